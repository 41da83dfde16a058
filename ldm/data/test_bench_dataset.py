"""``ldm.data.test_bench_dataset`` -> reface_amd.data (scripts/inference_test_bench.py:26-27).  CelebA and FFHQ test-split readers
exist; FF++ / COCO raise on construction."""
from reface_amd.data import CelebAdataset, FFHQdataset  # noqa: F401


class _Missing:
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__}: folder reader not built yet (SURVEY.md 8f.1); CelebAdataset / FFHQdataset are")


class FFdataset(_Missing):
    pass


class COCOImageDataset(_Missing):
    pass
