"""``ldm.data.test_bench_dataset`` -> reface_amd.data (scripts/inference_test_bench.py:26-27).  CelebA and FFHQ test-split readers
exist for CelebA, FFHQ and FF++; the COCO test bench raises on construction."""
from reface_amd.data import CelebAdataset, FFdataset, FFHQdataset  # noqa: F401


class _Missing:
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__}: folder reader not built yet (SURVEY.md 8f.1); the CelebA / FFHQ / FF++ readers are")


class COCOImageDataset(_Missing):
    pass
