"""``ldm.data.test_bench_dataset`` -> reface_amd.data (scripts/inference_test_bench.py:26-27).  Only the CelebA reader of the test
bench exists so far; FFHQ / FF++ raise on construction."""
from reface_amd.data import CelebAdataset  # noqa: F401


class _Missing:
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__}: folder reader not built yet (SURVEY.md 8f.1); CelebAdataset is")


class FFHQdataset(_Missing):
    pass


class FFdataset(_Missing):
    pass


class COCOImageDataset(_Missing):
    pass
