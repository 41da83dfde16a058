"""``ldm.data.video_swap_dataset`` -> reface_amd.data (scripts/inference_swap_selected.py:34, one_inference.py:34, inference_swap_video.py:34)."""
from reface_amd.data import VideoDataset  # noqa: F401
