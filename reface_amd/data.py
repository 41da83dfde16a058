"""Input contract of the test-bench datasets (ldm/data/test_bench_dataset.py:368): each item is
``(target[3,H,W] in [-1,1], prior, {inpaint_image, inpaint_mask, ref_imgs}, id_str)``.

``SyntheticPairs`` produces seeded items of that contract (no dataset is reachable offline); the real
CelebA / FFHQ / FF++ folder readers are a "next" row of the scope table."""
import torch
from torch.utils.data import Dataset

from .params import seeded_randn


class SyntheticPairs(Dataset):
    def __init__(self, n=8, image_size=512, seed=0, **_ignored):
        self.n, self.size, self.seed = n, image_size, seed
        H = image_size
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
        ell = (((yy - H / 2) / (0.30 * H)) ** 2 + ((xx - H / 2) / (0.38 * H)) ** 2) <= 1.0
        self.mask = (~ell).float()[None]                      # 1 = keep (test_bench_dataset.py:347)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        H = self.size
        target = torch.tanh(seeded_randn((3, H, H), self.seed * 100003 + 3 * i))
        ref = seeded_randn((1, 3, 224, 224), self.seed * 100003 + 3 * i + 1)      # CLIP-normalised reference (dataset adds a dim)
        inpaint = target * self.mask
        return target, target.clone(), {"inpaint_image": inpaint, "inpaint_mask": self.mask.clone(), "ref_imgs": ref}, f"{i:012d}"


def shard_indices(n, rank, world):
    """Pairs are independent: rank r takes indices r, r + world, ...."""
    return list(range(rank, n, world))
