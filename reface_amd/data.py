"""Input contract of the test-bench datasets (ldm/data/test_bench_dataset.py:368): each item is
``(target[3,H,W] in [-1,1], prior, {inpaint_image, inpaint_mask, ref_imgs}, id_str)``.

``SyntheticPairs`` produces seeded items of that contract (no dataset is reachable offline).  ``CelebAdataset`` reads the
CelebAMask-HQ folder layout the reference's test split uses (test_bench_dataset.py:130-222, 253-370), with PIL + numpy only:
the reference goes through cv2 / albumentations for the 224x224 resize of the source face (cv2 INTER_LINEAR), which are not in
this image -- PIL's bilinear filter stands in, so that one resize is "parity unpinned" (the tensors' shapes, ranges, mask
semantics and normalisations are the reference's)."""
import os

import numpy as np
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


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _to_tensor(img):
    """torchvision ToTensor: HWC uint8 -> CHW float in [0, 1] (an 'L' image gives one channel)."""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).float() / 255.0


def _normalize(t, mean, std):
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return (t - m) / s


class CelebAdataset(Dataset):
    """Test split of CelebAMask-HQ as the REFace test bench reads it (ids 28000-28999 are targets, 29000-29999 sources; pair i is
    (target i, source i)).  Item = (target [3,512,512] in [-1,1], prior, {inpaint_image, inpaint_mask (1 = keep), ref_imgs
    [1,3,224,224] CLIP-normalised source face x its face mask}, 12-digit id) -- test_bench_dataset.py:262-370.

    ``remove_mask_tar`` / ``preserve_mask_src`` are the segmentation labels (1 skin, 2 nose, ... 17 neck) that are cut out of the
    target / kept in the source; ``gray_outer_mask=False`` selects the `__getitem_black__` variant (full source image, one label
    list ``preserve_mask_src`` for both)."""
    IMG_DIR, IMG_FMT = "CelebA-HQ-img", "{}.jpg"
    MASK_DIR, MASK_FMT = "CelebA-HQ-mask/Overall_mask", "{}.png"
    REF_IMG_DIR = REF_MASK_DIR = None          # sources in other folders than the targets (FF++)

    def __init__(self, state="test", dataset_dir="dataset/FaceData/CelebAMask-HQ", gray_outer_mask=True, remove_mask_tar=None,
                 preserve_mask_src=None, preserve_mask=None, fraction=1.0, first_target=28000, n_targets=1000, first_source=29000,
                 raw=False, **_ignored):
        if state != "test":
            raise NotImplementedError("only the test split is on the inference path (train / validation splits feed main.py)")
        from PIL import Image
        self.Image = Image
        self.gray_outer_mask = bool(gray_outer_mask)
        # raw=True: items are the decoded / resized uint8 arrays only (target HWC, its label map, 224x224 source HWC, its label map, id);
        # normalisation, label masks and the mask products then run on the GPU (reface_amd/prep.py)
        self.raw = bool(raw)
        if preserve_mask is not None:
            remove_mask_tar = preserve_mask_src = preserve_mask
        self.remove_tar = list(remove_mask_tar if remove_mask_tar is not None else [1, 2, 4, 5, 8, 9, 6, 7, 10, 11, 12, 17])
        self.preserve_src = list(preserve_mask_src if preserve_mask_src is not None else [1, 2, 4, 5, 8, 9, 6, 7, 10, 11, 12, 13, 17])
        j = os.path.join
        ids_t = range(first_target, first_target + n_targets)
        ids_s = range(first_source, first_source + n_targets)
        self.imgs = sorted(j(dataset_dir, self.IMG_DIR, self.IMG_FMT.format(i)) for i in ids_t)
        self.labels = sorted(j(dataset_dir, self.MASK_DIR, self.MASK_FMT.format(i)) for i in ids_t)
        self.ref_imgs = sorted(j(dataset_dir, self.REF_IMG_DIR or self.IMG_DIR, self.IMG_FMT.format(i)) for i in ids_s)
        self.ref_labels = sorted(j(dataset_dir, self.REF_MASK_DIR or self.MASK_DIR, self.MASK_FMT.format(i)) for i in ids_s)
        n = int(len(self.imgs) * fraction)
        self.imgs, self.labels, self.ref_imgs, self.ref_labels = self.imgs[:n], self.labels[:n], self.ref_imgs[:n], self.ref_labels[:n]

    def __len__(self):
        return len(self.imgs)

    def _label_mask(self, path, keep):
        lab = np.array(self.Image.open(path).convert("L"))
        return self.Image.fromarray(np.where(np.isin(lab, keep), 255, 0).astype(np.uint8)).convert("L")

    def __getitem__(self, index):
        Image = self.Image
        img_p = Image.open(self.imgs[index]).convert("RGB").resize((512, 512))                 # PIL default filter, as the reference
        if self.raw:
            ref = Image.open(self.ref_imgs[index]).convert("RGB").resize((224, 224), Image.BILINEAR)
            u8 = lambda im: torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())
            return (u8(img_p), u8(Image.open(self.labels[index]).convert("L")), u8(ref),
                    u8(Image.open(self.ref_labels[index]).convert("L")), str(index).zfill(12))
        image_tensor = _normalize(_to_tensor(img_p), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        tar_labels = self.remove_tar if self.gray_outer_mask else self.preserve_src
        mask_tensor = 1.0 - _to_tensor(self._label_mask(self.labels[index], tar_labels))     # 1 = keep, 0 = region to generate
        ref = Image.open(self.ref_imgs[index]).convert("RGB").resize((224, 224), Image.BILINEAR)   # A.Resize(224, 224): cv2 linear
        ref_tensor = _normalize(_to_tensor(ref), CLIP_MEAN, CLIP_STD)
        if self.gray_outer_mask:
            ref_mask = _to_tensor(self._label_mask(self.ref_labels[index], self.preserve_src))
            # T.Resize((224, 224)) on a tensor: bilinear, no antialias (torchvision <= 0.14)
            ref_mask = torch.nn.functional.interpolate(ref_mask[None], size=(224, 224), mode="bilinear", align_corners=False)[0]
            ref_tensor = ref_tensor * ref_mask
        inpaint_tensor = image_tensor * mask_tensor
        return image_tensor, image_tensor, {"inpaint_image": inpaint_tensor, "inpaint_mask": mask_tensor,
                                            "ref_imgs": ref_tensor.unsqueeze(0)}, str(index).zfill(12)


class FFHQdataset(CelebAdataset):
    """FFHQ test split (test_bench_dataset.py:456-700): ``images512/%05d.png`` + ``BiSeNet_mask/%05d.png``, targets 68000-68999,
    sources 69000-69999, label lists ``remove_mask_tar_FFHQ`` / ``preserve_mask_src_FFHQ``; same item contract as CelebA."""
    IMG_DIR, IMG_FMT = "images512", "{:05d}.png"
    MASK_DIR, MASK_FMT = "BiSeNet_mask", "{:05d}.png"

    def __init__(self, state="test", dataset_dir="dataset/FaceData/FFHQ", remove_mask_tar_FFHQ=None, preserve_mask_src_FFHQ=None,
                 first_target=68000, first_source=69000, **kw):
        kw.pop("remove_mask_tar", None)
        kw.pop("preserve_mask_src", None)
        super().__init__(state=state, dataset_dir=dataset_dir, remove_mask_tar=remove_mask_tar_FFHQ, preserve_mask_src=preserve_mask_src_FFHQ,
                         first_target=first_target, first_source=first_source, **kw)


class FFdataset(FFHQdataset):
    """FaceForensics++ test split (test_bench_dataset.py:640-839): targets ``Val_target/%04d.png`` + ``target_mask`` (0-499),
    sources ``Val/%04d.png`` + ``src_mask`` (500-999), the *_FFHQ label lists; same item contract."""
    IMG_DIR, IMG_FMT = "Val_target", "{:04d}.png"
    MASK_DIR, MASK_FMT = "target_mask", "{:04d}.png"
    REF_IMG_DIR, REF_MASK_DIR = "Val", "src_mask"

    def __init__(self, state="test", dataset_dir="dataset/FaceData/FF++", first_target=0, first_source=500, n_targets=500, **kw):
        super().__init__(state=state, dataset_dir=dataset_dir, first_target=first_target, first_source=first_source, n_targets=n_targets, **kw)


class VideoDataset(Dataset):
    """Targets of the selected-swap / one-image / video callers (ldm/data/video_swap_dataset.py:86-295): ``data_path/<i>.png`` aligned
    target crops and ``mask_path/<i>.png`` their face-parsing label maps, i = 0 .. n-1.  Item = (target [3,512,512] in [-1,1], prior =
    target, {inpaint_image, inpaint_mask (1 = keep)}, 12-digit index) -- no ``ref_imgs``: the caller supplies ONE source face for
    every target.  ``gray_outer_mask=False`` uses the fixed label list [2, 3, 5, 6, 7] of ``__getitem_black__``."""

    def __init__(self, label_transform=None, data_path="Video_processing/target_frames", mask_path="Video_processing/target_masks", **args):
        from PIL import Image
        self.Image = Image
        self.gray_outer_mask = bool(args["gray_outer_mask"])
        # (the reference tests ``hasattr(args, 'preserve_mask')`` on a dict, which is always False: the *_FFHQ lists are what it uses)
        self.remove_tar = list(args["remove_mask_tar_FFHQ"])
        self.preserve_src = list(args["preserve_mask_src_FFHQ"])
        n_img, n_lab = len(os.listdir(data_path)), len(os.listdir(mask_path))
        assert n_img == n_lab, "The number of images must be equal to the number of labels"
        self.imgs = [os.path.join(data_path, f"{i}.png") for i in range(n_img)]
        self.labels = [os.path.join(mask_path, f"{i}.png") for i in range(n_lab)]

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, index):
        Image = self.Image
        img_p = Image.open(self.imgs[index]).convert("RGB").resize((512, 512))
        lab = np.array(Image.open(self.labels[index]).convert("L"))
        keep = self.remove_tar if self.gray_outer_mask else [2, 3, 5, 6, 7]
        mask_img = Image.fromarray(np.where(np.isin(lab, keep), 255, 0).astype(np.uint8)).convert("L")
        image_tensor = _normalize(_to_tensor(img_p), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        mask_tensor = 1.0 - _to_tensor(mask_img)
        return image_tensor, image_tensor, {"inpaint_image": image_tensor * mask_tensor, "inpaint_mask": mask_tensor}, str(index).zfill(12)


def load_source_reference(img_path, mask_path, preserve):
    """The ONE source face of the selected-swap callers (inference_swap_selected.py:525-553): RGB image resized to 224x224 (the reference
    uses A.Resize = cv2 INTER_LINEAR; PIL bilinear stands in, as in the test-bench readers: parity unpinned for that resize), CLIP
    normalisation, times its preserved-label mask resized to 224x224 (bilinear on the tensor) -> [1, 3, 224, 224]."""
    from PIL import Image
    ref = Image.open(img_path).convert("RGB").resize((224, 224), Image.BILINEAR)
    lab = np.array(Image.open(mask_path).convert("L"))
    m = _to_tensor(Image.fromarray(np.where(np.isin(lab, preserve), 255, 0).astype(np.uint8)).convert("L"))
    m = torch.nn.functional.interpolate(m[None], size=(224, 224), mode="bilinear", align_corners=False)[0]
    return (_normalize(_to_tensor(ref), CLIP_MEAN, CLIP_STD) * m).unsqueeze(0)


def shard_indices(n, rank, world):
    """Pairs are independent: rank r takes indices r, r + world, ...."""
    return list(range(rank, n, world))
