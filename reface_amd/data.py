"""Input contract of the test-bench datasets (ldm/data/test_bench_dataset.py:368): each item is
``(target[3,H,W] in [-1,1], prior, {inpaint_image, inpaint_mask, ref_imgs}, id_str)``.

``SyntheticPairs`` produces seeded items of that contract (no dataset is reachable offline).  ``CelebAdataset`` reads the
CelebAMask-HQ folder layout the reference's test split uses (test_bench_dataset.py:130-222, 253-370), with PIL + numpy only.  The
reference resizes the source face to 224x224 with albumentations' ``A.Resize`` = ``cv2.resize(..., INTER_LINEAR)``; cv2 is a third-party
dependency that is neither under /root/reference nor in this image, so ``resize_u8_linear`` restates OpenCV's published uint8
INTER_LINEAR algorithm (two taps, 11-bit fixed point, no antialiasing) and the readers use it -- PIL's BILINEAR, which widens its
kernel when shrinking, is NOT equivalent (mean |d| of 37 grey levels on a 1024 -> 224 noise image).  Images are decoded by PIL (the
reference: ``cv2.imread``; both wrap libjpeg / libpng -- decode parity itself is unpinned)."""
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


INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _linear_taps(n_src, n_dst, clamp_high):
    """Source index and the two 11-bit fixed-point weights of every destination coordinate, as OpenCV's `resize` builds its xofs / alpha
    (yofs / beta) tables for INTER_LINEAR (modules/imgproc/src/resize.cpp, `cv::hal::resize`): half-pixel centres,
    f = float((d + 0.5) * scale - 0.5) with scale in double, s = floor(f), weights (1 - f, f) scaled by 2048 and rounded to nearest-even
    to int16 (`saturate_cast<short>`).  Columns (`clamp_high`): s < 0 -> (s, f) = (0, 0); s >= n_src - 1 -> (n_src - 1, 0).  Rows keep
    their weights and have the two source rows clipped to [0, n_src - 1] instead."""
    # OpenCV's operation order (resize.cpp): inv_scale = dsize / ssize, then scale = 1. / inv_scale -- TWO roundings in double; n_src / n_dst in one
    # rounding can differ by an ulp for non-power-of-two sizes and flip a floor() at a boundary
    scale = 1.0 / (float(n_dst) / float(n_src))
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_high:
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0.0, n_src - 1
    w1 = np.rint(f * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int32)                      # rint: round half to even (cvRound)
    w0 = np.rint((np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int32)
    return s, w0, w1


def resize_u8_linear(img, out_h, out_w):
    """`cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR)` for uint8 images [H, W] / [H, W, C] -- what albumentations'
    `A.Resize(height, width)` runs on the source face (ldm/data/test_bench_dataset.py:141-148, 296-324; scripts/inference_swap_selected.py:
    525-553).  cv2 is not in this image; this restates OpenCV's published u8 algorithm (resize.cpp: `resizeGeneric_` with
    `HResizeLinear<uchar, int, short, 2048>` and the u8 specialisation of `VResizeLinear`) in the same integer arithmetic and operation order
    (pinned by hand vectors and against the HIP kernel; NOT yet against a cv2-generated golden -- cv2 is absent here -- so the last bit is unpinned):

      * two taps per axis at half-pixel centres, NO antialiasing whatever the ratio (PIL's BILINEAR widens its kernel when shrinking);
      * horizontal pass in int32:  Hrow[x] = S[sx] * a0 + S[sx + 1] * a1  with 11-bit weights (a0 + a1 = 2048);
      * vertical pass:  out = ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2;
      * an exact 2:1 reduction on both axes is rerouted by OpenCV to its fast INTER_AREA kernel: (a + b + c + d + 2) >> 2.

    Not restated (builds of OpenCV that take them may differ in the last bit): the IPP / OpenCL back-ends."""
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim not in (2, 3):
        raise TypeError("resize_u8_linear: uint8 [H, W] or [H, W, C] image expected")
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    H, W, _ = a.shape
    if (H, W) == (out_h, out_w):
        out = a.copy()
    elif H == 2 * out_h and W == 2 * out_w:
        v = a.astype(np.int32)
        out = ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    else:
        sx, a0, a1 = _linear_taps(W, out_w, True)
        sy, b0, b1 = _linear_taps(H, out_h, False)
        sx1 = np.minimum(sx + 1, W - 1)                           # (weight 0 wherever sx + 1 would leave the row)
        y0, y1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
        a0_, a1_ = a0[None, :, None], a1[None, :, None]

        def hpass(rows):                                          # horizontal pass of the source rows a destination row needs: [out_h, out_w, C] int32
            r = a[rows]
            return r[:, sx, :].astype(np.int32) * a0_ + r[:, sx1, :].astype(np.int32) * a1_
        h0, h1 = hpass(y0) >> 4, hpass(y1) >> 4
        out = ((((b0[:, None, None] * h0) >> 16) + ((b1[:, None, None] * h1) >> 16) + 2) >> 2).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


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
        self.raw = "full" if raw == "full" else bool(raw)
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
            # raw = True: the 224x224 source is resized here (host); raw = "full": the decoded source goes to the GPU at its own size and
            # rf_resize_u8_linear does the same arithmetic there (DevicePrep; sources of one batch must then share a size, else `raw_collate`
            # keeps them as a list)
            ref = np.asarray(Image.open(self.ref_imgs[index]).convert("RGB"), dtype=np.uint8)
            if self.raw != "full":
                ref = resize_u8_linear(ref, 224, 224)
            u8 = lambda im: torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())
            return (u8(img_p), u8(Image.open(self.labels[index]).convert("L")), u8(ref),
                    u8(Image.open(self.ref_labels[index]).convert("L")), str(index).zfill(12))
        image_tensor = _normalize(_to_tensor(img_p), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        tar_labels = self.remove_tar if self.gray_outer_mask else self.preserve_src
        mask_tensor = 1.0 - _to_tensor(self._label_mask(self.labels[index], tar_labels))     # 1 = keep, 0 = region to generate
        # A.Resize(224, 224) = cv2.resize(..., INTER_LINEAR) on the cv2.imread image (test_bench_dataset.py:141-148, 296-324)
        ref = resize_u8_linear(np.asarray(Image.open(self.ref_imgs[index]).convert("RGB"), dtype=np.uint8), 224, 224)
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
    """The ONE source face of the selected-swap callers (inference_swap_selected.py:525-553): RGB image resized to 224x224 (A.Resize =
    cv2 INTER_LINEAR: `resize_u8_linear`), CLIP normalisation, times its preserved-label mask resized to 224x224 (bilinear on the
    tensor) -> [1, 3, 224, 224]."""
    from PIL import Image
    ref = resize_u8_linear(np.asarray(Image.open(img_path).convert("RGB"), dtype=np.uint8), 224, 224)
    lab = np.array(Image.open(mask_path).convert("L"))
    m = _to_tensor(Image.fromarray(np.where(np.isin(lab, preserve), 255, 0).astype(np.uint8)).convert("L"))
    m = torch.nn.functional.interpolate(m[None], size=(224, 224), mode="bilinear", align_corners=False)[0]
    return (_normalize(_to_tensor(ref), CLIP_MEAN, CLIP_STD) * m).unsqueeze(0)


def raw_collate(items):
    """Collate of raw (uint8) items: every field stacked as default_collate does, except that full-size sources / their label maps of
    different sizes stay lists (DevicePrep resizes them one by one)."""
    from torch.utils.data import default_collate
    cols = list(zip(*items))
    out = []
    for col in cols:
        if torch.is_tensor(col[0]) and any(c.shape != col[0].shape for c in col):
            out.append(list(col))
        else:
            out.append(default_collate(list(col)))
    return out


def shard_indices(n, rank, world):
    """Pairs are independent: rank r takes indices r, r + world, ...."""
    return list(range(rank, n, world))
