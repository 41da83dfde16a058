"""Device-side input preparation of the test-bench items (second half).

The reference's dataset builds every tensor on the host, per image (ldm/data/test_bench_dataset.py:283-355): ToTensor + Normalize of
the 512x512 target, ``1 - isin(labels, remove)`` keep-mask, masked target, ToTensor + CLIP-Normalize of the 224x224 source, its
``isin(labels, preserve)`` mask resized to 224x224 (bilinear), product.  With ``raw=True`` the dataset workers only decode / resize
and hand over uint8 arrays (``raw="full"``: the source face at its own size, resized here by rf_resize_u8_linear -- cv2 INTER_LINEAR bit for bit); this module does the rest on the GPU with the same float operation order, so the target, keep-mask
and masked target are bit-identical to the host path and the masked source face agrees to one fp32 ulp (its mask goes through a
non-integer-ratio bilinear resize; tests/test_e2e_gpu.py::test_device_prep_matches_host).
"""
import torch

from . import ops
from .data import CLIP_MEAN, CLIP_STD


class DevicePrep:
    def __init__(self, remove_tar, preserve_src, gray_outer_mask=True, device="cuda"):
        self.dev = torch.device(device)
        self.gray = bool(gray_outer_mask)

        def lut(labels):
            t = torch.zeros(256, dtype=torch.uint8)
            t[torch.tensor(list(labels), dtype=torch.long)] = 1
            return t.to(self.dev)
        self.lut_tar = lut(remove_tar if self.gray else preserve_src)      # __getitem_black__ uses preserve_src for the target too
        self.lut_src = lut(preserve_src)
        f = lambda v: torch.tensor(v, dtype=torch.float32, device=self.dev)
        self.m05, self.s05 = f((0.5, 0.5, 0.5)), f((0.5, 0.5, 0.5))
        self.mclip, self.sclip = f(CLIP_MEAN), f(CLIP_STD)

    @torch.no_grad()
    def resize_sources(self, ref_u8):
        """Sources as uint8 [B, 224, 224, 3] on the device: already 224x224 (host resize), or full-size images -- a stacked batch or a
        list of different sizes -- through rf_resize_u8_linear (cv2 INTER_LINEAR arithmetic, test_bench_dataset.py:141-148)."""
        dev = self.dev
        if torch.is_tensor(ref_u8):
            ref_u8 = ref_u8.to(dev, non_blocking=True).contiguous()
            if tuple(ref_u8.shape[1:3]) == (224, 224):
                return ref_u8
            out = torch.empty((ref_u8.shape[0], 224, 224, 3), dtype=torch.uint8, device=dev)
            ops.resize_u8_linear(ref_u8, out)()
            return out
        out = torch.empty((len(ref_u8), 224, 224, 3), dtype=torch.uint8, device=dev)
        for i, r in enumerate(ref_u8):
            r = r.to(dev, non_blocking=True).contiguous()
            if tuple(r.shape[:2]) == (224, 224):
                out[i].copy_(r)
            else:
                ops.resize_u8_linear(r[None], out[i:i + 1])()
        return out

    @torch.no_grad()
    def __call__(self, tar_u8, tar_lab, ref_u8, ref_lab):
        """uint8 batches (host or device): tar [B,H,W,3], tar_lab [B,H,W], ref [B,224,224,3], ref_lab [B,Hl,Wl] ->
        (target [B,3,H,W] in [-1,1], {"inpaint_image", "inpaint_mask" [B,1,H,W] (1 = keep), "ref_imgs" [B,1,3,224,224]})."""
        dev = self.dev
        tar_u8, tar_lab = (t.to(dev, non_blocking=True).contiguous() for t in (tar_u8, tar_lab))
        B, H, W, _ = tar_u8.shape
        ref_u8 = self.resize_sources(ref_u8)
        if isinstance(ref_lab, (list, tuple)):          # label maps of different sizes: one mask resize per image below
            ref_lab = [t.to(dev, non_blocking=True).contiguous() for t in ref_lab]
        else:
            ref_lab = ref_lab.to(dev, non_blocking=True).contiguous()
        target = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        ops.u8_to_norm(tar_u8, self.m05, self.s05, target)()
        mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        ops.label_mask(tar_lab, self.lut_tar, mask, invert=True)()
        inpaint = torch.empty_like(target)
        ops.mul_mask(target, mask, inpaint)()
        ref = torch.empty((B, 3, 224, 224), dtype=torch.float32, device=dev)
        ops.u8_to_norm(ref_u8, self.mclip, self.sclip, ref)()
        if self.gray:
            m224 = torch.empty((B, 1, 224, 224), dtype=torch.float32, device=dev)
            groups = [(ref_lab, m224)] if torch.is_tensor(ref_lab) else [(l[None], m224[i:i + 1]) for i, l in enumerate(ref_lab)]
            for lab, dst in groups:
                m_full = torch.empty((lab.shape[0], 1) + tuple(lab.shape[1:]), dtype=torch.float32, device=dev)
                ops.label_mask(lab, self.lut_src, m_full, invert=False)()
                ops.bilinear_resize(m_full, dst)()                   # T.Resize((224, 224)) on a tensor: bilinear, no antialias
            ops.mul_mask(ref, m224, ref)()
        return target, {"inpaint_image": inpaint, "inpaint_mask": mask, "ref_imgs": ref.unsqueeze(1)}
