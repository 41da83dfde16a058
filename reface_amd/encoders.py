"""Conditioning encoders of REFace on the HIP kernels: CLIP ViT-L/14 vision tower + mapper
(``FrozenCLIPEmbedder``) and ArcFace IR-SE50 (``Backbone`` / ``IDLoss``).

Interfaces mirror ldm/modules/encoders/modules.py:211-264 (+ xf.py:31-130) and
src/Face_models/encoders/model_irse.py:9-69, helpers.py:56-119, ldm/models/diffusion/ddpm.py:91-124.
The CLIP vision arithmetic itself lives in HF ``transformers`` (reference pin 4.19.2) and is
restated from the published ViT algorithm: patch conv (no bias) -> [CLS | patches] + position
embedding -> pre-LN -> L x {LN, MHA, +, LN, MLP(quick_gelu), +} -> post-LN(CLS) -> visual_projection.

Engines run in fp32 by default (conditioning is computed once per image pair; ~1.9 % of the FLOPs).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .modules import ParamTree, flat_state, weights_version
from .params import CLIPVisionConfig, arcface_param_specs, arcface_units, clip_param_specs
from .unet import _Pool

F32 = torch.float32
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bn_affine(sd, p, eps=1e-5):
    """eval-mode BatchNorm as y = x * a + b."""
    a = 1.0 / torch.sqrt(sd[f"{p}.running_var"] + eps)
    if f"{p}.weight" in sd:
        a = a * sd[f"{p}.weight"]
    b = -sd[f"{p}.running_mean"] * a
    if f"{p}.bias" in sd:
        b = b + sd[f"{p}.bias"]
    return a.contiguous(), b.contiguous()


# =================================================================================================
# ArcFace IR-SE50
# =================================================================================================
class _ArcFaceEngine:
    CP = 8      # 3 input channels stored in 8

    def __init__(self, sd, B, dtype, device):
        self.B, self.dt, self.dev = B, dtype, device
        self.pool = _Pool(device)
        self.sd = {k: v.detach().to(device=device, dtype=F32) for k, v in sd.items() if v.dtype.is_floating_point}
        self.launches = []
        self.ws = ops.new_workspace(device)
        with ops.workspace_scope(self.ws):
            self._build()
        self.sd = None

    def _conv(self, x, wkey, cout, *, stride=1, ksize=3, bn=None, prelu=None, cin_pad=None):
        """conv (no bias) with an optional following BatchNorm folded into weights/bias and optional PReLU epilogue."""
        B, H, W, _ = x.shape
        w = self.sd[wkey]
        bias = None
        if bn is not None:
            a, b = _bn_affine(self.sd, bn)
            w = w * a.view(-1, 1, 1, 1)
            bias = b
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        y = self.pool.get((B, Ho, Wo, cout), self.dt)
        pad = (1, 1) if ksize == 3 else (0, 0)
        self.launches.append(ops.conv2d(x, ops.pack_conv_weight(w, self.dt, cin_pad=cin_pad), y, bias, ksize=ksize, stride=stride, pad=pad,
                                        act=ops.ACT_PRELU if prelu is not None else ops.ACT_NONE,
                                        act_vec=self.sd[prelu].contiguous() if prelu is not None else None, name=wkey))
        return y

    def _build(self):
        B, dev, sd = self.B, self.dev, self.sd
        # ddpm.py:112-121: un_norm_clip -> (x - 0.5) / 0.5 -> pool 256 -> crop [35:223, 32:220] -> pool 112
        self.x_in = torch.empty((B, 3, 224, 224), dtype=F32, device=dev)
        a = torch.tensor([2.0 * s for s in CLIP_STD], dtype=F32, device=dev)             # ((x*std+mean) - 0.5) / 0.5
        b = torch.tensor([2.0 * m - 1.0 for m in CLIP_MEAN], dtype=F32, device=dev)
        p256 = torch.empty((B, 3, 256, 256), dtype=F32, device=dev)
        self.launches.append(ops.adaptive_avgpool(self.x_in, p256, a=a, b=b, name="face_pool_1"))
        x = torch.zeros((B, 112, 112, self.CP), dtype=self.dt, device=dev)
        self.launches.append(ops.adaptive_avgpool(p256, x, crop=(35, 32, 188, 188), nhwc=True, name="face_pool_2"))
        self.x112 = x
        self.body_from = len(self.launches)
        x = self._conv(x, "input_layer.0.weight", 64, bn="input_layer.1", prelu="input_layer.2.weight", cin_pad=self.CP)
        for i, (cin, depth, stride) in enumerate(arcface_units()):
            p = f"body.{i}"
            Bn, H, W, _ = x.shape
            if cin == depth:
                sc, sc_stride = x, stride                                    # MaxPool2d(1, stride) = subsample (helpers.py:101)
            else:
                sc, sc_stride = self._conv(x, f"{p}.shortcut_layer.0.weight", depth, stride=stride, ksize=1, bn=f"{p}.shortcut_layer.1"), 1
            a0, b0 = _bn_affine(sd, f"{p}.res_layer.0")
            r = self.pool.get((Bn, H, W, cin), self.dt)
            self.launches.append(ops.channel_affine(x, a0, b0, r, name=f"{p}.res_layer.0"))
            r1 = self._conv(r, f"{p}.res_layer.1.weight", depth, prelu=f"{p}.res_layer.2.weight")
            self.pool.put(r)
            r2 = self._conv(r1, f"{p}.res_layer.3.weight", depth, stride=stride, bn=f"{p}.res_layer.4")
            self.pool.put(r1)
            # SE (helpers.py:56-72): mean -> fc1 -> ReLU -> fc2 -> sigmoid -> scale
            m = torch.empty((Bn, depth), dtype=F32, device=dev)
            self.launches.append(ops.spatial_mean(r2, m, name=f"{p}.se.pool"))
            h = torch.empty((Bn, depth // 16), dtype=F32, device=dev)
            s = torch.empty((Bn, depth), dtype=F32, device=dev)
            self.launches.append(ops.linear(m, sd[f"{p}.res_layer.5.fc1.weight"].reshape(depth // 16, depth).contiguous(), h, None,
                                            act=ops.ACT_RELU, name=f"{p}.se.fc1"))
            self.launches.append(ops.linear(h, sd[f"{p}.res_layer.5.fc2.weight"].reshape(depth, depth // 16).contiguous(), s, None,
                                            act=ops.ACT_SIGMOID, name=f"{p}.se.fc2"))
            y = self.pool.get(tuple(r2.shape), self.dt)
            self.launches.append(ops.se_scale_add(r2, s, sc, y, stride=sc_stride, name=f"{p}.se.scale_add"))
            self.pool.put(r2)
            if sc is not x:
                self.pool.put(sc)
            self.pool.put(x)
            x = y
        # output_layer: BN2d -> flatten (CHW order) -> Linear -> BN1d -> l2norm   (model_irse.py:24-28, 69)
        a, b = _bn_affine(sd, "output_layer.0")
        xf = torch.empty(tuple(x.shape), dtype=F32, device=dev)
        self.launches.append(ops.channel_affine(x, a, b, xf, name="output_layer.0"))
        Bn, H, W, Cc = x.shape
        wl = sd["output_layer.3.weight"].reshape(512, Cc, H * W).permute(0, 2, 1).reshape(512, H * W * Cc).contiguous()   # CHW -> HWC columns
        feat = torch.empty((Bn, 512), dtype=F32, device=dev)
        self.launches.append(ops.linear(xf.view(Bn, H * W * Cc), wl, feat, sd["output_layer.3.bias"].contiguous(), name="output_layer.3"))
        a, b = _bn_affine(sd, "output_layer.4")
        f2 = torch.empty((Bn, 512), dtype=F32, device=dev)
        self.launches.append(ops.channel_affine(feat, a, b, f2, name="output_layer.4"))
        self.out = torch.empty((Bn, 512), dtype=F32, device=dev)
        self.launches.append(ops.l2norm_rows(f2, self.out))


class Backbone(nn.Module):
    """Drop-in for ``Backbone(input_size=112, num_layers=50, mode='ir_se')`` (model_irse.py:9-69), eval mode."""

    def __init__(self, input_size=112, num_layers=50, mode="ir_se", drop_ratio=0.4, affine=True, compute_dtype=None):
        super().__init__()
        assert input_size == 112 and num_layers == 50 and mode == "ir_se" and affine, "only the REFace ArcFace configuration"
        self.compute_dtype = compute_dtype or F32
        tree = ParamTree(arcface_param_specs())
        for name, child in tree.named_children():
            self.add_module(name, child)
        self._engines = {}

    def _engine(self, B):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("reface_amd ArcFace runs on the GPU only (HIP kernels; there is no CPU fallback)")
        key = (B, self.compute_dtype, weights_version(self))
        eng = self._engines.get(key)
        if eng is None:
            self._engines = {k: v for k, v in self._engines.items() if k[-1] == key[-1]}
            eng = _ArcFaceEngine(flat_state(self), B, self.compute_dtype, dev)
            self._engines[key] = eng
        return eng

    @torch.no_grad()
    def forward(self, x, multi_scale=False):
        """x: [B, 3, 112, 112] fp32 NCHW -> [l2-normalised 512-d features]."""
        assert not multi_scale, "multi_scale_ID is False in the shipped REFace configuration"
        eng = self._engine(x.shape[0])
        ops.nchw_to_nhwc(x.to(F32).contiguous(), eng.x112)()
        ops.run(eng.launches[eng.body_from:])
        return [eng.out.clone()]

    @torch.no_grad()
    def forward_from_clip_image(self, ref224):
        """The whole ``IDLoss.extract_feats`` chain (ddpm.py:112-124) on a CLIP-normalised 224x224 image."""
        eng = self._engine(ref224.shape[0])
        eng.x_in.copy_(ref224.to(F32))
        ops.run(eng.launches)
        return [eng.out.clone()]


class IDLoss(nn.Module):
    """ddpm.py:91-124 -- only ``extract_feats`` is on the inference path."""

    def __init__(self, opts=None, multiscale=False):
        super().__init__()
        self.multiscale = multiscale
        self.facenet = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se")

    def extract_feats(self, x, clip_img=True):
        if not clip_img or x.shape[2] != 224:
            raise NotImplementedError("extract_feats expects the CLIP-normalised 224x224 reference image (REFace path)")
        return self.facenet.forward_from_clip_image(x)


# =================================================================================================
# CLIP ViT vision tower + mapper2
# =================================================================================================
class _CLIPEngine:
    def __init__(self, sd, cfg: CLIPVisionConfig, B, dtype, device):
        self.cfg, self.B, self.dt, self.dev = cfg, B, dtype, device
        self.sd = {k: v.detach().to(device=device, dtype=F32) for k, v in sd.items()}
        self.launches = []
        self.ws = ops.new_workspace(device)
        with ops.workspace_scope(self.ws):
            self._build()
        self.sd = None

    def w(self, k):
        return self.sd[k].to(self.dt).contiguous()

    def f(self, k):
        return self.sd[k].contiguous()

    def _build(self):
        cfg, B, dev, dt = self.cfg, self.B, self.dev, self.dt
        h, heads, P = cfg.hidden, cfg.heads, cfg.patch
        g = cfg.image // P
        NP, NT = g * g, g * g + 1
        v = "model.vision_model"
        CP = 4 if dt == F32 else 8
        self.x_in = torch.empty((B, 3, cfg.image, cfg.image), dtype=F32, device=dev)
        xcl = torch.zeros((B, cfg.image, cfg.image, CP), dtype=dt, device=dev)
        self.launches.append(ops.nchw_to_nhwc(self.x_in, xcl))
        patch = torch.empty((B, g, g, h), dtype=dt, device=dev)
        self.launches.append(ops.conv2d(xcl, ops.pack_conv_weight(self.sd[f"{v}.embeddings.patch_embedding.weight"], dt, cin_pad=CP), patch,
                                        None, ksize=P, stride=P, pad=(0, 0), name="patch_embedding"))
        x = torch.empty((B, NT, h), dtype=dt, device=dev)
        self.launches.append(ops.clip_tokens(patch.view(B, NP, h), self.f(f"{v}.embeddings.class_embedding"),
                                             self.f(f"{v}.embeddings.position_embedding.weight"), x))
        M = B * NT
        x2 = x.view(M, h)
        xa = torch.empty((M, h), dtype=dt, device=dev)
        self.launches.append(ops.layernorm(x2, self.f(f"{v}.pre_layrnorm.weight"), self.f(f"{v}.pre_layrnorm.bias"), xa, name="pre_layrnorm"))
        cur, other = xa, x2
        ln = torch.empty((M, h), dtype=dt, device=dev)
        qkv = torch.empty((M, 3 * h), dtype=dt, device=dev)
        att = torch.empty((M, h), dtype=dt, device=dev)
        mid = torch.empty((M, cfg.intermediate), dtype=dt, device=dev)
        d = h // heads
        for i in range(cfg.layers):
            p = f"{v}.encoder.layers.{i}"
            self.launches.append(ops.layernorm(cur, self.f(f"{p}.layer_norm1.weight"), self.f(f"{p}.layer_norm1.bias"), ln, name=f"{p}.ln1"))
            wqkv = torch.cat([self.sd[f"{p}.self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0).to(dt).contiguous()
            bqkv = torch.cat([self.sd[f"{p}.self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous()
            self.launches.append(ops.linear(ln, wqkv, qkv, bqkv, name=f"{p}.qkv"))
            q3 = qkv.view(B, NT, 3 * h)
            self.launches.append(ops.attention(q3[..., :h], q3[..., h:2 * h], q3[..., 2 * h:], att.view(B, NT, h), heads=heads,
                                               scale=d ** -0.5, name=f"{p}.attn"))
            self.launches.append(ops.linear(att, self.w(f"{p}.self_attn.out_proj.weight"), other, self.f(f"{p}.self_attn.out_proj.bias"),
                                            residual=cur, name=f"{p}.out_proj"))
            cur, other = other, cur
            self.launches.append(ops.layernorm(cur, self.f(f"{p}.layer_norm2.weight"), self.f(f"{p}.layer_norm2.bias"), ln, name=f"{p}.ln2"))
            self.launches.append(ops.linear(ln, self.w(f"{p}.mlp.fc1.weight"), mid, self.f(f"{p}.mlp.fc1.bias"), act=ops.ACT_QUICK_GELU, name=f"{p}.fc1"))
            self.launches.append(ops.linear(mid, self.w(f"{p}.mlp.fc2.weight"), other, self.f(f"{p}.mlp.fc2.bias"), residual=cur, name=f"{p}.fc2"))
            cur, other = other, cur
        # pooled = post_layernorm(CLS) -> visual_projection -> mapper2 -> final_ln2   (fp32 from here: B rows)
        cls_rows = cur.view(B, NT, h)[:, 0, :]                      # [B, h] view with row pitch NT*h
        pooled = torch.empty((B, h), dtype=F32, device=dev)
        self.launches.append(ops.layernorm(cls_rows, self.f(f"{v}.post_layernorm.weight"), self.f(f"{v}.post_layernorm.bias"), pooled, name="post_layernorm"))
        self.pooled = pooled
        w = cfg.proj
        z = torch.empty((B, w), dtype=F32, device=dev)
        self.launches.append(ops.linear(pooled, self.f("model.visual_projection.weight"), z, None, name="visual_projection"))
        t1 = torch.empty((B, w), dtype=F32, device=dev)
        vbuf = torch.empty((B, w), dtype=F32, device=dev)
        z2 = torch.empty((B, w), dtype=F32, device=dev)
        hid = torch.empty((B, 4 * w), dtype=F32, device=dev)
        for i in range(cfg.mapper_layers):
            p = f"mapper2.resblocks.{i}"
            # xf.py:60-77 with n_ctx = 1, heads = 1: softmax over one key == 1  =>  attention(x) == v == c_qkv(x)[2w:3w]
            self.launches.append(ops.layernorm(z, self.f(f"{p}.ln_1.weight"), self.f(f"{p}.ln_1.bias"), t1, name=f"{p}.ln_1"))
            self.launches.append(ops.linear(t1, self.sd[f"{p}.attn.c_qkv.weight"][2 * w:].contiguous(), vbuf,
                                            self.sd[f"{p}.attn.c_qkv.bias"][2 * w:].contiguous(), name=f"{p}.c_qkv.v"))
            self.launches.append(ops.linear(vbuf, self.f(f"{p}.attn.c_proj.weight"), z2, self.f(f"{p}.attn.c_proj.bias"), residual=z, name=f"{p}.c_proj"))
            self.launches.append(ops.layernorm(z2, self.f(f"{p}.ln_2.weight"), self.f(f"{p}.ln_2.bias"), t1, name=f"{p}.ln_2"))
            self.launches.append(ops.linear(t1, self.f(f"{p}.mlp.c_fc.weight"), hid, self.f(f"{p}.mlp.c_fc.bias"), act=ops.ACT_GELU, name=f"{p}.c_fc"))
            self.launches.append(ops.linear(hid, self.f(f"{p}.mlp.c_proj.weight"), z, self.f(f"{p}.mlp.c_proj.bias"), residual=z2, name=f"{p}.mlp.c_proj"))
        self.out = torch.empty((B, w), dtype=F32, device=dev)
        self.launches.append(ops.layernorm(z, self.f("final_ln2.weight"), self.f("final_ln2.bias"), self.out, name="final_ln2"))


class FrozenCLIPEmbedder(nn.Module):
    """Drop-in for ``ldm.modules.encoders.modules.FrozenCLIPEmbedder`` (image branch, modules.py:253-264).

    Holds only the tensors the inference path reads (vision tower, visual_projection, mapper2, final_ln2); the
    checkpoint's text tower / mapper / final_ln / projection_back keys are ignored by ``load_state_dict(strict=False)``.
    No network access: weights come from the REFace checkpoint (``cond_stage_model.*``)."""

    def __init__(self, version="openai/clip-vit-large-patch14", vision_config=None, compute_dtype=None):
        super().__init__()
        self.cfg = CLIPVisionConfig(**(vision_config or {}))
        self.compute_dtype = compute_dtype or F32
        tree = ParamTree(clip_param_specs(self.cfg))
        for name, child in tree.named_children():
            self.add_module(name, child)
        self._engines = {}

    def _engine(self, B):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("reface_amd CLIP runs on the GPU only (HIP kernels; there is no CPU fallback)")
        key = (B, self.compute_dtype, weights_version(self))
        eng = self._engines.get(key)
        if eng is None:
            self._engines = {k: v for k, v in self._engines.items() if k[-1] == key[-1]}
            eng = _CLIPEngine(flat_state(self), self.cfg, B, self.compute_dtype, dev)
            self._engines[key] = eng
        return eng

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # newer HF layouts nest the tower one level deeper (model.vision_model.vision_model.*): accept both
        deep = prefix + "model.vision_model.vision_model."
        for k in [k for k in state_dict if k.startswith(deep)]:
            state_dict[prefix + "model.vision_model." + k[len(deep):]] = state_dict.pop(k)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @torch.no_grad()
    def forward(self, image):
        """image: [B, 3, 224, 224] CLIP-normalised fp32 -> [B, 1, 768]."""
        eng = self._engine(image.shape[0])
        eng.x_in.copy_(image.to(F32))
        ops.run(eng.launches)
        return eng.out.clone().unsqueeze(1)

    def encode(self, image):
        return self(image)


# =================================================================================================
# glue used by LatentDiffusion.conditioning_with_feat
# =================================================================================================
@torch.no_grad()
def target_to_clip_input(tar):
    """ddpm.py:907-912: (tar + 1) / 2 -> CLIP normalise -> bilinear resize to 224 (no antialias)."""
    B = tar.shape[0]
    dev = tar.device
    a = torch.tensor([0.5 / s for s in CLIP_STD], dtype=F32, device=dev)
    b = torch.tensor([(0.5 - m) / s for m, s in zip(CLIP_MEAN, CLIP_STD)], dtype=F32, device=dev)
    out = torch.empty((B, 3, 224, 224), dtype=F32, device=dev)
    ops.bilinear_resize(tar.to(F32).contiguous(), out, a=a, b=b)()
    return out


@torch.no_grad()
def combine_conditioning(c_src, c_tar, c_id, lm, clip_w, id_w, lm_w, weight_division=True):
    """ddpm.py:915 (c = c_src + c_tar) and :1038-1039 (weighted mean).  All [B, 1, 768] fp32 (c_id / lm optional)."""
    c = torch.empty_like(c_src)
    ops.combine3(c_src.contiguous(), c_tar.contiguous(), None, c, wa=1.0, wb=1.0, wc=0.0, den=0.0)()
    out = torch.empty_like(c)
    den = (clip_w + (id_w if c_id is not None else 0.0) + (lm_w if lm is not None else 0.0)) if weight_division else 0.0
    ops.combine3(c, None if c_id is None else c_id.contiguous(), None if lm is None else lm.contiguous(), out,
                 wa=clip_w, wb=id_w, wc=lm_w, den=den)()
    return out
