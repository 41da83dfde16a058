"""Parameter specifications (reference checkpoint key layout) and seeded initialisation.

The key names and shapes below restate the ``state_dict`` layout of the reference modules so
that a REFace Lightning checkpoint loads unchanged:

* UNet      -- ldm/modules/diffusionmodules/openaimodel.py:631-836, ldm/modules/attention.py:152-276
* KL-VAE    -- ldm/modules/diffusionmodules/model.py:368-533, ldm/models/autoencoder.py:285-312
* CLIP      -- HF ``CLIPModel.vision_model`` names + ldm/modules/encoders/modules.py:211-233, xf.py:31-130
* ArcFace   -- src/Face_models/encoders/model_irse.py:9-43, helpers.py:29-119

``seeded_state_dict`` fills a spec with deterministic, variance-preserving random values (per-key
CPU generators, so the result does not depend on generation order).  There is no network in the
build/bench environment, so random weights of the right architecture are what the benchmark and
the parity tests run on; the same function produces the weights that the golden-vector
generator (tools/gen_golden.py) loads into the imported reference modules.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, List, Sequence, Tuple

import torch

Spec = "OrderedDict[str, Tuple[int, ...]]"


# --------------------------------------------------------------------------------------------
# UNet
# --------------------------------------------------------------------------------------------
class UNetConfig:
    """Constructor arguments of the reference UNetModel that shape the graph
    (openaimodel.py:558-589; values from configs/train.yaml:33-47)."""

    def __init__(self, in_channels=9, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), num_heads=8,
                 transformer_depth=1, context_dim=768, **_ignored):
        self.in_channels = int(in_channels)
        self.model_channels = int(model_channels)
        self.out_channels = int(out_channels)
        self.num_res_blocks = int(num_res_blocks)
        self.attention_resolutions = tuple(int(a) for a in attention_resolutions)
        self.channel_mult = tuple(int(m) for m in channel_mult)
        self.num_heads = int(num_heads)
        self.transformer_depth = int(transformer_depth)
        self.context_dim = int(context_dim)
        if self.transformer_depth != 1:
            raise NotImplementedError("transformer_depth != 1 is not on the REFace path")

    @property
    def time_embed_dim(self):
        return 4 * self.model_channels


def unet_plan(cfg: UNetConfig):
    """Block structure of the UNet as lists of layer descriptors.

    Mirrors the construction loops of openaimodel.py:666-830 (``legacy=False``,
    ``num_head_channels=-1`` => dim_head = ch // num_heads, ``resblock_updown=False``).
    Returns ``(input_blocks, middle_block, output_blocks)``; each block is a list of tuples
    ``("conv", cin, cout) | ("res", cin, cout) | ("st", ch, heads, dhead) | ("down", ch) | ("up", ch)``.
    """
    mc = cfg.model_channels
    input_blocks: List[List[tuple]] = [[("conv", cfg.in_channels, mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(("st", ch, cfg.num_heads, ch // cfg.num_heads))
            input_blocks.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            input_blocks.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch), ("st", ch, cfg.num_heads, ch // cfg.num_heads), ("res", ch, ch)]
    output_blocks: List[List[tuple]] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(("st", ch, cfg.num_heads, ch // cfg.num_heads))
            if level and i == cfg.num_res_blocks:
                layers.append(("up", ch))
                ds //= 2
            output_blocks.append(layers)
    return input_blocks, middle, output_blocks


def _res_specs(s, p, cin, cout, tdim):
    s[f"{p}.in_layers.0.weight"] = (cin,)
    s[f"{p}.in_layers.0.bias"] = (cin,)
    s[f"{p}.in_layers.2.weight"] = (cout, cin, 3, 3)
    s[f"{p}.in_layers.2.bias"] = (cout,)
    s[f"{p}.emb_layers.1.weight"] = (cout, tdim)
    s[f"{p}.emb_layers.1.bias"] = (cout,)
    s[f"{p}.out_layers.0.weight"] = (cout,)
    s[f"{p}.out_layers.0.bias"] = (cout,)
    s[f"{p}.out_layers.3.weight"] = (cout, cout, 3, 3)
    s[f"{p}.out_layers.3.bias"] = (cout,)
    if cin != cout:
        s[f"{p}.skip_connection.weight"] = (cout, cin, 1, 1)
        s[f"{p}.skip_connection.bias"] = (cout,)


def _st_specs(s, p, c, ctx):
    s[f"{p}.norm.weight"] = (c,)
    s[f"{p}.norm.bias"] = (c,)
    s[f"{p}.proj_in.weight"] = (c, c, 1, 1)
    s[f"{p}.proj_in.bias"] = (c,)
    t = f"{p}.transformer_blocks.0"
    for a, kdim in (("attn1", c), ("attn2", ctx)):
        s[f"{t}.{a}.to_q.weight"] = (c, c)
        s[f"{t}.{a}.to_k.weight"] = (c, kdim)
        s[f"{t}.{a}.to_v.weight"] = (c, kdim)
        s[f"{t}.{a}.to_out.0.weight"] = (c, c)
        s[f"{t}.{a}.to_out.0.bias"] = (c,)
    s[f"{t}.ff.net.0.proj.weight"] = (8 * c, c)
    s[f"{t}.ff.net.0.proj.bias"] = (8 * c,)
    s[f"{t}.ff.net.2.weight"] = (c, 4 * c)
    s[f"{t}.ff.net.2.bias"] = (c,)
    for n in ("norm1", "norm2", "norm3"):
        s[f"{t}.{n}.weight"] = (c,)
        s[f"{t}.{n}.bias"] = (c,)
    s[f"{p}.proj_out.weight"] = (c, c, 1, 1)
    s[f"{p}.proj_out.bias"] = (c,)


def _block_specs(s, prefix, layers, tdim, ctx):
    for j, l in enumerate(layers):
        p = f"{prefix}.{j}"
        if l[0] == "conv":
            s[f"{p}.weight"] = (l[2], l[1], 3, 3)
            s[f"{p}.bias"] = (l[2],)
        elif l[0] == "res":
            _res_specs(s, p, l[1], l[2], tdim)
        elif l[0] == "st":
            _st_specs(s, p, l[1], ctx)
        elif l[0] == "down":
            s[f"{p}.op.weight"] = (l[1], l[1], 3, 3)
            s[f"{p}.op.bias"] = (l[1],)
        elif l[0] == "up":
            s[f"{p}.conv.weight"] = (l[1], l[1], 3, 3)
            s[f"{p}.conv.bias"] = (l[1],)


def unet_param_specs(cfg: UNetConfig):
    """Keys relative to ``model.diffusion_model.``."""
    s = OrderedDict()
    mc, td = cfg.model_channels, cfg.time_embed_dim
    s["time_embed.0.weight"] = (td, mc)
    s["time_embed.0.bias"] = (td,)
    s["time_embed.2.weight"] = (td, td)
    s["time_embed.2.bias"] = (td,)
    ib, mid, ob = unet_plan(cfg)
    for i, layers in enumerate(ib):
        _block_specs(s, f"input_blocks.{i}", layers, td, cfg.context_dim)
    _block_specs(s, "middle_block", mid, td, cfg.context_dim)
    for i, layers in enumerate(ob):
        _block_specs(s, f"output_blocks.{i}", layers, td, cfg.context_dim)
    s["out.0.weight"] = (mc,)
    s["out.0.bias"] = (mc,)
    s["out.2.weight"] = (cfg.out_channels, mc, 3, 3)
    s["out.2.bias"] = (cfg.out_channels,)
    return s


# --------------------------------------------------------------------------------------------
# KL-VAE (AutoencoderKL)
# --------------------------------------------------------------------------------------------
class VAEConfig:
    """``ddconfig`` + ``embed_dim`` of the reference AutoencoderKL (configs/train.yaml:51-68)."""

    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, out_ch=3,
                 z_channels=4, embed_dim=4, double_z=True, attn_resolutions=(), resolution=256,
                 dropout=0.0, **_ignored):
        self.ch = int(ch)
        self.ch_mult = tuple(int(m) for m in ch_mult)
        self.num_res_blocks = int(num_res_blocks)
        self.in_channels = int(in_channels)
        self.out_ch = int(out_ch)
        self.z_channels = int(z_channels)
        self.embed_dim = int(embed_dim)
        self.double_z = bool(double_z)
        if len(tuple(attn_resolutions)) != 0:
            raise NotImplementedError("attn_resolutions != [] is not on the REFace path")


def _vae_res(s, p, cin, cout):
    s[f"{p}.norm1.weight"] = (cin,)
    s[f"{p}.norm1.bias"] = (cin,)
    s[f"{p}.conv1.weight"] = (cout, cin, 3, 3)
    s[f"{p}.conv1.bias"] = (cout,)
    s[f"{p}.norm2.weight"] = (cout,)
    s[f"{p}.norm2.bias"] = (cout,)
    s[f"{p}.conv2.weight"] = (cout, cout, 3, 3)
    s[f"{p}.conv2.bias"] = (cout,)
    if cin != cout:
        s[f"{p}.nin_shortcut.weight"] = (cout, cin, 1, 1)
        s[f"{p}.nin_shortcut.bias"] = (cout,)


def _vae_attn(s, p, c):
    s[f"{p}.norm.weight"] = (c,)
    s[f"{p}.norm.bias"] = (c,)
    for n in ("q", "k", "v", "proj_out"):
        s[f"{p}.{n}.weight"] = (c, c, 1, 1)
        s[f"{p}.{n}.bias"] = (c,)


def _vae_mid(s, p, c):
    _vae_res(s, f"{p}.block_1", c, c)
    _vae_attn(s, f"{p}.attn_1", c)
    _vae_res(s, f"{p}.block_2", c, c)


def vae_encoder_specs(cfg: VAEConfig):
    """Keys relative to ``first_stage_model.`` (model.py:368-431)."""
    s = OrderedDict()
    ch = cfg.ch
    s["encoder.conv_in.weight"] = (ch, cfg.in_channels, 3, 3)
    s["encoder.conv_in.bias"] = (ch,)
    in_mult = (1,) + cfg.ch_mult
    block_in = ch
    nres = len(cfg.ch_mult)
    for lvl in range(nres):
        block_in = ch * in_mult[lvl]
        block_out = ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks):
            _vae_res(s, f"encoder.down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != nres - 1:
            s[f"encoder.down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            s[f"encoder.down.{lvl}.downsample.conv.bias"] = (block_in,)
    _vae_mid(s, "encoder.mid", block_in)
    s["encoder.norm_out.weight"] = (block_in,)
    s["encoder.norm_out.bias"] = (block_in,)
    zc = 2 * cfg.z_channels if cfg.double_z else cfg.z_channels
    s["encoder.conv_out.weight"] = (zc, block_in, 3, 3)
    s["encoder.conv_out.bias"] = (zc,)
    s["quant_conv.weight"] = (2 * cfg.embed_dim, zc, 1, 1)
    s["quant_conv.bias"] = (2 * cfg.embed_dim,)
    return s


def vae_decoder_specs(cfg: VAEConfig):
    """Keys relative to ``first_stage_model.`` (model.py:462-533); ``up`` is indexed by level,
    ``up.{nres-1}`` runs first (model.py:506-525)."""
    s = OrderedDict()
    ch = cfg.ch
    nres = len(cfg.ch_mult)
    block_in = ch * cfg.ch_mult[nres - 1]
    s["post_quant_conv.weight"] = (cfg.z_channels, cfg.embed_dim, 1, 1)
    s["post_quant_conv.bias"] = (cfg.z_channels,)
    s["decoder.conv_in.weight"] = (block_in, cfg.z_channels, 3, 3)
    s["decoder.conv_in.bias"] = (block_in,)
    _vae_mid(s, "decoder.mid", block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks + 1):
            _vae_res(s, f"decoder.up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            s[f"decoder.up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            s[f"decoder.up.{lvl}.upsample.conv.bias"] = (block_in,)
    s["decoder.norm_out.weight"] = (block_in,)
    s["decoder.norm_out.bias"] = (block_in,)
    s["decoder.conv_out.weight"] = (cfg.out_ch, block_in, 3, 3)
    s["decoder.conv_out.bias"] = (cfg.out_ch,)
    return s


def vae_param_specs(cfg: VAEConfig):
    s = vae_encoder_specs(cfg)
    s.update(vae_decoder_specs(cfg))
    return s


# --------------------------------------------------------------------------------------------
# CLIP ViT vision tower + REFace mapper  (FrozenCLIPEmbedder)
# --------------------------------------------------------------------------------------------
class CLIPVisionConfig:
    """openai/clip-vit-large-patch14 vision tower dims (HF configuration_clip.py) and the
    REFace mapper (modules.py:226-231: Transformer(n_ctx=1, width=768, layers=5, heads=1))."""

    def __init__(self, hidden=1024, intermediate=4096, layers=24, heads=16, patch=14, image=224,
                 proj=768, mapper_layers=5):
        self.hidden, self.intermediate, self.layers, self.heads = hidden, intermediate, layers, heads
        self.patch, self.image, self.proj, self.mapper_layers = patch, image, proj, mapper_layers

    @property
    def tokens(self):
        return (self.image // self.patch) ** 2 + 1


def clip_param_specs(cfg: CLIPVisionConfig):
    """Keys relative to ``cond_stage_model.`` — only the tensors the inference path reads
    (vision tower, visual_projection, mapper2, final_ln2)."""
    s = OrderedDict()
    h = cfg.hidden
    v = "model.vision_model"
    s[f"{v}.embeddings.class_embedding"] = (h,)
    s[f"{v}.embeddings.patch_embedding.weight"] = (h, 3, cfg.patch, cfg.patch)
    s[f"{v}.embeddings.position_embedding.weight"] = (cfg.tokens, h)
    s[f"{v}.pre_layrnorm.weight"] = (h,)
    s[f"{v}.pre_layrnorm.bias"] = (h,)
    for i in range(cfg.layers):
        p = f"{v}.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[f"{p}.self_attn.{n}.weight"] = (h, h)
            s[f"{p}.self_attn.{n}.bias"] = (h,)
        s[f"{p}.layer_norm1.weight"] = (h,)
        s[f"{p}.layer_norm1.bias"] = (h,)
        s[f"{p}.mlp.fc1.weight"] = (cfg.intermediate, h)
        s[f"{p}.mlp.fc1.bias"] = (cfg.intermediate,)
        s[f"{p}.mlp.fc2.weight"] = (h, cfg.intermediate)
        s[f"{p}.mlp.fc2.bias"] = (h,)
        s[f"{p}.layer_norm2.weight"] = (h,)
        s[f"{p}.layer_norm2.bias"] = (h,)
    s[f"{v}.post_layernorm.weight"] = (h,)
    s[f"{v}.post_layernorm.bias"] = (h,)
    s["model.visual_projection.weight"] = (cfg.proj, h)
    w = cfg.proj
    for i in range(cfg.mapper_layers):
        p = f"mapper2.resblocks.{i}"
        s[f"{p}.attn.c_qkv.weight"] = (3 * w, w)
        s[f"{p}.attn.c_qkv.bias"] = (3 * w,)
        s[f"{p}.attn.c_proj.weight"] = (w, w)
        s[f"{p}.attn.c_proj.bias"] = (w,)
        s[f"{p}.ln_1.weight"] = (w,)
        s[f"{p}.ln_1.bias"] = (w,)
        s[f"{p}.mlp.c_fc.weight"] = (4 * w, w)
        s[f"{p}.mlp.c_fc.bias"] = (4 * w,)
        s[f"{p}.mlp.c_proj.weight"] = (w, 4 * w)
        s[f"{p}.mlp.c_proj.bias"] = (w,)
        s[f"{p}.ln_2.weight"] = (w,)
        s[f"{p}.ln_2.bias"] = (w,)
    s["final_ln2.weight"] = (w,)
    s["final_ln2.bias"] = (w,)
    return s


# --------------------------------------------------------------------------------------------
# ArcFace IR-SE50
# --------------------------------------------------------------------------------------------
def arcface_units():
    """(in_channel, depth, stride) per bottleneck, helpers.py:25-37 (num_layers=50)."""
    units = []
    for cin, depth, n in ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3)):
        units.append((cin, depth, 2))
        units += [(depth, depth, 1)] * (n - 1)
    return units


def _bn(s, p, c, affine=True):
    if affine:
        s[f"{p}.weight"] = (c,)
        s[f"{p}.bias"] = (c,)
    s[f"{p}.running_mean"] = (c,)
    s[f"{p}.running_var"] = (c,)
    s[f"{p}.num_batches_tracked"] = ()


def arcface_param_specs():
    """Keys relative to ``face_ID_model.facenet.`` (model_irse.py:20-43, helpers.py:97-119)."""
    s = OrderedDict()
    s["input_layer.0.weight"] = (64, 3, 3, 3)
    _bn(s, "input_layer.1", 64)
    s["input_layer.2.weight"] = (64,)
    _bn(s, "output_layer.0", 512)
    s["output_layer.3.weight"] = (512, 512 * 7 * 7)
    s["output_layer.3.bias"] = (512,)
    _bn(s, "output_layer.4", 512)
    for i, (cin, depth, _stride) in enumerate(arcface_units()):
        p = f"body.{i}"
        if cin != depth:
            s[f"{p}.shortcut_layer.0.weight"] = (depth, cin, 1, 1)
            _bn(s, f"{p}.shortcut_layer.1", depth)
        _bn(s, f"{p}.res_layer.0", cin)
        s[f"{p}.res_layer.1.weight"] = (depth, cin, 3, 3)
        s[f"{p}.res_layer.2.weight"] = (depth,)
        s[f"{p}.res_layer.3.weight"] = (depth, depth, 3, 3)
        _bn(s, f"{p}.res_layer.4", depth)
        s[f"{p}.res_layer.5.fc1.weight"] = (depth // 16, depth, 1, 1)
        s[f"{p}.res_layer.5.fc2.weight"] = (depth, depth // 16, 1, 1)
    return s


def cond_head_specs():
    """Top-level LatentDiffusion conditioning heads (ddpm.py:698-733)."""
    s = OrderedDict()
    s["learnable_vector"] = (1, 1, 768)
    s["ID_proj_out.weight"] = (768, 512)
    s["ID_proj_out.bias"] = (768,)
    s["landmark_proj_out.weight"] = (768, 136)
    s["landmark_proj_out.bias"] = (768,)
    s["proj_out_source.weight"] = (768, 768)
    s["proj_out_source.bias"] = (768,)
    s["proj_out_target.weight"] = (768, 768)
    s["proj_out_target.bias"] = (768,)
    return s


# --------------------------------------------------------------------------------------------
# Seeded initialisation
# --------------------------------------------------------------------------------------------
def _key_seed(key: str, seed: int) -> int:
    return (zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def seeded_tensor(key: str, shape: Sequence[int], seed: int) -> torch.Tensor:
    """Deterministic fp32 tensor for one parameter.

    * matrices / conv kernels: U(-b, b), b = sqrt(3 / fan_in)  (unit gain, keeps activations O(1))
    * 1-D ``weight`` (norm gains, PReLU slopes) and ``running_var``: U(0.8, 1.2)
    * 1-D ``bias`` / ``running_mean`` / embeddings: U(-0.1, 0.1)
    * ``learnable_vector`` and embedding tables: U(-1, 1) * 0.5
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(_key_seed(key, seed))
    shape = tuple(int(d) for d in shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    u = torch.rand(shape, generator=g, dtype=torch.float32)
    if leaf in ("learnable_vector", "class_embedding") or "position_embedding" in key:
        return u - 0.5
    if len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        b = math.sqrt(3.0 / fan_in)
        return (u * 2.0 - 1.0) * b
    if leaf in ("weight", "running_var"):
        return 0.8 + 0.4 * u
    return (u * 2.0 - 1.0) * 0.1


def seeded_state_dict(specs: Dict[str, Tuple[int, ...]], seed: int, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, shp in specs.items():
        out[prefix + k] = seeded_tensor(prefix + k, shp, seed)
    return out


def seeded_randn(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    """Deterministic N(0,1) CPU tensor (synthetic inputs for tests, fixtures and the bench)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale


def state_dict_digest(sd: Dict[str, torch.Tensor], keys: Iterable[str] = None) -> float:
    """Cheap drift detector: sum over tensors of (mean |x|), in float64."""
    tot = 0.0
    for k in (keys if keys is not None else sd.keys()):
        t = sd[k]
        if t.dtype.is_floating_point and t.numel():
            tot += float(t.double().abs().mean())
    return tot
