#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE itself (/root/reference).

Build-container only.  Imports the reference's Python modules through tools/ref_shims.py,
loads them with weights from ``reface_amd.params.seeded_state_dict`` (strict key/shape match --
this also pins the checkpoint key layout), runs them on seeded inputs and stores inputs/outputs
as small ``.npz`` fixtures under tests/golden/.  Only DATA is stored (tensors + the seeds/configs
needed to regenerate the weights); no reference source is copied.

Usage:  python tools/gen_golden.py [group ...]     (groups: schedule unet_ops unet_small unet_full
                                                    ddim vae arcface clip e2e)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()                      # puts /root/reference at sys.path[0]
sys.path.append(os.path.dirname(HERE))   # repo root AFTER the reference: `ldm` stays the reference's

from reface_amd import params as P  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)
torch.set_num_threads(int(os.environ.get("GEN_THREADS", "8")))


rnd = P.seeded_randn     # inputs are regenerated from (shape, seed) by the tests; only outputs are stored


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def load_strict(module, sd):
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])


# ---------------------------------------------------------------------------------------------
def gen_schedule():
    from ldm.modules.diffusionmodules.util import (make_beta_schedule, make_ddim_timesteps,
                                                   make_ddim_sampling_parameters)
    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120)
    ac = torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)   # ddpm.py:262-272
    out = {"betas": betas, "alphas_cumprod": ac}
    for S in (5, 50):
        ts = make_ddim_timesteps("uniform", S, 1000, verbose=False)
        for eta in (0.0, 0.5):
            sig, a, ap = make_ddim_sampling_parameters(ac, ts, eta, verbose=False)
            tag = f"S{S}_eta{int(eta*10)}"
            out[f"ts_{tag}"] = ts
            out[f"sigmas_{tag}"] = np.asarray(sig, dtype=np.float64)
            out[f"alphas_{tag}"] = a
            out[f"alphas_prev_{tag}"] = ap
            out[f"sqrt1m_{tag}"] = np.sqrt(1.0 - a)
    save("schedule", **out)


def _ref_unet(cfg: P.UNetConfig, seed):
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    m = UNetModel(image_size=32, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                  model_channels=cfg.model_channels, attention_resolutions=list(cfg.attention_resolutions),
                  num_res_blocks=cfg.num_res_blocks, channel_mult=list(cfg.channel_mult),
                  num_heads=cfg.num_heads, use_spatial_transformer=True, transformer_depth=1,
                  context_dim=cfg.context_dim, use_checkpoint=True, legacy=False,
                  add_conv_in_front_of_unet=False).eval()
    sd = P.seeded_state_dict(P.unet_param_specs(cfg), seed)
    load_strict(m, sd)
    return m, sd


def gen_unet_ops():
    """Single ResBlock / SpatialTransformer / timestep-embed at full REFace widths."""
    from ldm.modules.diffusionmodules.openaimodel import ResBlock, Downsample, Upsample
    from ldm.modules.attention import SpatialTransformer
    from ldm.modules.diffusionmodules.util import timestep_embedding
    import collections
    res = {}
    t = torch.tensor([981, 1, 500, 21], dtype=torch.long)
    res["temb_t"] = t
    res["temb_out"] = timestep_embedding(t, 320)
    for tag, cin, cout, hw in (("a", 320, 320, 16), ("b", 2560, 1280, 8), ("c", 960, 640, 8)):
        blk = ResBlock(cin, 1280, 0, out_channels=cout, dims=2, use_checkpoint=True).eval()
        s = collections.OrderedDict()
        P._res_specs(s, "r", cin, cout, 1280)
        sd = P.seeded_state_dict(s, 100)
        load_strict(blk, {k[2:]: v for k, v in sd.items()})
        x = rnd((2, cin, hw, hw), 1)
        emb = rnd((2, 1280), 2)
        res[f"res_{tag}_y"] = blk(x, emb)
    for tag, c, heads, hw in (("a", 320, 8, 16), ("b", 1280, 8, 8), ("c", 640, 8, 12)):
        st = SpatialTransformer(c, heads, c // heads, depth=1, context_dim=768).eval()
        s = collections.OrderedDict()
        P._st_specs(s, "s", c, 768)
        sd = P.seeded_state_dict(s, 101)
        load_strict(st, {k[2:]: v for k, v in sd.items()})
        x = rnd((2, c, hw, hw), 3)
        ctx = rnd((2, 1, 768), 4)
        res[f"st_{tag}_y"] = st(x, ctx)
    save("unet_ops", **res)


SMALL_UNET = dict(in_channels=9, model_channels=64, out_channels=4, num_res_blocks=2,
                  attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768)


def gen_unet_small():
    cfg = P.UNetConfig(**SMALL_UNET)
    m, _ = _ref_unet(cfg, 7)
    x = rnd((2, 9, 16, 16), 10)
    t = torch.tensor([981, 41], dtype=torch.long)
    ctx = rnd((2, 1, 768), 11)
    save("unet_small", t=t, y=m(x, t, context=ctx), seed=7)
    x = rnd((2, 9, 24, 24), 12)          # non power-of-two grid (768-px style: 24 -> 12 -> 6 -> 3)
    save("unet_small_24", t=t, y=m(x, t, context=ctx), seed=7)


def gen_unet_full():
    cfg = P.UNetConfig()
    t0 = time.time()
    m, _ = _ref_unet(cfg, 1234)
    print(f"  full UNet built in {time.time()-t0:.1f}s")
    x = rnd((2, 9, 8, 8), 20)
    t = torch.tensor([961, 961], dtype=torch.long)
    ctx = rnd((2, 1, 768), 21)
    y = m(x, t, context=ctx)
    save("unet_full_8", t=t, y=y, seed=1234)
    x = rnd((1, 9, 16, 16), 22)
    t = torch.tensor([21], dtype=torch.long)
    ctx = rnd((1, 1, 768), 23)
    save("unet_full_16", t=t, y=m(x, t, context=ctx), seed=1234)


def gen_unet_keys():
    """{state-dict key: shape} of the REFERENCE's full-width UNetModel (openaimodel.py:666-830, the REFace configuration) and of a small
    configuration with a different channel_mult / attention pattern: what oracle.unet.plan_from_shapes reads the block structure from."""
    import json
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    out = {}
    for tag, cfg in (("full", P.UNetConfig()),
                     ("small", P.UNetConfig(model_channels=32, channel_mult=(1, 2, 4), attention_resolutions=(2, 1), num_res_blocks=1, num_heads=4, context_dim=32))):
        with torch.device("meta"):
            m = UNetModel(image_size=32, in_channels=cfg.in_channels, out_channels=cfg.out_channels, model_channels=cfg.model_channels,
                          attention_resolutions=list(cfg.attention_resolutions), num_res_blocks=cfg.num_res_blocks, channel_mult=list(cfg.channel_mult),
                          num_heads=cfg.num_heads, use_spatial_transformer=True, transformer_depth=1, context_dim=cfg.context_dim, use_checkpoint=True,
                          legacy=False, add_conv_in_front_of_unet=False)
        out[tag] = {"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()},
                    "shapes": {k: list(v.shape) for k, v in m.state_dict().items()}}
    path = os.path.join(OUT, "unet_keys.json")
    json.dump(out, open(path, "w"), separators=(",", ":"), sort_keys=True)
    print(f"  wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB, {len(out['full']['shapes'])} + {len(out['small']['shapes'])} keys)")


class _StubLDM:
    """What DDIMSampler reads from the model (ddim.py:100,113-119,207,345)."""

    def __init__(self, unet):
        from ldm.modules.diffusionmodules.util import make_beta_schedule
        betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = 1000
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
        self.device = torch.device("cpu")
        self.unet = unet

    def apply_model(self, x, t, c):
        return self.unet(x, t, context=c)


def gen_ddim():
    from ldm.models.diffusion.ddim import DDIMSampler
    DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    cfg = P.UNetConfig(**SMALL_UNET)
    m, _ = _ref_unet(cfg, 7)
    sampler = DDIMSampler(_StubLDM(m))
    B, h = 2, 16
    x_T = rnd((B, 4, h, h), 30)
    z_inp = rnd((B, 4, h, h), 31)
    mask = (rnd((B, 1, h, h), 32) > 0).float()
    c = rnd((B, 1, 768), 33)
    uc = rnd((1, 1, 768), 34).repeat(B, 1, 1)
    for S in (5, 50):
        samples, inter = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False,
                                        unconditional_guidance_scale=3.5, unconditional_conditioning=uc,
                                        eta=0.0, x_T=x_T, log_every_t=100,
                                        test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
        save(f"ddim_small_S{S}", samples=samples,
             pred_x0_last=inter["pred_x0"][-1], n_inter=len(inter["x_inter"]), seed=7, scale=3.5)
    # eta > 0 with recorded noise (RNG stream: one randn per step, util.py:264-267)
    torch.manual_seed(99)
    samples, _ = sampler.sample(S=5, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False,
                                unconditional_guidance_scale=3.5, unconditional_conditioning=uc,
                                eta=0.5, x_T=x_T,
                                test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
    torch.manual_seed(99)
    noises = torch.stack([torch.randn((B, 4, h, h)) for _ in range(5)])
    save("ddim_small_S5_eta5", samples=samples, noises=noises, seed=7, scale=3.5)


def gen_ddim_full():
    """SURVEY 8d gate at FULL S and FULL width: the reference's own UNet (859.5 M parameters, seeded weights 1234) under the reference's
    DDIMSampler (ddim.py:96-251, 323-375), S = 50, B = 2 at 64x64 latents, CFG scale 3.5, eta 0, inpainting kwargs -- then the reference's
    full-width AutoencoderKL.decode (seeded weights 55) of the samples / 0.18215 (ddpm.py:1102-1113).  ~20 minutes on 8 CPU threads.
    Stored: the latents after 50 steps, the last pred_x0 and every 8th pixel of the decoded images (the full images would be 6 MB:
    the test decodes the stored latents with the oracle and checks that decode against these pixels)."""
    from ldm.models.diffusion.ddim import DDIMSampler
    DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    m, _ = _ref_unet(P.UNetConfig(), 1234)
    sampler = DDIMSampler(_StubLDM(m))
    B, h, S = 2, 64, 50
    x_T = rnd((B, 4, h, h), 480)
    mask = (rnd((B, 1, h, h), 482) > 0).float()
    z_inp = rnd((B, 4, h, h), 481) * mask
    c = rnd((B, 1, 768), 483)
    uc = rnd((1, 1, 768), 484).repeat(B, 1, 1)
    t0 = time.time()
    samples, inter = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False, unconditional_guidance_scale=3.5,
                                    unconditional_conditioning=uc, eta=0.0, x_T=x_T, log_every_t=100,
                                    test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
    print(f"  reference DDIM S={S} B={B} full width: {time.time() - t0:.0f}s")
    del m, sampler
    vae, _ = _ref_vae(P.VAEConfig(), 55)
    img = vae.decode(samples / 0.18215)
    save("ddim_full_S50_B2", samples=samples, pred_x0_last=inter["pred_x0"][-1], image_stride8=img[:, :, ::8, ::8].contiguous(),
         image_absmax=img.abs().max(), seed_unet=1234, seed_vae=55, scale=3.5, S=S)


def gen_plms():
    """PLMSSampler (plms.py) with CFG on the reduced-width UNet, and LatentDiffusion.q_sample (ddpm.py:412-415)."""
    from ldm.models.diffusion.plms import PLMSSampler
    PLMSSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    cfg = P.UNetConfig(**SMALL_UNET)
    m, _ = _ref_unet(cfg, 7)
    ldm = _StubLDM(m)
    sampler = PLMSSampler(ldm)
    B, h = 2, 16
    x_T = rnd((B, 4, h, h), 30)
    z_inp = rnd((B, 4, h, h), 31)
    mask = (rnd((B, 1, h, h), 32) > 0).float()
    c = rnd((B, 1, 768), 33)
    uc = rnd((1, 1, 768), 34).repeat(B, 1, 1)
    for S in (5, 10):
        samples, inter = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False,
                                        unconditional_guidance_scale=3.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                        test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
        save(f"plms_small_S{S}", samples=samples, pred_x0_last=inter["pred_x0"][-1], n_inter=len(inter["x_inter"]), seed=7, scale=3.5)
    # q_sample through the reference DDPM buffers (register_schedule, fp32)
    from ldm.models.diffusion.ddpm import DDPM
    from reface_amd.schedule import ddpm_buffers
    bufs = ddpm_buffers(1000, 0.00085, 0.0120)
    host = type("H", (), {})()
    host.sqrt_alphas_cumprod = bufs["sqrt_alphas_cumprod"]
    host.sqrt_one_minus_alphas_cumprod = bufs["sqrt_one_minus_alphas_cumprod"]
    z = rnd((B, 4, h, h), 35)
    noise = rnd((B, 4, h, h), 36)
    t = torch.tensor([999, 417])
    save("q_sample", x=DDPM.q_sample(host, z, t, noise), t=t)


SMALL_VAE = dict(ch=32, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, out_ch=3, z_channels=4,
                 embed_dim=4, double_z=True, attn_resolutions=(), resolution=256)


def _ref_vae(cfg: P.VAEConfig, seed):
    from ldm.models.autoencoder import AutoencoderKL
    dd = dict(double_z=True, z_channels=cfg.z_channels, resolution=256, in_channels=cfg.in_channels,
              out_ch=cfg.out_ch, ch=cfg.ch, ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
              attn_resolutions=[], dropout=0.0)
    m = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=cfg.embed_dim).eval()
    sd = P.seeded_state_dict(P.vae_param_specs(cfg), seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    return m, sd


def gen_vae():
    import collections
    from ldm.modules.diffusionmodules.model import ResnetBlock, AttnBlock
    cfg = P.VAEConfig(**SMALL_VAE)
    m, _ = _ref_vae(cfg, 55)
    x = torch.tanh(rnd((2, 3, 64, 64), 40))
    post = m.encode(x)
    z = rnd((2, 4, 8, 8), 41)
    save("vae_small", mean=post.mean, logvar=post.logvar, dec=m.decode(z), seed=55)
    # full-width single blocks
    res = {}
    for tag, cin, cout, hw in (("a", 512, 512, 16), ("b", 512, 256, 16), ("c", 128, 128, 32)):
        blk = ResnetBlock(in_channels=cin, out_channels=cout, temb_channels=0, dropout=0.0).eval()
        s = collections.OrderedDict()
        P._vae_res(s, "r", cin, cout)
        sd = P.seeded_state_dict(s, 56)
        load_strict(blk, {k[2:]: v for k, v in sd.items()})
        xx = rnd((1, cin, hw, hw), 42)
        res[f"res_{tag}_y"] = blk(xx, None)
    ab = AttnBlock(512).eval()
    s = collections.OrderedDict()
    P._vae_attn(s, "a", 512)
    sd = P.seeded_state_dict(s, 57)
    load_strict(ab, {k[2:]: v for k, v in sd.items()})
    xx = rnd((1, 512, 16, 16), 43)
    res["attn_y"] = ab(xx)
    save("vae_blocks", **res)


def gen_arcface():
    from src.Face_models.encoders.model_irse import Backbone
    import ldm.models.diffusion.ddpm as ddpm
    net = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se").eval()
    sd = P.seeded_state_dict(P.arcface_param_specs(), 77)
    load_strict(net, sd)
    idl = ddpm.IDLoss.__new__(ddpm.IDLoss)
    torch.nn.Module.__init__(idl)
    idl.multiscale = False
    idl.face_pool_1 = torch.nn.AdaptiveAvgPool2d((256, 256))
    idl.face_pool_2 = torch.nn.AdaptiveAvgPool2d((112, 112))
    idl.facenet = net
    ref = rnd((2, 3, 224, 224), 50)
    feats = idl.extract_feats(ref)[0]
    x112 = rnd((2, 3, 112, 112), 51)
    save("arcface", feats=feats, feats112=net(x112)[0], seed=77)


SMALL_CLIP = dict(hidden=128, intermediate=512, layers=2, heads=4, patch=14, image=224, proj=768, mapper_layers=5)


def _hf_clip(cfg: P.CLIPVisionConfig):
    from transformers import CLIPConfig, CLIPModel
    c = CLIPConfig(projection_dim=cfg.proj,
                   vision_config=dict(hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                                      num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                                      patch_size=cfg.patch, image_size=cfg.image, hidden_act="quick_gelu"),
                   text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                                    num_attention_heads=2, vocab_size=100, max_position_embeddings=77,
                                    hidden_act="quick_gelu"))
    return CLIPModel(c)


def _ref_clip_embedder(cfg: P.CLIPVisionConfig, seed):
    import ldm.modules.encoders.modules as M
    from transformers import CLIPModel, CLIPTokenizer
    orig_m, orig_t = CLIPModel.from_pretrained, CLIPTokenizer.from_pretrained
    CLIPModel.from_pretrained = classmethod(lambda cls, *a, **k: _hf_clip(cfg))
    CLIPTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: None)
    try:
        emb = M.FrozenCLIPEmbedder().eval()
    finally:
        CLIPModel.from_pretrained, CLIPTokenizer.from_pretrained = orig_m, orig_t
    sd = P.seeded_state_dict(P.clip_param_specs(cfg), seed)
    have = emb.state_dict()
    # HF >= 5 may nest the vision tower one level deeper; map our 4.19-layout keys onto it.
    remap = {}
    for k in sd:
        if k in have:
            remap[k] = k
        else:
            alt = k.replace("model.vision_model.", "model.vision_model.vision_model.")
            assert alt in have, k
            remap[k] = alt
    missing, unexpected = emb.load_state_dict({remap[k]: v for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected[:5]
    used = [k for k in missing if ("vision_model" in k or "visual_projection" in k or "mapper2" in k or "final_ln2" in k)
            and "position_ids" not in k]
    assert not used, used[:5]
    return emb, sd


def gen_clip():
    cfg = P.CLIPVisionConfig(**SMALL_CLIP)
    emb, _ = _ref_clip_embedder(cfg, 88)
    img = rnd((2, 3, 224, 224), 60)
    pooled = emb.model.vision_model(pixel_values=img).pooler_output
    save("clip_small", pooled=pooled, z=emb(img), seed=88)
    # one full-width ViT-L/14 layer stack (1 layer) to pin 1024/16-head shapes
    cfg1 = P.CLIPVisionConfig(layers=1)
    emb1, _ = _ref_clip_embedder(cfg1, 89)
    img1 = rnd((1, 3, 224, 224), 61)
    save("clip_l14_1layer", z=emb1(img1), seed=89)


def gen_e2e():
    """Whole reference chain (inference_test_bench.py:441-495) at reduced widths: LatentDiffusion
    built through the reference registry from a config dict shaped like configs/train.yaml."""
    import yaml
    from ldm.util import instantiate_from_config
    from ldm.models.diffusion.ddim import DDIMSampler
    from transformers import CLIPModel, CLIPTokenizer
    import tempfile
    DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    raw = yaml.safe_load(open("/root/reference/configs/train.yaml"))
    mp = raw["model"]["params"]
    mp["unet_config"]["params"]["model_channels"] = 64
    mp["first_stage_config"]["params"]["ddconfig"]["ch"] = 32
    ccfg = P.CLIPVisionConfig(**SMALL_CLIP)
    arc_sd = P.seeded_state_dict(P.arcface_param_specs(), 77)
    with tempfile.NamedTemporaryFile(suffix=".pth", delete=False) as f:
        torch.save(arc_sd, f.name)
        arc_path = f.name
    mp["cond_stage_config"]["other_params"]["arcface_path"] = arc_path
    cfg = ref_shims.to_attr(raw)
    orig_m, orig_t = CLIPModel.from_pretrained, CLIPTokenizer.from_pretrained
    CLIPModel.from_pretrained = classmethod(lambda cls, *a, **k: _hf_clip(ccfg))
    CLIPTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: None)
    try:
        model = instantiate_from_config(cfg.model).eval()
    finally:
        CLIPModel.from_pretrained, CLIPTokenizer.from_pretrained = orig_m, orig_t
        os.unlink(arc_path)
    # our seeded weights under the checkpoint prefixes
    sd = {}
    sd.update(P.seeded_state_dict(P.unet_param_specs(P.UNetConfig(**SMALL_UNET)), 7, "model.diffusion_model."))
    sd.update(P.seeded_state_dict(P.vae_param_specs(P.VAEConfig(**SMALL_VAE)), 55, "first_stage_model."))
    csd = P.seeded_state_dict(P.clip_param_specs(ccfg), 88, "cond_stage_model.")
    have = model.state_dict()
    for k, v in csd.items():
        if k not in have:
            k = k.replace("model.vision_model.", "model.vision_model.vision_model.")
            assert k in have, k
        sd[k] = v
    sd.update({"face_ID_model.facenet." + k: v for k, v in arc_sd.items()})
    sd.update(P.seeded_state_dict(P.cond_head_specs(), 9))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:8]
    crit = [k for k in missing if not any(s in k for s in (
        "text_model", "text_projection", "logit_scale", "mapper.", "final_ln.", "projection_back", "position_ids",
        "betas", "alphas", "sqrt_", "log_one", "posterior", "logvar", "lvlb"))]
    assert not crit, crit[:8]
    print("  e2e model built; unused-at-inference params left at ctor init:", len(missing))

    B, H = 2, 256
    target = torch.tanh(rnd((B, 3, H, H), 70))
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    ell = (((yy - H / 2) / (0.30 * H)) ** 2 + ((xx - H / 2) / (0.38 * H)) ** 2) <= 1.0
    inpaint_mask = (~ell).float()[None, None].repeat(B, 1, 1, 1)
    inpaint_image = target * inpaint_mask
    ref = rnd((B, 3, 224, 224), 71)
    x_T = rnd((B, 4, H // 8, H // 8), 72)

    sampler = DDIMSampler(model)
    uc = model.learnable_vector.repeat(B, 1, 1)
    landmarks = model.get_landmarks(target)                       # dlib stub: no faces -> zeros(136) -> proj
    c = model.conditioning_with_feat(ref, landmarks=landmarks, tar=target).float()
    torch.manual_seed(4242)
    z_inpaint = model.get_first_stage_encoding(model.encode_first_stage(inpaint_image)).detach()
    torch.manual_seed(4242)
    eps = torch.randn(z_inpaint.shape)
    post = model.encode_first_stage(inpaint_image)
    from torchvision.transforms import Resize
    mask64 = Resize([H // 8, H // 8])(inpaint_mask)
    samples, _ = sampler.sample(S=5, conditioning=c, batch_size=B, shape=[4, H // 8, H // 8], verbose=False,
                                unconditional_guidance_scale=3.5, unconditional_conditioning=uc, eta=0.0,
                                x_T=x_T, test_model_kwargs={"inpaint_image": z_inpaint, "inpaint_mask": mask64})
    x_dec = model.decode_first_stage(samples)
    x_img = torch.clamp((x_dec + 1.0) / 2.0, min=0.0, max=1.0)
    u8 = (255.0 * x_img.permute(0, 2, 3, 1).numpy()).astype(np.uint8)
    save("e2e_small", landmarks=landmarks, c=c, uc=uc, post_mean=post.mean, post_logvar=post.logvar, eps=eps,
         z_inpaint=z_inpaint, mask64=mask64, samples=samples, x_dec=x_dec, u8=u8)

    # ---- the on-disk output tree: the reference's OWN statements (scripts/inference_test_bench.py:500-552, read from the
    # reference tree at generation time, never stored) executed on this run's tensors; PIL's save is replaced by a recorder.
    import textwrap
    import types as _types
    from einops import rearrange
    from torchvision.utils import make_grid                   # tools/ref_shims.py restatement (torchvision is not installed)
    src = open("/root/reference/scripts/inference_test_bench.py").read().split("\n")
    assert src[499].strip() == "def un_norm(x):" and src[551].strip().startswith("ref_img.save("), (src[499], src[551])
    block = textwrap.dedent("\n".join(src[499:552]))
    written = {}

    class _Rec:
        def __init__(self, a):
            self.a = np.array(a)

        def save(self, path):
            written[os.path.basename(path)] = self.a

    cv2 = _types.SimpleNamespace(COLOR_GRAY2RGB=8, cvtColor=lambda a, code: np.repeat(a, 3, axis=2))      # GRAY2RGB replicates the channel
    ns = dict(torch=torch, np=np, os=os, rearrange=rearrange, make_grid=make_grid, Resize=lambda sz: Resize([H, H]), cv2=cv2,
              Image=_types.SimpleNamespace(fromarray=_Rec), opt=_types.SimpleNamespace(skip_save=False),
              x_checked_image_torch=x_img, test_batch=target, inpaint_image=inpaint_image, inpaint_mask=inpaint_mask,
              test_model_kwargs={"ref_imgs": ref.unsqueeze(1)}, segment_id_batch=["a", "b"], grid_path="g", result_path="r",
              sample_path="s", base_count=0)
    exec(block, ns)
    assert sorted(written) == ["a.png", "a_GT.png", "a_inpaint.png", "a_mask.png", "a_ref.png", "b.png", "b_GT.png", "b_inpaint.png",
                               "b_mask.png", "b_ref.png", "grid-a.png", "grid-b.png"], sorted(written)
    # item 0 only (random images do not compress): grid carries the GT / inpaint / ref / result panels
    g0 = written["grid-a.png"]
    for k, nm in enumerate(("a_GT.png", "a_inpaint.png", "a_ref.png", "a.png")):       # the individual files ARE the grid's panels
        assert np.array_equal(g0[2:2 + H, 2 + k * (H + 2):2 + k * (H + 2) + H], written[nm]), nm
    save("e2e_png", grid=g0, mask=written["a_mask.png"])


GROUPS = dict(ddim_full=gen_ddim_full, unet_keys=gen_unet_keys, plms=gen_plms, schedule=gen_schedule, unet_ops=gen_unet_ops, unet_small=gen_unet_small, unet_full=gen_unet_full,
              ddim=gen_ddim, vae=gen_vae, arcface=gen_arcface, clip=gen_clip, e2e=gen_e2e)

if __name__ == "__main__":
    sel = sys.argv[1:] or [g for g in GROUPS if g != "ddim_full"]          # (ddim_full: 20 minutes; ask for it by name)
    for g in sel:
        print(f"[{g}]")
        t0 = time.time()
        GROUPS[g]()
        print(f"  {time.time()-t0:.1f}s")
