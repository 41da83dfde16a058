"""Pin the CPU oracle (oracle/) against golden vectors produced by the REFERENCE itself
(tools/gen_golden.py, which imports /root/reference in the build container).

Inputs and weights are regenerated from seeds (reface_amd.params.seeded_*); the fixtures hold
the reference's outputs.  Tolerances are fp32 round-off class (same ATen CPU kernels, possibly
different op grouping).
"""
import collections
import os

import numpy as np
import pytest
import torch

from reface_amd import params as P
from oracle import schedule, unet, vae, ddim, encoders

rnd = P.seeded_randn
torch.set_grad_enabled(False)

SMALL_UNET = dict(in_channels=9, model_channels=64, out_channels=4, num_res_blocks=2,
                  attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768)
SMALL_VAE = dict(ch=32, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, out_ch=3, z_channels=4,
                 embed_dim=4, double_z=True, attn_resolutions=(), resolution=256)
SMALL_CLIP = dict(hidden=128, intermediate=512, layers=2, heads=4, patch=14, image=224, proj=768, mapper_layers=5)


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def close(a, b, atol, rtol=0.0):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    lim = atol + rtol * np.abs(b)
    assert (err <= lim).all(), f"max err {err.max():.3e} (limit {atol:.1e}+{rtol:.1e}*|ref|), ref absmax {np.abs(b).max():.3e}"


def test_schedule(golden_dir):
    g = G(golden_dir, "schedule")
    betas = schedule.make_beta_schedule()
    assert np.array_equal(betas, g["betas"])
    ac = schedule.alphas_cumprod()
    assert np.array_equal(ac.numpy(), g["alphas_cumprod"])
    for S in (5, 50):
        ts = schedule.ddim_timesteps(S)
        for eta in (0.0, 0.5):
            tag = f"S{S}_eta{int(eta*10)}"
            assert np.array_equal(ts, g[f"ts_{tag}"])
            p = schedule.ddim_parameters(ac, ts, eta)
            assert np.array_equal(p["alphas"].numpy(), g[f"alphas_{tag}"])
            assert np.array_equal(p["alphas_prev"], g[f"alphas_prev_{tag}"])
            assert np.array_equal(p["sqrt_one_minus_alphas"].numpy(), g[f"sqrt1m_{tag}"])
            np.testing.assert_allclose(p["sigmas"], g[f"sigmas_{tag}"], rtol=1e-12, atol=0)
    assert list(schedule.ddim_timesteps(50)[:3]) == [1, 21, 41] and schedule.ddim_timesteps(50)[-1] == 981


def test_timestep_embedding(golden_dir):
    g = G(golden_dir, "unet_ops")
    close(unet.timestep_embedding(torch.from_numpy(g["temb_t"]), 320), g["temb_out"], 1e-6)


@pytest.mark.parametrize("tag,cin,cout,hw", [("a", 320, 320, 16), ("b", 2560, 1280, 8), ("c", 960, 640, 8)])
def test_resblock(golden_dir, tag, cin, cout, hw):
    g = G(golden_dir, "unet_ops")
    s = collections.OrderedDict()
    P._res_specs(s, "r", cin, cout, 1280)
    sd = P.seeded_state_dict(s, 100)
    y = unet.res_block(sd, "r", rnd((2, cin, hw, hw), 1), rnd((2, 1280), 2))
    close(y, g[f"res_{tag}_y"], 2e-5, 1e-5)


@pytest.mark.parametrize("tag,c,hw", [("a", 320, 16), ("b", 1280, 8), ("c", 640, 12)])
def test_spatial_transformer(golden_dir, tag, c, hw):
    g = G(golden_dir, "unet_ops")
    s = collections.OrderedDict()
    P._st_specs(s, "s", c, 768)
    sd = P.seeded_state_dict(s, 101)
    y = unet.spatial_transformer(sd, "s", rnd((2, c, hw, hw), 3), rnd((2, 1, 768), 4), 8)
    close(y, g[f"st_{tag}_y"], 2e-5, 1e-5)


def test_unet_block_structure_from_the_reference_key_layout(golden_dir):
    """oracle.unet.plan_from_shapes reads the block structure off a checkpoint's keys / shapes.  On the REFERENCE's own key list
    (tests/golden/unet_keys.json: state_dict of the reference's UNetModel, openaimodel.py:666-830, for the REFace configuration and a small
    one; tools/gen_golden.py unet_keys) it must give what the product derives from the constructor arguments (params.unet_plan), and the
    product's parameter specs must BE that key list -- so a structural slip in either derivation shows up here, on the CPU."""
    import json
    ref = json.load(open(os.path.join(golden_dir, "unet_keys.json")))
    for tag in ("full", "small"):
        cfg = P.UNetConfig(**ref[tag]["config"])
        shapes = {k: tuple(v) for k, v in ref[tag]["shapes"].items()}
        assert unet.plan_from_shapes(shapes, cfg.num_heads) == tuple(P.unet_plan(cfg)), tag
        specs = {k: tuple(v) for k, v in P.unet_param_specs(cfg).items()}
        assert specs == shapes, (tag, sorted(set(specs) ^ set(shapes))[:6])
    n_res = sum(1 for blk in unet.plan_from_shapes({k: tuple(v) for k, v in ref["full"]["shapes"].items()})[0] + unet.plan_from_shapes(
        {k: tuple(v) for k, v in ref["full"]["shapes"].items()})[2] for l in blk if l[0] == "res")
    assert n_res == 20          # 8 + 12 ResBlocks around the middle block's two (openaimodel.py: 2 per level down, 3 per level up)


def _small_unet():
    cfg = P.UNetConfig(**SMALL_UNET)
    sd = P.seeded_state_dict(P.unet_param_specs(cfg), 7)
    return cfg, sd, unet.plan_of(sd, cfg.num_heads)          # the oracle reads the block structure off the checkpoint layout itself (oracle.unet.plan_from_shapes)


def test_unet_small(golden_dir):
    cfg, sd, plan = _small_unet()
    for name, hw, xs in (("unet_small", 16, 10), ("unet_small_24", 24, 12)):
        g = G(golden_dir, name)
        y = unet.unet_forward(sd, plan, rnd((2, 9, hw, hw), xs), torch.from_numpy(g["t"]), rnd((2, 1, 768), 11),
                              cfg.model_channels)
        close(y, g["y"], 2e-5, 1e-5)


@pytest.mark.slow
def test_unet_full_width(golden_dir):
    cfg = P.UNetConfig()
    sd = P.seeded_state_dict(P.unet_param_specs(cfg), 1234)
    assert sum(v.numel() for v in sd.values()) == 859_535_364      # 859.54 M
    plan = unet.plan_of(sd, cfg.num_heads)
    g = G(golden_dir, "unet_full_8")
    y = unet.unet_forward(sd, plan, rnd((2, 9, 8, 8), 20), torch.from_numpy(g["t"]), rnd((2, 1, 768), 21))
    close(y, g["y"], 5e-5, 1e-5)
    g = G(golden_dir, "unet_full_16")
    y = unet.unet_forward(sd, plan, rnd((1, 9, 16, 16), 22), torch.from_numpy(g["t"]), rnd((1, 1, 768), 23))
    close(y, g["y"], 5e-5, 1e-5)


def _ddim_inputs():
    B, h = 2, 16
    x_T = rnd((B, 4, h, h), 30)
    z_inp = rnd((B, 4, h, h), 31)
    mask = (rnd((B, 1, h, h), 32) > 0).float()
    c = rnd((B, 1, 768), 33)
    uc = rnd((1, 1, 768), 34).repeat(B, 1, 1)
    return x_T, z_inp, mask, c, uc


@pytest.mark.parametrize("S", [5, 50])
def test_ddim_small(golden_dir, S):
    cfg, sd, plan = _small_unet()
    eps = lambda x, t, c: unet.unet_forward(sd, plan, x, t, c, cfg.model_channels)
    x_T, z_inp, mask, c, uc = _ddim_inputs()
    g = G(golden_dir, f"ddim_small_S{S}")
    samples, inter = ddim.sample(eps, S, x_T, c, uc, z_inp, mask, 3.5)
    close(samples, g["samples"], 1e-4 if S == 5 else 5e-4)
    close(inter["pred_x0"][-1], g["pred_x0_last"], 1e-4 if S == 5 else 5e-4)
    assert len(inter["x_inter"]) == int(g["n_inter"])


def test_ddim_eta(golden_dir):
    cfg, sd, plan = _small_unet()
    eps = lambda x, t, c: unet.unet_forward(sd, plan, x, t, c, cfg.model_channels)
    x_T, z_inp, mask, c, uc = _ddim_inputs()
    g = G(golden_dir, "ddim_small_S5_eta5")
    samples, _ = ddim.sample(eps, 5, x_T, c, uc, z_inp, mask, 3.5, eta=0.5, noises=torch.from_numpy(g["noises"]))
    close(samples, g["samples"], 1e-4)


@pytest.mark.parametrize("S", [5, 10])
def test_plms_small(golden_dir, S):
    """oracle.ddim.plms_sample vs the reference's PLMSSampler (plms.py) with CFG 3.5."""
    cfg, sd, plan = _small_unet()
    eps = lambda x, t, c: unet.unet_forward(sd, plan, x, t, c, cfg.model_channels)
    x_T, z_inp, mask, c, uc = _ddim_inputs()
    g = G(golden_dir, f"plms_small_S{S}")
    samples, inter = ddim.plms_sample(eps, S, x_T, c, uc, z_inp, mask, 3.5)
    close(samples, g["samples"], 2e-4)
    close(inter["pred_x0"][-1], g["pred_x0_last"], 2e-4)
    assert len(inter["x_inter"]) == int(g["n_inter"])


def test_q_sample(golden_dir):
    g = G(golden_dir, "q_sample")
    x = ddim.q_sample(rnd((2, 4, 16, 16), 35), torch.from_numpy(g["t"]), rnd((2, 4, 16, 16), 36))
    close(x, g["x"], 1e-6)


def test_vae_small(golden_dir):
    cfg = P.VAEConfig(**SMALL_VAE)
    sd = P.seeded_state_dict(P.vae_param_specs(cfg), 55)
    g = G(golden_dir, "vae_small")
    mean, logvar = vae.encode_moments(sd, cfg, torch.tanh(rnd((2, 3, 64, 64), 40)))
    close(mean, g["mean"], 2e-5, 1e-5)
    close(logvar, g["logvar"], 2e-5, 1e-5)
    z = rnd((2, 4, 8, 8), 41)
    h = torch.nn.functional.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    close(vae.decoder(sd, cfg, h), g["dec"], 2e-5, 1e-5)


def test_vae_blocks(golden_dir):
    g = G(golden_dir, "vae_blocks")
    for tag, cin, cout, hw in (("a", 512, 512, 16), ("b", 512, 256, 16), ("c", 128, 128, 32)):
        s = collections.OrderedDict()
        P._vae_res(s, "r", cin, cout)
        sd = P.seeded_state_dict(s, 56)
        close(vae.resnet_block(sd, "r", rnd((1, cin, hw, hw), 42)), g[f"res_{tag}_y"], 2e-5, 1e-5)
    s = collections.OrderedDict()
    P._vae_attn(s, "a", 512)
    sd = P.seeded_state_dict(s, 57)
    close(vae.attn_block(sd, "a", rnd((1, 512, 16, 16), 43)), g["attn_y"], 2e-5, 1e-5)


def test_arcface(golden_dir):
    g = G(golden_dir, "arcface")
    sd = P.seeded_state_dict(P.arcface_param_specs(), 77)
    assert sum(v.numel() for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k) == 43_797_696
    units = P.arcface_units()
    close(encoders.extract_id_feats(sd, units, rnd((2, 3, 224, 224), 50)), g["feats"], 2e-6)
    close(encoders.arcface_backbone(sd, units, rnd((2, 3, 112, 112), 51)), g["feats112"], 2e-6)


def test_clip_small(golden_dir):
    cfg = P.CLIPVisionConfig(**SMALL_CLIP)
    sd = P.seeded_state_dict(P.clip_param_specs(cfg), 88)
    g = G(golden_dir, "clip_small")
    img = rnd((2, 3, 224, 224), 60)
    close(encoders.clip_vision_pooled(sd, cfg, img), g["pooled"], 2e-5, 1e-5)
    close(encoders.clip_embed(sd, cfg, img), g["z"], 2e-5, 1e-5)


def test_clip_l14_layer(golden_dir):
    cfg = P.CLIPVisionConfig(layers=1)
    sd = P.seeded_state_dict(P.clip_param_specs(cfg), 89)
    g = G(golden_dir, "clip_l14_1layer")
    close(encoders.clip_embed(sd, cfg, rnd((1, 3, 224, 224), 61)), g["z"], 2e-5, 1e-5)


def test_e2e_small(golden_dir):
    """The reference's whole chain (inference_test_bench.py:441-495) at reduced widths."""
    g = G(golden_dir, "e2e_small")
    ucfg = P.UNetConfig(**SMALL_UNET)
    usd = P.seeded_state_dict(P.unet_param_specs(ucfg), 7, "model.diffusion_model.")
    usd = {k[len("model.diffusion_model."):]: v for k, v in usd.items()}
    vcfg = P.VAEConfig(**SMALL_VAE)
    vsd = {k[len("first_stage_model."):]: v
           for k, v in P.seeded_state_dict(P.vae_param_specs(vcfg), 55, "first_stage_model.").items()}
    ccfg = P.CLIPVisionConfig(**SMALL_CLIP)
    csd = {k[len("cond_stage_model."):]: v
           for k, v in P.seeded_state_dict(P.clip_param_specs(ccfg), 88, "cond_stage_model.").items()}
    asd = P.seeded_state_dict(P.arcface_param_specs(), 77)
    heads = P.seeded_state_dict(P.cond_head_specs(), 9)

    B, H = 2, 256
    target = torch.tanh(rnd((B, 3, H, H), 70))
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    ell = (((yy - H / 2) / (0.30 * H)) ** 2 + ((xx - H / 2) / (0.38 * H)) ** 2) <= 1.0
    inpaint_mask = (~ell).float()[None, None].repeat(B, 1, 1, 1)
    inpaint_image = target * inpaint_mask
    ref = rnd((B, 3, 224, 224), 71)
    x_T = rnd((B, 4, H // 8, H // 8), 72)

    lm136 = torch.zeros(B, 136)      # dlib found no face (ddpm.py:1081-1083)
    close(torch.nn.functional.linear(lm136, heads["landmark_proj_out.weight"], heads["landmark_proj_out.bias"]),
          g["landmarks"], 1e-6)
    c = encoders.conditioning_with_feat(heads, csd, ccfg, asd, P.arcface_units(), ref, lm136, target)
    close(c, g["c"], 2e-5, 1e-5)
    uc = heads["learnable_vector"].repeat(B, 1, 1)
    close(uc, g["uc"], 0)
    mean, logvar = vae.encode_moments(vsd, vcfg, inpaint_image)
    close(mean, g["post_mean"], 5e-5, 1e-5)
    close(logvar, g["post_logvar"], 5e-5, 1e-5)
    z_inp = vae.first_stage_encoding(mean, logvar, torch.from_numpy(g["eps"]))
    close(z_inp, g["z_inpaint"], 2e-5, 1e-5)
    m64 = encoders.mask64(inpaint_mask)
    close(m64, g["mask64"], 0)
    # mask64 rule of SURVEY Appendix A: mean of pixels {8i+3,8i+4}x{8j+3,8j+4}
    alt = 0.25 * (inpaint_mask[..., 3::8, 3::8] + inpaint_mask[..., 3::8, 4::8]
                  + inpaint_mask[..., 4::8, 3::8] + inpaint_mask[..., 4::8, 4::8])
    close(alt, g["mask64"], 0)
    plan = unet.plan_of(usd, ucfg.num_heads)
    eps_fn = lambda x, t, cc: unet.unet_forward(usd, plan, x, t, cc, ucfg.model_channels)
    samples, _ = ddim.sample(eps_fn, 5, x_T, c, uc, z_inp, m64, 3.5)
    close(samples, g["samples"], 2e-4)
    x_dec = vae.decode_first_stage(vsd, vcfg, samples)
    close(x_dec, g["x_dec"], 5e-4)
    u8 = encoders.to_uint8_image(x_dec)
    assert (np.abs(u8.astype(int) - g["u8"].astype(int)) <= 1).all()
    assert (u8 != g["u8"]).mean() < 1e-3


def test_output_tree_vs_reference_golden(golden_dir):
    """reface_amd/output.py against the files the reference's own save block (inference_test_bench.py:500-552) produced for the
    e2e fixture: mask, GT and inpaint panels depend only on the inputs -> byte-for-byte; grid geometry exact."""
    from reface_amd import output as O
    g = np.load(os.path.join(golden_dir, "e2e_png.npz"))
    B, H = 2, 256
    target = torch.tanh(rnd((B, 3, H, H), 70))
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    ell = (((yy - H / 2) / (0.30 * H)) ** 2 + ((xx - H / 2) / (0.38 * H)) ** 2) <= 1.0
    inpaint_mask = (~ell).float()[None, None].repeat(B, 1, 1, 1)
    inpaint_image = target * inpaint_mask
    zero = np.zeros((3, H, H), np.float32)
    o = O.compose(zero, target[0].numpy(), inpaint_image[0].numpy(), inpaint_mask[0].numpy(), zero)
    assert o["grid"].shape == g["grid"].shape
    assert np.array_equal(o["mask"], g["mask"])
    for k, nm in enumerate(("GT", "inpaint")):
        assert np.array_equal(o[nm], g["grid"][2:2 + H, 2 + k * (H + 2):2 + k * (H + 2) + H]), nm
    pad = np.ones(g["grid"].shape[:2], bool)
    for k in range(4):
        pad[2:2 + H, 2 + k * (H + 2):2 + k * (H + 2) + H] = False
    assert (g["grid"][pad] == 0).all() and np.array_equal(o["grid"][pad], g["grid"][pad])


@pytest.mark.slow
def test_oracle_full_width_decode_vs_reference_ddim50_golden(golden_dir):
    """The full-S full-width fixture (tests/golden/ddim_full_S50_B2.npz: the REFERENCE's UNet + DDIMSampler + AutoencoderKL.decode, 50 CFG steps,
    B = 2) holds every 8th pixel of the reference's decoded images: the oracle's full-width 512x512 decode of the stored latents must
    reproduce them -- the GPU gate test compares the HIP path's image with this oracle decode."""
    import numpy as np
    from oracle import vae as ovae
    from reface_amd import params as P
    g = np.load(os.path.join(golden_dir, "ddim_full_S50_B2.npz"))
    cfg = P.VAEConfig()
    vsd = P.seeded_state_dict(P.vae_param_specs(cfg), int(g["seed_vae"]))
    with torch.no_grad():
        img = ovae.decode_first_stage(vsd, cfg, torch.from_numpy(g["samples"][:1]))
    e = (img[:, :, ::8, ::8] - torch.from_numpy(g["image_stride8"][:1])).abs().max().item()
    assert e < 1e-4, e
    assert abs(float(g["image_absmax"])) < 50 and np.isfinite(g["samples"]).all() and np.isfinite(g["pred_x0_last"]).all()
