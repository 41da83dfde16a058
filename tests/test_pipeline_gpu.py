"""GPU parity of the HIP engines (UNet, DDIM sampler, KL-VAE) against
 (a) golden vectors produced by the reference itself (tests/golden, tools/gen_golden.py) and
 (b) the CPU oracle on the same seeded inputs.
fp32 (exact-fp32 MFMA) mode carries the parity gate; bf16 mode is checked against the fp32 result with a
loose, explicitly stated tolerance."""
import collections
import os
import types

import numpy as np
import pytest
import torch

from reface_amd import params as P
from reface_amd.params import seeded_randn as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda"

SMALL_UNET = dict(in_channels=9, model_channels=64, out_channels=4, num_res_blocks=2,
                  attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768)
SMALL_VAE = dict(ch=32, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, out_ch=3, z_channels=4,
                 embed_dim=4, double_z=True, attn_resolutions=(), resolution=256)


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def maxerr(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a.astype(np.float64) - np.asarray(b).astype(np.float64)).max())


def make_unet(cfg_kwargs, seed, dtype=torch.float32):
    from reface_amd.unet import UNetModel
    m = UNetModel(image_size=32, use_spatial_transformer=True, transformer_depth=1, use_checkpoint=True, legacy=False,
                  compute_dtype=dtype, **cfg_kwargs)
    sd = P.seeded_state_dict(P.unet_param_specs(m.cfg), seed)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def test_unet_small_vs_reference_golden(golden_dir):
    m = make_unet(SMALL_UNET, 7)
    for name, hw, xs in (("unet_small", 16, 10), ("unet_small_24", 24, 12)):
        g = G(golden_dir, name)
        y = m(rnd((2, 9, hw, hw), xs).to(DEV), torch.from_numpy(g["t"]).to(DEV), context=rnd((2, 1, 768), 11).to(DEV))
        assert y.shape == g["y"].shape and y.dtype == torch.float32
        assert maxerr(y, g["y"]) < 5e-5, maxerr(y, g["y"])


def test_unet_full_width_vs_reference_golden(golden_dir):
    m = make_unet(dict(in_channels=9, model_channels=320, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1),
                       channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768), 1234)
    g = G(golden_dir, "unet_full_8")
    y = m(rnd((2, 9, 8, 8), 20).to(DEV), torch.from_numpy(g["t"]).to(DEV), context=rnd((2, 1, 768), 21).to(DEV))
    assert maxerr(y, g["y"]) < 1e-4, maxerr(y, g["y"])
    g = G(golden_dir, "unet_full_16")
    y = m(rnd((1, 9, 16, 16), 22).to(DEV), torch.from_numpy(g["t"]).to(DEV), context=rnd((1, 1, 768), 23).to(DEV))
    assert maxerr(y, g["y"]) < 1e-4, maxerr(y, g["y"])


@pytest.mark.parametrize("hw", [64, 96])
def test_unet_full_size_structural_reductions(hw):
    """BASELINE configs[1] / [3] latent sizes (64x64, 96x96), full-width UNet, bf16, one CFG pair: the structural reductions are
    size-independent identities -- (a) computing the CFG-shared stem once (cfg_pair) and (b) taking the GroupNorm statistics from
    the GEMM epilogues change the result only by bf16 rounding noise; (c) the two batch halves differ (the context entered), and
    (d) the bf16 engine tracks the exact-fp32 engine of the same weights."""
    import os
    from reface_amd.unet import UNetEngine
    from reface_amd.modules import flat_state
    full = dict(in_channels=9, model_channels=320, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1),
                channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768)
    m = make_unet(full, 1234, torch.bfloat16)
    sd = flat_state(m)
    x1 = rnd((1, hw, hw, 9), 300)
    x = torch.zeros((2, hw, hw, 16))
    x[..., :9] = x1                                  # both halves: same x (classifier-free guidance)
    ctx = rnd((2, 768), 301)
    t = torch.tensor([481.0])

    def run(dtype, pair, fuse):
        os.environ["REFACE_GN_FUSE"] = "1" if fuse else "0"
        try:
            eng = UNetEngine(sd, m.cfg, 2, hw, hw, dtype, torch.device(DEV), uniform_t=True, cfg_pair=pair)
        finally:
            os.environ.pop("REFACE_GN_FUSE", None)
        eng.x_in.copy_(x.to(DEV).to(dtype))
        eng.set_context(ctx.to(DEV))
        eng.set_timesteps(t)
        eng.run()
        torch.cuda.synchronize()
        out = eng.eps.float().cpu().clone()
        n_fused = eng.gn_fused
        del eng
        torch.cuda.empty_cache()
        return out, n_fused

    base, nf0 = run(torch.bfloat16, False, False)
    pair, _ = run(torch.bfloat16, True, False)
    fused, nf1 = run(torch.bfloat16, True, True)
    assert nf0 == 0 and nf1 >= (55 if hw == 64 else 40)              # (nearly) every statistics pass is gone; 96-latent levels with
                                                                     # H*W not a multiple of the tile rows keep the separate pass
    scale = base.abs().max().item()
    assert torch.isfinite(base).all() and scale > 1e-3
    assert (pair - base).abs().max().item() < 0.03 * scale, ((pair - base).abs().max().item(), scale)
    assert (fused - base).abs().max().item() < 0.03 * scale, ((fused - base).abs().max().item(), scale)
    assert (base[0] - base[1]).abs().max().item() > 1e-4 * scale     # conditional vs unconditional half
    if hw == 64:
        ref, _ = run(torch.float32, True, True)
        rel = ((fused - ref).norm() / ref.norm()).item()
        assert rel < 0.05, rel


@pytest.mark.parametrize("width", [64, 320])
def test_unet_sample_split_at_96(width):
    """BASELINE configs[3] (96x96 latent, 8 UNet samples = 4 CFG pairs): the top level's 3x3 convolutions are 8 x 36 = 288 tiles of 256 rows --
    1.125 rounds of the chip -- and are launched as 7 + 1 samples (UNetEngine._add_conv3).  A scheduling choice: against the unsplit engine the
    result moves by bf16 rounding noise only (the tail part runs other tiles / split-K: another order of the fp32 additions), and the GroupNorm
    statistics still come from the GEMM epilogues (two producer launches with different tile plans per tensor)."""
    import os
    from reface_amd.unet import UNetEngine
    from reface_amd.modules import flat_state
    cfg = dict(in_channels=9, model_channels=width, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1),
               channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768)
    m = make_unet(cfg, 4321, torch.bfloat16)
    sd = flat_state(m)
    hw, B = 96, 8
    x = torch.zeros((B, hw, hw, 16))
    x[: B // 2, ..., :9] = rnd((B // 2, hw, hw, 9), 310)
    x[B // 2:] = x[: B // 2]
    ctx = rnd((B, 768), 311)
    t = torch.tensor([481.0])

    def run(split):
        os.environ["REFACE_SAMPLE_SPLIT"] = "1" if split else "0"
        try:
            eng = UNetEngine(sd, m.cfg, B, hw, hw, torch.bfloat16, torch.device(DEV), uniform_t=True, cfg_pair=True)
        finally:
            os.environ.pop("REFACE_SAMPLE_SPLIT", None)
        eng.x_in.copy_(x.to(DEV).to(torch.bfloat16))
        eng.set_context(ctx.to(DEV))
        eng.set_timesteps(t)
        eng.run()
        torch.cuda.synchronize()
        out = eng.eps.float().cpu().clone()
        n, nf, nl = eng.n_sample_split, eng.gn_fused, len(eng.main)
        del eng
        torch.cuda.empty_cache()
        return out, n, nf, nl

    base, n0, nf0, nl0 = run(False)
    got, n1, nf1, nl1 = run(True)
    print(f"width {width}: {n1} convolutions split by samples, {nl0} -> {nl1} launches, statistics fused {nf0} -> {nf1}")
    assert n0 == 0 and n1 >= 8 and nl1 >= nl0 + n1 and nf1 == nf0
    scale = base.abs().max().item()
    assert torch.isfinite(got).all() and scale > 1e-3
    assert (got - base).abs().max().item() < 0.03 * scale, ((got - base).abs().max().item(), scale)


def test_unet_small_bf16_close_to_fp32(golden_dir):
    g = G(golden_dir, "unet_small")
    m = make_unet(SMALL_UNET, 7, torch.bfloat16)
    y = m(rnd((2, 9, 16, 16), 10).to(DEV), torch.from_numpy(g["t"]).to(DEV), context=rnd((2, 1, 768), 11).to(DEV))
    ref = g["y"]
    rel = maxerr(y, ref) / np.abs(ref).max()
    assert rel < 0.08, rel          # bf16 storage + bf16 MFMA through ~60 layers: percent-level agreement


def test_unet_small_fp16_close_to_fp32(golden_dir):
    """The fp16 mode on the reduced-width UNet against the REFERENCE's output (golden): an eighth of the bf16 mode's bound (11 significant bits against 8)."""
    g = G(golden_dir, "unet_small")
    m = make_unet(SMALL_UNET, 7, torch.float16)
    y = m(rnd((2, 9, 16, 16), 10).to(DEV), torch.from_numpy(g["t"]).to(DEV), context=rnd((2, 1, 768), 11).to(DEV))
    ref = g["y"]
    rel = maxerr(y, ref) / np.abs(ref).max()
    print(f"small UNet fp16 vs the reference's golden output: max |d| / max |y| = {rel:.2e}")
    assert rel < 0.01, rel


class _LDMStub:
    """What DDIMSampler reads from the pipeline model (ddim.py:100,113-119,207,345)."""

    def __init__(self, unet):
        from reface_amd.schedule import ddpm_buffers
        b = ddpm_buffers(1000, 0.00085, 0.0120)
        self.num_timesteps = 1000
        self.betas, self.alphas_cumprod, self.alphas_cumprod_prev = b["betas"], b["alphas_cumprod"], b["alphas_cumprod_prev"]
        self.device = torch.device(DEV)
        self.model = types.SimpleNamespace(diffusion_model=unet)


def _ddim_inputs():
    B, h = 2, 16
    x_T = rnd((B, 4, h, h), 30)
    z_inp = rnd((B, 4, h, h), 31)
    mask = (rnd((B, 1, h, h), 32) > 0).float()
    c = rnd((B, 1, 768), 33)
    uc = rnd((1, 1, 768), 34).repeat(B, 1, 1)
    return x_T, z_inp, mask, c, uc


@pytest.mark.parametrize("S,graph", [(5, False), (5, True), (50, True), (50, "per_step"), (5, "per_step")])
def test_ddim_vs_reference_golden(golden_dir, S, graph, monkeypatch):
    """graph=True: the whole S-step loop is ONE captured HIP graph (in-graph copies deliver the intermediates);
    "per_step": one graph of a single step replayed S times (REFACE_GRAPH_STEPS=1); False: eager launches."""
    from reface_amd.ddim import DDIMSampler
    unet = make_unet(SMALL_UNET, 7)
    if graph == "per_step":
        monkeypatch.setenv("REFACE_GRAPH_STEPS", "1")
    sampler = DDIMSampler(_LDMStub(unet), use_graph=bool(graph))
    x_T, z_inp, mask, c, uc = _ddim_inputs()
    g = G(golden_dir, f"ddim_small_S{S}")
    samples, inter = sampler.sample(S=S, conditioning=c.to(DEV), batch_size=2, shape=[4, 16, 16], verbose=False,
                                    unconditional_guidance_scale=3.5, unconditional_conditioning=uc.to(DEV), eta=0.0,
                                    x_T=x_T.to(DEV), test_model_kwargs={"inpaint_image": z_inp.to(DEV), "inpaint_mask": mask.to(DEV)})
    tol = 2e-4 if S == 5 else 1e-3
    assert maxerr(samples, g["samples"]) < tol, maxerr(samples, g["samples"])
    assert maxerr(inter["pred_x0"][-1], g["pred_x0_last"]) < tol
    assert len(inter["x_inter"]) == int(g["n_inter"])
    # second call re-uses the captured graph and must reproduce the first bit for bit
    samples2, _ = sampler.sample(S=S, conditioning=c.to(DEV), batch_size=2, shape=[4, 16, 16], verbose=False,
                                 unconditional_guidance_scale=3.5, unconditional_conditioning=uc.to(DEV), eta=0.0,
                                 x_T=x_T.to(DEV), test_model_kwargs={"inpaint_image": z_inp.to(DEV), "inpaint_mask": mask.to(DEV)})
    assert torch.equal(samples, samples2)


def test_ddim_eta_vs_reference_golden(golden_dir):
    from reface_amd.ddim import DDIMSampler
    unet = make_unet(SMALL_UNET, 7)
    sampler = DDIMSampler(_LDMStub(unet))
    x_T, z_inp, mask, c, uc = _ddim_inputs()
    g = G(golden_dir, "ddim_small_S5_eta5")
    samples, _ = sampler.sample(S=5, conditioning=c.to(DEV), batch_size=2, shape=[4, 16, 16], verbose=False,
                                unconditional_guidance_scale=3.5, unconditional_conditioning=uc.to(DEV), eta=0.5,
                                x_T=x_T.to(DEV), x_noise=torch.from_numpy(g["noises"]),
                                test_model_kwargs={"inpaint_image": z_inp.to(DEV), "inpaint_mask": mask.to(DEV)})
    assert maxerr(samples, g["samples"]) < 2e-4, maxerr(samples, g["samples"])


def test_ddim_no_cfg_matches_oracle():
    from oracle import ddim as oddim, unet as ounet
    from reface_amd.ddim import DDIMSampler
    unet = make_unet(SMALL_UNET, 7)
    cfg = P.UNetConfig(**SMALL_UNET)
    sd = P.seeded_state_dict(P.unet_param_specs(cfg), 7)
    plan = ounet.plan_of(sd, cfg.num_heads)
    x_T, z_inp, mask, c, uc = _ddim_inputs()
    ref, _ = oddim.sample(lambda x, t, cc: ounet.unet_forward(sd, plan, x, t, cc, 64), 5, x_T, c, None, z_inp, mask, 1.0)
    sampler = DDIMSampler(_LDMStub(unet))
    got, _ = sampler.sample(S=5, conditioning=c.to(DEV), batch_size=2, shape=[4, 16, 16], verbose=False, eta=0.0, x_T=x_T.to(DEV),
                            test_model_kwargs={"inpaint_image": z_inp.to(DEV), "inpaint_mask": mask.to(DEV)})
    assert maxerr(got, ref) < 2e-4, maxerr(got, ref)


@pytest.mark.parametrize("which", ["ddim", "plms"])
def test_sampler_leaves_device_rng_in_reference_state(which):
    """The reference draws x_T (ddim.py:211 / plms.py:125) and then ONE torch.randn(shape) on the device per step even at eta = 0
    (ddim.py:371 / plms.py:212 via util.py:264-267).  A seeded run must therefore leave the device generator where S + 1 draws leave it,
    and start from the same x_T -- batch i + 1 of a seeded CLI run then sees the reference's noise."""
    from reface_amd.ddim import DDIMSampler
    from reface_amd.plms import PLMSSampler
    unet = make_unet(SMALL_UNET, 7)
    _, z_inp, mask, c, uc = _ddim_inputs()
    S, shape = 5, (2, 4, 16, 16)
    sampler = (DDIMSampler if which == "ddim" else PLMSSampler)(_LDMStub(unet))
    kw = dict(S=S, conditioning=c.to(DEV), batch_size=2, shape=[4, 16, 16], verbose=False, eta=0.0, unconditional_guidance_scale=3.5,
              unconditional_conditioning=uc.to(DEV), test_model_kwargs={"inpaint_image": z_inp.to(DEV), "inpaint_mask": mask.to(DEV)})
    torch.manual_seed(1234)
    got, _ = sampler.sample(x_T=None, **kw)
    after = torch.randn(shape, device=DEV)
    torch.manual_seed(1234)
    x_T = torch.randn(shape, device=DEV)
    for _ in range(S):
        torch.randn(shape, device=DEV)
    expect_after = torch.randn(shape, device=DEV)
    assert torch.equal(after, expect_after)
    ref, _ = sampler.sample(x_T=x_T, **kw)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("S", [5, 10])
def test_plms_vs_reference_golden(golden_dir, S):
    """PLMSSampler on the HIP engine vs the reference's plms.py (CFG 3.5; improved-Euler first step, Adams-Bashforth after)."""
    from reface_amd.plms import PLMSSampler
    unet = make_unet(SMALL_UNET, 7)
    sampler = PLMSSampler(_LDMStub(unet))
    x_T, z_inp, mask, c, uc = _ddim_inputs()
    g = G(golden_dir, f"plms_small_S{S}")
    samples, inter = sampler.sample(S=S, conditioning=c.to(DEV), batch_size=2, shape=[4, 16, 16], verbose=False,
                                    unconditional_guidance_scale=3.5, unconditional_conditioning=uc.to(DEV), eta=0.0,
                                    x_T=x_T.to(DEV), test_model_kwargs={"inpaint_image": z_inp.to(DEV), "inpaint_mask": mask.to(DEV)})
    assert maxerr(samples, g["samples"]) < 3e-4, maxerr(samples, g["samples"])
    assert maxerr(inter["pred_x0"][-1], g["pred_x0_last"]) < 3e-4
    assert len(inter["x_inter"]) == int(g["n_inter"])
    with pytest.raises(ValueError, match="ddim_eta must be 0 for PLMS"):
        sampler.sample(S=S, conditioning=c.to(DEV), batch_size=2, shape=[4, 16, 16], verbose=False, eta=0.5, x_T=x_T.to(DEV),
                       test_model_kwargs={"inpaint_image": z_inp.to(DEV), "inpaint_mask": mask.to(DEV)})


def test_q_sample_vs_reference_golden(golden_dir):
    """LatentDiffusion.q_sample (--Start_from_target, ddpm.py:412-415) through the HIP combine kernel."""
    import types
    from reface_amd.ddpm import LatentDiffusion
    from reface_amd.schedule import ddpm_buffers
    g = G(golden_dir, "q_sample")
    b = ddpm_buffers(1000, 0.00085, 0.0120)
    host = types.SimpleNamespace(device=torch.device(DEV), sqrt_alphas_cumprod=b["sqrt_alphas_cumprod"],
                                 sqrt_one_minus_alphas_cumprod=b["sqrt_one_minus_alphas_cumprod"])
    x = LatentDiffusion.q_sample(host, rnd((2, 4, 16, 16), 35).to(DEV), torch.from_numpy(g["t"]), rnd((2, 4, 16, 16), 36).to(DEV))
    assert maxerr(x, g["x"]) < 1e-6, maxerr(x, g["x"])


def make_vae(cfg_kwargs, seed):
    from reface_amd.vae import AutoencoderKL
    dd = dict(cfg_kwargs)
    emb = dd.pop("embed_dim")
    dd["ch_mult"], dd["attn_resolutions"] = list(dd["ch_mult"]), []
    m = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=emb)
    m.load_state_dict(P.seeded_state_dict(P.vae_param_specs(m.cfg), seed), strict=True)
    return m.to(DEV).eval()


def test_vae_small_vs_reference_golden(golden_dir):
    m = make_vae(SMALL_VAE, 55)
    g = G(golden_dir, "vae_small")
    post = m.encode(torch.tanh(rnd((2, 3, 64, 64), 40)).to(DEV))
    assert maxerr(post.mean, g["mean"]) < 5e-5 and maxerr(post.logvar, g["logvar"]) < 5e-5
    for mode, lim in (("f32", 5e-5), ("bf16x3", 2e-4)):          # exact-fp32 MFMA; split-bf16 operands on the layers wide enough for it (default)
        m.decode_mode = mode
        m._engines.clear()
        dec = m.decode(rnd((2, 4, 8, 8), 41).to(DEV))
        assert dec.shape == g["dec"].shape
        assert maxerr(dec, g["dec"]) < lim, (mode, maxerr(dec, g["dec"]))
    # posterior sample + scale (distributions.py:35-37, ddpm.py:857) against the oracle formula
    eps = rnd((2, 4, 8, 8), 44)
    z = post.sample(noise=eps, scale=0.18215)
    ref = 0.18215 * (torch.from_numpy(g["mean"]) + torch.exp(0.5 * torch.from_numpy(g["logvar"])) * eps)
    assert maxerr(z, ref) < 2e-5


def test_vae_full_width_blocks_vs_reference_golden(golden_dir):
    """Full-width (512-channel) ResnetBlock / AttnBlock of the decoder via a 1-level VAE engine."""
    from reface_amd.vae import _VAEEngine
    from reface_amd import ops
    g = G(golden_dir, "vae_blocks")
    cfg = P.VAEConfig()

    def run_block(kind, specs_fn, seed, x, *shape_args, mode="f32"):
        s = collections.OrderedDict()
        specs_fn(s)
        sd = P.seeded_state_dict(s, seed)
        eng = _VAEEngine.__new__(_VAEEngine)
        from reface_amd.unet import _Pool
        B = x.shape[0]
        eng.cfg, eng.B, eng.dt, eng.dev = cfg, B, torch.float32, torch.device(DEV)
        eng.x3, eng.n_x3 = mode == "bf16x3", 0
        eng.pool = _Pool(eng.dev)
        eng.sd = {k: v.to(DEV) for k, v in sd.items()}
        eng.gn_partial = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=DEV)
        eng.launches = []
        from reface_amd.gnfuse import ProducerTracker
        eng.tracker, eng.gn_fuse, eng.gn_fused = ProducerTracker(), True, 0
        eng.pool.on_put = eng.tracker.forget
        xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        y = eng._res(*shape_args[:1], xin, *shape_args[1:]) if kind == "res" else eng._attn(shape_args[0], xin, shape_args[1])
        ops.run(eng.launches)
        torch.cuda.synchronize()
        return y.permute(0, 3, 1, 2)

    for tag, cin, cout, hw in (("a", 512, 512, 16), ("b", 512, 256, 16), ("c", 128, 128, 32)):
        for mode, lim in (("f32", 1e-4), ("bf16x3", 3e-4)):
            y = run_block("res", lambda s: P._vae_res(s, "r", cin, cout), 56, rnd((1, cin, hw, hw), 42), "r", cin, cout, mode=mode)
            assert maxerr(y, g[f"res_{tag}_y"]) < lim, (tag, mode, maxerr(y, g[f"res_{tag}_y"]))
    y = run_block("attn", lambda s: P._vae_attn(s, "a", 512), 57, rnd((1, 512, 16, 16), 43), "a", 512)
    assert maxerr(y, g["attn_y"]) < 1e-4, maxerr(y, g["attn_y"])
