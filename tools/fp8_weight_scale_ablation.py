#!/usr/bin/env python3
"""BASELINE configs[4] quality: how much of the fp8 mode's image error is the weight SCALE granularity?

The fp8 UNet stores every eligible GEMM weight as e4m3fn bytes with ONE power-of-two scale per output channel (rf_quantize_fp8_rows); the
MX-scaled MFMA could apply an E8M0 scale per 32 K elements for free.  This ablation fake-quantises the same full-width weights under five
schemes and runs them through the fp32-class engine ("f32x3": 2.4e-4 from exact fp32, far below any fp8 effect), so that ONLY the weight
rounding differs; full-width UNet + VAE, seeded weights, B = 2, 64x64 latents, CFG 3.5, S = 50, decoded [0, 1] images against the same
engine on the unquantised weights:

  row_pow2      amax of the row -> smallest power of two with amax / s <= 448          (what the library does)
  row_exact     s = amax / 448 (any fp32 value: the row's largest weight lands exactly on 448)
  blk32_pow2    one power-of-two (E8M0) scale per 32 consecutive K elements of a row    (what the MX MFMA applies for free)
  blk32_exact   s = block amax / 448 per 32 K elements                                  (upper bound of block scaling)
  row_pow2_e5m2 the row scheme with e5m2 (2 mantissa bits) -- the direction in which mantissa bits, not scales, decide
  row_pow2_3x3_convs_only   mixed precision: only the 3x3 convolutions (85 % of the FLOPs) quantised

  python tools/fp8_weight_scale_ablation.py > profiles/rNN_fp8_weight_scale_ablation.json
"""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reface_amd import ops  # noqa: E402
from reface_amd.ddim import DDIMSampler  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
unet, vae, ldm, cpu_sd = bench.build_models(torch.float32, dev, 0, 1, True)
usd = cpu_sd[0]
B, h, S, scale = 2, 64, 50, 3.5
x_T, z_inp, mask, c, uc = bench.synthetic_inputs(B, h, 4242, dev)
ENGINE = "f32x3"


def eligible(k, v):
    """The tensors the fp8 engine quantises (tests/test_fullsize_gpu.py::_dequantised_state_dict)."""
    if not k.endswith(".weight") or v.dim() < 2 or k.startswith("time_embed") or "emb_layers" in k or "attn2" in k or k == "out.2.weight":
        return False
    w2 = v.reshape(v.shape[0], -1)
    cin = v.shape[1] if v.dim() == 4 and v.shape[-1] == 3 else None
    return ops.fp8_eligible(w2.shape[1], cin)


def pow2_ceil(x):
    return torch.exp2(torch.ceil(torch.log2(x)))


def fake_quant(v, scheme):
    """v: weight in the reference layout ([Cout, Cin, 3, 3] / [Cout, Cin, 1, 1] / [N, K]) -> the values an fp8 store under `scheme` represents.
    32-blocks run over consecutive INPUT CHANNELS (the engine's K order inside a filter tap)."""
    fdt = torch.float8_e5m2 if scheme.endswith("e5m2") else torch.float8_e4m3fn
    fmax = 57344.0 if scheme.endswith("e5m2") else 448.0
    w = v.float()
    if w.dim() == 4:
        w = w.permute(0, 2, 3, 1)                      # [Cout, kh, kw, Cin]
    shp = w.shape
    if scheme.startswith("row"):
        g = w.reshape(shp[0], -1)
    else:
        assert shp[-1] % 32 == 0, shp
        g = w.reshape(-1, 32)
    amax = g.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    s = amax / fmax
    if "pow2" in scheme:
        s = pow2_ceil(s)
    q = (g / s).to(fdt).float() * s
    q = q.reshape(shp)
    if v.dim() == 4:
        q = q.permute(0, 3, 1, 2)
    return q.reshape(v.shape).contiguous()


def run(sd):
    unet.load_state_dict(sd, strict=True)
    unet.to(dev)
    unet.set_compute_dtype(ENGINE)
    sampler = DDIMSampler(ldm)
    samples, _ = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False, unconditional_guidance_scale=scale,
                                unconditional_conditioning=uc, eta=0.0, x_T=x_T, test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
    x = vae.decode(samples, inv_scale=1.0 / 0.18215)
    img = torch.empty_like(x)
    ops.to_image(x, img)()
    eng = unet.engine(2 * B, h, h, uniform_t=True, cfg_pair=True)
    xin = torch.cat([x_T, z_inp, mask], 1)
    ops.nchw_to_nhwc(torch.cat([xin, xin]).contiguous(), eng.x_in)()
    eng.set_context(torch.cat([uc, c]))
    eng.set_timesteps(torch.full((1,), 481.0, device=dev))
    eng.run()
    eps = eng.eps.clone()
    torch.cuda.synchronize()
    unet._engines.clear()
    torch.cuda.empty_cache()
    return img.double().cpu(), samples.double().cpu(), eps.double().cpu()


def dist(a, r):
    mse = ((a[0] - r[0]) ** 2).mean().item()
    return {"image_max_abs": (a[0] - r[0]).abs().max().item(), "image_mean_abs": (a[0] - r[0]).abs().mean().item(),
            "image_psnr_db": 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf"),
            "latent_rel_l2": ((a[1] - r[1]).norm() / r[1].norm()).item(),
            "one_evaluation_eps_rel_l2": ((a[2] - r[2]).norm() / r[2].norm()).item()}


ref = run(usd)
names = [k for k, v in usd.items() if eligible(k, v)]
out = {"setup": f"full-width UNet / VAE, seeded weights, B = {B}, {8 * h}x{8 * h}, S = {S}, CFG {scale}; engine {ENGINE} (fp32-class) on fake-quantised weights; "
                f"{len(names)} weight tensors quantised; reference = the same engine on the unquantised weights"}
for scheme in ("row_pow2", "row_exact", "blk32_pow2", "blk32_exact", "row_pow2_e5m2", "row_pow2_3x3_convs_only"):
    sd = dict(usd)
    werr = 0.0
    wnorm = 0.0
    for k in names:
        if scheme.endswith("3x3_convs_only") and not (usd[k].dim() == 4 and usd[k].shape[-1] == 3):
            continue          # mixed precision: only the 3x3 convolutions (85 % of the FLOPs) in fp8, every projection in bf16-or-better
        sd[k] = fake_quant(usd[k], scheme.replace("_3x3_convs_only", ""))
        werr += (sd[k].double() - usd[k].double()).pow(2).sum().item()
        wnorm += usd[k].double().pow(2).sum().item()
    d = dist(run(sd), ref)
    d["weight_rel_l2"] = math.sqrt(werr / wnorm)
    out[scheme] = d
    print(f"[{scheme}] {d}", file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
