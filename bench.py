#!/usr/bin/env python3
"""bench.py -- REFace hot path on MI355X: 512x512, 50-step DDIM images/sec (BASELINE.json metric).

One "step" = one batch of B synthetic image pairs through the timed region of SURVEY.md section 8d:
  50 x [pack 9-ch input x2 (CFG) -> UNet on 2B samples -> CFG + DDIM update]  +  fp32 KL-VAE decode + clamp.
Inputs (seeded, already resident in HBM): x_T ~ N(0,1), masked-image latent, ellipse keep-mask,
c ~ N(0,1) [B,1,768], learned-uncond vector, scale 3.5, eta 0.  Weights: seeded random init of the exact
REFace architecture (859.5 M-param UNet, 83.7 M-param VAE) -- no checkpoint is obtainable offline.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events per launch on the launch
stream; `cpu_baseline` times the CPU oracle (oracle/) on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_UNET_64 = 796.94e9          # algorithmic FLOP per sample per UNet evaluation, latent 64x64 (SURVEY.md 8d / BASELINE.md 2)
F_VAE_DEC_512 = 2514.5e9      # fp32 VAE decode per 512x512 image
PEAK = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA TFLOP/s (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def synthetic_inputs(B, h, seed, device):
    from reface_amd.params import seeded_randn as rnd
    x_T = rnd((B, 4, h, h), seed)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(h), indexing="ij")
    ell = (((yy - h / 2) / (0.30 * h)) ** 2 + ((xx - h / 2) / (0.38 * h)) ** 2) <= 1.0
    mask = (~ell).float()[None, None].repeat(B, 1, 1, 1)          # 1 = keep (test_bench_dataset.py:347)
    z_inp = rnd((B, 4, h, h), seed + 1) * mask
    c = rnd((B, 1, 768), seed + 2)
    uc = rnd((1, 1, 768), 7).repeat(B, 1, 1)
    return [t.to(device) for t in (x_T, z_inp, mask, c, uc)]


def build_models(dtype, device, rank, world, keep_cpu_sd):
    """Full-width UNet + VAE with seeded weights; rank 0 generates, RCCL-broadcasts to the other ranks."""
    import types
    from reface_amd import params as P
    from reface_amd.schedule import ddpm_buffers
    from reface_amd.unet import UNetModel
    from reface_amd.vae import AutoencoderKL
    unet = UNetModel(image_size=32, in_channels=9, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                     num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1,
                     context_dim=768, use_checkpoint=True, legacy=False, compute_dtype=dtype)
    vae = AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                                      ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0),
                        lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)
    cpu_sd = None
    if rank == 0:
        t0 = time.time()
        usd = P.seeded_state_dict(P.unet_param_specs(unet.cfg), 1234)
        vsd = P.seeded_state_dict(P.vae_param_specs(vae.cfg), 55)
        unet.load_state_dict(usd, strict=True)
        vae.load_state_dict(vsd, strict=True)
        if keep_cpu_sd:
            cpu_sd = (usd, vsd)
        log(f"[bench] seeded weights generated in {time.time() - t0:.1f}s")
    unet.to(device)
    vae.to(device)
    if world > 1:
        import torch.distributed as dist
        for mod in (unet, vae):
            ps = [p.data for p in mod.parameters()]
            flat = torch.cat([p.reshape(-1) for p in ps])
            dist.broadcast(flat, 0)                       # one RCCL broadcast per module over xGMI
            off = 0
            for p in ps:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            del flat
    b = ddpm_buffers(1000, 0.00085, 0.0120)
    ldm = types.SimpleNamespace(num_timesteps=1000, betas=b["betas"], alphas_cumprod=b["alphas_cumprod"],
                                alphas_cumprod_prev=b["alphas_cumprod_prev"], device=torch.device(device),
                                model=types.SimpleNamespace(diffusion_model=unet))
    return unet, vae, ldm, cpu_sd


def cpu_baseline(cpu_sd, cores):
    """Bounded CPU sample of the same workload on the host cores, via the oracle (a port, kind='port'):
    ONE CFG DDIM step (UNet on a CFG batch of 2 at latent 64x64, B = 1 image) + ONE fp32 VAE decode of a
    512x512 image; images/sec = 1 / (50 * t_step + t_decode)."""
    from oracle import unet as ounet, vae as ovae
    from reface_amd import params as P
    # pick the thread count that is actually fastest on this host (all logical CPUs oversubscribe badly)
    best, best_t = cores, float("inf")
    xs, ws = torch.randn(2, 320, 64, 64), torch.randn(320, 320, 3, 3)
    for n in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8), 32, 16, 8}):
        if n > cores:
            continue
        torch.set_num_threads(n)
        torch.nn.functional.conv2d(xs, ws, padding=1)
        t0 = time.time()
        for _ in range(3):
            torch.nn.functional.conv2d(xs, ws, padding=1)
        dt = time.time() - t0
        if dt < best_t:
            best, best_t = n, dt
    cores = best
    torch.set_num_threads(cores)
    usd, vsd = cpu_sd
    ucfg, vcfg = P.UNetConfig(), P.VAEConfig()
    plan = P.unet_plan(ucfg)
    x = P.seeded_randn((2, 9, 64, 64), 1)
    t = torch.full((2,), 981, dtype=torch.long)
    c = P.seeded_randn((2, 1, 768), 2)
    with torch.no_grad():
        t0 = time.time()
        ounet.unet_forward(usd, plan, x, t, c)
        t_step = time.time() - t0
        z = P.seeded_randn((1, 4, 64, 64), 3)
        t0 = time.time()
        ovae.decode_first_stage(vsd, vcfg, z)
        t_dec = time.time() - t0
    ips = 1.0 / (50 * t_step + t_dec)
    return {"value": ips, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"B=1: 1 CFG DDIM step (UNet batch 2, latent 64x64) = {t_step:.2f}s scaled x50, + 1 fp32 VAE decode 512x512 = {t_dec:.2f}s"}


def conditioning_line(vae, B, h, device, enc_dtype=torch.float32):
    """the once-per-image conditioning stage as a separate line -- CLIP ViT-L/14 on the reference and on the
    (resized) target, ArcFace IR-SE50 on the reference, fp32 KL-VAE encode of the 512x512 masked target -- fp32, batch B."""
    from reface_amd import params as P
    from reface_amd.encoders import Backbone, FrozenCLIPEmbedder, target_to_clip_input
    clip = FrozenCLIPEmbedder(compute_dtype=enc_dtype)
    clip.load_state_dict(P.seeded_state_dict(P.clip_param_specs(clip.cfg), 88), strict=True)
    arc = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se", compute_dtype=enc_dtype)
    arc.load_state_dict(P.seeded_state_dict(P.arcface_param_specs(), 77), strict=True)
    clip.to(device)
    arc.to(device)
    ref = P.seeded_randn((B, 3, 224, 224), 71).to(device)
    tar = torch.tanh(P.seeded_randn((B, 3, 8 * h, 8 * h), 70)).to(device)

    vae.encode_dtype = None if enc_dtype == torch.float32 else enc_dtype

    def stage():
        z_ref = clip.encode(ref)
        z_tar = clip.encode(target_to_clip_input(tar))
        fid = arc.forward_from_clip_image(ref)
        post = vae.encode(tar)
        return z_ref, z_tar, fid, post

    stage()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = stage()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    vae.encode_dtype = None
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all() and torch.isfinite(out[3].mean).all()
    tag = "fp32" if enc_dtype == torch.float32 else "bf16"
    return {"metric": f"conditioning images/sec (2x CLIP ViT-L/14 + ArcFace IR-SE50 + VAE encode 512x512, {tag})", "value": B / dt,
            "unit": "images/s", "ms_per_batch": dt * 1e3, "batch": B, "algorithmic_gflop_per_image": 1116.7 + 2 * 155.53 + 12.59}


def pmc_traffic(family):
    """HBM-side bytes per launch of a kernel family (read + write) from the newest committed PMC pass, or None.
    The counters cannot be collected inside this process: tools/pmc_traffic.sh runs this same command under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, FETCH_SIZE doubled as the gfx950 note in the
    microarchitecture guide prescribes) and the summary is committed as profiles/rNN_pmc_hbm_traffic.json."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_hbm_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f).get(family)
        return None if d is None else d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch_uncalibrated"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="image pairs per GPU per step")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--latent", type=int, default=64, help="latent side (64 = 512x512 images)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--scale", type=float, default=3.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-conditioning", action="store_true", help="skip the separate conditioning-stage (encoders) throughput line")
    ap.add_argument("--profile-json", default=None, help="write the per-kernel-family table here")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from reface_amd import ops
    from reface_amd.ddim import DDIMSampler
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    unet, vae, ldm, cpu_sd = build_models(dtype, device, rank, world, want_cpu)
    sampler = DDIMSampler(ldm)
    B, h, S = args.batch, args.latent, args.ddim_steps
    x_T, z_inp, mask, c, uc = synthetic_inputs(B, h, 42 + rank, device)
    img_out = torch.empty((B, 3, 8 * h, 8 * h), dtype=torch.float32, device=device)

    def one_batch():
        samples, _ = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, h, h], verbose=False,
                                    unconditional_guidance_scale=args.scale, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                    test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": mask})
        x = vae.decode(samples, inv_scale=1.0 / 0.18215)
        ops.to_image(x, img_out)()
        return img_out

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_batch()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_batch()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all(), "non-finite output image"
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    result = {
        "metric": "512x512 50-step DDIM images/sec", "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: {8*h}x{8*h}, {S} DDIM steps, CFG scale {args.scale}, batch {B} per GPU, "
                               f"{args.dtype} UNet + fp32 VAE decode, seeded random-init REFace weights",
                   "batch_per_gpu": B, "global_batch": B * world, "ddim_steps": S, "latent": h, "parallelism": f"dp{world} (pairs sharded, no collective in the step loop)"},
    }

    if rank == 0 and not args.no_roofline:
        from reface_amd import profiler
        plan = list(sampler._plans.values())[0]
        log(f"[bench] UNet launches per DDIM step: {len(plan['step'])} (GroupNorm statistics fused into GEMM epilogues: {plan['eng'].gn_fused} of 61)")
        timed = profiler.time_launches(plan["step"], reps=3)
        fam = profiler.summarize(timed)
        step_ms = sum(ms for _, ms in timed)
        dec = vae._engine("dec", B, h, h)
        dtimed = profiler.time_launches(dec.launches, reps=2)
        dfam = profiler.summarize(dtimed)
        dec_ms = sum(ms for _, ms in dtimed)
        key = f"rf_conv_gemm[{args.dtype}]"
        dom = fam[key]
        nb = 2 * B
        unet_alg = F_UNET_64 * nb * (h / 64.0) ** 2 if h == 64 else None
        roof = {"bound": "mfma", "kernel": key, "achieved": dom["tflops_per_s"], "peak": PEAK[args.dtype], "unit": "TFLOP/s",
                "frac": dom["tflops_per_s"] / PEAK[args.dtype], "traffic": pmc_traffic(key),
                "launches_per_ddim_step": dom["calls"], "avg_launch_us": dom["ms"] / dom["calls"] * 1e3,
                "alg_flop_per_ddim_step": dom["flops"], "ddim_step_ms_sum_of_kernels": step_ms}
        if unet_alg:
            roof["unet_mfma_util_whole_step"] = unet_alg / (step_ms * 1e-3) / 1e12 / PEAK[args.dtype]
        result["roofline"] = roof
        result["breakdown"] = {
            "ddim_step_ms": step_ms, "vae_decode_ms": dec_ms,
            "unet_step": {k: {"calls": v["calls"], "ms": round(v["ms"], 4), "tflops_per_s": round(v["tflops_per_s"], 2)} for k, v in fam.items()},
            "vae_decode": {k: {"calls": v["calls"], "ms": round(v["ms"], 4), "tflops_per_s": round(v["tflops_per_s"], 2)} for k, v in dfam.items()},
        }
        log("[bench] per-family (one DDIM step):")
        for k, v in fam.items():
            log(f"   {k:28s} calls {v['calls']:4d}  {v['ms']:9.3f} ms  {v['tflops_per_s']:8.1f} TFLOP/s")
        log(f"[bench] one DDIM step = {step_ms:.2f} ms (sum of kernels); VAE decode = {dec_ms:.2f} ms")
        for k, v in dfam.items():
            log(f"   dec {k:24s} calls {v['calls']:4d}  {v['ms']:9.3f} ms  {v['tflops_per_s']:8.1f} TFLOP/s")
        if args.profile_json:
            def row(l, ms):
                r = {"name": l.name, "family": profiler.launch_family(l), "ms": ms, "flop": profiler.gemm_flops(l) + profiler.attention_flops(l)}
                if l.fn.__name__ == "rf_conv_gemm":
                    d = l.keep[0]
                    r.update(M=d.M, N=d.N, K=d.K, act=d.act, batch=d.batch)
                return r
            with open(args.profile_json, "w") as f:
                json.dump({"families": fam, "vae_families": dfam, "step_launches": [row(l, ms) for l, ms in timed],
                           "vae_launches": [row(l, ms) for l, ms in dtimed]}, f, indent=1)
    if rank == 0 and world == 1 and not args.no_conditioning:
        try:
            result["conditioning"] = conditioning_line(vae, B, h, device)
            log(f"[bench] conditioning stage: {result['conditioning']['value']:.1f} images/s ({result['conditioning']['ms_per_batch']:.1f} ms per batch of {B})")
            if dtype == torch.bfloat16:         # what the CLI's --precision bf16 runs: bf16 towers and VAE encoder (decode stays fp32)
                result["conditioning_bf16"] = conditioning_line(vae, B, h, device, torch.bfloat16)
                log(f"[bench] conditioning stage, bf16: {result['conditioning_bf16']['value']:.1f} images/s")
        except Exception as e:        # the headline number must not depend on this side line
            result["conditioning"] = {"error": repr(e)}
    if want_cpu:
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        result["cpu_baseline"] = cpu_baseline(cpu_sd, cores)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
