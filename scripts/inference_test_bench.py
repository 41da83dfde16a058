#!/usr/bin/env python3
"""Test-bench CLI of REFace on the MI355X-native engines.

Same flags, loop and on-disk outputs as the reference's scripts/inference_test_bench.py:145-566:
  <outdir>/results/<id>.png, <outdir>/grid/grid-<id>.png (4-panel make_grid: GT, inpaint, ref, result),
  <outdir>/samples/<id>_{mask,GT,inpaint,ref}.png  -- composed by reface_amd/output.py, byte-compatible with :500-553
Differences, all at the edges of the scope table:
  * no module-import network access (the reference loads an HF safety checker it never calls);
  * ``--dataset synthetic`` (seeded items of the dataset tensor contract) is available because no dataset is
    reachable offline; CelebA / FFHQ / FF++ folder readers are the "next" row 8f.1;
  * ``--ckpt none`` runs on seeded random weights (no checkpoint is reachable offline);
  * ``--precision bf16`` selects the throughput mode: bf16 MFMA UNet, CLIP / ArcFace towers and VAE encoder, fp32 VAE decode
    (``full`` = exact-fp32 MFMA everywhere, the parity mode; ``autocast`` maps to bf16; ``fp16`` = the same UNet kernels on fp16 storage and the
    fp16 MFMA at the bf16 rate -- three more mantissa bits per operand -- with fp32 towers / VAE encoder);
  * one process per GPU under torch.distributed.run shards the pairs ``rank::world`` (weights broadcast once).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402
from ldm.util import instantiate_from_config  # noqa: E402
from reface_amd import config as rcfg  # noqa: E402
from reface_amd import ops  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--prompt", type=str, nargs="?", default="a photograph of an astronaut riding a horse")
    p.add_argument("--device_ID", type=int, default=5)
    p.add_argument("--outdir", type=str, nargs="?", default="results/debug")
    p.add_argument("--skip_grid", action="store_true")
    p.add_argument("--skip_save", action="store_true", help="do not save individual samples. For speed measurements.")
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--plms", action="store_true")
    p.add_argument("--laion400m", action="store_true")
    p.add_argument("--fixed_code", action="store_true")
    p.add_argument("--Guidance", action="store_true")
    p.add_argument("--Start_from_target", action="store_true")
    p.add_argument("--target_start_noise_t", type=int, default=1000)
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=2)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_samples", type=int, default=5)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=5)
    p.add_argument("--dataset", type=str, default="CelebA", help="CelebA | FFHQ | FF++ | synthetic")
    p.add_argument("--dataset_dir", type=str, default="dataset/FaceData/CelebAMask-HQ")
    p.add_argument("--from-file", type=str)
    p.add_argument("--config", type=str, default="models/REFace/configs/project_ffhq.yaml")
    p.add_argument("--ckpt", type=str, default="models/REFace/checkpoints/last.ckpt")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--rank", type=int, default=0)
    p.add_argument("--precision", type=str, choices=["full", "fullx3", "autocast", "bf16", "fp16", "fp8", "fp8c"], default="full")
    # additions (not in the reference)
    p.add_argument("--n_items", type=int, default=8, help="number of synthetic pairs (--dataset synthetic)")
    p.add_argument("--dump_tensors", type=str, default=None, help="directory for per-batch .npz dumps of the tensors fed to / produced by the engines (tests)")
    p.add_argument("--clip_vision_config", type=str, default=None, help="JSON dict overriding the CLIP ViT dims (tests)")
    p.add_argument("--num_workers", type=int, default=4, help="DataLoader workers of the folder readers (reference: 4)")
    p.add_argument("--png_level", type=int, default=None, help="zlib level of the PNG files (default: PIL's, 6 = the reference's files byte for byte; "
                   "lower = the same pixels in larger files for less host time)")
    p.add_argument("--fast_aux_png", action="store_true", help="samples/ and grid/ PNGs at zlib level 1 (same pixels, larger files); results/<id>.png keeps --png_level "
                   "(multi-GPU runs: the host's PNG encoding is what caps 8 processes per node)")
    p.add_argument("--gpu_prep", action="store_true", help="folder readers hand over uint8 arrays; normalise / mask / resize run on the GPU")
    return p


def load_model_from_config(config, ckpt, verbose=False):
    """inference_test_bench.py:96-113.  ``ckpt == 'none'``: seeded random weights of the same architecture."""
    model = instantiate_from_config(config.model)
    if ckpt and ckpt.lower() != "none":
        print(f"Loading model from {ckpt}")
        pl_sd = torch.load(ckpt, map_location="cpu")
        if "global_step" in pl_sd:
            print(f"Global Step: {pl_sd['global_step']}")
        m, u = model.load_state_dict(pl_sd["state_dict"], strict=False)
        if verbose:
            print("missing keys:", m, "\nunexpected keys:", u)
        model.check_engine_weights(m)            # a tensor the engines read but the checkpoint lacks is an error, not zeros
    else:
        from reface_amd import params as P
        sd = {}
        unet, vae, clip = model.model.diffusion_model, model.first_stage_model, model.cond_stage_model
        sd.update(P.seeded_state_dict(P.unet_param_specs(unet.cfg), 1234, "model.diffusion_model."))
        sd.update(P.seeded_state_dict(P.vae_param_specs(vae.cfg), 55, "first_stage_model."))
        sd.update(P.seeded_state_dict(P.clip_param_specs(clip.cfg), 88, "cond_stage_model."))
        sd.update(P.seeded_state_dict(P.arcface_param_specs(), 77, "face_ID_model.facenet."))
        sd.update(P.seeded_state_dict(P.cond_head_specs(), 9))
        m, u = model.load_state_dict(sd, strict=False)
        assert not u, u[:5]
        print(f"[reface_amd] --ckpt none: seeded random weights ({len(sd)} tensors)")
    model.cuda()
    model.eval()
    return model


def main(argv=None):
    opt = build_parser().parse_args(argv)
    print(opt)
    if opt.laion400m or opt.Guidance:
        raise NotImplementedError("--laion400m / --Guidance are outside the REFace sampling path (SURVEY.md section 2)")
    torch.manual_seed(opt.seed)
    np.random.seed(opt.seed)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)

    config = rcfg.load(opt.config)
    if opt.clip_vision_config:
        import json
        config.model.params.cond_stage_config["params"] = {"vision_config": json.loads(opt.clip_vision_config)}
    model = load_model_from_config(config, opt.ckpt)
    device = torch.device("cuda")
    model = model.to(device)
    if opt.precision in ("autocast", "bf16"):
        model.set_compute_dtype(torch.bfloat16, encoders=True)
    elif opt.precision == "fp16":                   # the bf16 mode's UNet kernels on fp16 storage / MFMA (same speed, ~17 dB closer to the exact-fp32 image);
        model.set_compute_dtype(torch.float16)      # the towers and the VAE encoder stay fp32: this mode is chosen for its distance to "full"
    elif opt.precision in ("fp8", "fp8c"):          # BASELINE configs[4]: fp8 x fp8 UNet GEMMs ("fp8c": the 3x3 convolutions only, bf16 projections -- 8 dB closer to fp32)
        model.set_compute_dtype(opt.precision, encoders=True)
    elif opt.precision == "fullx3":                 # the fast form of "full": fp32 storage, split-bf16 GEMM operands (3 bf16 MFMA passes)
        model.set_compute_dtype("f32x3")
    if world > 1:
        import torch.distributed as dist
        from reface_amd.multigpu import broadcast_module
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        broadcast_module(model, 0)                  # a few flat RCCL broadcasts (one buffer per dtype), not one per tensor
    if opt.plms:                                    # inference_test_bench.py:337-339
        from ldm.models.diffusion.plms import PLMSSampler
        sampler = PLMSSampler(model)
    else:
        sampler = DDIMSampler(model)

    outpath = opt.outdir
    sample_path, result_path, grid_path = (os.path.join(outpath, d) for d in ("samples", "results", "grid"))
    for d in (outpath, sample_path, result_path, grid_path):
        os.makedirs(d, exist_ok=True)
    batch_size = opt.n_samples

    if opt.dataset == "synthetic":
        from reface_amd.data import SyntheticPairs, shard_indices
        full = SyntheticPairs(n=opt.n_items, image_size=opt.H, seed=opt.seed)
        test_dataset = torch.utils.data.Subset(full, shard_indices(len(full), rank, world))
    elif opt.dataset in ("CelebA", "FFHQ", "FF++"):          # inference_test_bench.py:374-381: data.params.test.params of the config
        from reface_amd.data import CelebAdataset, FFdataset, FFHQdataset, shard_indices
        test_args = {}
        try:
            test_args = dict(config.data.params.test.params)
        except (AttributeError, KeyError, TypeError):
            pass
        if opt.dataset_dir is not None and "dataset_dir" not in test_args:
            test_args["dataset_dir"] = opt.dataset_dir
        test_args.setdefault("state", "test")
        if opt.gpu_prep:
            test_args["raw"] = "full"           # decode only: the 224x224 source resize (cv2 INTER_LINEAR arithmetic) runs on the GPU too
        if opt.dataset == "FF++" and opt.dataset_dir is not None:       # inference_test_bench.py:383: the flag overrides the config
            test_args["dataset_dir"] = opt.dataset_dir
        full = {"CelebA": CelebAdataset, "FFHQ": FFHQdataset, "FF++": FFdataset}[opt.dataset](**test_args)
        test_dataset = torch.utils.data.Subset(full, shard_indices(len(full), rank, world))
    else:
        raise NotImplementedError(f"--dataset {opt.dataset}: CelebA, FFHQ, FF++ and synthetic are available")
    # inference_test_bench.py:386-391: 4 worker processes, pinned host memory (the PIL decode / resize of batch i+1 overlaps batch i)
    nw = 0 if opt.dataset == "synthetic" else opt.num_workers
    collate = None
    if opt.gpu_prep and opt.dataset != "synthetic":
        from reface_amd.data import raw_collate
        collate = raw_collate                       # full-size sources of different sizes stay a list
    loader = torch.utils.data.DataLoader(test_dataset, batch_size=batch_size, num_workers=nw, pin_memory=True, shuffle=False, collate_fn=collate,
                                         drop_last=False, persistent_workers=False)

    start_code = None
    if opt.fixed_code:
        start_code = torch.randn([opt.n_samples, opt.C, opt.H // opt.f, opt.W // opt.f], device=device)

    n_done, n_batches, t_start = 0, 0, time.time()
    host = [None, None]
    writer = None
    if not opt.skip_save:
        from reface_amd.output import OutputWriter
        from reface_amd.output import default_writer_threads
        # PNG encodes run on worker threads, bounded by this process's share of the host (8 processes per node share it).  zlib level: PIL's
        # default (6) = the reference's files byte for byte, whatever the number of processes (--png_level / RF_PNG_LEVEL change it: same pixels,
        # other bytes).  --fast_aux_png writes the four samples/ panels and the grid/ file at level 1 and keeps results/<id>.png -- the file a user
        # compares -- at the reference's level: photo-like panels at level 6 cost 2.4 s of host time per batch of 8 with 8 processes at once against
        # 1.03 s at level 1 (tools/host_scaling_probe.py --natural, profiles/r04d_host_probe_natural*.json); with only results/ at level 6 the
        # probe's estimate is (5 x level-1 + 1 x level-6) / 6 of that -- see README "Known limits" for the 8-GPU host cap.
        lvl = os.environ.get("RF_PNG_LEVEL")
        level = int(lvl) if lvl else opt.png_level
        writer = OutputWriter(outpath, skip_grid=opt.skip_grid, threads=default_writer_threads(world), compress_level=level,
                              aux_compress_level=1 if opt.fast_aux_png else None)
    host_compose = os.environ.get("RF_HOST_COMPOSE") == "1"          # debug: the reference's per-image float passes on the host (round-3 form)
    def with_landmark_prefetch(batches):
        """Yield (batch, landmarks136 or None): the dlib landmarks of batch i+1 are detected on a worker thread while the GPU works
        on batch i -- the only serial CPU stage inside the reference's batch loop (ddpm.py:1068-1099)."""
        if not model.Landmark_cond:
            for bt in batches:
                yield bt, None
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            it = iter(batches)
            try:
                cur = next(it)
            except StopIteration:
                return
            fut = pool.submit(model.detect_landmarks, cur[0])
            while True:
                try:
                    nxt = next(it)
                except StopIteration:
                    nxt = None
                lm = fut.result()
                if nxt is not None:
                    fut = pool.submit(model.detect_landmarks, nxt[0])
                yield cur, lm
                if nxt is None:
                    return
                cur = nxt

    from reface_amd.pipeline import SwapRunner
    runner = SwapRunner(model, sampler, opt)
    prep = None
    if opt.gpu_prep and opt.dataset != "synthetic":
        from reface_amd.prep import DevicePrep
        prep = DevicePrep(full.remove_tar, full.preserve_src, full.gray_outer_mask)

    def unpack(batches):
        """raw uint8 items -> the dataset's tensor contract, prepared on the GPU (--gpu_prep); identity otherwise"""
        for item in batches:
            if prep is None:
                yield item
            else:
                tar_u8, tar_lab, ref_u8, ref_lab, ids = item
                target, kw = prep(tar_u8, tar_lab, ref_u8, ref_lab)
                t_host = target.cpu()
                yield t_host, t_host, kw, ids

    def stage_out(slot, tensors):
        """Enqueue the device -> host copies of one batch into the slot's pinned buffers (allocated on first use), then the slot's event."""
        bufs, ev = slot
        out = {}
        for name, t in tensors.items():
            if not t.is_cuda:
                out[name] = t
                continue
            if name not in bufs or bufs[name].shape[0] < t.shape[0] or bufs[name].shape[1:] != t.shape[1:]:
                bufs[name] = torch.empty(t.shape, dtype=torch.float32).pin_memory()
            bufs[name][:t.shape[0]].copy_(t, non_blocking=True)
            out[name] = bufs[name][:t.shape[0]]
        ev.record()
        return out

    def stage_out_u8(slot, x_img, ref512, target, inpaint, mask):
        """The output panels (+ grid) of the batch as packed uint8 records, composed by ONE kernel on the device (rf_compose_outputs_u8: the
        reference's un_norm / un_norm_clip / make_grid / astype(uint8), inference_test_bench.py:500-552) and copied to the slot's pinned
        buffer by ONE D2H: 7.1 MB per 512x512 image instead of 13 MB of fp32 panels, and no float passes left on the host."""
        from reface_amd import ops
        from reface_amd.output import record_layout
        bufs, ev = slot
        Bc, _, H, W = x_img.shape
        nbytes, _ = record_layout(H, W, with_grid=not opt.skip_grid)
        if "rec_dev" not in bufs or bufs["rec_dev"].shape[0] < Bc or bufs["rec_dev"].shape[1] != nbytes:
            bufs["rec_dev"] = torch.zeros((batch_size, nbytes), dtype=torch.uint8, device=device)      # (zeros: the grid's padding is never written)
            bufs["rec_host"] = torch.empty((batch_size, nbytes), dtype=torch.uint8).pin_memory()
        dev = lambda t: t.to(device, non_blocking=True).float().contiguous()
        ops.compose_outputs_u8(x_img.contiguous(), dev(target), dev(inpaint), dev(mask), ref512.contiguous(), bufs["rec_dev"][:Bc],
                               with_grid=not opt.skip_grid)()
        bufs["rec_host"][:Bc].copy_(bufs["rec_dev"][:Bc], non_blocking=True)
        ev.record()
        return {"records": bufs["rec_host"][:Bc], "H": H, "W": W}

    def flush(pending):
        """The batch whose copies were enqueued one iteration ago: wait for them, hand the arrays to the PNG writer threads."""
        ids, slot, out = pending
        slot[1].synchronize()
        if "records" in out:
            writer.submit_u8(ids, out["records"].numpy(), out["H"], out["W"])
            return
        writer.submit(ids, out["x"].numpy().copy(), out["target"].float().numpy(), out["inpaint"].float().numpy().copy(),
                      out["mask"].float().numpy().copy(), out["ref"].numpy().copy())

    pending = None
    timing = os.environ.get("RF_CLI_TIMING") == "1"      # per-batch host phases on stderr: loader / enqueue / flush (ms)
    t_mark = time.perf_counter()
    with torch.no_grad(), model.ema_scope():
        for (test_batch, prior, test_model_kwargs, segment_id_batch), lm136 in with_landmark_prefetch(unpack(loader)):
            t_load = time.perf_counter()
            if opt.Start_from_target:                   # inference_test_bench.py:414-435: noised target (or prior) latent as x_T
                start_code = runner.start_from_target(prior)      # `use_prior = True` is hard-wired in the reference (:402, :424-429)
            kw_in = test_model_kwargs                   # as the loader (host) or the device prep (GPU) made them
            test_model_kwargs = {n: kw_in[n].to(device, non_blocking=True) for n in kw_in}
            B = test_batch.shape[0]
            ref = test_model_kwargs["ref_imgs"].squeeze(1)
            x_img, _ = runner.run_batch(test_batch, test_model_kwargs, ref, start_code=start_code, landmarks136=lm136)
            # Two host staging slots: the copies of batch i are enqueued right behind its kernels, the host then enqueues batch i+1
            # and only afterwards waits for batch i's copies and hands them to the writer thread -- the GPU goes from one batch
            # straight into the next while the host converts / saves the previous one (tools/cli_idle.sh).
            slot = host[n_batches % 2]
            if slot is None:
                slot = host[n_batches % 2] = ({}, torch.cuda.Event())
            n_done += B
            n_batches += 1
            if not opt.skip_save:
                if host_compose:
                    out = stage_out(slot, {"x": x_img, "ref": runner.resized_reference(ref, opt.H, opt.W), "target": test_batch,
                                           "inpaint": kw_in["inpaint_image"], "mask": kw_in["inpaint_mask"]})
                else:
                    out = stage_out_u8(slot, x_img, runner.resized_reference(ref, opt.H, opt.W), test_batch, test_model_kwargs["inpaint_image"],
                                       test_model_kwargs["inpaint_mask"])
                t_enq = time.perf_counter()
                if pending is not None:
                    flush(pending)
                pending = (list(segment_id_batch), slot, out)
                if timing:
                    t_now = time.perf_counter()
                    print(f"[cli] batch {n_batches}: loader {1e3 * (t_load - t_mark):.0f} ms, enqueue {1e3 * (t_enq - t_load):.0f} ms, "
                          f"flush of the previous batch {1e3 * (t_now - t_enq):.0f} ms", file=sys.stderr, flush=True)
                    t_mark = t_now
            else:
                slot[1].record()
        if pending is not None:
            flush(pending)
    if writer is not None:
        writer.close()
    torch.cuda.synchronize()
    dt = time.time() - t_start
    print(f"Your samples are ready and waiting for you here: \n{outpath} \n ({n_done} images on rank {rank} in {dt:.1f}s)\nEnjoy.")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return n_done


if __name__ == "__main__":
    main()
