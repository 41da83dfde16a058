#!/usr/bin/env python3
"""Selected-faces swap CLI on the MI355X-native engines: every source face in ``<Base_dir>/source_cropped`` onto every target in
``<Base_dir>/target_cropped`` -- the sampling stage of the reference's scripts/inference_swap_selected.py:516-762.

The reference script has two stages.  Stage 1 (:440-512) aligns and crops the raw ``--target_folder`` / ``--src_folder`` images
(dlib / FFHQ alignment) and writes face-parsing label maps (BiSeNet, ``--faceParsing_ckpt``) into ``<Base_dir>/{target_cropped,
mask_frames,source_cropped,source_mask}``; those models are outside the scope of this build, so this CLI expects
that tree to exist (the reference's stage 1, or any tool writing ``<i>.png`` crops + label maps, produces it).  Stage 2 -- the
sampling loop -- is the same batch body as the test bench (reface_amd/pipeline.py) with ONE source face repeated over the batch
(:649-653); outputs as the reference writes them: ``<outdir>/results/<s>/<id>.png``, ``<outdir>/grid/<s>/grid-<id>.png``,
``<outdir>/<s>/<id>_{mask,GT,inpaint,ref}.png`` and the decoded ``pred_x0`` intermediates in ``<outdir>/model_outputs``.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402
from reface_amd import config as rcfg  # noqa: E402
from reface_amd import output as O  # noqa: E402
from inference_test_bench import load_model_from_config  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--prompt", type=str, nargs="?", default="a photograph of an astronaut riding a horse")
    p.add_argument("--outdir", type=str, nargs="?", default="results_video/debug")
    p.add_argument("--Base_dir", type=str, nargs="?", default="results_video")
    p.add_argument("--skip_grid", action="store_true")
    p.add_argument("--skip_save", action="store_true")
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--plms", action="store_true")
    p.add_argument("--laion400m", action="store_true")
    p.add_argument("--fixed_code", action="store_true", default=False)
    p.add_argument("--Start_from_target", action="store_true")
    p.add_argument("--only_target_crop", action="store_true", default=True)
    p.add_argument("--target_start_noise_t", type=int, default=1000)
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=2)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_samples", type=int, default=12)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=5)
    p.add_argument("--target_folder", type=str, default="examples/faceswap/Andy2.mp4")
    p.add_argument("--src_folder", type=str, default="examples/faceswap/source.jpg")
    p.add_argument("--src_image_mask", type=str, default=None)
    p.add_argument("--from-file", type=str, default=None)
    p.add_argument("--config", type=str, default="configs/debug.yaml")
    p.add_argument("--ckpt", type=str, default="models/REFace/checkpoints/last.ckpt")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--rank", type=int, default=0)
    p.add_argument("--precision", type=str, choices=["full", "fullx3", "autocast", "bf16", "fp16", "fp8"], default="autocast")
    p.add_argument("--faceParser_name", default="default", type=str)
    p.add_argument("--faceParsing_ckpt", type=str, default="Other_dependencies/face_parsing/79999_iter.pth")
    p.add_argument("--segnext_config", default="", type=str)
    p.add_argument("--save_vis", action="store_true")
    p.add_argument("--seg12", default=True, action="store_true")
    # additions (not in the reference)
    p.add_argument("--dump_tensors", type=str, default=None, help="directory for per-batch .npz dumps of the tensors fed to / produced by the engines (tests)")
    p.add_argument("--clip_vision_config", type=str, default=None, help="JSON dict overriding the CLIP ViT dims (tests)")
    p.add_argument("--num_workers", type=int, default=4)
    return p


def main(argv=None):
    opt = build_parser().parse_args(argv)
    print(opt)
    torch.manual_seed(opt.seed)
    np.random.seed(opt.seed)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    base = opt.Base_dir
    tc, tm, sc, sm = (os.path.join(base, d) for d in ("target_cropped", "mask_frames", "source_cropped", "source_mask"))
    missing = [d for d in (tc, tm, sc, sm) if not os.path.isdir(d) or not os.listdir(d)]
    if missing:
        raise SystemExit("inference_swap_selected: stage 1 of the reference (face alignment + BiSeNet parsing of --target_folder / "
                         "--src_folder, inference_swap_selected.py:440-512) is outside this build's scope; prepare\n  "
                         + "\n  ".join(missing) + "\nwith <i>.png crops and face-parsing label maps (the reference's stage 1 writes exactly this tree).")
    config = rcfg.load(opt.config)
    if opt.clip_vision_config:
        import json
        config.model.params.cond_stage_config["params"] = {"vision_config": json.loads(opt.clip_vision_config)}
    model = load_model_from_config(config, opt.ckpt)
    device = torch.device("cuda")
    if opt.precision in ("autocast", "bf16"):
        model.set_compute_dtype(torch.bfloat16, encoders=True)
    elif opt.precision == "fp16":                   # the bf16 mode's UNet kernels on fp16 storage / MFMA (same speed, ~17 dB closer to the exact-fp32 image);
        model.set_compute_dtype(torch.float16)      # the towers and the VAE encoder stay fp32: this mode is chosen for its distance to "full"
    elif opt.precision == "fp8":
        model.set_compute_dtype("fp8", encoders=True)
    elif opt.precision == "fullx3":                 # the fast form of "full": fp32 storage, split-bf16 GEMM operands (3 bf16 MFMA passes)
        model.set_compute_dtype("f32x3")
    if opt.plms:
        from ldm.models.diffusion.plms import PLMSSampler
        sampler = PLMSSampler(model)
    else:
        sampler = DDIMSampler(model)
    from reface_amd.data import VideoDataset, load_source_reference
    from reface_amd.pipeline import SwapRunner
    runner = SwapRunner(model, sampler, opt)
    outpath = opt.outdir
    model_out = os.path.join(outpath, "model_outputs")
    os.makedirs(model_out, exist_ok=True)
    test_args = dict(config.data.params.test.params)
    n_done = 0
    with torch.no_grad(), model.ema_scope():
        for s_idx, im in enumerate(sorted(os.listdir(sc))):
            dirs = {"results": os.path.join(outpath, "results", str(s_idx)), "grid": os.path.join(outpath, "grid", str(s_idx)),
                    "samples": os.path.join(outpath, str(s_idx))}
            for d in dirs.values():
                os.makedirs(d, exist_ok=True)
            ref1 = load_source_reference(os.path.join(sc, im), os.path.join(sm, im), test_args["preserve_mask_src_FFHQ"]).to(device)
            ds = VideoDataset(data_path=tc, mask_path=tm, **test_args)
            loader = torch.utils.data.DataLoader(ds, batch_size=opt.n_samples, num_workers=opt.num_workers, pin_memory=True, shuffle=False,
                                                 drop_last=False)
            start_code = None
            if opt.fixed_code:
                start_code = torch.randn([opt.n_samples, opt.C, opt.H // opt.f, opt.W // opt.f], device=device)
            for test_batch, prior, kw, ids in loader:
                if opt.Start_from_target:
                    start_code = runner.start_from_target(test_batch)        # `use_prior = False` here (:592)
                kw = {n: kw[n].to(device, non_blocking=True) for n in kw}
                B = test_batch.shape[0]
                ref = ref1.repeat(B, 1, 1, 1)
                x_img, inter = runner.run_batch(test_batch, kw, ref, start_code=start_code)
                from PIL import Image
                for k, px0 in enumerate(inter["pred_x0"]):       # :686-696: first sample of every logged pred_x0, decoded
                    img0 = runner.decode01(px0)[0].cpu().numpy()
                    Image.fromarray(O.to_u8_hwc(img0)).save(os.path.join(model_out, f"_intermediate_{k}.png"))
                n_done += B
                if opt.skip_save:
                    continue
                ref_np = runner.resized_reference(ref, opt.H, opt.W).cpu().numpy()
                res, tgt, inp, msk = x_img.cpu().numpy(), test_batch.float().numpy(), kw["inpaint_image"].cpu().numpy(), kw["inpaint_mask"].float().cpu().numpy()
                for i, sid in enumerate(ids):
                    arrs = O.compose(res[i], tgt[i], inp[i], msk[i], ref_np[i])
                    Image.fromarray(arrs["result"]).save(os.path.join(dirs["results"], sid + ".png"))
                    Image.fromarray(arrs["grid"]).save(os.path.join(dirs["grid"], "grid-" + sid + ".png"))
                    for nm in ("mask", "GT", "inpaint", "ref"):
                        Image.fromarray(arrs[nm]).save(os.path.join(dirs["samples"], f"{sid}_{nm}.png"))
    torch.cuda.synchronize()
    print(f"Your samples are ready and waiting for you here: \n{outpath} \n ({n_done} images)\nEnjoy.")
    return n_done


if __name__ == "__main__":
    main()
