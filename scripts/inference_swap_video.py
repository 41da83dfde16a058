#!/usr/bin/env python3
"""Video face-swap CLI on the MI355X-native engines: ONE source face onto every frame crop of a target video -- the sampling stage
of the reference's scripts/inference_swap_video.py:504-700.

The reference script has three stages.  Stage 1 (:430-500) decodes ``--target_video`` with OpenCV, aligns / crops every frame and the
``--src_image`` (dlib / FFHQ alignment, keeping the inverse transforms) and writes face-parsing label maps (BiSeNet); stage 3
(:690-760) warps every swapped crop back into its frame and encodes the mp4 with the original audio (moviepy).  Both are host-side
I/O around models that are outside this build, so this CLI takes stage 1's on-disk product, in the reference's
own layout, and leaves stage 3 to the reference:

  <Base_dir>/<video>cropped_face/<i>.png      aligned 1024^2 crops, one per frame           (:416, written at :489)
  <Base_dir>/<video>mask_frames/<i>.png       their face-parsing label maps                 (:417, :497)
  <outdir>/temp_results/<src>.png             the aligned source crop                       (:456-457)
  <outdir>/temp_results/<basename(src_image)> its label map                                 (:463)

Stage 2 is the test bench's batch body (reface_amd/pipeline.py) with the source repeated over the batch (:619-620) and
``drop_last=True`` (:541: a trailing partial batch of frames is NOT swapped -- kept, it is the reference's behaviour); every
swapped crop is written as the reference does before pasting, resized to 1024^2 (bilinear), to
``<outdir>/model_outputs/<frame id>.png`` (:690-691).
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
    p.add_argument("--fixed_code", action="store_true", default=True)        # (sic: on by default in the video caller, :231-235)
    p.add_argument("--Start_from_target", action="store_true")
    p.add_argument("--only_target_crop", action="store_true", default=True)
    p.add_argument("--target_start_noise_t", type=int, default=1000)
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=2)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_samples", type=int, default=10)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=5)
    p.add_argument("--target_video", type=str, default="examples/faceswap/Andy2.mp4")
    p.add_argument("--src_image", type=str, default="examples/faceswap/source.jpg")
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


def prepared_paths(opt):
    """The reference's names for what its stage 1 leaves on disk (:408-418, :456-463)."""
    video = os.path.basename(opt.target_video).split(".")[0]
    src = os.path.basename(opt.src_image).split(".")[0]
    tmp = os.path.join(opt.outdir, "temp_results")
    return {"frames": os.path.join(opt.Base_dir, video + "cropped_face"), "masks": os.path.join(opt.Base_dir, video + "mask_frames"),
            "src": os.path.join(tmp, src + ".png"), "src_mask": os.path.join(tmp, os.path.basename(opt.src_image))}


def main(argv=None):
    opt = build_parser().parse_args(argv)
    print(opt)
    torch.manual_seed(opt.seed)
    np.random.seed(opt.seed)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    pp = prepared_paths(opt)
    missing = [pp[k] for k in ("frames", "masks") if not os.path.isdir(pp[k]) or not os.listdir(pp[k])]
    missing += [pp[k] for k in ("src", "src_mask") if not os.path.isfile(pp[k])]
    if missing:
        raise SystemExit("inference_swap_video: stage 1 of the reference (frame extraction, face alignment and BiSeNet parsing of --target_video / "
                         "--src_image, inference_swap_video.py:430-500) is outside this build's scope; prepare\n  " + "\n  ".join(missing) +
                         "\n(the reference's stage 1 writes exactly these paths).")
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
    from PIL import Image
    from reface_amd.data import VideoDataset, load_source_reference
    from reface_amd.pipeline import SwapRunner
    runner = SwapRunner(model, sampler, opt)
    model_out = os.path.join(opt.outdir, "model_outputs")
    os.makedirs(model_out, exist_ok=True)
    os.makedirs(os.path.join(opt.outdir, "results"), exist_ok=True)      # stage 3 of the reference fills it (pasted frames)
    test_args = dict(config.data.params.test.params)
    ref1 = load_source_reference(pp["src"], pp["src_mask"], test_args["preserve_mask_src_FFHQ"]).to(device)
    ds = VideoDataset(data_path=pp["frames"], mask_path=pp["masks"], **test_args)
    loader = torch.utils.data.DataLoader(ds, batch_size=opt.n_samples, num_workers=opt.num_workers, pin_memory=True, shuffle=False, drop_last=True)
    start_code = None
    if opt.fixed_code:      # ONE latent, repeated over the batch (:549-552) -- not one per sample as in the selected-swap caller
        start_code = torch.randn([opt.C, opt.H // opt.f, opt.W // opt.f], device=device).unsqueeze(0).repeat(opt.n_samples, 1, 1, 1)
    n_done = 0
    with torch.no_grad(), model.ema_scope():
        for test_batch, prior, kw, ids in loader:
            if opt.Start_from_target:
                start_code = runner.start_from_target(test_batch)        # `use_prior = False` (:555)
            kw = {n: kw[n].to(device, non_blocking=True) for n in kw}
            B = test_batch.shape[0]
            x_img, _ = runner.run_batch(test_batch, kw, ref1.repeat(B, 1, 1, 1), start_code=start_code)
            n_done += B
            if opt.skip_save:
                continue
            res = x_img.cpu().numpy()
            for i, sid in enumerate(ids):
                Image.fromarray(O.to_u8_hwc(res[i])).resize((1024, 1024), Image.BILINEAR).save(os.path.join(model_out, sid + ".png"))
    torch.cuda.synchronize()
    print(f"Swapped crops of {n_done} frames are in {model_out}; the reference's stage 3 (paste back with the inverse alignment of every frame, "
          f"mp4 + audio) takes them from there.")
    return n_done


if __name__ == "__main__":
    main()
