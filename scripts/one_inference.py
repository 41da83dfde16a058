#!/usr/bin/env python3
"""One-pair swap caller on the MI355X-native engines -- the sampling core of the reference's scripts/one_inference.py.

The reference script is scripts/inference_swap_selected.py behind a Flask endpoint: `process_images` (one_inference.py:447-487) stores the
uploaded source / target under examples/FaceSwap/One_{source,target}, `run_inference(scale, steps)` (:521-790) runs stage 1 (alignment +
BiSeNet parsing into <Base_dir>/{target_cropped,mask_frames,source_cropped,source_mask}, :525-585) and stage 2 -- the same flags, the same
batch body, the same output tree -- and the endpoint returns <outdir>/results/0/000000000000.png as JPEG bytes.  Stage 1 and the web UI
are outside this build's scope; this module provides stage 2 with the reference's function names:

    run_inference(scale, steps)          stage 2 on the prepared <Base_dir> tree (one source, one target), through reface_amd/pipeline.py
    process_images(image1, image2, ...)  the endpoint's file choreography around it for ALREADY ALIGNED 512x512 crops + label maps

  python scripts/one_inference.py --Base_dir <prepared tree> --outdir <out> --config ... --ckpt ... [--scale 3.5 --ddim_steps 50]
"""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import inference_swap_selected as _sel  # noqa: E402

build_parser = _sel.build_parser        # one_inference.py:190-440 declares the flags of inference_swap_selected.py verbatim
_ARGV = None


def configure(argv):
    """Remember the command line `run_inference` completes with (scale, steps), as the module-level `opt` of the reference does."""
    global _ARGV
    _ARGV = list(argv)


def run_inference(scale=3.5, steps=50):
    """one_inference.py:521-790, stage 2: returns the path of the swapped crop of target 0 by source 0."""
    if _ARGV is None:
        raise RuntimeError("one_inference.configure(argv) first (the reference parses its flags at import time)")
    argv = [a for a in _ARGV]
    for flag, val in (("--scale", scale), ("--ddim_steps", steps)):
        if flag in argv:
            i = argv.index(flag)
            del argv[i:i + 2]
        argv += [flag, str(val)]
    opt = build_parser().parse_args(argv)
    _sel.main(argv)
    return os.path.join(opt.outdir, "results", "0", "000000000000.png")


def process_images(source_crop, source_labels, target_crop, target_labels, steps=50, scale=3.5):
    """The endpoint's flow (one_inference.py:447-487) for inputs that are already aligned crops + face-parsing label maps (PIL images
    or paths): writes the one-pair <Base_dir> tree, runs stage 2, returns the result as JPEG bytes (io.BytesIO), as the endpoint does."""
    from PIL import Image
    opt = build_parser().parse_args(_ARGV or [])
    base = opt.Base_dir
    for d, im in (("source_cropped", source_crop), ("source_mask", source_labels), ("target_cropped", target_crop), ("mask_frames", target_labels)):
        os.makedirs(os.path.join(base, d), exist_ok=True)
        (im if hasattr(im, "save") else Image.open(im)).save(os.path.join(base, d, "0.png"))
    out = Image.open(run_inference(scale=scale, steps=steps))
    buf = io.BytesIO()
    out.save(buf, "JPEG")
    buf.seek(0)
    return buf


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if "--serve" in argv:
        raise SystemExit("one_inference: the Flask UI of the reference (one_inference.py:443-519) is outside this build's scope; "
                         "import this module and call process_images / run_inference from your own endpoint")
    configure(argv)
    opt = build_parser().parse_args(argv)
    print(run_inference(scale=opt.scale, steps=opt.ddim_steps))


if __name__ == "__main__":
    main()
