#!/usr/bin/env python3
"""Host-side scaling probe for the 8-GPU run (the only scaling risk of the sharded path is host contention).

Runs the HOST half of scripts/inference_test_bench.py -- the CelebA reader behind a DataLoader (4 worker processes, as the reference),
the landmark-prefetch thread slot and the PNG writer thread (reface_amd/output.OutputWriter: 6 PNG files per image) -- with the GPU
replaced by a sleep of the measured per-batch device time, in 1 process and then in N processes at once (one per GPU of a node), on a
synthetic CelebAMask-HQ tree.  Reports per-process ms per batch of 8: the host half must stay under the device time of a batch
(~900 ms for 8 images on MI355X) in all N processes at once for the >= 7.5x target to be reachable.  CPU only; no GPU is touched.

  python tools/host_scaling_probe.py [--procs 8] [--batches 6] [--device-ms 900]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def photo_like(rng, h, w):
    """A smooth image with photographic statistics for the codecs (low-frequency structure + a little sensor noise): what JPEG / PNG see on
    face photographs, against uniform noise, their worst case."""
    import numpy as np
    from PIL import Image
    base = Image.fromarray(rng.integers(0, 256, (h // 32, w // 32, 3), dtype=np.uint8)).resize((w, h), Image.BICUBIC)
    a = np.asarray(base, dtype=np.int16) + rng.integers(-3, 4, (h, w, 3), dtype=np.int16)
    return np.clip(a, 0, 255).astype(np.uint8)


def make_tree(root, n, natural=False):
    import numpy as np
    from PIL import Image
    os.makedirs(os.path.join(root, "CelebA-HQ-img"), exist_ok=True)
    os.makedirs(os.path.join(root, "CelebA-HQ-mask", "Overall_mask"), exist_ok=True)
    rng = np.random.default_rng(0)
    for i in list(range(28000, 28000 + n)) + list(range(29000, 29000 + n)):
        img = photo_like(rng, 1024, 1024) if natural else rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
        Image.fromarray(img).save(os.path.join(root, "CelebA-HQ-img", f"{i}.jpg"), quality=90)
        lab = np.repeat(np.repeat(rng.integers(0, 19, (16, 16), dtype=np.uint8), 32, 0), 32, 1) if natural else rng.integers(0, 19, (512, 512), dtype=np.uint8)
        Image.fromarray(lab).save(os.path.join(root, "CelebA-HQ-mask", "Overall_mask", f"{i}.png"))


def worker(tree, outdir, batches, device_ms, B=8, legacy=False, world=1, png_level=None, gpu_prep=False, natural=False, aux_level=None, stage="both", loader_workers=4):
    import numpy as np
    import torch
    from reface_amd import output as O
    from reface_amd.data import CelebAdataset
    torch.set_num_threads(1)
    # --gpu-prep: the readers only decode / resize and hand over uint8 arrays (raw="full": normalisation, masks, the 224x224 source resize and
    # the products run on the GPU, reface_amd/prep.py) -- the host half the CLI has with --gpu_prep
    from reface_amd.data import raw_collate
    ds = CelebAdataset(dataset_dir=tree, n_targets=batches * B, raw="full" if gpu_prep else False)
    loader = torch.utils.data.DataLoader(ds, batch_size=B, num_workers=loader_workers, shuffle=False, collate_fn=raw_collate if gpu_prep else None)
    if stage == "writer":          # (--stage writer: the PNG half alone -- the same records submitted `batches` times, no reader)
        class _NoReader:
            def __iter__(self_):
                ids = [f"{i:012d}" for i in range(B)]
                tgt = torch.zeros((B, 3, 8, 8))
                for _ in range(batches):
                    yield (tgt, None, None, None, ids) if gpu_prep else (tgt, None, None, ids)
        loader = _NoReader()
    [os.makedirs(os.path.join(outdir, d), exist_ok=True) for d in ("results", "grid", "samples")]
    # round 4: the panels / grid are composed on the GPU and arrive as ONE packed uint8 record per image; the writer's worker count is the
    # process's share of the host.  --legacy: the round-3 host half (fp32 panels composed on the host, 8 workers per process)
    writer = O.OutputWriter(outdir) if legacy else O.OutputWriter(outdir, threads=O.default_writer_threads(world), compress_level=png_level, aux_compress_level=aux_level)
    nbytes, _ = O.record_layout(512, 512)
    rng = np.random.default_rng(0)
    pool_u8 = rng.integers(0, 256, (B, nbytes), dtype=np.uint8)              # stand-in for the D2H'd records (noise: zlib's worst case)
    if natural:                                                              # ... or photo-like panels and grid
        _, lay = O.record_layout(512, 512)
        for i in range(B):
            for k, (off, shp) in lay.items():
                pool_u8[i, off:off + int(np.prod(shp))] = photo_like(rng, (shp[0] + 31) // 32 * 32, (shp[1] + 31) // 32 * 32)[:shp[0], :shp[1]].reshape(-1)
    pool_f = rng.random((2, B, 3, 512, 512), dtype=np.float32)
    t0 = time.perf_counter()
    host_ms = []
    n = 0
    for item in loader:
        if gpu_prep:
            target, ids = item[0], item[4]
            kw = None
        else:
            target, prior, kw, ids = item
        t_host = time.perf_counter()
        time.sleep(device_ms / 1e3)                      # the device's share of the batch (sampling + decode are queued, the host is free)
        if stage == "reader":          # (--stage reader: the decode / resize half alone)
            pass
        elif legacy and not gpu_prep:
            writer.submit(list(ids), pool_f[0][:target.shape[0]], target.float().numpy(), kw["inpaint_image"].float().numpy(),
                          kw["inpaint_mask"].float().numpy(), pool_f[1][:target.shape[0]])
        else:
            writer.submit_u8(list(ids), pool_u8[:target.shape[0]], 512, 512)
        host_ms.append(1e3 * (time.perf_counter() - t_host) - device_ms)
        n += 1
        if n == 1:
            t0 = time.perf_counter()                     # steady state: the first batch carries the DataLoader workers' start-up
        if n >= batches:
            break
    writer.close()
    total = 1e3 * (time.perf_counter() - t0)
    print(json.dumps({"batches": n, "ms_per_batch": total / max(n - 1, 1), "host_ms_on_launch_thread": sum(host_ms[1:]) / max(len(host_ms) - 1, 1)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--device-ms", type=float, default=900.0)
    ap.add_argument("--worker", nargs=2, default=None)
    ap.add_argument("--legacy", action="store_true", help="the round-3 host half: fp32 panels composed on the host, 8 PNG workers per process")
    ap.add_argument("--world", type=int, default=1, help="(worker) processes sharing the host")
    ap.add_argument("--natural", action="store_true", help="photo-like images in and out (smooth content + sensor noise) instead of uniform noise, the codecs' worst case")
    ap.add_argument("--gpu-prep", action="store_true", help="readers hand over uint8 arrays only (the CLI's --gpu_prep)")
    ap.add_argument("--png-level", type=int, default=None, help="zlib level of the PNG files (default: PIL's 6, the reference's files)")
    ap.add_argument("--stage", default="both", choices=["both", "reader", "writer"], help="attribute the host half: the DataLoader readers alone, the PNG writer alone (run with --device-ms 0)")
    ap.add_argument("--loader-workers", type=int, default=4, help="DataLoader worker processes per rank (reference: 4)")
    ap.add_argument("--aux-png-level", type=int, default=None, help="zlib level of the samples/ and grid/ files only (the CLI's --fast_aux_png = 1); results/ keeps --png-level")
    a = ap.parse_args()
    if a.worker:
        worker(a.worker[0], a.worker[1], a.batches, a.device_ms, legacy=a.legacy, world=a.world, png_level=a.png_level, gpu_prep=a.gpu_prep, natural=a.natural,
               aux_level=a.aux_png_level, stage=a.stage, loader_workers=a.loader_workers)
        return
    with tempfile.TemporaryDirectory() as tmp:
        tree = os.path.join(tmp, "CelebAMask-HQ")
        make_tree(tree, a.batches * 8, natural=a.natural)
        out = {}
        for n in (1, a.procs):
            extra = (["--legacy"] if a.legacy else []) + (["--png-level", str(a.png_level)] if a.png_level is not None else []) + (["--gpu-prep"] if a.gpu_prep else []) + (["--natural"] if a.natural else []) + \
                    (["--aux-png-level", str(a.aux_png_level)] if a.aux_png_level is not None else []) + ["--stage", a.stage, "--loader-workers", str(a.loader_workers)]
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", tree, os.path.join(tmp, f"out_{n}_{r}"), "--batches", str(a.batches),
                                    "--device-ms", str(a.device_ms), "--world", str(n)] + extra, stdout=subprocess.PIPE, text=True) for r in range(n)]
            rs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
            out[n] = {"ms_per_batch_max": max(r["ms_per_batch"] for r in rs), "ms_per_batch_mean": sum(r["ms_per_batch"] for r in rs) / n,
                      "host_ms_on_launch_thread_max": max(r["host_ms_on_launch_thread"] for r in rs)}
        out["cpus"] = len(os.sched_getaffinity(0))
        from reface_amd.output import available_cpus
        out["cpus_under_cgroup_quota"] = available_cpus()          # what the container may really use (the mask alone says 256 on the test pool's boxes, the quota 16)
        out["host_half"] = "round 3 (fp32 panels composed on the host, 8 PNG workers)" if a.legacy else \
            f"round 4 (packed uint8 records from the device, PNG workers = share of the host, zlib level {a.png_level if a.png_level is not None else 6}" + \
            (f" for results/, {a.aux_png_level} for samples/ + grid/" if a.aux_png_level is not None else "") + \
            (", readers decode / resize only: --gpu_prep)" if a.gpu_prep else ")")
        out["content"] = "photo-like (smooth + sensor noise)" if a.natural else "uniform noise (worst case of JPEG decode and zlib)"
        out["device_ms_assumed"] = a.device_ms
        out["stage"] = a.stage
        out["loader_workers"] = a.loader_workers
        out["verdict"] = ("host half hides under the device time in all %d processes" % a.procs
                          if out[a.procs]["ms_per_batch_max"] < 1.05 * max(a.device_ms, out[1]["ms_per_batch_max"]) else
                          "host half EXCEEDS the device time with %d processes: scaling would be host-bound on this machine" % a.procs)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
