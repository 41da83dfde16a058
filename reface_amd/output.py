"""On-disk outputs of the test-bench CLI, byte-compatible with scripts/inference_test_bench.py:500-553 of the reference.

Per image id the reference writes
  results/<id>.png           255 * clamp((x_dec + 1) / 2, 0, 1), truncated to uint8                      (:493-495, :536-538)
  grid/grid-<id>.png         torchvision.utils.make_grid([GT, inpaint, ref, result]) (nrow 8, padding 2, pad value 0)   (:518-531)
  samples/<id>_mask.png      255 * (mask + 1) / 2 of the FULL-resolution {0, 1} keep-mask, grey -> RGB (0 -> 127)        (:540-543)
  samples/<id>_GT.png        255 * (target + 1) / 2                                                      (:544-546)
  samples/<id>_inpaint.png   255 * (masked target + 1) / 2                                               (:547-549)
  samples/<id>_ref.png       CLIP-un-normalised reference, bilinearly resized 224 -> 512 (no clamp)      (:523-525, :550-552)
All float -> uint8 conversions are numpy ``astype(np.uint8)`` of float32 arrays, exactly as the reference does them (truncation;
out-of-range values of the un-clamped panels wrap the same way on the same platform).

Host-side code (PIL PNG encoding).  ``OutputWriter`` runs the encodes on a worker thread so the GPU launch thread never waits
for zlib (the reference encodes inline, inference_test_bench.py:531-552).
"""
import os
import queue
import threading

import numpy as np

CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32).reshape(3, 1, 1)
CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32).reshape(3, 1, 1)


def un_norm(x):
    """inference_test_bench.py:500-501 (float32 arithmetic, as torch does it)."""
    return (np.asarray(x, dtype=np.float32) + np.float32(1.0)) / np.float32(2.0)


def un_norm_clip(x):
    """inference_test_bench.py:502-514: x * std + mean per channel, [3, H, W] float32."""
    return np.asarray(x, dtype=np.float32) * CLIP_STD + CLIP_MEAN


def make_grid(panels, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a list of [3, H, W] float32 panels (normalize=False): panels left to right, `nrow` per
    row, `padding` pixels of `pad_value` between panels and around the border -> [3, ymaps*(H+p)+p, xmaps*(W+p)+p]."""
    n = len(panels)
    _, H, W = panels[0].shape
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    hh, ww = H + padding, W + padding
    grid = np.full((3, hh * ymaps + padding, ww * xmaps + padding), pad_value, dtype=np.float32)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * hh + padding:y * hh + padding + H, x * ww + padding:x * ww + padding + W] = panels[k]
            k += 1
    return grid


def to_u8_hwc(chw):
    """255. * rearrange(x, 'c h w -> h w c') -> astype(uint8): the reference's conversion (truncation)."""
    return (np.float32(255.0) * np.transpose(np.asarray(chw, dtype=np.float32), (1, 2, 0))).astype(np.uint8)


def compose(result01, target, inpaint_image, inpaint_mask, ref512, skip_grid=False):
    """uint8 HWC arrays of one image's output files.

    result01 [3,H,W] in [0,1]; target / inpaint_image [3,H,W] in [-1,1]; inpaint_mask [1,H,W] in {0,1} (full resolution);
    ref512 [3,H,W]: the CLIP-normalised reference already resized to the image size."""
    gt, inp, ref = un_norm(target), un_norm(inpaint_image), un_norm_clip(ref512)
    out = {
        "result": to_u8_hwc(result01),
        "mask": np.repeat(to_u8_hwc(un_norm(inpaint_mask)), 3, axis=2),      # cv2.COLOR_GRAY2RGB replicates the channel
        "GT": to_u8_hwc(gt),
        "inpaint": to_u8_hwc(inp),
        "ref": to_u8_hwc(ref),
    }
    if not skip_grid:
        # the reference builds (and saves) the grid for every image even under --skip_grid (:518-531); the flag is honoured here
        out["grid"] = to_u8_hwc(make_grid([gt, inp, ref, np.asarray(result01, dtype=np.float32)]))
    return out


PANELS = ("result", "mask", "GT", "inpaint", "ref")          # order of the 5 [H][W][3] panels in a packed record (rf_compose_outputs_u8)


def record_layout(H, W, with_grid=True):
    """(record bytes, {name: (offset, shape)}) of one image's packed uint8 record as rf_compose_outputs_u8 writes it."""
    panel = H * W * 3
    lay = {n: (k * panel, (H, W, 3)) for k, n in enumerate(PANELS)}
    nbytes = 5 * panel
    if with_grid:
        lay["grid"] = (nbytes, (H + 4, 4 * W + 10, 3))
        nbytes += (H + 4) * (4 * W + 10) * 3
    return nbytes, lay


def available_cpus():
    """CPUs this process may actually use: the affinity mask capped by the container's CFS quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`).  The GPU
    boxes of the test pool show 256 CPUs in the mask under a quota of 16 (`profiles/r06i_host_half_stages.txt`): sizing pools by the mask alone
    oversubscribes the quota and every thread is throttled."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 8
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        else:
            quota = open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read().strip()
            period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
        if quota != "max" and int(quota) > 0:
            cpus = min(cpus, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cpus


def default_writer_threads(world=1):
    """PNG-encode workers per process.  One job is ONE file (48 per batch of 8), so idle workers cost nothing and what bounds the count is the
    process's share of the host next to its 4 loader processes: available_cpus() / (world * 2), between 2 and 8 (16 CPUs per rank, what a GPU slot
    of the test pool gets: 8 workers; with 5 workers and one job per IMAGE a batch of 8 took two rounds of the slowest image)."""
    return max(2, min(8, available_cpus() // (max(1, world) * 2)))


def paths(outdir, sid):
    s, r, g = (os.path.join(outdir, d) for d in ("samples", "results", "grid"))
    return {"result": os.path.join(r, sid + ".png"), "grid": os.path.join(g, "grid-" + sid + ".png"),
            "mask": os.path.join(s, sid + "_mask.png"), "GT": os.path.join(s, sid + "_GT.png"),
            "inpaint": os.path.join(s, sid + "_inpaint.png"), "ref": os.path.join(s, sid + "_ref.png")}


# files of one image, most expensive encode first (results/ at the reference's zlib level, then the 4-panel grid, then the panels): the jobs of a
# batch are queued in this order ACROSS its images, so the long encodes start first and the short ones fill the workers' tails
ENCODE_ORDER = ("result", "grid", "GT", "inpaint", "ref", "mask")


class OutputWriter:
    """PNG encoding off the launch thread: ``submit`` takes host arrays of one batch and returns at once; ``close`` drains."""

    def __init__(self, outdir, skip_grid=False, depth=4, threads=8, compress_level=None, aux_compress_level=None):
        # One job per FILE, `threads` workers: the six PNG encodes of an image are ~0.3 s of zlib on incompressible content (zlib runs
        # outside the GIL), so a single worker caps the CLI at ~3 images/s -- below one MI355X (tools/host_scaling_probe.py: 2.86 s per
        # batch of 8 with one worker against 0.9 s of device time)
        self.outdir, self.skip_grid = outdir, skip_grid
        # zlib level of the PNG files: None = PIL's default (6), what the reference's Image.save writes; a lower level gives the same
        # pixels in larger files for a third of the CPU time (RF_PNG_LEVEL / --png_level)
        # aux_compress_level (--fast_aux_png): its own level for the four samples/ panels and the grid/ file; results/<id>.png -- the file a user
        # compares with the reference's -- keeps `compress_level`
        self.save_kw = {} if compress_level is None else {"compress_level": int(compress_level)}
        self.aux_kw = self.save_kw if aux_compress_level is None else {"compress_level": int(aux_compress_level)}
        self.q = queue.Queue(maxsize=depth * 8 * len(ENCODE_ORDER))
        self.err = None
        self.n = 0                 # images whose files are all on disk
        self.left = {}             # image id -> files still to write
        self.lock = threading.Lock()
        self.ts = [threading.Thread(target=self._run, daemon=True) for _ in range(max(1, threads))]
        for t in self.ts:
            t.start()

    def _run(self):
        from PIL import Image
        while True:
            job = self.q.get()
            if job is None:
                return
            try:
                if len(job) == 6:          # float panels of one image (the host-composed path): compose here, off the launch thread, then
                    sid = job[0]           # hand the files to whoever is free -- or encode them here when the queue is full (never block a worker)
                    arrs = compose(*job[1:], skip_grid=self.skip_grid)
                    todo = [(sid, k, arrs[k], paths(self.outdir, sid)[k]) for k in ENCODE_ORDER if k in arrs]
                    with self.lock:
                        self.left[sid] = self.left.get(sid, 0) + len(todo) - 1
                    for j in todo[1:]:
                        try:
                            self.q.put_nowait(j)
                        except queue.Full:
                            self._encode(Image, *j)
                    job = todo[0]
                self._encode(Image, *job)
            except Exception as e:          # surfaced on the next submit / close
                self.err = e
            finally:
                self.q.task_done()

    def _encode(self, Image, sid, key, arr, path):
        Image.fromarray(arr).save(path, **(self.save_kw if key == "result" else self.aux_kw))
        with self.lock:
            self.left[sid] -= 1
            if self.left[sid] == 0:
                del self.left[sid]
                self.n += 1

    def _enqueue(self, per_image):
        """per_image: [(sid, {name: uint8 HWC array})] of one batch -> one job per file, long encodes first."""
        with self.lock:
            for sid, arrs in per_image:
                self.left[sid] = self.left.get(sid, 0) + len(arrs)
        for key in ENCODE_ORDER:
            for sid, arrs in per_image:
                if key in arrs:
                    self.q.put((sid, key, arrs[key], paths(self.outdir, sid)[key]))

    def submit(self, ids, result01, target, inpaint_image, inpaint_mask, ref512):
        if self.err is not None:
            raise self.err
        with self.lock:
            for sid in ids:
                self.left[sid] = self.left.get(sid, 0) + 1          # the compose job itself; the worker adds the image's files
        for i, sid in enumerate(ids):
            self.q.put((sid, result01[i], target[i], inpaint_image[i], inpaint_mask[i], ref512[i]))

    def submit_u8(self, ids, records, H, W):
        """records: uint8 [B, record_bytes] host array (one D2H copy of what rf_compose_outputs_u8 wrote); the rows are copied out of the
        staging buffer here, so the caller may reuse it at once."""
        if self.err is not None:
            raise self.err
        _, lay = record_layout(H, W, with_grid=not self.skip_grid)
        per_image = []
        for i, sid in enumerate(ids):
            rec = np.array(records[i], copy=True)
            per_image.append((sid, {k: rec[o:o + int(np.prod(shp))].reshape(shp) for k, (o, shp) in lay.items()}))
        self._enqueue(per_image)

    def close(self):
        self.q.join()              # (a compose job queues its files before it is done: nothing is behind the sentinels)
        for _ in self.ts:
            self.q.put(None)
        for t in self.ts:
            t.join()
        if self.err is not None:
            raise self.err
        return self.n
