"""UNetModel -- the 9-channel SD-inpainting UNet of REFace on hand-written HIP kernels.

Interface mirrors ldm/modules/diffusionmodules/openaimodel.py:528-907 (constructor kwargs from
configs/train.yaml:33-47, ``forward(x, timesteps, context)`` -> eps, NCHW fp32); parameters carry
the reference's ``state_dict`` names.  The arithmetic is a prepared list of kernel launches
(``UNetEngine``) over channels-last activations:

* ResBlock (openaimodel.py:255-275): GN+SiLU -> conv3x3 (+bias +timestep row-vector) -> GN+SiLU ->
  conv3x3 (+bias +skip) ; skip = identity or 1x1 conv.
* SpatialTransformer (attention.py:278-289, 239-243): GN -> 1x1 -> LN -> fused QKV GEMM -> fused
  attention -> out-proj (+bias +residual +cross-attention row-vector) -> LN -> GEGLU GEMM -> GEMM
  (+residual) -> 1x1 (+residual).
* Cross-attention (attention.py:179-221 with a 1-token context, ddpm.py:1012,1022): the softmax
  over a single key is exactly 1, so attn2(x) == to_out(to_v(c)) for every token -- a per-sample
  vector computed once per conditioning and added in the attn1 out-projection epilogue.  norm2 and
  attn2.to_q / to_k never influence the output.
* torch.cat([h, hs.pop()]) (openaimodel.py:898) is zero-copy: producers write straight into the
  two halves of a preallocated concat buffer (strided outputs).
* Upsample (openaimodel.py:116-118): nearest x2 is folded into the conv's input addressing.
* The timestep path (util.py:151-166, openaimodel.py:632-636, 218-224) runs in fp32 for every
  compute dtype: one GEMM produces the 22 ResBlock embedding vectors at once.
"""
import math

import os

import torch
import torch.nn as nn

from . import ops
from .gnfuse import ProducerTracker
from .modules import ParamTree, flat_state, weights_version
from .params import UNetConfig, unet_param_specs, unet_plan

F32 = torch.float32
H16 = (torch.bfloat16, torch.float16)          # the 16-bit storage modes: the same kernels (element type bf16_t / f16_t) and the same fused paths


class _Pool:
    """Free-list of device tensors keyed by (numel, dtype); launch order == allocation order."""

    def __init__(self, device):
        self.device, self.free, self.bytes = device, {}, 0
        self.on_put = None          # called with a tensor when its memory goes back to the free list

    def get(self, shape, dtype):
        n = 1
        for s in shape:
            n *= s
        lst = self.free.get((n, dtype))
        if lst:
            return lst.pop().view(shape)
        self.bytes += n * (4 if dtype == F32 else 2)
        return torch.empty(shape, dtype=dtype, device=self.device)

    def put(self, t):
        if self.on_put is not None:
            self.on_put(t)
        self.free.setdefault((t.numel(), t.dtype), []).append(t.reshape(-1))


class UNetEngine:
    """Prepared launch list for one (batch, height, width, dtype).

    ``uniform_t=True``: all samples share one timestep (DDIM sampling): the embedding buffer has one
    row that every sample reads (row-vector pitch 0).
    """

    CPAD = 16      # 9 input channels stored in 16 (multiple of the 16-byte vector for both dtypes)

    def __init__(self, sd, cfg: UNetConfig, B, H, W, dtype, device, uniform_t=False, emb_rows=None, cfg_pair=False):
        # dtype "fp8" (BASELINE configs[4]): bf16 activations, GEMM weights stored as fp8 e4m3fn + per-output-channel power-of-two
        # scales wherever rf_conv_gemm takes them (K and channel count multiples of 64), fp32 accumulate
        # "fp8": additionally fp8 ACTIVATIONS (e4m3fn + one E8M0 scale per 32 channels, written by the GroupNorm / LayerNorm passes) into the
        # ResBlock convolutions, proj_in, qkv and GEGLU projections, multiplied on the MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4);
        # "fp8w": fp8 weights only (bf16 activations on the bf16 MFMA)
        # "f32x3": fp32 storage, statistics, attention and epilogues; GEMM operands as split-bf16 pairs (hi + lo = 16 significant bits) in three
        # bf16 MFMA passes with fp32 accumulation (rf_conv_gemm RF_BF16X3) -- the fast form of the exact-fp32 parity mode
        # "fp8c" (round 4): the fp8 x fp8 path for the 3x3 CONVOLUTIONS only (ResBlock / resampling convs: 85 % of the FLOPs) -- every projection
        # (proj_in / qkv / to_out / GEGLU / ff.net.2 / proj_out / 1x1 skips) stays bf16.  tools/fp8_weight_scale_ablation.py: the image error of the
        # fp8 mode is the 3-mantissa-bit rounding of the PROJECTION weights (30.9 dB with everything quantised, 39.9 dB with the 3x3 convs alone;
        # no scale granularity changes that: per-32-block E8M0 scales give 30.9 dB again)
        self.conv_only8 = dtype == "fp8c"
        if self.conv_only8:
            dtype = "fp8"
        self.x3 = dtype == "f32x3"
        if self.x3:
            dtype = torch.float32
        self.n_x3 = 0
        self.w8 = dtype in ("fp8", "fp8w")
        self.a8 = dtype == "fp8"
        self.n_a8 = 0
        self._apool = {}
        if self.w8:
            dtype = torch.bfloat16
        self.cfg, self.B, self.H, self.W, self.dt, self.dev = cfg, B, H, W, dtype, device
        self.n_fp8 = 0
        self.uniform_t = uniform_t
        # cfg_pair: the caller guarantees that samples [0, B/2) and [B/2, B) carry the SAME x and timestep and differ only in
        # the context (classifier-free guidance, ddim.py:330-341).  Everything upstream of the first cross-attention is then
        # computed once for B/2 samples -- bit-identical to computing it twice.
        self.cfg_pair = bool(cfg_pair) and uniform_t and B % 2 == 0
        self.korder_on = os.environ.get("REFACE_KORDER", "0") == "1"
        # GroupNorm statistics come out of the epilogue of the GEMM that produced the tensor wherever its tile plan allows
        self.gn_fuse = os.environ.get("REFACE_GN_FUSE", "1") == "1"
        # GEGLU + ff.net.2 of the C = 320 blocks as ONE kernel (csrc/ffn.hip): 213 us against 150 + 80 for the pair inside the step -- the pair's
        # epilogues are store-bound and ff.net.2's residual segments cost ~20 us; -0.4 % per batch, same box (tools/archive/exp_r03_17.sh).  =0: the pair
        self.ffn_fuse = os.environ.get("REFACE_FFN_FUSE", "1") == "1"
        self.ffn_whole = float(os.environ.get("REFACE_FFN_WHOLE", "0.9"))          # least fill of the fused kernel's last round of 128-token blocks
        # norm3 inside the fused feed-forward kernel (C = 320 blocks): REFACE_LN_FOLD=0 keeps the separate LayerNorm pass
        self.ln_fold = os.environ.get("REFACE_LN_FOLD", "1") == "1"
        # norm1 / norm3 folded around their neighbour GEMMs (producer statistics + consumer epilogue affine; bf16 mode): REFACE_LN_FOLD_GEMM=0 off
        self.ln_fold_gemm = os.environ.get("REFACE_LN_FOLD_GEMM", "1") == "1"
        # row-extended A tiles for the 3x3 stride-1 convolutions (korder 2): REFACE_HX=0 keeps the tap-major K order
        self.hx_on = os.environ.get("REFACE_HX", "1") == "1"
        # SpatialTransformer.norm (GroupNorm, no SiLU) folded into proj_in's weights per sample (rf_groupnorm_fold_linear): REFACE_GN_FOLD=0 keeps the pass
        self.gn_fold_lin = os.environ.get("REFACE_GN_FOLD", "1") == "1"
        # proj_out fused behind the feed-forward kernel of the C = 320 blocks (rf_ffn_block): REFACE_TAIL_FUSE=0 keeps the separate proj_out launches
        self.tail_fuse = os.environ.get("REFACE_TAIL_FUSE", "1") == "1"
        self.n_tail_fused = 0
        # proj_out folded into ff.net.2 where the token-resident kernel does not reach (C = 640 / 1280, and C = 320 at sizes that kernel declines; 16-bit
        # modes): one GEMM over K = 5 C on premultiplied weights instead of two launches and a round trip of [M, C] (_st): REFACE_PO_FOLD=0 keeps the pair
        self.po_fold = os.environ.get("REFACE_PO_FOLD", "1") == "1"
        self.n_po_folded = 0
        # the token-resident FRONT of the C = 320 blocks: norm (folded) + proj_in + norm1 + qkv in one kernel (rf_attn_in, csrc/attnin.hip): REFACE_FRONT_FUSE=0 keeps the
        # two K = 320 GEMM launches
        self.front_fuse = os.environ.get("REFACE_FRONT_FUSE", "1") == "1"
        self.n_front_fused = 0
        # attn1.to_out (+ residual + cross-attention vector) in FRONT of the token-resident tail kernel (rf_ffn_desc.wo): REFACE_MID_FUSE=0 keeps its rf_conv_gemm launch(es)
        self.mid_fuse = os.environ.get("REFACE_MID_FUSE", "1") == "1"
        self.n_mid_fused = 0
        self.out_fuse = os.environ.get("REFACE_OUT_FUSE", "1") == "1"          # `out` head (GroupNorm + SiLU + 3x3 conv to 4 channels) as one pass (csrc/smallconv.hip)
        # the stem (3x3 conv from the 9 stored-in-16 input channels) as a pixels-on-lanes kernel that computes the CFG-duplicated half once and emits
        # the statistics of both its GroupNorm consumers (csrc/smallconv.hip): REFACE_STEM_FUSE=0 keeps the implicit GEMM + the statistics pass
        self.stem_fuse = os.environ.get("REFACE_STEM_FUSE", "1") == "1"
        self.n_stem_fused = 0
        # 3x3 convolutions whose 256-row tiles overhang whole rounds of the chip by a little are split by samples (_add_conv3): REFACE_SAMPLE_SPLIT=0 off
        self.sample_split = os.environ.get("REFACE_SAMPLE_SPLIT", "1") == "1"
        self.n_sample_split = 0
        self.gn_fold_maxc = int(os.environ.get("REFACE_GN_FOLD_MAXC", "640"))
        self.n_gn_folded = 0
        self.n_hx = 0
        self.n_ln_folded = 0
        self.n_cu = torch.cuda.get_device_properties(device).multi_processor_count if torch.cuda.is_available() else 256
        self.gn_fused = 0
        self.pool = _Pool(device)
        self._wpack = {}
        self.tracker = ProducerTracker()
        self.pool.on_put = self.tracker.forget
        self.sd = {k: v.detach().to(device=device, dtype=F32) for k, v in sd.items()}
        self.plan = unet_plan(cfg)
        nds = len(cfg.channel_mult) - 1
        if H % (1 << nds) or W % (1 << nds):
            raise ValueError(f"latent size {H}x{W} must be a multiple of {1 << nds}")
        self.x_in = torch.zeros((B, H, W, self.CPAD), dtype=dtype, device=device)
        self.eps = torch.empty((B, H, W, 4), dtype=F32, device=device)
        self.gn_partial = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=device)
        self.ws = ops.new_workspace(device)          # this engine's own split-K scratch (ops.workspace_scope)
        with ops.workspace_scope(self.ws):
            self._build_emb(emb_rows if emb_rows is not None else (1 if uniform_t else B))
            self._build_ctx()
            self.main = []
            self._build_main()
        self.sd = None      # packed copies are held by the launches
        self._wpack = None

    # ------------------------------------------------------------------ weights
    def w(self, key, dtype=None):
        return self.sd[key].to(dtype or self.dt).contiguous()          # (non-GEMM uses; GEMM weights go through gw())

    def f32(self, key):
        return self.sd[key].contiguous()

    def gw(self, w2d, cin=None):
        """A GEMM weight [N, K] in the engine's storage: the activation dtype, or (fp8 mode) e4m3fn bytes + per-row scales."""
        if self.w8 and ops.fp8_eligible(w2d.shape[1], cin):
            self.n_fp8 += 1
            return ops.quantize_fp8(w2d)
        return w2d.to(self.dt).contiguous()

    def aget(self, shape):
        """An Fp8Act buffer (free-list by shape; pad bytes / pad scales are initialised once and never written)."""
        lst = self._apool.get(tuple(shape))
        return lst.pop() if lst else ops.Fp8Act(tuple(shape), self.dev)

    def aput(self, a):
        self._apool.setdefault(tuple(a.shape), []).append(a)

    def x3_ok(self, K, cin=None):
        return self.x3 and ops.x3_eligible(K, cin)

    def split_in(self, x):
        """f32x3 mode: the split-bf16 form of an fp32 GEMM input that no normalisation pass rewrites (a pooled temporary; the caller
        returns it with self.pool.put once the consuming launch is appended)."""
        s_ = self.pool.get(tuple(x.shape[:-1]) + (2 * x.shape[-1],), torch.bfloat16)
        self.main.append(ops.split_bf16(x, s_, name="split_bf16"))
        return s_

    def lin(self, x, w2d, out, bias, name, **kw):
        """out = x W^T (+ epilogue) in the engine's GEMM form: bf16 / fp32 / fp8-weight operands, or (f32x3) split-bf16 pairs -- x may
        already be split ([M, 2K] bf16 from a normalisation pass), else a split pass is inserted."""
        K = w2d.shape[1]
        if self.x3_ok(K):
            xs = x if x.dtype == torch.bfloat16 else self.split_in(x)
            self.main.append(ops.linear(xs, ops.pack_x3(w2d), out, bias, x3=True, name=name, **kw))
            self.n_x3 += 1
            if xs is not x:
                self.pool.put(xs)
        else:
            self.main.append(ops.linear(x, self.gw(w2d), out, bias, name=name, **kw))

    def gw8(self, w2d, taps, cin):
        """fp8 weight of an fp8-activation GEMM: every tap's channel run zero-padded to the 128-byte K tile."""
        self.n_fp8 += 1
        self.n_a8 += 1
        return ops.quantize_fp8_padded(w2d.to(self.dev), taps, cin)

    # ------------------------------------------------------------------ timestep path (fp32)
    def _res_prefixes(self):
        ib, mid, ob = self.plan
        out = []
        for i, layers in enumerate(ib):
            out += [(f"input_blocks.{i}.{j}", l[2]) for j, l in enumerate(layers) if l[0] == "res"]
        out += [(f"middle_block.{j}", l[2]) for j, l in enumerate(mid) if l[0] == "res"]
        for i, layers in enumerate(ob):
            out += [(f"output_blocks.{i}.{j}", l[2]) for j, l in enumerate(layers) if l[0] == "res"]
        return out

    def _build_emb(self, rows):
        cfg, dev = self.cfg, self.dev
        mc, td = cfg.model_channels, cfg.time_embed_dim
        half = mc // 2
        # util.py:160-162 -- frequency table built exactly as the reference does (fp32 torch ops)
        self.freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=F32) / half).to(dev)
        res = self._res_prefixes()
        self.emb_off, off = {}, 0
        for p, cout in res:
            self.emb_off[p] = (off, cout)
            off += cout
        self.E = off
        wcat = torch.cat([self.f32(f"{p}.emb_layers.1.weight") for p, _ in res], 0).contiguous()
        bcat = torch.cat([self.f32(f"{p}.emb_layers.1.bias") for p, _ in res], 0).contiguous()
        self._emb_w = (self.f32("time_embed.0.weight"), self.f32("time_embed.0.bias"),
                       self.f32("time_embed.2.weight"), self.f32("time_embed.2.bias"), wcat, bcat)
        self.emb_rows = rows
        self.t_f32 = torch.zeros(rows, dtype=F32, device=dev)
        self.emb_table = torch.zeros((rows, self.E), dtype=F32, device=dev)
        self.emb_launches = self.make_emb_launches(self.t_f32, self.emb_table)

    def make_emb_launches(self, t_f32, table):
        """Launch chain  t[n] -> table[n, E]  (all ResBlock ``emb_layers`` outputs)."""
        n = t_f32.shape[0]
        mc, td, dev = self.cfg.model_channels, self.cfg.time_embed_dim, self.dev
        w0, b0, w2, b2, wcat, bcat = self._emb_w
        temb = torch.empty((n, mc), dtype=F32, device=dev)
        h = torch.empty((n, td), dtype=F32, device=dev)
        emb = torch.empty((n, td), dtype=F32, device=dev)
        semb = torch.empty((n, td), dtype=F32, device=dev)
        with ops.workspace_scope(self.ws):
            return [ops.timestep_embedding(t_f32, self.freqs, temb),
                    ops.linear(temb, w0, h, b0, act=ops.ACT_SILU, name="time_embed.0"),
                    ops.linear(h, w2, emb, b2, name="time_embed.2"),
                    ops.silu_f32(emb, semb),
                    ops.linear(semb, wcat, table, bcat, name="emb_layers")]

    def emb_vec(self, p):
        off, cout = self.emb_off[p]
        return self.emb_table[:, off:off + cout]

    # ------------------------------------------------------------------ conditioning path (fp32)
    def _st_prefixes(self):
        ib, mid, ob = self.plan
        out = []
        for i, layers in enumerate(ib):
            out += [(f"input_blocks.{i}.{j}", l[1]) for j, l in enumerate(layers) if l[0] == "st"]
        out += [(f"middle_block.{j}", l[1]) for j, l in enumerate(mid) if l[0] == "st"]
        for i, layers in enumerate(ob):
            out += [(f"output_blocks.{i}.{j}", l[1]) for j, l in enumerate(layers) if l[0] == "st"]
        return out

    def _build_ctx(self):
        B, dev = self.B, self.dev
        sts = self._st_prefixes()
        self.ctx_off, off = {}, 0
        for p, c in sts:
            self.ctx_off[p] = (off, c)
            off += c
        self.ctx_in = torch.zeros((B, self.cfg.context_dim), dtype=F32, device=dev)
        self.ctx_table = torch.zeros((B, max(off, 1)), dtype=F32, device=dev)
        self.ctx_launches = []
        for p, c in sts:
            t = f"{p}.transformer_blocks.0.attn2"
            v = torch.empty((B, c), dtype=F32, device=dev)
            o, _ = self.ctx_off[p]
            self.ctx_launches.append(ops.linear(self.ctx_in, self.f32(f"{t}.to_v.weight"), v, None, name=t + ".to_v"))
            self.ctx_launches.append(ops.linear(v, self.f32(f"{t}.to_out.0.weight"), self.ctx_table[:, o:o + c],
                                                self.f32(f"{t}.to_out.0.bias"), name=t + ".to_out"))

    def ctx_vec(self, p):
        off, c = self.ctx_off[p]
        return self.ctx_table[:, off:off + c]

    # ------------------------------------------------------------------ main graph
    def _conv3(self, x, wkey, out, bkey, name, **kw):
        """3x3 conv launch; K order chosen per layer (ops.conv_korder)."""
        cin = x.shape[3]
        if isinstance(x, ops.Fp8Act):
            return ops.conv2d(x, self.gw8(ops.pack_conv_weight(self.sd[wkey], F32), 9, cin), out, self.f32(bkey), name=name, **kw)
        if self.x3 and x.dtype == torch.bfloat16:          # split-bf16 input (from _gn(split=True) / split_in): cin is half the stored width
            self.n_x3 += 1
            return ops.conv2d(x, ops.pack_x3(ops.pack_conv_weight(self.sd[wkey], F32)), out, self.f32(bkey), x3=True, name=name, **kw)
        ko = ops.conv_korder(cin, self.dt) if self.korder_on else 0
        if (self.hx_on and self.dt in H16 and not self.w8 and not ko and cin % 64 == 0 and kw.get("stride", 1) == 1 and not kw.get("ups", 0)
                and x.shape[1:3] == out.shape[1:3]):
            # 3x3 stride-1 convolution: K order (filter row, channel chunk, filter column) -- one row-extended A tile serves the three horizontal
            # taps (a third of the A-operand fill, gemm.hip HX).  Whether the launch's tile can take it (whole image rows per tile) is the
            # library's answer: ask for the plan, fall back to the tap-major order otherwise.
            cand = ops.conv2d(x, self._packed(wkey, "hx", lambda: ops.pack_conv_weight(self.sd[wkey], self.dt, korder=2)), out, self.f32(bkey), korder=2, name=name, **kw)
            try:
                ops.gemm_plan2(cand)
                self.n_hx += 1
                return cand
            except Exception:
                pass
        def pack():
            wp = ops.pack_conv_weight(self.sd[wkey], F32, korder=ko)
            return wp.to(self.dt) if ko else self.gw(wp, cin)
        return ops.conv2d(x, self._packed(wkey, ("tap", ko), pack), out, self.f32(bkey), korder=ko, name=name, **kw)

    def _packed(self, wkey, kind, make):
        """One packed device copy per (weight, packing): a convolution launched in two sample slices (_add_conv3) shares it."""
        w = self._wpack.get((wkey, kind))
        if w is None:
            w = self._wpack[(wkey, kind)] = make()
        return w

    def _add_conv3(self, x, wkey, out, bkey, name, rowvec=None, residual=None, **kw):
        """_conv3 + _add, with the launch split BY SAMPLES when its 256-row tiles overhang a whole number of rounds of the chip by a little
        (768x768: 8 samples x 36 tiles = 288 = 1.125 rounds of 256 CUs -- the library then falls back to quarter tiles for the whole launch):
        the first k samples fill whole rounds of big tiles, the rest is a launch of its own (its plan: smaller tiles / split-K).
        `tools/conv_split_probe.py`: 96x96, 320 / 640 / 960 -> 320: 154 -> 135, 283 -> 236, 405 -> 338 us.  REFACE_SAMPLE_SPLIT=0: off."""
        B, k = x.shape[0], 0
        if (self.sample_split and isinstance(x, torch.Tensor) and not isinstance(out, ops.Fp8Act) and self.dt in H16 and not self.x3 and not self.w8
                and kw.get("stride", 1) == 1):
            HWo, N = out.shape[1] * out.shape[2], out.shape[3]
            tn = (N + 319) // 320
            if HWo % 256 == 0 and B > 1:
                ps = (HWo // 256) * tn                  # 256 x 320 tiles per sample
                t = B * ps
                full, rem = divmod(t, self.n_cu)
                if full >= 1 and 0 < rem <= self.n_cu // 4:
                    k = (self.n_cu * full) // ps
                    if not (0 < k < B and k * ps >= 0.9 * self.n_cu * full):
                        k = 0
        if not k:
            return self._add(self._conv3(x, wkey, out, bkey, name, **dict(kw, **({"rowvec": rowvec} if rowvec is not None else {}),
                                                                           **({"residual": residual} if residual is not None else {}))), out)
        self.n_sample_split += 1
        for a, b in ((0, k), (k, B)):
            kws = dict(kw)
            if rowvec is not None:
                kws["rowvec"] = rowvec[a:b]
            if residual is not None:
                kws["residual"] = residual[a:b]
            self._add(self._conv3(x[a:b], wkey, out[a:b], bkey, f"{name}[{a}:{b}]", **kws), out[a:b])

    def _add(self, launch, out=None):
        """Append a launch; GEMM outputs are remembered so that a later GroupNorm can ask their producers for its statistics."""
        self.main.append(launch)
        return self.tracker.record(out, launch)

    def _gn(self, x, key, eps, silu, fp8=False, split=False):
        if split:
            out = self.pool.get(tuple(x.shape[:3]) + (2 * x.shape[3],), torch.bfloat16)
        else:
            out = self.aget(x.shape) if fp8 else self.pool.get(tuple(x.shape), self.dt)
        fused = None
        if self.gn_fuse:
            prods = self.tracker.producers(x)
            if prods is not None:
                fused = ops.fuse_groupnorm_stats(x, prods)
        if fused is not None:
            self.main += fused[2]
            self.main.append(ops.groupnorm_apply(x, self.f32(key + ".weight"), self.f32(key + ".bias"), out, fused[0], fused[1],
                                                 eps=eps, silu=silu, split=split, name=key))
            self.gn_fused += 1
        else:
            self.main += ops.groupnorm(x, self.f32(key + ".weight"), self.f32(key + ".bias"), out, self.gn_partial, eps=eps,
                                       silu=silu, split=split, name=key)
        return out

    def _gn_stats(self, x):
        """GroupNorm(32) statistics of x alone (no normalised copy): from the producing GEMMs' epilogues when they can, else the statistics pass.
        -> (partial, chunks per sample); the launches are appended."""
        if self.gn_fuse:
            prods = self.tracker.producers(x)
            if prods is not None:
                fused = ops.fuse_groupnorm_stats(x, prods)
                if fused is not None:
                    self.main += fused[2]
                    self.gn_fused += 1
                    return fused[0], fused[1]
        l, n = ops.groupnorm_stats(x, self.gn_partial)
        self.main.append(l)
        return self.gn_partial, n

    def _res(self, p, x, cin, cout, dst):
        B, H, W, _ = x.shape
        t1 = self._gn(x, f"{p}.in_layers.0", 1e-5, True, fp8=self.a8, split=self.x3_ok(9 * cin, cin))
        h1 = self.pool.get((B, H, W, cout), self.dt)
        rv = self.emb_vec(p)
        if self.uniform_t:
            rv = rv.as_strided((B, cout), (0, 1), rv.storage_offset())      # every sample reads row 0 (B = this input's batch)
        self._add_conv3(t1, f"{p}.in_layers.2.weight", h1, f"{p}.in_layers.2.bias", f"{p}.in_layers.2", rowvec=rv)
        (self.aput if self.a8 else self.pool.put)(t1)
        t2 = self._gn(h1, f"{p}.out_layers.0", 1e-5, True, fp8=self.a8, split=self.x3_ok(9 * cout, cout))
        self.pool.put(h1)
        if cin != cout:
            skip = self.pool.get((B, H, W, cout), self.dt)
            w_sk = self.sd[f"{p}.skip_connection.weight"].reshape(cout, cin)
            if self.x3_ok(cin, cin):
                xs = self.split_in(x)
                self.main.append(ops.conv2d(xs, ops.pack_x3(w_sk), skip, self.f32(f"{p}.skip_connection.bias"), ksize=1, pad=(0, 0), x3=True,
                                            name=f"{p}.skip_connection"))
                self.pool.put(xs)
                self.n_x3 += 1
            else:
                w_skq = w_sk.to(self.dt).contiguous() if self.conv_only8 else self.gw(w_sk)          # ("fp8c": projections stay bf16)
                self.main.append(ops.conv2d(x, w_skq, skip, self.f32(f"{p}.skip_connection.bias"), ksize=1, pad=(0, 0),
                                            name=f"{p}.skip_connection"))
        else:
            skip = x
        y = dst if dst is not None else self.pool.get((B, H, W, cout), self.dt)
        self._add_conv3(t2, f"{p}.out_layers.3.weight", y, f"{p}.out_layers.3.bias", f"{p}.out_layers.3", residual=skip)
        (self.aput if self.a8 else self.pool.put)(t2)
        if cin != cout:
            self.pool.put(skip)
        return y

    def _st(self, p, x, c, heads, dst, pair=False):
        """SpatialTransformer (attention.py:218-289).  pair=True: ``x`` holds B/2 samples shared by both CFG halves; the block
        runs on B/2 samples up to the self-attention and fans out to the full batch where the context enters."""
        if self.conv_only8 and self.w8:          # "fp8c": the transformer block runs as in the bf16 mode
            self.w8 = self.a8 = False
            try:
                return self._st(p, x, c, heads, dst, pair=pair)
            finally:
                self.w8 = self.a8 = True
        B, H, W, _ = x.shape
        M, d = B * H * W, c // heads
        t = f"{p}.transformer_blocks.0"
        nb = 2 if pair else 1                   # batch fan-out at the cross-attention
        a8 = self.a8
        tok = self.pool.get((M, c), self.dt)
        w_pi = self.sd[f"{p}.proj_in.weight"].reshape(c, c)
        l_pi = None
        # C = 320: GroupNorm (folded into per-sample weights) + proj_in + norm1 + to_q / to_k / to_v as ONE token-resident kernel (rf_attn_in) where its 128-token blocks come in
        # nearly whole rounds of the chip: tok is written once (to_out's residual) and never re-read, the LayerNorm statistics never leave the registers that hold the row
        fblk = (M + 127) // 128
        front = (self.front_fuse and c == 320 and self.dt in H16 and not a8 and not self.w8 and self.gn_fold_lin and x.is_contiguous() and (H * W) % 128 == 0 and
                 fblk / (self.n_cu * ((fblk + self.n_cu - 1) // self.n_cu)) >= self.ffn_whole)
        if front:
            part, nch = self._gn_stats(x)
            fl, wps, rv = ops.groupnorm_fold_linear(w_pi.float().contiguous(), self.f32(f"{p}.norm.weight"), self.f32(f"{p}.norm.bias"),
                                                    self.f32(f"{p}.proj_in.bias"), part, nch, B=B, HW=H * W, eps=1e-6, dtype=self.dt, name=f"{p}.norm.fold")
            wq = torch.cat([self.sd[f"{t}.attn1.to_q.weight"].float() * (d ** -0.5 * ops.LOG2E), self.sd[f"{t}.attn1.to_k.weight"].float(),
                            self.sd[f"{t}.attn1.to_v.weight"].float()], 0)
            wqf, bqf = ops.fold_layernorm_geglu(wq, torch.zeros(3 * c, device=self.dev), self.sd[f"{t}.norm1.weight"], self.sd[f"{t}.norm1.bias"])
            qkv = self.pool.get((M, 3 * c), self.dt)
            self.main.append(fl)
            self.main.append(ops.attn_in(x.view(M, c), wps, rv, tok, wqf.to(self.dt).contiguous(), bqf.contiguous(), qkv, rows_per_sample=H * W, ln_eps=1e-5,
                                         name=f"{p}.proj_in+norm1+qkv"))
            self.n_gn_folded += 1
            self.n_ln_folded += 1
            self.n_front_fused += 1
            ln = self.pool.get((M, c), self.dt)
            return self._st_after_qkv(p, x, c, heads, dst, pair, tok, qkv, ln, a8)
        if self.gn_fold_lin and self.dt in H16 and not a8 and not self.w8 and c <= self.gn_fold_maxc and x.is_contiguous() and (H * W) % 256 == 0:
            # `norm` folded into proj_in (bf16 mode, C <= 640: B x C x C folded weights cost less than the pass they replace): the statistics of x
            # scale W's columns per sample, proj_in multiplies the UN-normalised x, the mean / beta terms ride in its per-sample vector
            part, nch = self._gn_stats(x)
            fl, wps, rv = ops.groupnorm_fold_linear(w_pi.float().contiguous(), self.f32(f"{p}.norm.weight"), self.f32(f"{p}.norm.bias"),
                                                    self.f32(f"{p}.proj_in.bias"), part, nch, B=B, HW=H * W, eps=1e-6, dtype=self.dt, name=f"{p}.norm.fold")
            cand = ops.linear(x.view(M, c), wps[0], tok, None, rowvec=rv, rows_per_sample=H * W, w_per_sample=wps, name=f"{p}.proj_in")
            ops.gemm_plan2(cand)          # (raises if a tile would straddle samples: H W is a multiple of every tile height)
            self.main.append(fl)
            l_pi = cand
            self.n_gn_folded += 1
        if l_pi is None:
            g = self._gn(x, f"{p}.norm", 1e-6, False, fp8=a8)
            l_pi = ops.linear(g.view(M, c), self.gw8(w_pi, 1, c) if a8 else self.gw(w_pi), tok, self.f32(f"{p}.proj_in.bias"), name=f"{p}.proj_in")
            (self.aput if a8 else self.pool.put)(g)
        self.main.append(l_pi)
        ln = self.aget((M, c)) if a8 else self.pool.get((M, c), self.dt)
        qkv = self.pool.get((M, 3 * c), self.dt)
        # to_q carries d^-0.5 * log2(e): the scores reach the attention kernels in the exp2 domain (scale = ln 2 below) -- the product
        # is rounded to the storage type once, as a weight, instead of the kernel re-rounding q * scale
        wqkv = torch.cat([self.sd[f"{t}.attn1.to_q.weight"].float() * (d ** -0.5 * ops.LOG2E), self.sd[f"{t}.attn1.to_k.weight"].float(),
                          self.sd[f"{t}.attn1.to_v.weight"].float()], 0)
        # norm1 folded around the two GEMMs (bf16 mode): proj_in's epilogue leaves per-row (mean, M2) records, the qkv GEMM reads the
        # UN-normalised tokens and applies  rstd (acc - mean u) + W beta  in its epilogue; gamma rides in the weights' columns -- the
        # rf_layernorm pass (one read + one write of [M, C]) and its launch are gone
        l_qkv = None
        if self.ln_fold_gemm and self.dt in H16 and not self.w8:
            w2, u2, b2 = ops.fold_layernorm_linear(wqkv, self.sd[f"{t}.norm1.weight"], self.sd[f"{t}.norm1.bias"], None, self.dt)
            cand = ops.linear(tok, w2, qkv, b2, ln_u=u2, name=f"{t}.attn1.qkv")
            if ops.layernorm_fold([(l_pi, 0, M)], cand, eps=1e-5, C_=c) is not None:
                l_qkv = cand
                self.n_ln_folded += 1
        if l_qkv is None:
            self.main.append(ops.layernorm(tok, self.f32(f"{t}.norm1.weight"), self.f32(f"{t}.norm1.bias"), ln, name=f"{t}.norm1"))
            l_qkv = ops.linear(ln, self.gw8(wqkv, 1, c) if a8 else self.gw(wqkv), qkv, None, name=f"{t}.attn1.qkv")
        self.main.append(l_qkv)
        if a8:
            self.aput(ln)
            ln = self.pool.get((M, c), self.dt)
        return self._st_after_qkv(p, x, c, heads, dst, pair, tok, qkv, ln, a8)

    def _st_after_qkv(self, p, x, c, heads, dst, pair, tok, qkv, ln, a8):
        """The SpatialTransformer block from the self-attention on (attention.py:239-243, 268-289): `tok` = proj_in's output (the residual), `qkv` the fused projection,
        `ln` a free [M, C] buffer."""
        B, H, W, _ = x.shape
        M, d = B * H * W, c // heads
        t = f"{p}.transformer_blocks.0"
        nb = 2 if pair else 1
        att = ln    # reuse the LayerNorm buffer for the attention output
        q3 = qkv.view(B, H * W, 3 * c)
        self.main.append(ops.attention(q3[..., :c], q3[..., c:2 * c], q3[..., 2 * c:], att.view(B, H * W, c), heads=heads,
                                       scale=ops.LN2, name=f"{t}.attn1"))
        # (one block of the fused feed-forward kernel per 128 tokens and CU: only where those blocks come in nearly whole rounds -- 576 blocks at configs[3]
        #  are 2.25 rounds of 256 CUs and lose 0.5 % per batch against the pair, tools/archive/exp_r03_20.sh)
        nblk = (nb * M + 127) // 128
        whole = nblk / (self.n_cu * ((nblk + self.n_cu - 1) // self.n_cu))
        fused_ffn = self.ffn_fuse and c == 320 and self.dt in H16 and not self.w8 and whole >= self.ffn_whole
        # proj_out folded into ff.net.2 (attention.py:243, 268-272, 288-289; exact in real arithmetic):
        #     y = Wpo (W2 h + b2 + x1) + bpo + x_in  =  [h | x1] [Wpo W2 | Wpo]^T + (Wpo b2 + bpo) + x_in
        # one GEMM over K = 5 C whose A operand is ONE buffer [M, 5 C]: the GEGLU projection writes h into columns [0, 4 C), the attention's out-projection
        # writes x1 into columns [4 C, 5 C) (strided outputs, as the decoder's [h | skip] concat) -- a single source keeps the direct-to-LDS main loop.  The
        # product Wpo W2 is formed in fp32 and rounded ONCE to the operand type; x2 = ff(x1) + x1 is never rounded to 16 bits, never written, never re-read:
        # a launch and a round trip of [M, C] per block fewer (11 blocks at 512x512).
        po_fold = self.po_fold and nb == 1 and self.dt in H16 and not self.w8 and not a8 and not fused_ffn and c % 64 == 0
        cat5 = self.pool.get((M, 5 * c), self.dt) if po_fold else None
        x1 = cat5[:, 4 * c:] if po_fold else self.pool.get((nb * M, c), self.dt)
        # attn1 out-projection + residual + the (token-independent) cross-attention output; with pair=True one launch per CFG
        # half: same A and residual, that half's context vectors
        w_out, b_out, cv = self.gw(self.sd[f"{t}.attn1.to_out.0.weight"]), self.f32(f"{t}.attn1.to_out.0.bias"), self.ctx_vec(p)
        # the whole rest of the block in ONE kernel where the token-resident tail applies: the out-projection runs in front of the feed-forward inside rf_ffn_block (x1 is
        # written once, as the feed-forward's residual, and not re-read as its input; the launch(es) of attn1.to_out are gone)
        tail_ok = fused_ffn and self.ln_fold and self.tail_fuse and x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and (B == 1 or x.stride(0) == H * x.stride(1))
        mid = self.mid_fuse and tail_ok and (H * W) % 128 == 0
        l_out = []
        if not mid:
            for hf in range(nb):
                l_out.append(ops.linear(att, w_out, x1[hf * M:(hf + 1) * M], b_out, residual=tok, rowvec=cv[hf * B:(hf + 1) * B],
                                        rows_per_sample=H * W, name=f"{t}.attn1.to_out"))
            self.main += l_out
        self.pool.put(qkv)
        if mid:
            w1f, b1f = ops.fold_layernorm_geglu(self.sd[f"{t}.ff.net.0.proj.weight"], self.sd[f"{t}.ff.net.0.proj.bias"], self.sd[f"{t}.norm3.weight"], self.sd[f"{t}.norm3.bias"])
            wgm, bgm = ops.pack_geglu(w1f, b1f, F32)
            y = dst if dst is not None else self.pool.get((nb * B, H, W, c), self.dt)
            y2 = y.as_strided((nb * M, c), (y.stride(2), 1))
            assert y.stride(3) == 1 and y.stride(1) == W * y.stride(2) and y.stride(0) == H * y.stride(1)
            x_rows = x.as_strided((M, c), (x.stride(2), 1))
            self._add(ops.ffn_block(att, wgm.to(self.dt).contiguous(), bgm, ops.pack_ffn_w2(self.sd[f"{t}.ff.net.2.weight"], self.dt), self.f32(f"{t}.ff.net.2.bias"), y2,
                                    residual=x1, wpo=self.sd[f"{p}.proj_out.weight"].reshape(c, c).to(self.dt).contiguous(), bpo=self.f32(f"{p}.proj_out.bias"),
                                    res2=x_rows, res2_rows=M if nb > 1 else 0, ln_eps=1e-5,
                                    front=dict(wo=w_out, bo=b_out, ctx=cv, rows_per_sample=H * W, res0=tok, front_rows=M if nb > 1 else 0, x1=x1),
                                    name=f"{t}.to_out+ff+proj_out"), y)
            for buf in (tok, ln, x1):          # (only now: the launch reads tok and the attention output, and owns x1)
                self.pool.put(buf)
            self.n_ln_folded += 1
            self.n_tail_fused += 1
            self.n_mid_fused += 1
            return y
        self.pool.put(tok)
        if pair or a8:
            self.pool.put(ln)
            ln = self.aget((nb * M, c)) if a8 else self.pool.get((nb * M, c), self.dt)
        fold = fused_ffn and self.ln_fold
        w1s, b1s = self.sd[f"{t}.ff.net.0.proj.weight"], self.sd[f"{t}.ff.net.0.proj.bias"]
        if fold:
            # ... and norm3 inside that kernel too: a lane pair holds the token's whole row, so the LayerNorm is ~400 VALU operations per lane in
            # registers; gamma rides in W1's columns, beta in b1 -- the rf_layernorm pass over [M, C] and its launch are gone
            self.pool.put(ln)
            w1s, b1s = ops.fold_layernorm_geglu(w1s, b1s, self.sd[f"{t}.norm3.weight"], self.sd[f"{t}.norm3.bias"])
            self.n_ln_folded += 1
        l_geglu_f = None
        if not fold and self.ln_fold_gemm and self.dt in H16 and not self.w8:
            # norm3 of the unfused feed-forward (C = 640 / 1280, and C = 320 where the fused kernel is not taken): the statistics come from the
            # to_out launch(es) that wrote x1, the GEGLU GEMM reads x1 itself and normalises in its epilogue
            wgp, bgp = ops.pack_geglu(w1s, b1s, F32)          # (the fold is per input column: it commutes with the row packing)
            w2, u2, b2 = ops.fold_layernorm_linear(wgp, self.sd[f"{t}.norm3.weight"], self.sd[f"{t}.norm3.bias"], None, self.dt)
            b2 = (b2 + bgp).contiguous()
            ggc = cat5[:, :4 * c] if po_fold else self.pool.get((nb * M, 4 * c), self.dt)
            cand = ops.linear(x1, w2, ggc, b2, act=ops.ACT_GEGLU, ln_u=u2, name=f"{t}.ff.net.0")
            if ops.layernorm_fold([(l, hf * M, M) for hf, l in enumerate(l_out)], cand, eps=1e-5, C_=c) is not None:
                l_geglu_f = (cand, ggc)
                self.n_ln_folded += 1
            elif not po_fold:
                self.pool.put(ggc)
        if not fold and l_geglu_f is None:
            self.main.append(ops.layernorm(x1, self.f32(f"{t}.norm3.weight"), self.f32(f"{t}.norm3.bias"), ln, name=f"{t}.norm3"))
        wg, bg = ops.pack_geglu(w1s, b1s, F32)
        if po_fold:
            if l_geglu_f is not None:
                self.main.append(l_geglu_f[0])
            else:          # (norm3 ran as its own pass into `ln`)
                self.main.append(ops.linear(ln, self.gw(wg), cat5[:, :4 * c], bg, act=ops.ACT_GEGLU, name=f"{t}.ff.net.0"))
            wpo32, w232 = self.sd[f"{p}.proj_out.weight"].reshape(c, c).float(), self.sd[f"{t}.ff.net.2.weight"].float()
            wf = torch.cat([wpo32 @ w232, wpo32], dim=1).contiguous()
            bf = (wpo32 @ self.sd[f"{t}.ff.net.2.bias"].float() + self.sd[f"{p}.proj_out.bias"].float()).contiguous()
            y = dst if dst is not None else self.pool.get((B, H, W, c), self.dt)
            self._add(ops.conv2d(cat5.view(B, H, W, 5 * c), self.gw(wf), y, bf, ksize=1, pad=(0, 0), residual=x, name=f"{t}.ff.net.2+proj_out"), y)
            self.pool.put(ln)
            self.pool.put(cat5)
            self.n_po_folded += 1
            return y
        if l_geglu_f is not None:
            l_g, gg = l_geglu_f
            self.main.append(l_g)
            x2 = ln
            self.main.append(ops.linear(gg, self.gw(self.sd[f"{t}.ff.net.2.weight"]), x2, self.f32(f"{t}.ff.net.2.bias"), residual=x1, name=f"{t}.ff.net.2"))
            self.pool.put(gg)
        elif fused_ffn and self.tail_fuse and x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and (B == 1 or x.stride(0) == H * x.stride(1)):
            # one kernel for the whole token-resident tail of the block: feed-forward (hidden tensor in registers) + residual + proj_out + `x + x_in`
            # (csrc/ffn.hip, rf_ffn_block) -- the feed-forward's output never leaves the CU, the proj_out launches and their re-read of it are gone;
            # the statistics of the GroupNorm that reads the block's output come from this kernel's epilogue (the launch is a statistics producer)
            y = dst if dst is not None else self.pool.get((nb * B, H, W, c), self.dt)
            y2 = y.as_strided((nb * M, c), (y.stride(2), 1))
            assert y.stride(3) == 1 and y.stride(1) == W * y.stride(2) and y.stride(0) == H * y.stride(1)
            x_rows = x.as_strided((M, c), (x.stride(2), 1))
            self._add(ops.ffn_block(x1 if fold else ln, wg.to(self.dt).contiguous(), bg, ops.pack_ffn_w2(self.sd[f"{t}.ff.net.2.weight"], self.dt),
                                    self.f32(f"{t}.ff.net.2.bias"), y2, residual=x1, wpo=self.sd[f"{p}.proj_out.weight"].reshape(c, c).to(self.dt).contiguous(),
                                    bpo=self.f32(f"{p}.proj_out.bias"), res2=x_rows, res2_rows=M if nb > 1 else 0, ln_eps=1e-5 if fold else 0.0,
                                    name=f"{t}.ff+proj_out"), y)
            if not fold:
                self.pool.put(ln)
            self.pool.put(x1)
            self.n_tail_fused += 1
            return y
        elif fused_ffn:
            # one kernel: the [M, 4C] hidden tensor (168 MB at 64x64) stays in registers (csrc/ffn.hip)
            x2 = self.pool.get((nb * M, c), self.dt)
            self.main.append(ops.ffn_geglu(x1 if fold else ln, wg.to(self.dt).contiguous(), bg, ops.pack_ffn_w2(self.sd[f"{t}.ff.net.2.weight"], self.dt),
                                           self.f32(f"{t}.ff.net.2.bias"), x2, residual=x1, ln_eps=1e-5 if fold else 0.0, name=f"{t}.ff"))
            if not fold:
                self.pool.put(ln)
        else:
            wg = self.gw8(wg, 1, c) if a8 else self.gw(wg)
            # fp8 mode: the gated hidden tensor leaves the GEGLU epilogue as fp8 + block scales (half the bytes of the widest tensor of the
            # block) and ff.net.2 runs on the fp8 MFMA too
            gg = self.aget((nb * M, 4 * c)) if a8 else self.pool.get((nb * M, 4 * c), self.dt)
            l_geglu = ops.linear(ln, wg, gg, bg, act=ops.ACT_GEGLU, name=f"{t}.ff.net.0")
            if a8:
                try:                      # the fp8-output epilogue needs the direct epilogue form: ask the library for this launch's plan
                    ops.gemm_plan(l_geglu)
                    self.main.append(l_geglu)
                except Exception:         # otherwise: bf16 hidden tensor + its own quantisation pass
                    gb = self.pool.get((nb * M, 4 * c), self.dt)
                    self.main.append(ops.linear(ln, wg, gb, bg, act=ops.ACT_GEGLU, name=f"{t}.ff.net.0"))
                    self.main.append(ops.quantize_act(gb, gg, name=f"{t}.ff.quantize"))
                    self.pool.put(gb)
            else:
                self.main.append(l_geglu)
            if a8:
                self.aput(ln)
                ln = self.pool.get((nb * M, c), self.dt)
            x2 = ln
            w2 = self.sd[f"{t}.ff.net.2.weight"]
            self.main.append(ops.linear(gg, self.gw8(w2, 1, 4 * c) if a8 else self.gw(w2), x2, self.f32(f"{t}.ff.net.2.bias"), residual=x1, name=f"{t}.ff.net.2"))
            (self.aput if a8 else self.pool.put)(gg)
        self.pool.put(x1)
        y = dst if dst is not None else self.pool.get((nb * B, H, W, c), self.dt)
        w_po, b_po = self.gw(self.sd[f"{p}.proj_out.weight"].reshape(c, c)), self.f32(f"{p}.proj_out.bias")
        for hf in range(nb):        # the residual x is shared by both halves
            self._add(ops.conv2d(x2.view(nb * B, H, W, c)[hf * B:(hf + 1) * B], w_po, y[hf * B:(hf + 1) * B], b_po,
                                 ksize=1, pad=(0, 0), residual=x, name=f"{p}.proj_out"), y[hf * B:(hf + 1) * B])
        self.pool.put(x2)
        return y

    def _st_x3(self, p, x, c, heads, dst, pair=False):
        """SpatialTransformer of the f32x3 mode: fp32 tensors, every GEMM on split-bf16 operand pairs -- written by the GroupNorm / LayerNorm
        passes where one precedes the GEMM, by a split pass otherwise; fp32 attention.  Same graph as _st."""
        B, H, W, _ = x.shape
        M, d = B * H * W, c // heads
        t = f"{p}.transformer_blocks.0"
        nb = 2 if pair else 1
        BF = torch.bfloat16
        g = self._gn(x, f"{p}.norm", 1e-6, False, split=True)
        tok = self.pool.get((M, c), F32)
        self.lin(g.view(M, 2 * c), self.sd[f"{p}.proj_in.weight"].reshape(c, c), tok, self.f32(f"{p}.proj_in.bias"), f"{p}.proj_in")
        self.pool.put(g)
        ln = self.pool.get((M, 2 * c), BF)
        self.main.append(ops.layernorm(tok, self.f32(f"{t}.norm1.weight"), self.f32(f"{t}.norm1.bias"), ln, name=f"{t}.norm1"))
        qkv = self.pool.get((M, 3 * c), F32)
        wqkv = torch.cat([self.sd[f"{t}.attn1.to_q.weight"].float() * (d ** -0.5 * ops.LOG2E), self.sd[f"{t}.attn1.to_k.weight"].float(),
                          self.sd[f"{t}.attn1.to_v.weight"].float()], 0)
        self.lin(ln, wqkv, qkv, None, f"{t}.attn1.qkv")
        self.pool.put(ln)
        att = self.pool.get((M, c), F32)
        q3 = qkv.view(B, H * W, 3 * c)
        # split-bf16 operand pairs for both contractions of the attention too (REFACE_X3_ATTN=0: the exact-fp32 MFMA kernel)
        self.main.append(ops.attention(q3[..., :c], q3[..., c:2 * c], q3[..., 2 * c:], att.view(B, H * W, c), heads=heads,
                                       scale=ops.LN2, x3=os.environ.get("REFACE_X3_ATTN", "1") == "1", name=f"{t}.attn1"))
        x1 = self.pool.get((nb * M, c), F32)
        atts = self.split_in(att)
        w_out, b_out, cv = ops.pack_x3(self.sd[f"{t}.attn1.to_out.0.weight"]), self.f32(f"{t}.attn1.to_out.0.bias"), self.ctx_vec(p)
        for hf in range(nb):
            self.main.append(ops.linear(atts, w_out, x1[hf * M:(hf + 1) * M], b_out, residual=tok, rowvec=cv[hf * B:(hf + 1) * B],
                                        rows_per_sample=H * W, x3=True, name=f"{t}.attn1.to_out"))
            self.n_x3 += 1
        for buf in (atts, att, qkv, tok):
            self.pool.put(buf)
        ln3 = self.pool.get((nb * M, 2 * c), BF)
        self.main.append(ops.layernorm(x1, self.f32(f"{t}.norm3.weight"), self.f32(f"{t}.norm3.bias"), ln3, name=f"{t}.norm3"))
        wg, bg = ops.pack_geglu(self.sd[f"{t}.ff.net.0.proj.weight"], self.sd[f"{t}.ff.net.0.proj.bias"], F32)
        gg = self.pool.get((nb * M, 4 * c), F32)
        self.lin(ln3, wg, gg, bg, f"{t}.ff.net.0", act=ops.ACT_GEGLU)
        self.pool.put(ln3)
        x2 = self.pool.get((nb * M, c), F32)
        self.lin(gg, self.sd[f"{t}.ff.net.2.weight"], x2, self.f32(f"{t}.ff.net.2.bias"), f"{t}.ff.net.2", residual=x1)
        self.pool.put(gg)
        self.pool.put(x1)
        y = dst if dst is not None else self.pool.get((nb * B, H, W, c), F32)
        x2s = self.split_in(x2.view(nb * B, H, W, c))
        w_po, b_po = ops.pack_x3(self.sd[f"{p}.proj_out.weight"].reshape(c, c)), self.f32(f"{p}.proj_out.bias")
        for hf in range(nb):
            self._add(ops.conv2d(x2s[hf * B:(hf + 1) * B], w_po, y[hf * B:(hf + 1) * B], b_po, ksize=1, pad=(0, 0), residual=x, x3=True,
                                 name=f"{p}.proj_out"), y[hf * B:(hf + 1) * B])
            self.n_x3 += 1
        self.pool.put(x2s)
        self.pool.put(x2)
        return y

    def _stem(self, p, x, cout, dst):
        """input_blocks.0.0 (openaimodel.py:666-671) as rf_conv3x3_stem when the shapes allow; with cfg_pair the second batch half of x is a copy of
        the first (the same latent under both conditionings): it is computed once and stored twice.  -> True when the launch was appended."""
        B, H, W, ci = x.shape
        if not (self.stem_fuse and self.dt in H16 and not self.x3 and dst is not None and ci == self.CPAD == 16 and
                cout in ops.SMALLCONV_CHANNELS and (H * W) % 128 == 0 and p == "input_blocks.0.0"):
            return False
        w = ops.pack_conv_weight(self.sd[f"{p}.weight"], self.dt, cin_pad=self.CPAD)
        if self.cfg_pair:
            h = B // 2
            l = ops.conv3x3_stem(x[:h], w, self.f32(f"{p}.bias"), dst[:h], dup=dst[h:], name=p)
            self._add(l, dst[:h])
            self.tracker.record(dst[h:], l)
        else:
            self._add(ops.conv3x3_stem(x, w, self.f32(f"{p}.bias"), dst, name=p), dst)
        self.n_stem_fused += 1
        return True

    def _block(self, prefix, layers, x, dst):
        """TimestepEmbedSequential (openaimodel.py:80-88); the last layer writes into ``dst``."""
        B = self.B
        pair = False
        if self.cfg_pair and prefix == "input_blocks.1" and [l[0] for l in layers] == ["res", "st"]:
            x, pair, B = x[:B // 2], True, B // 2          # both halves of x are identical: work on one
        for j, l in enumerate(layers):
            p = f"{prefix}.{j}"
            last = j == len(layers) - 1
            d = dst if last else None
            H, W = x.shape[1], x.shape[2]
            if l[0] == "conv" and self._stem(p, x, l[2], d):
                y = d
            elif l[0] == "conv":
                y = d if d is not None else self.pool.get((B, H, W, l[2]), self.dt)
                self._add(ops.conv2d(x, ops.pack_conv_weight(self.sd[f"{p}.weight"], self.dt, cin_pad=self.CPAD), y,
                                     self.f32(f"{p}.bias"), name=p), y)
            elif l[0] == "res":
                y = self._res(p, x, l[1], l[2], d)
            elif l[0] == "st":
                y = (self._st_x3 if self.x3 and l[1] % 64 == 0 else self._st)(p, x, l[1], l[2], d, pair=pair)
            elif l[0] in ("down", "up"):
                # fp8 mode: the residual-stream tensor is quantised by its own pass (1.5 bytes per element) -- the 3x3 convolution behind it has
                # 9 C of K per output and runs ~1.5x faster on the fp8 MFMA
                xc = x
                if self.a8:
                    xc = self.aget(x.shape)
                    self.main.append(ops.quantize_act(x, xc, name=f"{p}.quantize"))
                elif self.x3_ok(9 * x.shape[3], x.shape[3]):
                    xc = self.split_in(x)
                if l[0] == "down":
                    y = d if d is not None else self.pool.get((B, H // 2, W // 2, l[1]), self.dt)
                    self._add(self._conv3(xc, f"{p}.op.weight", y, f"{p}.op.bias", f"{p}.op", stride=2), y)
                else:
                    y = d if d is not None else self.pool.get((B, 2 * H, 2 * W, l[1]), self.dt)
                    self._add_conv3(xc, f"{p}.conv.weight", y, f"{p}.conv.bias", f"{p}.conv", ups=1)
                if self.a8:
                    self.aput(xc)
                elif xc is not x:
                    self.pool.put(xc)
            else:
                raise ValueError(l)
            if j > 0 and x is not None:
                self.pool.put(x)           # intra-block intermediate
            x = y
        return x

    @staticmethod
    def _out_ch(layers):
        for l in reversed(layers):
            if l[0] in ("res", "conv"):
                return l[2]
        return layers[-1][1]

    def _build_main(self):
        cfg, B, dev = self.cfg, self.B, self.dev
        ib, mid, ob = self.plan
        # shapes of every input-block output (the skips) and the h entering each output block
        skip_shapes, H, W = [], self.H, self.W
        for layers in ib:
            if layers[0][0] == "down":
                H, W = H // 2, W // 2
                skip_shapes.append((H, W, layers[0][1]))
            else:
                skip_shapes.append((H, W, self._out_ch(layers)))
        hch = self._out_ch(mid)
        cats, hH, hW = [], H, W
        stack = list(skip_shapes)
        for layers in ob:
            sh, sw, sc = stack.pop()
            assert (sh, sw) == (hH, hW), "skip / decoder resolution mismatch"
            cats.append(torch.empty((B, sh, sw, hch + sc), dtype=self.dt, device=dev))
            hch = self._out_ch(layers)
            if layers[-1][0] == "up":
                hH, hW = 2 * hH, 2 * hW
        n_in = len(ib)
        # input block i's output is the skip of output block (n_in - 1 - i): right part of that concat buffer
        x = self.x_in
        for i, layers in enumerate(ib):
            cat = cats[n_in - 1 - i]
            c1 = skip_shapes[i][2]
            dst = cat[..., cat.shape[3] - c1:]
            x = self._block(f"input_blocks.{i}", layers, x, dst)
        c0 = cats[0].shape[3] - skip_shapes[-1][2]
        self._block("middle_block", mid, x, cats[0][..., :c0])
        final = None
        for i, layers in enumerate(ob):
            if i + 1 < len(ob):
                nxt = cats[i + 1]
                c0n = nxt.shape[3] - skip_shapes[n_in - 2 - i][2]
                dst = nxt[..., :c0n]
            else:
                dst = None
            final = self._block(f"output_blocks.{i}", layers, cats[i], dst)
        c_fin = final.shape[3]
        if (self.out_fuse and self.dt in H16 and not self.x3 and final.dtype == self.dt and c_fin in ops.SMALLCONV_CHANNELS and
                self.sd["out.2.weight"].shape[0] <= 4 and final.is_contiguous()):
            # `out` = GroupNorm + SiLU + 3x3 conv to 4 channels: one pass over the raw tensor (csrc/smallconv.hip) instead of the normalisation pass
            # (a write + a read of the tensor) and an implicit GEMM that stages it nine times for a tile that is 94 % padding
            part, nch = self._gn_stats(final)
            self.out_ws = torch.empty((final.shape[0] * final.shape[1] * final.shape[2] * 40,), dtype=F32, device=dev)
            self.main.append(ops.gn_silu_conv3x3_small(final, self.f32("out.0.weight"), self.f32("out.0.bias"), part, nch,
                                                       ops.pack_conv_weight(self.sd["out.2.weight"], self.dt), self.f32("out.2.bias"), self.eps, eps=1e-5, silu=True,
                                                       workspace=self.out_ws, name="out"))
            self.cats = cats
            return
        g = self._gn(final, "out.0", 1e-5, True, split=self.x3_ok(9 * c_fin, c_fin))
        if g.dtype == torch.bfloat16 and self.x3:
            self.main.append(ops.conv2d(g, ops.pack_x3(ops.pack_conv_weight(self.sd["out.2.weight"], F32)), self.eps, self.f32("out.2.bias"), x3=True,
                                        name="out.2"))
            self.n_x3 += 1
        else:
            self.main.append(ops.conv2d(g, ops.pack_conv_weight(self.sd["out.2.weight"], self.dt), self.eps, self.f32("out.2.bias"), name="out.2"))
        self.cats = cats

    # ------------------------------------------------------------------ execution
    def set_context(self, context):
        """context: [B, 1, ctx_dim] or [B, ctx_dim] (device fp32)."""
        c = context.reshape(self.B, -1).to(device=self.dev, dtype=F32)
        self.ctx_in.copy_(c)
        ops.run(self.ctx_launches)

    def set_timesteps(self, t):
        self.t_f32.copy_(t.reshape(-1)[: self.emb_rows].to(device=self.dev, dtype=F32))
        ops.run(self.emb_launches)

    def run(self, stream=None):
        ops.run(self.main, stream)


class UNetModel(nn.Module):
    """Drop-in for ``ldm.modules.diffusionmodules.openaimodel.UNetModel`` (inference only)."""

    def __init__(self, image_size=32, in_channels=9, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), dropout=0, channel_mult=(1, 2, 4, 4), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 add_conv_in_front_of_unet=False, sep_head_att=False, land_mark_id_seperate_layers=False, head_splits=None,
                 compute_dtype=None):
        super().__init__()
        # the REFace path (configs/train.yaml:33-47); anything else is outside this build's scope
        unsupported = dict(dims=(dims, 2), num_classes=(num_classes, None), use_scale_shift_norm=(use_scale_shift_norm, False),
                           resblock_updown=(resblock_updown, False), n_embed=(n_embed, None), legacy=(legacy, False),
                           add_conv_in_front_of_unet=(add_conv_in_front_of_unet, False), sep_head_att=(sep_head_att, False),
                           land_mark_id_seperate_layers=(land_mark_id_seperate_layers, False), conv_resample=(conv_resample, True),
                           use_spatial_transformer=(use_spatial_transformer, True), num_head_channels=(num_head_channels, -1))
        for k, (got, want) in unsupported.items():
            if got != want:
                raise NotImplementedError(f"UNetModel({k}={got!r}) is not on the REFace inference path (expected {want!r})")
        if context_dim is None or num_heads == -1:
            raise NotImplementedError("UNetModel needs context_dim and num_heads (spatial-transformer configuration)")
        if isinstance(context_dim, (list, tuple)):
            context_dim = list(context_dim)[0]
        self.cfg = UNetConfig(in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                              num_res_blocks=num_res_blocks, attention_resolutions=tuple(attention_resolutions),
                              channel_mult=tuple(channel_mult), num_heads=num_heads, transformer_depth=transformer_depth,
                              context_dim=context_dim)
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, attention_resolutions, channel_mult
        self.num_heads, self.use_checkpoint, self.dtype = num_heads, use_checkpoint, torch.float32
        self.compute_dtype = compute_dtype or torch.float32
        tree = ParamTree(unet_param_specs(self.cfg))
        for name, child in tree.named_children():
            self.add_module(name, child)
        self._engines = {}

    def set_compute_dtype(self, dtype):
        """torch.float32 (exact-fp32 parity mode) | torch.bfloat16 (throughput mode) | torch.float16 (the throughput mode's kernels on fp16 storage and
        v_mfma_f32_32x32x16_f16: same rate, 11 significant bits instead of 8 -- the mode that brings the 50-step image closest to the exact-fp32 one at
        full speed; activations of a checkpoint must stay inside fp16's range, tools/act_range.py) | "fp8" (BASELINE configs[4]: fp8 e4m3fn GEMM weights
        everywhere + fp8 activations with E8M0 block scales into the ResBlock convs / proj_in / qkv / GEGLU on the fp8 MFMA) | "fp8w" (fp8
        weights, bf16 activations, bf16 MFMA)."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16, "fp8", "fp8w", "fp8c", "f32x3"):
            raise ValueError(f"unsupported UNet compute dtype {dtype!r}")
        self.compute_dtype = dtype
        self._engines.clear()

    def engine(self, B, H, W, uniform_t=False, emb_rows=None, cfg_pair=False):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("reface_amd.UNetModel runs on the GPU only (HIP kernels; there is no CPU fallback)")
        key = (B, H, W, self.compute_dtype, uniform_t, emb_rows, bool(cfg_pair), weights_version(self))
        eng = self._engines.get(key)
        if eng is None:
            self._engines = {k: v for k, v in self._engines.items() if k[-1] == key[-1]}   # drop engines of stale weights
            eng = UNetEngine(flat_state(self), self.cfg, B, H, W, self.compute_dtype, dev, uniform_t=uniform_t, emb_rows=emb_rows,
                             cfg_pair=cfg_pair)
            self._engines[key] = eng
        return eng

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, return_features=False, **kwargs):
        """x: [N, 9, h, w] fp32 NCHW; timesteps: [N]; context: [N, 1, 768] -> eps [N, 4, h, w] fp32."""
        assert y is None, "must specify y if and only if the model is class-conditional"
        if return_features:
            raise NotImplementedError("return_features is a training-time option")
        if context is None or context.shape[1] != 1:
            raise NotImplementedError("REFace conditioning is a single 768-d token per sample")
        N, C, H, W = x.shape
        eng = self.engine(N, H, W)
        ops.nchw_to_nhwc(x.float().contiguous(), eng.x_in)()
        eng.set_context(context)
        eng.set_timesteps(timesteps)
        eng.run()
        out = torch.empty((N, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        ops.nhwc_to_nchw(eng.eps, out)()
        return out
