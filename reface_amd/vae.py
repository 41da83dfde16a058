"""AutoencoderKL -- the KL-VAE first stage of REFace on the HIP kernels (fp32 by default).

Interface mirrors ldm/models/autoencoder.py:285-342 (``encode(x) -> posterior``, ``decode(z)``);
graph from ldm/modules/diffusionmodules/model.py:368-568:

* ResnetBlock (model.py:122-141): GN(eps 1e-6)+swish -> conv3x3 -> GN+swish -> conv3x3 (+shortcut).
* AttnBlock (model.py:178-202), 1 head of width C: scores = q k^T * C^-0.5 as a batched MFMA GEMM,
  row softmax, then P V as a GEMM against V^T (V^T is produced directly by a GEMM with swapped operand
  roles; since softmax rows sum to 1 the v-bias is added once in the P V epilogue).
* Upsample (model.py:53-57): nearest x2 folded into the conv addressing.
* Downsample (model.py:72-76): asymmetric (0,1,0,1) zero pad + stride-2 conv = pad_t = pad_l = 0.
* decode: z / scale_factor (ddpm.py:1284) is folded into post_quant_conv's alpha.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .gnfuse import ProducerTracker
from .modules import ParamTree, flat_state, weights_version
from .params import VAEConfig, vae_param_specs
from .unet import _Pool

F32 = torch.float32


class DiagonalGaussianDistribution(object):
    """ldm/modules/distributions/distributions.py:24-75 (inference subset)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters                      # [B, 2C, H, W] fp32 (mean | logvar)
        self.deterministic = deterministic

    @property
    def mean(self):
        return torch.chunk(self.parameters, 2, dim=1)[0]

    @property
    def logvar(self):
        return torch.clamp(torch.chunk(self.parameters, 2, dim=1)[1], -30.0, 20.0)

    def _draw(self, eps, scale):
        B, C2, H, W = self.parameters.shape
        out = torch.empty((B, C2 // 2, H, W), dtype=F32, device=self.parameters.device)
        ops.gaussian_sample(self.parameters.contiguous(), eps, out, scale=scale)()
        return out

    def sample(self, noise=None, scale=1.0):
        """mean + std * N(0,1).  The reference draws the noise on the CPU (distributions.py:36)."""
        if self.deterministic:
            return self._draw(None, scale)
        if noise is None:
            noise = torch.randn(self.mean.shape)
        return self._draw(noise.to(device=self.parameters.device, dtype=F32).contiguous(), scale)

    def mode(self):
        return self._draw(None, 1.0)


class _VAEEngine:
    def __init__(self, sd, cfg: VAEConfig, B, H, W, dtype, device, which):
        # dtype "bf16x3": fp32 storage of the residual stream, statistics, attention and every epilogue; the operands of the 3x3 / 1x1
        # convolutions as split-bf16 pairs (hi + lo = 16 significant bits) multiplied in three bf16 MFMA passes with fp32 accumulation
        # (rf_conv_gemm RF_BF16X3): ~2^-16 relative error per product instead of fp32's 2^-24, at a multiple of the fp32-MFMA rate
        self.x3 = dtype == "bf16x3"
        if self.x3:
            dtype = F32
        self.cfg, self.B, self.dt, self.dev = cfg, B, dtype, device
        self.n_x3 = 0
        self.pool = _Pool(device)
        self.tracker = ProducerTracker()
        self.pool.on_put = self.tracker.forget
        self.gn_fuse = os.environ.get("REFACE_GN_FUSE", "1") == "1"
        self.gn_fused = 0
        self.sd = {k: v.detach().to(device=device, dtype=F32) for k, v in sd.items()}
        self.gn_partial = torch.empty(B * ops.GN_MAX_CHUNKS * 64, dtype=torch.float64, device=device)
        self.launches = []
        self.ws = ops.new_workspace(device)          # own split-K scratch: the decode may run on another stream than the sampler
        with ops.workspace_scope(self.ws):
            (self._build_decoder if which == "dec" else self._build_encoder)(H, W)
        self.sd = None

    def w(self, key):
        return self.sd[key].to(self.dt).contiguous()

    def f32(self, key):
        return self.sd[key].contiguous()

    def _x3_ok(self, cin, ksize=3):
        return self.x3 and ops.x3_eligible(ksize * ksize * cin, cin)

    def _split(self, x):
        """fp32 residual-stream tensor -> its split-bf16 form (a conv input that no GroupNorm pass rewrites)."""
        B, H, W, c = x.shape
        s = self.pool.get((B, H, W, 2 * c), torch.bfloat16)
        self.launches.append(ops.split_bf16(x, s, name="split_bf16"))
        return s

    def _gn(self, x, key, silu=True, split=False):
        out = self.pool.get(tuple(x.shape[:3]) + (2 * x.shape[3],), torch.bfloat16) if split else self.pool.get(tuple(x.shape), self.dt)
        fused = None
        if self.gn_fuse:            # statistics from the epilogue of the GEMM that wrote x, when its tile plan allows
            prods = self.tracker.producers(x)
            if prods is not None:
                fused = ops.fuse_groupnorm_stats(x, prods)
        if fused is not None:
            self.launches += fused[2]
            self.launches.append(ops.groupnorm_apply(x, self.f32(key + ".weight"), self.f32(key + ".bias"), out, fused[0], fused[1],
                                                     eps=1e-6, silu=silu, split=split, name=key))
            self.gn_fused += 1
        else:
            self.launches += ops.groupnorm(x, self.f32(key + ".weight"), self.f32(key + ".bias"), out, self.gn_partial, eps=1e-6,
                                           silu=silu, split=split, name=key)
        return out

    def _conv3(self, key, x, cout, *, cin_pad=None, stride=1, pad=(1, 1), ups=0, residual=None, out=None):
        B, H, W, _ = x.shape
        Ho, Wo = (2 * H, 2 * W) if ups else ((H // 2, W // 2) if stride == 2 else (H, W))
        y = out if out is not None else self.pool.get((B, Ho, Wo, cout), self.dt)
        x3 = self.x3 and x.dtype == torch.bfloat16            # x is the split-bf16 form (from _gn(split=True) / _split)
        wp = ops.pack_conv_weight(self.sd[key + ".weight"], F32 if x3 else self.dt, cin_pad=cin_pad)
        if x3:
            wp = ops.pack_x3(wp)
            self.n_x3 += 1
        self._conv(x, wp, y, self.f32(key + ".bias"), stride=stride, pad=pad, ups=ups, residual=residual, x3=x3, name=key)
        return y

    def _conv(self, x, wp, y, bias, residual=None, **kw):
        """One convolution as 1, 2, 4 ... launches over batch slices: the direct-to-LDS main loop addresses its operand with 31-bit byte
        offsets, and the 512x512 level of a B = 8 decode holds 2.1 GB per tensor (fp32 or split-bf16 alike)."""
        B, n = x.shape[0], 1
        while (x.numel() // n) * x.element_size() >= 0x7fff0000 and B % (2 * n) == 0:
            n *= 2
        step = B // n
        for i in range(n):
            sl = slice(i * step, (i + 1) * step)
            self.launches.append(self.tracker.record(y[sl], ops.conv2d(x[sl], wp, y[sl], bias, residual=None if residual is None else residual[sl], **kw)))

    def _res(self, p, x, cin, cout):
        B, H, W, _ = x.shape
        t1 = self._gn(x, f"{p}.norm1", split=self._x3_ok(cin))
        h1 = self._conv3(f"{p}.conv1", t1, cout)
        self.pool.put(t1)
        t2 = self._gn(h1, f"{p}.norm2", split=self._x3_ok(cout))
        self.pool.put(h1)
        if cin != cout:
            sc = self.pool.get((B, H, W, cout), self.dt)
            if self._x3_ok(cin, 1):
                xs = self._split(x)
                self._conv(xs, ops.pack_x3(self.sd[f"{p}.nin_shortcut.weight"].reshape(cout, cin)), sc,
                           self.f32(f"{p}.nin_shortcut.bias"), ksize=1, pad=(0, 0), x3=True, name=f"{p}.nin_shortcut")
                self.pool.put(xs)
                self.n_x3 += 1
            else:
                self._conv(x, self.w(f"{p}.nin_shortcut.weight").reshape(cout, cin), sc,
                           self.f32(f"{p}.nin_shortcut.bias"), ksize=1, pad=(0, 0), name=f"{p}.nin_shortcut")
        else:
            sc = x
        y = self._conv3(f"{p}.conv2", t2, cout, residual=sc)
        self.pool.put(t2)
        if cin != cout:
            self.pool.put(sc)
        return y

    def _attn(self, p, x, c):
        B, H, W, _ = x.shape
        N = H * W
        M = B * N
        g = self._gn(x, f"{p}.norm", silu=False)
        g2 = g.view(M, c)
        wq, wk, wv = (self.sd[f"{p}.{n}.weight"].reshape(c, c) for n in ("q", "k", "v"))
        qk = self.pool.get((M, 2 * c), self.dt)
        self.launches.append(ops.linear(g2, torch.cat([wq, wk], 0).to(self.dt).contiguous(), qk,
                                        torch.cat([self.sd[f"{p}.q.bias"], self.sd[f"{p}.k.bias"]]).contiguous(), name=f"{p}.qk"))
        # V^T[b] = Wv . X_b^T   (rows = channels, columns = tokens)
        vt = self.pool.get((B, c, N), self.dt)
        self.launches.append(ops.conv_gemm(wv.to(self.dt).contiguous(), g2, vt, M=c, N=N, K=c, C0=c, ld0=c, Hin=1, Win=c, Hout=1, Wout=c,
                                           ldo=N, batch=B, sA=0, sW=N * c, sO=c * N, name=f"{p}.vT"))
        scores = self.pool.get((B, N, N), F32)
        self.launches.append(ops.conv_gemm(qk, qk[:, c:], scores, M=N, N=N, K=c, C0=c, ld0=2 * c, Hin=1, Win=N, Hout=1, Wout=N,
                                           ldo=N, alpha=float(int(c) ** (-0.5)), batch=B, sA=N * 2 * c, sW=N * 2 * c, sO=N * N, ldw=2 * c,
                                           name=f"{p}.scores"))
        self.launches.append(ops.softmax_rows(scores.view(B * N, N), name=f"{p}.softmax"))
        if self.dt != F32:
            probs = self.pool.get((B, N, N), self.dt)
            self.launches.append(ops.cast(scores, probs))
        else:
            probs = scores
        att = g     # GN output is dead after qk / vT
        self.launches.append(ops.conv_gemm(probs, vt, att, M=N, N=c, K=N, C0=N, ld0=N, Hin=1, Win=N, Hout=1, Wout=N, ldo=c,
                                           bias=self.f32(f"{p}.v.bias"), batch=B, sA=N * N, sW=c * N, sO=N * c, name=f"{p}.pv"))
        y = self.pool.get((B, H, W, c), self.dt)
        self.launches.append(self.tracker.record(y, ops.conv2d(att, self.w(f"{p}.proj_out.weight").reshape(c, c), y, self.f32(f"{p}.proj_out.bias"),
                                                               ksize=1, pad=(0, 0), residual=x, name=f"{p}.proj_out")))
        for t in (qk, vt, scores, att) + ((probs,) if probs is not scores else ()):
            self.pool.put(t)
        return y

    def _mid(self, p, h, c):
        a = self._res(f"{p}.block_1", h, c, c)
        self.pool.put(h)
        b = self._attn(f"{p}.attn_1", a, c)
        self.pool.put(a)
        d = self._res(f"{p}.block_2", b, c, c)
        self.pool.put(b)
        return d

    # ------------------------------------------------------------------ decoder (model.py:535-568)
    def _build_decoder(self, h, w):
        cfg, B, dev = self.cfg, self.B, self.dev
        ZP = 8
        self.z_in = torch.empty((B, cfg.embed_dim, h, w), dtype=F32, device=dev)         # NCHW latent (already sliced to 4 ch)
        z_cl = torch.zeros((B, h, w, ZP), dtype=self.dt, device=dev)
        self.launches.append(ops.nchw_to_nhwc(self.z_in, z_cl))
        zq = torch.zeros((B, h, w, ZP), dtype=self.dt, device=dev)                       # pad channels stay 0
        wpq = torch.zeros((cfg.z_channels, ZP), dtype=F32, device=dev)
        wpq[:, :cfg.embed_dim] = self.sd["post_quant_conv.weight"].reshape(cfg.z_channels, cfg.embed_dim)
        self.pq_alpha_launch_index = len(self.launches)
        self._pq = (z_cl, wpq.to(self.dt).contiguous(), zq)
        self.launches.append(None)      # placeholder: post_quant_conv with alpha = 1 / scale_factor, set by set_scale()
        nres = len(cfg.ch_mult)
        block_in = cfg.ch * cfg.ch_mult[nres - 1]
        hcur = self._conv3("decoder.conv_in", zq, block_in, cin_pad=ZP)
        hcur = self._mid("decoder.mid", hcur, block_in)
        for lvl in reversed(range(nres)):
            block_out = cfg.ch * cfg.ch_mult[lvl]
            for b in range(cfg.num_res_blocks + 1):
                nh = self._res(f"decoder.up.{lvl}.block.{b}", hcur, block_in, block_out)
                self.pool.put(hcur)
                hcur, block_in = nh, block_out
            if lvl != 0:
                hs = self._split(hcur) if self._x3_ok(block_in) else hcur
                nh = self._conv3(f"decoder.up.{lvl}.upsample.conv", hs, block_in, ups=1)
                if hs is not hcur:
                    self.pool.put(hs)
                self.pool.put(hcur)
                hcur = nh
        g = self._gn(hcur, "decoder.norm_out", split=self._x3_ok(block_in))
        self.pool.put(hcur)
        H, W = g.shape[1], g.shape[2]
        out4 = torch.empty((B, H, W, 4), dtype=F32, device=dev)
        self._conv3("decoder.conv_out", g, cfg.out_ch, out=out4[..., :cfg.out_ch])
        self.out = torch.empty((B, cfg.out_ch, H, W), dtype=F32, device=dev)
        self.launches.append(ops.nhwc_to_nchw(out4, self.out, C_=cfg.out_ch))
        self.set_scale(1.0)

    def set_scale(self, inv_scale):
        z_cl, wpq, zq = self._pq
        B, h, w, ZP = z_cl.shape
        self.launches[self.pq_alpha_launch_index] = ops.conv_gemm(
            z_cl, wpq, zq, M=B * h * w, N=self.cfg.z_channels, K=ZP, C0=ZP, ld0=ZP, Hin=1, Win=B * h * w, Hout=1, Wout=B * h * w,
            bias=self._pq_bias(), ldo=ZP, alpha=float(inv_scale), name="post_quant_conv")

    def _pq_bias(self):
        if not hasattr(self, "_pqb"):
            self._pqb = self.sd["post_quant_conv.bias"].contiguous()
        return self._pqb

    # ------------------------------------------------------------------ encoder (model.py:434-459)
    def _build_encoder(self, H, W):
        cfg, B, dev = self.cfg, self.B, self.dev
        XP = 8
        self.x_in = torch.empty((B, cfg.in_channels, H, W), dtype=F32, device=dev)
        x_cl = torch.zeros((B, H, W, XP), dtype=self.dt, device=dev)
        self.launches.append(ops.nchw_to_nhwc(self.x_in, x_cl))
        hcur = self._conv3("encoder.conv_in", x_cl, cfg.ch, cin_pad=XP)
        nres = len(cfg.ch_mult)
        in_mult = (1,) + cfg.ch_mult
        block_in = cfg.ch
        for lvl in range(nres):
            block_in = cfg.ch * in_mult[lvl]
            block_out = cfg.ch * cfg.ch_mult[lvl]
            for b in range(cfg.num_res_blocks):
                nh = self._res(f"encoder.down.{lvl}.block.{b}", hcur, block_in, block_out)
                self.pool.put(hcur)
                hcur, block_in = nh, block_out
            if lvl != nres - 1:
                hs = self._split(hcur) if self._x3_ok(block_in) else hcur
                nh = self._conv3(f"encoder.down.{lvl}.downsample.conv", hs, block_in, stride=2, pad=(0, 0))
                if hs is not hcur:
                    self.pool.put(hs)
                self.pool.put(hcur)
                hcur = nh
        hcur = self._mid("encoder.mid", hcur, block_in)
        g = self._gn(hcur, "encoder.norm_out", split=self._x3_ok(block_in))
        self.pool.put(hcur)
        zc = 2 * cfg.z_channels
        h8 = self._conv3("encoder.conv_out", g, zc)
        h, w = h8.shape[1], h8.shape[2]
        mom = torch.empty((B, h, w, 2 * cfg.embed_dim), dtype=F32, device=dev)
        self.launches.append(ops.conv2d(h8, self.w("quant_conv.weight").reshape(2 * cfg.embed_dim, zc), mom, self.f32("quant_conv.bias"),
                                        ksize=1, pad=(0, 0), name="quant_conv"))
        self.out = torch.empty((B, 2 * cfg.embed_dim, h, w), dtype=F32, device=dev)
        self.launches.append(ops.nhwc_to_nchw(mom, self.out))

    def run(self):
        ops.run(self.launches)


class AutoencoderKL(nn.Module):
    """Drop-in for ``ldm.models.autoencoder.AutoencoderKL`` (inference: encode / decode)."""

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, compute_dtype=None):
        super().__init__()
        dd = dict(ddconfig)
        assert dd["double_z"]
        self.cfg = VAEConfig(embed_dim=embed_dim, **dd)
        self.embed_dim, self.image_key = embed_dim, image_key
        self.compute_dtype = compute_dtype or torch.float32
        # how the fp32 decode multiplies: "bf16x3" (default) = split-bf16 operand pairs, three bf16 MFMA passes, fp32 accumulate and fp32
        # storage (pinned to the CPU oracle within the 1e-3 per-pixel gate by tests/test_fullsize_gpu.py); "f32" = exact fp32 MFMA
        self.decode_mode = os.environ.get("REFACE_VAE_DECODE", "bf16x3")
        tree = ParamTree(vae_param_specs(self.cfg))
        for name, child in tree.named_children():
            self.add_module(name, child)
        self._engines = {}
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu")["state_dict"]
            missing, _ = self.load_state_dict({k: v for k, v in sd.items() if not any(k.startswith(i) for i in ignore_keys)}, strict=False)
            lost = sorted(set(vae_param_specs(self.cfg)).intersection(missing))
            if lost:            # parameters are zero until loaded: a tensor the engine reads must come from the checkpoint
                raise RuntimeError(f"VAE checkpoint {ckpt_path} lacks {len(lost)} tensors the engine reads: {', '.join(lost[:6])}")

    def _engine(self, which, B, H, W):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("reface_amd.AutoencoderKL runs on the GPU only (HIP kernels; there is no CPU fallback)")
        # the encoder may run in its own dtype (throughput mode: bf16 encode of the masked target, fp32 decode of the result)
        dt = (getattr(self, "encode_dtype", None) or self.compute_dtype) if which == "enc" else self.compute_dtype
        if which == "dec" and dt == torch.float32 and self.decode_mode == "bf16x3":
            dt = "bf16x3"
        key = (which, B, H, W, dt, weights_version(self))
        eng = self._engines.get(key)
        if eng is None:
            self._engines = {k: v for k, v in self._engines.items() if k[-1] == key[-1]}
            eng = _VAEEngine(flat_state(self), self.cfg, B, H, W, dt, dev, which)
            self._engines[key] = eng
        return eng

    @torch.no_grad()
    def encode(self, x):
        B, _, H, W = x.shape
        eng = self._engine("enc", B, H, W)
        eng.x_in.copy_(x.to(dtype=F32))
        eng.run()
        return DiagonalGaussianDistribution(eng.out.clone())

    @torch.no_grad()
    def decode(self, z, inv_scale=1.0):
        """decoder(post_quant_conv(inv_scale * z)); ``inv_scale`` lets LatentDiffusion fold 1/scale_factor."""
        B, _, h, w = z.shape
        eng = self._engine("dec", B, h, w)
        if getattr(eng, "_inv_scale", None) != float(inv_scale):
            eng.set_scale(inv_scale)
            eng._inv_scale = float(inv_scale)
        eng.z_in.copy_(z.to(dtype=F32))
        eng.run()
        return eng.out.clone()

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
