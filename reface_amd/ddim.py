"""DDIMSampler -- the REFace sampling loop on the HIP UNet engine.

Same surface as ldm/models/diffusion/ddim.py:96-375: ``DDIMSampler(model).sample(S=..., batch_size=...,
shape=..., conditioning=..., unconditional_guidance_scale=..., unconditional_conditioning=...,
eta=..., x_T=..., test_model_kwargs={'inpaint_image', 'inpaint_mask'}) -> (samples, intermediates)``.

MI355X design: the step body [pack 9-channel input (x2 for CFG) -> UNet -> CFG + DDIM update] is a
fixed launch list; the whole S-step loop is unrolled into ONE HIP graph (REFACE_GRAPH_STEPS caps the steps
per graph; 1 = one replay per step).  Everything that depends on the step index lives in two small device tables the
unrolled steps index by position: the per-timestep ResBlock embedding vectors (precomputed for all S
timesteps by one GEMM chain) and five fp32 update coefficients per step -- so a sample() call is
three small copies and one graph replay.  Cross-attention reduces to a per-sample vector computed once per call.
"""
import numpy as np
import os

import torch

from . import ops
from .schedule import ddim_step_coefficients, make_ddim_sampling_parameters, make_ddim_timesteps

F32 = torch.float32


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.use_graph = kwargs.get("use_graph", os.environ.get("REFACE_NO_GRAPH", "0") != "1")      # counter collection cannot trace graph replays
        self._plans = {}

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """ddim.py:110-139 (only the buffers the sampling path reads are registered)."""
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        ac = alphas_cumprod.detach().to(device="cpu", dtype=F32)
        self.register_buffer("betas", self.model.betas.detach().clone().to(F32))
        self.register_buffer("alphas_cumprod", ac.clone())
        self.register_buffer("alphas_cumprod_prev", self.model.alphas_cumprod_prev.detach().clone().to(F32))
        ddim_sigmas, ddim_alphas, ddim_alphas_prev = make_ddim_sampling_parameters(alphacums=ac, ddim_timesteps=self.ddim_timesteps,
                                                                                   eta=ddim_eta, verbose=verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = ddim_sigmas, ddim_alphas, ddim_alphas_prev
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1. - ddim_alphas)
        self.ddim_coefs = ddim_step_coefficients(ddim_alphas, ddim_alphas_prev, ddim_sigmas)     # [S, 5] fp32 (host)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, src_im=None, tar=None, **kwargs):
        if conditioning is not None:
            cbs = conditioning[list(conditioning.keys())[0]].shape[0] if isinstance(conditioning, dict) else conditioning.shape[0]
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        for name, val, ok in (("mask", mask, None), ("x0", x0, None), ("score_corrector", score_corrector, None),
                              ("quantize_x0", quantize_x0, False)):
            if val is not ok and val != ok:
                raise NotImplementedError(f"DDIMSampler.sample({name}=...) is not on the REFace inference path")
        if noise_dropout != 0. or temperature != 1.:
            raise NotImplementedError("noise_dropout / temperature are not on the REFace inference path")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f"Data shape for DDIM sampling is {size}, eta {eta}")
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose, **kwargs)

    # ------------------------------------------------------------------------------------------
    def _plan(self, B, H, W, cfg_on, scale, with_noise):
        """Build (once per shape) the engine and the S-independent step launch list."""
        unet = self.model.model.diffusion_model
        key = (B, H, W, cfg_on, float(scale), with_noise, unet.compute_dtype, id(unet))
        plan = self._plans.get(key)
        if plan is not None and plan["eng"] is unet.engine(B * (2 if cfg_on else 1), H, W, uniform_t=True, cfg_pair=cfg_on):
            return plan
        dev = self.model.device
        nb = B * (2 if cfg_on else 1)
        eng = unet.engine(nb, H, W, uniform_t=True, cfg_pair=cfg_on)
        img = torch.empty((B, 4, H, W), dtype=F32, device=dev)
        z = torch.empty((B, 4, H, W), dtype=F32, device=dev)
        m = torch.empty((B, 1, H, W), dtype=F32, device=dev)
        px0 = torch.empty((B, 4, H, W), dtype=F32, device=dev)
        noise = torch.zeros((B, 4, H, W), dtype=F32, device=dev) if with_noise else None
        coef = torch.zeros(8, dtype=F32, device=dev)
        step = [ops.ddim_pack_input(img, z, m, eng.x_in, dup=2 if cfg_on else 1)]
        step += eng.main
        step.append(ops.ddim_update(eng.eps, img, px0, noise, coef, cfg=cfg_on, scale=scale))
        plan = dict(eng=eng, img=img, z=z, m=m, px0=px0, noise=noise, coef=coef, step=step, graphs={}, cfg_on=cfg_on, scale=scale,
                    x_start=torch.empty_like(img), warm=False)
        self._plans = {key: plan}
        return plan

    @staticmethod
    def _chunk(S, flagged, want):
        """DDIM steps per captured graph: the largest divisor of S that is <= `want` -- provided every step whose state the
        caller wants back (intermediates) is either inside a single whole-loop graph or at a chunk end; else 1."""
        c = max(d for d in range(1, S + 1) if S % d == 0 and d <= max(1, want))
        if c == S or all((i + 1) % c == 0 for i in flagged):
            return c
        return 1

    def _capture(self, plan, c, flagged_local):
        """One HIP graph of `c` unrolled DDIM steps.  Step j of the chunk reads row j of the chunk's timestep-embedding table
        and coefficient block (and its own noise slab when eta > 0): the per-step pointers are patched into the prepared
        descriptors while the launches are recorded (kernel parameters are copied at capture time)."""
        eng, dev = plan["eng"], self.model.device
        B, _, H, W = plan["img"].shape
        E = eng.E
        g = dict(tab=torch.zeros((c, E), dtype=F32, device=dev), coef=torch.zeros((c, 8), dtype=F32, device=dev),
                 noise=(torch.zeros((c,) + tuple(plan["img"].shape), dtype=F32, device=dev) if plan["noise"] is not None else None),
                 inter_x=[torch.empty_like(plan["img"]) for _ in flagged_local], inter_p=[torch.empty_like(plan["img"]) for _ in flagged_local])
        lo = eng.emb_table.data_ptr()
        hi = lo + eng.emb_table.numel() * 4
        patch = [(l.keep[0], l.keep[0].rowvec) for l in eng.main
                 if l.fn.__name__ == "rf_conv_gemm" and l.keep[0].rowvec and lo <= l.keep[0].rowvec < hi]
        updates = [ops.ddim_update(eng.eps, plan["img"], plan["px0"], None if g["noise"] is None else g["noise"][j], g["coef"][j],
                                   cfg=plan["cfg_on"], scale=plan["scale"]) for j in range(c)]
        pack = plan["step"][0]
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                for j in range(c):
                    for d, orig in patch:
                        d.rowvec = g["tab"].data_ptr() + j * E * 4 + (orig - lo)
                    pack()
                    ops.run(eng.main)
                    updates[j]()
                    if j in flagged_local:
                        k = flagged_local.index(j)
                        g["inter_x"][k].copy_(plan["img"])
                        g["inter_p"][k].copy_(plan["px0"])
        finally:
            for d, orig in patch:
                d.rowvec = orig
        g["graph"], g["keep"] = graph, updates
        return g

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, verbose=True, x_noise=None, **kwargs):
        """ddim.py:200-251 + :323-375.  ``x_noise`` (optional, [S, B, 4, h, w]) supplies the per-step N(0,1)
        draws when eta > 0 (otherwise they are drawn with torch.randn on the device)."""
        dev = self.model.device
        B, _, H, W = shape
        if "test_model_kwargs" in kwargs:
            tk = kwargs["test_model_kwargs"]
            z_inpaint, mask = tk["inpaint_image"], tk["inpaint_mask"]
        elif "rest" in kwargs:
            z_inpaint, mask = kwargs["rest"][:, :4], kwargs["rest"][:, 4:5]
        else:
            raise Exception("kwargs must contain either 'test_model_kwargs' or 'rest' key")
        cfg_on = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        timesteps = self.ddim_timesteps
        S = total_steps = timesteps.shape[0]
        time_range = np.flip(timesteps)
        sig_host = np.asarray(self.ddim_sigmas, dtype=np.float64)
        with_noise = bool((sig_host != 0).any())

        plan = self._plan(B, H, W, cfg_on, unconditional_guidance_scale, with_noise)
        eng, img, x_start = plan["eng"], plan["img"], plan["x_start"]
        # the start noise is drawn ONCE (ddim.py:213: one torch.randn per call), whatever the graph mode does afterwards
        x_start.copy_(torch.randn(shape, device=dev) if x_T is None else x_T.to(device=dev, dtype=F32))
        img.copy_(x_start)
        plan["z"].copy_(z_inpaint.to(device=dev, dtype=F32))
        plan["m"].copy_(mask.to(device=dev, dtype=F32))
        c = cond.to(device=dev, dtype=F32)
        eng.set_context(torch.cat([unconditional_conditioning.to(device=dev, dtype=F32), c]) if cfg_on else c)

        # all S timestep-embedding rows in one shot (timesteps are data-independent)
        t_all = torch.tensor(np.ascontiguousarray(time_range), dtype=F32, device=dev)
        table = torch.empty((S, eng.E), dtype=F32, device=dev)
        ops.run(eng.make_emb_launches(t_all, table))
        coefs = torch.flip(self.ddim_coefs, dims=[0]).contiguous().to(dev)      # row i <-> index S-1-i
        coefs = torch.cat([coefs, torch.zeros((S, 3), dtype=F32, device=dev)], dim=1).contiguous()
        # The reference draws torch.randn(shape) on the device at EVERY step, used or not (ddim.py:371 via util.py:264-267: at eta = 0 the
        # draw is multiplied by sigma = 0).  The same S separate draws are made here, up front and in the same order behind the x_T draw, so
        # that the device generator leaves a sample() call in the reference's state (seeded runs stay aligned from batch to batch) and, at
        # eta > 0, the steps see the reference's numbers.
        if x_noise is None:
            draws = [torch.randn(shape, device=dev) for _ in range(S)]
        if with_noise:
            noise_all = x_noise.to(device=dev, dtype=F32) if x_noise is not None else torch.stack(draws)

        flagged = [i for i in range(S) if (S - i - 1) % log_every_t == 0 or i == 0]       # ddim.py:247 (index == total_steps - 1 <=> i == 0)
        per_step_host = bool(callback or img_callback)
        c_steps = 1
        if self.use_graph and not per_step_host:
            c_steps = self._chunk(S, flagged, int(os.environ.get("REFACE_GRAPH_STEPS", "1000")))
        g = None
        if self.use_graph:
            if not plan["warm"]:
                # one eager step outside capture (module load, function attributes), on real coefficients; state restored after
                eng.emb_table.copy_(table[0:1])
                plan["coef"].copy_(coefs[0])
                stream = torch.cuda.Stream()
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):
                    ops.run(plan["step"])
                torch.cuda.current_stream().wait_stream(stream)
                img.copy_(x_start)
                plan["warm"] = True
            fl_local = [i for i in flagged if i < c_steps] if c_steps == S else []
            gkey = (c_steps, tuple(fl_local))
            g = plan["graphs"].get(gkey)
            if g is None:
                g = self._capture(plan, c_steps, fl_local)
                plan["graphs"] = {gkey: g}
                img.copy_(x_start)

        intermediates = {"x_inter": [img.clone()], "pred_x0": [img.clone()]}
        if verbose:
            print(f"Running DDIM Sampling with {total_steps} timesteps")
        if g is not None:
            for r in range(S // c_steps):
                i0, i1 = r * c_steps, (r + 1) * c_steps
                g["tab"].copy_(table[i0:i1])
                g["coef"].copy_(coefs[i0:i1])
                if with_noise:
                    g["noise"].copy_(noise_all[i0:i1])
                g["graph"].replay()
                for i in range(i0, i1):
                    if callback:
                        callback(i)
                    if img_callback:
                        img_callback(plan["px0"], i)
                if c_steps == S:
                    for k in range(len(g["inter_x"])):
                        intermediates["x_inter"].append(g["inter_x"][k].clone())
                        intermediates["pred_x0"].append(g["inter_p"][k].clone())
                elif (i1 - 1) in flagged:
                    intermediates["x_inter"].append(img.clone())
                    intermediates["pred_x0"].append(plan["px0"].clone())
        else:
            for i in range(S):
                eng.emb_table.copy_(table[i:i + 1])
                plan["coef"].copy_(coefs[i])
                if with_noise:
                    plan["noise"].copy_(noise_all[i])
                ops.run(plan["step"])
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(plan["px0"], i)
                if i in flagged:
                    intermediates["x_inter"].append(img.clone())
                    intermediates["pred_x0"].append(plan["px0"].clone())
        return img.clone(), intermediates
