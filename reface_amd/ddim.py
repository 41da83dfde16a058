"""DDIMSampler -- the REFace sampling loop on the HIP UNet engine.

Same surface as ldm/models/diffusion/ddim.py:96-375: ``DDIMSampler(model).sample(S=..., batch_size=...,
shape=..., conditioning=..., unconditional_guidance_scale=..., unconditional_conditioning=...,
eta=..., x_T=..., test_model_kwargs={'inpaint_image', 'inpaint_mask'}) -> (samples, intermediates)``.

MI355X design: the step body [pack 9-channel input (x2 for CFG) -> UNet -> CFG + DDIM update] is a
fixed launch list captured once into a HIP graph and replayed S times.  Everything that depends on
the step index lives in two small device buffers refreshed before each replay: the per-timestep
ResBlock embedding vectors (precomputed for all S timesteps by one GEMM chain) and five fp32
update coefficients.  Cross-attention reduces to a per-sample vector computed once per call.
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
        """Build (once per shape) the engine, the S-independent step launch list and its graph."""
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
        plan = dict(eng=eng, img=img, z=z, m=m, px0=px0, noise=noise, coef=coef, step=step, graph=None)
        self._plans = {key: plan}
        return plan

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
        total_steps = timesteps.shape[0]
        time_range = np.flip(timesteps)
        sig_host = np.asarray(self.ddim_sigmas, dtype=np.float64)
        with_noise = bool((sig_host != 0).any())

        plan = self._plan(B, H, W, cfg_on, unconditional_guidance_scale, with_noise)
        eng, img = plan["eng"], plan["img"]
        img.copy_(torch.randn(shape, device=dev) if x_T is None else x_T.to(device=dev, dtype=F32))
        plan["z"].copy_(z_inpaint.to(device=dev, dtype=F32))
        plan["m"].copy_(mask.to(device=dev, dtype=F32))
        c = cond.to(device=dev, dtype=F32)
        eng.set_context(torch.cat([unconditional_conditioning.to(device=dev, dtype=F32), c]) if cfg_on else c)

        # all S timestep-embedding rows in one shot (timesteps are data-independent)
        t_all = torch.tensor(np.ascontiguousarray(time_range), dtype=F32, device=dev)
        table = torch.empty((total_steps, eng.E), dtype=F32, device=dev)
        ops.run(eng.make_emb_launches(t_all, table))
        coefs = torch.flip(self.ddim_coefs, dims=[0]).contiguous().to(dev)      # row i <-> index S-1-i
        coefs = torch.cat([coefs, torch.zeros((total_steps, 3), dtype=F32, device=dev)], dim=1).contiguous()

        if self.use_graph and plan["graph"] is None:
            stream = torch.cuda.Stream()
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                ops.run(plan["step"])            # warm-up outside capture (module load, attribute calls)
                img.copy_(torch.randn(shape, device=dev) if x_T is None else x_T.to(device=dev, dtype=F32))
            torch.cuda.current_stream().wait_stream(stream)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                ops.run(plan["step"])
            plan["graph"] = g
            img.copy_(torch.randn(shape, device=dev) if x_T is None else x_T.to(device=dev, dtype=F32))

        intermediates = {"x_inter": [img.clone()], "pred_x0": [img.clone()]}
        if verbose:
            print(f"Running DDIM Sampling with {total_steps} timesteps")
        for i in range(total_steps):
            index = total_steps - i - 1
            eng.emb_table.copy_(table[i:i + 1])
            plan["coef"].copy_(coefs[i])
            if with_noise:
                plan["noise"].copy_(x_noise[i].to(dev) if x_noise is not None else torch.randn(shape, device=dev))
            if plan["graph"] is not None:
                plan["graph"].replay()
            else:
                ops.run(plan["step"])
            if callback:
                callback(i)
            if img_callback:
                img_callback(plan["px0"], i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img.clone())
                intermediates["pred_x0"].append(plan["px0"].clone())
        return img.clone(), intermediates
