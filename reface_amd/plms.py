"""PLMS sampler on the HIP UNet engine: mirror of ldm/models/diffusion/plms.py:116-237 (REFace's 9-channel variant, which
concatenates ``test_model_kwargs`` inpaint image / mask to x, plms.py:224-230).

Same engine, same fused pack / update kernels as the DDIM sampler; what changes is the step rule: the guided eps of the last
three steps is kept and combined by 2nd/3rd/4th-order Adams-Bashforth weights (`rf_combine3`), and the first step runs the
UNet twice (pseudo improved Euler).  eta must be 0 (plms.py:27-28).
"""
import numpy as np
import torch

from . import ops
from .ddim import DDIMSampler, F32


class PLMSSampler(DDIMSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        super().make_schedule(ddim_num_steps, ddim_discretize=ddim_discretize, ddim_eta=ddim_eta, verbose=verbose)

    def ddim_sampling(self, cond, shape, **kwargs):          # DDIMSampler.sample() dispatches here
        return self.plms_sampling(cond, shape, **kwargs)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, verbose=True, **kwargs):
        dev = self.model.device
        B, _, H, W = shape
        if "test_model_kwargs" not in kwargs:
            raise KeyError("test_model_kwargs")                 # plms.py:224 indexes it unconditionally
        tk = kwargs["test_model_kwargs"]
        z_inpaint, mask = tk["inpaint_image"], tk["inpaint_mask"]
        scale = float(unconditional_guidance_scale)
        cfg_on = not (unconditional_conditioning is None or scale == 1.)
        timesteps = self.ddim_timesteps
        total_steps = timesteps.shape[0]
        time_range = np.flip(timesteps)

        plan = self._plan(B, H, W, cfg_on, scale, False)
        eng, img, px0 = plan["eng"], plan["img"], plan["px0"]
        img.copy_(torch.randn(shape, device=dev) if x_T is None else x_T.to(device=dev, dtype=F32))
        for _ in range(total_steps):                              # plms.py:212: one (unused: sigma = 0) device draw per step -- keeps the generator
            torch.randn(shape, device=dev)                        # state of a seeded run aligned with the reference's from batch to batch
        plan["z"].copy_(z_inpaint.to(device=dev, dtype=F32))
        plan["m"].copy_(mask.to(device=dev, dtype=F32))
        c = cond.to(device=dev, dtype=F32)
        eng.set_context(torch.cat([unconditional_conditioning.to(device=dev, dtype=F32), c]) if cfg_on else c)
        t_all = torch.tensor(np.ascontiguousarray(time_range), dtype=F32, device=dev)
        table = torch.empty((total_steps, eng.E), dtype=F32, device=dev)
        ops.run(eng.make_emb_launches(t_all, table))
        coefs = torch.flip(self.ddim_coefs, dims=[0]).contiguous().to(dev)
        coefs = torch.cat([coefs, torch.zeros((total_steps, 3), dtype=F32, device=dev)], dim=1).contiguous()

        pack = plan["step"][0]
        eshape = (B, H, W, 4)                                    # the engine's eps layout (channels-last)
        ring = [torch.empty(eshape, dtype=F32, device=dev) for _ in range(4)]      # e_t of this and the last three steps
        tmp = torch.empty(eshape, dtype=F32, device=dev)
        e_prime = torch.empty(eshape, dtype=F32, device=dev)
        e_next = torch.empty(eshape, dtype=F32, device=dev)
        x_save = torch.empty_like(img)

        def model_output(row, dst):
            """UNet at timestep row `row` on the current img -> guided eps in dst (plms.py:190-204)."""
            eng.emb_table.copy_(table[row:row + 1])
            pack()
            eng.run()
            if cfg_on:
                e_u, e_c = eng.eps[:B], eng.eps[B:]
                ops.combine3(e_c, e_u, None, tmp, wa=1.0, wb=-1.0, wc=0.0, den=0.0)()          # e_c - e_u
                ops.combine3(e_u, tmp, None, dst, wa=1.0, wb=scale, wc=0.0, den=0.0)()         # e_u + s * (e_c - e_u)
            else:
                dst.copy_(eng.eps)

        def update(e):
            ops.ddim_update(e, img, px0, None, plan["coef"], cfg=False, scale=1.0)()

        intermediates = {"x_inter": [img.clone()], "pred_x0": [img.clone()]}
        if verbose:
            print(f"Running PLMS Sampling with {total_steps} timesteps")
        n_old = 0
        for i in range(total_steps):
            index = total_steps - i - 1
            e_t = ring[i % 4]
            old = [ring[(i - k) % 4] for k in (1, 2, 3)]
            plan["coef"].copy_(coefs[i])
            model_output(i, e_t)
            if n_old == 0:                                       # pseudo improved Euler (plms.py:227-231)
                x_save.copy_(img)
                update(e_t)
                model_output(min(i + 1, total_steps - 1), e_next)
                ops.combine3(e_t, e_next, None, e_prime, wa=1.0, wb=1.0, wc=0.0, den=2.0)()
                img.copy_(x_save)
            elif n_old == 1:
                ops.combine3(e_t, old[0], None, e_prime, wa=3.0, wb=-1.0, wc=0.0, den=2.0)()
            elif n_old == 2:
                ops.combine3(e_t, old[0], old[1], e_prime, wa=23.0, wb=-16.0, wc=5.0, den=12.0)()
            else:
                ops.combine3(e_t, old[0], old[1], tmp, wa=55.0, wb=-59.0, wc=37.0, den=0.0)()
                ops.combine3(tmp, old[2], None, e_prime, wa=1.0, wb=-9.0, wc=0.0, den=24.0)()
            update(e_prime)
            n_old = min(n_old + 1, 3)
            if callback:
                callback(i)
            if img_callback:
                img_callback(px0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img.clone())
                intermediates["pred_x0"].append(px0.clone())
        return img.clone(), intermediates
