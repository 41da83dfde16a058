"""The batch body every REFace caller runs (scripts/inference_test_bench.py:436-495 == inference_swap_selected.py:640-699 ==
one_inference.py / inference_swap_video.py): conditioning of the references -> VAE encode of the masked target -> 64x64 mask ->
DDIM / PLMS sampling -> fp32 VAE decode -> clamp to [0, 1].  One implementation, shared by the CLIs of this repository.
"""
import torch

from . import ops


class SwapRunner:
    """``model``: LatentDiffusion on the GPU; ``sampler``: DDIMSampler / PLMSSampler over it; ``opt``: the CLI namespace
    (ddim_steps, scale, ddim_eta, C, H, W, f)."""

    def __init__(self, model, sampler, opt):
        self.model, self.sampler, self.opt = model, sampler, opt
        self.device = torch.device("cuda")
        # --dump_tensors <dir> (an addition of this build's CLIs): every batch's inputs -- including the random draws the reference makes
        # implicitly (posterior noise on the CPU, distributions.py:36; x_T on the device, ddim.py:211) -- and intermediate results as
        # batch_<n>.npz, so that a test can run the CPU oracle chain on exactly the tensors a caller fed the engines
        self.dump_dir = getattr(opt, "dump_tensors", None)
        self.n_dumped = 0

    @torch.no_grad()
    def start_from_target(self, x0_img):
        """inference_test_bench.py:414-435: the noised latent of the target (or prior) image as x_T."""
        m, opt = self.model, self.opt
        z0 = m.get_first_stage_encoding(m.encode_first_stage(x0_img.to(self.device).float()))
        t0 = int(opt.target_start_noise_t)
        t_q = torch.randint(t0 - 1, t0, (z0.shape[0],), device="cpu").long()
        return m.q_sample(x_start=z0, t=t_q, noise=torch.randn_like(z0))

    @torch.no_grad()
    def run_batch(self, test_batch, test_model_kwargs, ref_imgs, start_code=None, landmarks136=None, log_every_t=100):
        """test_batch [B,3,H,W] in [-1,1] (host or device); test_model_kwargs: {"inpaint_image", "inpaint_mask"} on the device;
        ref_imgs [B,3,224,224] CLIP-normalised references on the device.  Returns (images [B,3,H,W] in [0,1] on the device,
        intermediates of the sampler)."""
        m, opt, dev = self.model, self.opt, self.device
        B = test_batch.shape[0]
        uc = m.learnable_vector.repeat(B, 1, 1) if opt.scale != 1.0 else None
        landmarks = m.get_landmarks(test_batch, landmarks136=landmarks136) if m.Landmark_cond else None
        c = m.conditioning_with_feat(ref_imgs.to(torch.float32), landmarks=landmarks, tar=test_batch.to(dev).to(torch.float32)).float()
        if len(c.shape) == 2:
            c = c.unsqueeze(1)
        inpaint_image, inpaint_mask = test_model_kwargs["inpaint_image"], test_model_kwargs["inpaint_mask"]
        post = m.encode_first_stage(inpaint_image)
        post_noise = torch.randn(post.mean.shape) if self.dump_dir else None          # (the draw get_first_stage_encoding makes itself otherwise)
        z_inpaint = m.get_first_stage_encoding(post, noise=post_noise).detach()
        h = z_inpaint.shape[-1]
        m64 = torch.empty((B, 1, h, h), dtype=torch.float32, device=dev)
        ops.bilinear_resize(inpaint_mask.float().contiguous(), m64)()        # torchvision Resize on a tensor (inference_test_bench.py:465)
        kw = dict(test_model_kwargs)
        kw["inpaint_image"], kw["inpaint_mask"] = z_inpaint, m64
        shape = [opt.C, opt.H // opt.f, opt.W // opt.f]
        x_T = None if start_code is None else start_code[:B]
        if x_T is None and self.dump_dir:
            x_T = torch.randn([B] + shape, device=dev)                                 # the draw the sampler makes itself otherwise
        samples, inter = self.sampler.sample(S=opt.ddim_steps, conditioning=c, batch_size=B, shape=shape, verbose=False,
                                             unconditional_guidance_scale=opt.scale, unconditional_conditioning=uc, eta=opt.ddim_eta,
                                             x_T=x_T, log_every_t=log_every_t, test_model_kwargs=kw)
        x_img = self.decode01(samples)
        if self.dump_dir:
            import os
            import numpy as np
            os.makedirs(self.dump_dir, exist_ok=True)
            f = lambda t: t.detach().float().cpu().numpy()
            np.savez(os.path.join(self.dump_dir, f"batch_{self.n_dumped:04d}.npz"), test_batch=f(test_batch), ref_imgs=f(ref_imgs),
                     inpaint_image=f(inpaint_image), inpaint_mask=f(inpaint_mask), post_noise=f(post_noise), x_T=f(x_T), c=f(c),
                     z_inpaint=f(z_inpaint), mask64=f(m64), samples=f(samples), x_img=f(x_img),
                     uc=f(uc) if uc is not None else np.zeros(0, np.float32))
            self.n_dumped += 1
        return x_img, inter

    @torch.no_grad()
    def decode01(self, latents):
        """clamp((decode(z) + 1) / 2, 0, 1) -- inference_test_bench.py:493-494."""
        x_dec = self.model.decode_first_stage(latents)
        x_img = torch.empty_like(x_dec)
        ops.to_image(x_dec, x_img)()
        return x_img

    @torch.no_grad()
    def resized_reference(self, ref_imgs, H, W):
        """The reference panel of the grid / <id>_ref.png: 224 -> image size, bilinear (torchvision Resize on a tensor, :523)."""
        ref = ref_imgs.float().contiguous()
        out = torch.empty((ref.shape[0], 3, H, W), dtype=torch.float32, device=ref.device)
        ops.bilinear_resize(ref, out)()
        return out
