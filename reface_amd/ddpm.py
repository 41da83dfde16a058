"""LatentDiffusion / DiffusionWrapper -- the REFace pipeline model (inference surface only).

Mirrors the attributes and methods that scripts/inference_test_bench.py:96-113, 330-334, 408-493 and
ldm/models/diffusion/ddim.py touch on ldm/models/diffusion/ddpm.py:574-2257:

  load_state_dict(sd, strict=False) / .cuda() / .eval() / .to(device) / ema_scope()
  learnable_vector, stack_feat, Landmark_cond, land_mark_id_seperate_layers, sep_head_att, proj_out
  get_landmarks(x), conditioning_with_feat(ref, landmarks=, tar=), get_learned_conditioning(c)
  encode_first_stage(x), get_first_stage_encoding(posterior), decode_first_stage(z), apply_model(x, t, c)
  num_timesteps, betas, alphas_cumprod, alphas_cumprod_prev, device, scale_factor

Training-only machinery (losses, optimizers, EMA, logging, patch fold/unfold) is out of scope.
All tensor arithmetic runs in the HIP engines of reface_amd.unet / vae / encoders.
"""
from contextlib import contextmanager

import numpy as np
import os

import torch
import torch.nn as nn

from . import ops
from .registry import cfg_get, instantiate_from_config
from .schedule import ddpm_buffers
from .vae import DiagonalGaussianDistribution

F32 = torch.float32
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _DeviceMixin:
    @property
    def device(self):
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")


class DiffusionWrapper(nn.Module, _DeviceMixin):
    """ddpm.py:2231-2257 (``crossattn`` conditioning only)."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, "concat", "crossattn", "hybrid", "adm"]

    def forward(self, x, t, c_concat=None, c_crossattn=None, return_features=False):
        if self.conditioning_key != "crossattn":
            raise NotImplementedError(f"conditioning_key={self.conditioning_key!r} is not on the REFace path")
        cc = torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc, return_features=return_features)


class Linear(nn.Module):
    """nn.Linear parameter names, HIP GEMM forward on [..., in] -> [..., out] fp32 tensors."""

    def __init__(self, fin, fout):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.zeros(fout, fin), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(fout), requires_grad=False)

    @torch.no_grad()
    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).to(F32)
        K = shp[-1]
        kp = (K + 3) // 4 * 4
        w = self.weight
        if kp != K:      # 136-d landmarks etc.: keep the 16-byte vector constraint
            x2 = torch.nn.functional.pad(x2, (0, kp - K))
            w = torch.nn.functional.pad(w, (0, kp - K))
        x2, w = x2.contiguous(), w.contiguous()
        out = torch.empty((x2.shape[0], self.out_features), dtype=F32, device=x.device)
        ops.linear(x2, w, out, self.bias)()
        return out.reshape(*shp[:-1], self.out_features)


class Identity(nn.Module):
    def forward(self, x):
        return x


class LatentDiffusion(nn.Module, _DeviceMixin):
    """Drop-in for ``ldm.models.diffusion.ddpm.LatentDiffusion`` on the inference path."""

    def __init__(self, first_stage_config, cond_stage_config, unet_config=None, num_timesteps_cond=None, cond_stage_key="image",
                 cond_stage_trainable=False, concat_mode=True, cond_stage_forward=None, conditioning_key=None, scale_factor=1.0,
                 scale_by_std=False, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 first_stage_key="image", image_size=256, channels=3, log_every_t=100, parameterization="eps", use_ema=True,
                 u_cond_percent=0, ckpt_path=None, ignore_keys=(), **training_only):
        super().__init__()
        assert parameterization == "eps"
        self.num_timesteps_cond = 1 if num_timesteps_cond is None else num_timesteps_cond
        assert self.num_timesteps_cond <= timesteps
        if conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        if cond_stage_config == "__is_unconditional__":
            conditioning_key = None
        if use_ema:
            raise NotImplementedError("use_ema=True (training-time EMA shadow weights) is not on the REFace inference path")
        self.parameterization, self.use_ema = parameterization, use_ema
        self.log_every_t, self.first_stage_key, self.image_size, self.channels = log_every_t, first_stage_key, image_size, channels
        self.cond_stage_key, self.cond_stage_trainable, self.cond_stage_forward = cond_stage_key, cond_stage_trainable, cond_stage_forward
        self.concat_mode, self.u_cond_percent, self.scale_by_std = concat_mode, u_cond_percent, scale_by_std
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        # schedule buffers (ddpm.py:255-307)
        for k, v in ddpm_buffers(timesteps, linear_start, linear_end).items():
            self.register_buffer(k, v)
        self.num_timesteps, self.linear_start, self.linear_end = int(timesteps), linear_start, linear_end

        # conditioning switches (ddpm.py:610-697)
        op = cfg_get(cond_stage_config, "other_params")
        if op is not None:
            self.clip_weight = cfg_get(op, "clip_weight")
            self.ID_weight = cfg_get(op, "ID_weight")
            self.Landmark_cond = cfg_get(op, "Landmark_cond")
            self.Landmarks_weight = cfg_get(op, "Landmarks_weight")
            ac = cfg_get(op, "Additional_config")
            self.multi_scale_ID = cfg_get(op, "multi_scale_ID", True)
            self.land_mark_id_seperate_layers = cfg_get(op, "land_mark_id_seperate_layers", False)
            self.sep_head_att = cfg_get(op, "sep_head_att", False)
            self.normalize = cfg_get(op, "normalize", False)
            self.concat_feat = cfg_get(op, "concat_feat", False)
            self.stack_feat = cfg_get(op, "stack_feat", False)
            self.weight_division = cfg_get(op, "weight_division", True)
            self.Target_CLIP_feat = cfg_get(ac, "Target_CLIP_feat", False)
            self.Source_CLIP_feat = cfg_get(ac, "Source_CLIP_feat", False)
            self.use_3dmm = cfg_get(ac, "use_3dmm", False)
            self.Landmark_loss_weight = cfg_get(ac, "Landmark_loss_weight", 0)
        else:
            self.clip_weight, self.ID_weight, self.Landmark_cond, self.Landmarks_weight = 1, 0, False, 0
            self.Landmark_loss_weight = 0
            self.multi_scale_ID = self.land_mark_id_seperate_layers = self.sep_head_att = self.normalize = False
            self.concat_feat = self.stack_feat = self.Target_CLIP_feat = self.Source_CLIP_feat = self.use_3dmm = False
            self.weight_division = True
        for flag in ("multi_scale_ID", "land_mark_id_seperate_layers", "sep_head_att", "normalize", "concat_feat", "stack_feat", "use_3dmm"):
            if getattr(self, flag):
                raise NotImplementedError(f"other_params.{flag}=True is not the shipped REFace configuration")
        self.update_weight = False
        self.learnable_vector = nn.Parameter(torch.randn((1, 1, 768)), requires_grad=False)
        if self.ID_weight > 0:
            from .encoders import IDLoss
            self.ID_proj_out = Linear(512, 768)
            self.face_ID_model = IDLoss(multiscale=False)
            # ddpm.py:101 of the reference: the ArcFace weights come from other_params.arcface_path BEFORE any checkpoint is applied
            self.arcface_path = cfg_get(op, "arcface_path", None) if op is not None else None
            self._arcface_loaded = False
            if self.arcface_path and os.path.exists(str(self.arcface_path)):
                self.face_ID_model.facenet.load_state_dict(torch.load(str(self.arcface_path), map_location="cpu"), strict=True)
                self._arcface_loaded = True
        self.detector = self.predictor = None
        if self.Landmark_cond or self.Landmark_loss_weight > 0:
            try:                                     # dlib is host-side and optional (ddpm.py:706-708)
                import dlib
                self.detector = dlib.get_frontal_face_detector()
                self.predictor = dlib.shape_predictor("Other_dependencies/DLIB_landmark_det/shape_predictor_68_face_landmarks.dat")
            except Exception as e:                   # noqa: BLE001
                print(f"[reface_amd] dlib landmark detector unavailable ({type(e).__name__}); "
                      "get_landmarks() will use caller-supplied landmarks or the no-face branch")
            self.landmark_proj_out = Linear(136, 768)
        target = cfg_get(cond_stage_config, "target")
        if target == "ldm.modules.encoders.modules.FrozenCLIPImageEmbedder":
            raise NotImplementedError("FrozenCLIPImageEmbedder is not the shipped REFace configuration")
        elif target == "ldm.modules.encoders.modules.FrozenCLIPEmbedder" and self.Source_CLIP_feat and self.Target_CLIP_feat:
            self.proj_out_source = Linear(768, 768)
            self.proj_out_target = Linear(768, 768)
            self.proj_out = Identity()
        elif target == "ldm.modules.encoders.modules.FrozenCLIPEmbedder":
            self.proj_out = Identity()
        if not scale_by_std:
            self.scale_factor = scale_factor
        else:
            self.register_buffer("scale_factor", torch.tensor(scale_factor))
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        self._sync_decode_mode(torch.float32)          # the model starts in the exact-fp32 parity mode: exact-fp32 decode too
        if cond_stage_config in ("__is_first_stage__", "__is_unconditional__"):
            raise NotImplementedError("REFace uses a CLIP cond stage")
        self.cond_stage_model = instantiate_from_config(cond_stage_config).eval()
        self.clip_denoised, self.bbox_tokenizer, self.restarted_from_ckpt = False, None, False
        for p in self.parameters():
            p.requires_grad = False
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu")
            sd = sd.get("state_dict", sd)
            missing, _ = self.load_state_dict({k: v for k, v in sd.items() if not any(k.startswith(i) for i in ignore_keys)}, strict=False)
            self.check_engine_weights(missing)
            self.restarted_from_ckpt = True

    def check_engine_weights(self, missing):
        """``strict=False`` tolerates a pruned checkpoint in the reference because its CLIP comes from ``CLIPModel.from_pretrained``
        and its ArcFace from ``other_params.arcface_path`` before the checkpoint is applied (ddpm.py:101, modules.py:230).  Here
        every parameter is zero until loaded, so a tensor the HIP engines read but the checkpoint lacks would silently stay zero:
        that is an error.  Keys the inference path never reads (text tower, training-only heads) may be absent."""
        from . import params as P
        need = set()
        need.update("model.diffusion_model." + k for k in P.unet_param_specs(self.model.diffusion_model.cfg))
        need.update("first_stage_model." + k for k in P.vae_param_specs(self.first_stage_model.cfg))
        need.update("cond_stage_model." + k for k in P.clip_param_specs(self.cond_stage_model.cfg))
        if hasattr(self, "face_ID_model") and not getattr(self, "_arcface_loaded", False):
            need.update("face_ID_model.facenet." + k for k in P.arcface_param_specs())
        need.update(k for k in P.cond_head_specs() if k in dict(self.named_parameters()))
        lost = sorted(need.intersection(missing))
        if lost:
            raise RuntimeError(f"checkpoint lacks {len(lost)} tensors the inference engines read (they would stay zero): "
                               + ", ".join(lost[:6]) + (" ..." if len(lost) > 6 else ""))

    # ------------------------------------------------------------------ plumbing
    @contextmanager
    def ema_scope(self, context=None):
        yield None                                   # use_ema is False on this path (ddpm.py:309-322)

    def set_compute_dtype(self, dtype, encoders=False):
        """Storage/MFMA dtype of the UNet (torch.float32 = exact-fp32 parity mode, torch.bfloat16 = throughput mode, torch.float16 = the throughput
        mode's kernels on fp16 operands).
        ``encoders=True`` also switches the CLIP ViT-L/14 / ArcFace towers and the VAE *encoder* (conditioning stage 4x faster; the
        conditioning vector and the inpaint latent then deviate ~1 % from fp32 -- throughput mode only; the VAE decode stays fp32)."""
        self.model.diffusion_model.set_compute_dtype(dtype)
        self._sync_decode_mode(dtype)
        if dtype in ("fp8", "fp8w", "fp8c"):       # fp8 GEMM operands are a UNet mode; the towers / VAE encoder take the bf16 activations' dtype
            dtype = torch.bfloat16
        elif dtype == "f32x3":             # split-bf16 UNet operands (fast parity mode): everything around it stays fp32
            dtype = torch.float32
        elif dtype == torch.float16:       # fp16 is a UNet mode (csrc/encoder.hip is bf16 / fp32): 16-bit towers / VAE encoder, if asked for, are bf16
            dtype = torch.bfloat16
        if encoders:
            for m in (getattr(self, "cond_stage_model", None), getattr(getattr(self, "face_ID_model", None), "facenet", None)):
                if m is not None and hasattr(m, "compute_dtype"):
                    m.compute_dtype = dtype
            if hasattr(self, "first_stage_model"):
                self.first_stage_model.encode_dtype = dtype        # the masked-target latent feeds a UNet of this dtype anyway

    def _sync_decode_mode(self, dtype):
        """The fp32 VAE decode follows the UNet mode unless REFACE_VAE_DECODE pins it: exact fp32 MFMA beside the exact-fp32 UNet (the mode the
        1e-3 gate is stated for), split-bf16 operand pairs (5e-5 from the oracle, 2.4x faster) beside every faster UNet mode."""
        fsm = getattr(self, "first_stage_model", None)
        if fsm is not None and hasattr(fsm, "decode_mode") and "REFACE_VAE_DECODE" not in os.environ:
            fsm.decode_mode = "f32" if dtype == torch.float32 else "bf16x3"

    # ------------------------------------------------------------------ conditioning (ddpm.py:859-1045, 1068-1099)
    def get_learned_conditioning(self, c):
        return self.cond_stage_model.encode(c)

    @torch.no_grad()
    def detect_landmarks(self, x):
        """Host half of get_landmarks (ddpm.py:1068-1096): dlib HOG detector + 68-point predictor per image on the CPU -> [B, 136]
        float tensor; zeros where no face is found or dlib is unavailable.  Touches no GPU state, so the CLI runs it for batch i+1
        on a worker thread while the GPU samples batch i."""
        lm = []
        if self.detector is not None and x is not None:
            img = (255.0 * ((x + 1.0) / 2.0).permute(0, 2, 3, 1).cpu().numpy()).astype(np.uint8)
            for i in range(len(img)):
                faces = self.detector(img[i], 1)
                if len(faces) == 0:
                    lm.append(np.zeros((1, 136), dtype=np.float32))
                    continue
                shape = self.predictor(img[i], faces[0])
                lm.append(np.array([[p.x, p.y] for p in shape.parts()]).reshape(1, 136))
        else:
            lm = [np.zeros((1, 136), dtype=np.float32) for _ in range(x.shape[0])]
        return torch.tensor(np.concatenate(lm, axis=0)).float()

    def get_landmarks(self, x, landmarks136=None):
        """Host-side dlib landmarks -> landmark_proj_out.  ``landmarks136`` ([B,136], e.g. a prefetched detect_landmarks result)
        bypasses the detection."""
        if landmarks136 is None:
            landmarks136 = self.detect_landmarks(x)
        landmarks136 = landmarks136.to(self.device)
        if self.Landmark_loss_weight > 0 and not self.Landmark_cond:
            return landmarks136
        return self.landmark_proj_out(landmarks136)

    @torch.no_grad()
    def conditioning_with_feat(self, x, landmarks=None, is_train=False, tar=None, tar_mask=None):
        """Live branches of ddpm.py:901-915 (source+target CLIP), :1009-1012 (ID), :1022-1039 (combine)."""
        from .encoders import combine_conditioning, target_to_clip_input
        if not (self.clip_weight > 0 and self.Source_CLIP_feat and self.Target_CLIP_feat and tar is not None):
            raise NotImplementedError("only the shipped branch (Source+Target CLIP, target image given) is implemented")
        c_src = self.proj_out_source(self.get_learned_conditioning(x))
        c_tar = self.proj_out_target(self.get_learned_conditioning(target_to_clip_input(tar.to(self.device))))
        c2 = None
        if self.ID_weight > 0:
            c2 = self.ID_proj_out(self.face_ID_model.extract_feats(x)[0]).unsqueeze(1)
        lm = None
        if self.Landmark_cond:
            lm = landmarks.unsqueeze(1) if landmarks.dim() != 3 else landmarks
        return combine_conditioning(c_src, c_tar, c2, lm, self.clip_weight, self.ID_weight,
                                    self.Landmarks_weight if self.Landmark_cond else 0.0, self.weight_division)

    # ------------------------------------------------------------------ first stage (ddpm.py:850-857, 1277-1337, 1402-1439)
    @torch.no_grad()
    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            return encoder_posterior.sample(noise=noise, scale=float(self.scale_factor))
        elif isinstance(encoder_posterior, torch.Tensor):
            raise NotImplementedError("tensor posteriors are not produced on the REFace path")
        raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        if predict_cids:
            raise NotImplementedError("VQ first stages are not on the REFace path")
        if self.first_stage_key == "inpaint":
            z = z[:, :4, :, :]
        return self.first_stage_model.decode(z.contiguous(), inv_scale=1.0 / float(self.scale_factor))

    # ------------------------------------------------------------------ UNet call (ddpm.py:1519-1617)
    def apply_model(self, x_noisy, t, cond, return_ids=False, return_features=False):
        if isinstance(cond, dict):
            pass
        else:
            if not isinstance(cond, list):
                cond = [cond]
            key = "c_concat" if self.model.conditioning_key == "concat" else "c_crossattn"
            cond = {key: cond}
        return self.model(x_noisy, t, **cond, return_features=return_features)

    @torch.no_grad()
    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:412-415: sqrt(acp[t]) * x_start + sqrt(1 - acp[t]) * noise, per sample (rf_combine3)."""
        x = x_start.to(device=self.device, dtype=torch.float32).contiguous()
        noise = torch.randn_like(x) if noise is None else noise.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        sa, s1 = self.sqrt_alphas_cumprod.detach().cpu(), self.sqrt_one_minus_alphas_cumprod.detach().cpu()
        for b, tb in enumerate(t.reshape(-1).tolist()):
            ops.combine3(x[b], noise[b], None, out[b], wa=float(sa[tb]), wb=float(s1[tb]), wc=0.0, den=0.0)()
        return out
