"""Oracle: conditioning encoders (CLIP ViT-L/14 vision + REFace mapper, ArcFace IR-SE50) and the
conditioning combine.  TEST INFRASTRUCTURE (see oracle/__init__.py).

CLIP arithmetic lives in HF ``transformers`` (reference pins transformers==4.19.2,
requirements.txt:3; not under /root/reference) -- restated from the published CLIP ViT
algorithm (modeling_clip: CLIPVisionEmbeddings / CLIPEncoderLayer / CLIPAttention / CLIPMLP with
``quick_gelu``) and pinned against the installed transformers by tools/gen_golden.py.
Call sites: ldm/modules/encoders/modules.py:253-261; mapper: ldm/modules/encoders/xf.py:31-130.
ArcFace: src/Face_models/encoders/model_irse.py:44-69, helpers.py:15-119; wrapper
ldm/models/diffusion/ddpm.py:58-70, 112-124.  Combine: ddpm.py:901-915, 1009-1039.
"""
import math

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------ CLIP vision + mapper2
def clip_vision_pooled(sd, cfg, pixel_values):
    """HF CLIPVisionTransformer.forward -> pooler_output (post-LN CLS token)."""
    v = "model.vision_model"
    b = pixel_values.shape[0]
    h, heads = cfg.hidden, cfg.heads
    d = h // heads
    x = F.conv2d(pixel_values, sd[f"{v}.embeddings.patch_embedding.weight"], stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)                                  # [B, 256, h]
    cls = sd[f"{v}.embeddings.class_embedding"].expand(b, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[f"{v}.embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (h,), sd[f"{v}.pre_layrnorm.weight"], sd[f"{v}.pre_layrnorm.bias"], 1e-5)
    n = x.shape[1]
    for i in range(cfg.layers):
        p = f"{v}.encoder.layers.{i}"
        r = x
        y = F.layer_norm(x, (h,), sd[f"{p}.layer_norm1.weight"], sd[f"{p}.layer_norm1.bias"], 1e-5)
        q = F.linear(y, sd[f"{p}.self_attn.q_proj.weight"], sd[f"{p}.self_attn.q_proj.bias"])
        k = F.linear(y, sd[f"{p}.self_attn.k_proj.weight"], sd[f"{p}.self_attn.k_proj.bias"])
        vv = F.linear(y, sd[f"{p}.self_attn.v_proj.weight"], sd[f"{p}.self_attn.v_proj.bias"])
        sp = lambda t: t.view(b, n, heads, d).transpose(1, 2)
        w = torch.matmul(sp(q), sp(k).transpose(-1, -2)) * (d ** -0.5)
        w = F.softmax(w, dim=-1, dtype=torch.float32)
        o = torch.matmul(w, sp(vv)).transpose(1, 2).reshape(b, n, h)
        x = r + F.linear(o, sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
        r = x
        y = F.layer_norm(x, (h,), sd[f"{p}.layer_norm2.weight"], sd[f"{p}.layer_norm2.bias"], 1e-5)
        y = F.linear(y, sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"])
        y = y * torch.sigmoid(1.702 * y)                               # quick_gelu
        x = r + F.linear(y, sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"])
    pooled = x[:, 0, :]
    return F.layer_norm(pooled, (h,), sd[f"{v}.post_layernorm.weight"], sd[f"{v}.post_layernorm.bias"], 1e-5)


def mapper2(sd, cfg, z):
    """xf.py:80-130 on a 1-token sequence (n_ctx=1, heads=1): the attention is the full
    restatement (scale d^-1/4 on q and k, fp32 softmax), not the 1-token shortcut."""
    w = cfg.proj
    for i in range(cfg.mapper_layers):
        p = f"mapper2.resblocks.{i}"
        y = F.layer_norm(z.float(), (w,), sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"], 1e-5)
        qkv = F.linear(y, sd[f"{p}.attn.c_qkv.weight"], sd[f"{p}.attn.c_qkv.bias"])
        bs, n_ctx, width = qkv.shape
        attn_ch = width // 1 // 3
        scale = 1 / math.sqrt(math.sqrt(attn_ch))
        qkv = qkv.view(bs, n_ctx, 1, -1)
        q, k, v = torch.split(qkv, attn_ch, dim=-1)
        wt = torch.einsum("bthc,bshc->bhts", q * scale, k * scale)
        wt = torch.softmax(wt.float(), dim=-1)
        a = torch.einsum("bhts,bshc->bthc", wt, v).reshape(bs, n_ctx, -1)
        z = z + F.linear(a, sd[f"{p}.attn.c_proj.weight"], sd[f"{p}.attn.c_proj.bias"])
        y = F.layer_norm(z.float(), (w,), sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"], 1e-5)
        y = F.gelu(F.linear(y, sd[f"{p}.mlp.c_fc.weight"], sd[f"{p}.mlp.c_fc.bias"]))
        z = z + F.linear(y, sd[f"{p}.mlp.c_proj.weight"], sd[f"{p}.mlp.c_proj.bias"])
    return z


def clip_embed(sd, cfg, image):
    """modules.py:253-261: vision_model -> pooler_output -> visual_projection -> unsqueeze(1)
    -> mapper2 -> final_ln2."""
    z = clip_vision_pooled(sd, cfg, image)
    z = F.linear(z, sd["model.visual_projection.weight"])
    z = z.unsqueeze(1)
    z = mapper2(sd, cfg, z)
    return F.layer_norm(z.float(), (cfg.proj,), sd["final_ln2.weight"], sd["final_ln2.bias"], 1e-5)


# ------------------------------------------------------------------ ArcFace IR-SE50
def _bn2d(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd.get(f"{p}.weight"),
                        sd.get(f"{p}.bias"), False, 0.0, eps)


def arcface_backbone(sd, units, x):
    """model_irse.py:44-69 (multi_scale=False), eval mode; keys relative to facenet."""
    x = F.conv2d(x, sd["input_layer.0.weight"], padding=1)
    x = _bn2d(sd, "input_layer.1", x)
    x = F.prelu(x, sd["input_layer.2.weight"])
    for i, (cin, depth, stride) in enumerate(units):
        p = f"body.{i}"
        if cin == depth:
            sc = F.max_pool2d(x, 1, stride)                            # helpers.py:101
        else:
            sc = F.conv2d(x, sd[f"{p}.shortcut_layer.0.weight"], stride=stride)
            sc = _bn2d(sd, f"{p}.shortcut_layer.1", sc)
        r = _bn2d(sd, f"{p}.res_layer.0", x)
        r = F.conv2d(r, sd[f"{p}.res_layer.1.weight"], padding=1)
        r = F.prelu(r, sd[f"{p}.res_layer.2.weight"])
        r = F.conv2d(r, sd[f"{p}.res_layer.3.weight"], stride=stride, padding=1)
        r = _bn2d(sd, f"{p}.res_layer.4", r)
        s = F.adaptive_avg_pool2d(r, 1)                                # SE: helpers.py:56-72
        s = F.relu(F.conv2d(s, sd[f"{p}.res_layer.5.fc1.weight"]))
        s = torch.sigmoid(F.conv2d(s, sd[f"{p}.res_layer.5.fc2.weight"]))
        x = r * s + sc
    x = _bn2d(sd, "output_layer.0", x)
    x = x.reshape(x.shape[0], -1)
    x = F.linear(x, sd["output_layer.3.weight"], sd["output_layer.3.bias"])
    x = F.batch_norm(x, sd["output_layer.4.running_mean"], sd["output_layer.4.running_var"],
                     sd.get("output_layer.4.weight"), sd.get("output_layer.4.bias"), False, 0.0, 1e-5)
    return x / torch.norm(x, 2, 1, True)                               # helpers.py:15-18


def arcface_preprocess(x):
    """ddpm.py:112-121: un_norm_clip -> (x-0.5)/0.5 -> pool 256 -> crop -> pool 112."""
    x = x * 1.0
    for c in range(3):
        x[:, c] = x[:, c] * CLIP_STD[c] + CLIP_MEAN[c]
    x = (x - 0.5) / 0.5
    if x.shape[2] != 256:
        x = F.adaptive_avg_pool2d(x, (256, 256))
    x = x[:, :, 35:223, 32:220]
    return F.adaptive_avg_pool2d(x, (112, 112))


def extract_id_feats(sd, units, ref_img):
    return arcface_backbone(sd, units, arcface_preprocess(ref_img))


# ------------------------------------------------------------------ conditioning combine
def target_to_clip_input(tar):
    """ddpm.py:907-912: (tar+1)/2 -> CLIP normalise -> bilinear resize to 224 (tensor input =>
    no antialias, align_corners=False)."""
    t = (tar * 1.0 + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    t = (t - mean) / std
    return F.interpolate(t, size=(224, 224), mode="bilinear", align_corners=False, antialias=False)


def conditioning_with_feat(heads, clip_sd, clip_cfg, arc_sd, arc_units, ref, landmarks136, tar,
                           clip_weight=1.0, id_weight=10.0, lm_weight=0.05):
    """ddpm.py:901-915 (source+target CLIP), :1009-1012 (ID), :1096 + :1022 (landmarks),
    :1038-1039 (weighted mean).  ``heads`` holds the top-level projection tensors."""
    c_src = F.linear(clip_embed(clip_sd, clip_cfg, ref), heads["proj_out_source.weight"], heads["proj_out_source.bias"])
    c_tar = F.linear(clip_embed(clip_sd, clip_cfg, target_to_clip_input(tar)),
                     heads["proj_out_target.weight"], heads["proj_out_target.bias"])
    c = c_src + c_tar
    c2 = F.linear(extract_id_feats(arc_sd, arc_units, ref), heads["ID_proj_out.weight"], heads["ID_proj_out.bias"]).unsqueeze(1)
    lm = F.linear(landmarks136, heads["landmark_proj_out.weight"], heads["landmark_proj_out.bias"]).unsqueeze(1)
    return (c * clip_weight + c2 * id_weight + lm * lm_weight) / (clip_weight + id_weight + lm_weight)


def mask64(inpaint_mask):
    """inference_test_bench.py:465: torchvision Resize([64,64]) on a tensor == bilinear, no
    antialias, align_corners=False."""
    h = inpaint_mask.shape[-1] // 8
    return F.interpolate(inpaint_mask, size=(h, h), mode="bilinear", align_corners=False, antialias=False)


def to_uint8_image(x):
    """inference_test_bench.py:494-495, 536-537: clamp((x+1)/2,0,1) -> HWC -> trunc(255*x)."""
    x = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
    x = x.permute(0, 2, 3, 1).numpy()
    return (255.0 * x).astype("uint8")
