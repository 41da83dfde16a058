"""Oracle: KL-VAE encoder / decoder over a reference-layout state dict (keys relative to
``first_stage_model.``).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows ldm/modules/diffusionmodules/model.py:38-202 (blocks), :434-459 (Encoder.forward),
:535-568 (Decoder.forward); ldm/models/autoencoder.py:324-333; ldm/modules/distributions/
distributions.py:24-37; ldm/models/diffusion/ddpm.py:850-857, 1277-1337.
"""
import torch
import torch.nn.functional as F


def _norm(sd, p, x):
    return F.group_norm(x, 32, sd[f"{p}.weight"], sd[f"{p}.bias"], 1e-6)   # model.py:38-39


def _swish(x):
    return x * torch.sigmoid(x)                                             # model.py:33-35


def resnet_block(sd, p, x):
    """model.py:122-141 (temb is None on this path)."""
    h = F.conv2d(_swish(_norm(sd, f"{p}.norm1", x)), sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.conv2d(_swish(_norm(sd, f"{p}.norm2", h)), sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.nin_shortcut.weight"], sd[f"{p}.nin_shortcut.bias"])
    return x + h


def attn_block(sd, p, x):
    """model.py:178-202: single head, scale C^-0.5, softmax over keys."""
    h_ = _norm(sd, f"{p}.norm", x)
    q = F.conv2d(h_, sd[f"{p}.q.weight"], sd[f"{p}.q.bias"])
    k = F.conv2d(h_, sd[f"{p}.k.weight"], sd[f"{p}.k.bias"])
    v = F.conv2d(h_, sd[f"{p}.v.weight"], sd[f"{p}.v.bias"])
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    h_ = F.conv2d(h_, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    return x + h_


def _mid(sd, p, h):
    h = resnet_block(sd, f"{p}.block_1", h)
    h = attn_block(sd, f"{p}.attn_1", h)
    return resnet_block(sd, f"{p}.block_2", h)


def encoder(sd, cfg, x):
    """model.py:434-459."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nres = len(cfg.ch_mult)
    for lvl in range(nres):
        for b in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"encoder.down.{lvl}.block.{b}", h)
        if lvl != nres - 1:      # model.py:72-76 asymmetric pad (0,1,0,1) + conv s2 p0
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"encoder.down.{lvl}.downsample.conv.weight"],
                         sd[f"encoder.down.{lvl}.downsample.conv.bias"], stride=2)
    h = _mid(sd, "encoder.mid", h)
    h = _swish(_norm(sd, "encoder.norm_out", h))
    return F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)


def encode_moments(sd, cfg, x):
    """autoencoder.py:324-328: Encoder -> quant_conv -> (mean, logvar clamp[-30,20])."""
    h = encoder(sd, cfg, x)
    m = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = torch.chunk(m, 2, dim=1)
    return mean, torch.clamp(logvar, -30.0, 20.0)


def first_stage_encoding(mean, logvar, eps, scale_factor=0.18215):
    """distributions.py:35-37 sample() with the caller's noise, then ddpm.py:857 scale."""
    return scale_factor * (mean + torch.exp(0.5 * logvar) * eps)


def decoder(sd, cfg, z):
    """model.py:535-568."""
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _mid(sd, "decoder.mid", h)
    nres = len(cfg.ch_mult)
    for lvl in reversed(range(nres)):
        for b in range(cfg.num_res_blocks + 1):
            h = resnet_block(sd, f"decoder.up.{lvl}.block.{b}", h)
        if lvl != 0:             # model.py:53-57 nearest x2 then conv 3x3
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up.{lvl}.upsample.conv.weight"],
                         sd[f"decoder.up.{lvl}.upsample.conv.bias"], padding=1)
    h = _swish(_norm(sd, "decoder.norm_out", h))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def decode_first_stage(sd, cfg, z, scale_factor=0.18215):
    """ddpm.py:1284 (z / scale), :1334-1335 (first 4 channels), autoencoder.py:330-333."""
    z = (1.0 / scale_factor) * z
    z = z[:, :4, :, :]
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return decoder(sd, cfg, z)
