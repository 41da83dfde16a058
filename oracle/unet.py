"""Oracle: UNetModel forward as pure functions over a reference-layout state dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Keys are relative to ``model.diffusion_model.``.
Follows ldm/modules/diffusionmodules/openaimodel.py:860-907 (forward), :255-275 (ResBlock),
:91-160 (Up/Downsample); ldm/modules/attention.py:179-289; ldm/modules/diffusionmodules/util.py:151-171.
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """util.py:151-166: [cos | sin], freqs = exp(-ln(max_period) * k / half), fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def time_embed(sd, t_emb):
    """openaimodel.py:632-636: Linear, SiLU, Linear."""
    h = F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    return F.linear(F.silu(h), sd["time_embed.2.weight"], sd["time_embed.2.bias"])


def group_norm(x, w, b, eps):
    return F.group_norm(x.float(), 32, w, b, eps).type(x.dtype)


def res_block(sd, p, x, emb):
    """openaimodel.py:255-275 (use_scale_shift_norm=False, no up/down); GroupNorm32 eps 1e-5."""
    h = group_norm(x, sd[f"{p}.in_layers.0.weight"], sd[f"{p}.in_layers.0.bias"], 1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.in_layers.2.weight"], sd[f"{p}.in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[f"{p}.emb_layers.1.weight"], sd[f"{p}.emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = group_norm(h, sd[f"{p}.out_layers.0.weight"], sd[f"{p}.out_layers.0.bias"], 1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.out_layers.3.weight"], sd[f"{p}.out_layers.3.bias"], padding=1)
    if f"{p}.skip_connection.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.skip_connection.weight"], sd[f"{p}.skip_connection.bias"])
    return x + h


def attention(sd, p, x, context, heads):
    """attention.py:179-221: q/k/v Linear (no bias), 'b n (h d) -> (b h) n d', scaled scores,
    softmax over keys, merge heads, to_out Linear+bias."""
    q = F.linear(x, sd[f"{p}.to_q.weight"])
    ctx = x if context is None else context
    k = F.linear(ctx, sd[f"{p}.to_k.weight"])
    v = F.linear(ctx, sd[f"{p}.to_v.weight"])
    b, n, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
    return F.linear(out, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def feed_forward(sd, p, x):
    """attention.py:37-64: GEGLU (value | gate chunk, exact erf GELU) then Linear."""
    h = F.linear(x, sd[f"{p}.net.0.proj.weight"], sd[f"{p}.net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    h = a * F.gelu(gate)
    return F.linear(h, sd[f"{p}.net.2.weight"], sd[f"{p}.net.2.bias"])


def spatial_transformer(sd, p, x, context, heads):
    """attention.py:278-289 + :239-243; Normalize = GroupNorm(32, eps=1e-6) (:76-77)."""
    b, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    x = F.conv2d(x, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = f"{p}.transformer_blocks.0"
    ln = lambda y, n: F.layer_norm(y, (c,), sd[f"{t}.{n}.weight"], sd[f"{t}.{n}.bias"], 1e-5)
    x = attention(sd, f"{t}.attn1", ln(x, "norm1"), None, heads) + x
    x = attention(sd, f"{t}.attn2", ln(x, "norm2"), context, heads) + x
    x = feed_forward(sd, f"{t}.ff", ln(x, "norm3")) + x
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    return x + x_in


def _run_block(sd, prefix, layers, h, emb, context):
    """TimestepEmbedSequential.forward (openaimodel.py:80-88)."""
    for j, l in enumerate(layers):
        p = f"{prefix}.{j}"
        if l[0] == "conv":
            h = F.conv2d(h, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
        elif l[0] == "res":
            h = res_block(sd, p, h, emb)
        elif l[0] == "st":
            h = spatial_transformer(sd, p, h, context, l[2])
        elif l[0] == "down":      # openaimodel.py:151-153 conv 3x3 stride 2 pad 1
            h = F.conv2d(h, sd[f"{p}.op.weight"], sd[f"{p}.op.bias"], stride=2, padding=1)
        elif l[0] == "up":        # openaimodel.py:116-118 nearest x2 then conv 3x3
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], padding=1)
    return h


def plan_from_shapes(shapes, num_heads=8):
    """The block structure, read off the CHECKPOINT itself: ``shapes`` = {state-dict key: shape} in the reference's key layout (what
    ``UNetModel.state_dict()`` of openaimodel.py:666-830 produces; tests/golden/unet_full_keys.json holds the reference's own list).
    Child j of ``input_blocks.i`` / ``output_blocks.i`` / ``middle_block`` is a ResBlock when it owns ``in_layers.2.weight`` (cin, cout from
    its shape), a SpatialTransformer when it owns ``proj_in.weight``, a Downsample when it owns ``op.weight``, an Upsample when it owns
    ``conv.weight``, the stem conv when the weight hangs on the child directly.  Independent of reface_amd.params.unet_plan (product
    code), which tests/test_oracle_golden.py compares it with.  Same tuple vocabulary as unet_forward's ``plan``."""
    def children(prefix):
        out = {}
        for k in shapes:
            if k.startswith(prefix + "."):
                out.setdefault(int(k[len(prefix) + 1:].split(".")[0]), []).append(k)
        return [out[j] for j in sorted(out)]

    def layers_of(prefix):
        ls = []
        for j, keys in enumerate(children(prefix)):
            q = f"{prefix}.{j}"
            if f"{q}.in_layers.2.weight" in shapes:
                co, ci = shapes[f"{q}.in_layers.2.weight"][:2]
                ls.append(("res", int(ci), int(co)))
            elif f"{q}.proj_in.weight" in shapes:
                ch = int(shapes[f"{q}.proj_in.weight"][0])
                ls.append(("st", ch, num_heads, ch // num_heads))
            elif f"{q}.op.weight" in shapes:
                ls.append(("down", int(shapes[f"{q}.op.weight"][0])))
            elif f"{q}.conv.weight" in shapes:
                ls.append(("up", int(shapes[f"{q}.conv.weight"][0])))
            elif f"{q}.weight" in shapes:
                co, ci = shapes[f"{q}.weight"][:2]
                ls.append(("conv", int(ci), int(co)))
            else:
                raise KeyError(f"unrecognised block {q}: {keys[:3]}")
        return ls

    n_in = 1 + max(int(k.split(".")[1]) for k in shapes if k.startswith("input_blocks."))
    n_out = 1 + max(int(k.split(".")[1]) for k in shapes if k.startswith("output_blocks."))
    return ([layers_of(f"input_blocks.{i}") for i in range(n_in)], layers_of("middle_block"), [layers_of(f"output_blocks.{i}") for i in range(n_out)])


def plan_of(sd, num_heads=8):
    """plan_from_shapes of a state dict (tensors)."""
    return plan_from_shapes({k: tuple(v.shape) for k, v in sd.items()}, num_heads)


def unet_forward(sd, plan, x, timesteps, context, model_channels=320):
    """openaimodel.py:860-907.  ``plan`` = the block structure: plan_from_shapes of the state dict (the oracle's own reading of the checkpoint),
    or reface_amd.params.unet_plan(cfg) -- tests/test_oracle_golden.py holds the two equal on the reference's key list."""
    ib, mid, ob = plan
    emb = time_embed(sd, timestep_embedding(timesteps, model_channels))
    hs = []
    h = x
    for i, layers in enumerate(ib):
        h = _run_block(sd, f"input_blocks.{i}", layers, h, emb, context)
        hs.append(h)
    h = _run_block(sd, "middle_block", mid, h, emb, context)
    for i, layers in enumerate(ob):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f"output_blocks.{i}", layers, h, emb, context)
    h = group_norm(h, sd["out.0.weight"], sd["out.0.bias"], 1e-5)
    return F.conv2d(F.silu(h), sd["out.2.weight"], sd["out.2.bias"], padding=1)
