"""ORACLE -- TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline).

CPU fp32 restatement, in plain torch ops, of the graphs that execute inside the reference's single hot call
``self.pipe(...)`` (/root/reference/src/pipeline.py:261-272).  The arithmetic itself lives in third-party
packages that are NOT vendored in the reference and NOT installed here (diffusers 0.35.2, README.md:349-351),
so each function names the upstream module it restates (SURVEY.md 8c "Files a CPU restatement must follow").

Pinning status: CLIP text is pinned against the locally importable `transformers` implementation
(tests/test_oracle_cpu.py); op-level primitives are torch's own (the reference's declared dependency,
requirements.txt:2); LCM closed forms and the RNG fixture are pinned by SURVEY A.5 / 8a-RNG known answers.
The composite UNet / ControlNet / VAE graphs have no importable reference here: **parity unpinned** beyond
their published parameter counts (2.567 B / 1.251 B / 83.7 M, reproduced by the same tables).

All functions take a state dict ``sd`` with diffusers key names and a config dict (see the product's presets;
the oracle does not import the product).  NCHW fp32 throughout.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- primitives
def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def timestep_embedding(t, dim):
    """diffusers models/embeddings.py::get_timestep_embedding with flip_sin_to_cos=True, freq_shift=0
    (f_i = exp(-ln(10000) i / half), emb = [cos, sin])."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def resnet_block(sd, p, x, temb, groups, eps):
    """diffusers models/resnet.py::ResnetBlock2D (output_scale_factor 1, SiLU, time_embedding_norm default)."""
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps))
    h = _conv(sd, p + "conv1", h)
    if temb is not None:
        h = h + _lin(sd, p + "time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
    h = _conv(sd, p + "conv2", h)
    if p + "conv_shortcut.weight" in sd:
        x = _conv(sd, p + "conv_shortcut", x, padding=0)
    return x + h


def _mha(q, k, v, heads, mask=None):
    b, n, c = q.shape
    d = c // heads
    q = q.view(b, n, heads, d).transpose(1, 2)
    k = k.view(b, k.shape[1], heads, d).transpose(1, 2)
    v = v.view(b, v.shape[1], heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    if mask is not None:
        s = s + mask
    o = torch.softmax(s, dim=-1) @ v
    return o.transpose(1, 2).reshape(b, n, c)


def basic_transformer_block(sd, p, x, ctx, heads):
    """diffusers models/attention.py::BasicTransformerBlock (LN -> self-attn, LN -> cross-attn, LN -> GEGLU FF)."""
    c = x.shape[-1]
    h = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    a = _mha(_lin(sd, p + "attn1.to_q", h), _lin(sd, p + "attn1.to_k", h), _lin(sd, p + "attn1.to_v", h), heads)
    x = x + _lin(sd, p + "attn1.to_out.0", a)
    h = F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    a = _mha(_lin(sd, p + "attn2.to_q", h), _lin(sd, p + "attn2.to_k", ctx), _lin(sd, p + "attn2.to_v", ctx), heads)
    x = x + _lin(sd, p + "attn2.to_out.0", a)
    h = F.layer_norm(x, (c,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
    g = _lin(sd, p + "ff.net.0.proj", h)
    val, gate = g.chunk(2, dim=-1)                 # GEGLU: hidden * gelu(gate), exact erf GELU
    x = x + _lin(sd, p + "ff.net.2", val * F.gelu(gate))
    return x


def transformer2d(sd, p, x, ctx, depth, head_dim, groups):
    """diffusers models/transformers/transformer_2d.py::Transformer2DModel, use_linear_projection=True."""
    b, c, hh, ww = x.shape
    res = x
    h = F.group_norm(x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    h = _lin(sd, p + "proj_in", h)
    for k in range(depth):
        h = basic_transformer_block(sd, f"{p}transformer_blocks.{k}.", h, ctx, c // head_dim)
    h = _lin(sd, p + "proj_out", h)
    return h.reshape(b, hh, ww, c).permute(0, 3, 1, 2) + res


def _cond_embedding(sd, cfg, t, text_embeds, time_ids):
    """UNet2DConditionModel time + 'text_time' addition embedding."""
    ch0 = cfg["block_out_channels"][0]
    temb = timestep_embedding(t, ch0)
    temb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", temb)))
    tid = timestep_embedding(time_ids.flatten(), cfg["addition_time_embed_dim"]).reshape(time_ids.shape[0], -1)
    add = torch.cat([text_embeds, tid], dim=-1)
    aemb = _lin(sd, "add_embedding.linear_2", F.silu(_lin(sd, "add_embedding.linear_1", add)))
    return temb + aemb


def _down_and_mid(sd, cfg, x, emb, ctx, down_res=None, mid_res=None, collect=None):
    """conv_in output `x` -> skip stack + mid output.  Shared by UNet and ControlNet."""
    g, eps, hd = cfg["norm_num_groups"], cfg["norm_eps"], cfg["head_dim"]
    chans = cfg["block_out_channels"]
    skips = [x]
    for i in range(len(chans)):
        for j in range(cfg["layers_per_block"]):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            d = cfg["down_attn"][i][j]
            if d:
                x = transformer2d(sd, f"down_blocks.{i}.attentions.{j}.", x, ctx, d, hd, g)
            skips.append(x)
        if i != len(chans) - 1:
            x = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            skips.append(x)
    if down_res is not None:      # ControlNet residual i is added to stack entry i (A.1)
        skips = [s + r for s, r in zip(skips, down_res)]
    x = resnet_block(sd, "mid_block.resnets.0.", x, emb, g, eps)
    for k in range(1, cfg["mid_resnets"]):
        if cfg["mid_attn"]:
            x = transformer2d(sd, f"mid_block.attentions.{k - 1}.", x, ctx, cfg["mid_attn"], hd, g)
        x = resnet_block(sd, f"mid_block.resnets.{k}.", x, emb, g, eps)
    if mid_res is not None:
        x = x + mid_res
    return skips, x


def unet_forward(sd, cfg, sample, t, ctx, text_embeds, time_ids, down_res=None, mid_res=None, taps=None):
    """diffusers models/unets/unet_2d_condition.py::UNet2DConditionModel.forward (SDXL family)."""
    g, eps, hd = cfg["norm_num_groups"], cfg["norm_eps"], cfg["head_dim"]
    t = torch.as_tensor(t, dtype=torch.float32).reshape(-1).expand(sample.shape[0])
    emb = _cond_embedding(sd, cfg, t, text_embeds, time_ids)
    x = _conv(sd, "conv_in", sample)
    skips, x = _down_and_mid(sd, cfg, x, emb, ctx, down_res, mid_res)
    if taps is not None:
        taps["emb"] = emb
        taps["mid"] = x
    rev = list(reversed(cfg["block_out_channels"]))
    for i in range(len(rev)):
        for j in range(cfg["layers_per_block"] + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            d = cfg["up_attn"][i][j]
            if d:
                x = transformer2d(sd, f"up_blocks.{i}.attentions.{j}.", x, ctx, d, hd, g)
        if i != len(rev) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(F.group_norm(x, g, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    return _conv(sd, "conv_out", x)


def controlnet_forward(sd, cfg, sample, t, ctx, cond, scale, text_embeds, time_ids):
    """diffusers models/controlnets/controlnet.py::ControlNetModel.forward.
    Returns (list of 9 down residuals, mid residual), all multiplied by `scale`."""
    t = torch.as_tensor(t, dtype=torch.float32).reshape(-1).expand(sample.shape[0])
    emb = _cond_embedding(sd, cfg, t, text_embeds, time_ids)
    x = _conv(sd, "conv_in", sample)
    p = "controlnet_cond_embedding."
    c = F.silu(_conv(sd, p + "conv_in", cond))
    nb = len(cfg["conditioning_embedding_out_channels"]) - 1
    for i in range(nb):
        c = F.silu(_conv(sd, f"{p}blocks.{2 * i}", c))
        c = F.silu(_conv(sd, f"{p}blocks.{2 * i + 1}", c, stride=2))
    c = _conv(sd, p + "conv_out", c)
    x = x + c
    skips, x = _down_and_mid(sd, cfg, x, emb, ctx)
    down = [_conv(sd, f"controlnet_down_blocks.{i}", s, padding=0) * scale for i, s in enumerate(skips)]
    mid = _conv(sd, "controlnet_mid_block", x, padding=0) * scale
    return down, mid


# --------------------------------------------------------------------------- VAE
def _vae_attn(sd, p, x, groups, eps):
    """diffusers models/attention_processor.py::Attention as used by the VAE mid block: single head, d = C."""
    b, c, hh, ww = x.shape
    h = F.group_norm(x, groups, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], eps)
    h = h.reshape(b, c, hh * ww).transpose(1, 2)
    a = _mha(_lin(sd, p + "to_q", h), _lin(sd, p + "to_k", h), _lin(sd, p + "to_v", h), 1)
    a = _lin(sd, p + "to_out.0", a)
    return x + a.transpose(1, 2).reshape(b, c, hh, ww)


def _vae_mid(sd, side, x, g, eps):
    x = resnet_block(sd, f"{side}.mid_block.resnets.0.", x, None, g, eps)
    x = _vae_attn(sd, f"{side}.mid_block.attentions.0.", x, g, eps)
    return resnet_block(sd, f"{side}.mid_block.resnets.1.", x, None, g, eps)


def vae_encode_moments(sd, cfg, x):
    """diffusers models/autoencoders/vae.py::Encoder + quant_conv -> (mean, logvar)."""
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    ch = cfg["block_out_channels"]
    x = _conv(sd, "encoder.conv_in", x)
    for i in range(len(ch)):
        for j in range(cfg["layers_per_block"]):
            x = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}.", x, None, g, eps)
        if i != len(ch) - 1:
            x = F.pad(x, (0, 1, 0, 1))            # asymmetric pad, then stride-2 conv with no padding
            x = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", x, stride=2, padding=0)
    x = _vae_mid(sd, "encoder", x, g, eps)
    x = F.silu(F.group_norm(x, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    x = _conv(sd, "encoder.conv_out", x)
    x = _conv(sd, "quant_conv", x, padding=0)
    mean, logvar = x.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_decode(sd, cfg, z):
    """diffusers models/autoencoders/vae.py::Decoder behind post_quant_conv."""
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    rev = list(reversed(cfg["block_out_channels"]))
    x = _conv(sd, "post_quant_conv", z, padding=0)
    x = _conv(sd, "decoder.conv_in", x)
    x = _vae_mid(sd, "decoder", x, g, eps)
    for i in range(len(rev)):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}.", x, None, g, eps)
        if i != len(rev) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
    return _conv(sd, "decoder.conv_out", x)


# --------------------------------------------------------------------------- CLIP text
def clip_text_forward(sd, cfg, ids):
    """transformers models/clip/modeling_clip.py::CLIPTextTransformer (local copy v5.15.0, EOS pooling :561-582,
    projection :843,886-887).  Returns (hidden_states list of layers+1 entries, pooled) where pooled is the
    final-LN'd EOS state, projected when the config has a projection (CLIPTextModelWithProjection)."""
    b, n = ids.shape
    h = cfg["hidden"]
    x = sd["text_model.embeddings.token_embedding.weight"][ids] + \
        sd["text_model.embeddings.position_embedding.weight"][:n][None]
    mask = torch.full((n, n), float("-inf")).triu(1)
    hs = [x]
    for i in range(cfg["layers"]):
        p = f"text_model.encoder.layers.{i}."
        y = F.layer_norm(x, (h,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg["eps"])
        a = _mha(_lin(sd, p + "self_attn.q_proj", y), _lin(sd, p + "self_attn.k_proj", y),
                 _lin(sd, p + "self_attn.v_proj", y), cfg["heads"], mask)
        x = x + _lin(sd, p + "self_attn.out_proj", a)
        y = F.layer_norm(x, (h,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg["eps"])
        y = _lin(sd, p + "mlp.fc1", y)
        y = y * torch.sigmoid(1.702 * y) if cfg["act"] == "quick_gelu" else F.gelu(y)
        x = x + _lin(sd, p + "mlp.fc2", y)
        hs.append(x)
    last = F.layer_norm(x, (h,), sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"], cfg["eps"])
    if cfg["eos_token_id"] == 2:                  # legacy configs: argmax of ids
        pos = ids.argmax(dim=-1)
    else:                                         # first occurrence of eos_token_id
        pos = (ids == cfg["eos_token_id"]).int().argmax(dim=-1)
    pooled = last[torch.arange(b), pos]
    if cfg["projection_dim"]:
        pooled = F.linear(pooled, sd["text_projection.weight"])
    return hs, pooled
