"""Parameter tables (diffusers / transformers key names) and weight sources.

The reference obtains weights with hub fetches (src/pipeline.py:89-154) which cannot run offline.  This module
supplies the two replacements:

* ``synth_state_dict(cfg, seed, ...)``  -- seeded, variance-preserving synthetic weights (SURVEY 8d "Weights").
* ``load_dir(path)``                     -- a local diffusers-layout directory (config.json + *.safetensors).

Keys follow the upstream naming so that real checkpoints drop in unchanged.  ``fold_lora`` restates what
``pipe.load_lora_weights`` (src/pipeline.py:154) does at run time, but folds it once: W += (alpha/r) * B @ A.
"""
import json
import math
import os

import torch

from .presets import time_embed_dim

# init kinds: "w" weight N(0, 1/fan_in); "wo" residual-branch output weight (scaled down); "z" zero-conv
# (N(0, 0.02^2) so the residual path is exercised); "b" bias 0; "g" norm gamma 1; "e" embedding N(0, 0.02^2)


def _resnet(p, cin, cout, temb):
    t = [(p + "norm1.weight", (cin,), "g"), (p + "norm1.bias", (cin,), "b"),
         (p + "conv1.weight", (cout, cin, 3, 3), "w"), (p + "conv1.bias", (cout,), "b")]
    if temb:
        t += [(p + "time_emb_proj.weight", (cout, temb), "w"), (p + "time_emb_proj.bias", (cout,), "b")]
    t += [(p + "norm2.weight", (cout,), "g"), (p + "norm2.bias", (cout,), "b"),
          (p + "conv2.weight", (cout, cout, 3, 3), "wo"), (p + "conv2.bias", (cout,), "b")]
    if cin != cout:
        # UNet shortcuts are 1x1 convs stored 4-D upstream
        t += [(p + "conv_shortcut.weight", (cout, cin, 1, 1), "w"), (p + "conv_shortcut.bias", (cout,), "b")]
    return t


def _transformer2d(p, c, depth, xdim):
    t = [(p + "norm.weight", (c,), "g"), (p + "norm.bias", (c,), "b"),
         (p + "proj_in.weight", (c, c), "w"), (p + "proj_in.bias", (c,), "b")]
    for k in range(depth):
        q = f"{p}transformer_blocks.{k}."
        t += [(q + "norm1.weight", (c,), "g"), (q + "norm1.bias", (c,), "b"),
              (q + "attn1.to_q.weight", (c, c), "w"), (q + "attn1.to_k.weight", (c, c), "w"),
              (q + "attn1.to_v.weight", (c, c), "w"),
              (q + "attn1.to_out.0.weight", (c, c), "wo"), (q + "attn1.to_out.0.bias", (c,), "b"),
              (q + "norm2.weight", (c,), "g"), (q + "norm2.bias", (c,), "b"),
              (q + "attn2.to_q.weight", (c, c), "w"), (q + "attn2.to_k.weight", (c, xdim), "w"),
              (q + "attn2.to_v.weight", (c, xdim), "w"),
              (q + "attn2.to_out.0.weight", (c, c), "wo"), (q + "attn2.to_out.0.bias", (c,), "b"),
              (q + "norm3.weight", (c,), "g"), (q + "norm3.bias", (c,), "b"),
              (q + "ff.net.0.proj.weight", (8 * c, c), "w"), (q + "ff.net.0.proj.bias", (8 * c,), "b"),
              (q + "ff.net.2.weight", (c, 4 * c), "wo"), (q + "ff.net.2.bias", (c,), "b")]
    t += [(p + "proj_out.weight", (c, c), "wo"), (p + "proj_out.bias", (c,), "b")]
    return t


def _embeddings(cfg):
    ch0 = cfg["block_out_channels"][0]
    te = time_embed_dim(cfg)
    pin = cfg["projection_class_embeddings_input_dim"]
    return [("time_embedding.linear_1.weight", (te, ch0), "w"), ("time_embedding.linear_1.bias", (te,), "b"),
            ("time_embedding.linear_2.weight", (te, te), "w"), ("time_embedding.linear_2.bias", (te,), "b"),
            ("add_embedding.linear_1.weight", (te, pin), "w"), ("add_embedding.linear_1.bias", (te,), "b"),
            ("add_embedding.linear_2.weight", (te, te), "w"), ("add_embedding.linear_2.bias", (te,), "b")]


def _encoder_half(cfg):
    """conv_in + down blocks + mid block, shared by UNet and ControlNet."""
    chans = cfg["block_out_channels"]
    te = time_embed_dim(cfg)
    xd = cfg["cross_attention_dim"]
    t = [("conv_in.weight", (chans[0], cfg["in_channels"], 3, 3), "w"), ("conv_in.bias", (chans[0],), "b")]
    t += _embeddings(cfg)
    cin = chans[0]
    for i, cout in enumerate(chans):
        for j in range(cfg["layers_per_block"]):
            t += _resnet(f"down_blocks.{i}.resnets.{j}.", cin, cout, te)
            cin = cout
            d = cfg["down_attn"][i][j]
            if d:
                t += _transformer2d(f"down_blocks.{i}.attentions.{j}.", cout, d, xd)
        if i != len(chans) - 1:
            t += [(f"down_blocks.{i}.downsamplers.0.conv.weight", (cout, cout, 3, 3), "w"),
                  (f"down_blocks.{i}.downsamplers.0.conv.bias", (cout,), "b")]
    c = chans[-1]
    t += _resnet("mid_block.resnets.0.", c, c, te)
    for k in range(1, cfg["mid_resnets"]):
        if cfg["mid_attn"]:
            t += _transformer2d(f"mid_block.attentions.{k - 1}.", c, cfg["mid_attn"], xd)
        t += _resnet(f"mid_block.resnets.{k}.", c, c, te)
    return t


def unet_table(cfg):
    chans = cfg["block_out_channels"]
    te = time_embed_dim(cfg)
    xd = cfg["cross_attention_dim"]
    t = _encoder_half(cfg)
    # skip-stack channel list in push order (SURVEY A.1 "Skip stack")
    skips = [chans[0]]
    for i, cout in enumerate(chans):
        skips += [cout] * cfg["layers_per_block"]
        if i != len(chans) - 1:
            skips.append(cout)
    rev = list(reversed(chans))
    prev = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            skip = skips.pop()
            t += _resnet(f"up_blocks.{i}.resnets.{j}.", prev + skip, cout, te)
            prev = cout
            d = cfg["up_attn"][i][j]
            if d:
                t += _transformer2d(f"up_blocks.{i}.attentions.{j}.", cout, d, xd)
        if i != len(rev) - 1:
            t += [(f"up_blocks.{i}.upsamplers.0.conv.weight", (cout, cout, 3, 3), "w"),
                  (f"up_blocks.{i}.upsamplers.0.conv.bias", (cout,), "b")]
    t += [("conv_norm_out.weight", (chans[0],), "g"), ("conv_norm_out.bias", (chans[0],), "b"),
          ("conv_out.weight", (cfg["out_channels"], chans[0], 3, 3), "wo"), ("conv_out.bias", (cfg["out_channels"],), "b")]
    return t


def controlnet_table(cfg):
    chans = cfg["block_out_channels"]
    t = _encoder_half(cfg)
    emb = cfg["conditioning_embedding_out_channels"]
    p = "controlnet_cond_embedding."
    t += [(p + "conv_in.weight", (emb[0], cfg["conditioning_channels"], 3, 3), "w"), (p + "conv_in.bias", (emb[0],), "b")]
    for i in range(len(emb) - 1):
        t += [(f"{p}blocks.{2 * i}.weight", (emb[i], emb[i], 3, 3), "w"), (f"{p}blocks.{2 * i}.bias", (emb[i],), "b"),
              (f"{p}blocks.{2 * i + 1}.weight", (emb[i + 1], emb[i], 3, 3), "w"),
              (f"{p}blocks.{2 * i + 1}.bias", (emb[i + 1],), "b")]
    t += [(p + "conv_out.weight", (chans[0], emb[-1], 3, 3), "z"), (p + "conv_out.bias", (chans[0],), "b")]
    skips = [chans[0]]
    for i, cout in enumerate(chans):
        skips += [cout] * cfg["layers_per_block"]
        if i != len(chans) - 1:
            skips.append(cout)
    for i, c in enumerate(skips):
        t += [(f"controlnet_down_blocks.{i}.weight", (c, c, 1, 1), "z"), (f"controlnet_down_blocks.{i}.bias", (c,), "b")]
    t += [("controlnet_mid_block.weight", (chans[-1], chans[-1], 1, 1), "z"), ("controlnet_mid_block.bias", (chans[-1],), "b")]
    return t


def _vae_attn(p, c):
    t = [(p + "group_norm.weight", (c,), "g"), (p + "group_norm.bias", (c,), "b")]
    for n, k in (("to_q", "w"), ("to_k", "w"), ("to_v", "w"), ("to_out.0", "wo")):
        t += [(f"{p}{n}.weight", (c, c), k), (f"{p}{n}.bias", (c,), "b")]
    return t


def vae_table(cfg):
    ch = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    lc = cfg["latent_channels"]
    t = [("encoder.conv_in.weight", (ch[0], cfg["in_channels"], 3, 3), "w"), ("encoder.conv_in.bias", (ch[0],), "b")]
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(L):
            t += _resnet(f"encoder.down_blocks.{i}.resnets.{j}.", cin, cout, 0)
            cin = cout
        if i != len(ch) - 1:
            t += [(f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (cout, cout, 3, 3), "w"),
                  (f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (cout,), "b")]
    c = ch[-1]
    for side in ("encoder", "decoder"):
        if side == "decoder":
            t += [("decoder.conv_in.weight", (c, lc, 3, 3), "w"), ("decoder.conv_in.bias", (c,), "b")]
        t += _resnet(f"{side}.mid_block.resnets.0.", c, c, 0)
        t += _vae_attn(f"{side}.mid_block.attentions.0.", c)
        t += _resnet(f"{side}.mid_block.resnets.1.", c, c, 0)
        if side == "encoder":
            t += [("encoder.conv_norm_out.weight", (c,), "g"), ("encoder.conv_norm_out.bias", (c,), "b"),
                  ("encoder.conv_out.weight", (2 * lc, c, 3, 3), "w"), ("encoder.conv_out.bias", (2 * lc,), "b"),
                  ("quant_conv.weight", (2 * lc, 2 * lc, 1, 1), "w"), ("quant_conv.bias", (2 * lc,), "b"),
                  ("post_quant_conv.weight", (lc, lc, 1, 1), "w"), ("post_quant_conv.bias", (lc,), "b")]
    rev = list(reversed(ch))
    cin = rev[0]
    for i, cout in enumerate(rev):
        for j in range(L + 1):
            t += _resnet(f"decoder.up_blocks.{i}.resnets.{j}.", cin, cout, 0)
            cin = cout
        if i != len(rev) - 1:
            t += [(f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (cout, cout, 3, 3), "w"),
                  (f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (cout,), "b")]
    t += [("decoder.conv_norm_out.weight", (ch[0],), "g"), ("decoder.conv_norm_out.bias", (ch[0],), "b"),
          ("decoder.conv_out.weight", (cfg["out_channels"], ch[0], 3, 3), "wo"),
          ("decoder.conv_out.bias", (cfg["out_channels"],), "b")]
    return t


def clip_table(cfg):
    h, f = cfg["hidden"], cfg["intermediate"]
    t = [("text_model.embeddings.token_embedding.weight", (cfg["vocab_size"], h), "e"),
         ("text_model.embeddings.position_embedding.weight", (cfg["max_positions"], h), "e")]
    for i in range(cfg["layers"]):
        p = f"text_model.encoder.layers.{i}."
        t += [(p + "layer_norm1.weight", (h,), "g"), (p + "layer_norm1.bias", (h,), "b")]
        for n, k in (("q_proj", "w"), ("k_proj", "w"), ("v_proj", "w"), ("out_proj", "wo")):
            t += [(f"{p}self_attn.{n}.weight", (h, h), k), (f"{p}self_attn.{n}.bias", (h,), "b")]
        t += [(p + "layer_norm2.weight", (h,), "g"), (p + "layer_norm2.bias", (h,), "b"),
              (p + "mlp.fc1.weight", (f, h), "w"), (p + "mlp.fc1.bias", (f,), "b"),
              (p + "mlp.fc2.weight", (h, f), "wo"), (p + "mlp.fc2.bias", (h,), "b")]
    t += [("text_model.final_layer_norm.weight", (h,), "g"), ("text_model.final_layer_norm.bias", (h,), "b")]
    if cfg["projection_dim"]:
        t += [("text_projection.weight", (cfg["projection_dim"], h), "w")]
    return t


_TABLES = {"unet": unet_table, "controlnet": controlnet_table, "vae": vae_table, "clip": clip_table}


def param_table(cfg):
    return _TABLES[cfg["kind"]](cfg)


def param_count(cfg):
    return sum(math.prod(s) for _, s, _ in param_table(cfg))


def synth_state_dict(cfg, seed=1234, device="cpu", dtype=torch.float32, out_scale=0.35):
    """Seeded synthetic weights.  Drawn in fp32 on `device` from one generator, in table order, then cast.
    Residual-branch output projections are scaled by `out_scale` so ~70 stacked blocks stay inside fp16."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape, kind in param_table(cfg):
        if kind == "g":
            w = torch.ones(shape, device=device)
        elif kind == "b":
            w = torch.zeros(shape, device=device)
        else:
            w = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
            if kind in ("w", "wo"):
                fan_in = math.prod(shape[1:])
                w *= (1.0 / math.sqrt(fan_in)) * (out_scale if kind == "wo" else 1.0)
            else:  # "z", "e"
                w *= 0.02
        sd[name] = w.to(dtype)
    return sd


def empty_state_dict(cfg, device="cpu", dtype=torch.float32):
    """Uninitialised tensors with the table's names/shapes (receive buffers for the weight broadcast)."""
    return {name: torch.empty(shape, device=device, dtype=dtype) for name, shape, _ in param_table(cfg)}


LORA_FILE_NAMES = ("pytorch_lora_weights.safetensors", "lcm_lora.safetensors")       # hub name first (latent-consistency/lcm-lora-sdxl)


def _lora_pairs(lora_sd, module_names):
    """Group a LoRA state dict into {module: (A/down, B/up, alpha or None)} for the UNet, whatever the key scheme:

      * PEFT / current diffusers:   [unet.]<module>.lora_A.weight / .lora_B.weight  (+ optional <module>.alpha)
      * older diffusers:            [unet.]<module>.lora.down.weight / .lora.up.weight, and the attention-processor form
                                    <block>.processor.to_q_lora.down.weight -> module <block>.to_q
      * kohya (sd-scripts):         lora_unet_<module with '.' -> '_'>.lora_down.weight / .lora_up.weight / .alpha

    `module_names` (the target state dict's module paths) resolves the kohya underscore names.  Text-encoder adapters
    (`text_encoder.`, `lora_te1_`, `lora_te2_`) are not part of the reference's LCM-LoRA and are skipped.  Returns
    (pairs, skipped_keys); unet-looking keys that resolve to no module raise."""
    flat = {m.replace(".", "_"): m for m in module_names}
    pairs, skipped, unresolved = {}, [], []
    suffixes = ((".lora_A.weight", "a"), (".lora_B.weight", "b"), (".lora.down.weight", "a"), (".lora.up.weight", "b"),
                (".lora_down.weight", "a"), (".lora_up.weight", "b"), (".lora_A.default.weight", "a"),
                (".lora_B.default.weight", "b"), (".down.weight", "a"), (".up.weight", "b"), (".alpha", "alpha"))
    for k, v in lora_sd.items():
        role = None
        for suf, r in suffixes:
            if k.endswith(suf):
                base, role = k[: -len(suf)], r
                break
        if role is None:
            skipped.append(k)
            continue
        if base.startswith(("text_encoder", "lora_te")):
            skipped.append(k)
            continue
        if base.startswith("lora_unet_"):
            mod = flat.get(base[len("lora_unet_"):])
        else:
            mod = base[len("unet."):] if base.startswith("unet.") else base
            if ".processor." in mod and mod.endswith("_lora"):          # attn1.processor.to_q_lora -> attn1.to_q
                mod = mod.replace(".processor.", ".")[: -len("_lora")]
                if mod.endswith(".to_out"):
                    mod += ".0"
            if mod not in module_names:
                mod = None
        if mod is None:
            unresolved.append(k)
            continue
        pairs.setdefault(mod, {})[role] = v
    if unresolved:
        hint = " (SGM-named kohya files, lora_unet_input_blocks_*, are not supported: convert to diffusers names)" \
            if any("input_blocks" in k or "output_blocks" in k for k in unresolved) else ""
        raise ValueError(f"LoRA: {len(unresolved)} keys match no UNet module, e.g. {unresolved[:3]}{hint}")
    bad = [m for m, d in pairs.items() if "a" not in d or "b" not in d]
    if bad:
        raise ValueError(f"LoRA: incomplete down/up pair for {bad[:3]}")
    return pairs, skipped


def fold_lora(sd, lora_sd, scale=1.0):
    """W += scale * (alpha/r) * B @ A for every adapter pair of `lora_sd` (what `pipe.load_lora_weights`,
    /root/reference/src/pipeline.py:154, evaluates at run time through peft, folded once).  Key schemes: _lora_pairs.
    Returns the number of folded modules."""
    modules = {k[: -len(".weight")] for k in sd if k.endswith(".weight")}
    pairs, _ = _lora_pairs(lora_sd, modules)
    for mod, d in pairs.items():
        a, b = d["a"].float(), d["b"].float()
        r = a.shape[0]
        alpha = float(d["alpha"]) if "alpha" in d else float(r)
        w = sd[mod + ".weight"]
        if a.dim() == 4:  # conv: A is k x k (r, cin, k, k), B is 1 x 1 (cout, r, 1, 1)
            delta = torch.einsum("or,rikl->oikl", b.flatten(1), a)
        else:
            delta = b @ a
        if delta.numel() != w.numel():
            raise ValueError(f"LoRA: {mod}: adapter shape {tuple(delta.shape)} does not fit weight {tuple(w.shape)}")
        sd[mod + ".weight"] = (w.float() + scale * (alpha / r) * delta.reshape(w.shape).to(w.device)).to(w.dtype)
    return len(pairs)


def find_lora(root):
    """The LCM-LoRA file of a weights directory: <root>/<name> or <root>/lcm_lora/<name> for the names above."""
    for d in (root, os.path.join(root, "lcm_lora"), os.path.join(root, "lcm-lora-sdxl")):
        for n in LORA_FILE_NAMES:
            p = os.path.join(d, n)
            if os.path.exists(p):
                return p
    return None


def synth_lora(cfg, seed=4321, rank=8, device="cpu"):
    """A small synthetic LoRA over attention projections (exercises fold_lora on the sdxl branch)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape, _ in param_table(cfg):
        if name.endswith((".to_q.weight", ".to_k.weight", ".to_v.weight", ".to_out.0.weight")) and len(shape) == 2:
            base = name[: -len(".weight")]
            out[base + ".lora_A.weight"] = torch.randn((rank, shape[1]), generator=g, device=device) / math.sqrt(shape[1])
            out[base + ".lora_B.weight"] = torch.randn((shape[0], rank), generator=g, device=device) * 0.05
    return out


def load_dir(path, variant=None):
    """Read a diffusers-layout component directory: config.json + (diffusion_pytorch_)model[.variant].safetensors.
    Uses safetensors only (nothing in the file is executed)."""
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        config = json.load(f)
    cands = []
    for stem in ("diffusion_pytorch_model", "model"):
        if variant:
            cands.append(f"{stem}.{variant}.safetensors")
        cands.append(f"{stem}.safetensors")
    for c in cands:
        p = os.path.join(path, c)
        if os.path.exists(p):
            return config, load_file(p)
    raise FileNotFoundError(f"no safetensors weights under {path} (tried {cands})")
