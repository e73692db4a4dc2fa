"""diffusers / transformers `config.json` -> the plain-data graph configs of presets.py.

`FastEditor.__init__` of the reference names hub checkpoints (/root/reference/src/pipeline.py:82-154) whose `config.json`
decides the topology upstream; SURVEY A.2 / A.3 could not verify the SSD-1B UNet and the "small" ControlNet offline, so a
local weights directory must be read CONFIG-FIRST: the graph is built from the directory's own config.json, never from a
preset guess.  Restates how the upstream constructors expand their arguments:

  * UNet2DConditionModel.__init__ / get_down_block / get_mid_block / get_up_block (models/unets/unet_2d_condition.py,
    unet_2d_blocks.py): `transformer_layers_per_block` int | list | nested list, `reverse_transformer_layers_per_block`,
    block types with / without cross attention, `mid_block_type` (UNetMidBlock2DCrossAttn: resnet, attention, resnet;
    UNetMidBlock2D as built by get_mid_block: num_layers = 0, add_attention = False -> ONE resnet), `attention_head_dim`
    holding the HEAD COUNTS (num_attention_heads = num_attention_heads or attention_head_dim), `time_cond_proj_dim` ignored
    because the SDXL-ControlNet-img2img pipeline never passes `timestep_cond`.
  * ControlNetModel.__init__ (models/controlnets/controlnet.py): the encoder half + `conditioning_embedding_out_channels`.
  * AutoencoderKL (models/autoencoders/autoencoder_kl.py); CLIPTextConfig (transformers models/clip/configuration_clip.py).

Anything this build has no kernels for (head dim != 64, non-linear projections, other block types) raises ValueError."""


def _per_block(v, n, what):
    if isinstance(v, (list, tuple)):
        if len(v) != n:
            raise ValueError(f"{what}: expected {n} entries, config has {len(v)}")
        return list(v)
    return [v] * n


def _depths(entry, layers, what):
    """One block's transformer depth spec (int | list per layer) -> tuple of `layers` ints."""
    if isinstance(entry, (list, tuple)):
        if len(entry) != layers:
            raise ValueError(f"{what}: {len(entry)} depths for {layers} layers")
        return tuple(int(x) for x in entry)
    return (int(entry),) * layers


def _common(c, kind):
    chans = tuple(c["block_out_channels"])
    n = len(chans)
    lpb = c.get("layers_per_block", 2)
    if isinstance(lpb, (list, tuple)):
        if len(set(lpb)) != 1:
            raise ValueError(f"{kind}: per-block layers_per_block {lpb} is not supported")
        lpb = lpb[0]
    down_types = c.get("down_block_types") or ["CrossAttnDownBlock2D"] * n
    for t in down_types:
        if t not in ("DownBlock2D", "CrossAttnDownBlock2D"):
            raise ValueError(f"{kind}: down block type {t!r} is not supported")
    tl = _per_block(c.get("transformer_layers_per_block", 1), n, f"{kind}.transformer_layers_per_block")
    down_attn = tuple(_depths(tl[i], lpb, f"{kind}.down[{i}]") if down_types[i] == "CrossAttnDownBlock2D" else (0,) * lpb
                      for i in range(n))
    heads = c.get("num_attention_heads") or c.get("attention_head_dim", 8)
    heads = _per_block(heads, n, f"{kind}.attention_head_dim")
    hd = {chans[i] // int(heads[i]) for i in range(n) if any(down_attn[i])}
    if any(down_attn[i] and chans[i] % int(heads[i]) for i in range(n)) or (hd and hd != {64}):
        raise ValueError(f"{kind}: attention head dims {sorted(hd)} -- the HIP attention kernels are built for 64")
    if any(any(d) for d in down_attn) and not c.get("use_linear_projection", False):
        raise ValueError(f"{kind}: use_linear_projection=False (conv proj_in/out) is not supported")
    if c.get("addition_embed_type") != "text_time":
        raise ValueError(f"{kind}: addition_embed_type {c.get('addition_embed_type')!r} (SDXL needs 'text_time')")
    mid_type = c.get("mid_block_type", "UNetMidBlock2DCrossAttn")
    if mid_type == "UNetMidBlock2DCrossAttn":
        last = tl[-1]
        mid_attn, mid_resnets = int(last[0] if isinstance(last, (list, tuple)) else last), 2
    elif mid_type == "UNetMidBlock2D":              # get_mid_block(): num_layers=0, add_attention=False
        mid_attn, mid_resnets = 0, 1
    else:
        raise ValueError(f"{kind}: mid_block_type {mid_type!r} is not supported")
    return dict(
        kind=kind, in_channels=c.get("in_channels", 4), block_out_channels=chans, layers_per_block=int(lpb),
        down_attn=down_attn, mid_attn=mid_attn, mid_resnets=mid_resnets, head_dim=64,
        cross_attention_dim=int(c["cross_attention_dim"] if not isinstance(c["cross_attention_dim"], (list, tuple))
                                else c["cross_attention_dim"][0]),
        norm_num_groups=c.get("norm_num_groups", 32), norm_eps=c.get("norm_eps", 1e-5),
        addition_time_embed_dim=c["addition_time_embed_dim"],
        projection_class_embeddings_input_dim=c["projection_class_embeddings_input_dim"]), tl, down_types


def unet_cfg(c, name="unet(config.json)"):
    cfg, tl, down_types = _common(c, "unet")
    n, lpb = len(cfg["block_out_channels"]), cfg["layers_per_block"]
    up_types = c.get("up_block_types") or ["CrossAttnUpBlock2D"] * n
    for t in up_types:
        if t not in ("UpBlock2D", "CrossAttnUpBlock2D"):
            raise ValueError(f"unet: up block type {t!r} is not supported")
    nested = any(isinstance(x, (list, tuple)) for x in tl)
    rev = c.get("reverse_transformer_layers_per_block")
    if nested and rev is None:
        raise ValueError("Must provide 'reverse_transformer_layers_per_block` if using asymmetric UNet.")   # upstream's message
    rev = list(reversed(tl)) if rev is None else _per_block(rev, n, "unet.reverse_transformer_layers_per_block")
    cfg["up_attn"] = tuple(_depths(rev[i], lpb + 1, f"unet.up[{i}]") if up_types[i] == "CrossAttnUpBlock2D" else (0,) * (lpb + 1)
                           for i in range(n))
    cfg["out_channels"] = c.get("out_channels", 4)
    cfg["name"] = name
    return cfg


def controlnet_cfg(c, name="controlnet(config.json)"):
    cfg, _, _ = _common(c, "controlnet")
    cfg["conditioning_channels"] = c.get("conditioning_channels", 3)
    cfg["conditioning_embedding_out_channels"] = tuple(c.get("conditioning_embedding_out_channels", (16, 32, 96, 256)))
    cfg["name"] = name
    return cfg


def vae_cfg(c, name="vae(config.json)"):
    for t in (c.get("down_block_types") or []):
        if t != "DownEncoderBlock2D":
            raise ValueError(f"vae: block type {t!r} is not supported")
    return dict(kind="vae", name=name, in_channels=c.get("in_channels", 3), out_channels=c.get("out_channels", 3),
                latent_channels=c.get("latent_channels", 4), block_out_channels=tuple(c["block_out_channels"]),
                layers_per_block=c.get("layers_per_block", 2), norm_num_groups=c.get("norm_num_groups", 32), norm_eps=1e-6,
                scaling_factor=c.get("scaling_factor", 0.18215), force_upcast=bool(c.get("force_upcast", True)))


def clip_cfg(c, name="clip(config.json)", pad_token_id=None):
    """`c` = a CLIPTextConfig dict.  The projection exists iff the checkpoint's architecture is CLIPTextModelWithProjection
    (text_encoder_2); text_encoder's config also carries a projection_dim but the class has no projection."""
    with_proj = "CLIPTextModelWithProjection" in (c.get("architectures") or [])
    hidden, heads = c["hidden_size"], c["num_attention_heads"]
    if hidden // heads != 64 or hidden % heads:
        raise ValueError(f"clip: head dim {hidden / heads} -- the HIP attention kernels are built for 64")
    act = c.get("hidden_act", "quick_gelu")
    if act not in ("quick_gelu", "gelu"):
        raise ValueError(f"clip: hidden_act {act!r} is not supported")
    eos = c.get("eos_token_id", 2)
    return dict(kind="clip", name=name, vocab_size=c["vocab_size"], max_positions=c.get("max_position_embeddings", 77),
                hidden=hidden, layers=c["num_hidden_layers"], heads=heads, intermediate=c["intermediate_size"], act=act,
                projection_dim=c.get("projection_dim", hidden) if with_proj else 0, eps=c.get("layer_norm_eps", 1e-5),
                # legacy configs (eos_token_id == 2, as shipped with SDXL's text encoders) pool at argmax(ids): modeling_clip.py
                eos_token_id=eos, pad_token_id=c.get("pad_token_id", 1) if pad_token_id is None else pad_token_id,
                bos_token_id=c.get("bos_token_id", 0))


BUILDERS = {"unet": unet_cfg, "controlnet": controlnet_cfg, "vae": vae_cfg, "clip_l": clip_cfg, "clip_g": clip_cfg}


def to_diffusers(cfg):
    """Inverse of the builders above: a graph config -> the config.json dict a diffusers-layout directory would hold
    (used to export synthetic stacks as directories: tools/export_synthetic_dir.py, tests)."""
    k = cfg["kind"]
    if k == "vae":
        n = len(cfg["block_out_channels"])
        return dict(_class_name="AutoencoderKL", in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                    latent_channels=cfg["latent_channels"], block_out_channels=list(cfg["block_out_channels"]),
                    layers_per_block=cfg["layers_per_block"], norm_num_groups=cfg["norm_num_groups"],
                    scaling_factor=cfg["scaling_factor"], down_block_types=["DownEncoderBlock2D"] * n,
                    up_block_types=["UpDecoderBlock2D"] * n, force_upcast=False)
    if k == "clip":
        return dict(architectures=["CLIPTextModelWithProjection" if cfg["projection_dim"] else "CLIPTextModel"],
                    vocab_size=cfg["vocab_size"], max_position_embeddings=cfg["max_positions"], hidden_size=cfg["hidden"],
                    num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], intermediate_size=cfg["intermediate"],
                    hidden_act=cfg["act"], projection_dim=cfg["projection_dim"] or cfg["hidden"], layer_norm_eps=cfg["eps"],
                    eos_token_id=cfg["eos_token_id"], pad_token_id=cfg["pad_token_id"], bos_token_id=cfg["bos_token_id"])
    chans = list(cfg["block_out_channels"])
    n = len(chans)
    mid = "UNetMidBlock2DCrossAttn" if cfg["mid_resnets"] == 2 else "UNetMidBlock2D"
    tl = [list(d) if len(set(d)) > 1 else int(d[0]) for d in cfg["down_attn"]]
    if mid == "UNetMidBlock2DCrossAttn" and cfg["mid_attn"] != (tl[-1][0] if isinstance(tl[-1], list) else tl[-1]):
        raise ValueError("mid-block depth must equal the last down block's first depth to be expressible in config.json")
    out = dict(
        _class_name="UNet2DConditionModel" if k == "unet" else "ControlNetModel", in_channels=cfg["in_channels"],
        block_out_channels=chans, layers_per_block=cfg["layers_per_block"],
        down_block_types=["CrossAttnDownBlock2D" if any(d) else "DownBlock2D" for d in cfg["down_attn"]],
        mid_block_type=mid, transformer_layers_per_block=[t if (isinstance(t, list) or t) else 1 for t in tl],
        attention_head_dim=[c // cfg["head_dim"] for c in chans], cross_attention_dim=cfg["cross_attention_dim"],
        use_linear_projection=True, norm_num_groups=cfg["norm_num_groups"], norm_eps=cfg["norm_eps"],
        addition_embed_type="text_time", addition_time_embed_dim=cfg["addition_time_embed_dim"],
        projection_class_embeddings_input_dim=cfg["projection_class_embeddings_input_dim"])
    if k == "unet":
        out["out_channels"] = cfg["out_channels"]
        out["up_block_types"] = ["CrossAttnUpBlock2D" if any(d) else "UpBlock2D" for d in cfg["up_attn"]]
        rev = [list(d) if len(set(d)) > 1 else int(d[0]) for d in cfg["up_attn"]]
        rev = [t if (isinstance(t, list) or t) else 1 for t in rev]
        if rev != list(reversed(out["transformer_layers_per_block"])) or any(isinstance(t, list) for t in out["transformer_layers_per_block"]):
            out["reverse_transformer_layers_per_block"] = rev
        out["time_cond_proj_dim"] = None
    else:
        out["conditioning_channels"] = cfg["conditioning_channels"]
        out["conditioning_embedding_out_channels"] = list(cfg["conditioning_embedding_out_channels"])
    return out
