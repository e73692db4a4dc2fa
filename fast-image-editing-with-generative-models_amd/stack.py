"""Assemble the model stack `FastEditor.__init__` asks for (reference: src/pipeline.py:82-161).

The reference fetches five hub checkpoints by name; offline that is impossible, so the stack comes from
  * ``weights_dir`` (or $FIE_WEIGHTS_DIR): a local directory with diffusers-layout sub-folders
    ``unet/ controlnet/ vae/ text_encoder/ text_encoder_2/`` (+ ``tokenizer*/`` and, for sdxl, the LCM-LoRA
    ``pytorch_lora_weights.safetensors``); every graph is built from the sub-folder's own config.json, or
  * seeded synthetic weights of the preset architecture, generated directly on the target device.
"""
import json
import os

import torch

from . import config, presets, weights
from .tokenizer import BpeTokenizer, StandInTokenizer

KEYS = ("unet", "controlnet", "vae", "clip_l", "clip_g")


def stack_configs(model_name, use_full_controlnet):
    st = presets.STACKS[model_name]
    return dict(unet=st["unet"], controlnet=st["controlnet_full" if use_full_controlnet else "controlnet_small"],
                vae=st["vae"], clip_l=st["clip_l"], clip_g=st["clip_g"])


def synthetic_stack(model_name, use_full_controlnet=False, device="cpu", dtype=torch.float16, seed=1234, lora=None):
    cfgs = stack_configs(model_name, use_full_controlnet)
    sds = {k: weights.synth_state_dict(cfgs[k], seed=seed + i, device=device, dtype=dtype) for i, k in enumerate(KEYS)}
    if lora is None:
        lora = presets.STACKS[model_name]["lcm_lora"]
    if lora:        # sdxl branch: LCM-LoRA folded once at load (reference attaches it unfused, src/pipeline.py:154)
        weights.fold_lora(sds["unet"], weights.synth_lora(cfgs["unet"], seed=seed + 99, device=device))
    return cfgs, sds


def broadcast_stack(model_name, use_full_controlnet=False, device="cpu", dtype=torch.float16, seed=1234):
    """Multi-GPU start-up (collective C1 of SURVEY 2.3): rank 0 materialises the fp16 weights, every other rank allocates
    receive buffers, then ONE bucketed broadcast per ~1 GiB moves them over RCCL/xGMI (gloo in the CPU tests)."""
    import torch.distributed as dist
    from . import dist as fdist
    cfgs = stack_configs(model_name, use_full_controlnet)
    if dist.get_rank() == 0:
        _, sds = synthetic_stack(model_name, use_full_controlnet, device=device, dtype=dtype, seed=seed)
    else:
        sds = {k: weights.empty_state_dict(cfgs[k], device=device, dtype=dtype) for k in KEYS}
    fdist.broadcast_state_dicts(sds, src=0, device=device)
    return cfgs, sds


def _tokenizer(root, sub, default_pad):
    """tokenizer*/vocab.json + merges.txt -> BpeTokenizer with the pad id the tokenizer's own config names (CLIPTokenizer
    pads encoder 1 with <|endoftext|> and encoder 2 with "!" = id 0); stand-in tokenizer when the files are absent."""
    d = os.path.join(root, sub)
    v, m = os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt")
    if not (os.path.exists(v) and os.path.exists(m)):
        return StandInTokenizer(default_pad)
    pad = default_pad
    for cfg_name in ("special_tokens_map.json", "tokenizer_config.json"):
        p = os.path.join(d, cfg_name)
        if os.path.exists(p):
            with open(p, encoding="utf-8") as f:
                tok = json.load(f).get("pad_token")
            if isinstance(tok, dict):
                tok = tok.get("content")
            if isinstance(tok, str):
                with open(v, encoding="utf-8") as f:
                    vocab = json.load(f)
                key = tok if tok in vocab else tok + "</w>"
                if key in vocab:
                    pad = vocab[key]
                break
    return BpeTokenizer(v, m, pad)


SUBDIRS = dict(unet="unet", controlnet="controlnet", vae="vae", clip_l="text_encoder", clip_g="text_encoder_2")


def directory_stack(root, model_name, use_full_controlnet=False, variant="fp16", lora_path=None):
    """Real weights from a local diffusers-layout directory, CONFIG-FIRST: every graph is built
    from the sub-folder's own config.json (fie_amd/config.py), never from a preset -- the SSD-1B UNet and the "small"
    ControlNet presets are labelled guesses, and `use_full_controlnet=False` is the reference's default.  The state dict is
    then checked against the parameter table of THAT graph (missing / mis-shaped keys raise).  A config.json without
    architecture fields (hand-made minimal directories) falls back to the preset and says so in the config name.

    The `sdxl` branch attaches the LCM-LoRA (/root/reference/src/pipeline.py:154, mandatory there): the file
    (`pytorch_lora_weights.safetensors`, PEFT or kohya keys) must exist, else this raises -- base SDXL run for 4 LCM steps
    without it would silently produce garbage."""
    preset = stack_configs(model_name, use_full_controlnet)
    cfgs, sds = {}, {}
    for k in KEYS:
        d = os.path.join(root, SUBDIRS[k])
        cj, sd = weights.load_dir(d, variant=variant)
        if any(f in cj for f in ("block_out_channels", "hidden_size")):
            cfgs[k] = config.BUILDERS[k](cj, name=f"{SUBDIRS[k]}/config.json")
            if k in ("clip_l", "clip_g"):
                cfgs[k]["pad_token_id"] = preset[k]["pad_token_id"]      # the TOKENIZER's pad id (model configs carry legacy ids)
        else:
            cfgs[k] = dict(preset[k], name=preset[k]["name"] + " (preset: config.json has no architecture fields)")
        if k in ("clip_l", "clip_g"):                                    # transformers >= 5 drops the `text_model.` prefix
            sd = {(n if n.startswith(("text_model.", "text_projection.")) else "text_model." + n): v for n, v in sd.items()}
        want = {n: s for n, s, _ in weights.param_table(cfgs[k])}
        missing = [n for n in want if n not in sd]
        bad = [n for n in want if n in sd and tuple(sd[n].shape) != tuple(want[n])]
        if missing or bad:
            raise ValueError(f"{d} does not match its graph {cfgs[k]['name']}: missing {missing[:3]} "
                             f"(+{max(len(missing) - 3, 0)}), mis-shaped {bad[:3]}")
        sds[k] = {n: sd[n] for n in want}
    if presets.STACKS[model_name]["lcm_lora"]:
        lora_path = lora_path or os.environ.get("FIE_LORA_PATH") or weights.find_lora(root)
        if not lora_path or not os.path.exists(lora_path):
            raise FileNotFoundError(f"model {model_name!r} needs the LCM-LoRA (reference: load_lora_weights, src/pipeline.py:154) but none of "
                                    f"{weights.LORA_FILE_NAMES} exists under {root} (or pass lora_path / $FIE_LORA_PATH)")
        from safetensors.torch import load_file
        n = weights.fold_lora(sds["unet"], load_file(lora_path))
        if n == 0:
            raise ValueError(f"{lora_path}: no UNet adapter pairs found")
    toks = (_tokenizer(root, "tokenizer", cfgs["clip_l"]["pad_token_id"]),
            _tokenizer(root, "tokenizer_2", cfgs["clip_g"]["pad_token_id"]))
    return cfgs, sds, toks
