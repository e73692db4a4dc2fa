"""Assemble the model stack `FastEditor.__init__` asks for (reference: src/pipeline.py:82-161).

The reference fetches five hub checkpoints by name; offline that is impossible, so the stack comes from
  * ``weights_dir`` (or $FIE_WEIGHTS_DIR): a local directory with diffusers-layout sub-folders
    ``unet/ controlnet/ vae/ text_encoder/ text_encoder_2/`` (+ ``tokenizer*/`` and optional ``lcm_lora.safetensors``), or
  * seeded synthetic weights of the preset architecture, generated directly on the target device.
"""
import os

import torch

from . import presets, weights
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


def _tokenizer(root, sub, pad_id):
    v, m = os.path.join(root, sub, "vocab.json"), os.path.join(root, sub, "merges.txt")
    if os.path.exists(v) and os.path.exists(m):
        return BpeTokenizer(v, m, pad_id)
    return StandInTokenizer(pad_id)


def directory_stack(root, model_name, use_full_controlnet=False, variant="fp16"):
    """Real weights: architecture presets are kept (config-driven presets are labelled [L] in presets.py for SSD-1B /
    small ControlNet; a mismatch with the checkpoint shows up as a missing/mis-shaped key here, loudly)."""
    cfgs = stack_configs(model_name, use_full_controlnet)
    sub = dict(unet="unet", controlnet="controlnet", vae="vae", clip_l="text_encoder", clip_g="text_encoder_2")
    sds = {}
    for k in KEYS:
        _, sd = weights.load_dir(os.path.join(root, sub[k]), variant=variant)
        want = {n: s for n, s, _ in weights.param_table(cfgs[k])}
        missing = [n for n in want if n not in sd]
        bad = [n for n in want if n in sd and tuple(sd[n].shape) != tuple(want[n])]
        if missing or bad:
            raise ValueError(f"{root}/{sub[k]} does not match preset {cfgs[k]['name']}: missing {missing[:3]} "
                             f"(+{max(len(missing) - 3, 0)}), mis-shaped {bad[:3]}")
        sds[k] = {n: sd[n] for n in want}
    lora_path = os.path.join(root, "lcm_lora.safetensors")
    if presets.STACKS[model_name]["lcm_lora"] and os.path.exists(lora_path):
        from safetensors.torch import load_file
        weights.fold_lora(sds["unet"], load_file(lora_path))
    toks = (_tokenizer(root, "tokenizer", cfgs["clip_l"]["pad_token_id"]),
            _tokenizer(root, "tokenizer_2", cfgs["clip_g"]["pad_token_id"]))
    return cfgs, sds, toks
