"""Architecture presets for the graphs `FastEditor.__init__` selects
(reference: src/pipeline.py:30-43 MODEL_CONFIGS, :82-105 ControlNet/VAE choice, :110-154 UNet choice).

The reference never states these shapes itself -- it names hub checkpoints -- so the presets restate
the published architecture of those checkpoints (confidence tags kept there).
Everything is plain data so that a real ``config.json`` can replace a preset (see weights.load_dir).

Conventions
-----------
``down_attn[i][j]``  transformer depth after resnet j of down block i (0 = no attention)
``up_attn[i][j]``    same for up block i (up blocks have layers_per_block+1 resnets)
``mid_attn``         transformer depth of the mid block (0 = attention removed)
``mid_resnets``      number of resnets in the mid block (diffusers: 1 + num_layers)
Head dim is always 64 (diffusers' misleadingly named ``attention_head_dim`` = heads 5/10/20).
"""
from copy import deepcopy

UNET_SDXL = dict(
    kind="unet", name="sdxl-base",
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_attn=((0, 0), (2, 2), (10, 10)),
    mid_attn=10, mid_resnets=2,
    up_attn=((10, 10, 10), (2, 2, 2), (0, 0, 0)),
    head_dim=64, cross_attention_dim=2048, norm_num_groups=32, norm_eps=1e-5,
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
)

# SSD-1B / lcm-ssd-1b.  [L] confidence: A' = one mid resnet, no mid attention (1.331 B params,
# closest to the advertised 1.3 B); A = two mid resnets; B = mid block keeps depth-4 attention.
UNET_SSD1B_A1 = dict(
    UNET_SDXL, name="ssd-1b-A1",
    down_attn=((0, 0), (2, 2), (4, 4)),
    mid_attn=0, mid_resnets=1,
    up_attn=((4, 4, 10), (2, 1, 1), (0, 0, 0)),
)
UNET_SSD1B_A = dict(UNET_SSD1B_A1, name="ssd-1b-A", mid_resnets=2)
UNET_SSD1B_B = dict(UNET_SSD1B_A1, name="ssd-1b-B", mid_resnets=2, mid_attn=4)

# diffusers/controlnet-canny-sdxl-1.0 (full) [H: 1.251 B params reproduced]
CONTROLNET_FULL = dict(
    kind="controlnet", name="controlnet-canny-sdxl-full",
    in_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_attn=((0, 0), (2, 2), (10, 10)),
    mid_attn=10, mid_resnets=2,
    head_dim=64, cross_attention_dim=2048, norm_num_groups=32, norm_eps=1e-5,
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
    conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256),
)
# diffusers/controlnet-canny-sdxl-1.0-small [L]: depth-1 everywhere is a labelled guess (293 M vs advertised 320 M)
CONTROLNET_SMALL = dict(
    CONTROLNET_FULL, name="controlnet-canny-sdxl-small-depth1-guess",
    down_attn=((0, 0), (1, 1), (1, 1)), mid_attn=1,
)

# stabilityai/sdxl-vae and madebyollin/sdxl-vae-fp16-fix share this graph [H: 83.7 M params]
VAE_SDXL = dict(
    kind="vae", name="sdxl-vae",
    in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
    layers_per_block=2, norm_num_groups=32, norm_eps=1e-6, scaling_factor=0.13025,
)

CLIP_L = dict(
    kind="clip", name="clip-vit-l-14-text",
    vocab_size=49408, max_positions=77, hidden=768, layers=12, heads=12, intermediate=3072,
    act="quick_gelu", projection_dim=0, eps=1e-5, pad_token_id=49407, eos_token_id=49407, bos_token_id=49406,
)
CLIP_BIGG = dict(
    kind="clip", name="openclip-vit-bigg-14-text",
    vocab_size=49408, max_positions=77, hidden=1280, layers=32, heads=20, intermediate=5120,
    act="gelu", projection_dim=1280, eps=1e-5, pad_token_id=0, eos_token_id=49407, bos_token_id=49406,
)

# LCMScheduler config inherited from the base model's scheduler
LCM_SCHED = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5, set_alpha_to_one=False,
)


def _tiny(unet_like):
    """Shrink a UNet/ControlNet preset to something the CPU oracle runs in well under a second.
    Keeps every code path: attention + plain blocks, asymmetric depths, shortcut convs, skip concats."""
    c = deepcopy(unet_like)
    c["name"] = "tiny-" + c["name"]
    c["block_out_channels"] = (64, 128, 256)
    c["cross_attention_dim"] = 192           # tiny CLIP-L hidden 128 + tiny CLIP-G hidden 64
    c["addition_time_embed_dim"] = 32
    # pooled text dim 64 (tiny bigG projection) + 6 * 32
    c["projection_class_embeddings_input_dim"] = 64 + 6 * 32
    c["down_attn"] = ((0, 0), (1, 1), (2, 2)) if c["mid_attn"] else ((0, 0), (1, 1), (2, 1))
    if c["kind"] == "unet":
        c["up_attn"] = ((2, 1, 2), (1, 1, 1), (0, 0, 0))
    c["mid_attn"] = 2 if c["mid_attn"] else 0       # = first depth of the last down block: the only mid depth a diffusers config.json can express
    if c["kind"] == "controlnet":
        c["conditioning_embedding_out_channels"] = (16, 32, 32, 64)
    return c


TINY_UNET = _tiny(UNET_SDXL)
TINY_UNET_NOMID = _tiny(UNET_SSD1B_A1)
TINY_CONTROLNET = _tiny(CONTROLNET_FULL)
TINY_VAE = dict(VAE_SDXL, name="tiny-vae", block_out_channels=(32, 64, 64, 64))
TINY_CLIP_L = dict(CLIP_L, name="tiny-clip-l", hidden=128, layers=2, heads=2, intermediate=256, vocab_size=49408)
TINY_CLIP_G = dict(CLIP_BIGG, name="tiny-clip-g", hidden=64, layers=3, heads=1, intermediate=128, projection_dim=64)

# what `FastEditor(model_name=...)` resolves to; "tiny" is a test-only stack with the same topology
STACKS = {
    "sdxl": dict(unet=UNET_SDXL, controlnet_small=CONTROLNET_SMALL, controlnet_full=CONTROLNET_FULL,
                 vae=VAE_SDXL, clip_l=CLIP_L, clip_g=CLIP_BIGG, lcm_lora=True),
    "ssd-1b": dict(unet=UNET_SSD1B_A1, controlnet_small=CONTROLNET_SMALL, controlnet_full=CONTROLNET_FULL,
                   vae=VAE_SDXL, clip_l=CLIP_L, clip_g=CLIP_BIGG, lcm_lora=False),
    "tiny": dict(unet=TINY_UNET, controlnet_small=TINY_CONTROLNET, controlnet_full=TINY_CONTROLNET,
                 vae=TINY_VAE, clip_l=TINY_CLIP_L, clip_g=TINY_CLIP_G, lcm_lora=False),
    "tiny-nomid": dict(unet=TINY_UNET_NOMID, controlnet_small=TINY_CONTROLNET, controlnet_full=TINY_CONTROLNET,
                       vae=TINY_VAE, clip_l=TINY_CLIP_L, clip_g=TINY_CLIP_G, lcm_lora=True),
}


def time_embed_dim(cfg):
    return cfg["block_out_channels"][0] * 4
