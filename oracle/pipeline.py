"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU fp32 restatement of the one call that is the reference's hot path,
``self.pipe(prompt=..., image=..., control_image=..., strength=..., ...)`` at
/root/reference/src/pipeline.py:261-272, i.e. upstream diffusers 0.35.2
``pipelines/controlnet/pipeline_controlnet_sd_xl_img2img.py::__call__`` (call order per SURVEY.md 3.2 steps 1-9),
plus the host steps of ``FastEditor.edit`` around it (:243-258: generator, LANCZOS resize, Canny).

Composite result: **parity unpinned** (no importable upstream here); sub-steps are pinned as listed in nets.py.
"""
import numpy as np
import torch
from PIL import Image

from . import nets
from .canny import canny_rgb
from .lcm import LCMOracle


def pil_to_float(img, normalize):
    """VaeImageProcessor.preprocess on one PIL image: /255 then (optionally) 2x-1; NCHW fp32."""
    a = np.asarray(img).astype(np.float32) / 255.0
    x = torch.from_numpy(a).permute(2, 0, 1)[None].contiguous()
    return 2.0 * x - 1.0 if normalize else x


def float_to_u8(x):
    """VaeImageProcessor.postprocess(..., 'pil'): (x/2+0.5).clamp(0,1) -> NHWC -> (x*255).round() -> uint8."""
    y = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().numpy()
    return (y * 255).round().astype("uint8")


def encode_prompt(sds, cfgs, ids_l, ids_g):
    """encode_prompt(): penultimate hidden state of both encoders concatenated; pooled from encoder 2."""
    hs_l, _ = nets.clip_text_forward(sds["clip_l"], cfgs["clip_l"], ids_l)
    hs_g, pooled = nets.clip_text_forward(sds["clip_g"], cfgs["clip_g"], ids_g)
    return torch.cat([hs_l[-2], hs_g[-2]], dim=-1), pooled


@torch.no_grad()
def run(sds, cfgs, image, control_image, ids, neg_ids, strength=0.8, num_inference_steps=4, guidance_scale=1.5,
        controlnet_conditioning_scale=0.5, generator=None, sched_cfg=None, trace=None):
    """sds/cfgs: dicts with keys unet, controlnet, vae, clip_l, clip_g.  ids/neg_ids: (ids_l, ids_g) int64 [1,77].
    image/control_image: PIL RGB of equal size (multiple of 8).  Returns uint8 HxWx3 array."""
    do_cfg = guidance_scale > 1.0
    pe, pooled = encode_prompt(sds, cfgs, *ids)
    if do_cfg:
        npe, npooled = encode_prompt(sds, cfgs, *neg_ids)
        pe, pooled = torch.cat([npe, pe]), torch.cat([npooled, pooled])
    x_img = pil_to_float(image, True)
    cond = pil_to_float(control_image, False)
    if do_cfg:
        cond = torch.cat([cond, cond])
    h, w = x_img.shape[-2:]

    sch = LCMOracle(**(sched_cfg or {}))
    sch.set_timesteps(num_inference_steps)
    timesteps, _ = sch.get_timesteps(num_inference_steps, strength)

    # prepare_latents: RNG draw #1 (posterior sample), draw #2 (init noise)
    vae_cfg = cfgs["vae"]
    mean, logvar = nets.vae_encode_moments(sds["vae"], vae_cfg, x_img)
    std = torch.exp(0.5 * logvar)
    z0 = (mean + std * torch.randn(mean.shape, generator=generator, dtype=torch.float32)) * vae_cfg["scaling_factor"]
    noise = torch.randn(z0.shape, generator=generator, dtype=torch.float32)
    lat = sch.add_noise(z0, noise, timesteps[0]) if timesteps else z0
    if trace is not None:
        trace.update(prompt_embeds=pe, pooled=pooled, z0=z0, latents0=lat, eps=[], latents=[])

    tid = torch.tensor([[h, w, 0, 0, h, w]], dtype=torch.float32).repeat(pe.shape[0], 1)
    for t in timesteps:
        x_in = torch.cat([lat, lat]) if do_cfg else lat
        down, mid = nets.controlnet_forward(sds["controlnet"], cfgs["controlnet"], x_in, t, pe, cond,
                                            controlnet_conditioning_scale, pooled, tid)
        eps = nets.unet_forward(sds["unet"], cfgs["unet"], x_in, t, pe, pooled, tid, down, mid)
        if do_cfg:
            eu, ec = eps.chunk(2)
            eps = eu + guidance_scale * (ec - eu)
        last = sch.step_index == sch.num_inference_steps - 1
        z = None if last else torch.randn(eps.shape, generator=generator, dtype=torch.float32)
        lat, _ = sch.step(eps, t, lat, z)
        if trace is not None:
            trace["eps"].append(eps)
            trace["latents"].append(lat)
    dec = nets.vae_decode(sds["vae"], vae_cfg, lat / vae_cfg["scaling_factor"])
    if trace is not None:
        trace["decoded"] = dec
    return float_to_u8(dec)[0]


def edit(sds, cfgs, image, ids, neg_ids, size=1024, canny_low=100, canny_high=200, seed=None, **kw):
    """FastEditor.edit (src/pipeline.py:212-274) on the oracle: CPU generator, LANCZOS resize, Canny, run()."""
    gen = torch.Generator(device="cpu").manual_seed(seed) if seed is not None else None
    inp = image.resize((size, size), Image.LANCZOS)
    ctrl = Image.fromarray(canny_rgb(np.array(inp), canny_low, canny_high))
    return run(sds, cfgs, inp, ctrl, ids, neg_ids, generator=gen, **kw)
