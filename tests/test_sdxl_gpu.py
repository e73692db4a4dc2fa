"""BASELINE configs 3 and 5 on the workload they NAME: the SDXL-base stack (mid-block attention, depth-10 transformers, LCM-LoRA folded;
reference branch /root/reference/src/pipeline.py:143-154) + ControlNet-full, at BASELINE size, against the CPU oracle (VERDICT r3,
next-round item 2: until round 4 the oracle only ever met the SSD-1B stack at full size, and fp8 was only ever run on SSD-1B).

  * fp16: one ControlNet + UNet evaluation at 128x128 latents / a 1024x1024 edge map, HIP vs oracle/nets.py       (bar 2e-2, as SSD-1B)
  * fp8 (config 5): the same evaluation through the W8A8 pipeline against the ORACLE evaluated on the dequantised weights -- a
    HIP-vs-oracle number for the fp8 configuration, not only HIP-vs-HIP: what remains is the e4m3 rounding of the activations (7.8e-2)
  * fp8 (config 5): whole 1024x1024 edit, fp8 pipeline vs the fp16 HIP pipeline on the same weights and noise     (SSIM >= 0.99)
The batch-8 == 8 serial edits test of config 3 lives in tests/test_realwidth_gpu.py.

Oracle cost: ~10 TFLOP per SDXL evaluation, ~15-25 s on the GPU box's 16 host threads; two evaluations in this module."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def qdq_e4m3(w):
    """Quantise -> dequantise a weight the way fie_pack_rows_f8 / fie_pack_conv3x3_f8 do (include/fie.h): one scale amax / 448 per output
    channel (row of w.view(N, -1)), round to nearest even to e4m3.  The result is a fixed point of that quantiser, so a pipeline built from
    it with weight_dtype="f8e4m3" holds EXACTLY these values wherever it stores fp8 (and the same fp16 values wherever it keeps fp16)."""
    rows = w.float().reshape(w.shape[0], -1)
    s = rows.abs().amax(1, keepdim=True).clamp_min(1e-30) / 448.0
    q = (rows / s).to(torch.float8_e4m3fn).float()
    return (q * s).reshape(w.shape).half()


@pytest.fixture(scope="module")
def sdxl_weights():
    from fie_amd import stack
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    return stack.synthetic_stack("sdxl", True, device="cpu", dtype=torch.float16)


def test_sdxl_eval_fp16_vs_oracle(sdxl_weights, fie):
    """Config 3's model: SDXL-base + ControlNet-full, one evaluation at BASELINE size, fp16 HIP against the fp32 oracle."""
    import time
    from fie_amd.pipe import HipImg2ImgPipeline
    from test_fullsize_gpu import eval_vs_oracle
    cfgs, sds = sdxl_weights
    assert cfgs["unet"]["mid_attn"] and max(max(r) for r in cfgs["unet"]["down_attn"]) == 10, "this module is about the SDXL-base topology"
    pipe = HipImg2ImgPipeline(fie, cfgs, sds, noise_dtype=torch.float32)
    sds32 = {k: {n: v.float() for n, v in sds[k].items()} for k in ("unet", "controlnet")}
    t0 = time.time()
    e, worst = eval_vs_oracle(cfgs, sds32, pipe, fie)
    print(f"SDXL-base eval vs oracle @128x128 latents: eps rel_err={e:.2e}, worst ControlNet residual rel_err={worst:.2e} ({time.time() - t0:.0f} s)")
    assert worst < 2e-2 and e < 2e-2


def test_sdxl_eval_fp8_vs_oracle_on_dequantised_weights(sdxl_weights, fie):
    """Config 5 on SDXL: the W8A8 HIP evaluation against the oracle (fp32 activations) on the SAME, already-dequantised weights.  The error
    left is the fp8 configuration's own: e4m3 rounding (3 mantissa bits: up to 6 % per element) of the activations the projections / resnet
    convs read, at unit scale, through ~70 transformer blocks, on top of the fp16 path's.  Measured (round 4): eps 7.8e-2, worst ControlNet
    residual 7.5e-2 of the tensor's max-abs -- against 8.4e-4 / 1.3e-3 for the fp16 path on the same check; the whole edit still meets the
    fp16 HIP edit at SSIM 0.99916 (next test).  Bar 1.2e-1: it catches a broken scale or a saturating layer, not the rounding itself."""
    from fie_amd.pipe import HipImg2ImgPipeline
    from test_fullsize_gpu import eval_vs_oracle
    cfgs, sds = sdxl_weights
    sdq = dict(sds)
    for k in ("unet", "controlnet"):
        sdq[k] = {n: (qdq_e4m3(v) if v.ndim >= 2 else v) for n, v in sds[k].items()}
    p8 = HipImg2ImgPipeline(fie, cfgs, sdq, noise_dtype=torch.float32, weight_dtype="f8e4m3")
    from fie_amd import hip
    blk = next(iter(p8.unet.transformers())).blocks[0]
    assert isinstance(blk.ff1.wp, hip.W8) and blk.a8, "the fp8 configuration must run fp8 weights and fp8 activations here"
    # the packed fp8 weight holds exactly the dequantised values the oracle gets (the quantiser's fixed point)
    name = "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.2.weight"
    got = blk.ff2.wp.dequant()[: blk.ff2.n, : blk.ff2.k].cpu()
    want = sdq["unet"][name].float()
    assert ((got - want).abs() <= 2e-3 * want.abs()).float().mean() > 0.999       # (a tie that rounds the other way after the fp16 round trip: one e4m3 code on a handful of elements)
    sds32 = {k: {n: v.float() for n, v in sdq[k].items()} for k in ("unet", "controlnet")}
    cache = {}
    e, worst = eval_vs_oracle(cfgs, sds32, p8, fie, cache=cache)
    print(f"SDXL-base fp8 (W8A8) eval vs oracle on the dequantised weights: eps rel_err={e:.2e}, worst ControlNet residual rel_err={worst:.2e}")
    assert worst < 1.2e-1 and e < 1.2e-1
    # the same evaluation with CALIBRATED activation scales (pipe.calibrate_fp8: one edit, per-tensor amax -> power-of-two scale): every scale a
    # power of two, most of them < 1 on these unit-variance activations (the values move up into e4m3's normal binades); the error against the
    # oracle must not grow (it barely moves here: e4m3's relative step is the same in every normal binade and nothing was clipping)
    import math
    from PIL import Image
    from fie_amd import hip
    rng = np.random.default_rng(11)
    a = rng.integers(0, 255, (1024, 1024, 3), dtype=np.uint8)
    a[200:700, 300:800] = 40
    scales = p8.calibrate_fp8(prompt="a [blue] house", image=Image.fromarray(a), control_image=Image.fromarray(hip.canny_rgb(a)), negative_prompt="",
                              strength=0.5, num_inference_steps=4, guidance_scale=1.5, controlnet_conditioning_scale=0.5,
                              generator=torch.Generator("cpu").manual_seed(1))
    flat = [v for sc in scales.values() for v in sc]
    assert len(scales) > 100 and all(v > 0 and math.log2(v) == round(math.log2(v)) for v in flat) and min(flat) < 1.0
    assert blk.s8 == scales["unet.transformer0.block0"] and any(v != 1.0 for v in blk.s8)
    e2, worst2 = eval_vs_oracle(cfgs, sds32, p8, fie, cache=cache)
    print(f"  ... with calibrated activation scales ({len(flat)} tensors, 2^{math.log2(min(flat)):.0f} .. 2^{math.log2(max(flat)):.0f}): eps rel_err={e2:.2e}, "
          f"worst ControlNet residual rel_err={worst2:.2e}")
    assert e2 < max(1.15 * e, 8e-2) and worst2 < max(1.15 * worst, 8e-2)
    p8.load_fp8_scales(scales)                       # round trip of the dict a deployment would store next to the weights
    with pytest.raises(ValueError, match="layer names differ"):
        p8.load_fp8_scales({"unet.nonsense": [1.0]})


def test_sdxl_fp8_full_size_ssim_vs_fp16(sdxl_weights, fie):
    """Config 5 as BASELINE.json words it (SDXL fp8, 4-step LCM, 1024x1024): the fp8 pipeline's edit against the fp16 HIP pipeline's on the
    same weights and noise, SSIM >= 0.99 at the metric resolution of src/metrics.py (512x512); tests/test_fp8_gpu.py has the SSD-1B twin."""
    from PIL import Image
    from fie_amd import hip
    from fie_amd.pipe import HipImg2ImgPipeline
    from oracle import metrics
    cfgs, sds = sdxl_weights
    p16 = HipImg2ImgPipeline(fie, cfgs, sds, noise_dtype=torch.float32)
    p8 = HipImg2ImgPipeline(fie, cfgs, sds, noise_dtype=torch.float32, weight_dtype="f8e4m3")
    rng = np.random.default_rng(7)
    a = np.zeros((1024, 1024, 3), np.uint8)
    a[:] = rng.integers(0, 255, 3)
    for _ in range(12):
        x0, y0 = rng.integers(0, 900, 2)
        a[y0:y0 + rng.integers(30, 300), x0:x0 + rng.integers(30, 300)] = rng.integers(0, 255, 3)
    img = Image.fromarray(a)
    ctrl = Image.fromarray(hip.canny_rgb(a))
    kw = dict(prompt="a [blue] house", negative_prompt="", image=img, control_image=ctrl, strength=0.5, num_inference_steps=4,
              guidance_scale=1.5, controlnet_conditioning_scale=0.5)
    o16 = p16(generator=torch.Generator("cpu").manual_seed(42), **kw).images[0]
    o8 = p8(generator=torch.Generator("cpu").manual_seed(42), **kw).images[0]
    assert p8.last_stats == dict(unet_evals=2, cfg_batch=2, latent_hw=(128, 128), images=1)
    s512 = metrics.ssim(o8, o16)
    sfull = metrics.ssim(o8, np.asarray(o16), size=None)
    d = np.abs(np.asarray(o8).astype(int) - np.asarray(o16).astype(int))
    print(f"SDXL fp8 vs fp16 at BASELINE size: SSIM {s512:.5f} (512x512) / {sfull:.5f} (full), mean |du8| {d.mean():.3f}, max {d.max()}")
    assert np.asarray(o8).std() > 5 and s512 >= 0.99
