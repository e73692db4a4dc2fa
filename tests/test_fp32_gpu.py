"""fp32 path (`use_full_precision=True` / --full_precision / --quality_mode): op parity against torch fp32 (tolerance 2e-5
relative: exact-fp32 MFMA vs torch's CPU summation order) and end-to-end parity against the CPU oracle on the tiny stack
(both fp32, identical weights and noise: the decoded images agree to <= 1 u8 level, SSIM >= 0.9999)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def f32():
    import fie_amd  # noqa: F401
    from fie_amd import hip
    return hip.context(0, torch.float32)


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("m,n,k", [(300, 200, 72), (1024, 1280, 640), (77, 77, 64), (5, 1280, 2816)])
def test_gemm_f32(f32, m, n, k):
    from fie_amd import hip
    a, w, bias, res = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3), rnd(m, n, seed=4)
    out = f32.gemm(a.to(DEV), f32.pack_linear(w.to(DEV)), n, bias=bias.to(DEV), residual=res.to(DEV), scale=0.5, act=hip.ACT_SILU)
    assert out.dtype == torch.float32 and rel_err(out, F.silu(a @ w.T + bias) * 0.5 + res) < 2e-5
    if n % 2 == 0:
        g = f32.gemm(a.to(DEV), f32.pack_linear(w.to(DEV), geglu=True), n,
                     bias=torch.stack([bias[: n // 2], bias[n // 2:]], 1).reshape(-1).to(DEV), act=hip.ACT_GEGLU)
        v, gate = (a @ w.T + bias).chunk(2, -1)
        assert rel_err(g, v * F.gelu(gate)) < 2e-5


@pytest.mark.parametrize("b,h,w,cin,cout,stride,pad_mode,ups", [(1, 16, 16, 64, 32, 1, 0, False), (2, 12, 20, 8, 24, 2, 0, False),
                                                                (1, 16, 16, 32, 16, 2, 1, False), (1, 8, 8, 16, 40, 1, 0, True)])
def test_conv_f32(f32, b, h, w, cin, cout, stride, pad_mode, ups):
    x, wt, bias = rnd(b, cin, h, w, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(cout, seed=3)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    if pad_mode == 1:
        xi = F.pad(xi, (0, 1, 0, 1))
    ref = F.conv2d(xi, wt, bias, stride=stride, padding=1 if pad_mode == 0 else 0)
    out = f32.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(DEV), f32.pack_conv3x3(wt.to(DEV)), cout, stride=stride, pad_mode=pad_mode,
                      upsample=ups, bias=bias.to(DEV))
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize("b,hn,tq,tk,d,causal", [(2, 3, 200, 333, 64, False), (1, 2, 77, 77, 64, True), (1, 1, 256, 256, 512, False)])
def test_attention_f32(f32, b, hn, tq, tk, d, causal):
    c = hn * d
    qkv = [rnd(b * t, c, seed=i) for i, t in enumerate((tq, tk, tk))]
    sp = lambda x, t: x.view(b, t, hn, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(qkv[0], tq), sp(qkv[1], tk), sp(qkv[2], tk), is_causal=causal).transpose(1, 2).reshape(b * tq, c)
    out = f32.attention(*(t.to(DEV) for t in qkv), hn, d, tq, tk, b, causal=causal)
    assert rel_err(out, ref) < 2e-5


def test_norms_f32(f32):
    x1, x2 = rnd(2, 256, 64, seed=1) + 0.5, rnd(2, 256, 32, seed=2) * 2
    g, bt = rnd(96, seed=3), rnd(96, seed=4)
    ref = F.silu(F.group_norm(torch.cat([x1, x2], -1).transpose(1, 2), 32, g, bt, 1e-5).transpose(1, 2))
    assert rel_err(f32.groupnorm(x1.to(DEV), g.to(DEV), bt.to(DEV), 32, 1e-5, True, x2=x2.to(DEV)), ref) < 2e-5
    x1, g, bt = rnd(2, 1024, 1280, seed=8) + 0.5, rnd(1280, seed=9), rnd(1280, seed=10)       # single-pass kernel shape
    ref = F.silu(F.group_norm(x1.transpose(1, 2), 32, g, bt, 1e-5).transpose(1, 2))
    assert rel_err(f32.groupnorm(x1.to(DEV), g.to(DEV), bt.to(DEV), 32, 1e-5, True), ref) < 2e-5
    x, g, bt = rnd(77, 768, seed=5) * 3 + 1, rnd(768, seed=6), rnd(768, seed=7)
    assert rel_err(f32.layernorm(x.to(DEV), g.to(DEV), bt.to(DEV)), F.layer_norm(x, (768,), g, bt, 1e-5)) < 2e-5


@pytest.mark.parametrize("stack_name", ["tiny", "tiny-nomid"])
def test_full_pipeline_fp32_vs_oracle(f32, stack_name):
    from fie_amd import stack
    from fie_amd.pipe import HipImg2ImgPipeline
    from oracle import canny, metrics, pipeline as opipe
    from test_pipeline_gpu import synth_image
    cfgs, sds = stack.synthetic_stack(stack_name, True, device="cpu", dtype=torch.float32)
    pipe = HipImg2ImgPipeline(f32, cfgs, sds)
    assert pipe.noise_dtype == torch.float32
    img = synth_image(5, 128)
    ctrl = Image.fromarray(canny.canny_rgb(np.asarray(img)))
    prompt = "a [purple] triangle"
    ids = lambda t: (pipe.tok_l([t]), pipe.tok_g([t]))
    for use_graph in (False, True):
        pipe.use_graph = use_graph
        out = pipe(prompt=prompt, negative_prompt="", image=img, control_image=ctrl, strength=0.8, num_inference_steps=4,
                   guidance_scale=1.5, controlnet_conditioning_scale=0.5, generator=torch.Generator("cpu").manual_seed(42)).images[0]
        ref = opipe.run(sds, cfgs, img, ctrl, ids(prompt), ids(""), strength=0.8, num_inference_steps=4, guidance_scale=1.5,
                        controlnet_conditioning_scale=0.5, generator=torch.Generator("cpu").manual_seed(42))
        diff = np.abs(np.asarray(out).astype(int) - ref.astype(int))
        assert diff.max() <= 1 and metrics.ssim(out, ref, size=None) >= 0.9999
        assert (diff > 0).mean() < 0.01                 # fp32 vs fp32: only rounding-boundary pixels may differ


def test_editor_full_precision_flag(f32):
    from src.pipeline import FastEditor
    from test_pipeline_gpu import synth_image
    ed = FastEditor(model_name="tiny", enable_cpu_offload=False, use_full_precision=True, use_full_controlnet=True)
    assert ed.dtype == torch.float32
    a = ed.edit(synth_image(2, 96), "a [cat]", seed=42, strength=0.5, guidance_scale=1.0)
    ed16 = FastEditor(model_name="tiny", enable_cpu_offload=False, use_full_controlnet=True, noise_dtype=torch.float32)
    b = ed16.edit(synth_image(2, 96), "a [cat]", seed=42, strength=0.5, guidance_scale=1.0)
    from oracle import metrics
    assert a.size == (1024, 1024) and metrics.ssim(a, b) >= 0.99        # the fp16 path against the fp32 path on the GPU
