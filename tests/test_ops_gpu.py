"""Op-level parity of every HIP kernel (through the C ABI) against a plain torch fp32 reference of the same op.
Tolerances: fp16 storage + fp32 accumulation -> rel. error of the output tensor <= 3e-3 of its max-abs
(stated per test); integer/byte outputs are exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (1024, 1280, 1280), (77, 640, 2048), (3, 1280, 320), (4096, 640, 2560),
                                   (130, 200, 72), (16384, 320, 320)])
def test_gemm_plain(fie, m, n, k):
    a, w = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5)
    bias = rnd(n, seed=3)
    ref = a.float() @ w.float().T + bias.float()
    wp = fie.pack_linear(w.to(DEV))
    out = fie.gemm(a.to(DEV), wp, n, bias=bias.to(DEV))
    assert rel_err(out, ref) < 3e-3


def test_gemm_asymmetric_identity(fie):
    # A = I with an asymmetric W catches a swapped row/col map (guide: "Always A=I-check with ASYMMETRIC B")
    n = k = 128
    a = torch.eye(k).half()
    w = (torch.arange(n)[:, None] * 0.5 + torch.arange(k)[None, :] * 0.01).half()
    out = fie.gemm(a.to(DEV), fie.pack_linear(w.to(DEV)), n)
    assert torch.equal(out.cpu(), w.T.contiguous())


@pytest.mark.parametrize("act", ["silu", "gelu", "quick_gelu", "geglu"])
def test_gemm_epilogues(fie, act):
    from fie_amd import hip
    m, n, k = 300, 512, 192
    a, w, bias, res = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3), rnd(m, n, seed=4)
    rowb = rnd(3, n, seed=5)
    lin = a.float() @ w.float().T + bias.float()
    if act == "geglu":
        v, g = lin.chunk(2, dim=-1)
        ref = v * F.gelu(g) * 0.5
        out = fie.gemm(a.to(DEV), fie.pack_linear(w.to(DEV), geglu=True), n,
                       bias=torch.stack([bias[: n // 2], bias[n // 2:]], 1).reshape(-1).contiguous().to(DEV),
                       scale=0.5, act=hip.ACT_GEGLU)
    else:
        lin = lin + rowb.float().repeat_interleave(100, 0)
        fn = {"silu": F.silu, "gelu": F.gelu, "quick_gelu": lambda x: x * torch.sigmoid(1.702 * x)}[act]
        ref = fn(lin) * 0.5 + res.float()
        code = {"silu": hip.ACT_SILU, "gelu": hip.ACT_GELU, "quick_gelu": hip.ACT_QUICK_GELU}[act]
        out = fie.gemm(a.to(DEV), fie.pack_linear(w.to(DEV)), n, bias=bias.to(DEV), rowbias=rowb.to(DEV),
                       rows_per_batch=100, residual=res.to(DEV), scale=0.5, act=code)
    assert rel_err(out, ref) < 3e-3


def test_gemm_concat_a(fie):
    m, k1, k2, n = 500, 128, 64, 256
    a1, a2, w = rnd(m, k1, seed=1), rnd(m, k2, seed=2), rnd(n, k1 + k2, seed=3, scale=0.07)
    ref = torch.cat([a1, a2], 1).float() @ w.float().T
    out = fie.gemm(a1.to(DEV), fie.pack_linear(w.to(DEV)), n, a2=a2.to(DEV))
    assert rel_err(out, ref) < 3e-3


@pytest.mark.parametrize("b,h,w,cin,cout,stride,pad_mode,ups", [
    (1, 32, 32, 64, 64, 1, 0, False), (2, 16, 16, 320, 640, 1, 0, False), (1, 32, 32, 128, 128, 2, 0, False),
    (1, 32, 32, 64, 64, 2, 1, False), (1, 16, 16, 64, 128, 1, 0, True), (1, 64, 64, 8, 32, 1, 0, False),
    (1, 24, 40, 16, 16, 1, 0, False), (1, 32, 32, 96, 256, 2, 0, False), (1, 128, 128, 320, 4, 1, 0, False)])
def test_conv3x3(fie, b, h, w, cin, cout, stride, pad_mode, ups):
    x = rnd(b, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=3)
    xi = x.float()
    if ups:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    if pad_mode == 1:
        xi = F.pad(xi, (0, 1, 0, 1))
    ref = F.conv2d(xi, wt.float(), bias.float(), stride=stride, padding=1 if pad_mode == 0 else 0)
    out = fie.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(DEV), fie.pack_conv3x3(wt.to(DEV)), cout, stride=stride,
                      pad_mode=pad_mode, upsample=ups, bias=bias.to(DEV))
    assert out.shape[1:3] == ref.shape[2:]
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 3e-3


def test_conv3x3_padded_cin_and_epilogue(fie):
    from fie_amd import hip
    # 3-channel image stored 8-channel padded; rowbias (time embedding) + residual + SiLU
    b, h, w, cin, cout = 2, 32, 32, 3, 64
    x, wt = rnd(b, cin, h, w, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.2)
    tb, res = rnd(b, cout, seed=3), rnd(b, cout, h, w, seed=4)
    ref = F.silu(F.conv2d(x.float(), wt.float(), None, padding=1) + tb.float()[:, :, None, None]) + res.float()
    xp = torch.zeros(b, h, w, 8, dtype=torch.float16)
    xp[..., :3] = x.permute(0, 2, 3, 1)
    out = fie.conv3x3(xp.to(DEV), fie.pack_conv3x3(wt.to(DEV), cin_pad=8), cout, rowbias=tb.to(DEV),
                      residual=res.permute(0, 2, 3, 1).contiguous().to(DEV), act=hip.ACT_SILU)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 3e-3


@pytest.mark.parametrize("b,hn,tq,tk,d,causal", [(1, 10, 4096, 4096, 64, False), (2, 20, 1024, 1024, 64, False),
                                                 (2, 10, 4096, 77, 64, False), (1, 12, 77, 77, 64, True),
                                                 (1, 3, 200, 333, 64, False), (1, 1, 1024, 1024, 512, False),
                                                 (1, 1, 100, 100, 512, False)])
def test_attention(fie, b, hn, tq, tk, d, causal):
    c = hn * d
    q, k, v = rnd(b * tq, c, seed=1), rnd(b * tk, c, seed=2), rnd(b * tk, c, seed=3)
    qf = q.float().view(b, tq, hn, d).transpose(1, 2)
    kf = k.float().view(b, tk, hn, d).transpose(1, 2)
    vf = v.float().view(b, tk, hn, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2).reshape(b * tq, c)
    out = fie.attention(q.to(DEV), k.to(DEV), v.to(DEV), hn, d, tq, tk, b, causal=causal)
    assert rel_err(out, ref) < 4e-3


@pytest.mark.parametrize("variant", [2, 3, 4, 5])
def test_attention_ab_variants_agree(fie, variant):
    """The A/B forms of the d = 64 kernel (fie_debug_attn_variant: 2 / 3 = 128 / 64 queries per block, 4 = two waves x 32 queries, 5 = three-stage
    K / V ring with a counted vmcnt) compute what the default does: self-attention, the 77-key cross-attention, CLIP's causal 77 x 77, a ragged
    length that ends inside a key tile."""
    from fie_amd import hip
    g = torch.Generator().manual_seed(variant)
    cases = [(2, 10, 1024, 1024, False), (2, 20, 256, 77, False), (2, 12, 77, 77, True), (1, 4, 333, 200, False)]
    try:
        for b, hn, tq, tk, causal in cases:
            q, k, v = (torch.randn(b * t, hn * 64, generator=g).half().to(DEV) for t in (tq, tk, tk))
            assert hip.lib().fie_debug_attn_variant(fie.h, 0) == 0
            ref = fie.attention(q, k, v, hn, 64, tq, tk, b, causal=causal).float()
            assert hip.lib().fie_debug_attn_variant(fie.h, variant) == 0
            out = fie.attention(q, k, v, hn, 64, tq, tk, b, causal=causal).float()
            assert rel_err(out, ref) < 2e-3, (b, hn, tq, tk, causal)
    finally:
        hip.lib().fie_debug_attn_variant(fie.h, 0)


def test_attention_fused_qkv_strides(fie):
    # q, k, v as column slices of one fused projection buffer (row stride 3C)
    b, hn, t, d = 1, 4, 256, 64
    c = hn * d
    qkv = rnd(b * t, 3 * c, seed=7).to(DEV)
    out = fie.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], hn, d, t, t, b)
    f = qkv.float().cpu()
    sp = lambda x: x.reshape(b, t, hn, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(f[:, :c]), sp(f[:, c:2 * c]), sp(f[:, 2 * c:])).transpose(1, 2).reshape(b * t, c)
    assert rel_err(out, ref) < 4e-3


def test_attention_forced_rescale(fie):
    # spike one late key so the running max jumps mid-sequence (exercises the online-softmax rescale path)
    b, hn, t, d = 1, 1, 512, 64
    q, k, v = rnd(t, d, seed=1), rnd(t, d, seed=2), rnd(t, d, seed=3)
    k[300] = q[5] * 4
    ref = F.scaled_dot_product_attention(q.float()[None, None], k.float()[None, None], v.float()[None, None])[0, 0]
    out = fie.attention(q.to(DEV), k.to(DEV), v.to(DEV), hn, d, t, t, b)
    assert rel_err(out, ref) < 4e-3


@pytest.mark.parametrize("b,rows,c1,c2,groups,silu", [(1, 1024, 320, 0, 32, True), (2, 256, 1280, 640, 32, True),
                                                      (1, 4096, 128, 0, 32, False), (1, 256, 1280, 1280, 32, True),
                                                      (1, 100, 64, 0, 32, True), (1, 65536, 128, 0, 32, True),
                                                      (2, 1024, 640, 320, 32, True),
                                                      # single-pass kernel shapes (32x32 / 64x64 latent maps), ragged rows, concat
                                                      (2, 1024, 1280, 0, 32, True), (2, 1024, 1280, 1280, 32, True),
                                                      (2, 4096, 640, 0, 32, True), (1, 1000, 1280, 0, 32, False),
                                                      (2, 1024, 640, 640, 32, True), (1, 2051, 640, 0, 32, True)])
def test_groupnorm(fie, b, rows, c1, c2, groups, silu):
    x1 = rnd(b, rows, c1, seed=1) + 0.5
    x2 = rnd(b, rows, c2, seed=2) * 2 if c2 else None
    c = c1 + c2
    gamma, beta = rnd(c, seed=3), rnd(c, seed=4)
    xc = torch.cat([x1, x2], -1) if c2 else x1
    ref = F.group_norm(xc.float().transpose(1, 2), groups, gamma.float(), beta.float(), 1e-5).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    out = fie.groupnorm(x1.to(DEV), gamma.to(DEV), beta.to(DEV), groups, 1e-5, silu, x2=x2.to(DEV) if c2 else None)
    assert rel_err(out, ref) < 3e-3


@pytest.mark.parametrize("b,rows,c1,c2", [(2, 1024, 1280, 0), (2, 1024, 1280, 1280), (2, 4096, 640, 0), (1, 4096, 128, 0)])
def test_groupnorm_single_pass_matches_three_kernel_path(fie, b, rows, c1, c2):
    from fie_amd import hip
    x1 = (rnd(b, rows, c1, seed=1) + 0.5).to(DEV)
    x2 = (rnd(b, rows, c2, seed=2) * 2).to(DEV) if c2 else None
    gamma, beta = rnd(c1 + c2, seed=3).to(DEV), rnd(c1 + c2, seed=4).to(DEV)
    one = fie.groupnorm(x1, gamma, beta, 32, 1e-5, True, x2=x2)
    try:
        hip.lib().fie_debug_gn_onepass(fie.h, 0)
        three = fie.groupnorm(x1, gamma, beta, 32, 1e-5, True, x2=x2)
    finally:
        hip.lib().fie_debug_gn_onepass(fie.h, 1)
    assert rel_err(one, three.float()) < 1e-3


@pytest.mark.parametrize("rows,c", [(4096, 640), (1024, 1280), (77, 768), (5, 64)])
def test_layernorm(fie, rows, c):
    x, g, bta = rnd(rows, c, seed=1) * 3 + 1, rnd(c, seed=2), rnd(c, seed=3)
    ref = F.layer_norm(x.float(), (c,), g.float(), bta.float(), 1e-5)
    out = fie.layernorm(x.to(DEV), g.to(DEV), bta.to(DEV))
    assert rel_err(out, ref) < 3e-3


@pytest.mark.parametrize("m,n,k,geglu,tile", [(2048, 3840, 1280, False, 96), (2048, 1280, 1280, False, 42), (8192, 640, 640, False, 42),
                                             (2048, 10240, 1280, True, 64), (1000, 640, 640, False, 42), (154, 1280, 1280, False, 42),
                                             (300, 2560, 320, True, 64), (8192, 1920, 640, False, 96), (8192, 1920, 640, False, 42)])
def test_gemm_with_layernorm_folded_in(fie, m, n, k, geglu, tile):
    """fie_gemm_ln_f16 (LayerNorm folded into the consumer GEMM: statistics from the activation fragments, rstd * (acc - mean * colsum) + b' in the epilogue)
    against LayerNorm -> Linear (-> GEGLU) in fp32 on the same f16 inputs, on each of the three tiles it is offered on; rows with a large common offset
    (mean ~ 3 sigma) so the mean correction carries weight; ragged M.  Also against the two-launch HIP sequence."""
    from fie_amd import hip
    x = (rnd(m, k, seed=1) * 2 + rnd(m, 1, seed=7) * 6)
    w, b = rnd(n, k, seed=2) / math.sqrt(k), rnd(n, seed=3) * 0.1
    g, bta = 1 + 0.2 * rnd(k, seed=4), 0.1 * rnd(k, seed=5)
    y = F.layer_norm(x.float(), (k,), g.float(), bta.float(), 1e-5) @ w.float().t() + b.float()
    ref = y[:, : n // 2] * F.gelu(y[:, n // 2:]) if geglu else y
    wp, tab = fie.fold_layernorm(w, b, g, bta, geglu=geglu)
    act = hip.ACT_GEGLU if geglu else hip.ACT_NONE
    fie.force_tile(tile)
    try:
        out = fie.gemm_ln(x.to(DEV), wp, n, tab, act=act)
        kern = hip.last_gemm_kernel(fie)
    finally:
        fie.force_tile(0)
    assert f"tile code {tile}" in kern, kern
    two = fie.gemm(fie.layernorm(x.to(DEV), g.to(DEV), bta.to(DEV)), fie.pack_linear(w, geglu=geglu), n, act=act,
                   bias=(torch.stack([b[: n // 2], b[n // 2:]], 1).reshape(-1) if geglu else b).to(DEV))
    e_fold, e_two = rel_err(out, ref), rel_err(two, ref)
    print(f"LN fold M={m} N={n} K={k} tile {tile}: folded {e_fold:.2e}, LayerNorm + GEMM {e_two:.2e} (vs fp32)")
    assert e_fold < 3e-3 and e_fold < 2 * e_two + 1e-4
    # whichever tile the rule takes adds K (and the row sums) in the same order as the forced one: bit-equal
    assert torch.equal(out, fie.gemm_ln(x.to(DEV), wp, n, tab, act=act))


@pytest.mark.parametrize("m,n,k,geglu,tile", [(8192, 640, 640, False, 42), (8192, 1920, 640, False, 96), (16384, 640, 640, False, 42), (4096, 1280, 1280, False, 42),
                                             (2048, 3840, 1280, False, 96), (8192, 5120, 640, True, 64), (2048, 10240, 1280, True, 64)])
def test_gemm_with_layernorm_folded_in_is_stable_on_a_loaded_chip(fie, m, n, k, geglu, tile):
    """Twenty launches of one LayerNorm-folded GEMM at grid sizes that keep several blocks resident per CU: bit-equal every time and within the fp32 bar.
    (Round 4: the 128x80 instantiation dropped the mean correction of single columns in 16-row spots at these sizes, different places every launch,
    never at M = 256 -- profiles/r04_ln_fold_tile48_anomaly.md; it is not offered.  This is the screen for the tiles that are.)"""
    from fie_amd import hip
    x = (rnd(m, k, seed=11) * 2 + rnd(m, 1, seed=17) * 6)
    w, b = rnd(n, k, seed=12) / math.sqrt(k), rnd(n, seed=13) * 0.1
    g, bta = 1 + 0.2 * rnd(k, seed=14), 0.1 * rnd(k, seed=15)
    y = F.layer_norm(x.float(), (k,), g.float(), bta.float(), 1e-5) @ w.float().t() + b.float()
    ref = y[:, : n // 2] * F.gelu(y[:, n // 2:]) if geglu else y
    wp, tab = fie.fold_layernorm(w, b, g, bta, geglu=geglu)
    act = hip.ACT_GEGLU if geglu else hip.ACT_NONE
    xd = x.to(DEV)
    fie.force_tile(tile)
    try:
        first = fie.gemm_ln(xd, wp, n, tab, act=act).clone()
        differ = sum(int(not torch.equal(fie.gemm_ln(xd, wp, n, tab, act=act), first)) for _ in range(20))
    finally:
        fie.force_tile(0)
    assert rel_err(first, ref) < 3e-3 and differ == 0, (rel_err(first, ref), differ)


def test_gemm_with_layernorm_folded_in_refuses_other_tiles(fie):
    from fie_amd import hip
    w, b, g, bta = rnd(640, 640, seed=2) / 25, rnd(640, seed=3), 1 + 0.2 * rnd(640, seed=4), 0.1 * rnd(640, seed=5)
    wp, tab = fie.fold_layernorm(w, b, g, bta)
    fie.force_tile(48)           # not an LN tile: the rule's choice runs instead
    try:
        fie.gemm_ln(rnd(256, 640, seed=1).to(DEV), wp, 640, tab)
        assert "tile code 42" in hip.last_gemm_kernel(fie)
    finally:
        fie.force_tile(0)
    wp, tab = fie.fold_layernorm(w[:384], b[:384], g, bta, geglu=True)
    with pytest.raises(hip.FieError):
        fie.gemm_ln(rnd(256, 640, seed=1).to(DEV), wp, 384, tab, act=hip.ACT_GEGLU)      # GEGLU needs N % 320 == 0 (the 256x320 tile)


def test_sinusoid_known_answers(fie):
    # SURVEY A.1 KAT: t = 499, dim 320
    out = torch.zeros(1, 320, device=DEV, dtype=torch.float16)
    fie.sinusoid(torch.tensor([[499.0]], device=DEV), 320, out)
    o = out.float().cpu()[0]
    assert torch.allclose(o[0:3], torch.tensor([-0.87116218, 0.98838931, 0.19755381]), atol=2e-3)
    assert torch.allclose(o[160:163], torch.tensor([0.49099535, -0.15194249, -0.98029202]), atol=2e-3)


def test_pixels_roundtrip_exact(fie):
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.integers(0, 256, (64, 48, 3), dtype=np.uint8))
    x = fie.pixels_in(img.to(DEV), True, copies=2)
    ref = 2.0 * (img.float() / 255.0) - 1.0
    assert torch.allclose(x[1, ..., :3].float().cpu(), ref, atol=1e-3) and x[..., 3:].abs().max() == 0
    back = fie.pixels_out(x[:1])
    assert torch.equal(back.cpu(), img)          # u8 -> f16 -> u8 is the identity


def test_lcm_step_and_latent_prep(fie):
    hw = 32 * 32
    g = torch.Generator().manual_seed(0)
    mom = torch.randn(hw, 8, generator=g).half()
    e1, e2 = torch.randn(4, hw, generator=g), torch.randn(4, hw, generator=g)
    lat = torch.empty(hw, 4, device=DEV)
    mi = torch.empty(2, hw, 8, device=DEV, dtype=torch.float16)
    fie.latent_prep(mom.to(DEV), e1.to(DEV), e2.to(DEV), hw, 0.13025, 0.5269, 0.8499, lat, mi)
    mean, logvar = mom.float()[:, :4], mom.float()[:, 4:].clamp(-30, 20)
    z0 = (mean + torch.exp(0.5 * logvar) * e1.T) * 0.13025
    ref = 0.5269 * z0 + 0.8499 * e2.T
    assert torch.allclose(lat.cpu(), ref, atol=1e-5)
    assert torch.allclose(mi[1, :, :4].float().cpu(), ref, atol=2e-3) and mi[..., 4:].abs().max() == 0
    eps = torch.randn(2, hw, 4, generator=g).half()
    z = torch.randn(4, hw, generator=g)
    dec = torch.empty(hw, 8, device=DEV, dtype=torch.float16)
    fie.lcm_step(eps.to(DEV), 2, 1.5, lat, z.to(DEV), hw, 0.5269, 0.8499, 1e-8, 1.0, 0.8118, 0.5840, mi, 1 / 0.13025, dec)
    e = eps[0].float() + 1.5 * (eps[1].float() - eps[0].float())
    x0 = (ref - 0.8499 * e) / 0.5269
    ref2 = 0.8118 * (1.0 * x0 + 1e-8 * ref) + 0.5840 * z.T
    assert torch.allclose(lat.cpu(), ref2, atol=1e-4)
    assert torch.allclose(dec[:, :4].float().cpu(), ref2 / 0.13025, rtol=2e-3, atol=2e-3)


def test_clip_embed(fie):
    tok, pos = rnd(1000, 64, seed=1), rnd(77, 64, seed=2)
    ids = torch.randint(0, 1000, (2, 77), dtype=torch.int32)
    out = fie.clip_embed(ids.to(DEV), tok.to(DEV), pos.to(DEV))
    ref = (tok.float()[ids.long()] + pos.float()[None]).reshape(-1, 64)
    assert rel_err(out, ref) < 2e-3


@pytest.mark.parametrize("code", [1, 2, 3, 42, 43, 44, 46, 47, 48, 20048, 51, 52, 54, 61, 62, 81, 82, 95, 96, 1042, 2042, 1062, 2081,
                                  20096, 30096, 40096, 30095, 20051, 30047, 20054, 40052, 30042, 20043, 21096, 32047])
def test_gemm_conv_every_shipped_kernel(fie, code):
    """Every kernel / tile the launch table can select (gemm_conv.hip kTiles; + 2000 = m-tiles-fastest order; + 10000 * s =
    split-K over s blocks per tile, which falls back to fewer slices where a slice would get under 4 K-steps) gives the
    reference result on GEMMs with ragged M / N / K tails and on convs with stride 2, asymmetric pad and fused upsample; a
    code the shape is not eligible for raises instead of launching."""
    from fie_amd import hip
    try:
        fie.force_tile(code)
        for m, n, k in [(300, 200, 72), (1024, 1280, 1280), (77, 640, 2048), (128, 128, 64), (600, 520, 192), (2048, 1280, 1280)]:
            a, w, bias = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3)
            out = fie.gemm(a.to(DEV), fie.pack_linear(w.to(DEV)), n, bias=bias.to(DEV))
            assert rel_err(out, a.float() @ w.float().T + bias.float()) < 3e-3, (m, n, k)
        # A = [A1 | A2] column concatenation (the up-block shortcut), seam on a K-tile boundary
        a1, a2, w = rnd(700, 128, seed=4), rnd(700, 192, seed=5), rnd(264, 320, seed=6, scale=320 ** -0.5)
        out = fie.gemm(a1.to(DEV), fie.pack_linear(w.to(DEV)), 264, a2=a2.to(DEV))
        assert rel_err(out, torch.cat([a1, a2], 1).float() @ w.float().T) < 3e-3
        for b, h, w_, cin, cout, stride, pad_mode, ups in [(1, 32, 32, 64, 64, 1, 0, False), (2, 16, 16, 320, 128, 1, 0, False),
                                                           (1, 32, 32, 128, 64, 2, 1, False), (1, 16, 16, 64, 128, 1, 0, True),
                                                           (1, 24, 40, 16, 16, 1, 0, False), (1, 20, 20, 96, 32, 2, 0, False),
                                                           (1, 9, 7, 192, 64, 1, 0, False), (2, 32, 48, 128, 192, 1, 0, False),
                                                           (1, 64, 64, 64, 4, 1, 0, False), (1, 48, 40, 128, 320, 1, 0, False)]:
            x = rnd(b, cin, h, w_, seed=1)
            wt = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
            xi = x.float()
            if ups:
                xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
            if pad_mode == 1:
                xi = F.pad(xi, (0, 1, 0, 1))
            ref = F.conv2d(xi, wt.float(), None, stride=stride, padding=1 if pad_mode == 0 else 0)
            try:
                out = fie.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(DEV), fie.pack_conv3x3(wt.to(DEV)), cout,
                                  stride=stride, pad_mode=pad_mode, upsample=ups)
            except hip.FieError as e:        # the LDS-DMA kernels refuse (loudly) shapes outside their contract
                assert code % 1000 >= 40 and "not eligible" in str(e) and cin % 64 != 0
                continue
            assert rel_err(out.permute(0, 3, 1, 2), ref) < 3e-3, (b, h, w_, cin, cout, stride, pad_mode, ups)
    finally:
        fie.force_tile(0)
    with pytest.raises(hip.FieError, match="unknown tile code"):
        try:
            fie.force_tile(66)               # never a tile code
            fie.gemm(rnd(64, 64).to(DEV), fie.pack_linear(rnd(64, 64).to(DEV)), 64)
        finally:
            fie.force_tile(0)


@pytest.mark.parametrize("code", [63, 64])
def test_gemm_view_only_tiles(fie, code):
    """Tile code built for the GEMM view only: 63 (256x320, 8 waves, wave tile 128x80, two-stage ring: the exact-fit tile of the FF1
    projection, M 2048 x N 10240 = 256 tiles = one per CU).  GEGLU at the real shape, ragged M / N / K with bias + row bias + SiLU + scale + in-place
    residual, the [A1 | A2] column concatenation, against fp32 torch; the conv view refuses the code loudly."""
    from fie_amd import hip
    try:
        fie.force_tile(code)
        a, w, b = rnd(2048, 1280, seed=1), rnd(10240, 1280, seed=2, scale=1280 ** -0.5), rnd(10240, seed=3)
        out = fie.gemm(a.to(DEV), fie.pack_linear(w.to(DEV), geglu=True), 10240, act=hip.ACT_GEGLU,
                       bias=torch.stack([b[:5120], b[5120:]], 1).reshape(-1).contiguous().to(DEV))
        assert ("256x320" if code in (63, 64) else "256x128") in hip.last_gemm_kernel(fie)
        full = a.float() @ w.float().T + b.float()
        assert rel_err(out, full[:, :5120] * F.gelu(full[:, 5120:])) < 3e-3
        for m, n, k in [(1000, 640, 200), (300, 328, 72), (2048, 1280, 5120), (77, 960, 2048)]:
            a, w, bias, res, rb = rnd(m, k, seed=m), rnd(n, k, seed=n, scale=k ** -0.5), rnd(n, seed=3), rnd(m, n, seed=4), rnd(2, n, seed=5)
            inplace = res.to(DEV).clone()
            out = fie.gemm(a.to(DEV), fie.pack_linear(w.to(DEV)), n, bias=bias.to(DEV), rowbias=rb.to(DEV), rows_per_batch=(m + 1) // 2,
                           residual=inplace, out=inplace, scale=0.5, act=hip.ACT_SILU)
            ref = a.float() @ w.float().T + bias.float() + rb.float().repeat_interleave((m + 1) // 2, 0)[:m]
            assert rel_err(out, F.silu(ref) * 0.5 + res.float()) < 3e-3, (m, n, k)
        a1, a2, w = rnd(700, 128, seed=4), rnd(700, 192, seed=5), rnd(320, 320, seed=6, scale=320 ** -0.5)
        out = fie.gemm(a1.to(DEV), fie.pack_linear(w.to(DEV)), 320, a2=a2.to(DEV))
        assert rel_err(out, torch.cat([a1, a2], 1).float() @ w.float().T) < 3e-3
        with pytest.raises(hip.FieError, match="GEMM view only"):
            fie.conv3x3(rnd(1, 16, 16, 64, seed=7).to(DEV), fie.pack_conv3x3(rnd(64, 64, 3, 3, seed=8).to(DEV)), 64)
    finally:
        fie.force_tile(0)


@pytest.mark.parametrize("code", [42, 43, 44, 47, 48, 51, 52, 54, 62, 95, 96])
def test_epilogue_operands_loaded_ahead_of_the_k_loop(fie, code):
    """Ring kernels read the bias row and the residual tile BEFORE the K loop (gemm_common.h: EpiPre; round 3).  Same values, same
    arithmetic: bit-identical to the in-epilogue loads (fie_debug_epilogue_prefetch 0), ragged edges and an in-place residual included,
    for the GEMM and the conv view of every ring tile."""
    try:
        fie.force_tile(code)
        for m, n, k in [(2048, 1280, 1280), (300, 328, 200), (77, 960, 2048)]:
            a, w, bias, res = rnd(m, k, seed=m).to(DEV), fie.pack_linear(rnd(n, k, seed=n, scale=k ** -0.5).to(DEV)), rnd(n, seed=3).to(DEV), rnd(m, n, seed=4)
            outs = []
            for pre in (False, True):
                fie.epi_prefetch = pre
                inplace = res.to(DEV).clone()
                outs.append(fie.gemm(a, w, n, bias=bias, residual=inplace, out=inplace).clone())
            assert torch.equal(outs[0], outs[1]), (code, m, n, k)
            # the plain-Linear launches take the kernel with the LEAN epilogue (absent operands read as zero): every combination of bias / residual
            for use_bias, use_res in ((False, False), (True, False), (False, True)):
                outs = []
                for pre in (False, True):
                    fie.epi_prefetch = pre
                    outs.append(fie.gemm(a, w, n, bias=bias if use_bias else None, residual=res.to(DEV) if use_res else None).clone())
                assert torch.equal(outs[0], outs[1]), (code, m, n, k, use_bias, use_res)
                ref = a.float() @ rnd(n, k, seed=n, scale=k ** -0.5).to(DEV).float().T + (bias.float() if use_bias else 0) + (res.to(DEV).float() if use_res else 0)
                assert rel_err(outs[1], ref) < 3e-3
        x, wc, bias, res = rnd(2, 24, 20, 128, seed=5).to(DEV), fie.pack_conv3x3(rnd(192, 128, 3, 3, seed=6, scale=1152 ** -0.5).to(DEV)), rnd(192, seed=7).to(DEV), rnd(2, 24, 20, 192, seed=8).to(DEV)
        outs = []
        for pre in (False, True):
            fie.epi_prefetch = pre
            outs.append(fie.conv3x3(x, wc, 192, bias=bias, residual=res).clone())
        assert torch.equal(outs[0], outs[1]), code
    finally:
        fie.epi_prefetch = True
        fie.force_tile(0)


def test_phased_256x256_kernel_large_and_odd_ktiles(fie):
    """gemm8_kernel (tile code 81) at the sizes it is selected for, against the ring kernel and fp32 torch: K-tile counts 1, 2,
    3 (odd: the buffer-parity tail), 20 and 64; epilogues (bias, row bias, SiLU, GEGLU, scale + residual); conv with tap
    changes mid-stream; an identity check with an ASYMMETRIC weight (catches a transposed C write)."""
    from fie_amd import hip
    try:
        for m, n, k in [(512, 512, 64), (512, 256, 128), (768, 512, 192), (2048, 1280, 1280), (4096, 4096, 4096), (1000, 700, 200)]:
            a, w = rnd(m, k, seed=m), rnd(n, k, seed=n + 1, scale=k ** -0.5)
            bias, res = rnd(n, seed=3), rnd(m, n, seed=4)
            rb = rnd(2, n, seed=5)
            wp = fie.pack_linear(w.to(DEV))
            outs = {}
            for code in (81, 42):
                fie.force_tile(code)
                outs[code] = fie.gemm(a.to(DEV), wp, n, bias=bias.to(DEV), residual=res.to(DEV), scale=0.5, act=hip.ACT_SILU,
                                      rowbias=rb.to(DEV), rows_per_batch=(m + 1) // 2)
            ref = a.float() @ w.float().T + bias.float() + rb.float().repeat_interleave((m + 1) // 2, 0)[:m]
            ref = F.silu(ref) * 0.5 + res.float()
            assert rel_err(outs[81], ref) < 3e-3, (m, n, k)
            assert rel_err(outs[81], outs[42].float()) < 2e-3
        fie.force_tile(81)
        # GEGLU epilogue
        m, n, k = 1024, 2560, 320
        a, w, bias = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3)
        bias_i = torch.stack([bias[: n // 2], bias[n // 2:]], 1).reshape(-1).contiguous()       # (value, gate) interleaved like the packed rows
        out = fie.gemm(a.to(DEV), fie.pack_linear(w.to(DEV), geglu=True), n, bias=bias_i.to(DEV), act=hip.ACT_GEGLU)
        full = a.float() @ w.float().T + bias.float()
        assert rel_err(out, full[:, : n // 2] * F.gelu(full[:, n // 2:])) < 3e-3
        # identity activations x asymmetric weight: C must equal W^T exactly
        eye = torch.eye(512, dtype=torch.float16)
        w = (torch.arange(512 * 512, dtype=torch.float32).reshape(512, 512) % 251 - 125).half()
        out = fie.gemm(eye.to(DEV), fie.pack_linear(w.to(DEV)), 512)
        assert torch.equal(out.cpu(), w.T.contiguous())
        # conv: 9 taps x {1, 2, 5} channel steps, stride 2 with asymmetric pad, fused upsample
        for b, h, w_, cin, cout, stride, pad_mode, ups in [(1, 64, 64, 64, 256, 1, 0, False), (2, 32, 32, 128, 512, 1, 0, False),
                                                           (1, 32, 32, 320, 256, 1, 0, True), (1, 66, 62, 128, 256, 2, 1, False)]:
            x = rnd(b, cin, h, w_, seed=1)
            wt = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
            xi = x.float()
            if ups:
                xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
            if pad_mode == 1:
                xi = F.pad(xi, (0, 1, 0, 1))
            ref = F.conv2d(xi, wt.float(), None, stride=stride, padding=1 if pad_mode == 0 else 0)
            out = fie.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(DEV), fie.pack_conv3x3(wt.to(DEV)), cout,
                              stride=stride, pad_mode=pad_mode, upsample=ups)
            assert hip.last_gemm_kernel(fie).startswith("gemm8_kernel")
            assert rel_err(out.permute(0, 3, 1, 2), ref) < 3e-3
    finally:
        fie.force_tile(0)


def test_phased_kernel_is_race_free_over_repeats(fie):
    """The counted-vmcnt / barrier protocol of gemm8_kernel: 40 back-to-back launches on a K = 5120 GEMM and on a conv must be
    bit-identical (an early LDS read or a late refill shows up as rare wrong tiles)."""
    try:
        fie.force_tile(81)
        a, w = rnd(2048, 5120, seed=1).to(DEV), fie.pack_linear(rnd(1280, 5120, seed=2, scale=5120 ** -0.5).to(DEV))
        first = fie.gemm(a, w, 1280).clone()
        x = rnd(1, 64, 64, 256, seed=3).to(DEV)
        wc = fie.pack_conv3x3(rnd(256, 256, 3, 3, seed=4, scale=2304 ** -0.5).to(DEV))
        cfirst = fie.conv3x3(x, wc, 256).clone()
        for _ in range(40):
            assert torch.equal(fie.gemm(a, w, 1280), first)
            assert torch.equal(fie.conv3x3(x, wc, 256), cfirst)
    finally:
        fie.force_tile(0)


@pytest.mark.parametrize("seed,h,w,lo,hi", [(0, 64, 64, 100, 200), (1, 97, 131, 100, 200), (2, 256, 256, 50, 150), (3, 1024, 1024, 100, 200),
                                            (4, 1024, 1024, 20, 60), (5, 33, 70, 200, 100)])
def test_canny_device_bit_exact(fie, seed, h, w, lo, hi):
    """Device Canny == host C++ entry == numpy oracle, bit for bit; long weak chains need several hysteresis passes."""
    from fie_amd import hip
    from oracle import canny as ocanny
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([128 + 100 * np.sin(xx / rng.uniform(5, 30) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(5, 30)) for _ in range(3)], 2)
    for _ in range(10):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1, max(2, min(h, w) / 4))
        img[((xx - cx) ** 2 + (yy - cy) ** 2) < r * r] = rng.uniform(0, 255, 3)
    img = (img + rng.normal(0, 5, img.shape)).clip(0, 255).astype(np.uint8)
    got = fie.canny_device(torch.from_numpy(img).to(DEV), lo, hi).cpu().numpy()
    assert np.array_equal(got, hip.canny_rgb(img, lo, hi))
    assert np.array_equal(got, ocanny.canny_rgb(img, lo, hi))
    assert fie.canny_passes >= 2
    # the two-call form FastEditor.edit() uses (fie_canny_rgb_device_begin_u8 / _finish_u8): nothing is waited for in begin, same bits after finish
    src = torch.from_numpy(img).to(DEV)
    edges, state = fie.canny_begin(src, lo, hi, rounds=4)
    assert fie.canny_finish(state) is edges and np.array_equal(edges.cpu().numpy(), got) and fie.canny_more == 0 and fie.canny_passes == 16


def test_canny_device_long_chain(fie):
    """A weak ramp edge 1000 pixels long seeded by one strong pixel: the closure must cross ~32 tiles."""
    from oracle import canny as ocanny
    img = np.full((64, 1024, 3), 100, np.uint8)
    img[32:, :, :] = 130                      # L1 gradient 4*30 = 120: weak everywhere (100 < 120 <= 200)
    img[32:, :8, :] = 200                     # strong seed at the left end
    got = fie.canny_device(torch.from_numpy(img).to(DEV), 100, 200).cpu().numpy()
    edges, state = fie.canny_begin(torch.from_numpy(img).to(DEV), 100, 200, rounds=4)       # 16 passes cannot finish this one: finish runs the further rounds
    assert np.array_equal(fie.canny_finish(state).cpu().numpy(), got) and fie.canny_more > 0 and fie.canny_passes > 30
    ref = ocanny.canny_rgb(img, 100, 200)
    assert np.array_equal(got, ref) and ref[:, 900:].max() == 255 and fie.canny_passes > 8


@pytest.mark.parametrize("h,w,oh,ow", [(512, 512, 1024, 1024), (333, 517, 1024, 1024), (1500, 1100, 1024, 1024), (64, 48, 96, 80),
                                       (1024, 700, 1024, 1024), (700, 1024, 1024, 1024), (1024, 1024, 1024, 1024)])
def test_resize_lanczos_device_bit_exact_with_pillow(fie, h, w, oh, ow):
    from PIL import Image
    rng = np.random.default_rng(h * 7 + w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 2] = (np.linspace(0, 255, w)[None, :, None] * np.ones((h // 2, 1, 3))).astype(np.uint8)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS))
    out = fie.resize_lanczos(torch.from_numpy(a).to(DEV), oh, ow).cpu().numpy()
    assert np.array_equal(out, ref)


def test_tile_override_steers_one_shape(fie):
    """`fie_debug_tile_override` (per ctx) steers ONE shape (the whole-UNet trial tools rely on it); other shapes keep the
    heuristic; clearing restores it."""
    from fie_amd import hip
    m, n, k = 4096, 5120, 128
    a, w, bias = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3)
    ad, bd = a.to(DEV), bias.to(DEV)
    wp = fie.pack_linear(w.to(DEV))
    ref = fie.gemm(ad, wp, n, bias=bd).clone()
    fie.gemm(ad[:256], wp, n, bias=bd)
    default_small = hip.last_gemm_kernel(fie)
    try:
        assert fie.tile_override(f"0,{m},{n},{k}=3;1,1,1,1=42") == 2
        out = fie.gemm(ad, wp, n, bias=bd).clone()
        assert "tile code 3)" in hip.last_gemm_kernel(fie)
        other = fie.gemm(ad[:256], wp, n, bias=bd).clone()            # a different M: not overridden
        assert hip.last_gemm_kernel(fie) == default_small
    finally:
        assert fie.tile_override(None) == 0
    assert rel_err(out, ref.float()) < 2e-3 and rel_err(other, ref[:256].float()) < 2e-3
    assert rel_err(ref, a.float() @ w.float().T + bias.float()) < 3e-3


def test_gemm_autotune_picks_by_measurement_and_keeps_results(fie):
    """fie_gemm_autotune (include/fie.h): the first eager launch of a shape times the eligible tiles (weights flushed cold) and
    the context remembers one; results are bit-identical to the built-in rule's (every tile accumulates K in the same order), an
    in-place residual (res == C) survives the timing launches (they write a scratch output), a stream capture never tunes,
    mode 2 keeps the remembered code without tuning new shapes, mode 0 returns to the rule."""
    from fie_amd import hip
    m, n, k = 2176, 1152, 576                           # shapes no pipeline in this session has met
    a, w, bias = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3)
    ad, bd = a.to(DEV), bias.to(DEV)
    wp = fie.pack_linear(w.to(DEV))
    x = rnd(2, 24, 24, 128, seed=4).to(DEV)
    wc = fie.pack_conv3x3(rnd(192, 128, 3, 3, seed=5, scale=(9 * 128) ** -0.5).to(DEV))
    res0 = rnd(m, n, seed=6).to(DEV)
    ref = fie.gemm(ad, wp, n, bias=bd).clone()
    ref_res = fie.gemm(ad, wp, n, bias=bd, residual=res0, out=res0.clone()).clone()
    ref_conv = fie.conv3x3(x, wc, 192).clone()
    rule_kernel = hip.last_gemm_kernel(fie)
    remembered = fie.autotune_report()[1]               # the session's choices (the pinned table, or what a live session tuned so far): restored below
    fie.tune_exclude("")                                # this test starts from an empty memory, whatever the table holds
    n0 = fie.autotune_report()[0]
    assert n0 == 0
    try:
        fie.splitk(False)                               # split-K changes the fp32 summation order: its own test below
        fie.autotune(1)
        inplace = res0.clone()
        got_res = fie.gemm(ad, wp, n, bias=bd, residual=inplace, out=inplace).clone()      # tuned on this call
        got = fie.gemm(ad, wp, n, bias=bd).clone()
        got_conv = fie.conv3x3(x, wc, 192).clone()
        count, report = fie.autotune_report()
        assert count == n0 + 2 and f"gemm M={m} N={n} K={k}" in report and "conv M=1152 N=192 K=1152" in report
        assert torch.equal(got, ref) and torch.equal(got_res, ref_res) and torch.equal(got_conv, ref_conv)
        fie.autotune(2)                                                                     # frozen: a new shape is not tuned
        fie.gemm(ad[:512], wp, n, bias=bd)
        assert fie.autotune_report()[0] == n0 + 2
        fie.autotune(1)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            small = ad[:1024].clone()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                cap = fie.gemm(small, wp, n, bias=bd)                                      # unseen shape under capture: rule, not tuned
            g.replay()
        torch.cuda.synchronize()
        assert fie.autotune_report()[0] == n0 + 2 and torch.equal(cap, ref[:1024])
    finally:
        fie.autotune(0)
        fie.splitk(True)
        fie.tune_exclude("")
        assert hip.lib().fie_gemm_autotune_load(fie.h, remembered.encode()) == remembered.count(" -> ")
    fie.conv3x3(x, wc, 192)
    assert hip.last_gemm_kernel(fie) == rule_kernel


@pytest.mark.parametrize("code", [71, 72])
def test_halo_resident_conv(fie, code):
    """csrc/conv_halo.hip (SURVEY 2.2 K1 "LDS-staged halo tiles"): the stride-1 3x3 conv with a 16x16 output patch's 18x18 halo resident in LDS per
    64-channel chunk.  Code 71 = one tile per block with the shared epilogue, 72 = persistent blocks with deferred stores (the rule for eligible
    shapes).  Against torch's fp32 conv2d with every epilogue option the resnets use (bias, per-image row bias, residual, GroupNorm sums for the
    consumer); shapes with one and with several tiles per block (400 and 1 200 tiles on 256 CUs), two column tiles, two images; a conv the
    kernel does not take (SiLU epilogue, 24x24 map) falls back / is refused; repeats are bit-identical (the race screen: tools/halo_race.py)."""
    from fie_amd import hip
    g = torch.Generator().manual_seed(code)
    try:
        for b, h, w, cin, cout, opts in [(2, 32, 48, 128, 128, "bias,res"), (2, 64, 64, 256, 256, "bias,gn"), (1, 320, 320, 128, 128, "bias,res,gn"),
                                         (2, 160, 240, 128, 256, "bias,rowbias,gn"), (1, 48, 32, 192, 320, "bias,rowbias"), (1, 16, 16, 64, 128, ""),
                                         (1, 160, 160, 512, 512, "bias,gn"), (2, 96, 96, 640, 640, "bias,res,gn")]:     # 16 channels per group; 20 (as 4-channel quads)
            x = torch.randn(b, h, w, cin, generator=g).half().to(DEV)
            wt = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half()
            bias = torch.randn(cout, generator=g).half().to(DEV) if "bias" in opts else None
            res = torch.randn(b, h, w, cout, generator=g).half().to(DEV) if "res" in opts else None
            rb = torch.randn(b, cout, generator=g).half().to(DEV) if "rowbias" in opts else None
            wp = fie.pack_conv3x3(wt.to(DEV))
            ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float().to(DEV), bias.float() if bias is not None else None, padding=1)
            if rb is not None:
                ref = ref + rb.float()[:, :, None, None]
            if res is not None:
                ref = ref + res.float().permute(0, 3, 1, 2)
            fie.force_tile(code)
            outs = [fie.conv3x3(x, wp, cout, bias=bias, residual=res, rowbias=rb, gn_groups=32 if "gn" in opts else None) for _ in range(6)]
            assert "conv_halo" in hip.last_gemm_kernel(fie), hip.last_gemm_kernel(fie)
            y = outs[0]
            assert rel_err(y.permute(0, 3, 1, 2), ref) < 3e-3, (code, b, h, w, cin, cout, opts)
            assert all(torch.equal(o, y) for o in outs[1:]), (code, b, h, w, cin, cout, opts)
            if "gn" in opts:
                assert getattr(outs[-1], "_gn_tag", None) is not None
                gam, bet = (1 + 0.2 * torch.randn(cout, generator=g)).half().to(DEV), (0.1 * torch.randn(cout, generator=g)).half().to(DEV)
                fie.force_tile(0)
                gn = fie.groupnorm(outs[-1], gam, bet, 32, 1e-5, True)          # consumes the sums the conv's tile ends wrote
                gref = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, gam.float(), bet.float(), 1e-5))
                assert rel_err(gn.permute(0, 3, 1, 2), gref) < 4e-3, (code, opts)
        # not this kernel's: a 24x24 map is refused when forced; an activation epilogue is served by the one-tile-per-block form
        fie.force_tile(code)
        x = torch.randn(1, 24, 24, 128, generator=g).half().to(DEV)
        wp = fie.pack_conv3x3((torch.randn(128, 128, 3, 3, generator=g) * 0.03).half().to(DEV))
        with pytest.raises(hip.FieError, match="halo-resident conv"):
            fie.conv3x3(x, wp, 128)
        x = torch.randn(1, 32, 32, 128, generator=g).half().to(DEV)
        y = fie.conv3x3(x, wp, 128, act=hip.ACT_SILU)
        assert hip.last_gemm_kernel(fie).startswith("conv_halo")
        fie.force_tile(0)
        assert rel_err(y, fie.conv3x3(x, wp, 128, act=hip.ACT_SILU).float()) < 3e-3
    finally:
        fie.force_tile(0)


def test_groupnorm_fused_into_the_halo_conv(fie):
    """GroupNorm + SiLU -> conv3x3 as ONE launch (include/fie.h: fie_groupnorm_coef_f16 + fie_conv3x3_gn_nhwc_f16; csrc/conv_halo.hip, GNA): the conv reads
    the un-normalised tensor and normalises every halo chunk in LDS.  Against the two-launch sequence on the same kernel family (GroupNorm from the
    producer's sums, then the conv on tile code 72): bit for bit, the output's own GroupNorm sums included; with and without a residual, 128 to 1024 input
    channels, one and several tiles per block, a map whose patches touch all four borders; repeats bit-identical; against torch fp32 to rounding."""
    from fie_amd import hip
    g = torch.Generator().manual_seed(77)
    fuse0, fie.gn_fuse_conv = fie.gn_fuse_conv, True      # the product leaves this form off (measured slower: profiles/r04_gn_apply_in_the_halo_conv_negative.log); the op is kept correct
    try:
        for h, w, cin, cout, use_res in [(320, 320, 128, 128, True), (160, 160, 256, 256, False), (128, 160, 512, 256, True), (64, 64, 1024, 128, False),
                                         (32, 48, 128, 128, False), (16, 16, 256, 128, True)]:
            x0 = torch.randn(1, h, w, 64, generator=g).half().to(DEV)
            w0 = (torch.randn(cin, 64, 3, 3, generator=g) * (9 * 64) ** -0.5).half().to(DEV)
            wt = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half().to(DEV)
            b0, b1 = torch.randn(cin, generator=g).half().to(DEV), torch.randn(cout, generator=g).half().to(DEV)
            gam, bet = (1 + 0.3 * torch.randn(cin, generator=g)).half().to(DEV), (0.2 * torch.randn(cin, generator=g)).half().to(DEV)
            res = torch.randn(1, h, w, cout, generator=g).half().to(DEV) if use_res else None
            wp0, wp = fie.pack_conv3x3(w0), fie.pack_conv3x3(wt)
            fie.force_tile(0)
            x = fie.conv3x3(x0, wp0, cin, bias=b0, gn_groups=32)                    # a producer that leaves its sums on x
            fie.force_tile(72)
            ref = fie.conv3x3(fie.groupnorm(x, gam, bet, 32, 1e-6, True), wp, cout, bias=b1, residual=res, gn_groups=32)
            ref_gn = fie.groupnorm(ref, torch.ones(cout).half().to(DEV), torch.zeros(cout).half().to(DEV), 32, 1e-6, False)
            fie.force_tile(0)
            outs = []
            for _ in range(4):
                x2 = fie.conv3x3(x0, wp0, cin, bias=b0, gn_groups=32)
                assert torch.equal(x2, x) and fie.conv3x3_gn_ok(x2, cout, 32)
                coef = fie.groupnorm_coef(x2, gam, bet, 32, 1e-6)
                outs.append(fie.conv3x3_gn(x2, coef, True, wp, cout, bias=b1, residual=res, gn_groups=32))
                assert "conv_halo2" in hip.last_gemm_kernel(fie)
            got_gn = fie.groupnorm(outs[-1], torch.ones(cout).half().to(DEV), torch.zeros(cout).half().to(DEV), 32, 1e-6, False)
            tiles = (h // 16) * (w // 16) * (cout // 128)
            if tiles > 256:                                                         # the two-launch reference ran the same persistent kernel: same bits
                assert torch.equal(outs[0], ref) and torch.equal(got_gn, ref_gn), (h, w, cin, cout, use_res)
            else:                                                                   # it ran the one-tile-per-block form: another epilogue, same sums to rounding
                assert rel_err(outs[0], ref.float()) < 2e-3 and rel_err(got_gn, ref_gn.float()) < 4e-3, (h, w, cin, cout, use_res)
            assert all(torch.equal(o, outs[0]) for o in outs[1:]), (h, w, cin, cout, use_res)
            y32 = F.silu(F.group_norm(x.float().permute(0, 3, 1, 2), 32, gam.float(), bet.float(), 1e-6))
            t32 = F.conv2d(y32, wt.float(), b1.float(), padding=1) + (res.float().permute(0, 3, 1, 2) if use_res else 0)
            assert rel_err(outs[0].permute(0, 3, 1, 2), t32) < 4e-3, (h, w, cin, cout, use_res)
        # where the fused form is not built the query says so (and the entry refuses): two images, a 24-pixel map, 64 input channels
        x = torch.randn(2, 32, 32, 128, generator=g).half().to(DEV)
        x._gn_tag = ("x",)
        assert not fie.conv3x3_gn_ok(x, 128, 32)
        assert not hip.lib().fie_conv3x3_gn_ok(fie.h, 1, 24, 32, 128, 128, 32) and not hip.lib().fie_conv3x3_gn_ok(fie.h, 1, 32, 32, 64, 128, 32)
        assert not hip.lib().fie_conv3x3_gn_ok(fie.h, 1, 32, 32, 128, 320, 32)
    finally:
        fie.force_tile(0)
        fie.gn_fuse_conv = fuse0


def test_halo_resident_conv_with_1x1_side_inputs(fie):
    """A resnet's conv2 + its 1x1 shortcut as one launch (include/fie.h: fie_conv3x3_plus_nhwc_f16) on the halo-resident kernel (code 72): the side
    inputs run as centre-tap K-steps behind the nine-tap chunks.  Against torch fp32; one and several tiles per block, one and two side inputs,
    GroupNorm sums for the consumer; repeats bit-identical; the ring kernel (code 52) agrees to the last bits' difference of a different K order."""
    from fie_amd import hip
    g = torch.Generator().manual_seed(72)
    try:
        for b, h, w, cin, cout, c2, c3, gn in [(1, 64, 64, 128, 128, 256, 0, False), (2, 64, 96, 256, 256, 128, 64, True), (1, 320, 320, 128, 128, 256, 0, True),
                                                (2, 64, 64, 640, 640, 1280, 640, True)]:
            x = torch.randn(b, h, w, cin, generator=g).half().to(DEV)
            x2 = torch.randn(b * h * w, c2, generator=g).half().to(DEV)
            x3 = torch.randn(b * h * w, c3, generator=g).half().to(DEV) if c3 else None
            wt = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half().to(DEV)
            w1 = (torch.randn(cout, c2 + c3, generator=g) * (c2 + c3) ** -0.5).half().to(DEV)
            bias = torch.randn(cout, generator=g).half().to(DEV)
            wplus = torch.cat([fie.pack_conv3x3(wt)[:, :9 * cin], fie.pack_linear(w1)[:, :c2 + c3]], 1).contiguous()
            side = x2 if x3 is None else torch.cat([x2, x3], 1)
            ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), bias.float(), padding=1) \
                + (side.float() @ w1.float().T).view(b, h, w, cout).permute(0, 3, 1, 2)
            fie.force_tile(72)
            outs = [fie.conv3x3_plus(x, wplus, cout, x2, x3, bias=bias, gn_groups=32 if gn else None) for _ in range(5)]
            assert "conv_halo2" in hip.last_gemm_kernel(fie), hip.last_gemm_kernel(fie)
            assert rel_err(outs[0].permute(0, 3, 1, 2), ref) < 4e-3, (b, h, w, cin, cout, c2, c3)
            assert all(torch.equal(o, outs[0]) for o in outs[1:]), (b, h, w, cin, cout, c2, c3)
            fie.force_tile(52)
            ring = fie.conv3x3_plus(x, wplus, cout, x2, x3, bias=bias)
            assert "conv_halo" not in hip.last_gemm_kernel(fie) and rel_err(outs[0], ring.float()) < 2e-3
            if gn:
                fie.force_tile(0)
                gnv = fie.groupnorm(outs[-1], torch.ones(cout).half().to(DEV), torch.zeros(cout).half().to(DEV), 32, 1e-5, True)
                gref = F.silu(F.group_norm(outs[-1].float().permute(0, 3, 1, 2), 32, eps=1e-5))
                assert rel_err(gnv.permute(0, 3, 1, 2), gref) < 4e-3
    finally:
        fie.force_tile(0)


def test_split_k_in_launch_reduction(fie):
    """Split-K for the M = 2048 class (include/fie.h: fie_splitk_workspace; gemm_common.h: splitk_reduce): the K-steps of a tile are
    dealt to s blocks, the block that arrives last sums the fp32 slabs in slice order and runs the epilogue.  Checked: the result
    against fp32 torch and against the unsplit kernel (same tile) with every epilogue option (bias, row bias, activation, scale,
    in-place residual, GEGLU); conv with its 1x1 side inputs, the 2x-upsampling parity conv, GroupNorm sums from the epilogue;
    bit-identical repeats (the sum order does not depend on which block came last) back to back on ONE workspace with changing
    inputs (a stale slab or counter would show), also while a second stream runs split GEMMs on ITS workspace; the counters are
    left zero; without a workspace the launch falls back to the unsplit kernel."""
    from fie_amd import hip
    lib = hip.lib()
    m, n, k = 2048, 1280, 5120
    w = rnd(n, k, seed=2, scale=k ** -0.5)
    wp = fie.pack_linear(w.to(DEV))
    bias, rb = rnd(n, seed=3).to(DEV), rnd(2, n, seed=5).to(DEV)
    try:
        for code in (30096, 20096, 40096, 20095, 30047, 20054, 30052):
            a, res = rnd(m, k, seed=code), rnd(m, n, seed=code + 1)
            ad = a.to(DEV)
            ref = (a.float() @ w.float().T + bias.float().cpu() + rb.float().cpu().repeat_interleave(m // 2, 0))
            ref = F.silu(ref) * 0.5 + res.float()
            fie.force_tile(code)
            inplace = res.to(DEV).clone()
            out = fie.gemm(ad, wp, n, bias=bias, rowbias=rb, rows_per_batch=m // 2, residual=inplace, out=inplace, scale=0.5, act=hip.ACT_SILU)
            assert f"split-K {code // 10000}" in hip.last_gemm_kernel(fie), hip.last_gemm_kernel(fie)
            assert rel_err(out, ref) < 3e-3, code
            fie.force_tile(code % 10000)
            plain = fie.gemm(ad, wp, n, bias=bias, rowbias=rb, rows_per_batch=m // 2, residual=res.to(DEV), scale=0.5, act=hip.ACT_SILU)
            assert "split-K" not in hip.last_gemm_kernel(fie)
            assert rel_err(out, plain.float()) < 1.5e-3, code           # fp32 summation order differs: last f16 bit only
        # GEGLU (FF1-type) and ragged M / N / K tails
        fie.force_tile(30096)
        a, wg, bg = rnd(1000, 1288, seed=11), rnd(520, 1288, seed=12, scale=1288 ** -0.5), rnd(520, seed=13)
        out = fie.gemm(a.to(DEV), fie.pack_linear(wg.to(DEV), geglu=True), 520, act=hip.ACT_GEGLU,
                       bias=torch.stack([bg[:260], bg[260:]], 1).reshape(-1).contiguous().to(DEV))        # (value, gate) interleaved like the rows
        assert "split-K 3" in hip.last_gemm_kernel(fie)
        full = a.float() @ wg.float().T + bg.float()
        assert rel_err(out, full[:, :260] * F.gelu(full[:, 260:])) < 3e-3
        # repeats on one workspace with changing inputs; a second stream splitting on its own workspace meanwhile
        fie.force_tile(30096)
        side = torch.cuda.Stream()
        a0, a1 = rnd(m, k, seed=21).to(DEV), rnd(m, k, seed=22).to(DEV)
        first = [fie.gemm(a0, wp, n, bias=bias).clone(), fie.gemm(a1, wp, n, bias=bias).clone()]
        torch.cuda.synchronize()
        for it in range(12):
            with torch.cuda.stream(side):
                for _ in range(3):
                    other = fie.gemm(a1 if it % 2 else a0, wp, n, bias=bias)
            got = fie.gemm(a0 if it % 2 else a1, wp, n, bias=bias)
            torch.cuda.synchronize()
            assert torch.equal(got, first[(it + 1) % 2]) and torch.equal(other, first[it % 2]), it
        for ws in fie._sk_ws.values():
            assert int(ws[:16384].view(torch.int32).abs().sum()) == 0           # every launch leaves its arrival counters zero
        # self-healing counters (round 4; GPUTEST r03's failure): two streams issue split launches on ONE workspace with no ordering -- a
        # violation of the per-stream rule whose overlapping sums are undefined -- and afterwards every counter is zero again and the next
        # (lawful) launches are exact.  With the old "== S - 1, store 0" ticket a counter stayed at 1 and corrupted every later launch.
        fie.sync_stream()
        fie._bind_splitk()
        fie._sk_pinned = True                              # both streams keep the default stream's workspace
        try:
            for it in range(8):
                with torch.cuda.stream(side):
                    for _ in range(4):
                        fie.gemm(a1, wp, n, bias=bias)
                for _ in range(4):
                    fie.gemm(a0, wp, n, bias=bias)
            torch.cuda.synchronize()
        finally:
            fie._sk_pinned = False
            fie._sk_bound = None
        assert fie.splitk_counters_clear()
        assert torch.equal(fie.gemm(a0, wp, n, bias=bias), first[0]) and torch.equal(fie.gemm(a1, wp, n, bias=bias), first[1])
        # conv 3x3 + its 1x1 side inputs, 2x-upsampling parity conv, GroupNorm sums from the split epilogue
        x = rnd(2, 32, 32, 256, seed=31).to(DEV)
        wt = rnd(512, 256, 3, 3, seed=32, scale=(9 * 256) ** -0.5)
        wc = fie.pack_conv3x3(wt.to(DEV))
        gam, bet = torch.ones(512, dtype=torch.float16, device=DEV), torch.zeros(512, dtype=torch.float16, device=DEV)
        ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wt.float(), None, padding=1)
        for code in (30096, 20095, 40052):
            fie.force_tile(code)
            y = fie.conv3x3(x, wc, 512, gn_groups=32)
            assert "split-K" in hip.last_gemm_kernel(fie)
            assert rel_err(y.permute(0, 3, 1, 2), ref) < 3e-3, code
            assert getattr(y, "_gn_tag", None) is not None
            fie.force_tile(0)
            gn = fie.groupnorm(y, gam, bet, 32, 1e-5, True)              # consumes the epilogue's sums
            gref = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, eps=1e-5))
            assert rel_err(gn.permute(0, 3, 1, 2), gref) < 4e-3, code
            fie.force_tile(code)
            w4 = fie.pack_conv_up2x(wt.to(DEV))
            up = fie.conv_up2x(x, w4, 512)
            uref = F.conv2d(F.interpolate(x.float().cpu().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"), wt.float(), None, padding=1)
            assert rel_err(up.permute(0, 3, 1, 2), uref) < 4e-3, code
            x2, x3 = rnd(2 * 32 * 32, 128, seed=33).to(DEV), rnd(2 * 32 * 32, 64, seed=34).to(DEV)
            w1 = rnd(512, 192, seed=35, scale=192 ** -0.5)
            wplus = torch.cat([wc[:, :9 * 256], fie.pack_linear(w1.to(DEV))[:, :192]], 1).contiguous()
            yp = fie.conv3x3_plus(x, wplus, 512, x2, x3)
            pref = ref + (torch.cat([x2, x3], 1).float().cpu() @ w1.float().T).view(2, 32, 32, 512).permute(0, 3, 1, 2)
            assert rel_err(yp.permute(0, 3, 1, 2), pref) < 3e-3, code
        # no workspace bound: the same forced code runs unsplit
        fie.force_tile(30096)
        assert lib.fie_splitk_workspace(fie.h, None, 0) == 0
        fie._sk_bound = "unbound"
        saved, fie.splitk_bytes = fie.splitk_bytes, 0
        try:
            out = fie.gemm(a0, wp, n, bias=bias)
            assert "split-K" not in hip.last_gemm_kernel(fie) and rel_err(out, first[0].float()) < 1.5e-3
        finally:
            fie.splitk_bytes = saved
            fie._sk_bound = None
    finally:
        fie.force_tile(0)


@pytest.mark.parametrize("cout,code", [(128, 0), (256, 54), (512, 81), (128, 52), (256, 62), (128, 42), (512, 96), (256, 43)])
def test_groupnorm_statistics_from_the_producing_conv(fie, cout, code):
    """fie_gn_stats_target + fie_groupnorm_stats_nhwc_f16 (include/fie.h): the conv's epilogue leaves per-granule (sum, sum of squares) of its
    f16-rounded output; GroupNorm from those equals the three-kernel GroupNorm of the same tensor (which re-reads it) and torch's;
    4 / 8 / 16 channels per group (the VAE's 128 / 256 / 512-channel maps), two images, residual + bias in the epilogue, every tile
    family, a ragged last row tile (M = 2 * 2304 = 4608 is not a multiple of 192 / 256)."""
    from fie_amd import hip
    b, h, w, cin, groups = 2, 48, 48, 64, 32
    x = rnd(b, h, w, cin, seed=1).to(DEV)
    wc = fie.pack_conv3x3(rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5).to(DEV))
    bias, res = rnd(cout, seed=3).to(DEV), rnd(b, h, w, cout, seed=4).to(DEV)
    gamma, beta = (1 + 0.1 * rnd(cout, seed=5)).to(DEV), (0.1 * rnd(cout, seed=6)).to(DEV)
    fie.force_tile(code)
    try:
        y = fie.conv3x3(x, wc, cout, bias=bias, residual=res, gn_groups=groups)
    finally:
        fie.force_tile(0)
    assert y._gn_tag is not None
    fast = fie.groupnorm(y, gamma, beta, groups, 1e-6, True)
    slow = fie.groupnorm(y.clone(), gamma, beta, groups, 1e-6, True)           # the clone carries no tag: gn_partial / finalize / apply
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(y.float().permute(0, 3, 1, 2), groups, gamma.float(), beta.float(), 1e-6)).permute(0, 2, 3, 1)
    assert rel_err(fast, slow.float()) < 1e-3 and rel_err(fast, ref) < 4e-3
    # a producer in between invalidates the tag (its sums overwrote the buffer): the plain path runs and still agrees
    fie.conv3x3(x, wc, cout, gn_groups=groups)
    again = fie.groupnorm(y, gamma, beta, groups, 1e-6, True)
    assert torch.equal(again, slow)
    # GEMM producer (the VAE mid-block attention's output projection, + residual)
    a2d, wl = rnd(b * h * w, 64, seed=7).to(DEV), fie.pack_linear(rnd(cout, 64, seed=8, scale=0.125).to(DEV))
    o = fie.gemm(a2d, wl, cout, bias=bias, residual=res.view(b * h * w, cout), gn_stats=(h * w, groups))
    o4 = o.view(b, h, w, cout)
    o4._gn_tag = o._gn_tag
    assert o._gn_tag is not None
    assert rel_err(fie.groupnorm(o4, gamma, beta, groups, 1e-6, False), fie.groupnorm(o4.clone(), gamma, beta, groups, 1e-6, False).float()) < 1e-3


@pytest.mark.parametrize("cout,code", [(640, 0), (1280, 96), (640, 42), (1280, 30096), (640, 52)])
def test_groupnorm_statistics_for_20_and_40_channel_groups(fie, cout, code):
    """The UNet's group widths (640 / 1280 channels in 32 groups = 20 / 40 channels: not a whole number of a lane's 4 output columns per
    16-column fragment): the producer writes one slot per 4-channel QUAD (armed with N / 4 'groups'), the consumer sums the 5 / 10 quads of
    each real group.  Conv, conv + 1x1 side inputs and GEMM (+ residual) producers at 64x64 'latents' (4096 rows per image: below that the
    single-pass GroupNorm runs and nothing is armed), against the three-kernel GroupNorm of the same tensor and torch."""
    b, h, w, cin, groups = 2, 64, 64, 128, 32
    x = rnd(b, h, w, cin, seed=1).to(DEV)
    wc = fie.pack_conv3x3(rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5).to(DEV))
    bias = rnd(cout, seed=3).to(DEV)
    gamma, beta = (1 + 0.1 * rnd(cout, seed=5)).to(DEV), (0.1 * rnd(cout, seed=6)).to(DEV)
    fie.force_tile(code)
    try:
        y = fie.conv3x3(x, wc, cout, bias=bias, gn_groups=groups)
        a2d, wl = rnd(b * h * w, 128, seed=7).to(DEV), fie.pack_linear(rnd(cout, 128, seed=8, scale=128 ** -0.5).to(DEV))
        o = fie.gemm(a2d, wl, cout, bias=bias, residual=y.view(b * h * w, cout), gn_stats=(h * w, groups))
    finally:
        fie.force_tile(0)
    assert o._gn_tag is not None and o._gn_tag[7] == cout // 4
    o4 = o.view(b, h, w, cout)
    o4._gn_tag = o._gn_tag
    fast = fie.groupnorm(o4, gamma, beta, groups, 1e-5, True)
    slow = fie.groupnorm(o4.clone(), gamma, beta, groups, 1e-5, True)
    ref = F.silu(F.group_norm(o4.float().permute(0, 3, 1, 2), groups, gamma.float(), beta.float(), 1e-5)).permute(0, 2, 3, 1)
    assert rel_err(fast, slow.float()) < 1e-3 and rel_err(fast, ref) < 4e-3
    # the conv's own sums were overwritten by the GEMM producer: its GroupNorm takes the plain path and still agrees
    assert torch.equal(fie.groupnorm(y, gamma, beta, groups, 1e-5, True), fie.groupnorm(y.clone(), gamma, beta, groups, 1e-5, True))
    y2 = fie.conv3x3(x, wc, cout, bias=bias, gn_groups=groups)
    assert y2._gn_tag is not None and rel_err(fie.groupnorm(y2, gamma, beta, groups, 1e-5, False), fie.groupnorm(y2.clone(), gamma, beta, groups, 1e-5, False).float()) < 1e-3
    small = fie.conv3x3(x[:, :32, :32].contiguous(), wc, cout, gn_groups=groups)       # 1024 rows per image: single-pass GroupNorm, nothing armed
    assert small._gn_tag is None


def test_gn_stats_target_is_disarmed_by_a_failing_call(fie):
    """ADVICE r2 (gemm_conv.hip:831): the one-shot GroupNorm target is taken at the top of every GEMM / conv entry, so a call that fails
    its argument checks does not leave it armed for the next, unrelated launch."""
    from fie_amd import hip
    lib = hip.lib()
    buf = torch.full((1 << 16,), 7.0, device=DEV)
    a, w = rnd(256, 128, seed=1).to(DEV), fie.pack_linear(rnd(128, 128, seed=2, scale=128 ** -0.5).to(DEV))
    out = torch.empty(256, 128, device=DEV, dtype=torch.float16)
    assert lib.fie_gn_stats_target(fie.h, buf.data_ptr(), 64, 32) == 0
    rc = lib.fie_gemm_f16(fie.h, a.data_ptr(), 128, 128, None, 0, w.data_ptr(), w.stride(0), out.data_ptr(), 128, 256, 128, 124, None, None, 0, 0,
                          None, 0, 1.0, 0)                           # K = 124: not a multiple of 8 -> FIE_EINVAL
    assert rc != 0
    fie.gemm(a, w, 128, out=out)                                  # must NOT write GroupNorm sums anywhere
    torch.cuda.synchronize()
    assert bool((buf == 7.0).all())


@pytest.mark.parametrize("b,h,w,cin,cout,code", [(2, 16, 16, 128, 128, 0), (1, 24, 40, 64, 256, 54), (2, 32, 32, 256, 512, 81), (1, 20, 12, 128, 192, 42)])
def test_upsampling_conv_as_four_parity_convs(fie, b, h, w, cin, cout, code):
    """fie_conv_up2x_nhwc_f16 (include/fie.h): conv3x3(nearest-2x(x)) from four 2x2 convs with pre-summed taps equals the 9-tap kernel with the
    upsampling folded into its addressing (up to the f16 rounding of the summed weights) and torch; borders (zero padding of the upsampled
    image), bias + row bias + SiLU, non-square maps, every tile family; with GroupNorm sums from its epilogue the normalised result
    matches too."""
    x = rnd(b, h, w, cin, seed=1).to(DEV)
    w4d = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    bias, rb = rnd(cout, seed=3).to(DEV), rnd(b, cout, seed=4).to(DEV)
    wp, wp4 = fie.pack_conv3x3(w4d.to(DEV)), fie.pack_conv_up2x(w4d)
    from fie_amd import hip
    fie.force_tile(code)
    try:
        nine = fie.conv3x3(x, wp, cout, upsample=True, bias=bias, rowbias=rb, act=hip.ACT_SILU)
        four = fie.conv_up2x(x, wp4, cout, bias=bias, rowbias=rb, act=hip.ACT_SILU, gn_groups=32 if cout // 32 in (4, 8, 16) and (h * w) % 32 == 0 else None)
    finally:
        fie.force_tile(0)
    up = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(up, w4d.to(DEV).float(), bias.float(), padding=1) + rb.float()[:, :, None, None]
    ref = torch.nn.functional.silu(ref).permute(0, 2, 3, 1)
    assert four.shape == nine.shape == ref.shape
    assert rel_err(four, ref) < 4e-3 and rel_err(four, nine.float()) < 3e-3
    if four._gn_tag is not None:
        gamma, beta = (1 + 0.1 * rnd(cout, seed=5)).to(DEV), (0.1 * rnd(cout, seed=6)).to(DEV)
        assert rel_err(fie.groupnorm(four, gamma, beta, 32, 1e-6, True), fie.groupnorm(four.clone(), gamma, beta, 32, 1e-6, True).float()) < 1e-3


@pytest.mark.parametrize("b,hw,cin,c2,c3,cout,code", [(2, 16, 128, 64, 0, 128, 0), (2, 32, 128, 192, 64, 256, 54), (1, 24, 64, 128, 128, 320, 42), (2, 16, 256, 128, 0, 192, 96),
                                                    (2, 16, 64, 64, 64, 128, 43), (2, 16, 64, 64, 64, 128, 44), (2, 16, 64, 64, 64, 128, 46), (2, 16, 64, 64, 64, 128, 51),
                                                    (2, 16, 64, 64, 64, 128, 52), (2, 16, 64, 64, 64, 128, 95), (2, 16, 64, 64, 64, 128, 62), (2, 16, 64, 64, 64, 128, 61)])
def test_conv3x3_with_its_1x1_shortcut_in_one_gemm(fie, b, hw, cin, c2, c3, cout, code):
    """fie_conv3x3_plus_nhwc_f16 (include/fie.h): conv2(h) + conv_shortcut([x | skip]) of a resnet as one GEMM equals the conv with the shortcut
    GEMM's output as residual (the unfused route) and torch; one or two side inputs, a side input that is a column slice of a wider tensor
    (row stride > C2), bias + SiLU, ragged N, the ring tile families."""
    from fie_amd import hip
    h = rnd(b, hw, hw, cin, seed=1).to(DEV)
    wide = rnd(b * hw * hw, c2 + 64, seed=2).to(DEV)
    x2 = wide[:, 32:32 + c2]                                   # a view: ld2 = c2 + 64
    x3 = rnd(b * hw * hw, c3, seed=3).to(DEV) if c3 else None
    wc, wsc = rnd(cout, cin, 3, 3, seed=4, scale=(9 * cin) ** -0.5), rnd(cout, c2 + c3, seed=5, scale=(c2 + c3) ** -0.5)
    bc, bsc = rnd(cout, seed=6).to(DEV), rnd(cout, seed=7).to(DEV)
    wp_c, wp_s = fie.pack_conv3x3(wc.to(DEV)), fie.pack_linear(wsc.to(DEV))
    wp = torch.cat([wp_c[:, :9 * cin], wp_s[:, :c2 + c3]], 1).contiguous()
    fie.force_tile(code)
    try:
        fused = fie.conv3x3_plus(h, wp, cout, x2, x3, bias=bc + bsc, act=hip.ACT_SILU)
    finally:
        fie.force_tile(0)
    xcat = torch.cat([x2, x3], 1) if c3 else x2
    ref = torch.nn.functional.conv2d(h.float().permute(0, 3, 1, 2), wc.to(DEV).float(), bc.float(), padding=1).permute(0, 2, 3, 1)
    ref = torch.nn.functional.silu(ref + (xcat.float() @ wsc.to(DEV).float().T + bsc.float()).view(b, hw, hw, cout))
    assert rel_err(fused, ref) < 4e-3
    res = fie.gemm(x2, wp_s, cout, a2=x3, bias=bsc).view(b, hw, hw, cout)
    # the unfused route adds the residual AFTER the activation, so compare it without one
    plain = fie.conv3x3(h, wp_c, cout, bias=bc, residual=res)
    assert rel_err(fie.conv3x3_plus(h, wp, cout, x2, x3, bias=bc + bsc), plain.float()) < 3e-3


def test_time_embed_fused(fie):
    """K7 fused kernel against the unfused route it replaces (embeddings.py): sinusoid -> Linear -> SiLU -> Linear, + the
    text-time embedding, SiLU; SDXL dims (320 -> 1280 -> 1280) and the tiny stack's (64 -> 256), batch 1 / 2 / 4, t = 499 KAT
    of SURVEY A.1 on the way; 50 back-to-back launches on one workspace are bit-identical (the in-launch barrier between the
    layers resets its counters; a stale hidden vector would show)."""
    import math
    for c0, e in ((320, 1280), (64, 256)):
        ws = fie.time_embed_workspace(e)
        for b in (1, 2, 4):
            t = torch.tensor([499.0, 259.0, 999.0, 0.0])[:b]
            w1, b1 = rnd(e, c0, seed=1, scale=c0 ** -0.5), rnd(e, seed=2, scale=0.1)
            w2, b2 = rnd(e, e, seed=3, scale=e ** -0.5), rnd(e, seed=4, scale=0.1)
            add = rnd(b, e, seed=5)
            dev = [v.to(DEV) for v in (t, w1, b1, w2, b2)]
            out = fie.time_embed(*dev, ws, add=add.to(DEV))
            f = torch.exp(-math.log(10000.0) * torch.arange(c0 // 2, dtype=torch.float32) / (c0 // 2))
            x = torch.cat([torch.cos(t[:, None] * f), torch.sin(t[:, None] * f)], 1)
            if c0 == 320:
                assert torch.allclose(x[0, :3], torch.tensor([-0.87116218, 0.98838931, 0.19755381]), atol=1e-5)
            h = F.silu(x.half().float() @ w1.float().T + b1.float()).half().float()
            assert rel_err(out, F.silu(h @ w2.float().T + b2.float() + add.float())) < 3e-3
            assert rel_err(fie.time_embed(*dev, ws), F.silu(h @ w2.float().T + b2.float())) < 3e-3
            first = fie.time_embed(*dev, ws, add=add.to(DEV)).clone()
            for _ in range(50):
                assert torch.equal(fie.time_embed(*dev, ws, add=add.to(DEV)), first)
        assert int(ws[-16:-8].view(torch.int32).abs().sum()) == 0          # both barrier counters back at zero


@pytest.mark.parametrize("b,h,w,cin,cout,stride,pad_mode,act", [
    (1, 512, 1024, 64, 4, 1, 0, "none"),        # row-reuse form, 8 rows per wave (the decoder's conv_out shape class)
    (1, 256, 1024, 32, 8, 1, 0, "none"),        # one 32-channel chunk
    (1, 512, 1024, 8, 16, 1, 0, "silu"),        # general form, 8 rows per wave (conditioning embedding: 3 -> 16 on the 8-channel padded image)
    (2, 40, 24, 64, 12, 1, 0, "none"),          # row-reuse form, 2 rows per wave, ragged strip (24 = 16 + 8) and ragged row block
    (2, 128, 128, 320, 4, 1, 0, "none"),        # the UNet's conv_out
    (1, 50, 30, 16, 16, 2, 1, "silu"),          # general form: stride 2, asymmetric pad
    (1, 33, 17, 96, 8, 2, 0, "none"),           # stride 2 with Cin % 32 == 0 (general form: the reuse form is stride 1 only)
    (1, 19, 21, 24, 4, 1, 1, "none")])          # Cin % 32 != 0, a lane group straddles nothing (Cin % 8 == 0), K = 216 -> 7 steps, last one partly past K
def test_thin_conv(fie, b, h, w, cin, cout, stride, pad_mode, act):
    """3x3 convs with at most 16 output channels run on the direct-load strip kernel (csrc/conv_thin.hip, tile code 77) by rule: against an fp32
    conv2d, and against an im2col tile of the launch table (a forced code goes past the rule)."""
    from fie_amd import hip
    x = rnd(b, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=3)
    xi = x.float()
    if pad_mode == 1:
        xi = F.pad(xi, (0, 1, 0, 1))
    ref = F.conv2d(xi, wt.float(), bias.float(), stride=stride, padding=1 if pad_mode == 0 else 0)
    if act == "silu":
        ref = F.silu(ref)
    ref = ref * 0.5
    xd, wp, bd = x.permute(0, 2, 3, 1).contiguous().to(DEV), fie.pack_conv3x3(wt.to(DEV)), bias.to(DEV)
    kw = dict(stride=stride, pad_mode=pad_mode, bias=bd, scale=0.5, act=hip.ACT_SILU if act == "silu" else hip.ACT_NONE)
    by_rule = b * ref.shape[2] * ref.shape[3] >= 131072                # smaller maps keep their ring tiles by rule: the code is forced for them
    try:
        if not by_rule:
            fie.force_tile(77)
        out = fie.conv3x3(xd, wp, cout, **kw)
        assert "conv_thin_kernel" in hip.last_gemm_kernel(fie)
        assert torch.equal(out, fie.conv3x3(xd, wp, cout, **kw))       # deterministic
        o8 = fie.conv3x3(xd, wp, cout, ldc=8, **kw) if cout == 4 else None
    finally:
        fie.force_tile(0)
    assert out.shape[1:3] == ref.shape[2:]
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 3e-3
    try:
        fie.force_tile(2)                                               # a pinned code goes past the thin-conv rule (as an override would)
        old = fie.conv3x3(xd, wp, cout, **kw)
        assert "conv_thin_kernel" not in hip.last_gemm_kernel(fie)
    finally:
        fie.force_tile(0)
    assert rel_err(out, old) < 2e-3
    # padded output rows (the decoder writes 3 channels as 4): ldc > Cout leaves the pad column alone
    if o8 is not None:
        assert torch.equal(o8[..., :4], out) and float(o8[..., 4:].abs().max()) == 0.0
