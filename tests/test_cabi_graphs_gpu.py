"""The graph-level C-ABI forwards that ARE forwards (include/fie.h: fie_clip_text_forward_f16, fie_vae_encode_f16, fie_controlnet_forward_f16,
fie_unet_forward_f16, fie_vae_decode_f16; csrc/graphs.cpp -- SURVEY 8b): the five model calls of the pipeline call at
/root/reference/src/pipeline.py:261-272 walked in C++ on weights registered by their diffusers names.  Checked against the Python walks
(fie_amd/{clip,vae,nn}.py: CLIP and VAE bit for bit -- same kernels, same fusions, same order; UNet / ControlNet to rounding -- the Python walk
keeps the one-launch timestep embedding and the zero-conv adds for itself), against the CPU oracle on the tiny stack, inside a hipGraph, and
for their error behaviour (unregistered weight, short workspace)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()


@pytest.mark.parametrize("stack_name,lat", [("tiny", 16), ("ssd-1b", 128)])
def test_vae_decode_and_encode_walked_in_cpp(fie, stack_name, lat):
    from fie_amd import cabi, hip, stack, weights
    from fie_amd.vae import VAE
    cfgs = stack.stack_configs(stack_name, True)
    sd = weights.synth_state_dict(cfgs["vae"], seed=1236, device="cpu", dtype=torch.float16)
    vae = VAE(fie, cfgs["vae"], sd)
    g = torch.Generator().manual_seed(4)
    z = torch.zeros(1, lat, lat, 8, dtype=torch.float16)
    z[..., :4] = torch.randn(1, lat, lat, 4, generator=g).half()
    zd = z.to(fie.device)
    ref = vae.decode(zd).float()
    cabi.register_vae(vae, f"vae_{stack_name}.")       # a prefix of this test's own: the session's context also serves the product pipelines' walks
    out = cabi.vae_decode(vae, zd)
    assert out.shape == ref.shape
    err = _rel(out, ref)
    print(f"C++ decoder walk vs Python walk ({stack_name}, {lat}x{lat} latents): rel. max-abs error {err:.2e}")
    # since round 4 the C++ walk takes the Python walk's fusions (GroupNorm sums from the epilogues, conv2 + shortcut in one launch, parity
    # up-samplers): same kernels in the same order, so the same bits
    assert torch.equal(out, ref.half()) and out[..., 3].abs().max() == 0
    # encoder: pixels -> moments
    img = torch.zeros(1, lat * 8, lat * 8, 8, dtype=torch.float16)
    img[..., :3] = (torch.rand(1, lat * 8, lat * 8, 3, generator=g) * 2 - 1).half()
    imgd = img.to(fie.device)
    mref, _ = vae.encode_moments(imgd)
    mout = cabi.vae_encode(vae, imgd)
    err = _rel(mout, mref)
    print(f"C++ encoder walk vs Python walk ({stack_name}, {lat * 8}^2 pixels): rel. max-abs error {err:.2e}")
    assert mout.shape == mref.shape and torch.equal(mout, mref)
    if stack_name == "tiny":
        from oracle import nets
        with torch.no_grad():
            oref = nets.vae_decode({k: v.float() for k, v in sd.items()}, cfgs["vae"], z[..., :4].permute(0, 3, 1, 2).float())
        assert _rel(out[0, ..., :3].permute(2, 0, 1).cpu(), oref[0]) < 2e-2
        # capturable: the whole C++ walk inside a hipGraph, replayed on new latents
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())      # zd and the walks above were issued on the default stream: order the side stream behind them
        with torch.cuda.stream(s):
            static_z = zd.clone()
            cabi.vae_decode(vae, static_z)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                cap = cabi.vae_decode(vae, static_z)
            static_z.copy_(zd * 0.5)
            gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(cap, cabi.vae_decode(vae, zd * 0.5))
        # a short workspace and a weight that was never registered: FIE_EINVAL with the reason, nothing launched past it
        vc = cabi.vae_config(cfgs["vae"], lat, lat, f"vae_{stack_name}.")
        need = hip.lib().fie_vae_decode_workspace_bytes(ctypes.byref(vc), lat, lat)
        ws = torch.empty(need, device=fie.device, dtype=torch.uint8)
        with pytest.raises(hip.FieError, match="workspace too small"):
            hip._chk(hip.lib().fie_vae_decode_f16(fie.h, ctypes.byref(vc), zd.data_ptr(), out.data_ptr(), ws.data_ptr(), need // 4))
        hip._chk(hip.lib().fie_weights_clear_prefix(fie.h, f"vae_{stack_name}.".encode()))
        with pytest.raises(hip.FieError, match="post_quant_conv.weight"):
            cabi.vae_decode(vae, zd)


@pytest.mark.parametrize("stack_name", ["tiny", "ssd-1b"])
def test_clip_text_walked_in_cpp(fie, stack_name):
    """Both text encoders: penultimate hidden state and (with a projection) the pooled projection, C++ walk == Python walk bit for bit (same kernels,
    same order, no fusion differs); tiny also against the oracle."""
    from fie_amd import cabi, stack, weights
    from fie_amd.clip import ClipText, eos_positions
    cfgs = stack.stack_configs(stack_name, True)
    g = torch.Generator().manual_seed(11)
    for key, prefix in (("clip_l", "text_encoder."), ("clip_g", "text_encoder_2.")):
        cfg = cfgs[key]
        sd = weights.synth_state_dict(cfg, seed=77, device="cpu", dtype=torch.float16)
        clip = ClipText(fie, cfg, sd)
        ids = torch.randint(1000, 40000, (2, 77), generator=g)
        ids[:, 0] = 49406
        ids[0, 9:] = 49407
        ids[1, 30:] = 49407
        eos = eos_positions(ids, cfg["eos_token_id"])
        eos_rows = (torch.arange(2) * 77 + eos).to(fie.device, torch.int32)
        ids_d = ids.to(fie.device, torch.int32).contiguous()
        pen_ref, pooled_ref = clip(ids_d, eos_rows=eos_rows)
        cabi.register_clip(clip, prefix)
        pen, pooled = cabi.clip_forward(clip, prefix, ids_d, eos_rows)
        assert torch.equal(pen, pen_ref), key
        if cfg["projection_dim"]:
            assert torch.equal(pooled, pooled_ref), key
        else:
            assert pooled_ref is None
        if stack_name == "tiny":
            from oracle import nets
            with torch.no_grad():
                hs, opooled = nets.clip_text_forward({k: v.float() for k, v in sd.items()}, cfg, ids)
            assert _rel(pen.cpu().view(2, 77, -1), hs[-2]) < 2e-2
            if cfg["projection_dim"]:
                assert _rel(pooled.cpu(), opooled) < 2e-2


def _cond_inputs(cfgs, lat, seed, dev):
    g = torch.Generator().manual_seed(seed)
    cfg = cfgs["unet"]
    x = torch.zeros(2, lat, lat, 8, dtype=torch.float16)
    x[..., :4] = torch.randn(2, lat, lat, 4, generator=g).half()
    text = torch.randn(2 * 77, cfg["cross_attention_dim"], generator=g).half()
    pdim = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
    pooled = torch.randn(2, pdim, generator=g).half()
    cond = torch.zeros(2, lat * 8, lat * 8, 8, dtype=torch.float16)
    cond[..., :3] = (torch.rand(2, lat * 8, lat * 8, 1, generator=g) > 0.9).half()
    tid = torch.tensor([[lat * 8., lat * 8., 0, 0, lat * 8., lat * 8.]]).repeat(2, 1)
    t = torch.full((2, 1), 499.0)
    return [v.to(dev) for v in (x, text, pooled, cond, tid, t)]


@pytest.mark.parametrize("stack_name,lat", [("tiny", 16), ("ssd-1b", 128)])
def test_controlnet_and_unet_walked_in_cpp(fie, stack_name, lat):
    """ControlNetModel.forward and UNet2DConditionModel.forward (with the ControlNet's residuals) as C++ walks, against the Python walks of the
    product path: every down / mid residual and the predicted noise.  The Python walk adds `scale * zero_conv(skip)` into the UNet's skips inside the
    zero-conv epilogue; upstream (and the C++ entries) return the residuals and add them in the UNet -- compared as residuals here."""
    from fie_amd import cabi, hip, stack, weights
    from fie_amd.nn import ControlNet, UNet
    cfgs = stack.stack_configs(stack_name, True)
    dev = fie.device
    unet = UNet(fie, cfgs["unet"], weights.synth_state_dict(cfgs["unet"], seed=1234, device="cpu", dtype=torch.float16))
    cn = ControlNet(fie, cfgs["controlnet"], weights.synth_state_dict(cfgs["controlnet"], seed=1235, device="cpu", dtype=torch.float16))
    x, text, pooled, cond, tid, t = _cond_inputs(cfgs, lat, 5, dev)
    scale = 0.5
    # Python walks
    unet.begin_image(pooled, tid)
    cn.begin_image(pooled, tid)
    tb_u, tb_c = unet.time_rowbias(t), cn.time_rowbias(t)
    c_skips, c_mid = cn.encode_cond(x, cn.cond_embedding(cond), tb_c, text, 77)
    skips, mid = unet.encode(unet.conv_in(fie, x), tb_u, text, 77)
    zeros = [torch.zeros_like(s) for s in skips]
    res_ref, mid_ref = cn.add_residuals(c_skips, c_mid, scale, zeros, torch.zeros_like(mid))       # the bare residuals
    skips2, mid2 = cn.add_residuals(c_skips, c_mid, scale, skips, mid)
    eps_ref = unet.decode(mid2, skips2, tb_u, text, 77)
    # C++ walks
    cabi.register_unet(unet)
    cabi.register_controlnet(cn)
    downs, midr = cabi.controlnet_forward(cn, "controlnet.", x, t, text, pooled, tid, cond, scale)
    assert len(downs) == len(res_ref)
    worst = max(_rel(d, r) for d, r in zip(downs + [midr], res_ref + [mid_ref]))
    print(f"C++ ControlNet walk vs Python walk ({stack_name}, {lat}x{lat} latents): worst residual rel. max-abs error {worst:.2e}")
    assert worst < 1e-2
    eps = cabi.unet_forward(unet, "unet.", x, t, text, pooled, tid, downs, midr)
    err = _rel(eps, eps_ref)
    print(f"C++ UNet walk vs Python walk ({stack_name}, {lat}x{lat} latents): rel. max-abs error {err:.2e}")
    assert eps.shape == eps_ref.shape and err < 1e-2
    # without residuals: the plain UNet forward
    eps0 = cabi.unet_forward(unet, "unet.", x, t, text, pooled, tid)
    eps0_ref = unet.decode(mid, skips, tb_u, text, 77)
    assert _rel(eps0, eps0_ref) < 1e-2
    # step cache (include/fie.h: fie_step_cache_bind): the first forward after a bind fills it with the text K / V of every transformer block (ControlNet:
    # and the conditioning embedding), the next ones read it -- the same bits as computing them per call, at another timestep too; unbound again afterwards
    b_, lh_, lw_, _ = x.shape
    cabi.step_cache_begin(unet, "unet.", b_, lh_, lw_, 77, False)
    cabi.step_cache_begin(cn, "controlnet.", b_, lh_, lw_, 77, True)
    try:
        fill = cabi.unet_forward(unet, "unet.", x, t, text, pooled, tid)
        t2 = t * 0.5
        hit = cabi.unet_forward(unet, "unet.", x, t2, text, pooled, tid)
        dfill, mfill = cabi.controlnet_forward(cn, "controlnet.", x, t, text, pooled, tid, cond, scale)
        dhit, mhit = cabi.controlnet_forward(cn, "controlnet.", x, t2, text, pooled, tid, cond, scale)
    finally:
        cabi.step_cache_end(unet, "unet.")
        cabi.step_cache_end(cn, "controlnet.")
    assert torch.equal(fill, eps0) and all(torch.equal(a_, b_) for a_, b_ in zip(dfill + [mfill], downs + [midr]))
    assert torch.equal(hit, cabi.unet_forward(unet, "unet.", x, t2, text, pooled, tid))
    dref, mref2 = cabi.controlnet_forward(cn, "controlnet.", x, t2, text, pooled, tid, cond, scale)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(dhit + [mhit], dref + [mref2]))
    short = torch.empty(1024, device=x.device, dtype=torch.uint8)
    hip._chk(hip.lib().fie_step_cache_bind(fie.h, b"unet.", short.data_ptr(), short.numel()))
    try:
        with pytest.raises(hip.FieError, match="step cache too small"):
            cabi.unet_forward(unet, "unet.", x, t, text, pooled, tid)
    finally:
        cabi.step_cache_end(unet, "unet.")
    if stack_name == "tiny":
        from oracle import nets
        f32 = lambda sd: {k: v.float() for k, v in sd.items()}
        usd = f32(weights.synth_state_dict(cfgs["unet"], seed=1234, device="cpu", dtype=torch.float16))
        with torch.no_grad():
            o = nets.unet_forward(usd, cfgs["unet"], x[..., :4].permute(0, 3, 1, 2).float().cpu(), t.view(-1).cpu(), text.view(2, 77, -1).float().cpu(),
                                  pooled.float().cpu(), tid.cpu())
            csd = f32(weights.synth_state_dict(cfgs["controlnet"], seed=1235, device="cpu", dtype=torch.float16))
            od, om = nets.controlnet_forward(csd, cfgs["controlnet"], x[..., :4].permute(0, 3, 1, 2).float().cpu(), t.view(-1).cpu(),
                                             text.view(2, 77, -1).float().cpu(), cond[..., :3].permute(0, 3, 1, 2).float().cpu(), scale, pooled.float().cpu(), tid.cpu())
        assert _rel(eps0.permute(0, 3, 1, 2).cpu(), o) < 2e-2
        assert max(_rel(d.permute(0, 3, 1, 2).cpu(), r) for d, r in zip(downs + [midr], od + [om])) < 2e-2
        # capturable, deterministic
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())      # the inputs and eps0 were produced on the default stream
        with torch.cuda.stream(s):
            cabi.unet_forward(unet, "unet.", x, t, text, pooled, tid)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                cap = cabi.unet_forward(unet, "unet.", x, t, text, pooled, tid)
            gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(cap, eps0)
        uc = cabi.unet_config(cfgs["unet"], 2, lat, lat, 77)
        uc.head_dim = 40
        assert hip.lib().fie_unet_workspace_bytes(ctypes.byref(uc)) == -1
