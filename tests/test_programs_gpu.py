"""Graph-level C-ABI entries (include/fie.h: fie_program_*, fie_unet_forward, fie_controlnet_forward, fie_vae_encode,
fie_vae_decode, fie_clip_text_forward -- SURVEY 8b).  A model forward is walked once while a launch program records it; the named
entry then re-issues the whole graph from C++.  Checked here: replaying after the CONTENTS of the static inputs changed equals an
eager evaluation on the new inputs, bit for bit (same kernels, same order); launch counts; error behaviour."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rig(fie):
    from fie_amd import stack
    from fie_amd.pipe import HipImg2ImgPipeline
    cfgs, sds = stack.synthetic_stack("tiny", True, device="cpu", dtype=torch.float16)
    return HipImg2ImgPipeline(fie, cfgs, sds, noise_dtype=torch.float32)


def _unet_inputs(pipe, seed):
    dev = pipe.ctx.device
    g = torch.Generator().manual_seed(seed)
    cfg = pipe.cfgs["unet"]
    x = torch.zeros(2, 16, 16, 8, dtype=torch.float16)
    x[..., :4] = torch.randn(2, 16, 16, 4, generator=g).half()
    text = torch.randn(2 * 77, cfg["cross_attention_dim"], generator=g).half()
    pdim = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
    pooled = torch.randn(2, pdim, generator=g).half()
    cond = torch.zeros(2, 128, 128, 8, dtype=torch.float16)
    cond[..., :3] = (torch.rand(2, 128, 128, 1, generator=g) > 0.9).half()
    return x.to(dev), text.to(dev), pooled.to(dev), cond.to(dev)


def test_unet_and_controlnet_forward_programs(rig, fie):
    from fie_amd import hip
    pipe = rig
    dev = fie.device
    x, text, pooled, cond = _unet_inputs(pipe, 0)
    tid = torch.tensor([[128., 128., 0, 0, 128., 128.]]).repeat(2, 1).to(dev)
    t_dev = torch.full((2, 1), 499.0, device=dev)
    pipe.unet.begin_image(pooled, tid)
    pipe.controlnet.begin_image(pooled, tid)
    tb_u, tb_c = pipe.unet.time_rowbias(t_dev), pipe.controlnet.time_rowbias(t_dev)
    cemb = pipe.controlnet.cond_embedding(cond)

    def controlnet():
        return pipe.controlnet.encode_cond(x, cemb, tb_c, text, 77)

    def unet(c_skips, c_mid):
        skips, mid = pipe.unet.encode(pipe.unet.conv_in(fie, x), tb_u, text, 77)
        skips, mid = pipe.controlnet.add_residuals(c_skips, c_mid, 0.5, skips, mid)
        return pipe.unet.decode(mid, skips, tb_u, text, 77)

    with fie.record() as pc:
        c_skips, c_mid = controlnet()
    with fie.record() as pu:
        eps = unet(c_skips, c_mid)
    assert len(pc) > 50 and len(pu) > 100 and len(pu.keep) > 50
    pc.register("controlnet_forward")
    pu.register("unet_forward")
    first = eps.clone()
    # new CONTENTS in the same static buffers
    x2, text2, _, _ = _unet_inputs(pipe, 1)
    x.copy_(x2)
    text.copy_(text2)
    for t in list(pipe.unet.transformers()) + list(pipe.controlnet.transformers()):
        t.reset()                                   # eager reference below recomputes the text K/V, as the programs do
    fie.run_named("controlnet_forward")
    fie.run_named("unet_forward")
    torch.cuda.synchronize()
    replayed = eps.clone()
    ref = unet(*controlnet())
    assert not torch.equal(replayed, first)
    assert torch.equal(replayed, ref), f"replay differs from the eager walk: max |d| = {(replayed.float() - ref.float()).abs().max().item()}"
    pu.close()
    with pytest.raises(hip.FieError, match="no program registered"):
        fie.run_named("unet_forward")
    pc.close()


def test_vae_and_clip_programs(rig, fie):
    pipe = rig
    dev = fie.device
    g = torch.Generator().manual_seed(3)
    z = torch.zeros(1, 16, 16, 8, dtype=torch.float16)
    z[..., :4] = torch.randn(1, 16, 16, 4, generator=g).half()
    z = z.to(dev)
    img = (torch.rand(1, 128, 128, 8, generator=g) * 2 - 1).half().to(dev)
    ids = pipe.tok_g(["a [red] cube", ""]).to(dev, torch.int32)
    eos = torch.tensor([3, 78], device=dev, dtype=torch.int32)      # rows of the pooled tokens in the [B*77, C] state
    with fie.record() as pd:
        dec = pipe.vae.decode(z)
    with fie.record() as pe:
        mom, _ = pipe.vae.encode_moments(img)
    with fie.record() as pt:
        pen, pooled = pipe.clip_g(ids, eos_rows=eos)
    pd.register("vae_decode")
    pe.register("vae_encode")
    pt.register("clip_text_forward")
    z.copy_(torch.roll(z, 3, dims=1))
    img.copy_(torch.roll(img, 5, dims=2))
    ids.copy_(pipe.tok_g(["a [blue] sphere on a table", "x"]).to(dev, torch.int32))
    for name in ("vae_decode", "vae_encode", "clip_text_forward"):
        fie.run_named(name)
    torch.cuda.synchronize()
    got = dec.clone(), mom.clone(), pen.clone(), pooled.clone()
    ref_pen, ref_pooled = pipe.clip_g(ids, eos_rows=eos)
    assert torch.equal(got[0], pipe.vae.decode(z)) and torch.equal(got[1], pipe.vae.encode_moments(img)[0])
    assert torch.equal(got[2], ref_pen) and torch.equal(got[3], ref_pooled)
    assert len(pd) > 60 and len(pt) > 10
    for p in (pd, pe, pt):
        p.close()


def test_program_inside_a_hipgraph(rig, fie):
    """A program is a list of ordinary launches: replaying it under stream capture puts the whole forward into a hipGraph."""
    pipe = rig
    z = torch.zeros(1, 16, 16, 8, dtype=torch.float16, device=fie.device)
    z[..., :4] = 0.3
    with fie.record() as pd:
        dec = pipe.vae.decode(z)
    ref = dec.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())          # z, dec and ref were produced on the default stream (include/fie.h, ORDERING CONTRACT (3))
    with torch.cuda.stream(s):
        pd.run()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            pd.run()
        dec.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(dec, ref)
    assert fie.splitk_counters_clear(), "a split-K arrival counter was left non-zero"
    pd.close()


def test_program_is_ordered_against_itself_across_streams(rig, fie):
    """ORDERING CONTRACT (1) of include/fie.h: a program re-run on ANOTHER stream without any caller-side ordering still waits for its previous
    pass (the library records an event behind every pass), and a program recorded with split-K launches owns its workspace, so eager split
    GEMMs on the recording stream may run beside it.  Forced split-K (tile override) so that the hazard of GPUTEST r03 is really present:
    unordered, two in-flight copies of one split launch interleave on one arrival counter."""
    from fie_amd import hip
    dev = fie.device
    g = torch.Generator().manual_seed(11)
    m, n, k = 256, 256, 4096
    a = torch.randn(m, k, generator=g).half().to(dev)
    w = fie.pack_linear((torch.randn(n, k, generator=g) * k ** -0.5).half())
    fie.tile_override(f"0,{m},{n},{k}=30051")            # split-K 3 on the 128x128 ring tile: 4 tiles x 3 slices
    prog = None
    try:
        with fie.record() as prog:
            outs = [fie.gemm(a, w, n) for _ in range(8)]
        assert "split-K 3" in hip.last_gemm_kernel(fie)
        ref = [o.clone() for o in outs]
        torch.cuda.synchronize()
        side = [torch.cuda.Stream() for _ in range(3)]
        for rep in range(6):                              # the same program hopping streams back to back, eager split GEMMs beside it
            with torch.cuda.stream(side[rep % 3]):
                prog.run()
            eager = fie.gemm(a, w, n)                     # default stream: its own workspace, not the program's
        torch.cuda.synchronize()
        assert all(torch.equal(o, r) for o, r in zip(outs, ref)) and torch.equal(eager, ref[0])
        assert fie.splitk_counters_clear(), "a split-K arrival counter was left non-zero"
    finally:
        fie.tile_override(None)
        if prog is not None:
            prog.close()
