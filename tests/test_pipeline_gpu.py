"""Stage-by-stage and end-to-end parity of the HIP hot path against the CPU fp32 oracle on identical seeds,
weights (fp16-rounded on both sides) and noise (drawn fp32 on a CPU generator: SURVEY 8a-RNG build rule).

Tolerances (fp16 storage, fp32 accumulation vs an fp32 reference): per-tensor max-abs error relative to the
tensor's max-abs <= 2e-2 for deep composite graphs; the north_star image gate is SSIM >= 0.99."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def synth_image(seed, size):
    """Low-frequency colour field + filled shapes, so Canny(100,200) finds real edges."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    img = np.stack([0.5 + 0.4 * np.sin(6.0 * xx + rng.uniform(0, 6)) * np.cos(4.0 * yy + rng.uniform(0, 6))
                    for _ in range(3)], axis=2)
    for _ in range(6):
        cx, cy, r = rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), rng.uniform(0.05, 0.2)
        mask = ((xx - cx) ** 2 + (yy - cy) ** 2) < r * r
        img[mask] = rng.uniform(0, 1, 3)
    x0, x1 = sorted(rng.integers(0, size, 2))
    y0, y1 = sorted(rng.integers(0, size, 2))
    img[y0:y1, x0:x1] = rng.uniform(0, 1, 3)
    return Image.fromarray((img.clip(0, 1) * 255).astype(np.uint8))


@pytest.fixture(scope="module", params=["tiny", "tiny-nomid"])
def rig(request, fie):
    from fie_amd import stack
    from fie_amd.pipe import HipImg2ImgPipeline
    cfgs, sds = stack.synthetic_stack(request.param, True, device="cpu", dtype=torch.float16)
    sds32 = {k: {n: v.float() for n, v in sd.items()} for k, sd in sds.items()}
    pipe = HipImg2ImgPipeline(fie, cfgs, sds, noise_dtype=torch.float32)
    return cfgs, sds32, pipe


def _ids(pipe, texts):
    return pipe.tok_l(texts), pipe.tok_g(texts)


def test_clip_text_parity(rig):
    from oracle import nets
    cfgs, sds32, pipe = rig
    il, ig = _ids(pipe, ["a [rusty] bicycle on the road", ""])
    for enc, cfg, sd, ids in ((pipe.clip_l, cfgs["clip_l"], sds32["clip_l"], il), (pipe.clip_g, cfgs["clip_g"], sds32["clip_g"], ig)):
        pen, pooled = enc(ids)
        hs, ref_pooled = nets.clip_text_forward(sd, cfg, ids)
        assert rel_err(pen.view(2, 77, -1), hs[-2]) < 1e-2
        if pooled is not None:
            assert rel_err(pooled, ref_pooled) < 1e-2


def test_vae_roundtrip_parity(rig, fie):
    from oracle import nets, pipeline as opipe
    cfgs, sds32, pipe = rig
    img = synth_image(3, 128)
    x = opipe.pil_to_float(img, True)
    mean, logvar = nets.vae_encode_moments(sds32["vae"], cfgs["vae"], x)
    u8 = torch.from_numpy(np.asarray(img)).cuda()
    mom, (lh, lw) = pipe.vae.encode_moments(fie.pixels_in(u8, True))
    ref = torch.cat([mean, logvar], 1)[0].permute(1, 2, 0).reshape(lh * lw, 8)
    assert rel_err(mom, ref) < 2e-2
    z = torch.zeros(1, lh, lw, 8, dtype=torch.float16)
    z[..., :4] = mean[0].permute(1, 2, 0).half()
    dec = pipe.vae.decode(z.cuda())
    ref_dec = nets.vae_decode(sds32["vae"], cfgs["vae"], mean.half().float())
    assert rel_err(dec[0, ..., :3].permute(2, 0, 1), ref_dec[0]) < 2e-2


def test_unet_controlnet_eval_parity(rig, fie):
    """One ControlNet + UNet evaluation with CFG batch 2 on shared inputs."""
    from oracle import nets
    cfgs, sds32, pipe = rig
    g = torch.Generator().manual_seed(5)
    lh = lw = 16
    lat = torch.randn(2, 4, lh, lw, generator=g).half().float()
    cond = (torch.rand(2, 3, lh * 8, lw * 8, generator=g) > 0.9).float()
    xd = cfgs["unet"]["cross_attention_dim"]
    text = torch.randn(2, 77, xd, generator=g).half().float()
    pdim = cfgs["unet"]["projection_class_embeddings_input_dim"] - 6 * cfgs["unet"]["addition_time_embed_dim"]
    pooled = torch.randn(2, pdim, generator=g).half().float()
    tid = torch.tensor([[128., 128., 0, 0, 128., 128.]]).repeat(2, 1)
    t = 499
    down, mid = nets.controlnet_forward(sds32["controlnet"], cfgs["controlnet"], lat, t, text, cond, 0.5, pooled, tid)
    ref = nets.unet_forward(sds32["unet"], cfgs["unet"], lat, t, text, pooled, tid, down, mid)

    dev = fie.device
    model_in = torch.zeros(2, lh, lw, 8, dtype=torch.float16, device=dev)
    model_in[..., :4] = lat.permute(0, 2, 3, 1).half().to(dev)
    cond8 = torch.zeros(2, lh * 8, lw * 8, 8, dtype=torch.float16, device=dev)
    cond8[..., :3] = cond.permute(0, 2, 3, 1).half().to(dev)
    text_d = text.reshape(2 * 77, xd).half().to(dev)
    pipe.unet.begin_image(pooled.half().to(dev), tid.to(dev))
    pipe.controlnet.begin_image(pooled.half().to(dev), tid.to(dev))
    cemb = pipe.controlnet.cond_embedding(cond8)
    t_dev = torch.full((2, 1), float(t), device=dev)
    tb_u, tb_c = pipe.unet.time_rowbias(t_dev), pipe.controlnet.time_rowbias(t_dev)
    skips, m = pipe.unet.encode(pipe.unet.conv_in(fie, model_in), tb_u, text_d, 77)
    c_skips, c_mid = pipe.controlnet.encode_cond(model_in, cemb, tb_c, text_d, 77)
    skips2, m2 = pipe.controlnet.add_residuals(c_skips, c_mid, 0.5, skips, m)
    eps = pipe.unet.decode(m2, skips2, tb_u, text_d, 77)
    assert rel_err(eps.permute(0, 3, 1, 2), ref) < 2e-2


@pytest.mark.parametrize("guidance,strength", [(1.5, 0.8), (1.0, 0.5), (2.0, 1.0)])
def test_full_pipeline_parity(rig, guidance, strength):
    from oracle import canny, metrics, pipeline as opipe
    cfgs, sds32, pipe = rig
    img = synth_image(11, 128)
    ctrl = Image.fromarray(canny.canny_rgb(np.asarray(img)))
    prompt, neg = "a [red] circle next to a square", ""
    out = pipe(prompt=prompt, negative_prompt=neg, image=img, control_image=ctrl, strength=strength,
               num_inference_steps=4, guidance_scale=guidance, controlnet_conditioning_scale=0.5,
               generator=torch.Generator("cpu").manual_seed(42)).images[0]
    tr = {}
    ref = opipe.run(sds32, cfgs, img, ctrl, _ids(pipe, [prompt]), _ids(pipe, [neg]), strength=strength,
                    num_inference_steps=4, guidance_scale=guidance, controlnet_conditioning_scale=0.5,
                    generator=torch.Generator("cpu").manual_seed(42), trace=tr)
    assert pipe.last_stats["unet_evals"] == len(tr["eps"]) == int(4 * strength)
    s = metrics.ssim(out, ref, size=None)
    diff = np.abs(np.asarray(out).astype(int) - ref.astype(int))
    print(f"ssim={s:.5f} max|du8|={diff.max()} mean|du8|={diff.mean():.4f}")
    assert s >= 0.99


def test_editor_api_surface(fie):
    from src.pipeline import FastEditor
    with pytest.raises(ValueError):
        FastEditor(model_name="nope")
    ed = FastEditor(model_name="tiny", enable_cpu_offload=False, use_full_controlnet=True)
    assert set(FastEditor.MODEL_CONFIGS) == {"sdxl", "ssd-1b"}
    ed.pipe.set_progress_bar_config(disable=True)
    img = synth_image(1, 96)
    edge = ed.preprocess_image(img)
    assert edge.size == img.size and set(np.unique(np.asarray(edge))) <= {0, 255}
    out = ed.edit(img, "a photo of a [cat]", seed=42, strength=0.5, guidance_scale=1.0)
    assert out.size == (1024, 1024) and out.mode == "RGB"
    out2 = ed.edit(img, "a photo of a [cat]", seed=42, strength=0.5, guidance_scale=1.0)
    assert np.array_equal(np.asarray(out), np.asarray(out2))         # same seed -> same image
    mem = ed.get_memory_usage()
    assert mem["allocated_gb"] > 0 and "reserved_gb" in mem
    with pytest.raises(ValueError):
        ed.edit(img, "x", strength=1.5)
    ed.clear_memory()


def test_graph_replay_matches_eager(rig):
    """The captured hipGraph replays the same launches: bit-identical output, also after swapping inputs."""
    from oracle import canny
    cfgs, sds32, pipe = rig
    outs = {}
    for mode in (False, True):
        pipe.use_graph = mode
        for seed in (21, 22):
            img = synth_image(seed, 128)
            ctrl = Image.fromarray(canny.canny_rgb(np.asarray(img)))
            outs[(mode, seed)] = np.asarray(pipe(prompt=f"a [green] shape {seed}", negative_prompt="", image=img,
                                                 control_image=ctrl, strength=0.8, num_inference_steps=4, guidance_scale=1.5,
                                                 controlnet_conditioning_scale=0.5,
                                                 generator=torch.Generator("cpu").manual_seed(seed)).images[0])
    pipe.use_graph = True
    assert np.array_equal(outs[(False, 21)], outs[(True, 21)]) and np.array_equal(outs[(False, 22)], outs[(True, 22)])
    assert not np.array_equal(outs[(True, 21)], outs[(True, 22)])


def test_run_batch_shard_end_to_end(fie, tmp_path, capsys):
    """run_batch's per-image loop on a synthetic PIE-Bench-shaped directory: outputs land under the same relative paths,
    --skip_existing makes reruns idempotent, a missing source / empty prompt / traversal path is counted, not fatal."""
    import json
    import run_batch
    from src.pipeline import FastEditor
    src, out = tmp_path / "src", tmp_path / "out"
    mapping = {}
    for i in range(3):
        rel = f"{i}_cat/{i:012d}.jpg"
        (src / f"{i}_cat").mkdir(parents=True)
        synth_image(i, 96).save(src / rel)
        mapping[f"{i:012d}"] = {"image_path": rel, "editing_prompt": f"a [blue] thing {i}", "editing_type_id": str(i)}
    mapping["nosrc"] = {"image_path": "9_cat/none.jpg", "editing_prompt": "x", "editing_type_id": "9"}
    mapping["noprompt"] = {"image_path": "0_cat/000000000000.jpg", "editing_prompt": "", "editing_type_id": "0"}
    mapping["evil"] = {"image_path": "../../etc/passwd", "editing_prompt": "x", "editing_type_id": "0"}
    args = run_batch.build_parser().parse_args(["--source_dir", str(src), "--output_dir", str(out), "--seed", "42",
                                                "--guidance", "1.0", "--strength", "0.5", "--skip_existing"])
    ed = FastEditor(model_name="tiny", enable_cpu_offload=False)
    edited = out / "e"
    entries = [(i, k, e) for i, (k, e) in enumerate(mapping.items())]
    r1 = run_batch.process_shard(ed, entries, args, str(edited), str(out / "c"))
    assert (r1["processed"], r1["skipped"], r1["failed"]) == (3, 1, 2)      # "noprompt" finds image 0's output: skipped
    assert [row["index"] for row in r1["rows"]] == [0, 1, 2] and r1["total_time"] > 0
    for i in range(3):
        im = Image.open(edited / f"{i}_cat/{i:012d}.jpg")
        assert im.size == (1024, 1024)
    r2 = run_batch.process_shard(ed, entries, args, str(edited), str(out / "c"))
    assert (r2["processed"], r2["skipped"], r2["failed"]) == (0, 4, 2)
    assert "Invalid path" in capsys.readouterr().out


def test_edit_with_unfinished_hysteresis_repeats_the_job(fie):
    """FastEditor.edit() launches the device Canny with a FIXED number of hysteresis rounds and issues the edit behind it without waiting for the flags
    (src/pipeline.py: edit, CANNY_ROUNDS; include/fie.h: fie_canny_rgb_device_begin_u8).  An image whose weak chain needs more passes than that -- a
    weak ramp edge across the whole width, seeded by one strong blob -- must still give exactly the result of the waiting path (preprocess_image() +
    the pipeline call): the flags say so when the result arrives, the remaining rounds run and the device job is repeated on the final edge map."""
    from PIL import Image
    from src.pipeline import FastEditor
    ed = FastEditor(model_name="tiny", enable_cpu_offload=False, use_full_controlnet=True)
    a = np.full((1024, 1024, 3), 100, np.uint8)
    a[512:, :, :] = 130                       # L1 gradient 120 along the row: weak everywhere (100 < 120 <= 200)
    a[512:, :8, :] = 200                      # one strong seed at the left end: the closure crosses 32 tiles
    img = Image.fromarray(a)
    kw = dict(strength=0.5, num_inference_steps=4, guidance_scale=1.5, controlnet_conditioning_scale=0.5)
    got = np.asarray(ed.edit(img, "a [red] line", seed=5, **kw))
    assert ed.pipe.repeated_jobs == 1 and ed.pipe.ctx.canny_more > 0
    ctrl = ed.preprocess_image(img)             # the waiting path: every round, then the call on the PIL edge map as the reference does
    assert np.asarray(ctrl)[500:524, 900:].max() == 255
    want = np.asarray(ed.pipe(prompt="a [red] line", negative_prompt="", image=img, control_image=ctrl,
                              generator=torch.Generator("cuda").manual_seed(5), **kw).images[0])
    assert np.array_equal(got, want)
    # an ordinary image: no repeat
    n = ed.pipe.repeated_jobs
    ed.edit(synth_image(3, 96), "a [toy]", seed=5, **kw)
    assert ed.pipe.repeated_jobs == n and ed.pipe.ctx.canny_more == 0


def test_two_edits_in_flight_match_serial(fie):
    """Worker threads on separate graph slots / streams produce exactly the serial results (--in_flight 2)."""
    from concurrent.futures import ThreadPoolExecutor
    from src.pipeline import FastEditor
    ed = FastEditor(model_name="tiny", enable_cpu_offload=False)
    imgs = [synth_image(30 + i, 96) for i in range(6)]
    serial = [np.asarray(ed.edit(im, f"a [toy] number {i}", seed=7, strength=0.5)) for i, im in enumerate(imgs)]
    ed.set_in_flight(2)
    ed.calibrate_in_flight(imgs[0], "a [toy] number 0", seed=7, strength=0.5)      # may re-draw the slot streams; results unchanged

    def work(slot):
        ed.worker_slot(slot)
        return [(i, np.asarray(ed.edit(imgs[i], f"a [toy] number {i}", seed=7, strength=0.5))) for i in range(slot, 6, 2)]

    with ThreadPoolExecutor(max_workers=2) as pool:
        got = dict(sum(pool.map(work, range(2)), []))
    for i in range(6):
        assert np.array_equal(got[i], serial[i]), i


def test_edit_batch_matches_serial_edits(fie):
    """BASELINE config "batch=8": n images through one UNet / ControlNet / CLIP call give the serial results (per-image
    generators, image-major rows); tolerance = fp16 tile / GroupNorm-chunk order effects only."""
    from src.pipeline import FastEditor
    from oracle import metrics
    ed = FastEditor(model_name="tiny", enable_cpu_offload=False)
    imgs = [synth_image(50 + i, 96) for i in range(3)]
    prompts = [f"a [toy] number {i}" for i in range(3)]
    serial = [np.asarray(ed.edit(im, p, seed=11, strength=0.5)) for im, p in zip(imgs, prompts)]
    batch = [np.asarray(o) for o in ed.edit_batch(imgs, prompts, seed=11, strength=0.5)]
    assert len(batch) == 3
    for a, b in zip(serial, batch):
        assert a.shape == b.shape == (1024, 1024, 3)
        assert np.abs(a.astype(int) - b.astype(int)).max() <= 2
        assert metrics.ssim(Image.fromarray(b), a, size=None) > 0.999
    # a different prompt order must permute the outputs, not mix them
    swapped = [np.asarray(o) for o in ed.edit_batch(imgs[::-1], prompts[::-1], seed=11, strength=0.5)]
    assert np.abs(swapped[0].astype(int) - batch[2].astype(int)).max() <= 2


def test_run_batch_in_flight_and_batched_match_serial(fie, tmp_path):
    """run_batch --in_flight 2 (worker threads) and --batch_size 2 (one device job per two images) write the same images as
    the serial loop, with the same counters."""
    import run_batch
    from src.pipeline import FastEditor
    src = tmp_path / "src"
    mapping = {}
    for i in range(5):
        rel = f"{i}_cat/{i:012d}.png"
        (src / f"{i}_cat").mkdir(parents=True)
        synth_image(70 + i, 96).save(src / rel)
        mapping[f"{i:012d}"] = {"image_path": rel, "editing_prompt": f"a [green] thing {i}", "editing_type_id": str(i)}
    mapping["nosrc"] = {"image_path": "9_cat/none.png", "editing_prompt": "x", "editing_type_id": "9"}
    entries = [(i, k, e) for i, (k, e) in enumerate(mapping.items())]
    ed = FastEditor(model_name="tiny", enable_cpu_offload=False)
    outs = {}
    for tag, extra in (("serial", []), ("threads", ["--in_flight", "2"]), ("batched", ["--batch_size", "2"])):
        out = tmp_path / tag
        args = run_batch.build_parser().parse_args(["--source_dir", str(src), "--output_dir", str(out), "--seed", "42",
                                                    "--strength", "0.5"] + extra)
        r = run_batch.process_shard(ed, entries, args, str(out / "e"), str(out / "c"))
        ed.set_in_flight(1)
        assert (r["processed"], r["skipped"], r["failed"]) == (5, 0, 1), tag
        assert [row["index"] for row in r["rows"]] == [0, 1, 2, 3, 4]
        outs[tag] = [np.asarray(Image.open(out / "e" / f"{i}_cat/{i:012d}.png")).astype(int) for i in range(5)]
    for i in range(5):
        assert np.array_equal(outs["threads"][i], outs["serial"][i])
        assert np.abs(outs["batched"][i] - outs["serial"][i]).max() <= 2


def test_forked_graph_budget(rig):
    """Keys beyond MAX_FORKED_GRAPHS are captured on one stream (same results): many forked graphs in one process end up
    sharing hardware queues and replay 50 % slower."""
    cfgs, sds32, pipe = rig
    img = synth_image(21, 128)
    ctrl = Image.fromarray(np.zeros((128, 128, 3), np.uint8))
    old = pipe.MAX_FORKED_GRAPHS, pipe._n_forked, pipe.use_graph
    try:
        pipe.use_graph = True
        pipe.MAX_FORKED_GRAPHS = pipe._n_forked + 1
        outs = []
        for g in (1.31, 1.32):           # two new keys: the first is captured forked, the second on one stream
            outs.append(np.asarray(pipe(prompt="a [toy]", image=img, control_image=ctrl, strength=0.5, guidance_scale=g,
                                        generator=torch.Generator("cpu").manual_seed(1)).images[0]))
        forks = sorted(k[1] for k in pipe._graphs if k[0][3] in (1.31, 1.32))
        assert forks == [False, True]
        assert np.abs(outs[0].astype(int) - outs[1].astype(int)).max() <= 8          # guidance differs by 0.01 only
    finally:
        pipe.MAX_FORKED_GRAPHS, pipe.use_graph = old[0], old[2]


def test_graph_cache_is_bounded_and_entries_keep_their_own_latents(rig):
    """ADVICE r1 (pipe.py:41): the hipGraph cache is capped (keys beyond the cap run eagerly, same result), and
    output_type='latent' returns the latents of the graph that was REPLAYED, not of the most recently captured one."""
    cfgs, sds32, pipe = rig
    img = synth_image(23, 128)
    ctrl = Image.fromarray(np.zeros((128, 128, 3), np.uint8))
    call = lambda g, ot="np": pipe(prompt="a [toy]", image=img, control_image=ctrl, strength=0.5, guidance_scale=g,
                                   generator=torch.Generator("cpu").manual_seed(3), output_type=ot).images[0]
    old = pipe.max_graphs, pipe.use_graph, pipe.eager_overflow
    try:
        pipe.use_graph = True
        lat_a = call(1.71, "latent").clone()           # captures key A
        lat_b = call(1.95, "latent").clone()           # captures key B (different latents)
        assert not torch.equal(lat_a, lat_b)
        assert torch.equal(call(1.71, "latent"), lat_a)           # replay of A returns A's latents although B was captured later
        assert pipe.last_stats["unet_evals"] == 2
        pipe.max_graphs = len(pipe._graphs)                      # cache full from here on
        n = len(pipe._graphs)
        eager = call(1.83)
        assert len(pipe._graphs) == n and pipe.eager_overflow == old[2] + 1
        pipe.max_graphs = n + 1
        graphed = call(1.83)
        assert len(pipe._graphs) == n + 1 and np.array_equal(eager, graphed)
    finally:
        pipe.max_graphs, pipe.use_graph = old[0], old[1]


def test_overflow_on_slot_1_beside_a_replaying_slot_0(rig):
    """ADVICE r2 (pipe.py:297): a full graph cache serves a call eagerly; on slot 1 that eager run must use slot 1's workspaces
    (GroupNorm scratch, split-K slabs, the timestep-embedding barrier counters are keyed by (stream, graph slot)) while slot 0's graph
    replays on its own stream.  Worker threads as in `run_batch.py --in_flight 2`; results must equal the graphed ones."""
    from concurrent.futures import ThreadPoolExecutor
    cfgs, sds32, pipe = rig
    img = synth_image(29, 128)
    ctrl = Image.fromarray(np.zeros((128, 128, 3), np.uint8))
    call = lambda g, slot: np.asarray(pipe(prompt="a [toy]", image=img, control_image=ctrl, strength=0.5, guidance_scale=g, slot=slot,
                                           generator=torch.Generator("cpu").manual_seed(3), output_type="np").images[0])
    old = pipe.max_graphs, pipe.use_graph, pipe.eager_overflow
    try:
        pipe.use_graph = True
        ref0, ref1 = call(1.62, 0), call(1.77, 0)                 # graphed on slot 0 (the second key only as the reference image)
        pipe.max_graphs = len(pipe._graphs)                       # full: slot 1 overflows into the eager path
        outs = {}

        def work(slot):
            for _ in range(3):
                outs.setdefault(slot, []).append(call(1.62, 0) if slot == 0 else call(1.77, 1))

        with ThreadPoolExecutor(max_workers=2) as pool:
            list(pool.map(work, range(2)))
        assert pipe.eager_overflow >= old[2] + 3 and pipe.ctx.ws_tag == 0
        assert all(np.array_equal(o, ref0) for o in outs[0]) and all(np.array_equal(o, ref1) for o in outs[1])
    finally:
        pipe.max_graphs, pipe.use_graph = old[0], old[1]


def test_device_error_word_reaches_the_host(fie):
    """include/fie.h FIE_DEVERR_*: a kernel that has to give up (the bounded spin of the timestep-embedding barrier) sets the context's
    device error word; the host reads it behind every edit's D2H copy and raises.  The word is set by hand here."""
    from fie_amd import hip
    fie.check_device_errors()
    fie._err[0] = 1
    with pytest.raises(hip.FieError, match="barrier timed out"):
        fie.check_device_errors()
    fie.check_device_errors()                                     # cleared


def test_editor_from_a_weights_directory_whose_topology_matches_no_preset(fie, tmp_path):
    """FastEditor(weights_dir=...) builds its graphs from the directory's config.json files (default ctor flags: ssd-1b would
    otherwise pick the guessed 'small' ControlNet preset) and matches the oracle run on the same files."""
    from test_oracle_cpu import _write_stack
    from fie_amd import stack, weights
    from oracle import canny, metrics, pipeline as opipe
    from src.pipeline import FastEditor
    base = stack.stack_configs("tiny", True)
    unet = dict(base["unet"], name="odd-unet", down_attn=((0, 0), (1, 2), (2, 1)), mid_attn=0, mid_resnets=1,
                up_attn=((1, 2, 1), (2, 1, 1), (0, 0, 0)))
    cn = dict(base["controlnet"], name="odd-cn", down_attn=((0, 0), (0, 0), (1, 1)), mid_attn=1,
              conditioning_embedding_out_channels=(8, 16, 24, 40))
    cfgs = dict(base, unet=unet, controlnet=cn, clip_g=dict(base["clip_g"], eos_token_id=2))
    sds = {k: weights.synth_state_dict(cfgs[k], seed=90 + i, dtype=torch.float16) for i, k in enumerate(stack.KEYS)}
    _write_stack(str(tmp_path), cfgs, sds)
    ed = FastEditor(model_name="ssd-1b", enable_cpu_offload=False, weights_dir=str(tmp_path), noise_dtype=torch.float32)
    assert ed.presets["unet"] == "unet/config.json" and ed.pipe.cfgs["unet"]["up_attn"] == unet["up_attn"]
    assert ed.pipe.cfgs["controlnet"]["conditioning_embedding_out_channels"] == (8, 16, 24, 40)
    img = synth_image(13, 128)
    ctrl = Image.fromarray(canny.canny_rgb(np.asarray(img)))
    pipe = ed.pipe
    prompt = "a [red] circle"
    out = pipe(prompt=prompt, negative_prompt="", image=img, control_image=ctrl, strength=0.8, num_inference_steps=4,
               guidance_scale=1.5, controlnet_conditioning_scale=0.5, generator=torch.Generator("cpu").manual_seed(42)).images[0]
    sds32 = {k: {n: v.float() for n, v in sd.items()} for k, sd in sds.items()}
    ref = opipe.run(sds32, pipe.cfgs, img, ctrl, _ids(pipe, [prompt]), _ids(pipe, [""]), strength=0.8, num_inference_steps=4,
                    guidance_scale=1.5, controlnet_conditioning_scale=0.5, generator=torch.Generator("cpu").manual_seed(42))
    assert metrics.ssim(out, ref, size=None) >= 0.99
