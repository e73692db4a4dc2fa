#!/usr/bin/env python3
"""Headline benchmark: PIE-Bench images/sec @1024^2, SSD-1B fp16, nominal 4-step LCM img2img with Canny ControlNet
(BASELINE.json metric; workload = configs[1]).  One "step" = one full `FastEditor.edit()` of one synthetic
PIE-Bench-shaped item, PIL in -> PIL out, timed exactly where /root/reference/run_batch.py:208-221 puts its timer:
LANCZOS 1024^2 + Canny + tokenise + RNG + H2D + (CLIP x2, VAE encode, evals x (ControlNet + UNet), CFG + LCM steps, VAE
decode, u8) + D2H.  `value` = K serial edits of K different items / wall time (one edit at a time, as the reference runs).
The device-resident hipGraph replay rate, two edits in flight and batch-N are reported as named extra keys, never as `value`.
Image-parallel over N GPUs: every rank edits its own items with a full replica; no collective inside the timed region
except the bracketing barriers.

    python bench.py [--gpus N] [--steps K] [--warmup W]        # N > 1 without a launcher: spawns its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (dominant kernel + UNet forward, fp16 MFMA
bound, HIP-event timed) and `cpu_baseline` (oracle/ fp32 restatement timed on this host, rank 0, N=1 only)."""
import argparse
import csv
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")    # before HIP initialises; see fie_amd.py
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_DENSE_TFLOPS = 2500.0      # MI355X_MICROARCH.md: Peak BF16/FP16 MFMA ~2.5 PF dense


def synth_item_image(i, size=512):
    """PIE-Bench-shaped synthetic source image: low-frequency colour fields + filled shapes."""
    from PIL import Image
    rng = np.random.default_rng(i)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    img = np.stack([0.5 + 0.4 * np.sin(rng.uniform(2, 9) * xx + rng.uniform(0, 6)) * np.cos(rng.uniform(2, 9) * yy + rng.uniform(0, 6))
                    for _ in range(3)], axis=2)
    for _ in range(8):
        cx, cy, r = rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), rng.uniform(0.04, 0.2)
        img[((xx - cx) ** 2 + (yy - cy) ** 2) < r * r] = rng.uniform(0, 1, 3)
    for _ in range(3):
        x0, x1 = sorted(rng.integers(0, size, 2))
        y0, y1 = sorted(rng.integers(0, size, 2))
        img[y0:y1, x0:x1] = rng.uniform(0, 1, 3)
    return Image.fromarray((img.clip(0, 1) * 255).astype(np.uint8))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Threads the CPU baseline may use: the process's CPU share (affinity / cgroup quota), capped at 16 (a one-GPU
    box's share) -- os.cpu_count() reports the whole host and oversubscribes by an order of magnitude."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def load_items():
    with open(os.path.join(ROOT, "tests", "golden", "pie_bench_items.csv")) as f:
        return list(csv.DictReader(f))


def cpu_baseline(editor, cfgs, job_args, evals, nb, strength, guidance):
    """ONE whole edit of the benchmark configuration through the CPU fp32 oracle (oracle.pipeline.run: CLIP x2 at batch nb, VAE
    encode, evals x (ControlNet + UNet) at CFG batch nb, LCM steps, VAE decode), timed end to end on this host's cores: the same
    workload as one timed GPU step, not an assembly of parts."""
    from oracle import pipeline as opipe
    from fie_amd import weights
    cores = host_cores()
    torch.set_num_threads(cores)
    seeds = {"unet": 0, "controlnet": 1, "vae": 2, "clip_l": 3, "clip_g": 4}
    # same synthetic generator as the product, fp16-rounded like the device copy (values do not affect timing)
    sds = {}
    for k, s in seeds.items():
        log(f"cpu_baseline: generating {k} weights on the host")
        sds[k] = {n: v.float() for n, v in weights.synth_state_dict(cfgs[k], seed=1234 + s, dtype=torch.float16).items()}
    pipe = editor.pipe
    img, ctrl, prompt = job_args
    ids = lambda texts: (pipe.tok_l(texts), pipe.tok_g(texts))
    log(f"cpu_baseline: weights ready, timing one oracle edit on {cores} threads")
    tr = {}
    t0 = time.time()
    out = opipe.run(sds, cfgs, img, ctrl, ids([prompt]), ids([""]), strength=strength, num_inference_steps=4, guidance_scale=guidance,
                    controlnet_conditioning_scale=0.5, generator=torch.Generator("cpu").manual_seed(42), trace=tr)
    dt = time.time() - t0
    assert out.shape == (1024, 1024, 3) and len(tr["eps"]) == evals
    log(f"cpu_baseline: one edit {dt:.1f}s")
    return {"value": 1.0 / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": (f"oracle/ fp32 torch-CPU restatement (oracle.pipeline.run), ONE whole 1024^2 edit of this configuration timed end to end: "
                       f"CLIPx2 at batch {nb}, VAE encode, {evals} x (ControlNet + UNet) at CFG batch {nb}, LCM steps, VAE decode = {dt:.1f}s")}


def _graph_ms(fn, reps=5):
    """Device time of `fn`'s launches captured as ONE hipGraph on the current stream and replayed (min of `reps`): what the production
    path (hipGraph replay) spends on them, without the host's per-launch cost that an eager pass adds to chains of small kernels."""
    fn()                                                # eager warm-up: lazy workspaces; tiles were tuned by the edits before
    torch.cuda.synchronize()
    from fie_amd import hip
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=hip.context(torch.cuda.current_device()).capture_stream()):   # split-K workspace bound before the capture
        out = fn()
    g.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, out, g


def time_unet_forward(pipe, job, iters=5):
    """Device time of one UNet forward (encode + decode, CFG batch) alone on one stream, replayed from a hipGraph."""
    ctx, dev = pipe.ctx, pipe.ctx.device
    h, w = job["hw"]
    nb = job["nb"]
    lh, lw = h // 8, w // 8
    xd = pipe.cfgs["unet"]["cross_attention_dim"]
    pdim = pipe.cfgs["unet"]["projection_class_embeddings_input_dim"] - 6 * pipe.cfgs["unet"]["addition_time_embed_dim"]
    g = torch.Generator(device=dev).manual_seed(0)
    text = torch.randn((nb * 77, xd), generator=g, device=dev, dtype=torch.float16)
    pooled = torch.randn((nb, pdim), generator=g, device=dev, dtype=torch.float16)
    x = torch.zeros((nb, lh, lw, 8), device=dev, dtype=torch.float16)
    x[..., :4] = torch.randn((nb, lh, lw, 4), generator=g, device=dev, dtype=torch.float16)
    pipe.unet.begin_image(pooled, job["time_ids"])
    tb = pipe.unet.time_rowbias(job["t_dev"][0])

    def fwd():
        skips, mid = pipe.unet.encode(pipe.unet.conv_in(ctx, x), tb, text, 77)
        return pipe.unet.decode(mid, skips, tb, text, 77)

    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        ms, _, _ = _graph_ms(fwd, iters)
    return ms


def stage_graph_ms(pipe, job, fl, reps=5):
    """Every stage of one edit captured as its OWN single-stream hipGraph and replayed: device ms and TFLOP/s per stage (the
    production graph additionally overlaps CLIP with the VAE encode and the UNet encoder with the ControlNet trunk, so the whole
    edit is shorter than the sum).  Mirrors pipe.run_device() for one image."""
    ctx, dev = pipe.ctx, pipe.ctx.device
    h, w = job["hw"]
    nb, steps = job["nb"], job["steps"]
    lh, lw = h // 8, w // 8
    hw = lh * lw
    sf = pipe.cfgs["vae"]["scaling_factor"]
    st0 = steps[0]
    out, keep = {}, []
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        def clip():
            pl, _ = pipe.clip_l(job["ids_l"])
            pg, pooled = pipe.clip_g(job["ids_g"], eos_rows=job["eos_rows"])
            return torch.cat([pl, pg], dim=1), pooled
        ms, (text, pooled), g = _graph_ms(clip, reps); keep.append(g)
        out["clip"] = (ms, nb * fl["clip"])
        latents = torch.empty((1, hw, 4), device=dev, dtype=torch.float32)
        model_in = torch.empty((nb, lh, lw, 8), device=dev, dtype=ctx.dtype)

        def enc():
            x_img = ctx.pixels_in(job["img_u8"], True)
            cond = ctx.pixels_in(job["ctl_u8"], False)
            moments, _ = pipe.vae.encode_moments(x_img)
            ctx.latent_prep(moments, job["noises"][0], job["noises"][1], hw, sf, st0["sqrt_ab"], st0["sqrt_1mab"], latents[0], model_in)
            return cond
        ms, cond, g = _graph_ms(enc, reps); keep.append(g)
        out["vae_encode"] = (ms, fl["vae_encode"])
        pipe.unet.begin_image(pooled, job["time_ids"])
        pipe.controlnet.begin_image(pooled, job["time_ids"])
        cond_emb = pipe.controlnet.cond_embedding(cond)
        if nb > 1:
            cond_emb = cond_emb.repeat_interleave(nb, dim=0)
        t_dev = job["t_dev"][0]
        tb_u, tb_c = pipe.unet.time_rowbias(t_dev), pipe.controlnet.time_rowbias(t_dev)

        def cn():
            return pipe.controlnet.encode_cond(model_in, cond_emb, tb_c, text, 77)
        ms, (c_skips, c_mid), g = _graph_ms(cn, reps); keep.append(g)
        out["controlnet_trunk (per eval)"] = (ms, nb * fl["controlnet"])

        def uenc():
            return pipe.unet.encode(pipe.unet.conv_in(ctx, model_in), tb_u, text, 77)
        ms, (skips, mid), g = _graph_ms(uenc, reps); keep.append(g)

        def udec():
            sk, m = pipe.controlnet.add_residuals(c_skips, c_mid, job["cn_scale"], skips, mid)
            return pipe.unet.decode(m, sk, tb_u, text, 77)
        ms2, eps, g = _graph_ms(udec, reps); keep.append(g)
        out["unet_encoder (per eval)"] = (ms, None)
        out["unet_decoder + zero-conv adds (per eval)"] = (ms2, None)
        out["unet (per eval)"] = (ms + ms2, nb * fl["unet"])
        decode_in = torch.zeros((1, lh, lw, 8), device=dev, dtype=ctx.dtype)
        decode_in[..., :4] = (latents.view(1, lh, lw, 4) / sf).to(ctx.dtype)

        def dec():
            return ctx.pixels_out(pipe.vae.decode(decode_in))
        ms, _, g = _graph_ms(dec, reps); keep.append(g)
        out["vae_decode"] = (ms, fl["vae_decode_executed"] if "vae_decode_executed" in fl else fl["vae_decode"])
    torch.cuda.synchronize()
    return {k: {"ms": round(ms, 3), **({"tflops": round(f / (ms * 1e-3) / 1e12, 1), "frac_of_mfma_peak": round(f / (ms * 1e-3) / 1e12 / PEAK_F16_DENSE_TFLOPS, 4)} if f else {})}
            for k, (ms, f) in out.items()}


def pmc_traffic(kernel, m, n, k, path=None):
    """(MB per launch, source CSVs) from the PMC record of `kernel` on shape (m, n, k), or (None, None): the record file holds one
    record per tile code the autotuner may pick (tools/collect_dominant_pmc.sh); the run's kernel name carries the code it used."""
    try:
        with open(path or os.path.join(ROOT, "profiles", "dominant_kernel_pmc.json")) as f:
            rec = json.load(f)
        if "fp8" in kernel:
            raise KeyError("the PMC records are for the f16-weight kernels")
        if "records" in rec:
            code = kernel.split("tile code ")[-1].split(")")[0]
            rec = dict(rec["records"][code], shape=rec["shape"])
        if [rec["shape"]["M"], rec["shape"]["N"], rec["shape"]["K"]] != [m, n, k]:
            return None, None
        # gfx950: FETCH_SIZE counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM) -> 2 x FETCH + WRITE
        return round((2 * rec["fetch_size_kib"] + rec["write_size_kib"]) * 1024 / 1e6, 1), rec.get("source")
    except (OSError, KeyError, ValueError):
        return None, None


def class_table(path=None):
    """The "By class" rows of the committed per-shape in-situ profile (tools/shape_profile.py under rocprofv3 --kernel-trace: one eager single-stream
    edit of this workload, launch log paired with the trace): launches, kernel ms and aggregate rate per kernel class, largest first.  Read from
    profiles/, never measured here (a kernel trace cannot be taken inside the bench); the source file is named beside it."""
    path = path or os.path.join(ROOT, "profiles", "r04_per_shape_roofline.md")
    rows = []
    try:
        with open(path) as f:
            lines = f.read().split("\n")
        i = lines.index("## By class")
        for l in lines[i + 4:]:
            c = [x.strip() for x in l.strip().strip("|").split("|")]
            if len(c) < 5 or not c[1].isdigit():
                break
            if c[0].startswith("_Z"):
                continue
            rate = c[4].split()
            frac = None
            if len(rate) == 2 and rate[1] in ("TF/s", "GB/s"):
                frac = round(float(rate[0]) / (PEAK_F16_DENSE_TFLOPS if rate[1] == "TF/s" else 8000.0), 4)
            rows.append({"class": c[0], "launches": int(c[1]), "ms": float(c[2]), "pct": float(c[3]), "rate": c[4], "frac_of_bound": frac})
    except (OSError, ValueError, IndexError):
        return None, None
    return rows[:8], os.path.relpath(path, ROOT)


def time_rotating_gemm(pipe, m, n, k, geglu, residual, iters=24):
    """One GEMM problem of the UNet as it runs INSIDE the network: the weight matrix rotates over enough copies (> 256 MB) that every launch streams
    it from HBM, not from the Infinity Cache (a UNet evaluation streams 2.6 GB of weights: no layer finds its own in cache), activations warm.
    Average launch duration by HIP events on the launch stream; algorithmic FLOPs = 2 M N K."""
    from fie_amd import hip
    ctx, dev = pipe.ctx, pipe.ctx.device
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn((m, k), generator=g, device=dev, dtype=torch.float16)
    copies = max(2, int(300e6 / (n * k * 2)) + 1)
    w8, ctx.w8 = ctx.w8, getattr(pipe, "weight_dtype", "f16") == "f8e4m3"       # the fp8 configuration times its own kernel
    try:
        ws = [ctx.pack_linear(torch.randn((n, k), generator=g, device=dev, dtype=torch.float16) * k ** -0.5, geglu=geglu) for _ in range(copies)]
    finally:
        ctx.w8 = w8
    bias = torch.randn((n,), generator=g, device=dev, dtype=torch.float16)
    a8w8 = isinstance(ws[0], hip.W8) and ctx.a8 and ws[0].stride(0) % 128 == 0       # fp8 configuration: e4m3 activations in (and e4m3 GEGLU out), as in the network
    if a8w8:
        a = ctx.quantize_f8(a)
    nout = n // 2 if geglu else n
    out8 = a8w8 and geglu
    out = torch.empty((m, nout), device=dev, dtype=torch.uint8 if out8 else torch.float16)
    res = torch.randn((m, nout), generator=g, device=dev, dtype=torch.float16) if residual else None
    act = hip.ACT_GEGLU if geglu else hip.ACT_NONE

    def launch(w):
        ctx.gemm(a, w, n, out=out, bias=bias, act=act, residual=res, out_f8=out8)
    for w in ws[:3]:
        launch(w)
    kernel = hip.last_gemm_kernel()
    iters = max(iters, copies)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        launch(ws[i % copies])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tf = 2.0 * m * n * k / (us * 1e-6) / 1e12
    return {"kernel": kernel, "shape": {"M": m, "N": n, "K": k}, "avg_us": round(us, 2), "launches": iters, "weight_copies": copies,
            "achieved": round(tf, 1), "frac": round(tf / PEAK_F16_DENSE_TFLOPS, 4), "algorithmic_gflop_per_launch": round(2.0 * m * n * k / 1e9, 2),
            "algorithmic_mb_per_launch": round((m * k + n * k + m * nout) * 2 / 1e6, 1)}


def time_dominant_kernel(pipe, nb):
    """The problem with the largest share of an edit's kernel time (top row of profiles/r04_per_shape_roofline.md): the GEMM of the UNet's
    32x32-latent FF1 projection (M = nb * 1024 tokens, N = 10240, K = 1280, bias + GEGLU epilogue), timed as it runs in the network (cold weights:
    time_rotating_gemm).  Beside it the largest problem of the kernel FAMILY with the largest share (the M = 2048 x N = 1280 projections on the small
    ring tiles): the FF2 projection, K = 5120, + residual.  `traffic` is NOT measured here (PMC counters cannot be collected inside the bench): it is
    read from profiles/dominant_kernel_pmc.json, which names the rocprofv3 --pmc CSVs it was derived from, and is reported only when that record is
    for this shape and the tile the run used."""
    m = nb * 1024
    ff1 = time_rotating_gemm(pipe, m, 10240, 1280, geglu=True, residual=False)
    ff2 = time_rotating_gemm(pipe, m, 1280, 5120, geglu=False, residual=True)
    ff1["kernel"] += " (FF1 GEGLU projection, 32x32 latents)"
    ff2["kernel"] += " (FF2 projection + residual, 32x32 latents)"
    ff1["traffic_mb_per_launch"], ff1["traffic_source"] = pmc_traffic(ff1["kernel"], m, 10240, 1280)
    return ff1, ff2


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU, and exit with the launcher's code.
    On a box with fewer than N GPUs this is a rehearsal: gloo backend, ranks folded onto the available devices (at most 6
    processes may share one card on the pool's boxes)."""
    import socket
    import subprocess
    ngpu = torch.cuda.device_count()                 # does not initialise HIP on this image
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ngpu < args.gpus:
        if ngpu == 0 or args.gpus > 6 * ngpu:
            raise SystemExit(f"--gpus {args.gpus}: only {ngpu} GPU(s) visible and at most 6 ranks may share one")
        env["FIE_DIST_BACKEND"] = "gloo"
        log(f"{ngpu} GPU(s) for {args.gpus} ranks: gloo rehearsal, ranks share devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="ssd-1b", choices=["ssd-1b", "sdxl"])
    ap.add_argument("--controlnet", default="full", choices=["full", "small"])
    ap.add_argument("--strength", type=float, default=0.5)
    ap.add_argument("--guidance", type=float, default=1.5)
    ap.add_argument("--weights", default="f16", choices=["f16", "f8e4m3"], help="f8e4m3: BASELINE config 5 (separate config, never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (skip the device-resident / in-flight / batch extras)")
    ap.add_argument("--in-flight", type=int, default=2, help="extras: edits in flight per GPU (independent hipGraph slots on separate streams)")
    ap.add_argument("--batch", type=int, default=0, help="extras: also time N images per device job (BASELINE config 'batch=8'); reported as "
                                                         "batched_images_per_sec, never as `value`")
    ap.add_argument("--no-graph", action="store_true", help="issue the launches eagerly instead of replaying the captured hipGraph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    dev_index = local % torch.cuda.device_count()      # == local on a real N-GPU node; folds ranks in a one-GPU gloo rehearsal
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FIE_DIST_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)

    import contextlib
    import io
    from src.pipeline import FastEditor
    from fie_amd import flops
    ekw = {}
    if args.weights != "f16":
        ekw["weight_dtype"] = args.weights
    with contextlib.redirect_stdout(io.StringIO() if rank else sys.stderr):
        editor = FastEditor(model_name=args.model, device=f"cuda:{dev_index}" if world > 1 else "cuda",
                            enable_cpu_offload=False, use_full_controlnet=args.controlnet == "full", **ekw)
    pipe = editor.pipe
    pipe.use_graph = not args.no_graph
    cfgs = pipe.cfgs
    items = load_items()
    total = args.warmup + args.steps

    # image-parallel shard: rank r takes items r, r+W, ...; weak scaling = K DIFFERENT items per rank.  The source
    # images are decoded PIL images in host memory, as at run_batch.py:199 (image open / JPEG decode is outside the timer there)
    work = []
    for s in range(total):
        it = items[(rank + s * world) % len(items)]
        work.append((synth_item_image(int(it["image_id"]) % 100000 + s), it["editing_prompt"]))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fp8_scales = None
    if args.weights == "f8e4m3":
        # config 5 runs with CALIBRATED activation scales (one eager edit on the first item: per-tensor amax -> power-of-two scale, pipe.calibrate_fp8)
        img0 = work[0][0].resize((1024, 1024))
        sc = pipe.calibrate_fp8(prompt=work[0][1], image=img0, control_image=editor.preprocess_image(img0), negative_prompt="", strength=args.strength,
                                num_inference_steps=4, guidance_scale=args.guidance, controlnet_conditioning_scale=0.5,
                                generator=torch.Generator("cpu").manual_seed(42))
        flat = [v for vs in sc.values() for v in vs]
        fp8_scales = {"tensors": len(flat), "min": min(flat), "max": max(flat)}

    def edit(s):
        # exactly the call of run_batch.py:209-219 (PIL in -> PIL out: LANCZOS 1024^2, Canny, tokenise, RNG, H2D, CLIP x2, VAE
        # encode, evals x (ControlNet + UNet), CFG + LCM steps, VAE decode, u8, D2H); strength is the additive flag
        return editor.edit(work[s][0], work[s][1], negative_prompt="", strength=args.strength, num_inference_steps=4,
                           guidance_scale=args.guidance, controlnet_conditioning_scale=0.5, seed=42)

    # ---- HEADLINE: K serial edits, one at a time, timer placed as run_batch.py:208-221 (sum over the K calls)
    for s in range(args.warmup):
        edit(s)                                        # first call captures the hipGraph, allocates Canny / resize tables
    barrier()
    t_ = time.perf_counter()
    for s in range(args.warmup, total):
        out_img = edit(s)
    barrier()
    elapsed = time.perf_counter() - t_
    assert out_img.size == (1024, 1024)
    per_rank = [elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if dist.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        g = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(g, t)                      # C2: gather of per-rank timings (metrics) on every rank
        per_rank = [float(x.item()) for x in g]
        elapsed = max(per_rank)
    log(f"timed region: {elapsed:.3f}s for {args.steps} serial PIL-in -> PIL-out edits")
    evals, nb = pipe.last_stats["unet_evals"], pipe.last_stats["cfg_batch"]
    value = args.steps * world / elapsed

    # ---- extras (named keys, never `value`): device-resident hipGraph replay, two edits in flight, batch N
    extras = {}
    from PIL import Image
    first_src, first_prompt = work[args.warmup % total]
    first_inp = first_src.resize((1024, 1024), Image.LANCZOS)
    first_ctrl = editor.preprocess_image(first_inp)
    if not args.no_extras and not args.no_graph:
        nrep = min(args.steps, 8)
        jobs = []
        for s in range(nrep + 2):
            src, prompt = work[s % total]
            inp = src.resize((1024, 1024), Image.LANCZOS)
            gen = torch.Generator(device="cpu").manual_seed(42)
            jobs.append(pipe.prepare(prompt, "", inp, editor.preprocess_image(inp), args.strength, 4, args.guidance, 0.5, gen))
        torch.cuda.synchronize()
        nfl = max(1, args.in_flight)
        streams = [pipe.new_slot_stream(i, equal_priority=True) for i in range(nfl)]
        if nfl > 1:                                          # untimed set-up: make sure the in-flight streams really overlap
            streams = pipe.calibrate_streams([jobs[i] for i in range(nfl)], streams, log=log)

        def replay_pass(n_streams):
            for s in range(2):
                with torch.cuda.stream(streams[s % n_streams]):
                    pipe.run_device_graphed(jobs[s], slot=s % n_streams)
            barrier()
            t0 = time.perf_counter()
            for s in range(2, nrep + 2):
                with torch.cuda.stream(streams[s % n_streams]):
                    pipe.run_device_graphed(jobs[s], slot=s % n_streams)
            barrier()
            return nrep * world / (time.perf_counter() - t0)

        extras["device_resident_images_per_sec"] = round(replay_pass(1), 4)
        if nfl > 1:
            extras["device_resident_in_flight_images_per_sec"] = {"in_flight": nfl, "value": round(replay_pass(nfl), 4)}
            from concurrent.futures import ThreadPoolExecutor
            editor.set_in_flight(nfl)
            editor.calibrate_in_flight(first_src, first_prompt, strength=args.strength, guidance_scale=args.guidance, seed=42)
            per = max(2, args.steps // nfl)

            def worker(slot, lo, n):
                editor.worker_slot(slot)
                for s in range(n):
                    edit((lo + s * nfl + slot) % total)

            with ThreadPoolExecutor(max_workers=nfl) as pool:
                list(pool.map(lambda sl: worker(sl, 0, 1), range(nfl)))          # capture the per-slot graphs
                barrier()
                t1 = time.perf_counter()
                list(pool.map(lambda sl: worker(sl, args.warmup, per), range(nfl)))
                barrier()
                extras["e2e_in_flight_images_per_sec"] = {"in_flight": nfl, "value": round(per * nfl * world / (time.perf_counter() - t1), 4)}
            editor.set_in_flight(1)
            editor.worker_slot(0)
        if args.batch > 1:                                   # N images per job: UNet / ControlNet / CLIP at batch N x CFG
            srcs = [work[s % total][0] for s in range(args.batch)]
            prompts = [work[s % total][1] for s in range(args.batch)]
            editor.edit_batch(srcs, prompts, strength=args.strength, guidance_scale=args.guidance, seed=42)
            barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                editor.edit_batch(srcs, prompts, strength=args.strength, guidance_scale=args.guidance, seed=42)
            barrier()
            extras["batched_images_per_sec"] = {"batch": args.batch, "what": "PIL in -> PIL out, edit_batch()",
                                                "value": round(3 * args.batch * world / (time.perf_counter() - t0), 4)}
        log(f"extras: {extras}")

    # ---- roofline of the UNet forward, HIP-event timed on the launch stream (untimed-for-throughput passes)
    gen = torch.Generator(device="cpu").manual_seed(42)
    job0 = pipe.prepare(first_prompt, "", first_inp, first_ctrl, args.strength, 4, args.guidance, 0.5, gen)
    stage = {}
    for s in range(2):
        pipe.timing = []
        pipe.run_device(job0)
        for k, v in pipe.stage_ms().items():
            stage[k] = stage.get(k, 0.0) + v / 2
    pipe.timing = None
    dominant, second = time_dominant_kernel(pipe, nb)
    classes, classes_src = class_table()
    fl = flops.image_flops(cfgs, evals, nb)
    stage_graphs = stage_graph_ms(pipe, job0, fl) if not args.no_graph else {}
    log(f"stage graphs (single stream each): {stage_graphs}")
    # roofline pass: UNet alone on one stream (the overlapped production schedule interleaves ControlNet kernels)
    unet_ms_per_fwd = time_unet_forward(pipe, job0)
    unet_tflops = fl["unet"] * nb / (unet_ms_per_fwd * 1e-3) / 1e12
    image_tflops = fl["total"] / (sum(stage.values()) * 1e-3) / 1e12

    # ---- where one serial edit's host time goes (4 more edits with the three host-side stages wrapped)
    marks = {}

    def wrap(obj, name):
        fn = getattr(obj, name)

        def timed(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            marks[name] = marks.get(name, 0.0) + (time.perf_counter() - t0) * 1e3 / 4
            return r
        setattr(obj, name, timed)
        return fn

    saved = [(o, n, wrap(o, n)) for o, n in ((editor, "_canny_device"), (pipe, "prepare"), (pipe, "run_device_graphed"), (pipe, "_to_host"))]
    t1 = time.perf_counter()
    for s in range(4):
        edit(args.warmup + s % max(args.steps, 1))
    e2e_ms = (time.perf_counter() - t1) / 4 * 1e3
    for o, n, fn in saved:
        setattr(o, n, fn)
    log(f"stage ms: { {k: round(v, 1) for k, v in stage.items()} }; host ms/edit { {k: round(v, 1) for k, v in marks.items()} } of {e2e_ms:.1f}")

    if rank == 0:
        out = {
            "metric": "PIE-Bench images/sec @1024^2 SSD-1B fp16 4-step", "value": round(value, 4), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if args.weights == "f16" else ("e4m3 x e4m3 on the block-scaled MFMA (fp32 accumulate) for the transformer-block projections of the UNet / ControlNet: "
                                                                           "e4m3 weights with per-channel scales, e4m3 activations written by LayerNorm / attention / GEGLU / GroupNorm at calibrated power-of-two scales (saturating, NaN kept); "
                                                                           "e4m3 weights x f16 activations for their other GEMMs / convs; f16 elsewhere"),
            "data": "synthetic (seeded PIE-Bench-shaped 512^2 images, real PIE-Bench prompts, seeded random-init weights, stand-in tokenizer)",
            "timed_region": "K serial FastEditor.edit() calls on K different items, PIL in -> PIL out, timer placed as run_batch.py:208-221 "
                            "(includes LANCZOS, Canny, tokenise, RNG, H2D, the device graph, D2H, PIL); excludes model load and image decode",
            "config": {"workload": f"{args.model} fp16 + ControlNet-Canny({args.controlnet}) LCM img2img, num_inference_steps=4, "
                                   f"strength={args.strength}, guidance={args.guidance}, 1024x1024, batch=1 image per GPU, one edit at a time",
                       "unet_preset": cfgs["unet"]["name"], "controlnet_preset": cfgs["controlnet"]["name"],
                       "unet_evals": evals, "cfg_batch": nb, "parallelism": f"image-parallel x{world}",
                       "launch": "eager" if args.no_graph else "hipGraph replay", "weights": args.weights, "fp8_activation_scales": fp8_scales,
                       "tflop_per_image": round(fl["total"] / 1e12, 2)},
            # the dominant problem (top row of profiles/r04_per_shape_roofline.md), HIP-event timed above on cold weights; the second kernel,
            # the per-class table of the committed kernel trace and the whole UNet forward (every launch between the events bracketing
            # unet.encode + unet.decode, alone on one stream, replayed from a hipGraph) are priced beside it
            "roofline": {"bound": "mfma", "kernel": dominant["kernel"], "shape": dominant["shape"],
                         "achieved": dominant["achieved"], "peak": PEAK_F16_DENSE_TFLOPS, "unit": "TFLOP/s",
                         "frac": dominant["frac"], "traffic": dominant["traffic_mb_per_launch"],
                         "traffic_unit": "MB/launch, fabric side: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes",
                         "traffic_source": dominant["traffic_source"],
                         "avg_us": dominant["avg_us"], "launches": dominant["launches"], "weight_copies": dominant["weight_copies"],
                         "avg_us_what": "HIP events on the launch stream, the weight matrix rotated over > 256 MB of copies: the in-network (cold-weight) duration",
                         "algorithmic_gflop_per_launch": dominant["algorithmic_gflop_per_launch"],
                         "algorithmic_mb_per_launch": dominant["algorithmic_mb_per_launch"],
                         "second": second,
                         "second_what": "largest problem of the kernel family with the largest share of device time (M = 2048 x N = 1280 projections on the small ring tiles)",
                         "class_table": classes, "class_table_source": classes_src,
                         "class_table_what": "kernel classes of one edit, largest first: launches, kernel ms, share, aggregate rate, fraction of the bound (2.5 PF MFMA / 8 TB/s HBM); from the committed rocprofv3 kernel trace, not measured in this run",
                         "unet_forward": {"what": f"{cfgs['unet']['name']}, batch {nb}, random inputs", "achieved": round(unet_tflops, 2),
                                          "frac": round(unet_tflops / PEAK_F16_DENSE_TFLOPS, 4),
                                          "algorithmic_tflop": round(fl["unet"] * nb / 1e12, 3), "ms": round(unet_ms_per_fwd, 3)}},
            "stage_ms": {k: round(v, 2) for k, v in stage.items()},
            "stage_ms_what": "eager single pass with HIP events (includes the host's per-launch cost on chains of small kernels: CLIP)",
            "stage_graph_ms": stage_graphs,
            "stage_graph_ms_what": "every stage as its own single-stream hipGraph, replayed: device time + algorithmic TFLOP/s; the production graph also overlaps CLIP || VAE encode and UNet encoder || ControlNet trunk",
            "host_ms_per_edit": {k: round(v, 2) for k, v in marks.items()},
            "image_tflops": round(image_tflops, 2),
            "per_rank_seconds": [round(x, 4) for x in per_rank],
            "dist_backend": None if dist is None else dist.get_backend(),
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(editor, cfgs, (first_inp, first_ctrl, first_prompt), evals, nb, args.strength, args.guidance)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
