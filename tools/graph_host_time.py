#!/usr/bin/env python3
"""Host cost of replaying the ~2 500-node edit graph: time inside graph.replay() vs the GPU time of the replay; then one replay
per thread from two Python threads (two graph slots on two streams)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib, io  # noqa: E401,E402
from PIL import Image  # noqa: E402
from bench import synth_item_image  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    ed = FastEditor(model_name="ssd-1b", enable_cpu_offload=False, use_full_controlnet=True)
pipe = ed.pipe
jobs = []
for i in range(4):
    inp = synth_item_image(i).resize((1024, 1024), Image.LANCZOS)
    jobs.append(pipe.prepare("a [red] house", "", inp, ed.preprocess_image(inp), 0.5, 4, 1.5, 0.5, torch.Generator("cpu").manual_seed(42)))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for slot in range(2):
    with torch.cuda.stream(streams[slot]):
        pipe.run_device_graphed(jobs[slot], slot=slot)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    pipe.run_device_graphed(jobs[0], slot=0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"single replay: host {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms", flush=True)


def worker(slot, n):
    with torch.cuda.stream(streams[slot]):
        for i in range(n):
            pipe.run_device_graphed(jobs[slot + 2 * (i % 2)], slot=slot)
        streams[slot].synchronize()


for nthreads in (1, 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(s, 6)) for s in range(nthreads)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nthreads} thread(s) x 6 edits: {dt * 1e3:.1f} ms -> {nthreads * 6 / dt:.2f} images/s", flush=True)
