#!/usr/bin/env python3
"""Does in-process stream concurrency fill the CUs left idle by small-grid kernels?  100 small GEMMs on one stream vs the
same 2 x 100 on two streams (eager launches and hipGraph replays)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
m, n, k = 2048, 1280, 1280
a = [torch.randn(m, k, device="cuda", dtype=torch.float16) for _ in range(2)]
w = [ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * 0.03) for _ in range(2)]
o = [torch.empty(m, n, device="cuda", dtype=torch.float16) for _ in range(2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]


def work(i, reps=100):
    for _ in range(reps):
        ctx.gemm(a[i], w[i], n, out=o[i])


def timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


work(0, 5)
one = timed(lambda: work(0))
seq = timed(lambda: (work(0), work(1)))


def two_streams():
    for i in range(2):
        with torch.cuda.stream(s[i]):
            work(i)


two = timed(two_streams)
print(f"eager : 100 GEMMs {one:.2f} ms | 200 on one stream {seq:.2f} ms | 2 x 100 on two streams {two:.2f} ms", flush=True)

graphs = []
for i in range(2):
    with torch.cuda.stream(s[i]):
        work(i, 3)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s[i]):
        work(i)
    graphs.append(g)
g_one = timed(lambda: graphs[0].replay())
g_seq = timed(lambda: (graphs[0].replay(), graphs[1].replay()))


def g_two():
    for i in range(2):
        with torch.cuda.stream(s[i]):
            graphs[i].replay()


g2 = timed(g_two)
print(f"graphs: one {g_one:.2f} ms | two replays on one stream {g_seq:.2f} ms | two replays on two streams {g2:.2f} ms", flush=True)
