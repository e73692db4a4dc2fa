"""Per-shape device time of one eager edit (single stream, HIP events around every C-ABI op).  GPU tool, not a test.

    python tools/shape_tally.py [model] > gpurun_out/shape_tally.txt
"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image  # noqa: E402
from fie_amd import hip  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "ssd-1b"
ed = FastEditor(model_name=model, use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
pipe.fork_streams = False
ctx = pipe.ctx
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42))
pipe.run_device(job)
torch.cuda.synchronize()

records = []


def wrap(name, keyfn):
    fn = getattr(ctx, name)

    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        records.append((name, keyfn(*a, **k), e0, e1))
        return r
    setattr(ctx, name, timed)


def gemm_key(a, wp, n, out=None, a2=None, bias=None, rowbias=None, rows_per_batch=0, residual=None, scale=1.0, act=0, k=None):
    kk = a.shape[1] + (a2.shape[1] if a2 is not None else 0)
    return f"M={a.shape[0]} N={n} K={kk} act={act} res={int(residual is not None)}"


def conv_key(x, wp, cout, out=None, stride=1, pad_mode=0, upsample=False, **kw):
    return f"B={x.shape[0]} {x.shape[1]}x{x.shape[2]} {x.shape[3]}->{cout} s{stride} u{int(upsample)}"


def attn_key(q, k, v, heads, head_dim, tq, tk, batch, **kw):
    return f"B={batch} H={heads} D={head_dim} Tq={tq} Tk={tk}"


def gn_key(x1, gamma, beta, groups, eps, silu, x2=None, out=None):
    return f"B={x1.shape[0]} rows={x1[0].numel() // x1.shape[-1]} C={x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)}"


def ln_key(x, *a, **k):
    return f"rows={x.shape[0]} C={x.shape[1]}"


wrap("gemm", gemm_key)
wrap("conv3x3", conv_key)
wrap("attention", attn_key)
wrap("groupnorm", gn_key)
wrap("layernorm", ln_key)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
pipe.run_device(job)
t1.record()
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for name, key, e0, e1 in records:
    a = agg[(name, key)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
total = t0.elapsed_time(t1)
covered = sum(v[1] for v in agg.values())
print(f"# {model}: one eager edit {total:.1f} ms (event-instrumented), ops covered {covered:.1f} ms")
print(f"{'op':10s} {'shape':52s} {'n':>5s} {'ms':>8s} {'us/call':>8s} {'%':>5s}")
for (name, key), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{name:10s} {key:52s} {n:5d} {ms:8.2f} {ms / n * 1e3:8.1f} {ms / total * 100:5.1f}")
by_op = defaultdict(float)
for (name, key), (n, ms) in agg.items():
    by_op[name] += ms
print({k: round(v, 1) for k, v in by_op.items()})
