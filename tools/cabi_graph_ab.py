"""One edit as a hipGraph, two ways, one process, alternating: the product's Python-walk graph (single stream) and the same edit with EVERY model call going
through a C-ABI forward (fie_amd/cabi.py::run_edit: csrc/graphs.cpp walks on registered weights), captured on the same stream.  VERDICT r3 item 6 asks for the
C++ walks within 2 % of the Python-walk graph.  Prints the replay times and the u8 difference of the two outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image  # noqa: E402
from fie_amd import cabi  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx = pipe.ctx
pipe.fork_streams = False
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42))
cabi.register_pipeline(pipe)
ref = pipe.run_device_graphed(job).clone()          # tunes (eager warm-up) and captures the product graph
torch.cuda.synchronize()

# the C-ABI edit: eager once (the tuner has met every shape above; the walks meet the same ones), then captured on the context's capture stream
ctx.autotune(pipe._tune_mode())
out = cabi.run_edit(pipe, job)
torch.cuda.synchronize()
ctx.autotune(2 if pipe.autotune else 0)
s = ctx.capture_stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    cap = cabi.run_edit(pipe, job)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
d = (cap.int() - ref.int()).abs()
print(f"C-ABI graph vs product graph: max |du8| {int(d.max())}, mean {d.float().mean().item():.4f}; C-ABI graph == C-ABI eager: {torch.equal(cap, out)}")


def timed(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rnd in range(3):
    a = timed(lambda: pipe.run_device_graphed(job))
    b = timed(g.replay)
    print(f"round {rnd}: product graph (Python walks, one stream) {a:.2f} ms   C-ABI forwards graph {b:.2f} ms   ratio {b / a:.3f}", flush=True)
