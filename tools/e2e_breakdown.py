"""Where the end-to-end time of FastEditor.edit() goes (host + device), per call, before and after other streams /
graph slots have been used in the process (as bench.py does).  GPU tool, not a test.

    python tools/e2e_breakdown.py [n_calls]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402  (sets GPU_MAX_HW_QUEUES before torch initialises HIP)
import torch  # noqa: E402

from bench import synth_item_image  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
img = synth_item_image(3)
pipe = ed.pipe
marks = []


def wrap(obj, name, label, sync=False):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        marks.append((label, (time.perf_counter() - t) * 1e3))
        if sync:
            t = time.perf_counter()
            torch.cuda.current_stream().synchronize()
            marks.append(("device wait", (time.perf_counter() - t) * 1e3))
        return r
    setattr(obj, name, timed)


wrap(ed, "_canny_device", "resize+canny(sync)")
wrap(pipe, "prepare", "prepare")
wrap(pipe, "run_device_graphed", "graph launch (host)", sync=True)
wrap(pipe.ctx, "canny_finish", "canny flags (after the edit)")
wrap(pipe, "_to_host", "D2H")


def phase(tag):
    for i in range(n):
        marks.clear()
        t0 = time.perf_counter()
        ed.edit(img, "a photo of a [red] house", strength=0.5, guidance_scale=1.5, seed=42)
        total = (time.perf_counter() - t0) * 1e3
        acc = sum(v for _, v in marks)
        print(f"{tag} call {i}: {total:7.1f} ms  " + "  ".join(f"{k} {v:.1f}" for k, v in marks) + f"  D2H+PIL {total - acc:.1f}", flush=True)


phase("fresh")
# what bench.py does before its end-to-end calls: jobs replayed back to back on two other streams / graph slots
inp = img.resize((1024, 1024))
ctrl = ed.preprocess_image(inp)
jobs = [pipe.prepare("a photo of a [red] house", "", inp, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42)) for _ in range(2)]
streams = [torch.cuda.Stream(priority=-(i % 2)) for i in range(2)]
for r in range(4):
    for s in range(2):
        with torch.cuda.stream(streams[s]):
            pipe.run_device_graphed.__wrapped__(jobs[s], slot=s) if hasattr(pipe.run_device_graphed, "__wrapped__") else pipe.run_device_graphed(jobs[s], slot=s)
torch.cuda.synchronize()
phase("after 2-in-flight")
if os.environ.get("EAGER"):
    pipe.timing = []
    pipe.run_device(jobs[0])
    print({k: round(v, 1) for k, v in pipe.stage_ms().items()})
    pipe.timing = None
    phase("after eager")
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
