"""Does a second / third captured edit graph run as fast as the first?  (hardware-queue assignment of the graphs' branch streams)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
pipe.fork_streams = os.environ.get("FORK", "1") == "1"
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)


def timed(job, n=6):
    pipe.run_device_graphed(job)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.run_device_graphed(job)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


jobs = [pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, g, 0.5, torch.Generator().manual_seed(42)) for g in (1.5, 1.6, 1.7, 1.8)]
with torch.cuda.stream(pipe.slot_stream(0)):
    for i, j in enumerate(jobs):
        print(f"capture + replay graph {i}: {timed(j):.2f} ms", flush=True)
    for i, j in enumerate(jobs):
        print(f"replay graph {i} again: {timed(j):.2f} ms", flush=True)
    if os.environ.get("CLEAR"):
        pipe._graphs.clear()
        for i, j in enumerate(jobs[:2]):
            print(f"after clear, capture + replay graph {i}: {timed(j):.2f} ms", flush=True)
