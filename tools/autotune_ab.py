"""Whole-edit (hipGraph replay) and UNet-forward time with the built-in tile rule versus the per-shape autotune, one process,
alternating captures (a captured graph keeps the kernels chosen at capture time).  usage: tools/autotune_ab.py [model] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image, time_unet_forward  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "ssd-1b"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ed = FastEditor(model_name=model, use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx = pipe.ctx
pipe.fork_streams = False      # single-stream graphs: immune to the hardware-queue collisions that many forked graphs in one process cause
pipe.max_graphs = 64
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
n_cap = 0
for rnd in range(rounds):
    for tag, on in (("rule", False), ("autotune", True)):
        pipe.autotune = on
        job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5 + 1e-4 * n_cap, 0.5, torch.Generator().manual_seed(42))
        n_cap += 1
        pipe.run_device_graphed(job)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            pipe.run_device_graphed(job)
        e1.record()
        torch.cuda.synchronize()
        fwd = time_unet_forward(pipe, job, iters=4)
        print(f"round {rnd} [{tag}]: edit {e0.elapsed_time(e1) / 8:.2f} ms, unet fwd {fwd:.2f} ms", flush=True)
n, rep = ctx.autotune_report()
print(f"{n} tuned problems")
print(rep)
