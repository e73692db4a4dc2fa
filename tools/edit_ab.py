"""Whole-edit (hipGraph replay) and UNet-forward time for a list of per-shape tile overrides, one process, alternating.
usage: tools/edit_ab.py "" "0,2048,1280,1280=43" ...   (each argument is a fie_debug_tile_override spec; "" = default selection)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image, time_unet_forward  # noqa: E402
from fie_amd import hip  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

specs = sys.argv[1:] or [""]
ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx = pipe.ctx
pipe.fork_streams = False      # single-stream graphs: immune to the hardware-queue collisions that many forked graphs in one process cause
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
# one graph per (round, spec): a captured graph keeps the kernels chosen at capture time; every capture gets its own key
# through a slightly different guidance value (graphs are never destroyed).
n_cap = 0
for rnd in range(3):
    for sp in specs:
        ctx.tile_override(sp or None)
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
        print(f"round {rnd} [{sp or 'default'}]: edit {e0.elapsed_time(e1) / 8:.2f} ms, unet fwd {fwd:.2f} ms", flush=True)
ctx.tile_override(None)
