"""Whole-UNet A/B of per-shape tile choices (cold weights, real activations): `fie_debug_tile_override` + bench.time_unet_forward.
The per-op microbenchmark re-runs ONE layer, so its weights sit in the Infinity Cache; inside the UNet every layer's weights
come from HBM, and choices that win there can lose here.  usage: tools/tile_trials.py [spec ...]  (default: built-in list)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image, time_unet_forward  # noqa: E402
from fie_amd import hip  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

CODES = [42, 43, 44, 46, 51, 95, 52, 54, 96, 81]
TRIALS = {name: CODES for name in (
    "FF1   0,2048,10240,1280", "FF2   0,2048,1280,5120", "proj  0,2048,1280,1280", "QKV   0,2048,3840,1280", "kv?   0,2048,1280,640",
    "FF1h  0,8192,5120,640", "projh 0,8192,640,640", "FF2h  0,8192,640,2560", "QKVh  0,8192,1920,640", "sc    0,8192,640,1280",
    "c32a  1,2048,1280,5760", "c32b  1,2048,1280,11520", "c32c  1,2048,1280,17280", "c32d  1,2048,1280,23040",
    "c64a  1,8192,640,5760", "c64b  1,8192,640,11520", "c64c  1,8192,640,17280", "c64u  1,8192,1280,11520",
    "c128a 1,32768,320,2880", "c128b 1,32768,320,5760", "c128c 1,32768,320,8640", "c128u 1,32768,640,5760")}

ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx = pipe.ctx
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42))
ctx.autotune(1)                # the base of every row is the cold-timed per-op autotune's choice
pipe.run_device(job)
torch.cuda.synchronize()
ctx.autotune(2)
print(ctx.autotune_report()[1])


def measure(spec):
    ctx.tile_override(spec or None)
    return min(time_unet_forward(pipe, job, iters=4) for _ in range(2))


specs = sys.argv[1:]
if specs:
    base = measure("")
    for sp in specs:
        t = measure(sp)
        print(f"{sp:60s} {t:7.3f} ms  ({(t / base - 1) * 100:+.2f} % vs {base:.3f})", flush=True)
else:
    for name, codes in TRIALS.items():
        shape = name.split()[1]
        base = measure("")
        row = [f"{name:28s} base {base:7.3f}"]
        for c in codes:
            t = measure(f"{shape}={c}")
            row.append(f"{c}: {(t / base - 1) * 100:+.2f}%")
        print("  ".join(row), flush=True)
ctx.tile_override(None)
