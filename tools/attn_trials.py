"""Whole-UNet A/B of the attention block size (fie_debug_attn_variant 0 = heuristic, 2 = 128 queries per block, 3 = 64)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image, time_unet_forward  # noqa: E402
from fie_amd import hip  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx = pipe.ctx
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42))
pipe.run_device(job)
torch.cuda.synchronize()
for rnd in range(3):
    row = []
    for v in (int(x) for x in (sys.argv[1:] or ['0', '5', '4', '2'])):
        hip.lib().fie_debug_attn_variant(ctx.h, v)
        row.append(f"variant {v}: {min(time_unet_forward(pipe, job, iters=4) for _ in range(2)):.3f} ms")
    print("  ".join(row), flush=True)
hip.lib().fie_debug_attn_variant(ctx.h, 0)
