"""UNet forward + whole-edit time with the default kernel selection or with FIE_PRODUCER_TILES=1 / FIE_160_TILES=1 (run alternately)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image, time_unet_forward  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42))
pipe.run_device_graphed(job)
torch.cuda.synchronize()
res = []
for rep in range(3):
    ms = time_unet_forward(pipe, job, iters=5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        pipe.run_device_graphed(job)
    e1.record()
    torch.cuda.synchronize()
    res.append((ms, e0.elapsed_time(e1) / 6))
print("FIE_PRODUCER_TILES=1" if os.environ.get("FIE_PRODUCER_TILES") else "default selection", " | ".join(f"unet fwd {a:.2f} ms, edit {b:.2f} ms" for a, b in res))
