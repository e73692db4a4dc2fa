"""Whole-edit (hipGraph replay) A/B of the thin-conv kernel (csrc/conv_thin.hip, tile code 77) against the tiles those convs had before, one process,
alternating captures.  The switch is fie_debug_tune_exclude("77"), which also forgets the tuner's remembered choices: run with FIE_AUTOTUNE=0 so that both
arms use the built-in rule for every other shape.  usage: FIE_AUTOTUNE=0 tools/thin_conv_edit_ab.py [model] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "ssd-1b"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ed = FastEditor(model_name=model, use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx = pipe.ctx
pipe.fork_streams = os.environ.get("FIE_AB_FORK", "0") == "1"
pipe.max_graphs = 64
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
n_cap, outs = 0, {}
for rnd in range(rounds):
    for thin in (False, True):
        ctx.tune_exclude("" if thin else "77")
        job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5 + 1e-4 * n_cap, 0.5, torch.Generator().manual_seed(42))
        n_cap += 1
        out = pipe.run_device_graphed(job)
        torch.cuda.synchronize()
        outs[thin] = out.clone() if torch.is_tensor(out) else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            pipe.run_device_graphed(job)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd} [thin conv {'on' if thin else 'off'}]: edit {e0.elapsed_time(e1) / 8:.2f} ms", flush=True)
ctx.tune_exclude("")
if outs.get(True) is not None and outs.get(False) is not None:
    d = (outs[True].float() - outs[False].float()).abs()
    print(f"output difference between the two settings (guidance differs by 1e-4): max {d.max().item():.0f}, mean {d.mean().item():.4f} (u8 levels)")
