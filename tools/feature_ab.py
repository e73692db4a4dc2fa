"""Whole-edit (hipGraph replay) A/B of one boolean switch of the HIP context, one process, alternating captures (a captured graph keeps
what was decided at capture time).  usage: tools/feature_ab.py <ctx attribute, e.g. gn_from_epilogue | pipe.<attribute>> [model] [rounds]
FIE_AB_FORK=1 keeps the two-stream graphs (needed for switches that move work between the streams; at most 6 captures = 3 rounds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

attr = sys.argv[1]
model = sys.argv[2] if len(sys.argv) > 2 else "ssd-1b"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ed = FastEditor(model_name=model, use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx = pipe.ctx
target = ctx
if attr.startswith("pipe."):
    target, attr = pipe, attr[5:]
assert isinstance(getattr(target, attr), bool), attr
if os.environ.get("FIE_AB_FORK", "0") != "1":
    pipe.fork_streams = False      # single-stream graphs: immune to the hardware-queue collisions that many forked graphs in one process cause
pipe.max_graphs = 64
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
n_cap, outs = 0, {}
for rnd in range(rounds):
    for on in (False, True):
        setattr(target, attr, on)
        job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5 + 1e-4 * n_cap, 0.5, torch.Generator().manual_seed(42))
        n_cap += 1
        out = pipe.run_device_graphed(job)
        torch.cuda.synchronize()
        outs[on] = out.clone() if torch.is_tensor(out) else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            pipe.run_device_graphed(job)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd} [{attr}={on}]: edit {e0.elapsed_time(e1) / 8:.2f} ms", flush=True)
if outs.get(True) is not None and outs.get(False) is not None:
    d = (outs[True].float() - outs[False].float()).abs()
    print(f"output difference between the two settings (guidance differs by 1e-4): max {d.max().item():.0f}, mean {d.mean().item():.4f} (u8 levels)")
