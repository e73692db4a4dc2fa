"""Whole-edit (hipGraph replay) A/B of the tuner's candidate set, one process, alternating captures: each argument is a list of tile codes
the tuner must NOT offer ("" = everything allowed, "10000" = no split-K variant, "63" = no 256x320 tile, "63,10000" = neither); every
setting re-tunes from scratch during the eager warm-up of its capture (a captured graph keeps what was chosen then).
usage: tools/tuner_ab.py "" "10000" "63" "63,10000" [--rounds N] [--model ssd-1b]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

args = sys.argv[1:]
rounds, model = 2, "ssd-1b"
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
if "--model" in args:
    i = args.index("--model"); model = args[i + 1]; del args[i:i + 2]
specs = args or ["", "10000"]
ed = FastEditor(model_name=model, use_full_controlnet=True, enable_cpu_offload=False)
pipe, ctx = ed.pipe, ed.pipe.ctx
pipe.fork_streams = False      # single-stream graphs: immune to the hardware-queue collisions that many forked graphs in one process cause
pipe.max_graphs = 64
img = synth_item_image(3).resize((1024, 1024))
ctrl = ed.preprocess_image(img)
n_cap, outs = 0, {}
for rnd in range(rounds):
    for sp in specs:
        ctx.tune_exclude(sp)
        job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5 + 1e-4 * n_cap, 0.5, torch.Generator().manual_seed(42))
        n_cap += 1
        out = pipe.run_device_graphed(job)
        torch.cuda.synchronize()
        outs[sp] = out.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            pipe.run_device_graphed(job)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd} [exclude '{sp}']: edit {e0.elapsed_time(e1) / 8:.2f} ms", flush=True)
        if rnd == rounds - 1:
            n, rep = ctx.autotune_report()
            picked = [l for l in rep.splitlines() if int(l.rsplit("-> ", 1)[1]) >= 10000 or l.endswith("-> 63")]
            print(f"   {n} problems tuned; split-K / 256x320 choices: {len(picked)}")
            for l in picked:
                print("     " + l)
base = outs[specs[0]].float()
for sp in specs[1:]:
    d = (outs[sp].float() - base).abs()
    print(f"output ['{sp}'] vs ['{specs[0]}'] (guidance differs by <= {1e-4 * n_cap:.4f}): max {d.max().item():.0f}, mean {d.mean().item():.4f} u8 levels")
ctx.tune_exclude("")
