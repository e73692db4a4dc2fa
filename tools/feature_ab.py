#!/usr/bin/env python3
"""In-process A/B of graph-level fusion switches (fie_amd.nn.FOLD_LN, BATCH_TEXT_KV, ...): one pipeline per setting on the same
synthetic weights, hipGraph replays alternated over several rounds (cdna_hip_programming.md rule 24: devices differ by ~5 %, so
only a same-process comparison ranks two builds).
usage: tools/feature_ab.py FLAG=0 [FLAG2=0 ...]      (each argument = one variant: the named nn switches set as given; the
                                                      first variant is always "all defaults")"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from bench import synth_item_image, time_unet_forward  # noqa: E402
from fie_amd import nn  # noqa: E402
from src.pipeline import FastEditor  # noqa: E402

variants = [""] + sys.argv[1:]
defaults = {k: getattr(nn, k) for k in dir(nn) if k.isupper() and isinstance(getattr(nn, k), bool)}
eds = []
img = synth_item_image(3).resize((1024, 1024))
for v in variants:
    for k, d in defaults.items():
        setattr(nn, k, d)
    for kv in filter(None, v.split(",")):
        k, val = kv.split("=")
        assert k in defaults, f"unknown switch {k}: {sorted(defaults)}"
        setattr(nn, k, bool(int(val)))
    ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
    ed.pipe.fork_streams = os.environ.get("FIE_AB_FORK", "1") == "1"
    ctrl = ed.preprocess_image(img)
    job = ed.pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42))
    out = ed.pipe.run_device_graphed(job).clone()
    eds.append((v or "defaults", ed, job, out))
for k, d in defaults.items():
    setattr(nn, k, d)
ref = eds[0][3].float()
for name, _, _, out in eds[1:]:
    print(f"[{name}] max |du8| vs defaults = {int((out.float() - ref).abs().max())}", flush=True)
for rnd in range(4):
    for name, ed, job, _ in eds:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ed.pipe.run_device_graphed(job)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd} [{name}]: edit {e0.elapsed_time(e1) / 8:.2f} ms", flush=True)
