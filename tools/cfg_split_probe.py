"""Would running the two CFG halves (cond / uncond) as two concurrent batch-1 streams inside one edit beat the batch-2 pass?
Times, as hipGraph replays: one UNet forward at batch 2 on one stream against two batch-1 forwards forked onto two streams (the
ControlNet trunk behaves the same).  Two edits in flight gain 14 % from overlapping kernel tails; this asks whether one edit can
get part of that by itself.  usage: tools/cfg_split_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: F401,E402
import torch  # noqa: E402

from src.pipeline import FastEditor  # noqa: E402

ed = FastEditor(model_name="ssd-1b", use_full_controlnet=True, enable_cpu_offload=False)
pipe = ed.pipe
ctx, dev = pipe.ctx, pipe.ctx.device
cfg = pipe.cfgs["unet"]
xd = cfg["cross_attention_dim"]
pdim = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
g = torch.Generator(device=dev).manual_seed(0)


def inputs(nb):
    text = torch.randn((nb * 77, xd), generator=g, device=dev, dtype=torch.float16)
    pooled = torch.randn((nb, pdim), generator=g, device=dev, dtype=torch.float16)
    x = torch.zeros((nb, 128, 128, 8), device=dev, dtype=torch.float16)
    x[..., :4] = torch.randn((nb, 128, 128, 4), generator=g, device=dev, dtype=torch.float16)
    tid = torch.tensor([[1024., 1024., 0, 0, 1024., 1024.]]).repeat(nb, 1).to(dev)
    t = torch.full((nb, 1), 499.0, device=dev)
    return text, pooled, x, tid, t


def forward(inp, tag):
    text, pooled, x, tid, t = inp
    ctx.ws_tag = tag
    for tr in pipe.unet.transformers():
        tr.reset()
    pipe.unet.begin_image(pooled, tid)
    tb = pipe.unet.time_rowbias(t)
    skips, mid = pipe.unet.encode(pipe.unet.conv_in(ctx, x), tb, text, 77)
    out = pipe.unet.decode(mid, skips, tb, text, 77)
    ctx.ws_tag = 0
    return out


i2, i1a, i1b = inputs(2), inputs(1), inputs(1)
ctx.autotune(1)
forward(i2, 20); forward(i1a, 21); forward(i1b, 22)          # eager: workspaces, tile autotune for both batch sizes
torch.cuda.synchronize()
ctx.autotune(2)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def capture(fn):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s0):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s0):
            fn()
    return gr


def batched():
    forward(i2, 20)


def split():
    s1.wait_stream(torch.cuda.current_stream())
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        forward(i1a, 21)
    with torch.cuda.stream(s2):
        forward(i1b, 22)
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)


def single():
    forward(i1a, 21)


graphs = {"batch 2, one stream": capture(batched), "2 x batch 1, two streams": capture(split), "batch 1 alone": capture(single)}
for rnd in range(3):
    for name, gr in graphs.items():
        with torch.cuda.stream(s0):
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
        print(f"round {rnd} {name:26s}: {e0.elapsed_time(e1) / 6:.2f} ms per UNet forward (CFG pair)", flush=True)
