#!/usr/bin/env python3
"""The halo-resident 3x3 conv (csrc/conv_halo.hip, tile code 71) against the im2col kernels on the VAE's conv shapes: correctness against
torch's conv2d (fp32 on the device), then interleaved rounds of every code in ONE process (cdna_hip_programming.md 5.4 rule 24), inputs
rotated over several copies so that activations come from HBM / the Infinity Cache as inside the decoder.
usage: tools/halo_conv.py [codes] [--stamps] [--quick]"""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"

# (B, H, Cin, Cout): the decoder / encoder maps of the SDXL VAE at 1024^2, the 128^2-latent UNet level
SHAPES = [(1, 1024, 128, 128), (1, 1024, 256, 128), (1, 512, 256, 256), (1, 512, 512, 256), (1, 512, 128, 256), (1, 256, 512, 512), (1, 256, 256, 512),
          (1, 128, 512, 512), (2, 128, 320, 320), (2, 128, 640, 320), (2, 64, 640, 640)]


def check(code=71):
    g = torch.Generator(device=DEV).manual_seed(0)
    worst = 0.0
    for b, h, w, cin, cout, opts in [(1, 16, 16, 64, 128, ""), (2, 32, 48, 128, 128, "bias,res"), (1, 48, 32, 192, 320, "bias,rowbias,silu"), (2, 64, 64, 256, 512, "bias,gn"),
                                     (2, 128, 128, 320, 320, "bias,rowbias,gn"), (2, 64, 64, 640, 640, "bias,res,gn"), (1, 512, 512, 128, 256, "res"),
                                     (1, 1024, 1024, 128, 128, "bias,res,gn"), (1, 1024, 1024, 128, 128, "bias")]:
        x = torch.randn(b, h, w, cin, generator=g, device=DEV, dtype=torch.float16)
        wt = torch.randn(cout, cin, 3, 3, generator=g, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5
        bias = torch.randn(cout, generator=g, device=DEV, dtype=torch.float16) if "bias" in opts else None
        res = torch.randn(b, h, w, cout, generator=g, device=DEV, dtype=torch.float16) if "res" in opts else None
        rb = torch.randn(b, cout, generator=g, device=DEV, dtype=torch.float16) if "rowbias" in opts else None
        wp = ctx.pack_conv3x3(wt)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), bias.float() if bias is not None else None, padding=1)
        if rb is not None:
            ref = ref + rb.float()[:, :, None, None]
        if "silu" in opts:
            ref = F.silu(ref)
        if res is not None:
            ref = ref + res.float().permute(0, 3, 1, 2)
        ctx.force_tile(code)
        y = ctx.conv3x3(x, wp, cout, bias=bias, residual=res, rowbias=rb, act=hip.ACT_SILU if "silu" in opts else hip.ACT_NONE, gn_groups=32 if "gn" in opts else None)
        assert "conv_halo" in hip.last_gemm_kernel(ctx), hip.last_gemm_kernel(ctx)
        torch.cuda.synchronize()
        err = ((y.float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()).item()
        ctx.force_tile(0)
        y0 = ctx.conv3x3(x, wp, cout, bias=bias, residual=res, rowbias=rb, act=hip.ACT_SILU if "silu" in opts else hip.ACT_NONE)
        err0 = ((y0.float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()).item()
        extra = ""
        if "gn" in opts and getattr(y, "_gn_tag", None) is not None:
            gam, bet = torch.ones(cout, device=DEV, dtype=torch.float16), torch.zeros(cout, device=DEV, dtype=torch.float16)
            gn = ctx.groupnorm(y, gam, bet, 32, 1e-5, True)
            gref = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, eps=1e-5))
            e2 = ((gn.float().permute(0, 3, 1, 2) - gref).abs().max() / gref.abs().max()).item()
            extra = f", GroupNorm from the epilogue's sums rel err {e2:.2e}"
            worst = max(worst, e2)
        print(f"check code {code} ({hip.last_gemm_kernel(ctx).split(' (')[0]}) B={b} {h}x{w} {cin}->{cout} [{opts}]: halo rel err {err:.2e} (rule kernel {err0:.2e}){extra}", flush=True)
        worst = max(worst, err)
    assert worst < 4e-3, worst
    print("correctness ok", flush=True)


def check_plus(code=72):
    """conv3x3(x) + [x2 | x3] W1x1^T (fie_conv3x3_plus_nhwc_f16: a resnet's conv2 + its 1x1 shortcut) on the halo-resident kernel against torch and against the
    ring kernels; repeats bit-identical."""
    g = torch.Generator(device=DEV).manual_seed(1)
    for b, h, w, cin, cout, c2, c3, gn in [(1, 64, 64, 128, 128, 256, 0, False), (2, 64, 96, 256, 256, 128, 64, True), (1, 320, 320, 128, 128, 256, 0, True),
                                            (2, 64, 64, 640, 640, 1280, 640, True), (1, 1024, 1024, 128, 128, 256, 0, True)]:
        x = torch.randn(b, h, w, cin, generator=g, device=DEV, dtype=torch.float16)
        x2 = torch.randn(b * h * w, c2, generator=g, device=DEV, dtype=torch.float16)
        x3 = torch.randn(b * h * w, c3, generator=g, device=DEV, dtype=torch.float16) if c3 else None
        wt = torch.randn(cout, cin, 3, 3, generator=g, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5
        w1 = torch.randn(cout, c2 + c3, generator=g, device=DEV, dtype=torch.float16) * (c2 + c3) ** -0.5
        bias = torch.randn(cout, generator=g, device=DEV, dtype=torch.float16)
        wc = ctx.pack_conv3x3(wt)
        wplus = torch.cat([wc[:, :9 * cin], ctx.pack_linear(w1)[:, :c2 + c3]], 1).contiguous()
        side = x2 if x3 is None else torch.cat([x2, x3], 1)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), bias.float(), padding=1) + (side.float() @ w1.float().T).view(b, h, w, cout).permute(0, 3, 1, 2)
        ctx.force_tile(code)
        outs = [ctx.conv3x3_plus(x, wplus, cout, x2, x3, bias=bias, gn_groups=32 if gn else None) for _ in range(5)]
        name = hip.last_gemm_kernel(ctx)
        assert "conv_halo2" in name, name
        torch.cuda.synchronize()
        err = ((outs[0].float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()).item()
        same = all(torch.equal(o, outs[0]) for o in outs[1:])
        ctx.force_tile(0)
        extra = ""
        if gn:
            gam, bet = torch.ones(cout, device=DEV, dtype=torch.float16), torch.zeros(cout, device=DEV, dtype=torch.float16)
            gnv = ctx.groupnorm(outs[-1], gam, bet, 32, 1e-5, True)
            gref = F.silu(F.group_norm(outs[-1].float().permute(0, 3, 1, 2), 32, eps=1e-5))
            extra = f", GroupNorm from its sums {((gnv.float().permute(0, 3, 1, 2) - gref).abs().max() / gref.abs().max()).item():.2e}"
        print(f"check +1x1 code {code} B={b} {h}x{w} {cin}->{cout} side {c2}+{c3}: rel err {err:.2e}, repeats identical: {same}{extra}", flush=True)
        assert err < 4e-3 and same


def bench_plus(codes):
    for b, h, cin, cout, c2, c3 in [(1, 1024, 128, 128, 256, 0), (1, 512, 256, 256, 512, 0), (1, 512, 256, 256, 128, 0), (1, 256, 512, 512, 256, 0), (2, 64, 640, 640, 1280, 640), (2, 64, 640, 640, 640, 640)]:
        copies = max(2, min(6, int(600e6 / (b * h * h * cin * 2)) + 1))
        xs = [torch.randn(b, h, h, cin, device=DEV, dtype=torch.float16) for _ in range(copies)]
        x2 = torch.randn(b * h * h, c2, device=DEV, dtype=torch.float16)
        x3 = torch.randn(b * h * h, c3, device=DEV, dtype=torch.float16) if c3 else None
        wc = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
        wplus = torch.cat([wc[:, :9 * cin], ctx.pack_linear(torch.randn(cout, c2 + c3, device=DEV, dtype=torch.float16) * 0.02)[:, :c2 + c3]], 1).contiguous()
        bias = torch.randn(cout, device=DEV, dtype=torch.float16)
        fns = [lambda x=x: ctx.conv3x3_plus(x, wplus, cout, x2, x3, bias=bias, gn_groups=32) for x in xs]
        flops = 2.0 * b * h * h * cout * (cin * 9 + c2 + c3)
        times = {c: [] for c in codes}
        ok = {}
        for c in codes:
            ctx.force_tile(c)
            try:
                fns[0]()
                ok[c] = True
            except hip.FieError:
                ok[c] = False
        torch.cuda.synchronize()
        iters = max(4, min(40, int(0.02 / (flops / 0.8e15))))
        for _ in range(5):
            for c in codes:
                if ok[c]:
                    ctx.force_tile(c)
                    times[c].append(time_rot(fns, iters))
        ctx.force_tile(0)
        cells = [f"{c}: n/a" if not ok[c] else f"{c}: {statistics.median(times[c]) * 1e6:7.1f} us {flops / statistics.median(times[c]) / 1e12:6.0f} TF/s" for c in codes]
        print(f"conv+1x1 B={b} {h}x{h} {cin}->{cout} side {c2}+{c3}  " + "  ".join(cells), flush=True)


def time_rot(fns, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def bench(codes, shapes):
    for b, h, cin, cout in shapes:
        copies = max(2, min(6, int(600e6 / (b * h * h * cin * 2)) + 1))
        xs = [torch.randn(b, h, h, cin, device=DEV, dtype=torch.float16) for _ in range(copies)]
        wp = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
        bias = torch.randn(cout, device=DEV, dtype=torch.float16)
        out = torch.empty(b, h, h, cout, device=DEV, dtype=torch.float16)
        if "--resgn" in sys.argv and cin == cout:           # the resnet's conv2: + identity shortcut, GroupNorm sums for the next block's norm1
            res = torch.randn(b, h, h, cout, device=DEV, dtype=torch.float16)
            fns = [lambda x=x: ctx.conv3x3(x, wp, cout, out=out, bias=bias, residual=res, gn_groups=32) for x in xs]
        elif "--resgn" in sys.argv:                         # conv1: GroupNorm sums for norm2
            fns = [lambda x=x: ctx.conv3x3(x, wp, cout, out=out, bias=bias, gn_groups=32) for x in xs]
        else:
            fns = [lambda x=x: ctx.conv3x3(x, wp, cout, out=out, bias=bias) for x in xs]
        flops = 2.0 * b * h * h * cout * cin * 9
        times = {c: [] for c in codes}
        names = {}
        for c in codes:                                   # first use + eligibility
            ctx.force_tile(c)
            try:
                fns[0]()
                names[c] = hip.last_gemm_kernel(ctx).split(" (")[0]
            except hip.FieError:
                names[c] = None
        torch.cuda.synchronize()
        iters = max(4, min(40, int(0.02 / (flops / 0.8e15))))
        for _ in range(5):                                # interleaved rounds
            for c in codes:
                if names[c] is None:
                    continue
                ctx.force_tile(c)
                times[c].append(time_rot(fns, iters))
        ctx.force_tile(0)
        cells = []
        for c in codes:
            if names[c] is None:
                cells.append(f"{c}: n/a")
            else:
                t = statistics.median(times[c])
                cells.append(f"{c}: {t * 1e6:7.1f} us {flops / t / 1e12:6.0f} TF/s ({flops / t / 2.5e15:.3f})")
        print(f"conv B={b} {h}x{h} {cin}->{cout:4d}  " + "  ".join(cells), flush=True)


def stamps(shapes, code=73):
    """Where a K-step of the halo kernel spends its cycles (code 73: v1, 74: v2): per-wave s_memtime sums, median over blocks, per K-step."""
    for b, h, cin, cout in shapes:
        x = torch.randn(b, h, h, cin, device=DEV, dtype=torch.float16)
        wp = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
        ntile = b * (h // 16) ** 2 * ((cout + 127) // 128)
        nblk = ntile if code == 73 else min(ntile, 256) // ((cout + 127) // 128) * ((cout + 127) // 128)
        nslot = 8 if code == 73 else 12
        buf = torch.zeros(nblk * 8 * nslot, device=DEV, dtype=torch.int32)
        ctx.gemm_stamps(buf)
        ctx.force_tile(code)
        ctx.conv3x3(x, wp, cout)
        ctx.conv3x3(x, wp, cout)
        torch.cuda.synchronize()
        ctx.force_tile(0)
        ctx.gemm_stamps(None)
        s = buf.view(nblk, 8, nslot).float()
        nk = 9 * cin // 64 * (ntile // nblk)
        med = s.median(dim=0).values            # [wave][segment]
        names = ["DMA issue", "reads", "wait+barrier", "MFMA issue", "barrier", "prologue", "epilogue / flush", "tile-end math"]
        for grp, ws in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
            m = med[ws].mean(dim=0)
            per = ", ".join(f"{names[i]} {m[i].item() / nk:6.0f}" for i in range(5))
            print(f"stamps code {code} B={b} {h}x{h} {cin}->{cout} {grp}: per K-step: {per} | prologue {m[5].item():.0f} epilogue / flush {m[6].item():.0f} tile-end math {m[7].item() / (ntile // nblk):.0f} per tile; K loop {sum(m[i].item() for i in range(5)):.0f} cycles over {ntile // nblk} tiles", flush=True)
            if nslot == 12:
                tiles, nch = ntile // nblk, cin // 64
                cnt = [9 * (tiles - 1), 9 * ((nch - 2) * tiles + 1), 9 * (tiles - 1), 9]       # K-steps per kind: first (tiles 2..), middle (+ the first tile's first chunk), last with prefetch, last without
                print("      whole K-steps by chunk kind: " + ", ".join(f"{nm} {m[8 + k].item() / max(cnt[k], 1):.0f}" for k, nm in enumerate(["first (stores)", "middle", "last + next tile's prefetch", "last"])), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    codes = [int(c) for c in args[0].split(",")] if args else [0, 71, 52, 96, 81]
    shapes = SHAPES[:3] if "--quick" in sys.argv else SHAPES
    for c in codes:
        if 71 <= c <= 76:
            check(c)
    if "--plus" in sys.argv:
        check_plus(72)
        bench_plus([0, 72, 52, 96, 42])
        sys.exit(0)
    if "--stamps" in sys.argv:
        stamps(shapes[:3], 73)
        stamps(shapes[:3], 74)
    bench(codes, shapes)
