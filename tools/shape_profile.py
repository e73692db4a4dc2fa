#!/usr/bin/env python3
"""Per-shape, in-situ roofline table of ONE edit.

  run     rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/shape_profile.py run DIR/oplog.txt
          One serial edit of the bench workload issued EAGERLY on ONE stream (so dispatch order == launch order) after two
          warm-up edits (the first tunes the tiles), with the library's launch log on (include/fie.h: fie_debug_oplog):
          every launch leaves "kernel|blocks|threads|lds|description", GEMM / conv / attention / norm ops describe their
          problem (shape, tile code, algorithmic FLOPs or bytes); `#stage` lines mark the stage boundaries of pipe.run_device.
  report  python3 tools/shape_profile.py report DIR OUT.md
          Pairs the log lines with the LAST launches of the kernel trace (by order; grid sizes are cross-checked) and writes
          one row per distinct problem: launches, in-network average / min / total us, TFLOP/s or GB/s, and the fraction of its
          bound: MFMA dense fp16 2.5 PFLOP/s for GEMM / conv / attention, HBM 8 TB/s for the norm kernels.

The in-situ durations include cold weights and whatever the previous kernel left in the caches, but NOT the overlap of the
two-stream graph (one stream here), so the sum is the kernel time of an edit, not its wall time."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_TF, PEAK_GB = 2500.0, 8000.0
# LDS-fill traffic of the tiled kernels: every K-step of 64 stages (BM + BN) x 128 B per tile out of the XCD's L2.  MI355X_MICROARCH.md ("Indexed rows: gather into
# LDS") measures 66-73 GB/s per CU = 16.8-18.8 TB/s chip-wide for such fills: the bound of the small tiles (BM BN / (BM + BN) FLOP per staged byte).
TILE = {1: (128, 128), 2: (128, 64), 3: (64, 64), 42: (128, 64), 43: (64, 64), 44: (128, 64), 46: (64, 64), 47: (128, 96), 48: (128, 80), 51: (128, 128), 52: (128, 128),
        54: (192, 128), 61: (256, 256), 62: (256, 128), 63: (256, 320), 64: (256, 320), 81: (256, 256), 82: (256, 256), 95: (128, 128), 96: (256, 128)}
FILL_TB = 17.8


def staged_bytes(desc):
    """LDS-fill bytes of one GEMM / conv launch from its log description, or None."""
    m = re.search(r"M=(\d+) N=(\d+) K=(\d+).* code=(\d+)", desc)
    if not m:
        return None
    M, N, K, code = (int(v) for v in m.groups())
    if code in (71, 72):          # halo-resident conv (csrc/conv_halo.hip): per K-step 128 weight rows x 128 B + a ninth of a 41 KB halo, per (16x16 patch, 128 channels) tile
        return (M // 256) * -(-N // 128) * -(-K // 64) * (128 * 128.0 + 41984.0 / 9)
    t = TILE.get(code % 1000)
    if t is None:
        return None
    bm, bn = t
    return -(-M // bm) * -(-N // bn) * -(-K // 64) * (bm + bn) * 128.0


def run(out_path, model="ssd-1b"):
    import fie_amd  # noqa: F401
    import torch
    from bench import synth_item_image
    from src.pipeline import FastEditor
    kw = {"weight_dtype": os.environ["FIE_PROFILE_WEIGHTS"]} if os.environ.get("FIE_PROFILE_WEIGHTS") else {}      # "f8e4m3": BASELINE config 5
    ed = FastEditor(model_name=model, use_full_controlnet=True, enable_cpu_offload=False, **kw)
    pipe, ctx = ed.pipe, ed.pipe.ctx
    pipe.fork_streams = False
    pipe.use_graph = False
    img = synth_item_image(3).resize((1024, 1024))
    ctrl = ed.preprocess_image(img)
    job = pipe.prepare("a photo of a [red] house", "", img, ctrl, 0.5, 4, 1.5, 0.5, torch.Generator().manual_seed(42))
    ctx.autotune(1 if pipe.autotune else 0)
    pipe.run_device(job)                     # tunes
    torch.cuda.synchronize()
    ctx.autotune(2 if pipe.autotune else 0)
    pipe.run_device(job)
    torch.cuda.synchronize()
    if os.environ.get("FIE_PROFILE_PROBE"):      # timing-only probes of the LDS-DMA GEMM / conv kernels (include/fie.h: fie_debug_gemm_probe; outputs are wrong):
        ctx.gemm_probe(int(os.environ["FIE_PROFILE_PROBE"]))      # 2 = every tile fetches tile (0,0)'s operands: the in-network times with all operand loads hitting L2
    ctx.oplog(True)
    pipe.run_device(job)
    torch.cuda.synchronize()
    lines = ctx.oplog_read()
    ctx.oplog(False)
    ctx.gemm_probe(0)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"{sum(1 for l in lines if not l.startswith('#'))} launches logged -> {out_path}")


def _ours(name):
    return not (name.startswith("void at::") or name.startswith("at::") or "at::native" in name or name.startswith("__amd_rocclr"))


def report(d, out_md):
    oplog = [l.rstrip("\n") for l in open(os.path.join(d, "oplog.txt")) if l.strip()]
    trace_files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for tf in trace_files:
        rows += list(csv.DictReader(open(tf)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if _ours(r["Kernel_Name"])]
    launches = [l for l in oplog if not l.startswith("#")]
    n = len(launches)
    assert len(rows) >= n, f"trace has {len(rows)} library launches, log has {n}"
    rows = rows[-n:]
    stage, per, stage_us, bad = "start", {}, {}, 0
    it = iter(rows)
    # a stage mark names the stage that ENDS there (pipe._mark): collect, then label backwards
    seq = []
    for l in oplog:
        if l.startswith("#"):
            seq.append(("mark", l[1:]))
            continue
        r = next(it)
        sym, blocks, threads, lds, desc = l.split("|", 4)
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        if grid != int(blocks) * int(threads):
            bad += 1
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        seq.append(("k", sym, desc, us, r["Kernel_Name"], int(blocks), int(lds)))
    assert bad == 0, f"{bad} log lines do not match the trace row they were paired with (grid size)"
    cur = []
    for item in seq:
        if item[0] == "mark":
            for k in cur:
                stage_us[item[1]] = stage_us.get(item[1], 0.0) + k[3]
            cur = []
        else:
            cur.append(item)
            _, sym, desc, us, kname, blocks, lds = item
            short = re.sub(r"\(anonymous namespace\)::|void |fie_gemm::", "", kname).split("(")[0][:60]
            key = (re.sub(r" (flop|bytes)=[0-9.e+]+", "", desc) or short, short, blocks, lds)
            e = per.setdefault(key, dict(n=0, us=0.0, mn=1e30, flop=0.0, bytes=0.0))
            e["n"] += 1
            e["us"] += us
            e["mn"] = min(e["mn"], us)
            m = re.search(r"flop=([0-9.e+]+)", desc)
            if m:
                e["flop"] = float(m.group(1))
            m = re.search(r"bytes=([0-9.e+]+)", desc)
            if m:
                e["bytes"] = float(m.group(1))
    tot = sum(e["us"] for e in per.values())
    with open(out_md, "w") as f:
        f.write(f"# Per-shape in-situ roofline of ONE edit (SSD-1B-A1 + ControlNet-full, 1024^2, 2 evals, CFG batch 2)\n\n"
                f"`tools/shape_profile.py`: one eager single-stream edit under `rocprofv3 --kernel-trace`, launch log paired with the trace by order "
                f"(grid sizes cross-checked). {n} launches, {tot / 1e3:.2f} ms of kernel time. Bounds: MFMA dense fp16 {PEAK_TF:.0f} TFLOP/s "
                f"(GEMM / conv / attention rows, executed FLOPs), HBM {PEAK_GB:.0f} GB/s (norm rows, algorithmic bytes).\n\n")
        f.write("## By stage (kernel time, single stream)\n\n| stage | ms |\n|---|---:|\n")
        for k, v in stage_us.items():
            f.write(f"| {k} | {v / 1e3:.2f} |\n")
        f.write("\nLast column: LDS-fill traffic of the tiled kernels (tiles x K-steps x (BM + BN) x 128 B) over the launch's time, against the ~17.8 TB/s at which the 256 CUs can fill "
                "LDS out of their L2s (MI355X_MICROARCH.md: 66-73 GB/s per CU): the bound the small-tile GEMMs run into long before the MFMA peak.\n")
        f.write("\n## By problem, largest first\n\n| problem | kernel | blocks | launches | avg us | min us | total us | % | rate | frac of bound | LDS fill TB/s (frac of 17.8) |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        classes = {}
        for (desc, short, blocks, lds), e in sorted(per.items(), key=lambda kv: -kv[1]["us"]):
            avg = e["us"] / e["n"]
            if e["flop"]:
                rate, frac = f"{e['flop'] / avg / 1e6:.0f} TF/s", e["flop"] / avg / 1e6 / PEAK_TF
            elif e["bytes"]:
                rate, frac = f"{e['bytes'] / avg / 1e3:.0f} GB/s", e["bytes"] / avg / 1e3 / PEAK_GB
            else:
                rate, frac = "", None
            sb = staged_bytes(desc)
            fill = "" if sb is None else f"{sb / avg / 1e6:.1f} ({sb / avg / 1e6 / FILL_TB:.2f})"
            f.write(f"| {desc} | `{short}` | {blocks} | {e['n']} | {avg:.1f} | {e['mn']:.1f} | {e['us']:.0f} | {100 * e['us'] / tot:.1f} | {rate} | "
                    f"{'' if frac is None else f'{frac:.3f}'} | {fill} |\n")
            cls = desc.split(" ")[0] if desc else short
            m = re.search(r"M=(\d+)", desc)
            if cls in ("gemm", "conv") and m:
                cls += " M<=2048" if int(m.group(1)) <= 2048 else " M<=8192" if int(m.group(1)) <= 8192 else " M>8192"
            c = classes.setdefault(cls, dict(us=0.0, flop=0.0, bytes=0.0, n=0))
            c["us"] += e["us"]
            c["flop"] += e["flop"] * e["n"]
            c["bytes"] += e["bytes"] * e["n"]
            c["n"] += e["n"]
        f.write("\n## By class\n\n| class | launches | total ms | % | aggregate rate |\n|---|---:|---:|---:|---:|\n")
        for cls, c in sorted(classes.items(), key=lambda kv: -kv[1]["us"]):
            rate = f"{c['flop'] / c['us'] / 1e6:.0f} TF/s" if c["flop"] else f"{c['bytes'] / c['us'] / 1e3:.0f} GB/s" if c["bytes"] else ""
            f.write(f"| {cls} | {c['n']} | {c['us'] / 1e3:.2f} | {100 * c['us'] / tot:.1f} | {rate} |\n")
    print(open(out_md).read())


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], *(sys.argv[3:4]))
    else:
        report(sys.argv[2], sys.argv[3])
