#!/usr/bin/env python3
"""Evaluation harness -- drop-in for /root/reference/evaluate.py (same flags, same `metrics.csv` columns and
`summary.json` layout, reference :193-271), on the metrics this build restates (`src/metrics.py`: SSIM, PSNR, MSE).
LPIPS / CLIP score / DINO distance need hub checkpoints: their cells are empty in the CSV and `null` in the JSON.

Under `torch.distributed.run` the mapping entries are sharded image-parallel over the ranks (fie_amd.dist) and rank 0
writes the merged files -- the "gather of metrics" of SURVEY.md 8e.

    python evaluate.py --outputs_dir outputs/batch/edited/ssd-1b_fp16
"""
import argparse
import csv
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")    # before HIP initialises; see fie_amd.py
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

METRICS = ("ssim", "lpips", "clip_score", "psnr", "mse", "dino_distance")
FIELDS = ["image_id", "image_path", "editing_type_id", "editing_prompt", *METRICS]
KNOWN_SUFFIXES = ("sdxl_fp32", "sdxl_fp16", "ssd-1b_fp32", "ssd-1b_fp16")


def build_parser():
    p = argparse.ArgumentParser(description="Evaluate edited images")
    p.add_argument("--mapping_file", type=str, default="data/PIE-Bench_v1/mapping_file.json", help="Path to PIE-Bench mapping file")
    p.add_argument("--source_dir", type=str, default="data/PIE-Bench_v1/annotation_images", help="Directory containing source images")
    p.add_argument("--outputs_dir", type=str, required=True,
                   help="Directory containing edited images (e.g., outputs/batch/edited/sdxl_fp32)")
    p.add_argument("--results_file", type=str, default=None,
                   help="Output CSV file for metrics (auto-detected from outputs_dir if not specified)")
    p.add_argument("--summary_file", type=str, default=None,
                   help="Output JSON file for summary statistics (auto-detected from outputs_dir if not specified)")
    p.add_argument("--device", type=str, default="cuda", help="Device to use for metrics computation")
    return p


def default_paths(args):
    tail = os.path.basename(args.outputs_dir.rstrip("/"))
    sub = f"{tail}/" if args.outputs_dir.rstrip("/").endswith(KNOWN_SUFFIXES) else ""
    args.results_file = args.results_file or f"results/{sub}metrics.csv"
    args.summary_file = args.summary_file or f"results/{sub}summary.json"
    for path in (args.results_file, args.summary_file):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)


def _stats(values, with_median):
    vals = [v for v in values if v is not None]
    if not vals:
        return {"mean": None, "std": None, **({"median": None} if with_median else {})}
    out = {"mean": float(np.mean(vals)), "std": float(np.std(vals))}
    if with_median:
        out["median"] = float(np.median(vals))
    return out


def summarize(rows):
    """reference :201-268: overall mean/std/median per metric + per-category mean/std and count."""
    summary = {"total_images": len(rows), "overall": {m: _stats([r[m] for r in rows], True) for m in METRICS}, "by_category": {}}
    cats = {}
    for r in rows:
        cats.setdefault(r["editing_type_id"], []).append(r)
    for cat, rs in cats.items():
        summary["by_category"][cat] = {"count": len(rs), **{m: _stats([r[m] for r in rs], False) for m in METRICS}}
    return summary


def evaluate_entries(entries, args, calc, progress=None):
    rows, skipped = [], 0
    for index, image_id, entry in (progress(entries) if progress else entries):
        rel = entry["image_path"]
        src, out = os.path.join(args.source_dir, rel), os.path.join(args.outputs_dir, rel)
        if not (os.path.exists(out) and os.path.exists(src)):
            skipped += 1
            continue
        try:
            size = (512, 512)
            a, b = Image.open(src).convert("RGB"), Image.open(out).convert("RGB")
            a = a if a.size == size else a.resize(size, Image.LANCZOS)
            b = b if b.size == size else b.resize(size, Image.LANCZOS)
            m = calc.calculate_all_metrics(source_img=a, edited_img=b, prompt=entry.get("editing_prompt", ""))
            rows.append(dict(index=index, image_id=image_id, image_path=rel, editing_type_id=entry.get("editing_type_id", "unknown"),
                             editing_prompt=entry.get("editing_prompt", ""), **{k: m[k] for k in METRICS}))
        except Exception as e:  # per-image isolation (reference :179-182)
            print(f"\n      Error processing {image_id}: {e}")
            skipped += 1
    return rows, skipped


def main(argv=None):
    args = build_parser().parse_args(argv)
    import fie_amd  # noqa: F401
    from fie_amd import dist as fdist
    from src.metrics import MetricsCalculator
    rank, local, world = fdist.init()
    say = print if rank == 0 else (lambda *a, **k: None)
    default_paths(args)
    say(f"\n[1/4] Loading mapping file from {args.mapping_file}")
    with open(args.mapping_file) as fh:
        mapping = json.load(fh)
    say(f"      Total entries: {len(mapping)}")
    say(f"\n[2/4] Scanning outputs in {args.outputs_dir}")
    say(f"\n[3/4] Initializing metrics on {args.device}...")
    device = args.device if world == 1 or not args.device.startswith("cuda") else f"cuda:{local}"
    calc = MetricsCalculator(device=device)
    mine = fdist.shard([(i, k, e) for i, (k, e) in enumerate(mapping.items())], rank, world)
    progress = None
    if rank == 0:
        try:
            from tqdm import tqdm
            progress = lambda it: tqdm(it, desc="Evaluating")
        except ImportError:
            pass
    rows, skipped = evaluate_entries(mine, args, calc, progress)
    gathered = fdist.gather_results(dict(rows=rows, skipped=skipped))
    if rank != 0:
        return
    rows = sorted((r for g in gathered for r in g["rows"]), key=lambda r: r["index"])
    skipped = sum(g["skipped"] for g in gathered)
    print(f"\n      Processed: {len(rows)} images\n      Skipped:   {skipped} images")
    if not rows:
        print("\n      No images were processed. Exiting.")
        return
    print("\n[4/4] Saving results...")
    with open(args.results_file, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=FIELDS, extrasaction="ignore")
        w.writeheader()
        w.writerows([{k: ("" if v is None else v) for k, v in r.items()} for r in rows])
    print(f"      Saved detailed metrics to: {args.results_file}")
    summary = summarize(rows)
    with open(args.summary_file, "w") as fh:
        json.dump(summary, fh, indent=2)
    print(f"      Saved summary statistics to: {args.summary_file}")
    fmt = lambda s, spec: "unavailable offline" if s["mean"] is None else f"{format(s['mean'], spec)} ± {format(s['std'], spec)}"
    bar = "=" * 60
    print(f"\n{bar}\nEVALUATION SUMMARY\n{bar}\n\nTotal Images Evaluated: {len(rows)}\n\nOverall Metrics:")
    labels = (("ssim", "SSIM:      ", ".4f"), ("lpips", "LPIPS:     ", ".4f"), ("psnr", "PSNR:      ", ".2f"), ("mse", "MSE:       ", ".6f"),
              ("clip_score", "CLIP Score:", ".2f"), ("dino_distance", "DINO Dist.:", ".4f"))
    for k, label, spec in labels:
        print(f"  {label} {fmt(summary['overall'][k], spec)}")
    print("\nMetrics by Category:")
    for cat in sorted(summary["by_category"]):
        c = summary["by_category"][cat]
        print(f"\n  Category {cat} ({c['count']} images):")
        for k, label, spec in labels:
            print(f"    {label} {fmt(c[k], spec)}")
    print(f"\n{bar}\n\nDone!")
    calc.clear_memory()


if __name__ == "__main__":
    main()
