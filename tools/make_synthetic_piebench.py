#!/usr/bin/env python3
"""Write a PIE-Bench-SHAPED synthetic dataset: `mapping_file.json` + one 512x512 JPEG per item under
the 700 real relative paths, with the real prompts / editing-type ids from tests/golden/pie_bench_items.csv and seeded synthetic
pictures (low-frequency colour fields + filled shapes, so Canny(100,200) finds real edges).

    python tools/make_synthetic_piebench.py --out data/PIE-Bench_v1 [--num 700]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_batch.py --model ssd-1b --no_cpu_offload --seed 42
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_item_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="data/PIE-Bench_v1")
    ap.add_argument("--num", type=int, default=700)
    args = ap.parse_args()
    with open(os.path.join(ROOT, "tests", "golden", "pie_bench_items.csv")) as f:
        items = list(csv.DictReader(f))[: args.num]
    mapping = {}
    for i, it in enumerate(items):
        path = os.path.join(args.out, "annotation_images", it["image_path"])
        os.makedirs(os.path.dirname(path), exist_ok=True)
        synth_item_image(i).save(path, quality=95)
        mapping[it["image_id"]] = {"image_path": it["image_path"], "editing_prompt": it["editing_prompt"],
                                   "editing_type_id": it["editing_type_id"]}
    with open(os.path.join(args.out, "mapping_file.json"), "w") as f:
        json.dump(mapping, f, indent=1)
    print(f"wrote {len(mapping)} items under {args.out}")


if __name__ == "__main__":
    main()
