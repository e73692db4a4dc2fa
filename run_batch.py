#!/usr/bin/env python3
"""PIE-Bench batch driver -- drop-in for /root/reference/run_batch.py (same flags, defaults, output layout and
summary), running the MI355X-native `FastEditor`.  New: launched under `torch.distributed.run` with N processes it
shards the selected entries image-parallel over N GPUs (fie_amd.dist) and rank 0 prints the merged summary.

    python run_batch.py --num_images 50 --editing_types 0 1 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run_batch.py --model ssd-1b

Additive flags (the reference has none of them): --strength, --weights_dir, --device, --results_json.
"""
import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")    # before HIP initialises; see fie_amd.py
import sys
import time

from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def load_mapping_file(mapping_path):
    with open(mapping_path, "r") as fh:
        return json.load(fh)


def safe_join(base_dir, user_path):
    """Join `user_path` under `base_dir`; absolute paths and `..` escapes raise ValueError (reference :25-41,
    including its string-prefix containment check)."""
    rel = os.path.normpath(user_path)
    if os.path.isabs(rel) or rel.startswith(".."):
        raise ValueError(f"Invalid path: {rel}")
    root = os.path.abspath(base_dir)
    joined = os.path.abspath(os.path.join(base_dir, rel))
    if not joined.startswith(root):
        raise ValueError(f"Path traversal detected: {rel}")
    return joined


def build_parser():
    p = argparse.ArgumentParser(description="Batch image editing on PIE-Bench")
    a = p.add_argument
    a("--mapping_file", type=str, default="data/PIE-Bench_v1/mapping_file.json", help="Path to PIE-Bench mapping file")
    a("--source_dir", type=str, default="data/PIE-Bench_v1/annotation_images", help="Directory containing source images")
    a("--output_dir", type=str, default="outputs", help="Output directory")
    a("--model", type=str, default="sdxl", choices=["sdxl", "ssd-1b"],
      help="Model to use: sdxl (full quality, ~6GB) or ssd-1b (faster, ~4GB)")
    a("--num_images", type=int, default=None, help="Number of images to process (default: all)")
    a("--editing_types", nargs="+", type=str, default=None, help="Filter by editing type IDs (e.g., 0 1 2)")
    a("--image_ids", nargs="+", type=str, default=None, help="Process specific image IDs")
    a("--steps", type=int, default=4, help="Number of inference steps")
    a("--guidance", type=float, default=1.5, help="Guidance scale")
    a("--control_scale", type=float, default=0.5, help="ControlNet conditioning scale")
    a("--canny_low", type=int, default=100, help="Canny low threshold")
    a("--canny_high", type=int, default=200, help="Canny high threshold")
    a("--seed", type=int, default=None, help="Random seed")
    a("--negative_prompt", type=str, default="", help="Negative prompt")
    a("--no_cpu_offload", action="store_true", help="Disable CPU offloading (faster but needs more VRAM)")
    a("--quality_mode", action="store_true", help="Maximum quality mode (fp32, full ControlNet) - A100 recommended")
    a("--full_precision", action="store_true", help="Use fp32 instead of fp16 (better quality, 2x VRAM)")
    a("--full_controlnet", action="store_true", help="Use full-size ControlNet instead of small variant")
    a("--skip_existing", action="store_true", help="Skip images that already have outputs")
    a("--save_comparisons", action="store_true", help="Save side-by-side comparison images")
    # additive
    a("--strength", type=float, default=None, help="[additive] img2img strength (default: FastEditor.edit's 0.80)")
    a("--weights_dir", type=str, default=None, help="[additive] local diffusers-layout weights directory")
    a("--results_json", type=str, default=None, help="[additive] write the merged per-image rows + summary here")
    a("--batch_size", type=int, default=1, help="[additive] images per device job (UNet / ControlNet / CLIP batched; the reference "
                                                  "is serial = 1); per-image time = job time / batch")
    a("--in_flight", type=int, default=1, help="[additive] edits in flight per GPU (worker threads, one hipGraph slot each); "
                                                 "per-image times then overlap, throughput is the summary's wall-clock rate")
    return p


def select_entries(mapping, args, say=print):
    """Reference :115-140: explicit ids win; else filter by type, then truncate to --num_images."""
    if args.image_ids:
        say("\n[2/3] Filtering by image IDs...")
        chosen = [(i, mapping[i]) for i in args.image_ids if i in mapping]
        say(f"      Selected {len(chosen)} images by ID")
        return chosen
    if args.editing_types:
        say(f"\n[2/3] Filtering by editing types: {args.editing_types}")
        chosen = [(i, e) for i, e in mapping.items() if e.get("editing_type_id") in args.editing_types]
        say(f"      Selected {len(chosen)} images by type")
    else:
        chosen = list(mapping.items())
        say(f"\n[2/3] Processing all images: {len(chosen)}")
    if args.num_images and args.num_images < len(chosen):
        chosen = chosen[: args.num_images]
        say(f"      Limited to first {args.num_images} images")
    return chosen


def save_comparison(path, source_img, edited_img, model, prompt):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    os.makedirs(os.path.dirname(path), exist_ok=True)
    fig, axes = plt.subplots(1, 2, figsize=(12, 6))
    axes[0].imshow(source_img)
    axes[0].set_title("Source Image")
    axes[1].imshow(edited_img)
    title = f'"{prompt[:60]}..."' if len(prompt) > 60 else f'"{prompt}"'
    axes[1].set_title(f"Edited ({model.upper()})\n{title}")
    for ax in axes:
        ax.axis("off")
    plt.tight_layout()
    plt.savefig(path, dpi=150, bbox_inches="tight")
    plt.close(fig)


def process_shard(editor, entries, args, edited_dir, comparisons_dir, progress=None):
    """The per-image loop (reference :176-261) over this rank's (index, image_id, entry) triples; with --in_flight > 1 the
    entries are dealt to worker threads (one graph slot each) and the per-worker results are merged."""
    n = max(1, getattr(args, "in_flight", 1))
    if n == 1:
        return _process_entries(editor, entries, args, edited_dir, comparisons_dir, progress)
    from concurrent.futures import ThreadPoolExecutor
    editor.set_in_flight(n)
    if hasattr(editor, "calibrate_in_flight"):        # untimed set-up: pick slot streams on which the n edits really overlap
        for _, _, entry in entries:
            try:
                probe = Image.open(safe_join(args.source_dir, entry["image_path"])).convert("RGB")
                extra = {} if args.strength is None else {"strength": args.strength}
                editor.calibrate_in_flight(probe, entry.get("editing_prompt") or "calibration", num_inference_steps=args.steps,
                                           guidance_scale=args.guidance, controlnet_conditioning_scale=args.control_scale,
                                           seed=args.seed, **extra)
                break
            except (OSError, ValueError, KeyError):
                continue

    def work(slot):
        editor.worker_slot(slot)
        return _process_entries(editor, entries[slot::n], args, edited_dir, comparisons_dir, progress if slot == 0 else None)

    with ThreadPoolExecutor(max_workers=n) as pool:
        parts = list(pool.map(work, range(n)))
    res = dict(processed=0, skipped=0, failed=0, total_time=0.0, rows=[])
    for part in parts:
        for k in ("processed", "skipped", "failed", "total_time"):
            res[k] += part[k]
        res["rows"] += part["rows"]
    res["rows"].sort(key=lambda r: r["index"])
    return res


def _process_entries(editor, entries, args, edited_dir, comparisons_dir, progress=None):
    res = dict(processed=0, skipped=0, failed=0, total_time=0.0, rows=[])
    extra = {} if args.strength is None else {"strength": args.strength}
    bs = max(1, getattr(args, "batch_size", 1))
    pending = []                                   # (index, image_id, rel, output_path, source_img, prompt) awaiting one device job

    def flush():
        if not pending:
            return
        kw = dict(num_inference_steps=args.steps, guidance_scale=args.guidance, controlnet_conditioning_scale=args.control_scale,
                  canny_low_threshold=args.canny_low, canny_high_threshold=args.canny_high, seed=args.seed, **extra)
        try:
            t0 = time.time()
            if bs == 1:
                edited = [editor.edit(image=pending[0][4], prompt=pending[0][5], negative_prompt=args.negative_prompt, **kw)]
            else:
                edited = editor.edit_batch(images=[p[4] for p in pending], prompts=[p[5] for p in pending],
                                           negative_prompts=[args.negative_prompt] * len(pending), **kw)
            dt = (time.time() - t0) / len(pending)
            for (index, image_id, rel, output_path, source_img, prompt), out in zip(pending, edited):
                res["total_time"] += dt
                out.save(output_path)
                res["processed"] += 1
                res["rows"].append(dict(index=index, image_id=image_id, image_path=rel, elapsed_s=dt))
                if args.save_comparisons:
                    save_comparison(os.path.join(comparisons_dir, rel.replace(".jpg", ".png")), source_img, out, args.model, prompt)
                if res["processed"] % 10 == 0:
                    editor.clear_memory()
        except Exception as e:  # per-image isolation, as the reference does (a failing batch fails its images)
            ids = ", ".join(str(p[1]) for p in pending)
            print(f"\n      Error processing {ids} ({type(e).__name__}): {e}")
            res["failed"] += len(pending)
        pending.clear()

    for index, image_id, entry in (progress(entries) if progress else entries):
        try:
            rel = entry["image_path"]
            source_path = safe_join(args.source_dir, rel)
            output_path = os.path.join(edited_dir, rel)
            if args.skip_existing and os.path.exists(output_path):
                res["skipped"] += 1
                continue
            if not os.path.exists(source_path):
                res["failed"] += 1
                continue
            os.makedirs(os.path.dirname(output_path), exist_ok=True)
            source_img = Image.open(source_path).convert("RGB")
            prompt = entry.get("editing_prompt", "")
            if not prompt:
                res["failed"] += 1
                continue
            pending.append((index, image_id, rel, output_path, source_img, prompt))
            if len(pending) == bs:
                flush()
        except FileNotFoundError as e:
            print(f"\n      File not found for {image_id}: {e}")
            res["failed"] += 1
        except ValueError as e:
            print(f"\n      Invalid path for {image_id}: {e}")
            res["failed"] += 1
    flush()
    return res


def print_summary(tot, args, edited_dir, comparisons_dir, world, wall):
    bar = "=" * 60
    print(f"\n{bar}\nBATCH PROCESSING SUMMARY\n{bar}")
    print(f"\nProcessed:  {tot['processed']} images")
    print(f"Skipped:    {tot['skipped']} images")
    print(f"Failed:     {tot['failed']} images")
    if tot["processed"] > 0:
        print(f"\nAverage time per image: {tot['total_time'] / tot['processed']:.2f}s")
        print(f"Total time: {tot['total_time']:.2f}s ({tot['total_time'] / 60:.1f} minutes)")
        print(f"Throughput: {tot['processed'] / wall:.2f} images/sec wall-clock on {world} GPU(s)")
    else:
        print("\n⚠ WARNING: No images were successfully processed!")
        print("  Check that:")
        print(f"    - Source images exist at: {args.source_dir}")
        print(f"    - Mapping file is correct: {args.mapping_file}")
        print("    - Selected filters match available images")
    print("\nOutputs saved to:")
    print(f"  - Edited images: {edited_dir}")
    if args.save_comparisons:
        print(f"  - Comparisons: {comparisons_dir}")
    print(bar)


def main(argv=None):
    args = build_parser().parse_args(argv)
    import fie_amd  # noqa: F401
    from fie_amd import dist as fdist
    rank, local, world = fdist.init()
    say = print if rank == 0 else (lambda *a, **k: None)
    if args.quality_mode:
        args.full_precision = args.full_controlnet = args.no_cpu_offload = True
        say("[Quality Mode] Enabled: fp32 + full ControlNet + no CPU offload")
    model_suffix = f"{args.model}_{'fp32' if args.full_precision else 'fp16'}"
    edited_dir = os.path.join(args.output_dir, "batch", "edited", model_suffix)
    comparisons_dir = os.path.join(args.output_dir, "batch", "comparisons", model_suffix)
    os.makedirs(edited_dir, exist_ok=True)
    if args.save_comparisons:
        os.makedirs(comparisons_dir, exist_ok=True)

    say(f"\n[1/3] Loading mapping file from {args.mapping_file}")
    mapping = load_mapping_file(args.mapping_file)
    say(f"      Total entries in mapping file: {len(mapping)}")
    selected = select_entries(mapping, args, say)
    if not selected:
        say("\n      No images selected. Exiting.")
        return
    mine = fdist.shard([(i, k, e) for i, (k, e) in enumerate(selected)], rank, world)

    say(f"\n[3/3] Initializing FastEditor ({model_suffix})...")
    from src.pipeline import FastEditor
    import contextlib
    import io
    with contextlib.redirect_stdout(sys.stdout if rank == 0 else io.StringIO()):
        import torch
        editor = FastEditor(model_name=args.model, device="cuda" if world == 1 else f"cuda:{local % max(torch.cuda.device_count(), 1)}",
                            enable_cpu_offload=not args.no_cpu_offload, use_full_precision=args.full_precision,
                            use_full_controlnet=args.full_controlnet, weights_dir=args.weights_dir)
    if hasattr(editor, "pipe"):
        editor.pipe.set_progress_bar_config(disable=True)
    mem = editor.get_memory_usage()
    say(f"      GPU Memory: {mem['allocated_gb']:.2f}GB allocated, {mem['reserved_gb']:.2f}GB reserved")
    say(f"\n      Processing {len(selected)} images on {world} GPU(s)...")
    say(f"      Parameters: steps={args.steps}, guidance={args.guidance}, control_scale={args.control_scale}")
    if args.negative_prompt:
        say(f"      Negative prompt: {args.negative_prompt}")
    say(f"      Canny thresholds: low={args.canny_low}, high={args.canny_high}")

    progress = None
    if rank == 0:
        try:
            from tqdm import tqdm
            progress = lambda it: tqdm(it, desc="Editing")
        except ImportError:
            pass
    fdist.barrier()
    t0 = time.time()
    res = process_shard(editor, mine, args, edited_dir, comparisons_dir, progress)
    fdist.barrier()
    wall = time.time() - t0
    gathered = fdist.gather_results(res)
    if rank == 0:
        tot = fdist.merge_results(gathered)
        print_summary(tot, args, edited_dir, comparisons_dir, world, wall)
        if args.results_json:
            with open(args.results_json, "w") as fh:
                json.dump(dict(world_size=world, wall_s=wall, **tot), fh, indent=1)
        print("\nDone! Next steps:")
        print(f"  1. Review outputs: ls {edited_dir}")
        print(f"  2. Run evaluation: python evaluate.py --outputs_dir {edited_dir}")
    editor.clear_memory()


if __name__ == "__main__":
    main()
