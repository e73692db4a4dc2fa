#!/usr/bin/env python3
"""Single-image driver -- drop-in for /root/reference/run_single_image.py (same flags and output layout) on the
MI355X-native `FastEditor`.

    python run_single_image.py --image path/to/image.jpg --prompt "a rusty bicycle"

`--compute_metrics` reports the metrics this build restates (SSIM, PSNR, MSE); LPIPS / CLIP score / DINO need
checkpoints that cannot be fetched offline and are reported as unavailable instead of silently skipped."""
import argparse
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")    # before HIP initialises; see fie_amd.py
import sys
import time
from datetime import datetime

from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build_parser():
    p = argparse.ArgumentParser(description="Fast image editing on a single image")
    a = p.add_argument
    a("--image", type=str, required=True, help="Path to input image")
    a("--prompt", type=str, required=True, help="Editing prompt")
    a("--model", type=str, default="sdxl", choices=["sdxl", "ssd-1b"],
      help="Model to use: sdxl (full quality, ~6GB) or ssd-1b (faster, ~4GB)")
    a("--negative_prompt", type=str, default="", help="Negative prompt")
    a("--steps", type=int, default=4, help="Number of inference steps")
    a("--guidance", type=float, default=1.5, help="Guidance scale")
    a("--control_scale", type=float, default=0.5, help="ControlNet conditioning scale")
    a("--canny_low", type=int, default=100, help="Canny low threshold")
    a("--canny_high", type=int, default=200, help="Canny high threshold")
    a("--seed", type=int, default=None, help="Random seed")
    a("--output_dir", type=str, default="outputs", help="Output directory")
    a("--no_cpu_offload", action="store_true", help="Disable CPU offloading (faster but needs more VRAM)")
    a("--quality_mode", action="store_true", help="Maximum quality mode (fp32, full ControlNet) - A100 recommended")
    a("--full_precision", action="store_true", help="Use fp32 instead of fp16 (better quality, 2x VRAM)")
    a("--full_controlnet", action="store_true", help="Use full-size ControlNet instead of small variant")
    a("--compute_metrics", action="store_true", help="Compute metrics")
    a("--show_plot", action="store_true", help="Show comparison plot")
    a("--strength", type=float, default=None, help="[additive] img2img strength (default: FastEditor.edit's 0.80)")
    a("--weights_dir", type=str, default=None, help="[additive] local diffusers-layout weights directory")
    return p


def save_plot(path, source_img, edited_img, model, prompt):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, axes = plt.subplots(1, 2, figsize=(12, 6))
    axes[0].imshow(source_img)
    axes[0].set_title("Source Image")
    axes[1].imshow(edited_img)
    axes[1].set_title(f'Edited Image ({model.upper()})\n"{prompt}"')
    for ax in axes:
        ax.axis("off")
    plt.tight_layout()
    plt.savefig(path, dpi=150, bbox_inches="tight")
    plt.close(fig)
    print(f"      Saved comparison plot to: {path}")


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.quality_mode:
        args.full_precision = args.full_controlnet = args.no_cpu_offload = True
        print("[Quality Mode] Enabled: fp32 + full ControlNet + no CPU offload")
    if not os.path.exists(args.image):
        print(f"Error: Image not found at {args.image}")
        return
    model_suffix = f"{args.model}_{'fp32' if args.full_precision else 'fp16'}"
    edited_dir = os.path.join(args.output_dir, "single", "edited", model_suffix)
    comparisons_dir = os.path.join(args.output_dir, "single", "comparisons", model_suffix)
    os.makedirs(edited_dir, exist_ok=True)
    os.makedirs(comparisons_dir, exist_ok=True)

    print(f"\n[1/4] Loading image from {args.image}")
    source_img = Image.open(args.image).convert("RGB")
    print(f"      Image size: {source_img.size}")

    print("\n[2/4] Initializing FastEditor...")
    from src.pipeline import FastEditor
    editor = FastEditor(model_name=args.model, device="cuda", enable_cpu_offload=not args.no_cpu_offload,
                        use_full_precision=args.full_precision, use_full_controlnet=args.full_controlnet,
                        weights_dir=args.weights_dir)
    mem = editor.get_memory_usage()
    print(f"      GPU Memory: {mem['allocated_gb']:.2f}GB allocated, {mem['reserved_gb']:.2f}GB reserved")

    print("\n[3/4] Running image editing...")
    print(f"      Prompt: {args.prompt}")
    print(f"      Steps: {args.steps}, Guidance: {args.guidance}, Control Scale: {args.control_scale}")
    extra = {} if args.strength is None else {"strength": args.strength}
    t0 = time.time()
    edited_img = editor.edit(image=source_img, prompt=args.prompt, negative_prompt=args.negative_prompt,
                             num_inference_steps=args.steps, guidance_scale=args.guidance,
                             controlnet_conditioning_scale=args.control_scale, canny_low_threshold=args.canny_low,
                             canny_high_threshold=args.canny_high, seed=args.seed, **extra)
    elapsed = time.time() - t0
    print(f"      Editing completed in {elapsed:.2f} seconds")
    mem = editor.get_memory_usage()
    print(f"      GPU Memory: {mem['allocated_gb']:.2f}GB allocated, {mem['reserved_gb']:.2f}GB reserved")

    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    output_path = os.path.join(edited_dir, f"edited_{stamp}.jpg")
    edited_img.save(output_path)
    print(f"\n      Saved edited image to: {output_path}")

    if args.compute_metrics:
        print("\n[4/4] Computing metrics...")
        from src.metrics import MetricsCalculator
        calc = MetricsCalculator(device="cuda")
        metrics = calc.calculate_all_metrics(source_img=source_img, edited_img=edited_img, prompt=args.prompt)
        labels = [("ssim", "SSIM (structure preservation):  ", ".4f", ""), ("lpips", "LPIPS (perceptual distance):    ", ".4f", ""),
                  ("psnr", "PSNR (signal quality):          ", ".2f", " dB"), ("mse", "MSE (pixel difference):         ", ".6f", ""),
                  ("clip_score", "CLIP Score (text alignment):    ", ".2f", "")]
        fmt = lambda k, f: "unavailable offline" if metrics.get(k) is None else format(metrics[k], f)
        print("\n      Metrics:")
        for k, label, f, unit in labels:
            print(f"        {label}{fmt(k, f)}{unit if metrics.get(k) is not None else ''}")
        metrics_path = os.path.join(edited_dir, f"metrics_{stamp}.txt")
        with open(metrics_path, "w") as fh:
            fh.write(f"Image: {args.image}\nPrompt: {args.prompt}\nModel: {args.model}\nTime: {elapsed:.2f}s\n\nMetrics:\n")
            for k, label, f, unit in labels:
                fh.write(f"  {k}: {fmt(k, f)}\n")
        print(f"      Saved metrics to: {metrics_path}")
        print("\n      Saving comparison plot...")
        save_plot(os.path.join(comparisons_dir, f"comparison_{stamp}.png"), source_img, edited_img, args.model, args.prompt)
        calc.clear_memory()
    elif args.show_plot:
        print("\n      Saving comparison plot...")
        save_plot(os.path.join(comparisons_dir, f"comparison_{stamp}.png"), source_img, edited_img, args.model, args.prompt)
    editor.clear_memory()
    print("\nDone!")


if __name__ == "__main__":
    main()
