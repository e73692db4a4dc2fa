"""CPU suite: the C-ABI shared library loads and exports every symbol include/fie.h declares (no GPU compute is
called); the host-side Canny entry (the one ABI function that needs no device) matches the numpy oracle exactly;
argument validation of device entries fails loudly without touching a GPU; CLI surfaces match the reference's flags."""
import os
import re

import numpy as np
import pytest

import fie_amd  # noqa: F401
from fie_amd import hip
from oracle import canny as ocanny

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    hip.build()
    return hip.lib()


def test_header_symbols_all_exported(lib):
    text = open(os.path.join(ROOT, "include", "fie.h")).read()
    declared = set(re.findall(r"\b(fie_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fie.h but not exported"
    assert declared - {"fie_last_error"} == set(hip.SIGNATURES), "ctypes signature table out of sync with the header"
    assert lib.fie_version() >= 100


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        hip.Context(0)
    from src.pipeline import FastEditor
    with pytest.raises(RuntimeError):
        FastEditor(model_name="ssd-1b", device="cpu")
    with pytest.raises(ValueError, match="Unknown model"):
        FastEditor(model_name="sd15")


def _scene(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([128 + 100 * np.sin(xx / rng.uniform(5, 30) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(5, 30))
                    for _ in range(3)], 2)
    for _ in range(10):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1, max(2, min(h, w) / 4))
        img[((xx - cx) ** 2 + (yy - cy) ** 2) < r * r] = rng.uniform(0, 255, 3)
    img += rng.normal(0, 6, img.shape)
    return img.clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("seed,h,w,lo,hi", [(0, 64, 64, 100, 200), (1, 97, 131, 100, 200), (2, 256, 256, 50, 150),
                                            (3, 128, 64, 200, 100), (4, 1024, 1024, 100, 200), (5, 8, 8, 10, 20)])
def test_canny_cabi_bit_exact_vs_oracle(lib, seed, h, w, lo, hi):
    img = _scene(seed, h, w)
    got = hip.canny_rgb(img, lo, hi)
    ref = ocanny.canny_rgb(img, lo, hi)
    assert got.dtype == np.uint8 and got.shape == img.shape
    assert np.array_equal(got, ref)
    if h >= 64:
        assert 0 < (got > 0).mean() < 0.5


def test_canny_edge_cases(lib):
    assert hip.canny_rgb(np.zeros((1, 1, 3), np.uint8)).max() == 0
    noise = np.random.default_rng(9).integers(0, 256, (50, 70, 3), dtype=np.uint8)
    assert np.array_equal(hip.canny_rgb(noise), ocanny.canny_rgb(noise))
    idem = hip.canny_rgb(noise)
    assert set(np.unique(idem)) <= {0, 255}
    with pytest.raises(hip.FieError):
        hip._chk(lib.fie_canny_rgb_u8(None, 4, 4, 1, 2, None))
    assert b"bad argument" in lib.fie_last_error()


def _flags(parser):
    return {a.option_strings[0]: (a.default, a.type, tuple(a.choices) if a.choices else None)
            for a in parser._actions if a.option_strings and a.option_strings[0] != "-h"}


def test_cli_flags_match_reference():
    """Flag names/defaults of the reference CLIs (run_batch.py:45-87, run_single_image.py:19-41), plus additive ones."""
    import run_batch
    import run_single_image
    fb = _flags(run_batch.build_parser())
    ref_batch = {"--mapping_file": "data/PIE-Bench_v1/mapping_file.json", "--source_dir": "data/PIE-Bench_v1/annotation_images",
                 "--output_dir": "outputs", "--model": "sdxl", "--num_images": None, "--editing_types": None,
                 "--image_ids": None, "--steps": 4, "--guidance": 1.5, "--control_scale": 0.5, "--canny_low": 100,
                 "--canny_high": 200, "--seed": None, "--negative_prompt": "", "--no_cpu_offload": False,
                 "--quality_mode": False, "--full_precision": False, "--full_controlnet": False, "--skip_existing": False,
                 "--save_comparisons": False}
    for k, d in ref_batch.items():
        assert k in fb and fb[k][0] == d, k
    assert fb["--model"][2] == ("sdxl", "ssd-1b")
    assert set(fb) - set(ref_batch) == {"--strength", "--weights_dir", "--results_json", "--in_flight", "--batch_size"}
    fs = _flags(run_single_image.build_parser())
    for k in ("--image", "--prompt", "--model", "--negative_prompt", "--steps", "--guidance", "--control_scale",
              "--canny_low", "--canny_high", "--seed", "--output_dir", "--no_cpu_offload", "--quality_mode",
              "--full_precision", "--full_controlnet", "--compute_metrics", "--show_plot"):
        assert k in fs, k
    assert fs["--guidance"][0] == 1.5 and fs["--steps"][0] == 4


def test_safe_join_and_selection():
    import argparse
    import run_batch
    assert run_batch.safe_join("/data/src", "0_random/a.jpg") == "/data/src/0_random/a.jpg"
    for bad in ("../x.jpg", "/etc/passwd", "a/../../x"):
        with pytest.raises(ValueError):
            run_batch.safe_join("/data/src", bad)
    mapping = {f"{i:03d}": {"image_path": f"{i % 3}_c/{i}.jpg", "editing_prompt": "p", "editing_type_id": str(i % 3)} for i in range(10)}
    ns = lambda **k: argparse.Namespace(image_ids=None, editing_types=None, num_images=None, **k)
    mute = lambda *a, **k: None
    assert len(run_batch.select_entries(mapping, ns(), mute)) == 10
    a = ns()
    a.editing_types, a.num_images = ["0", "1"], 3
    sel = run_batch.select_entries(mapping, a, mute)
    assert [k for k, _ in sel] == ["000", "001", "003"]
    a = ns()
    a.image_ids = ["007", "nope", "002"]
    assert [k for k, _ in run_batch.select_entries(mapping, a, mute)] == ["007", "002"]


def test_evaluate_schema_matches_reference_results(tmp_path):
    """evaluate.py writes the reference's metrics.csv columns and summary.json layout (evaluate.py:193-271); the
    reference's own results/ files define the schema -- their header / key sets are restated here."""
    import json
    import evaluate
    from PIL import Image
    rng = np.random.default_rng(0)
    mapping = {}
    for i in range(5):
        rel = f"{i % 2}_cat/{i:012d}.jpg"
        for root, jitter in (("src", 0), ("out", 12)):
            p = tmp_path / root / rel
            p.parent.mkdir(parents=True, exist_ok=True)
            a = np.random.default_rng(i).integers(0, 255, (64, 64, 3)).astype(int) + rng.integers(-jitter, jitter + 1, (64, 64, 3))
            Image.fromarray(a.clip(0, 255).astype(np.uint8)).save(p)
        mapping[f"{i:012d}"] = {"image_path": rel, "editing_prompt": f"a [thing] {i}", "editing_type_id": str(i % 2)}
    mapping["missing"] = {"image_path": "9_none/x.jpg", "editing_prompt": "x", "editing_type_id": "9"}
    (tmp_path / "map.json").write_text(json.dumps(mapping))
    evaluate.main(["--mapping_file", str(tmp_path / "map.json"), "--source_dir", str(tmp_path / "src"),
                   "--outputs_dir", str(tmp_path / "out"), "--results_file", str(tmp_path / "r" / "metrics.csv"),
                   "--summary_file", str(tmp_path / "r" / "summary.json"), "--device", "cpu"])
    header = (tmp_path / "r" / "metrics.csv").read_text().splitlines()[0]
    assert header == "image_id,image_path,editing_type_id,editing_prompt,ssim,lpips,clip_score,psnr,mse,dino_distance"
    s = json.loads((tmp_path / "r" / "summary.json").read_text())
    assert s["total_images"] == 5 and set(s) == {"total_images", "overall", "by_category"}
    assert set(s["overall"]) == {"ssim", "lpips", "clip_score", "psnr", "mse", "dino_distance"}
    assert set(s["overall"]["ssim"]) == {"mean", "std", "median"} and 0 < s["overall"]["ssim"]["mean"] < 1
    assert set(s["by_category"]) == {"0", "1"} and s["by_category"]["0"]["count"] == 3
    assert set(s["by_category"]["0"]["psnr"]) == {"mean", "std"} and s["overall"]["lpips"]["mean"] is None


@pytest.mark.parametrize("h,w,oh,ow", [(512, 512, 1024, 1024), (333, 517, 1024, 1024), (1500, 1100, 1024, 1024), (64, 48, 96, 80), (1024, 700, 1024, 1024)])
def test_lanczos_tables_match_pillow(h, w, oh, ow):
    """The restated coefficient / bounds tables of Pillow's 8-bit LANCZOS resample (host side of csrc/resize.hip) reproduce
    `Image.resize(..., Image.LANCZOS)` bit for bit (the two passes emulated in numpy)."""
    from PIL import Image
    from fie_amd import resize
    rng = np.random.default_rng(h * 7 + w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 2] = (np.linspace(0, 255, w)[None, :, None] * np.ones((h // 2, 1, 3))).astype(np.uint8)      # smooth half + noise half
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS))
    assert np.array_equal(resize.resample_numpy(a, oh, ow), ref)


def test_graph_walk_workspace_queries_run_on_the_host(lib):
    """fie_*_workspace_bytes (csrc/graphs.cpp) replay the C++ walks' allocation sequence WITHOUT launching: pure host code, so the CPU suite can
    check the walks' bookkeeping -- sane sizes at the BASELINE configuration, monotone in batch and resolution, -1 for configs the walks refuse."""
    import ctypes
    from fie_amd import cabi, stack
    cfgs = stack.stack_configs("ssd-1b", True)
    uc = cabi.unet_config(cfgs["unet"], 2, 128, 128, 77)
    unet = lib.fie_unet_workspace_bytes(ctypes.byref(uc))
    cc = cabi.unet_config(cfgs["controlnet"], 2, 128, 128, 77)
    cn = lib.fie_controlnet_workspace_bytes(ctypes.byref(cc))
    assert 100e6 < unet < 2e9 and 100e6 < cn < 2e9, (unet, cn)
    assert lib.fie_unet_num_residuals(ctypes.byref(uc)) == len(cabi.skip_shapes(cfgs["unet"], 2, 128, 128)[0]) == 9
    uc1 = cabi.unet_config(cfgs["unet"], 1, 128, 128, 77)
    uc64 = cabi.unet_config(cfgs["unet"], 2, 64, 64, 77)
    assert lib.fie_unet_workspace_bytes(ctypes.byref(uc1)) < unet and lib.fie_unet_workspace_bytes(ctypes.byref(uc64)) < unet
    uc.latent_h = 126                                        # not a multiple of 2^(blocks - 1)
    assert lib.fie_unet_workspace_bytes(ctypes.byref(uc)) == -1
    cc.num_cond_channels = 0
    assert lib.fie_controlnet_workspace_bytes(ctypes.byref(cc)) == -1
    vc = cabi.vae_config(cfgs["vae"], 128, 128)
    dec, enc = lib.fie_vae_decode_workspace_bytes(ctypes.byref(vc), 128, 128), lib.fie_vae_encode_workspace_bytes(ctypes.byref(vc))
    assert 0.5e9 < dec < 4e9 and 0.5e9 < enc < 4e9, (dec, enc)           # three or four live 1024x1024x128 f16 maps (268 MB each) + GroupNorm scratch
    vc.num_blocks = 0
    assert lib.fie_vae_encode_workspace_bytes(ctypes.byref(vc)) == -1
    for key in ("clip_l", "clip_g"):
        c = cfgs[key]
        k = hip.ClipConfig(2, 77, c["hidden"], c["heads"], c["layers"], c["intermediate"], c["projection_dim"] or 0, 1 if c["act"] == "quick_gelu" else 0, c["eps"])
        need = lib.fie_clip_text_workspace_bytes(ctypes.byref(k))
        assert 154 * c["hidden"] * 2 * 3 < need < 154 * c["intermediate"] * 2 * 8, (key, need)
