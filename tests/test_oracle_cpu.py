"""CPU suite (-m "not gpu"): pins the oracle against the golden vectors / known answers this path has, and checks the
host logic of the product that needs no GPU (schedule tables, tokenizer, weight tables, FLOP model, LoRA fold)."""
import json
import os

import numpy as np
import pytest
import torch

import fie_amd  # noqa: F401
from fie_amd import flops, lcm, presets as P, tokenizer, weights
from oracle import canny as ocanny
from oracle import lcm as olcm
from oracle import metrics as ometrics
from oracle import nets

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLD, "lcm_known_answers.json")))


# ------------------------------------------------------------------ LCM schedule (SURVEY A.5 known answers)
def test_lcm_timesteps_and_tables_oracle_and_product():
    o = olcm.LCMOracle()
    assert o.set_timesteps(4) == KAT["timesteps_4"]
    p = lcm.LCMSchedule(**P.LCM_SCHED, timestep_spacing="trailing")      # spacing stored, ignored
    assert p.timesteps(4) == KAT["timesteps_4"]
    for t, a in KAT["alpha_bar"].items():
        assert abs(float(o.alphas_cumprod[int(t)]) - a) < 2e-8
        assert abs(float(p.alphas_cumprod[int(t)]) - a) < 2e-8
    for t in KAT["c_skip"]:
        cs, co = o.boundary(int(t))
        assert abs(cs - KAT["c_skip"][t]) / KAT["c_skip"][t] < 1e-3
        assert abs(co - KAT["c_out"][t]) < 1e-9


@pytest.mark.parametrize("strength,evals", [(1.0, 4), (0.8, 3), (0.5, 2), (0.3, 1), (0.2, 0)])
def test_strength_to_eval_count(strength, evals):
    o = olcm.LCMOracle()
    o.set_timesteps(4)
    ts, _ = o.get_timesteps(4, strength)
    plan = lcm.LCMSchedule(**P.LCM_SCHED).plan(4, strength)
    assert len(ts) == len(plan) == evals == KAT["evals_by_strength"][str(strength)]
    assert [s["t"] for s in plan] == ts
    if plan:
        assert [s["last"] for s in plan] == [False] * (evals - 1) + [True]


def test_product_schedule_scalars_match_oracle_step():
    o = olcm.LCMOracle()
    o.set_timesteps(4)
    ts, _ = o.get_timesteps(4, 0.8)
    plan = lcm.LCMSchedule(**P.LCM_SCHED).plan(4, 0.8)
    g = torch.Generator().manual_seed(0)
    x, eps, z = (torch.randn(1, 4, 8, 8, generator=g) for _ in range(3))
    for st, t in zip(plan, ts):
        ref, _ = o.step(eps, t, x, None if st["last"] else z)
        x0 = (x - st["sqrt_1mab"] * eps) / st["sqrt_ab"]
        den = st["c_out"] * x0 + st["c_skip"] * x
        mine = den if st["last"] else st["sqrt_ab_prev"] * den + st["sqrt_1mab_prev"] * z
        assert torch.allclose(mine, ref, atol=1e-5)


def test_rng_fixture():
    g = torch.Generator(device="cpu").manual_seed(42)
    d = torch.randn((1, 4, 128, 128), generator=g, dtype=torch.float32).flatten()[:4]
    assert torch.allclose(d, torch.tensor(KAT["rng_seed42_fp32_first4"]), atol=0, rtol=0)
    g16 = torch.Generator(device="cpu").manual_seed(42)
    h = torch.randn((1, 4, 128, 128), generator=g16, dtype=torch.float16)
    assert (h.float().flatten()[:4] - d).abs().max() > 1e-2      # fp16 draws are a different stream


def test_timestep_embedding_known_answer():
    e = nets.timestep_embedding(torch.tensor([499.0]), 320)[0]
    k = KAT["timestep_embedding_t499_dim320"]
    assert torch.allclose(e[0:3], torch.tensor(k["0:3"]), atol=1e-5)
    assert torch.allclose(e[160:163], torch.tensor(k["160:163"]), atol=1e-5)


# ------------------------------------------------------------------ CLIP text: pinned by transformers golden vectors
@pytest.mark.parametrize("tag", ["l", "g"])
def test_clip_oracle_matches_transformers_golden(tag):
    z = np.load(os.path.join(GOLD, "clip_text_golden.npz"))
    sd = {k.split("::", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_w::")}
    hidden = sd["text_model.embeddings.position_embedding.weight"].shape[1]
    cfg = dict(hidden=hidden, layers=max(int(k.split(".")[3]) for k in sd if ".layers." in k) + 1,
               heads=2 if tag == "l" else 1, act="quick_gelu" if tag == "l" else "gelu",
               projection_dim=64 if tag == "g" else 0, eps=1e-5, eos_token_id=299)
    ids = torch.from_numpy(z[f"{tag}_ids"])
    hs, pooled = nets.clip_text_forward(sd, cfg, ids)
    assert torch.allclose(hs[-2], torch.from_numpy(z[f"{tag}_penultimate"]), atol=2e-5)
    assert torch.allclose(pooled, torch.from_numpy(z[f"{tag}_pooled"]), atol=2e-5)


def test_clip_oracle_matches_transformers_live():
    tr = pytest.importorskip("transformers")
    cfg = tr.CLIPTextConfig(vocab_size=500, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                            num_attention_heads=2, max_position_embeddings=77, hidden_act="gelu", projection_dim=32,
                            eos_token_id=499, pad_token_id=0, bos_token_id=498)
    torch.manual_seed(3)
    m = tr.CLIPTextModelWithProjection(cfg).eval()
    ids = torch.zeros(1, 77, dtype=torch.long)
    ids[0, :5] = torch.tensor([498, 11, 12, 13, 499])
    with torch.no_grad():
        r = m(ids, output_hidden_states=True)
    mine = dict(hidden=128, layers=3, heads=2, act="gelu", projection_dim=32, eps=1e-5, eos_token_id=499)
    sd = {(k if k.startswith(("text_model.", "text_projection.")) else "text_model." + k): v for k, v in m.state_dict().items()}
    hs, pooled = nets.clip_text_forward(sd, mine, ids)
    assert len(r.hidden_states) == 4 and torch.allclose(hs[-2], r.hidden_states[-2], atol=2e-5)
    assert torch.allclose(pooled, r.text_embeds, atol=2e-5)


# ------------------------------------------------------------------ architecture tables
@pytest.mark.parametrize("cfg,millions", [(P.UNET_SDXL, 2567.46), (P.CONTROLNET_FULL, 1251.01), (P.VAE_SDXL, 83.65),
                                          (P.UNET_SSD1B_A1, 1331.33), (P.CLIP_L, 123.06), (P.CLIP_BIGG, 694.66)])
def test_published_parameter_counts(cfg, millions):
    assert abs(weights.param_count(cfg) / 1e6 - millions) < 0.01


def test_flop_model_matches_survey():
    assert abs(flops.unet_like_flops(P.UNET_SDXL)["total"] / 1e12 - 6.761) < 2e-3
    assert abs(flops.unet_like_flops(P.UNET_SSD1B_A)["total"] / 1e12 - 4.268) < 2e-3
    assert abs(flops.unet_like_flops(P.CONTROLNET_FULL)["total"] / 1e12 - 3.020) < 2e-3
    v = flops.vae_flops(P.VAE_SDXL)
    assert abs(v["encode"] / 1e12 - 4.879) < 2e-3 and abs(v["decode"] / 1e12 - 10.470) < 2e-3


def test_oracle_graph_consumes_every_parameter():
    """Every tensor of the tables is read by the oracle graphs (NaN-poison each weight -> output must change)."""
    cfgs = {k: P.STACKS["tiny"][k2] for k, k2 in (("unet", "unet"), ("controlnet", "controlnet_full"))}
    sd = weights.synth_state_dict(cfgs["unet"], seed=1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 8, 8, generator=g)
    text, pooled = torch.randn(1, 77, 192, generator=g), torch.randn(1, 64, generator=g)
    tid = torch.tensor([[64., 64., 0, 0, 64., 64.]])
    used = set()

    class Spy(dict):
        def __getitem__(self, k):
            used.add(k)
            return dict.__getitem__(self, k)

        def get(self, k, d=None):
            used.add(k)
            return dict.get(self, k, d)

    nets.unet_forward(Spy(sd), cfgs["unet"], x, 499, text, pooled, tid)
    assert used >= set(sd), sorted(set(sd) - used)[:5]


def test_fold_lora_equals_runtime_adapter():
    cfg = P.TINY_UNET
    sd = weights.synth_state_dict(cfg, seed=2)
    lora = weights.synth_lora(cfg, seed=3, rank=4)
    name = "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q"
    w0 = sd[name + ".weight"].clone()
    n = weights.fold_lora(sd, lora)
    assert n > 0
    x = torch.randn(5, w0.shape[1])
    runtime = x @ w0.T + (x @ lora[name + ".lora_A.weight"].T) @ lora[name + ".lora_B.weight"].T
    assert torch.allclose(x @ sd[name + ".weight"].T, runtime, atol=1e-5)


# ------------------------------------------------------------------ tokenizer, canny, metrics
def test_standin_tokenizer_shape_and_padding():
    t1, t2 = tokenizer.StandInTokenizer(49407), tokenizer.StandInTokenizer(0)
    a, b = t1(["a [rusty] bike", ""]), t2(["a [rusty] bike", ""])
    assert a.shape == b.shape == (2, 77) and a[0, 0] == 49406 and a[0, 4] == 49407
    assert a[1, 1] == 49407 and (a[1, 2:] == 49407).all() and (b[1, 2:] == 0).all()
    assert torch.equal(a[0, :5], b[0, :5])
    long = t1(" ".join(["w"] * 200))
    assert long.shape == (1, 77) and long[0, -1] == 49407


def test_bpe_tokenizer_with_synthetic_vocab(tmp_path):
    b2u = tokenizer._bytes_to_unicode()
    vocab = {c + "</w>": i for i, c in enumerate("abc")}
    vocab.update({c: 10 + i for i, c in enumerate("abc")})
    vocab.update({"ab": 20, "ab</w>": 21, "abc</w>": 22})
    (tmp_path / "vocab.json").write_text(json.dumps(vocab))
    (tmp_path / "merges.txt").write_text("#version\na b\nab c</w>\na b</w>\n")
    tk = tokenizer.BpeTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"), pad_id=0)
    ids = tk(["abc ab"])[0]
    assert ids[:4].tolist() == [49406, 22, 21, 49407] and (ids[4:] == 0).all()
    assert len(b2u) == 256


def test_canny_oracle_properties():
    assert ocanny.rgb_to_gray(np.array([[[255, 255, 255]]], np.uint8))[0, 0] == 255
    assert ocanny.rgb_to_gray(np.array([[[255, 0, 0]]], np.uint8))[0, 0] == 76
    flat = np.full((32, 32, 3), 128, np.uint8)
    assert ocanny.canny_rgb(flat).max() == 0
    step = flat.copy()
    step[:, 16:] = 250
    e = ocanny.canny_rgb(step)
    assert set(np.unique(e)) == {0, 255} and e[:, 14:18].max() == 255 and e[:, :12].max() == 0 and e[:, 20:].max() == 0
    assert (e[..., 0] == e[..., 1]).all() and (e[..., 1] == e[..., 2]).all()
    weak = flat.copy()
    weak[:, 16:] = 160              # L1 gradient 4*32 = 128: above low (100), below high (200) -> no strong seed
    assert ocanny.canny_rgb(weak).max() == 0
    assert np.array_equal(ocanny.canny_rgb(step, 200, 100), e)   # thresholds are swapped when reversed


def test_ssim_against_an_independent_float64_restatement():
    """SSIM (reference src/metrics.py:227-239 -> torchmetrics StructuralSimilarityIndexMeasure(data_range=1.0): 11x11 Gaussian, sigma 1.5, k1 0.01,
    k2 0.03, reflect pad then crop, mean over the map -- torchmetrics is not installable here, so this is NOT a pin against it) checked against a second,
    independent code path: separable float64 correlation with scipy.ndimage on the un-padded image, evaluated only where the window never touches the
    border (the region torchmetrics keeps after its crop), formula straight from Wang et al. 2004 eq. 13.  Known answers: SSIM(a, a) = 1; a constant
    image against another constant has SSIM = luminance term (2 mu_x mu_y + c1) / (mu_x^2 + mu_y^2 + c1)."""
    from PIL import Image
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(3)
    base = rng.integers(0, 255, (96, 80, 3), dtype=np.uint8)
    other = (base.astype(int) + rng.integers(-25, 25, base.shape)).clip(0, 255).astype(np.uint8)

    def ssim64(u8a, u8b):
        x, y = u8a.astype(np.float64) / 255.0, u8b.astype(np.float64) / 255.0
        d = np.arange(-5, 6, dtype=np.float64)
        g = np.exp(-(d / 1.5) ** 2 / 2)
        g /= g.sum()
        blur = lambda t: correlate1d(correlate1d(t, g, axis=0, mode="constant"), g, axis=1, mode="constant")[5:-5, 5:-5]
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        vals = []
        for ch in range(3):
            mx, my = blur(x[..., ch]), blur(y[..., ch])
            sxx, syy, sxy = blur(x[..., ch] ** 2) - mx * mx, blur(y[..., ch] ** 2) - my * my, blur(x[..., ch] * y[..., ch]) - mx * my
            vals.append(((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2)))
        return float(np.mean(vals))

    got = ometrics.ssim(Image.fromarray(base), Image.fromarray(other), size=None)
    assert abs(got - ssim64(base, other)) < 2e-5, (got, ssim64(base, other))
    flat_a, flat_b = np.full((64, 64, 3), 100, np.uint8), np.full((64, 64, 3), 140, np.uint8)
    mx, my, c1 = 100 / 255.0, 140 / 255.0, 1e-4
    # both variances are zero: the structure term is c2 / c2 = 1 and only the luminance term remains (to fp32 cancellation in E[x^2] - mu^2 against c2 = 9e-4)
    assert abs(ometrics.ssim(Image.fromarray(flat_a), Image.fromarray(flat_b), size=None) - (2 * mx * my + c1) / (mx * mx + my * my + c1)) < 5e-4


def test_ssim_oracle_and_product_metrics_agree():
    from PIL import Image
    from src.metrics import MetricsCalculator
    rng = np.random.default_rng(0)
    a = Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8))
    b = Image.fromarray((np.asarray(a).astype(int) + rng.integers(-20, 20, (64, 64, 3))).clip(0, 255).astype(np.uint8))
    assert abs(ometrics.ssim(a, a) - 1.0) < 1e-6
    mc = MetricsCalculator(device="cpu")
    assert abs(mc.calculate_ssim(a, b) - ometrics.ssim(a, b)) < 1e-6
    assert abs(mc.calculate_psnr(a, b) - ometrics.psnr(a, b)) < 1e-4
    m = mc.calculate_all_metrics(a, b, "x")
    assert set(m) == {"ssim", "lpips", "clip_score", "psnr", "mse", "dino_distance"} and m["lpips"] is None


def test_oracle_pipeline_determinism_and_step_count():
    from PIL import Image
    from oracle import pipeline as opipe
    st = P.STACKS["tiny"]
    cfgs = dict(unet=st["unet"], controlnet=st["controlnet_full"], vae=st["vae"], clip_l=st["clip_l"], clip_g=st["clip_g"])
    sds = {k: weights.synth_state_dict(cfgs[k], seed=10 + i) for i, k in enumerate(cfgs)}
    img = Image.fromarray(np.random.default_rng(1).integers(0, 255, (64, 64, 3), dtype=np.uint8))
    ids = (tokenizer.StandInTokenizer(49407)(["a cat"]), tokenizer.StandInTokenizer(0)(["a cat"]))
    tr = {}
    a = opipe.edit(sds, cfgs, img, ids, ids, size=64, seed=7, strength=0.5, guidance_scale=1.0, trace=tr)
    b = opipe.edit(sds, cfgs, img, ids, ids, size=64, seed=7, strength=0.5, guidance_scale=1.0)
    assert len(tr["eps"]) == 2 and a.shape == (64, 64, 3) and np.array_equal(a, b)


def _write_stack(root, cfgs, sds, variant="fp16"):
    """A diffusers-layout directory for a stack: <sub>/config.json (fie_amd.config.to_diffusers) + safetensors."""
    from safetensors.torch import save_file
    from fie_amd import config, stack
    for k, d in stack.SUBDIRS.items():
        os.makedirs(os.path.join(root, d), exist_ok=True)
        stem = "diffusion_pytorch_model" if k in ("unet", "controlnet", "vae") else "model"
        save_file(sds[k], os.path.join(root, d, f"{stem}.{variant}.safetensors"))
        with open(os.path.join(root, d, "config.json"), "w") as f:
            json.dump(config.to_diffusers(cfgs[k]), f)


def _same_graph(a, b):
    drop = ("name", "force_upcast")
    return {k: v for k, v in a.items() if k not in drop} == {k: v for k, v in b.items() if k not in drop}


def test_directory_stack_roundtrip(tmp_path):
    """Disk format on the weights side of the path: a diffusers-layout directory written with
    safetensors is read back exactly, shape mismatches fail loudly, LoRA is folded, BPE files switch the tokenizer."""
    from safetensors.torch import save_file
    from fie_amd import stack
    cfgs = stack.stack_configs("tiny-nomid", True)
    sds = {k: weights.synth_state_dict(cfgs[k], seed=50 + i, dtype=torch.float16) for i, k in enumerate(stack.KEYS)}
    _write_stack(str(tmp_path), cfgs, sds)
    with pytest.raises(FileNotFoundError, match="needs the LCM-LoRA"):        # tiny-nomid is an lcm_lora stack: mandatory, as :154
        stack.directory_stack(str(tmp_path), "tiny-nomid", True)
    lora = weights.synth_lora(cfgs["unet"], seed=9, rank=4)
    save_file(lora, str(tmp_path / "pytorch_lora_weights.safetensors"))
    os.makedirs(tmp_path / "tokenizer")
    (tmp_path / "tokenizer" / "vocab.json").write_text(json.dumps({"a</w>": 5, "!": 0, "<|startoftext|>": 49406, "<|endoftext|>": 49407}))
    (tmp_path / "tokenizer" / "merges.txt").write_text("#version\n")
    (tmp_path / "tokenizer" / "special_tokens_map.json").write_text(json.dumps({"pad_token": {"content": "!"}}))
    c2, loaded, toks = stack.directory_stack(str(tmp_path), "tiny-nomid", True)
    assert all(_same_graph(c2[k], cfgs[k]) for k in cfgs) and c2["unet"]["name"] == "unet/config.json"
    assert torch.equal(loaded["vae"]["decoder.conv_out.weight"], sds["vae"]["decoder.conv_out.weight"])
    q = "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight"
    assert not torch.equal(loaded["unet"][q], sds["unet"][q])          # LoRA folded (tiny-nomid is an lcm_lora stack)
    ref = dict(sds["unet"])
    weights.fold_lora(ref, lora)
    assert torch.equal(loaded["unet"][q], ref[q])
    assert isinstance(toks[0], tokenizer.BpeTokenizer) and isinstance(toks[1], tokenizer.StandInTokenizer)
    assert toks[0](["a"])[0, :4].tolist() == [49406, 5, 49407, 0]      # pad id from the tokenizer's own special_tokens_map
    bad = dict(sds["vae"])
    bad["decoder.conv_out.weight"] = torch.zeros(3, 7, 3, 3, dtype=torch.float16)
    save_file(bad, str(tmp_path / "vae" / "diffusion_pytorch_model.fp16.safetensors"))
    with pytest.raises(ValueError, match="does not match its graph"):
        stack.directory_stack(str(tmp_path), "tiny-nomid", True)


def test_directory_stack_builds_the_graph_from_config_json(tmp_path):
    """A checkpoint whose topology differs from EVERY preset runs from its own config.json (VERDICT r1 missing #2): asymmetric
    nested depths + reverse depths, a plain mid block (UNetMidBlock2D), a ControlNet with other depths and conditioning
    channels, a CLIP config with the legacy eos_token_id = 2.  The oracle evaluates the loaded stack on the loaded config."""
    from fie_amd import config, stack
    base = stack.stack_configs("tiny", True)
    unet = dict(base["unet"], name="odd-unet", down_attn=((0, 0), (1, 2), (2, 1)), mid_attn=0, mid_resnets=1,
                up_attn=((1, 2, 1), (2, 1, 1), (0, 0, 0)))
    cn = dict(base["controlnet"], name="odd-cn", down_attn=((0, 0), (0, 0), (1, 1)), mid_attn=1,
              conditioning_embedding_out_channels=(8, 16, 24, 40))
    clip_g = dict(base["clip_g"], eos_token_id=2)
    cfgs = dict(base, unet=unet, controlnet=cn, clip_g=clip_g)
    for c in (unet, cn):
        assert not any(_same_graph(c, q) for q in vars(P).values() if isinstance(q, dict) and q.get("kind") == c["kind"])
    sds = {k: weights.synth_state_dict(cfgs[k], seed=70 + i, dtype=torch.float16) for i, k in enumerate(stack.KEYS)}
    _write_stack(str(tmp_path), cfgs, sds)
    c2, loaded, _ = stack.directory_stack(str(tmp_path), "ssd-1b", use_full_controlnet=False)      # the reference's default ctor flags
    assert all(_same_graph(c2[k], cfgs[k]) for k in cfgs)
    assert c2["unet"]["up_attn"] == unet["up_attn"] and c2["unet"]["mid_resnets"] == 1 and c2["clip_g"]["eos_token_id"] == 2
    assert set(loaded["unet"]) == set(sds["unet"]) and set(loaded["controlnet"]) == set(sds["controlnet"])
    # the loaded (config, weights) pair is a runnable graph: one oracle ControlNet + UNet evaluation
    f32 = {k: {n: v.float() for n, v in sd.items()} for k, sd in loaded.items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 16, 16, generator=g)
    text, pooled = torch.randn(1, 77, unet["cross_attention_dim"], generator=g), torch.randn(1, 64, generator=g)
    tid = torch.tensor([[128., 128., 0, 0, 128., 128.]])
    down, mid = nets.controlnet_forward(f32["controlnet"], c2["controlnet"], x, 499, text, torch.rand(1, 3, 128, 128, generator=g), 0.5, pooled, tid)
    eps = nets.unet_forward(f32["unet"], c2["unet"], x, 499, text, pooled, tid, down, mid)
    assert eps.shape == (1, 4, 16, 16) and torch.isfinite(eps).all()
    # unsupported topologies are refused, not mis-run
    j = config.to_diffusers(unet)
    with pytest.raises(ValueError, match="head dims"):
        config.unet_cfg(dict(j, attention_head_dim=[1, 4, 8]))
    with pytest.raises(ValueError, match="reverse_transformer_layers_per_block"):
        config.unet_cfg({k: v for k, v in j.items() if k != "reverse_transformer_layers_per_block"})
    with pytest.raises(ValueError, match="mid_block_type"):
        config.unet_cfg(dict(j, mid_block_type=None))


def test_config_json_of_the_published_sdxl_checkpoints_maps_to_the_presets():
    """The config.json fields of stabilityai/stable-diffusion-xl-base-1.0 (unet), diffusers/controlnet-canny-sdxl-1.0 and
    madebyollin/sdxl-vae-fp16-fix, as published, give exactly the [H] presets."""
    from fie_amd import config
    unet = dict(block_out_channels=[320, 640, 1280], layers_per_block=2, transformer_layers_per_block=[1, 2, 10],
                down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], mid_block_type="UNetMidBlock2DCrossAttn",
                attention_head_dim=[5, 10, 20], cross_attention_dim=2048, use_linear_projection=True, norm_num_groups=32,
                norm_eps=1e-5, addition_embed_type="text_time", addition_time_embed_dim=256,
                projection_class_embeddings_input_dim=2816, in_channels=4, out_channels=4, time_cond_proj_dim=None)
    assert _same_graph(config.unet_cfg(unet), P.UNET_SDXL)
    cn = {k: v for k, v in unet.items() if k not in ("up_block_types", "out_channels", "time_cond_proj_dim")}
    cn.update(conditioning_channels=3, conditioning_embedding_out_channels=[16, 32, 96, 256])
    assert _same_graph(config.controlnet_cfg(cn), P.CONTROLNET_FULL)
    ssd = dict(unet, transformer_layers_per_block=[1, [2, 2], [4, 4]], reverse_transformer_layers_per_block=[[4, 4, 10], [2, 1, 1], 1],
               mid_block_type="UNetMidBlock2D", time_cond_proj_dim=256)
    assert _same_graph(config.unet_cfg(ssd), P.UNET_SSD1B_A1)       # the A' preset = get_mid_block("UNetMidBlock2D"): one resnet
    vae = dict(block_out_channels=[128, 256, 512, 512], layers_per_block=2, latent_channels=4, norm_num_groups=32,
               scaling_factor=0.13025, in_channels=3, out_channels=3, force_upcast=False)
    assert _same_graph(config.vae_cfg(vae), P.VAE_SDXL)


@pytest.mark.parametrize("scheme", ["peft", "peft_unet_prefix", "diffusers_old", "processor", "kohya"])
def test_fold_lora_key_schemes(scheme):
    """pytorch_lora_weights.safetensors of latent-consistency/lcm-lora-sdxl in PEFT and kohya key schemes (+ the two older
    diffusers spellings): all fold to the same weights as the plain lora_A / lora_B form, alpha / rank applied."""
    cfg = P.TINY_UNET
    sd0 = weights.synth_state_dict(cfg, seed=2)
    lora = weights.synth_lora(cfg, seed=3, rank=4)
    want = dict(sd0)
    assert weights.fold_lora(want, lora, scale=0.5) > 0             # kohya files carry alpha: emulate alpha = r/2 with scale
    conv = "down_blocks.0.resnets.0.conv1"
    g = torch.Generator().manual_seed(5)
    ca, cb = torch.randn(4, 64, 3, 3, generator=g) * 0.05, torch.randn(64, 4, 1, 1, generator=g) * 0.05
    want[conv + ".weight"] = want[conv + ".weight"] + 0.5 * torch.einsum("or,rikl->oikl", cb.flatten(1), ca)
    ren = {}
    mods = sorted({k[: -len(".lora_A.weight")] for k in lora if k.endswith(".lora_A.weight")}) + [conv]
    for m in mods:
        a, b = (lora[m + ".lora_A.weight"], lora[m + ".lora_B.weight"]) if m != conv else (ca, cb)
        if scheme == "peft":
            ren[m + ".lora_A.weight"], ren[m + ".lora_B.weight"], ren[m + ".alpha"] = a, b, torch.tensor(2.0)
        elif scheme == "peft_unet_prefix":
            ren["unet." + m + ".lora_A.weight"], ren["unet." + m + ".lora_B.weight"], ren["unet." + m + ".alpha"] = a, b, torch.tensor(2.0)
        elif scheme == "diffusers_old":
            ren["unet." + m + ".lora.down.weight"], ren["unet." + m + ".lora.up.weight"], ren["unet." + m + ".alpha"] = a, b, torch.tensor(2.0)
        elif scheme == "processor" and ".attn" in m:
            blk, proj = m.rsplit(".to_", 1)
            nm = f"unet.{blk}.processor.to_{proj.replace('.0', '')}_lora"
            ren[nm + ".down.weight".replace(".down", ".down")] = a
            ren[nm.replace("_lora", "_lora") + ".up.weight"] = b
            ren[nm + ".alpha"] = torch.tensor(2.0)
        elif scheme == "processor":
            ren["unet." + m + ".lora_A.weight"], ren["unet." + m + ".lora_B.weight"], ren["unet." + m + ".alpha"] = a, b, torch.tensor(2.0)
        else:
            k = "lora_unet_" + m.replace(".", "_")
            ren[k + ".lora_down.weight"], ren[k + ".lora_up.weight"], ren[k + ".alpha"] = a, b, torch.tensor(2.0)
    ren["lora_te1_text_model_encoder_layers_0_mlp_fc1.lora_down.weight"] = torch.zeros(4, 8)     # text-encoder adapters are skipped
    got = dict(sd0)
    assert weights.fold_lora(got, ren) == len(mods)
    for m in mods:
        assert torch.allclose(got[m + ".weight"], want[m + ".weight"], atol=1e-6), m
    with pytest.raises(ValueError, match="match no UNet module"):
        weights.fold_lora(dict(sd0), {"lora_unet_input_blocks_4_1_proj_in.lora_down.weight": torch.zeros(4, 8),
                                      "lora_unet_input_blocks_4_1_proj_in.lora_up.weight": torch.zeros(8, 4)})
