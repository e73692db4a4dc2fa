#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/ (run in the build container, where /root/reference and the local
`transformers` package exist; nothing here runs on the GPU box).

  pie_bench_items.csv     columns 1-4 of the reference's results/ssd-1b_fp16/metrics.csv (data, not code)
  clip_text_golden.npz    tiny seeded CLIP text models evaluated by the installed transformers (the one upstream
                          implementation of the hot path that IS importable here): ids, weights, hidden_states[-2],
                          pooled / projected outputs
  lcm_known_answers.json  SURVEY.md A.5 closed forms + the 8a-RNG fixture
"""
import csv
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def pie_items():
    src = "/root/reference/results/ssd-1b_fp16/metrics.csv"
    rows = list(csv.reader(open(src)))
    with open(os.path.join(HERE, "pie_bench_items.csv"), "w", newline="") as f:
        csv.writer(f).writerows(r[:4] for r in rows)


def clip_golden():
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    out = {}
    for tag, act, proj, layers, heads, hidden in (("l", "quick_gelu", 0, 2, 2, 128), ("g", "gelu", 64, 3, 1, 64)):
        cfg = CLIPTextConfig(vocab_size=300, hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=layers,
                             num_attention_heads=heads, max_position_embeddings=77, hidden_act=act,
                             projection_dim=proj or 64, eos_token_id=299, pad_token_id=0, bos_token_id=298)
        torch.manual_seed(7)
        model = (CLIPTextModelWithProjection if proj else CLIPTextModel)(cfg).eval()
        ids = torch.zeros(2, 77, dtype=torch.long)
        ids[0, :6] = torch.tensor([298, 5, 17, 200, 3, 299])
        ids[1, :3] = torch.tensor([298, 42, 299])
        if not proj:
            ids[ids == 0] = 299        # encoder 1 pads with EOS
            ids[0, 0] = ids[1, 0] = 298
        with torch.no_grad():
            r = model(ids, output_hidden_states=True)
        out[f"{tag}_ids"] = ids.numpy()
        out[f"{tag}_penultimate"] = r.hidden_states[-2].numpy()
        out[f"{tag}_pooled"] = (r.text_embeds if proj else r.pooler_output).numpy()
        for k, v in model.state_dict().items():
            # transformers 5.x drops the `text_model.` prefix on CLIPTextModel; store the 4.57 names the reference used
            if not k.startswith(("text_model.", "text_projection.")):
                k = "text_model." + k
            out[f"{tag}_w::{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "clip_text_golden.npz"), **out)


def lcm_answers():
    g = torch.Generator(device="cpu").manual_seed(42)
    draw = torch.randn((1, 4, 128, 128), generator=g, dtype=torch.float32).flatten()[:4].tolist()
    kat = {
        "timesteps_4": [999, 759, 499, 259],
        "alpha_bar": {"999": 0.00466010, "759": 0.05221289, "499": 0.27766943, "259": 0.65897524},
        "c_skip": {"999": 2.5050e-09, "759": 4.3397e-09, "499": 1.0040e-08, "259": 3.7268e-08},
        "c_out": {"999": 0.9999999987, "759": 0.9999999978, "499": 0.9999999950, "259": 0.9999999814},
        "evals_by_strength": {"1.0": 4, "0.8": 3, "0.5": 2, "0.3": 1, "0.2": 0},
        "rng_seed42_fp32_first4": [1.9269152879714966, 1.4872840642929077, 0.9007171988487244, -2.1055209636688232],
        "rng_seed42_fp32_first4_measured_here": draw,
        "timestep_embedding_t499_dim320": {"0:3": [-0.87116218, 0.98838931, 0.19755381],
                                           "160:163": [0.49099535, -0.15194249, -0.98029202]},
    }
    with open(os.path.join(HERE, "lcm_known_answers.json"), "w") as f:
        json.dump(kat, f, indent=1)


if __name__ == "__main__":
    pie_items()
    clip_golden()
    lcm_answers()
    print("fixtures written to", HERE)
