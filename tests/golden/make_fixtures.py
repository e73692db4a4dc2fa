#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/ (run in the build container, where /root/reference and the local
`transformers` package exist; nothing here runs on the GPU box).

  pie_bench_items.csv     columns 1-4 of the reference's results/ssd-1b_fp16/metrics.csv (data, not code)
  clip_text_golden.npz    tiny seeded CLIP text models evaluated by the installed transformers (the one upstream
                          implementation of the hot path that IS importable here): ids, weights, hidden_states[-2],
                          pooled / projected outputs
  lcm_known_answers.json  SURVEY.md A.5 closed forms + the 8a-RNG fixture
  bpe_vocab.json, bpe_merges.txt, bpe_golden.json
                          a SYNTHETIC CLIP-style byte-level BPE vocabulary (512 byte symbols + merges learned here from the
                          700 PIE-Bench prompts + the two special tokens; the real 49 408-entry vocabulary is not available
                          offline) and the ids the installed transformers.CLIPTokenizer produces with it for a list of
                          prompts (plain, PIE-Bench bracketed, punctuation, digits, contractions, non-ASCII, > 77 tokens)
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


BPE_PROMPTS = [
    "a photo of a cat", "a [rusty] bicycle leaning on the [red] wall", "", "   two   spaces\tand a tab  ",
    "it's a dog's life, isn't it? we'll see -- they've won!", "12 monkeys & 3 apples cost $4.50 (approx.)",
    "caf\u00e9 cr\u00e8me br\u00fbl\u00e9e na\u00efve fa\u00e7ade", "\u00fcber stra\u00dfe \u4e2d\u6587 \u65e5\u672c\u8a9e \u0440\u0443\u0441\u0441\u043a\u0438\u0439",
    "UPPER case And MiXeD", "<|startoftext|> literal specials <|endoftext|> inside", "e\u0301 combining accent vs \u00e9",
    "emoji \U0001F600 and symbols \u00a9\u00ae\u2122 ... !!! ???", "x" * 90, " ".join(["word"] * 120),
    "a_b under_score __ mixed_9 7up", "tabs\nnewlines\r\nand more",
]


def bpe_golden(n_merges=900):
    """Learn a small byte-level BPE on the PIE-Bench prompts (plain greedy pair counting on the CLIP pre-tokenisation), write
    vocab.json / merges.txt in CLIP's file formats, then record what transformers.CLIPTokenizer does with them."""
    import collections
    import regex
    from transformers import CLIPTokenizer
    from fie_amd.tokenizer import _bytes_to_unicode
    b2u = _bytes_to_unicode()
    pat = regex.compile(r"'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+")
    rows = list(csv.DictReader(open(os.path.join(HERE, "pie_bench_items.csv"))))
    words = collections.Counter()
    for r in rows:
        for tok in pat.findall(r["editing_prompt"].lower()):
            sym = [b2u[b] for b in tok.encode("utf-8")]
            sym[-1] += "</w>"
            words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), cnt = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        if cnt < 2:
            break
        merges.append((a, b))
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and w[i] == a and w[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new[tuple(out)] += c
        words = new
    alphabet = [b2u[b] for b in sorted(b2u)]                  # CLIP's vocab order: byte symbols, their </w> forms, merges, specials
    vocab = {s: i for i, s in enumerate(alphabet + [s + "</w>" for s in alphabet] + [a + b for a, b in merges])}
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    with open(os.path.join(HERE, "bpe_vocab.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(HERE, "bpe_merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    tok = CLIPTokenizer(vocab=vocab, merges=list(merges))
    prompts = BPE_PROMPTS + [r["editing_prompt"] for r in rows[::35]]
    ids = tok(prompts, padding="max_length", max_length=77, truncation=True)["input_ids"]
    with open(os.path.join(HERE, "bpe_golden.json"), "w", encoding="utf-8") as f:
        json.dump({"transformers": __import__("transformers").__version__, "pad_id": tok.pad_token_id,
                   "prompts": prompts, "input_ids": ids}, f, ensure_ascii=True)


if __name__ == "__main__":
    pie_items()
    clip_golden()
    lcm_answers()
    bpe_golden()
    print("fixtures written to", HERE)
