"""`fie_amd.tokenizer.BpeTokenizer` pinned against the upstream implementation that IS importable here: the installed
`transformers.CLIPTokenizer` (what `encode_prompt()` of the diffusers pipeline behind /root/reference/src/pipeline.py:261-272
calls with padding="max_length", max_length=77, truncation=True).  The real 49 408-entry CLIP vocabulary is not available
offline, so both sides use the SYNTHETIC vocabulary of tests/golden/bpe_vocab.json + bpe_merges.txt (learned from the 700
PIE-Bench prompts by tests/golden/make_fixtures.py); the golden ids in bpe_golden.json were produced by transformers 5.15."""
import json
import os

import pytest
import torch

import fie_amd  # noqa: F401
from fie_amd.tokenizer import BpeTokenizer, StandInTokenizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "bpe_golden.json"), encoding="utf-8") as f:
        return json.load(f)


def _tok(pad_id):
    return BpeTokenizer(os.path.join(GOLD, "bpe_vocab.json"), os.path.join(GOLD, "bpe_merges.txt"), pad_id)


def test_bpe_matches_transformers_golden_ids(gold):
    tok = _tok(gold["pad_id"])
    ids = tok(gold["prompts"])
    assert ids.shape == (len(gold["prompts"]), 77) and ids.dtype == torch.int64
    for p, mine, ref in zip(gold["prompts"], ids.tolist(), gold["input_ids"]):
        assert mine == ref, f"prompt {p[:40]!r}"
    # coverage of the cases the golden list was built for
    joined = " ".join(gold["prompts"])
    assert any(ord(ch) > 127 for ch in joined) and "[" in joined and "'" in joined and any(len(p) == 0 for p in gold["prompts"])
    assert any(r[76] == tok.eos and r[75] != gold["pad_id"] for r in gold["input_ids"])        # a truncated (> 77 token) prompt


def test_bpe_matches_transformers_live(gold):
    tr = pytest.importorskip("transformers")
    vocab = json.load(open(os.path.join(GOLD, "bpe_vocab.json"), encoding="utf-8"))
    merges = [tuple(l.split()) for l in open(os.path.join(GOLD, "bpe_merges.txt"), encoding="utf-8").read().split("\n")[1:] if l]
    hf = tr.CLIPTokenizer(vocab=vocab, merges=merges)
    prompts = ["A [wooden] table with 2 cups; it's nice!", "¿qué tal? Ångström 漢字 42", "  ", "don't you'd we're I'm they'll",
               "semi-colon;colon:dash—ellipsis…", "tokens " * 100]
    ref = hf(prompts, padding="max_length", max_length=77, truncation=True)["input_ids"]
    for pad in (hf.pad_token_id, 0):                   # encoder 1 pads with EOS, encoder 2 with id 0
        mine = _tok(pad)(prompts).tolist()
        for m, r in zip(mine, ref):
            n = r.index(hf.eos_token_id) + 1           # through the first EOS the ids agree; behind it only the pad id differs
            assert m[:n] == r[:n] and all(x == pad for x in m[n:])


def test_unknown_piece_maps_to_unk_not_keyerror(gold, tmp_path):
    vocab = json.load(open(os.path.join(GOLD, "bpe_vocab.json"), encoding="utf-8"))
    del vocab["z</w>"]
    p = tmp_path / "vocab.json"
    p.write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
    tok = BpeTokenizer(str(p), os.path.join(GOLD, "bpe_merges.txt"), 0)
    ids = tok(["z"])[0].tolist()
    assert ids[:3] == [tok.bos, tok.unk, tok.eos]


def test_stand_in_tokenizer_contract():
    t = StandInTokenizer(0)
    ids = t(["a [blue] square", ""])
    assert ids.shape == (2, 77) and ids[0, 0] == 49406 and ids[1, 1] == 49407 and ids[1, 2] == 0
    assert (ids[0] == 49407).nonzero()[0].item() == 4
