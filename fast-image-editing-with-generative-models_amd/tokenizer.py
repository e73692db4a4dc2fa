"""Tokenisation for the two CLIP text encoders (upstream CLIPTokenizer; reached from encode_prompt()).

The CLIP BPE vocabulary is not in the reference tree nor in this image, so two tokenizers exist:
  * BpeTokenizer(vocab.json, merges.txt) -- the real byte-level BPE, used when a weights directory supplies the files.
                                             Restates the installed transformers CLIPTokenizer (tokenization_clip.py: NFC,
                                             whitespace collapse, lower-case, the \\p{L} / \\p{N} split pattern, byte-level
                                             alphabet, `</w>` end-of-word merges, unknown pieces -> unk = <|endoftext|>) and is
                                             pinned against it on a synthetic vocabulary: tests/golden/bpe_*.json,
                                             tests/test_host_cpu.py.  (transformers 4.57's slow tokenizer additionally ran
                                             ftfy / BasicTokenizer cleaning; on plain prompts such as PIE-Bench's it agrees.)
  * StandInTokenizer                      -- deterministic stand-in (SURVEY 8d "Tokens"): BOS, one id per whitespace
                                             word = crc32(word) mod 49406, EOS, padded to 77.  "synthetic tokenisation".
Both truncate to 77 and pad with `pad_id` (49407 for encoder 1, 0 for encoder 2)."""
import json
import re
import unicodedata
import zlib

try:                    # \p{L} / \p{N} classes need the third-party `regex` module (installed with transformers)
    import regex as _re_u
except ImportError:     # pragma: no cover
    _re_u = None

import torch

BOS, EOS, MAXLEN = 49406, 49407, 77


class StandInTokenizer:
    synthetic = True

    def __init__(self, pad_id):
        self.pad_id = pad_id

    def __call__(self, texts):
        if isinstance(texts, str):
            texts = [texts]
        out = torch.full((len(texts), MAXLEN), self.pad_id, dtype=torch.int64)
        for i, t in enumerate(texts):
            ids = [BOS] + [zlib.crc32(w.encode("utf-8")) % BOS for w in t.lower().split()][: MAXLEN - 2] + [EOS]
            out[i, : len(ids)] = torch.tensor(ids)
        return out


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, map(chr, cs)))


class BpeTokenizer:
    synthetic = False

    def __init__(self, vocab_path, merges_path, pad_id):
        with open(vocab_path, encoding="utf-8") as f:
            self.vocab = json.load(f)
        with open(merges_path, encoding="utf-8") as f:
            merges = [tuple(l.split()) for l in f.read().split("\n")[1:] if len(l.split()) == 2]
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.b2u = _bytes_to_unicode()
        self.pad_id = pad_id
        self.bos = self.vocab.get("<|startoftext|>", BOS)
        self.eos = self.unk = self.vocab.get("<|endoftext|>", EOS)
        if _re_u is not None:
            self.pat = _re_u.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+")
        else:           # ASCII approximation of the classes (identical on ASCII prompts)
            self.pat = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+|_+")
        self.cache = {}

    def _bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            new, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    new.append(a + b)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
        self.cache[token] = word
        return word

    def __call__(self, texts):
        if isinstance(texts, str):
            texts = [texts]
        out = torch.full((len(texts), MAXLEN), self.pad_id, dtype=torch.int64)
        for i, t in enumerate(texts):
            t = re.sub(r"\s+", " ", unicodedata.normalize("NFC", t)).lower()
            ids = [self.bos]
            for tok in self.pat.findall(t):
                if tok in ("<|startoftext|>", "<|endoftext|>"):
                    ids.append(self.vocab.get(tok, self.unk))
                    continue
                tok = "".join(self.b2u[b] for b in tok.encode("utf-8"))
                ids += [self.vocab.get(p, self.unk) for p in self._bpe(tok)]
            ids = ids[: MAXLEN - 1] + [self.eos]
            out[i, : len(ids)] = torch.tensor(ids)
        return out
