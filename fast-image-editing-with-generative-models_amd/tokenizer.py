"""Tokenisation for the two CLIP text encoders (upstream CLIPTokenizer; reached from encode_prompt()).

The CLIP BPE vocabulary is not in the reference tree nor in this image, so two tokenizers exist:
  * BpeTokenizer(vocab.json, merges.txt) -- the real byte-level BPE, used when a weights directory supplies the files
  * StandInTokenizer                      -- deterministic stand-in (SURVEY 8d "Tokens"): BOS, one id per whitespace
                                             word = crc32(word) mod 49406, EOS, padded to 77.  "synthetic tokenisation".
Both truncate to 77 and pad with `pad_id` (49407 for encoder 1, 0 for encoder 2)."""
import json
import re
import zlib

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
        self.pat = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[a-zA-Z]+|[0-9]|[^\sa-zA-Z0-9]+")
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
            t = re.sub(r"\s+", " ", t.strip()).lower()
            ids = [BOS]
            for tok in self.pat.findall(t):
                tok = "".join(self.b2u[b] for b in tok.encode("utf-8"))
                ids += [self.vocab[p] for p in self._bpe(tok)]
            ids = ids[: MAXLEN - 1] + [EOS]
            out[i, : len(ids)] = torch.tensor(ids)
        return out
