"""CLIP's byte-level BPE tokenizer on the host (the `tokenizer` component of the pipeline, pipe:206-243, used at pipe:560-571,
640-646 as `tokenizer(prompt, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids`).

The algorithm is the published one of openai/CLIP `simple_tokenizer.py` (which `transformers.CLIPTokenizer` - the class the
reference loads from `<sd15>/tokenizer` - restates): lower-case, collapse whitespace, split with the CLIP pattern, map bytes to
printable unicode, merge pairs by rank from `merges.txt` with `</w>` closing a word, look the pieces up in `vocab.json`, wrap in
<|startoftext|> ... <|endoftext|>, pad with the pad token.  Files: `vocab.json`, `merges.txt` and (optional) `tokenizer_config.json` /
`special_tokens_map.json` for the pad token (SD-1.5 pads with <|endoftext|>).  tests/test_pipeline_construct_cpu.py checks it
against the installed transformers implementation on a synthetic vocabulary.
"""
import json
import os
from functools import lru_cache

import torch

try:
    import regex as _re
    _PAT = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", _re.IGNORECASE)
except ImportError:                                  # pragma: no cover  (ASCII-only approximation of the unicode classes)
    import re as _re
    _PAT = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+|_+", _re.IGNORECASE)


@lru_cache()
def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class _Encoding(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class CLIPTokenizer:
    model_max_length = 77

    def __init__(self, vocab, merges, pad_token="<|endoftext|>", bos_token="<|startoftext|>", eos_token="<|endoftext|>",
                 model_max_length=77):
        self.encoder = dict(vocab)
        self.bpe_ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.byte_encoder = _bytes_to_unicode()
        self.bos_token_id, self.eos_token_id = self.encoder[bos_token], self.encoder[eos_token]
        self.pad_token_id = self.encoder[pad_token]
        self.unk_token_id = self.eos_token_id
        self.model_max_length = model_max_length
        self._cache = {bos_token: bos_token, eos_token: eos_token}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **_ignored):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "vocab.json"), encoding="utf-8") as f:
            vocab = json.load(f)
        with open(os.path.join(d, "merges.txt"), encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(ln.split()) for ln in lines if ln and not ln.startswith("#version") and len(ln.split()) == 2]
        kw = {}
        for name in ("special_tokens_map.json", "tokenizer_config.json"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                with open(p, encoding="utf-8") as f:
                    cfg = json.load(f)
                for key in ("pad_token", "bos_token", "eos_token"):
                    v = cfg.get(key)
                    if isinstance(v, dict):
                        v = v.get("content")
                    if isinstance(v, str) and key not in kw:
                        kw[key] = v
                if isinstance(cfg.get("model_max_length"), int) and cfg["model_max_length"] < 10 ** 6:
                    kw.setdefault("model_max_length", cfg["model_max_length"])
        return cls(vocab, merges, **kw)

    def _bpe(self, token):
        if token in self._cache:
            return self._cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if best not in self.bpe_ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        res = " ".join(word)
        self._cache[token] = res
        return res

    def encode(self, text):
        """Token ids WITHOUT the start / end tokens."""
        text = " ".join(text.split()).strip().lower()
        ids = []
        for tok in _PAT.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder.get(piece, self.unk_token_id) for piece in self._bpe(tok).split(" "))
        return ids

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors=None, **_ignored):
        texts = [text] if isinstance(text, str) else list(text)
        max_length = max_length or self.model_max_length
        rows, masks = [], []
        for t in texts:
            ids = [self.bos_token_id] + self.encode(t) + [self.eos_token_id]
            if truncation and len(ids) > max_length:
                ids = ids[: max_length - 1] + [self.eos_token_id]
            mask = [1] * len(ids)
            if padding == "max_length":
                mask += [0] * (max_length - len(ids))
                ids += [self.pad_token_id] * (max_length - len(ids))
            rows.append(ids)
            masks.append(mask)
        if padding in (True, "longest"):
            n = max(len(r) for r in rows)
            masks = [m + [0] * (n - len(m)) for m in masks]
            rows = [r + [self.pad_token_id] * (n - len(r)) for r in rows]
        if return_tensors == "pt":
            return _Encoding(input_ids=torch.tensor(rows, dtype=torch.int64), attention_mask=torch.tensor(masks, dtype=torch.int64))
        return _Encoding(input_ids=rows if not isinstance(text, str) else rows[0],
                         attention_mask=masks if not isinstance(text, str) else masks[0])
