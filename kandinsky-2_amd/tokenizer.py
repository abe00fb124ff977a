"""CLIP byte-pair tokenizer for the 2.1 prior / CLIP text tower (host side, once per prompt).

The reference builds `kandinsky2.model.prior.CustomizedTokenizer()` (prior.py:387-416): OpenAI `clip`'s `SimpleTokenizer` on the
merges file that ships inside that package (`bpe_simple_vocab_16e6.txt.gz`) plus `padded_tokens_and_mask`.  `clip` is an un-vendored,
un-pinned dependency that is not installed here, so the published algorithm of its `simple_tokenizer.py` is restated:

  text -> html-unescape twice, strip, collapse whitespace, lower-case            (the reference also runs ftfy.fix_text first: a no-op
                                                                                  on well-formed text; ftfy is not a dependency here)
       -> regex split: the two specials | 's 't 're 've 'm 'll 'd | letters+ | ONE digit | other non-space runs
       -> UTF-8 bytes of each piece mapped to 256 printable unicode characters (GPT-2's byte table)
       -> byte-pair merges, lowest rank first, the last symbol of a piece carrying "</w>"
  vocabulary = 256 byte symbols, the same 256 with "</w>", the first 48 894 merges of the file, <|startoftext|>, <|endoftext|> (49 408).

`ClipBPETokenizer(path)` reads the merges from `bpe_simple_vocab_16e6.txt.gz` (or the same list as a plain-text `merges.txt`, the
form the HF CLIP repositories carry); `find_bpe_file(cache_dir)` is where `get_kandinsky2(cache_dir=...)` looks for it.  Checked in
tests/test_host_cpu.py against transformers' independent implementation of the same algorithm (`CLIPTokenizer`) on a generated
merges list.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import List, Optional, Tuple

import torch

N_MERGES = 49152 - 256 - 2   # merges the OpenAI vocabulary keeps (the file holds more lines)
BPE_FILE_NAMES = ("bpe_simple_vocab_16e6.txt.gz", "bpe_simple_vocab_16e6.txt", "merges.txt")


@lru_cache()
def bytes_to_unicode():
    """byte -> printable unicode character: the printable latin-1 ranges map to themselves, the other 68 bytes to 256, 257, ..."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _byte_symbols():
    """the 256 byte symbols in the order the vocabulary lists them: the self-mapped bytes first, then the remapped ones"""
    t = bytes_to_unicode()
    keep = [b for b in range(256) if ord(t[b]) < 256]
    rest = [b for b in range(256) if ord(t[b]) >= 256]
    return [t[b] for b in keep + rest]


def _read_merges(path: str) -> List[Tuple[str, str]]:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        lines = f.read().decode("utf-8").split("\n")
    lines = lines[1:N_MERGES + 1]                      # line 0 is the "#version" header
    return [tuple(ln.split()) for ln in lines if len(ln.split()) == 2]


def find_bpe_file(cache_dir: Optional[str]) -> Optional[str]:
    for d in ([cache_dir, os.path.join(cache_dir, "text_encoder")] if cache_dir else []):
        for n in BPE_FILE_NAMES:
            p = os.path.join(d, n)
            if os.path.exists(p):
                return p
    return None


class ClipBPETokenizer:
    def __init__(self, bpe_path: str):
        import regex
        merges = _read_merges(bpe_path)
        if not merges:
            raise ValueError(f"{bpe_path}: no byte-pair merges found")
        base = _byte_symbols()
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {s: i for i, s in enumerate(vocab)}
        self.decoder = {i: s for s, i in self.encoder.items()}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {c: b for b, c in self.byte_encoder.items()}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
        self.sot_token = self.encoder["<|startoftext|>"]
        self.eot_token = self.encoder["<|endoftext|>"]

    # -- byte-pair merging of one regex piece (already mapped to byte symbols) -------------------------------------------
    def bpe(self, piece: str) -> str:
        hit = self.cache.get(piece)
        if hit is not None:
            return hit
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word[:-1], word[1:]):
                r = self.ranks.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    merged.append(best[0] + best[1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[piece] = out
        return out

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).lower()       # \s+ -> one space, strip (str.split() splits on the same unicode whitespace)

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in self.pat.findall(self.clean(text)):
            sym = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(sym).split(" "))
        return ids

    def decode(self, ids) -> str:
        s = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in s).decode("utf-8", errors="replace").replace("</w>", " ")

    # -- the reference's CustomizedTokenizer.padded_tokens_and_mask (prior.py:394-416) -----------------------------------
    def padded_tokens_and_mask(self, texts, text_ctx):
        if not isinstance(texts, list) or not all(isinstance(t, str) for t in texts):
            raise AssertionError("texts should be a list of strings")
        rows = [[self.sot_token] + self.encode(t) + [self.eot_token] for t in texts]
        mask = torch.tensor([[True] * min(text_ctx, len(r)) + [False] * max(text_ctx - len(r), 0) for r in rows], dtype=torch.bool)
        out = torch.zeros(len(rows), text_ctx, dtype=torch.int)
        for i, r in enumerate(rows):
            if len(r) > text_ctx:
                r = r[:text_ctx]
                r[-1] = self.eot_token          # a truncated prompt still ends in <|endoftext|>
            out[i, : len(r)] = torch.tensor(r)
        return out, mask
