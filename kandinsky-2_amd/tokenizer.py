"""CLIP byte-pair tokenizer for the 2.1 prior / CLIP text tower (host side, once per prompt).

The reference builds `kandinsky2.model.prior.CustomizedTokenizer()` (prior.py:387-416): OpenAI `clip`'s `SimpleTokenizer` on the
merges file that ships inside that package (`bpe_simple_vocab_16e6.txt.gz`) plus `padded_tokens_and_mask`.  `clip` is an un-vendored,
un-pinned dependency that is not installed here, so the published algorithm of its `simple_tokenizer.py` is restated:

  text -> ftfy.fix_text (ftfy itself when importable, else the fixes of its default configuration that touch well-encoded text,
          restated in fix_text: uncurled quotes, fullwidth -> ASCII, ligatures, control characters, line breaks, NFC)
       -> html-unescape twice, strip, collapse whitespace, lower-case
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
    def fix_text(text: str) -> str:
        """The reference's basic_clean starts with ftfy.fix_text.  With ftfy installed that is what runs; without it (this image) the
        fixes of ftfy's DEFAULT configuration that change the tokens of well-encoded text are restated: curly quotes -> straight ones
        (uncurl_quotes), fullwidth Latin letters / digits / punctuation -> ASCII and halfwidth katakana -> standard (fix_character_width),
        Latin ligatures -> letters (fix_latin_ligatures), C0 / C1 control characters except tab, newline and carriage return removed
        (remove_control_chars), line breaks \r\n, \r, U+2028, U+2029, U+0085 -> \n (fix_line_breaks), then NFC.  ftfy's mojibake repair
        (text that was decoded with the wrong codec) is NOT restated: such prompts tokenize as they are."""
        try:
            import ftfy
            return ftfy.fix_text(text)
        except ImportError:
            pass
        import unicodedata
        quotes = {0x2018: "'", 0x2019: "'", 0x201a: "'", 0x201b: "'", 0x2032: "'", 0x201c: '"', 0x201d: '"', 0x201e: '"', 0x201f: '"', 0x2033: '"'}
        ligatures = {0xfb00: "ff", 0xfb01: "fi", 0xfb02: "fl", 0xfb03: "ffi", 0xfb04: "ffl", 0xfb05: "ſt", 0xfb06: "st", 0x0132: "IJ", 0x0133: "ij",
                     0x01c4: "DŽ", 0x01c5: "Dž", 0x01c6: "dž", 0x01c7: "LJ", 0x01c8: "Lj", 0x01c9: "lj", 0x01ca: "NJ", 0x01cb: "Nj", 0x01cc: "nj"}
        text = text.replace("\r\n", "\n").replace("\r", "\n").replace("\u2028", "\n").replace("\u2029", "\n").replace("\u0085", "\n")
        out = []
        for ch in text:
            o = ord(ch)
            if o in quotes:
                out.append(quotes[o])
            elif o in ligatures:
                out.append(ligatures[o])
            elif 0xff01 <= o <= 0xff5e:                  # fullwidth ASCII block
                out.append(chr(o - 0xfee0))
            elif o == 0x3000:                            # ideographic space
                out.append(" ")
            elif 0xff61 <= o <= 0xffdc or 0xffe0 <= o <= 0xffee:   # halfwidth CJK punctuation / katakana / hangul, fullwidth signs
                out.append(unicodedata.normalize("NFKC", ch))
            elif (o < 0x20 and ch not in "\t\n") or 0x7f <= o <= 0x9f or o == 0xfeff:
                continue
            else:
                out.append(ch)
        return unicodedata.normalize("NFC", "".join(out))

    @classmethod
    def clean(cls, text: str) -> str:
        """basic_clean + whitespace_clean + lower of the reference's tokenizer (OpenAI CLIP simple_tokenizer, used through
        kandinsky2/model/prior.py:394-416 and the text encoders)."""
        text = html.unescape(html.unescape(cls.fix_text(text))).strip()
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
