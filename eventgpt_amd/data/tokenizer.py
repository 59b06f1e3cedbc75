"""Tokenizer glue.

``tokenizer_event_token`` splits a prompt on ``<event>``, tokenises the chunks
and re-joins them with the ``EVENT_TOKEN_INDEX`` sentinel, keeping exactly one
BOS (behaviour of reference: common/common.py:43-62).

``ByteTokenizer`` is a dependency-free tokenizer with the LLaMA special-token
layout (unk=0, bos=1, eos=2, bytes at 3..258) used for random-init / synthetic
runs and tests, where no sentencepiece model exists.  Real checkpoints load
their sentencepiece tokenizer through ``load_tokenizer``.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional, Sequence, Union

import torch

from ..constants import DEFAULT_EVENT_TOKEN, EVENT_TOKEN_INDEX


def tokenizer_event_token(prompt: str, tokenizer, event_token_index: int = EVENT_TOKEN_INDEX,
                          return_tensors: Optional[str] = None):
    chunks = [tokenizer(c).input_ids for c in prompt.split(DEFAULT_EVENT_TOKEN)]
    has_bos = bool(chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id)
    ids: List[int] = [chunks[0][0]] if has_bos else []
    skip = 1 if has_bos else 0
    for i, chunk in enumerate(chunks):
        if i > 0:
            ids.append(event_token_index)
        ids.extend(chunk[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


class _Encoding:
    def __init__(self, input_ids):
        self.input_ids = input_ids
        self.attention_mask = [1] * len(input_ids)

    def __getitem__(self, k):
        return getattr(self, k)


class ByteTokenizer:
    """UTF-8 byte tokenizer padded to an arbitrary vocabulary size."""

    unk_token_id = 0
    bos_token_id = 1
    eos_token_id = 2
    pad_token_id = 0
    _BYTE_BASE = 3

    def __init__(self, vocab_size: int = 32000, add_bos: bool = True):
        if vocab_size < self._BYTE_BASE + 256:
            raise ValueError("vocab_size must be at least 259")
        self.base_vocab_size = vocab_size
        self.add_bos = add_bos
        self.added: Dict[str, int] = {}

    # -- vocabulary ---------------------------------------------------------
    def __len__(self):
        return self.base_vocab_size + len(self.added)

    @property
    def vocab_size(self):
        return self.base_vocab_size

    def add_tokens(self, tokens: Iterable[str], special_tokens: bool = False) -> int:
        n = 0
        for t in tokens:
            if t not in self.added:
                self.added[t] = len(self)
                n += 1
        return n

    def convert_tokens_to_ids(self, token: str) -> int:
        return self.added.get(token, self.unk_token_id)

    def get_vocab(self) -> Dict[str, int]:
        """Specials, the 256 byte tokens (``<0xNN>`` like sentencepiece byte fallback)
        and the added tokens; unused filler ids have no string."""
        v = {"<unk>": self.unk_token_id, "<s>": self.bos_token_id, "</s>": self.eos_token_id}
        v.update({f"<0x{b:02X}>": b + self._BYTE_BASE for b in range(256)})
        v.update(self.added)
        return v

    # -- encode / decode ----------------------------------------------------
    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids: List[int] = [self.bos_token_id] if (self.add_bos and add_special_tokens) else []
        pos = 0
        specials = sorted(self.added, key=len, reverse=True)
        while pos < len(text):
            for s in specials:
                if text.startswith(s, pos):
                    ids.append(self.added[s])
                    pos += len(s)
                    break
            else:
                ids.extend(b + self._BYTE_BASE for b in text[pos].encode("utf-8"))
                pos += 1
        return ids

    def __call__(self, text, add_special_tokens: bool = True, return_tensors=None, **_):
        if isinstance(text, (list, tuple)):
            return _Encoding([self.encode(t, add_special_tokens) for t in text])
        ids = self.encode(text, add_special_tokens)
        if return_tensors == "pt":
            return _Encoding(torch.tensor([ids], dtype=torch.long))
        return _Encoding(ids)

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        inv = {v: k for k, v in self.added.items()}
        out = bytearray()
        pieces: List[str] = []
        for i in ids:
            if self._BYTE_BASE <= i < self._BYTE_BASE + 256:
                out.append(i - self._BYTE_BASE)
                continue
            if out:
                pieces.append(out.decode("utf-8", errors="replace"))
                out = bytearray()
            if i in inv and not skip_special_tokens:
                pieces.append(inv[i])
            elif not skip_special_tokens and i in (self.bos_token_id, self.eos_token_id):
                pieces.append("<s>" if i == self.bos_token_id else "</s>")
            # ids outside the byte range and not special decode to nothing
        if out:
            pieces.append(out.decode("utf-8", errors="replace"))
        return "".join(pieces)

    def batch_decode(self, batch, skip_special_tokens: bool = False) -> List[str]:
        if isinstance(batch, torch.Tensor):
            batch = batch.tolist()
        return [self.decode(ids, skip_special_tokens) for ids in batch]

    # -- persistence --------------------------------------------------------
    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "byte_tokenizer.json"), "w") as f:
            json.dump({"vocab_size": self.base_vocab_size, "add_bos": self.add_bos,
                       "added": self.added}, f)

    @classmethod
    def from_pretrained(cls, path: str) -> "ByteTokenizer":
        with open(os.path.join(path, "byte_tokenizer.json")) as f:
            meta = json.load(f)
        tok = cls(meta["vocab_size"], meta.get("add_bos", True))
        tok.added = {k: int(v) for k, v in meta.get("added", {}).items()}
        return tok


def load_tokenizer(model_path: str, vocab_size: int = 32000):
    """Sentencepiece LLaMA tokenizer if the directory has one, else ``ByteTokenizer``."""
    if os.path.exists(os.path.join(model_path, "byte_tokenizer.json")):
        return ByteTokenizer.from_pretrained(model_path)
    if os.path.exists(os.path.join(model_path, "tokenizer.model")) or \
            os.path.exists(os.path.join(model_path, "tokenizer.json")):
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(model_path, use_fast=False)
    return ByteTokenizer(vocab_size)
