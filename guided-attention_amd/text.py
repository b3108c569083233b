"""Prompt side of the pipeline (callers of the hot path; reproduced, not accelerated).

The reference uses CLIP's tokenizer and text encoder through transformers
(pipeline_guided_attention.py:64-199).  Their vocabulary / weights are not available offline, so:
  * `load_clip(path)` loads the real `CLIPTokenizer` / `CLIPTextModel` from a local diffusers folder;
  * otherwise `WordTokenizer` + `SyntheticTextEncoder` give a deterministic stand-in with CLIP's
    framing (BOS, one id per word, EOS, padding to 77) so that token indices of plain-word prompts
    match CLIP's and every shape downstream is the real one.  Synthetic embeddings carry no semantics;
    results with them are labelled "synthetic" by the bench.
"""
import hashlib
from pathlib import Path
from types import SimpleNamespace

import torch
import torch.nn as nn

BOS, EOS = 49406, 49407


class WordTokenizer:
    model_max_length = 77

    def __init__(self, pad_token_id=EOS):
        self.pad_token_id = pad_token_id
        self._words = {}

    def _id(self, word):
        wid = self._words.get(word)
        if wid is None:
            wid = int.from_bytes(hashlib.sha256(word.encode()).digest()[:4], "little") % 49000 + 256
            self._words[word] = wid
        return wid

    def _ids(self, text):
        return [BOS] + [self._id(w) for w in text.lower().split()] + [EOS]

    def __call__(self, text, padding=None, max_length=None, truncation=False, return_tensors=None):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [self._ids(t) for t in texts]
        if padding == "max_length":
            n = max_length or self.model_max_length
            rows = [(r[:n - 1] + [EOS] if len(r) > n else r) + [self.pad_token_id] * max(0, n - len(r)) for r in rows]
        elif padding == "longest":
            n = max(len(r) for r in rows)
            rows = [r + [self.pad_token_id] * (n - len(r)) for r in rows]
        if return_tensors == "pt":
            ids = torch.tensor(rows, dtype=torch.long)
            return SimpleNamespace(input_ids=ids, attention_mask=torch.ones_like(ids))
        return {"input_ids": rows[0] if isinstance(text, str) else rows}

    def decode(self, token_id):
        if isinstance(token_id, (list, tuple)):
            return " ".join(self.decode(t) for t in token_id)
        token_id = int(token_id)
        for w, i in self._words.items():
            if i == token_id:
                return w
        return {BOS: "<|startoftext|>", EOS: "<|endoftext|>"}.get(token_id, "?")

    def batch_decode(self, ids):
        return [self.decode(list(map(int, row))) for row in ids]


class SyntheticTextEncoder(nn.Module):
    """(B, 77) ids -> (B, 77, dim): per-id pseudo-random unit-variance vectors + positional term, layer-normed."""

    def __init__(self, dim=768, seed=1234):
        super().__init__()
        self.config = SimpleNamespace(hidden_size=dim)
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("basis", torch.randn(4096, dim, generator=g))
        self.register_buffer("pos", torch.randn(77, dim, generator=g) * 0.3)

    @property
    def dtype(self):
        return self.basis.dtype

    def forward(self, input_ids, attention_mask=None):
        ids = input_ids.to(self.basis.device)
        e = self.basis[ids % 4096] + self.basis[(ids // 7) % 4096] * 0.5 + self.pos[: ids.shape[1]][None]
        e = torch.nn.functional.layer_norm(e.float(), e.shape[-1:]).to(self.basis.dtype)
        return (e,)


def load_clip(folder):
    """Real CLIP tokenizer + text encoder from a local diffusers checkpoint folder, else None."""
    folder = Path(folder)
    if not (folder / "tokenizer").is_dir() or not (folder / "text_encoder").is_dir():
        return None
    from transformers import CLIPTextModel, CLIPTokenizer
    return CLIPTokenizer.from_pretrained(folder / "tokenizer"), CLIPTextModel.from_pretrained(folder / "text_encoder")
