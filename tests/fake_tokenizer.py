"""A tiny stand-in for CLIPTokenizer (no vocabulary files exist offline): words and punctuation marks become ids by
first appearance, ``<|...|>`` trigger tokens stay whole, and encode() adds begin / end ids like CLIP does.  Shared by
tests/golden/make_golden_prompt.py (which drives the REAL reference functions with it) and tests/test_prompt_utils.py."""
import re

import torch


class FakeTokenizer:
    model_max_length = 77
    pad_token_id = 2
    bos_token_id, eos_token_id = 0, 1

    def __init__(self):
        self.vocab = {"<|startoftext|>": 0, "<|endoftext|>": 1, "<|pad|>": 2, "<|image|>": 3, "<|facial|>": 4}

    def convert_tokens_to_ids(self, token):
        return self.vocab.get(token)

    def _ids(self, text):
        out = []
        for tok in re.findall(r"<\|\w+\|>|\w+|[^\w\s]", text.lower()):
            out.append(self.vocab.setdefault(tok, len(self.vocab)))
        return out

    def encode(self, text):
        return [self.bos_token_id] + self._ids(text) + [self.eos_token_id]

    def __call__(self, text, max_length=None, padding=None, truncation=False, return_tensors=None):
        ids = self.encode(text)
        if truncation and max_length and len(ids) > max_length:
            ids = ids[:max_length]
        if padding == "max_length" and max_length and len(ids) < max_length:
            ids = ids + [self.pad_token_id] * (max_length - len(ids))

        class Out:
            input_ids = torch.tensor([ids], dtype=torch.long)
        return Out
