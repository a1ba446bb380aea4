"""Prompt / trigger-token utilities of the pre-loop: they turn (prompt, face caption, the keys of
the face-parsing result) into the cleaned token ids and the ``<|facial|>`` position masks that FacialEncoder
consumes (``HipIDConditioner(facial_token_mask=...)``).

Plain host Python over strings and small integer tensors -- nothing here touches the GPU.  Behaviour follows the
reference function by function (quirks included, because the checkpoints were trained on its output):

  process_text_with_markers            /root/reference/functions.py:39-117
  tokenize_and_mask_noun_phrases_ends  /root/reference/functions.py:119-164
  prepare_image_token_idx              /root/reference/functions.py:166-205
  encode_prompt_with_trigger_word      /root/reference/pipline_StableDiffusion_ConsistentID.py:311-347

``tokenizer`` is anything with the CLIPTokenizer members the reference uses: ``encode(text)``, ``__call__(text,
max_length=, padding=, truncation=, return_tensors=)`` -> ``.input_ids``, ``model_max_length``, ``pad_token_id``,
``convert_tokens_to_ids(token)``.  tests/test_prompt_utils.py replays vectors produced by the real functions.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch

FACIAL_KEYWORDS = ("face", "ears", "eyes", "nose", "mouth")
# face-parsing region -> the caption word it is tied to, in the reference's scan order
REGION_WORD = (("Face", "face"), ("Left_Ear", "ears"), ("Right_Ear", "ears"), ("Left_Eye", "eyes"),
               ("Right_Eye", "eyes"), ("Nose", "nose"), ("Upper_Lip", "mouth"), ("Lower_Lip", "mouth"))
CLAUSE_END = ",.;"
MAX_CAPTION_CHARS = 330          # ref :329: longer face captions are dropped altogether


def extract_first_sentence(text: str) -> str:
    """functions.py:14-20"""
    dot = text.find(".")
    return (text if dot < 0 else text[:dot + 1]).strip()


def remove_duplicate_keywords(text: str, keywords: Sequence[str] = FACIAL_KEYWORDS) -> str:
    """Keep only the first occurrence (case-insensitive) of every keyword; the text comes back re-joined from its
    word / punctuation tokens with single blanks, a dropped word leaving an empty token behind (functions.py:23-37)."""
    tokens = re.findall(r"\b\w+\b|[.,;!?]", text)
    for kw in keywords:
        seen = False
        for i, tok in enumerate(tokens):
            if tok.lower() == kw.lower():
                if seen:
                    tokens[i] = ""
                seen = True
    return " ".join(tokens)


def process_text_with_markers(text: str, parsing_mask_list: Dict[str, object]) -> Tuple[str, Dict[str, object]]:
    """Face caption -> its clauses about the parsed facial regions, each ending in ``<|facial|>`` right after the
    region word, in region order; regions the caption never names are removed from ``parsing_mask_list`` (IN PLACE,
    like the reference: the caller's dict is the one returned)."""
    text = remove_duplicate_keywords(text)
    words: List[str] = []
    for region, word in REGION_WORD:
        if region in parsing_mask_list and word not in words:
            words.append(word)
    marked = text
    for word in reversed(words):
        new = re.sub(rf"\b{word}\b", f"{word} <|{word}|>", marked, count=1)
        if new == marked:                                   # the caption does not mention this region
            for region, w in REGION_WORD:
                if w == word and region in parsing_mask_list:
                    del parsing_mask_list[region]
        marked = new
    marked = marked.replace("\n", "")
    clauses: List[str] = []
    for word in words:
        marker = f"<|{word}|>"
        lo = marked.find(marker)                            # (-1 when absent: the reference's arithmetic is kept as is)
        hi = lo + len(marker)
        while lo > 0 and marked[lo - 1] not in CLAUSE_END:
            lo -= 1
        while hi < len(marked) and marked[hi] not in CLAUSE_END:
            hi += 1
        clause = marked[lo:hi].strip()
        if clause:
            clauses.append(clause + ",")
            marked = marked[:lo] + marked[hi:]
    out = " ".join(clauses)
    for word in ("face", "ears", "nose", "eyes", "mouth"):
        out = out.replace(f"<|{word}|>", "<|facial|>")
    return out, parsing_mask_list


def tokenize_and_mask_noun_phrases_ends(text: str, image_token_id: Optional[int], facial_token_id: Optional[int], tokenizer):
    """Token ids with the trigger tokens removed + masks of the positions just BEFORE each trigger token
    (``<|image|>`` -> image mask, ``<|facial|>`` -> facial mask), all padded / cut to ``tokenizer.model_max_length``.
    Returns ([1, T] long, [1, T] bool, [1, T] bool)."""
    ids = tokenizer.encode(text)
    img_mask = [False] * len(ids)
    fac_mask = [False] * len(ids)
    clean: List[int] = []
    n_img = 0
    for tok in ids:
        if tok == image_token_id:
            img_mask[len(clean) + n_img - 1] = True
            n_img += 1
        elif tok == facial_token_id:
            fac_mask[len(clean) - 1] = True
        else:
            clean.append(tok)
    T = tokenizer.model_max_length

    def fit(seq, pad):
        return seq[:T] if len(seq) > T else seq + [pad] * (T - len(seq))

    return (torch.tensor(fit(clean, tokenizer.pad_token_id), dtype=torch.long).unsqueeze(0),
            torch.tensor(fit(img_mask, False), dtype=torch.bool).unsqueeze(0),
            torch.tensor(fit(fac_mask, False), dtype=torch.bool).unsqueeze(0))


def _positions(mask: torch.Tensor, at_least: int):
    idx = torch.nonzero(mask, as_tuple=True)[1]
    valid = torch.ones_like(idx, dtype=torch.bool)
    short = at_least - len(idx)
    if short > 0:
        idx = torch.cat([idx, torch.zeros(short, dtype=torch.long)])
        valid = torch.cat([valid, torch.zeros(short, dtype=torch.bool)])
    return idx.unsqueeze(0), valid.unsqueeze(0)


def prepare_image_token_idx(image_token_mask: torch.Tensor, facial_token_mask: torch.Tensor, max_num_objects: int = 2,
                            max_num_facials: int = 5):
    """Mask -> (positions, validity), zero-padded up to the maxima (never truncated).  Returns
    (image_idx, image_idx_mask, facial_idx, facial_idx_mask), each [1, n]."""
    return _positions(image_token_mask, max_num_objects) + _positions(facial_token_mask, max_num_facials)


def encode_prompt_with_trigger_word(tokenizer, prompt: str, face_caption: str, key_parsing_mask_list: Dict[str, object],
                                    image_token: str = "<|image|>", facial_token: str = "<|facial|>",
                                    max_num_facials: int = 5, num_id_images: int = 1):
    """The reference pipeline method of the same name.  Returns (prompt_text_only, clean_input_id [1, T],
    key_parsing_mask_list_align, facial_token_mask [1, T], facial_token_idx [1, n], facial_token_idx_mask [1, n])."""
    caption_align, masks_align = process_text_with_markers(face_caption, key_parsing_mask_list)
    prompt_face = prompt + "Detail:" + caption_align
    n_tok = len(tokenizer(prompt_face, max_length=tokenizer.model_max_length, padding="max_length", truncation=False,
                          return_tensors="pt").input_ids[0])
    if n_tok != 77:                                          # too long for one window: caption first, prompt after
        prompt_face = "Detail:" + caption_align + " Caption:" + prompt
    if len(face_caption) > MAX_CAPTION_CHARS:
        prompt_face = prompt
    prompt_text_only = prompt_face.replace("<|facial|>", "").replace("<|image|>", "")
    facial_id = tokenizer.convert_tokens_to_ids(facial_token)
    clean_ids, image_mask, facial_mask = tokenize_and_mask_noun_phrases_ends(prompt_face, None, facial_id, tokenizer)
    _, _, facial_idx, facial_idx_mask = prepare_image_token_idx(image_mask, facial_mask, num_id_images, max_num_facials)
    return prompt_text_only, clean_ids, masks_align, facial_mask, facial_idx, facial_idx_mask
