"""consistentid_amd.prompt_utils against vectors produced by the REAL reference functions
(tests/golden/make_golden_prompt.py ran /root/reference/functions.py:23-205 and replayed
pipline_StableDiffusion_ConsistentID.py:311-347 on them)."""
import json
from pathlib import Path

import torch

from consistentid_amd import prompt_utils as pu
from fake_tokenizer import FakeTokenizer

GOLD = json.loads((Path(__file__).parent / "golden" / "prompt_utils.json").read_text())


def test_dedup_and_first_sentence():
    for c in GOLD["dedup"]:
        assert pu.remove_duplicate_keywords(c["text"]) == c["out"]
        assert pu.extract_first_sentence(c["text"]) == c["first"]


def test_process_text_with_markers():
    assert len(GOLD["markers"]) >= 40
    for c in GOLD["markers"]:
        d = {k: i for i, k in enumerate(c["keys"])}
        text, left = pu.process_text_with_markers(c["text"], d)
        assert text == c["out"], (c["text"], c["keys"])
        assert list(left.keys()) == c["left"] and left is d          # the caller's dict, pruned in place


def test_tokenize_and_token_positions():
    tok = FakeTokenizer()
    for c in GOLD["tokenize"]:
        ids, im, fm = pu.tokenize_and_mask_noun_phrases_ends(c["text"], tok.convert_tokens_to_ids("<|image|>"),
                                                             tok.convert_tokens_to_ids("<|facial|>"), tok)
        assert ids.tolist() == c["ids"] and ids.dtype == torch.long and ids.shape == (1, 77)
        assert im.int().tolist() == c["image_mask"] and fm.int().tolist() == c["facial_mask"] and fm.dtype == torch.bool
        idx = pu.prepare_image_token_idx(im, fm, 2, 5)
        assert [t.int().tolist() for t in idx] == c["idx"]


def test_encode_prompt_with_trigger_word():
    for c in GOLD["encode"]:
        tok = FakeTokenizer()
        d = {k: i for i, k in enumerate(c["keys"])}
        text_only, ids, left, fmask, fidx, fidx_mask = pu.encode_prompt_with_trigger_word(tok, c["prompt"], c["caption"], d,
                                                                                          num_id_images=1)
        assert text_only == c["text_only"]
        assert ids.tolist() == c["ids"] and list(left.keys()) == c["left"]
        assert fmask.int().tolist() == c["facial_mask"]
        assert fidx.tolist() == c["facial_idx"] and fidx_mask.int().tolist() == c["facial_idx_mask"]
