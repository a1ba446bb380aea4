#!/usr/bin/env python
"""Golden vectors for the prompt / trigger-token utilities, produced by the REAL reference functions
(/root/reference/functions.py:23-205; ``cv2`` shimmed, it is imported but unused by them) driven with
tests/fake_tokenizer.FakeTokenizer, plus the composition of pipline_StableDiffusion_ConsistentID.py:311-347
(``encode_prompt_with_trigger_word`` is a pipeline method: its body is replayed here on the real helper functions).

Run (only where /root/reference exists):  python tests/golden/make_golden_prompt.py  ->  tests/golden/prompt_utils.json
"""
import copy
import importlib.util
import json
import sys
import types
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT.parent))
from fake_tokenizer import FakeTokenizer  # noqa: E402

CAPTIONS = [
    "The person has one nose, two eyes, two ears, and a mouth. The face is round.",
    "A young woman with a bright face, her eyes are blue; her nose is small, and the mouth smiles. Her ears are hidden.",
    "He wears glasses.\nHis eyes look tired, eyes half closed, and his Face is pale",
    "nose",
    "",
    "The man's mouth is open, mouth wide. Two ears, one nose; eyes, face.",
    "A portrait of a person standing in a field with mountains in the background, no features described",
    "face face face, eyes and ears and nose and mouth; face again.",
]
KEYSETS = [
    ["Face", "Left_Ear", "Right_Ear", "Left_Eye", "Right_Eye", "Nose", "Upper_Lip", "Lower_Lip"],
    ["Face", "Left_Eye", "Nose"],
    ["Right_Ear", "Lower_Lip"],
    ["Hair", "Face", "Background"],
    [],
]
PROMPTS = ["A man, in a forest, adventuring", "cinematic photo of a woman img " + "very detailed " * 30]


def main():
    m = types.ModuleType("cv2")
    sys.modules["cv2"] = m
    spec = importlib.util.spec_from_file_location("ref_functions", REF / "functions.py")
    F = importlib.util.module_from_spec(spec)
    sys.modules["ref_functions"] = F
    spec.loader.exec_module(F)
    cases = {"markers": [], "dedup": [], "tokenize": [], "encode": []}
    for cap in CAPTIONS:
        cases["dedup"].append({"text": cap, "out": F.remove_duplicate_keywords(cap, ["face", "ears", "eyes", "nose", "mouth"]),
                               "first": F.extract_first_sentence(cap)})
        for keys in KEYSETS:
            d = {k: i for i, k in enumerate(keys)}
            text, left = F.process_text_with_markers(cap, d)
            cases["markers"].append({"text": cap, "keys": keys, "out": text, "left": list(left.keys())})
    tok = FakeTokenizer()
    for text in ["a photo of a man <|image|> with a round face <|facial|> , blue eyes <|facial|> .",
                 "<|facial|> starts, two <|image|> <|image|> tokens", "no trigger tokens at all",
                 "long " * 90 + "face <|facial|>"]:
        ids, im, fm = F.tokenize_and_mask_noun_phrases_ends(text, tok.convert_tokens_to_ids("<|image|>"),
                                                            tok.convert_tokens_to_ids("<|facial|>"), tok)
        idx = F.prepare_image_token_idx(im, fm, 2, 5)
        cases["tokenize"].append({"text": text, "ids": ids.tolist(), "image_mask": im.int().tolist(), "facial_mask": fm.int().tolist(),
                                  "idx": [t.int().tolist() for t in idx]})
    # encode_prompt_with_trigger_word (pipeline method, ref :311-347) replayed on the real helpers
    for prompt in PROMPTS:
        for cap in CAPTIONS[:3] + ["x" * 400 + " eyes"]:
            for keys in KEYSETS[:2]:
                tok = FakeTokenizer()
                d = {k: i for i, k in enumerate(keys)}
                cap_align, d_align = F.process_text_with_markers(cap, d)
                prompt_face = prompt + "Detail:" + cap_align
                if len(tok(prompt_face, max_length=tok.model_max_length, padding="max_length", truncation=False,
                           return_tensors="pt").input_ids[0]) != 77:
                    prompt_face = "Detail:" + cap_align + " Caption:" + prompt
                if len(cap) > 330:
                    prompt_face = prompt
                text_only = prompt_face.replace("<|facial|>", "").replace("<|image|>", "")
                ids, im, fm = F.tokenize_and_mask_noun_phrases_ends(prompt_face, None, tok.convert_tokens_to_ids("<|facial|>"), tok)
                _, _, fidx, fidx_mask = F.prepare_image_token_idx(im, fm, 1, 5)
                cases["encode"].append({"prompt": prompt, "caption": cap, "keys": keys, "text_only": text_only,
                                        "ids": ids.tolist(), "left": list(d_align.keys()), "facial_mask": fm.int().tolist(),
                                        "facial_idx": fidx.tolist(), "facial_idx_mask": fidx_mask.int().tolist()})
    (OUT / "prompt_utils.json").write_text(json.dumps(cases, indent=0))
    print({k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
