"""API-surface parity of the prompt glue: build_inputs / process_response of the drop-in
class reproduce the reference's outputs captured in tests/golden/g6_prompt.json (tools/gen_prompt_goldens.py),
token for token and error text for error text.  Pure host logic: no GPU, no HIP library needed."""
import json
import os

import pytest

from tests.fake_tokenizer import SPECIALS, FakeTokenizer
from valley_amd.valley_model import ValleyLlamaForCausalLM as M

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g6_prompt.json")))
CASES = {
    "video": [{"role": "system", "content": "You are Valley."}, {"role": "user", "content": "Describe this video concisely.\n<video>"}],
    "image_multi_turn": [{"role": "system", "content": "sys"}, {"role": "user", "content": "what is in <image> ?"},
                         {"role": "assistent", "content": "a cat"}, {"role": "user", "content": "and here <video> now"}],
}


def tok():
    t = FakeTokenizer()
    t.add_tokens(SPECIALS, special_tokens=True)
    return t


@pytest.mark.parametrize("case", list(CASES))
def test_build_inputs_matches_reference(case):
    t = tok()
    ids = M.build_inputs(None, t, CASES[case]).input_ids
    assert ids == G["build_inputs"][case]
    assert t.padding_side == "left"                     # the reference flips the tokenizer to left padding (:400)
    row = ids[0]
    s = t.special
    i = row.index(s["<im_start>"])
    assert row[i + 1:i + 257] == [s["<im_patch>"]] * 256 and row[i + 257] == s["<im_end>"]
    assert row[i + 258] == s["<vi_start>"] and row[i + 259:i + 267] == [s["<vi_frame>"]] * 8 and row[i + 267] == s["<vi_end>"]


def test_build_inputs_errors_match_reference():
    for k, m in {"no_video": [{"role": "user", "content": "hello"}], "bad_role": [{"role": "robot", "content": "<video>"}]}.items():
        with pytest.raises(ValueError) as e:
            M.build_inputs(None, tok(), m)
        assert G["errors"][k] == f"ValueError: {e.value}"


def test_process_response_matches_reference():
    assert M.process_response(None, G["process_response"]["inputs"]) == G["process_response"]["outputs"]


def test_cli_surface_matches_the_reference_entry_point():
    """valley/inference/run_valley.py keeps the reference's names (init_vision_token, main, DEFAULT_SYSTEM) and its five
    options; init_vision_token binds the six token ids where the splice reads them (reference run_valley.py:13-18)."""
    from types import SimpleNamespace

    import valley.inference.run_valley as rv
    from tests.fake_tokenizer import SPECIALS, FakeTokenizer
    tok = FakeTokenizer()
    tok.add_tokens(SPECIALS, special_tokens=True)
    cfg = SimpleNamespace()
    model = SimpleNamespace(get_model=lambda: SimpleNamespace(vision_tower=SimpleNamespace(config=cfg)))
    rv.init_vision_token(model, tok)
    want = dict(zip(("im_patch_token", "vi_frame_token", "im_start_token", "im_end_token", "vi_start_token", "vi_end_token"),
                    tok.convert_tokens_to_ids(SPECIALS)))
    assert vars(cfg) == want
    args = rv.parse_args([])
    assert args.query == "Describe this video concisely.\n<video>" and args.system_prompt == "" and args.vision_tower is None
    assert args.model_name.endswith("stable-valley-13b-v1") and args.video_file.endswith(".mp4")
    assert rv.DEFAULT_SYSTEM.startswith("You are Valley, a large language and vision assistant") and callable(rv.main)
