"""CPU checks of the analysis helpers under tools/ (they back statements made in DESIGN.md)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lds_swizzles_under_the_documented_bank_model():
    """The shipped 16-row fragment layout is conflict-free for ds_read_b128 (PMC agrees: 0.03 conflict cycles per LDS
    instruction); the 32-row fragments of the 32x32x16 path conflict 2-way under the same key and are
    conflict-free under (row >> 1) & 7 — the reason VLY_MFMA32=2 exists."""
    t = _load("lds_bank_check")
    assert t.frag16(lambda r: r & 7) == 4
    assert t.frag16(lambda r: 0) > 4
    assert t.frag32(lambda r: r & 7) == 8
    assert t.frag32(lambda r: (r >> 1) & 7) == 4
