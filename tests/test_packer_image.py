"""The packer's output is pinned word for word: FNV-1a hashes of the device image after the two packing phases of
``tsim_program_finalize`` (``TSIM_AMD_DEBUG=finalize,imghash``) against ``tests/golden/image_hashes.json`` - 46 programs x 2
formulations.  Runs without a GPU (the host phases come first).  See ``tests/golden/gen_image_hashes.py``."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_image_hashes_unchanged():
    import gen_image_hashes as G

    with open(G.OUT) as f:
        want = json.load(f)
    got = G.collect()
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    assert sum(len(v) for v in got.values()) >= 2 * len(got), "the finalize marks are missing (TSIM_AMD_DEBUG=finalize,imghash)"
    bad = [(k, ph, got[k].get(ph), w) for k, v in want.items() for ph, w in v.items() if got[k].get(ph) != w]
    assert not bad, bad[:5]
