"""Generate tests/golden/channels_golden.npz from the REFERENCE's own ChannelSampler.

Run in the build container only (needs /root/reference; the module depends on numpy alone and is
loaded by file path, so the jax-dependent package __init__ is never imported):

    python tests/golden/gen_channels_golden.py

The fixture is data only - seeded inputs (channel probability arrays, error_transform) and the
reference's outputs (simplified channel tables, sampled f arrays).
"""

import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/src/tsim/noise/channels.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "channels_golden.npz")


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_channels", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_channels"] = mod
    spec.loader.exec_module(mod)
    return mod


def random_case(rng, ref, n_channels, num_f):
    probs, nbits = [], 0
    for _ in range(n_channels):
        kind = rng.integers(0, 5)
        p = rng.uniform(0.001, 0.08)
        if kind == 0:
            pr = ref.error_probs(p)
        elif kind == 1:
            pr = ref.pauli_channel_1_probs(p, p / 2, p / 3)
        elif kind == 2:
            q = rng.uniform(0.0005, 0.004, size=15)
            pr = ref.pauli_channel_2_probs(*q)
        elif kind == 3:
            pr = ref.heralded_pauli_channel_1_probs(p, p / 2, p / 4, p / 5)
        else:
            pr = ref.correlated_error_probs([p, p / 2, p / 3])
        probs.append(pr)
        nbits += int(np.log2(len(pr)))
    # error_transform with repeated, zero and nested columns so every simplification pass fires
    base = (rng.random((num_f, max(3, nbits // 3))) < 0.35).astype(np.uint8)
    cols = []
    for _ in range(nbits):
        r = rng.random()
        if r < 0.12:
            cols.append(np.zeros(num_f, np.uint8))
        else:
            cols.append(base[:, rng.integers(0, base.shape[1])])
    T = np.stack(cols, axis=1)
    return probs, T


def main():
    ref = load_reference()
    rng = np.random.default_rng(20260928)
    out = {}
    # (channels, num_f, distinct columns): the last entry, when set, rebuilds error_transform from that few
    # distinct columns plus zero columns, so that duplicate folding, multi-bit marginalisation, merging and
    # subset absorption all fire many times (added in round 2, after the six original cases)
    cases = [(3, 4, 0), (6, 5, 0), (10, 8, 0), (16, 12, 0), (25, 20, 0), (40, 33, 0),
             (8, 3, 2), (14, 6, 2), (20, 7, 3), (30, 9, 3), (12, 70, 4), (24, 130, 5)]
    out["n_cases"] = np.int64(len(cases))
    for ci, (nch, num_f, distinct) in enumerate(cases):
        probs, T = random_case(rng, ref, nch, num_f)
        if distinct:
            base = (rng.random((num_f, distinct)) < 0.5).astype(np.uint8)
            T = np.stack([base[:, rng.integers(0, distinct)] if rng.random() > 0.3 else np.zeros(num_f, np.uint8)
                          for _ in range(T.shape[1])], axis=1)
        seed = 1000 + ci
        s = ref.ChannelSampler(probs, T, seed=seed)
        out[f"c{ci}_n_channels"] = np.int64(nch)
        for i, p in enumerate(probs):
            out[f"c{ci}_probs{i}"] = p
        out[f"c{ci}_transform"] = T
        out[f"c{ci}_seed"] = np.int64(seed)
        out[f"c{ci}_signature_matrix"] = s.signature_matrix
        out[f"c{ci}_n_simplified"] = np.int64(len(s.channels))
        for i, ch in enumerate(s.channels):
            out[f"c{ci}_s{i}_probs"] = ch.probs
            out[f"c{ci}_s{i}_ids"] = np.asarray(ch.unique_col_ids, dtype=np.int64)
        # three consecutive draws: the RNG stream must be consumed identically
        out[f"c{ci}_sample_a"] = s.sample(257)
        out[f"c{ci}_sample_b"] = s.sample(1)
        out[f"c{ci}_sample_c"] = s.sample(4096)
    # the reference's seeded detector KAT (test/integration/test_sampler_circuits.py:25-37):
    # X_ERROR(0.3), det sampler seed=1 -> channel seed = default_rng(1).integers(0, 2**30)
    ch_seed = int(np.random.default_rng(1).integers(0, 2**30))
    s = ref.ChannelSampler([ref.error_probs(0.3)], np.array([[1]], dtype=np.uint8), seed=ch_seed)
    out["kat_channel_seed"] = np.int64(ch_seed)
    out["kat_sample10"] = s.sample(10)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
