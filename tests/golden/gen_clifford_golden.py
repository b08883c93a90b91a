"""Regenerates tests/golden/clifford_golden.npz.

SELF-GENERATED fixture (not reference output: Stim / pyzx are not available here): the compiled form
of two generated circuits as produced by tsim_amd.clifford at the time the front-end passed its
independent checks (explicit-Pauli tableau replay, frame Monte Carlo, the reference's seeded KATs).
It pins the conventions that decide bit-exact agreement with the reference's sampler stream - error
variable order, channel tables, f basis, direct table - against accidental change.

    python tests/golden/gen_clifford_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tsim_amd.circuits import repetition_code_memory, rotated_surface_code_memory  # noqa: E402
from tsim_amd.clifford import CliffordCircuit  # noqa: E402

CASES = {
    "surface_d3_r3_x": rotated_surface_code_memory(3, 3, basis="X", after_clifford_depolarization=0.001,
                                                   before_round_data_depolarization=0.002,
                                                   before_measure_flip_probability=0.003,
                                                   after_reset_flip_probability=0.004),
    "repetition_d5_r4": repetition_code_memory(5, 4, before_round_data_flip=0.02, measure_flip=0.01),
}


def compiled(text):
    program, probs, et = CliffordCircuit(text).compile()
    return dict(
        direct_f_indices=np.asarray(program.direct_f_indices, np.int32),
        direct_flips=np.asarray(program.direct_flips, np.uint8),
        output_order=np.asarray(program.output_order, np.int32),
        error_transform=np.packbits(et, axis=1, bitorder="little"),
        et_shape=np.asarray(et.shape, np.int64),
        channel_sizes=np.asarray([len(p) for p in probs], np.int32),
        channel_probs=np.concatenate([np.asarray(p, np.float64) for p in probs]),
        sample_seed7=CliffordCircuit(text).compile_detector_sampler(seed=7).sample(64, append_observables=True),
    )


if __name__ == "__main__":
    out = {}
    for name, text in CASES.items():
        for k, v in compiled(text).items():
            out[f"{name}.{k}"] = v
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "clifford_golden.npz"), **out)
    print("wrote", len(out), "arrays")
