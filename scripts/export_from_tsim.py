#!/usr/bin/env python
"""Export real compiled tsim programs to ``.npz`` - run this UNCHANGED wherever ``tsim`` is installed.

    pip install tsim            # (jax, equinox, stim, pyzx-param come with it)
    python export_from_tsim.py --out exported/                      # the north-star circuits (BASELINE.json configs)
    python export_from_tsim.py --stim my_circuit.stim --out exported/   # any tsim circuit text

For every circuit it
  1. builds the ``tsim.Circuit`` (the 35- and 85-qubit magic-state distillation circuits with the reference's own
     encoders, ``tsim/utils/encoder.py:176-260``, following ``docs/demos/magic_state_distillation.ipynb`` cell 20; a
     d = 5 rotated surface code memory with a T gate injected on data qubit 0);
  2. runs the reference's compile pipeline exactly as ``Circuit.compile_detector_sampler`` does
     (``tsim/sampler.py:173-236``: ``prepare_graph`` -> ``compile_program(mode="sequential", strategy=...)``);
  3. writes ``<name>.npz`` with the ``CompiledProgram`` (every array of ``core/types.py:55-107`` /
     ``compile/compile.py:21-37`` / ``compile/terms.py:42-207``, byte for byte) and the noise model
     (``channel_probs`` and ``error_transform`` as ``ChannelSampler`` receives them, ``noise/channels.py:531``);
  4. optionally (``--golden N``) samples N shots with the reference itself (``sampler.sample(N, batch_size=N)``,
     seed as given) and stores them: golden vectors for the parity tests of the MI355X engine.

The only import from this repository is optional: with ``tsim_amd`` importable the file is written by
``tsim_amd.program.save_npz`` (which also validates it); otherwise the same keys are written by the plain-numpy
copy of that function below, so the script has no dependency besides tsim and numpy.

On the MI355X side:
    python bench.py --program exported/distill35.npz                # the benchmark on the real program
    from tsim_amd.sampler import CompiledDetectorSampler
    s = CompiledDetectorSampler.from_npz("exported/distill35.npz", seed=0); s.sample(10**6, batch_size=10**5)
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

THETA = -np.arccos(np.sqrt(1.0 / 3.0)) / np.pi  # distillation angle (magic_state_distillation.ipynb cell 4)

_LEVEL_FIELDS = (
    ("a_phases", "node_phases", "phases"), ("a_params", "node_phases", "params"), ("a_counts", "node_phases", "counts"),
    ("b_coeffs", "halfpi_phases", "coeffs"), ("b_params", "halfpi_phases", "params"),
    ("c_psi_const", "pi_products", "psi_const"), ("c_psi_params", "pi_products", "psi_params"),
    ("c_phi_const", "pi_products", "phi_const"), ("c_phi_params", "pi_products", "phi_params"),
    ("d_alpha", "phase_pairs", "alpha"), ("d_alpha_params", "phase_pairs", "alpha_params"),
    ("d_beta", "phase_pairs", "beta"), ("d_beta_params", "phase_pairs", "beta_params"), ("d_counts", "phase_pairs", "counts"),
    ("p_phase_indices", "prefactor", "phase_indices"), ("p_floatfactor", "prefactor", "floatfactor"),
    ("p_power2", "prefactor", "power2"), ("p_approx", "prefactor", "approximate_floatfactors"),
)
_DTYPES = {"a_counts": np.int32, "d_counts": np.int32, "p_floatfactor": np.int32, "p_power2": np.int32, "p_approx": np.complex64}


def save_npz_plain(path, program, **extra) -> None:
    """The keys ``tsim_amd.program.load_npz`` reads, from a live tsim ``CompiledProgram`` (jax leaves -> numpy)."""
    out = {
        "num_outputs": np.int64(int(program.num_outputs)),
        "num_detectors": np.int64(int(program.num_detectors)),
        "direct_f_indices": np.ascontiguousarray(np.asarray(program.direct_f_indices), dtype=np.int32),
        "direct_flips": np.ascontiguousarray(np.asarray(program.direct_flips), dtype=np.bool_),
        "output_order": np.ascontiguousarray(np.asarray(program.output_order), dtype=np.int32),
        "num_components": np.int64(len(program.components)),
    }
    for ci, comp in enumerate(program.components):
        out[f"c{ci}_output_indices"] = np.asarray([int(i) for i in comp.output_indices], dtype=np.int32)
        out[f"c{ci}_f_selection"] = np.ascontiguousarray(np.asarray(comp.f_selection), dtype=np.int32)
        out[f"c{ci}_num_levels"] = np.int64(len(comp.compiled_scalar_graphs))
        for k, lv in enumerate(comp.compiled_scalar_graphs):
            pre = f"c{ci}_l{k}_"
            out[pre + "num_graphs"] = np.int64(int(lv.num_graphs))
            out[pre + "n_params"] = np.int64(int(lv.n_params))
            out[pre + "has_approx"] = np.bool_(bool(lv.prefactor.has_approximate_floatfactors))
            for key, fam, attr in _LEVEL_FIELDS:
                a = np.asarray(getattr(getattr(lv, fam), attr))
                out[pre + key] = np.ascontiguousarray(a, dtype=_DTYPES.get(key, np.uint8))
            out[pre + "p_floatfactor"] = out[pre + "p_floatfactor"].reshape(-1, 4)
    for k, v in extra.items():
        out["x_" + k] = np.asarray(v)
    np.savez_compressed(path, **out)


def distillation_circuit(code: str, p: float, basis: str = "Z"):
    """5-to-1 magic-state distillation on transversally encoded qubits (ipynb cell 20): 35 qubits with the Steane
    code, 85 with the [[17,1,5]] colour code; preparation noise p, gate noise p / 5."""
    from tsim.utils.encoder import ColorEncoder5, SteaneEncoder

    noise = p / 5
    enc = SteaneEncoder() if code == "steane" else ColorEncoder5()
    enc.initialize(f"""
        R 0 1 2 3 4
        R_X({THETA}) 0 1 2 3 4
        T_DAG 0 1 2 3 4
        DEPOLARIZE1({p}) 0 1 2 3 4
        """)
    enc.encode_transversally(f"""
        SQRT_X 0 1 4
        DEPOLARIZE1({noise}) 0 1 4
        CZ 0 1 2 3
        DEPOLARIZE2({noise}) 0 1 2 3
        SQRT_Y 0 3
        DEPOLARIZE1({noise}) 0 3
        CZ 0 2 3 4
        DEPOLARIZE2({noise}) 0 2 3 4
        TICK
        SQRT_X_DAG 0
        DEPOLARIZE1({noise}) 0
        CZ 0 4
        DEPOLARIZE2({noise}) 0 4
        TICK
        CZ 1 3
        DEPOLARIZE2({noise}) 1 3
        TICK
        SQRT_X_DAG 0 1 2 3 4
        DEPOLARIZE1({noise}) 0 1 2 3 4
        """ + ("H 0" if basis == "X" else "H_YZ 0" if basis == "Y" else "") + """
        M 0 1 2 3 4
        DETECTOR rec[-5]
        DETECTOR rec[-4]
        DETECTOR rec[-3]
        DETECTOR rec[-2]
        DETECTOR rec[-1]
        OBSERVABLE_INCLUDE(0) rec[-5]
        OBSERVABLE_INCLUDE(1) rec[-4]
        OBSERVABLE_INCLUDE(2) rec[-3]
        OBSERVABLE_INCLUDE(3) rec[-2]
        OBSERVABLE_INCLUDE(4) rec[-1]
        """)
    return enc.circuit


def surface_code_with_t(distance: int, p: float):
    """Rotated surface code memory (stim's generator) with a T gate injected on data qubit 0 before the rounds."""
    import stim
    import tsim

    base = stim.Circuit.generated("surface_code:rotated_memory_z", distance=distance, rounds=distance,
                                  after_clifford_depolarization=p, before_measure_flip_probability=p,
                                  after_reset_flip_probability=p, before_round_data_depolarization=p)
    text = str(base)
    data0 = min(int(t) for ln in text.splitlines() if ln.startswith("R ") for t in ln.split()[1:])
    lines, done = [], False
    for ln in text.splitlines():
        lines.append(ln)
        if not done and ln.startswith("R "):
            lines.append(f"T {data0}")
            done = True
    return tsim.Circuit("\n".join(lines))


def export(name: str, circuit, out_dir: str, *, strategy: str, seed: int, golden: int) -> None:
    from tsim.compile.pipeline import compile_program
    from tsim.core.graph import prepare_graph

    prepared = prepare_graph(circuit, sample_detectors=True)
    program = compile_program(prepared, mode="sequential", strategy=strategy)
    probs = [np.asarray(c, dtype=np.float64) for c in prepared.channel_probs]
    extra = {"n_channels": np.int64(len(probs)), "error_transform": np.asarray(prepared.error_transform, dtype=np.uint8),
             "num_f": np.int64(np.asarray(prepared.error_transform).shape[0]), "strategy": np.bytes_(strategy),
             "circuit_text": np.bytes_(str(circuit))}
    for i, c in enumerate(probs):
        extra[f"channel_probs_{i}"] = c
    if golden > 0:
        sampler = circuit.compile_detector_sampler(strategy=strategy, seed=seed)
        det, obs = sampler.sample(shots=golden, batch_size=golden, separate_observables=True)
        extra.update(golden_seed=np.int64(seed), golden_shots=np.int64(golden), golden_detectors=np.packbits(det, axis=1, bitorder="little"),
                     golden_observables=np.packbits(obs, axis=1, bitorder="little"))
    path = os.path.join(out_dir, name + ".npz")
    try:
        from tsim_amd.program import save_npz
        save_npz(path, program, **extra)
    except ImportError:
        save_npz_plain(path, program, **extra)
    graphs = [[int(lv.num_graphs) for lv in c.compiled_scalar_graphs] for c in program.components]
    print(f"{path}: {int(program.num_outputs)} outputs ({int(program.num_detectors)} detectors), {len(np.asarray(program.direct_f_indices))} direct, "
          f"components {graphs}, num_f {int(extra['num_f'])}, {len(probs)} channels"
          + (f", {golden} golden shots (seed {seed})" if golden > 0 else ""))


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default="exported")
    ap.add_argument("--p", type=float, default=1e-3, help="noise strength of the generated circuits (BASELINE configs: 1e-3)")
    ap.add_argument("--strategy", default="cat5", choices=["cat5", "bss", "cutting"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--golden", type=int, default=0, help="also store this many shots sampled by the reference itself")
    ap.add_argument("--stim", action="append", default=[], help="tsim / stim circuit text file(s) to export instead of the built-in circuits")
    ap.add_argument("--only", default="", help="comma-separated subset of: distill35, distill85, surface5_t")
    args = ap.parse_args()
    try:
        import tsim
    except ImportError:
        sys.exit("tsim is not installed here: run this script where `import tsim` works (pip install tsim)")
    os.makedirs(args.out, exist_ok=True)
    kw = dict(strategy=args.strategy, seed=args.seed, golden=args.golden)
    if args.stim:
        for fn in args.stim:
            export(os.path.splitext(os.path.basename(fn))[0], tsim.Circuit(open(fn).read()), args.out, **kw)
        return
    want = set(filter(None, args.only.split(","))) or {"distill35", "distill85", "surface5_t"}
    if "distill35" in want:
        export("distill35", distillation_circuit("steane", args.p), args.out, **kw)
    if "distill85" in want:
        export("distill85", distillation_circuit("color5", args.p), args.out, **kw)
    if "surface5_t" in want:
        export("surface5_t", surface_code_with_t(5, args.p), args.out, **kw)


if __name__ == "__main__":
    main()
