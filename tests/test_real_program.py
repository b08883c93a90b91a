"""The consumer side of the exporter's golden shots (``tsim_amd.golden``), without a GPU: the file plumbing - ``save_npz``
of a duck-typed foreign program with its noise model and golden rows, ``load_npz``, seed / key chain / channel seed
derivation (reference ``src/tsim/sampler.py:198-203``), batch arithmetic (``:377-399``), detector / observable split and
bit-packing (``:850-868``) - with the reference's orchestration restated by the oracle on both sides of the file."""

import numpy as np
import pytest

import tsim_amd.sampler as sampler_module
from foreign import to_foreign
from oracle import oracle_np as O
from tsim_amd import golden, synth
from tsim_amd.channels import error_probs, pauli_channel_1_probs
from tsim_amd.program import save_npz
from tsim_amd.sampler import CompiledDetectorSampler


def oracle_sample_program(program, f_params, key):
    return O.sample_program(program, np.asarray(f_params), key)


def write_stand_in(path, monkeypatch, *, shots=257, seed=5, name="C2"):
    """What scripts/export_from_tsim.py writes, from the stand-in: the "reference" that samples the golden rows is the
    numpy oracle behind the reference's own orchestration (numpy channel stream, oracle sample_program)."""
    prog, cfg = synth.config_program(name)
    nf = cfg["num_f"]
    probs = [error_probs(0.03)] * (nf - 1) + [pauli_channel_1_probs(0.01, 0.02, 0.005)]
    et = np.concatenate([np.eye(nf, dtype=np.uint8), np.eye(nf, dtype=np.uint8)[:, :1]], axis=1)  # nf + 1 error bits (the last channel has two)
    monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)
    ref = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=et, seed=seed)
    ref._channel_sampler._native = None  # numpy's own generator: the reference's stream by construction
    det, obs = ref.sample(shots, batch_size=shots, separate_observables=True)
    monkeypatch.undo()
    extra = {"n_channels": np.int64(len(probs)), "error_transform": et, "num_f": np.int64(nf)}
    for i, c in enumerate(probs):
        extra[f"channel_probs_{i}"] = np.asarray(c, np.float64)
    extra.update(golden_seed=np.int64(seed), golden_shots=np.int64(shots), golden_detectors=np.packbits(det, axis=1, bitorder="little"),
                 golden_observables=np.packbits(obs, axis=1, bitorder="little"))
    save_npz(path, to_foreign(prog), **extra)
    return prog


def test_golden_check_through_the_oracle_seam(tmp_path, monkeypatch):
    path = tmp_path / "stand_in.npz"
    write_stand_in(path, monkeypatch)
    monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)
    rep = golden.check_golden(path)
    assert rep["ok"] and rep["mismatching_shots"] == 0 and rep["verdict"] == "identical", rep


def test_golden_check_reports_a_wrong_seed_as_systematic(tmp_path, monkeypatch):
    path = tmp_path / "stand_in.npz"
    write_stand_in(path, monkeypatch)
    z = dict(np.load(path))
    z["x_golden_seed"] = np.int64(6)
    np.savez_compressed(path, **z)
    monkeypatch.setattr(sampler_module, "sample_program", oracle_sample_program)
    rep = golden.check_golden(path)
    assert not rep["ok"] and rep["mismatching_fraction"] > 0.01 and rep["verdict"].startswith("systematic"), rep


def test_files_without_golden_rows_say_so(tmp_path):
    prog, _ = synth.config_program("C2")
    save_npz(tmp_path / "plain.npz", prog)
    with pytest.raises(KeyError):
        golden.check_golden(tmp_path / "plain.npz")
