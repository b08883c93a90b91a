"""Exported programs with golden shots sampled by the reference (tests/golden/real/*.npz, see its README) through the
product path on the GPU, bit for bit; and - always - the same check on a stand-in file whose golden rows come from the
oracle behind the reference's orchestration, so that the consumer is proven before a real file arrives."""

import glob
import os

import pytest

from test_real_program import write_stand_in
from tsim_amd import golden

pytestmark = pytest.mark.gpu

REAL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "real", "*.npz")))


@pytest.mark.parametrize("path", REAL or [None])
def test_exported_programs_reproduce_the_reference_shots(hip, path):
    if path is None:
        pytest.skip("no exported program in tests/golden/real (tsim is not available in the build image)")
    rep = golden.check_golden(path)
    assert rep["ok"], rep


@pytest.mark.parametrize("name,shots", [("C2", 5000), ("C4", 1200), ("C5", 3000)])
def test_stand_in_file_through_the_hip_path(hip, tmp_path, monkeypatch, name, shots):
    path = tmp_path / f"{name}.npz"
    write_stand_in(path, monkeypatch, shots=shots, seed=11, name=name)
    rep = golden.check_golden(path)
    assert rep["ok"], rep


def test_bench_check_golden_flag(hip, tmp_path, monkeypatch):
    """`bench.py --program x.npz --check-golden`: the JSON line carries the comparison with the file's golden shots."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = tmp_path / "stand_in.npz"
    write_stand_in(path, monkeypatch, shots=2000, seed=3, name="C2")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--program", str(path), "--check-golden", "--steps", "4", "--warmup", "2",
                        "--shots", "100000", "--repeats", "2", "--nf", "4", "--no-cpu-baseline", "--no-extra-legs"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["golden_check"]["ok"] is True and d["golden_check"]["shots"] == 2000, d["golden_check"]
