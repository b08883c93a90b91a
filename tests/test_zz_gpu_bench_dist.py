"""The N > 1 code path of bench.py (RCCL collection of the bit-packed rows through libtsim_hip.so's own
communicator, lanes joined on the batch lane) forced on one GPU: it must run to completion, print the
contract's JSON line, and - TSIM_BENCH_VERIFY=1 - the bytes that arrive in the receive buffers must equal the
rows of a serial full-kernel run, for a complete gather group and for a PARTIAL one (exact count, no stale tail),
with both collectives.  The real N > 1 runs are the driver's; this keeps the plumbing honest between rounds.
(Named test_zz_* so that it runs last: a subprocess with its own HIP context.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["alltoall", "root0", "default", "measured"])
def test_bench_distributed_path_on_one_gpu(hip, mode):
    env = dict(os.environ, TSIM_BENCH_FORCE_DIST="1", TSIM_BENCH_VERIFY="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if mode == "default":  # the defaults: the north star's gather to rank 0, group size from --steps (8 batches at 21 steps)
        env.pop("TSIM_BENCH_GATHER", None)
        env.pop("TSIM_BENCH_GATHER_EVERY", None)
    else:
        env.update(TSIM_BENCH_GATHER=mode, TSIM_BENCH_GATHER_EVERY="8")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "21", "--warmup", "2", "--shots", "200000",
           "--no-cpu-baseline", "--no-full-leg"]
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        pytest.fail("bench.py (forced distributed path) did not finish in 15 minutes")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 21 and d["value"] > 0 and d["scaling"] == "weak"
    if mode == "measured":  # the N > 1 default: both collectives timed in place, the faster one used (and its bytes checked below)
        cal = d["gather_calibration"]
        assert cal["root0_ms"] > 0 and cal["alltoall_ms"] > 0 and cal["chosen"] in ("root0", "alltoall")
        assert ("all-to-all" in d["config"]["sharding"]) == (cal["chosen"] == "alltoall")
    else:
        assert ("all-to-all" in d["config"]["sharding"]) == (mode == "alltoall")
    if mode == "default":
        assert "every 8 batches" in d["config"]["sharding"] and "gather to rank 0" in d["config"]["sharding"]
    assert d["roofline"]["achieved"] > 0 and d["roofline"]["launches"] >= 3
    assert "no torch.distributed" in d["config"]["sharding"] and d["repeats"] >= 1
    # the collected bytes were compared with a serial full-kernel run: 8 batches of the complete group + 5 of the partial one
    assert d["verify"]["ok"] is True and d["verify"]["batches_checked_on_rank0"] == 13


def test_bench_refuses_a_world_size_it_was_not_asked_for(hip):
    """`--gpus 2` under a launcher that provides one rank: an error, not an n_gpus = 1 line."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--no-cpu-baseline", "--no-extra-legs"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_on_an_exported_program_with_its_own_noise_model(hip, tmp_path):
    """`bench.py --program x.npz`: the file scripts/export_from_tsim.py writes (here: the C2 shape saved through the same
    save_npz, with a channel_probs / error_transform noise model) - f batches are drawn from the file's noise model."""
    import numpy as np

    from tsim_amd import synth
    from tsim_amd.program import save_npz

    prog, cfg = synth.config_program("C2")
    nf = cfg["num_f"]
    probs = {f"channel_probs_{i}": np.array([0.99, 0.01]) for i in range(nf)}
    path = str(tmp_path / "exported.npz")
    save_npz(path, prog, n_channels=np.int64(nf), error_transform=np.eye(nf, dtype=np.uint8), num_f=np.int64(nf), **probs)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--program", path, "--steps", "6", "--warmup", "2", "--shots", "200000",
                        "--repeats", "2", "--spinup-ms", "0", "--no-cpu-baseline", "--no-extra-legs"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["value"] > 0 and "exported program" in d["config"]["variant"] and "its own noise model" in d["config"]["workload"]
