"""The N > 1 code path of bench.py (RCCL collection of the bit-packed rows through libtsim_hip.so's own
communicator, lanes joined on the batch lane) forced on one GPU: it must run to completion and print the
contract's JSON line.  The real N > 1 runs are the driver's; this keeps the plumbing from rotting between
rounds.  (Named test_zz_* so that it runs last: a subprocess with its own HIP context.)"""
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


@pytest.mark.parametrize("mode", ["alltoall", "root0", "auto"])
def test_bench_distributed_path_on_one_gpu(hip, mode):
    env = dict(os.environ, TSIM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if mode == "auto":  # the defaults: spread roots, group size from --steps (8 batches per collective at 21 steps)
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
    assert ("all-to-all" in d["config"]["sharding"]) == (mode != "root0")
    if mode == "auto":
        assert "every 8 batches" in d["config"]["sharding"]
    assert d["roofline"]["achieved"] > 0 and d["roofline"]["launches"] >= 8
    assert "no torch.distributed" in d["config"]["sharding"] and d["repeats"] >= 1
