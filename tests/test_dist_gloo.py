"""world_size-2 (and 3) gloo tests of the shot-sharding path on CPU.

There is no GPU here, so the per-shard ``sample_fn`` is the C oracle (tests may use it as the
checker); what is under test is the sharding/offset/gather logic of tsim_amd/dist.py: the
gathered result must equal the unsharded one bit for bit.
"""

import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tsim_amd import dist as tdist
from tsim_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    import torch.distributed as dist

    from oracle import oracle_c as OC

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prog, cfg = synth.config_program("C2")
        f = synth.synth_f(B, cfg["num_f"], 0.05, seed=9)
        op = OC.OracleProgram(prog)

        def sample_fn(program, f_rows, key, shot_offset):
            return op.sample_program(f_rows, key, shot_offset=shot_offset, threads=1)

        out = tdist.sample_program_sharded(prog, f, (21, 22), sample_fn=sample_fn)
        if rank == 0:
            full = op.sample_program(f, (21, 22), threads=2)
            q.put(bool(np.array_equal(out.view(np.bool_), full)) and out.shape == (B, prog.num_outputs))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize(("world", "B"), [(2, 1001), (3, 10), (2, 1)])
def test_sharded_equals_unsharded_gloo(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_bounds():
    assert tdist.shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert tdist.shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert tdist.shard_bounds(0, 2) == [(0, 0), (0, 0)]
    for B in (1, 7, 1000, 1 << 20):
        for R in (1, 2, 4, 8):
            b = tdist.shard_bounds(B, R)
            assert b[0][0] == 0 and b[-1][1] == B and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    with pytest.raises(ValueError):
        tdist.shard_bounds(5, 0)
