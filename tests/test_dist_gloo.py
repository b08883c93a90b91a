"""world_size-2 (and 3) gloo tests of the shot-sharding path on CPU.

There is no GPU here, so the per-shard ``sample_fn`` is the C oracle (tests may use it as the
checker) and the transport is a ``gloo`` gather (test infrastructure; the product's transport is the RCCL
communicator inside libtsim_hip.so, tests/test_gpu_dist.py); what is under test is the sharding/offset/padding
logic of tsim_amd/dist.py: the assembled result must equal the unsharded one bit for bit.  Also here: the
socket rendezvous that carries the ncclUniqueId, with a fake id (no RCCL call on a GPU-less box).
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

        def gather(local):
            import torch

            t = torch.from_numpy(local)
            parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
            dist.gather(t, parts, dst=0)
            return None if parts is None else [x.numpy() for x in parts]

        out = tdist.sample_program_sharded(prog, f, (21, 22), rank=rank, world=world, sample_fn=sample_fn, gather=gather)
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


def _rdzv_worker(rank, world, port, q):
    ident = tdist.rendezvous_tcp(rank, world, port=port, timeout=60, make_id=lambda: bytes(range(128)))
    q.put((rank, ident))


@pytest.mark.parametrize("world", [2, 4])
def test_tcp_rendezvous_delivers_the_id_to_every_rank(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in reversed(range(world))]  # rank 0 last
    for p in procs:
        p.start()
    got = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(got) == list(range(world)) and all(v == bytes(range(128)) for v in got.values())


def _stray_then_real(port, q):
    """A stray connection that says nothing, one that says garbage, then rank 1 twice (a retry)."""
    import time

    for payload in (None, b"GET / HTTP/1.0\r\n\r\n"):
        for _ in range(200):
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=1.0)
                break
            except OSError:
                time.sleep(0.05)
        if payload:
            c.sendall(payload)
        c.close()
    a = tdist.rendezvous_tcp(1, 3, port=port, timeout=60)
    b = tdist.rendezvous_tcp(1, 3, port=port, timeout=60)  # the same rank again: served, not counted twice
    q.put((1, a, b))


def test_rendezvous_counts_distinct_ranks_not_connections():
    """Rank 0 keeps serving until ranks 1 AND 2 have the id, whatever else connects in between."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    p0 = ctx.Process(target=_rdzv_worker, args=(0, 3, port, q))
    p1 = ctx.Process(target=_stray_then_real, args=(port, q))
    p0.start()
    p1.start()
    r1 = q.get(timeout=60)
    assert r1[0] == 1 and r1[1] == r1[2] == bytes(range(128))
    assert p0.is_alive()  # still waiting for rank 2
    p2 = ctx.Process(target=_rdzv_worker, args=(2, 3, port, q))
    p2.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in (p0, p1, p2):
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == got[2] == bytes(range(128))


def test_rendezvous_rejects_a_foreign_service():
    import threading

    srv = socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    port = srv.getsockname()[1]

    def serve():
        c, _ = srv.accept()
        c.sendall(b"HTTP/1.1 200 OK\r\n\r\n")
        c.close()

    t = threading.Thread(target=serve)
    t.start()
    with pytest.raises((RuntimeError, TimeoutError)):
        tdist.rendezvous_tcp(1, 2, port=port, timeout=1.5)
    t.join()
    srv.close()
    assert tdist.rendezvous_tcp(0, 1, port=1, make_id=lambda: b"x" * 128) == b"x" * 128  # world 1: no socket at all


def test_bench_launches_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment spawns two ranks itself (one process per GPU,
    tsim_amd.dist's socket rendezvous, no torch.distributed); TSIM_BENCH_LAUNCH_ONLY stops each rank where the RCCL
    communicator would be created - with the same 128 bytes in hand on both.  (No GPU needed up to that point.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["TSIM_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert sorted(d["rank"] for d in lines) == [0, 1] and all(d["world"] == 2 and d["launch_only"] for d in lines)
    assert sorted(d["local_rank"] for d in lines) == [0, 1]
    assert lines[0]["id_sha"] == lines[1]["id_sha"]
    # and a launcher's world size that contradicts --gpus is an error, never a line with another n_gpus
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3"], env=env2, cwd=root,
                        capture_output=True, text=True, timeout=60)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in r2.stderr and not r2.stdout.strip()


def test_no_torch_in_the_product_package():
    """The product path (tsim_amd/) imports neither torch nor jax: RCCL is driven through the C ABI."""
    import pathlib
    import re

    root = pathlib.Path(tdist.__file__).resolve().parent
    for path in root.glob("*.py"):
        text = path.read_text()
        assert not re.search(r"^\s*(import|from)\s+torch\b", text, re.M), path


def _payload(sender: int, step: int, unit: int) -> np.ndarray:
    """What rank `sender` writes for running step `step` of a region: distinguishable bytes."""
    return ((np.arange(unit, dtype=np.int64) * 7 + sender * 37 + step * 101) % 251).astype(np.uint8)


def _group_worker(rank, world, port, mode, q):
    """bench.py's group arithmetic (tsim_amd.dist: gather_group_size, group_pieces, collective_kind, received_layout) over a
    gloo transport: regions of 20, 7 and 64 steps, complete and partial groups, both double-buffer slots."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        unit = 24  # bytes per step and rank (B * row bytes in the benchmark)

        def transport(kind, send, count):
            t = torch.from_numpy(np.ascontiguousarray(send))
            if kind == "root0":
                parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
                dist.gather(t, parts, dst=0)
                return None if parts is None else np.concatenate([x.numpy() for x in parts])
            parts = [torch.empty_like(t) for _ in range(world)]  # (gloo has no all-to-all: every rank's group, then this rank's chunk of each)
            dist.all_gather(parts, t)
            per = count // world * unit
            return np.concatenate([x.numpy()[rank * per:(rank + 1) * per] for x in parts])

        ok = True
        for steps in (20, 7, 64):
            every = tdist.gather_group_size(steps, world, mode == "alltoall")
            col = tdist.GroupCollector(world, rank, every, mode, unit, transport)
            for rep in range(2):  # two regions on one collector: the slot parity carries over, drain() resets the counters
                col.collected.clear()
                done = 0
                for k in (3, steps - 3) if steps > 3 else (steps,):  # a region as two calls: pieces that straddle a group boundary
                    col.steps([_payload(rank, done + i, unit) for i in range(k)])
                    done += k
                col.drain()
                assert sum(c for _, _, c, _ in col.collected) == steps
                for first, kind, count, got in col.collected:
                    layout = tdist.received_layout(kind, count, world, rank)
                    if not layout:
                        ok = ok and got is None
                        continue
                    want = np.concatenate([_payload(s, first + b, unit) for s, fb, nb in layout for b in range(fb, fb + nb)])
                    ok = ok and got is not None and np.array_equal(got, want)
                    ok = ok and (kind == "alltoall") == (mode == "alltoall" and count % world == 0)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("mode", ["root0", "alltoall"])
def test_group_collection_arithmetic_over_gloo(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got) == list(range(world)) and all(got.values()), got


def test_collection_prediction_and_strong_scaling_rows():
    pred = tdist.collection_prediction(8e10, 3)
    assert pred["1"]["root0"] == 8e10
    assert pred["8"]["root0"] < 3.5 * 8e10 < 8 * 8e10 <= pred["8"]["alltoall"] + 1  # the single root caps the node, spread roots do not
    # strong scaling: one global batch cut as SURVEY 8(e) words it
    for B, R in ((1_000_000, 8), (1000, 3)):
        b = tdist.shard_bounds(B, R)
        assert b[0][0] == 0 and b[-1][1] == B
