#!/usr/bin/env python
"""bench.py - detector shots/sec of the fused sampling kernel on N MI355X.

Contract (see the round prompt): ``python bench.py --gpus N --steps K --warmup W`` prints ONE
JSON line on rank 0.  For N > 1 it is launched by ``torch.distributed.run`` with one rank per
GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

Workload (``config.workload``): BASELINE.json ``configs[1]`` - the 35-qubit magic-state
distillation shape (15 direct detectors + one 5-output component,
sum G = 148 stabiliser terms, num_f = 64, per-bit fire probability 0.02, seed 42), as a seeded
synthetic program because the reference's compile pipeline cannot run here.  A *step* is one
pass of the hot path (``sample_program``) over one batch of ``--shots`` shots PER GPU (weak
scaling: rank r owns in-batch rows [r*shots, (r+1)*shots) of a global batch of N*shots, the
Threefry counter is the global row index, so the sharded result equals the unsharded one).
The packed ``f`` batch is resident in HBM before the timed region starts; for N > 1 every step
ends with the RCCL gather of the packed detector/observable bits to rank 0.

Torch is plumbing only (``torch.distributed`` rendezvous/barrier/gather, ``torch.cuda.synchronize``);
all sampling arithmetic is in ``tsim_amd/libtsim_hip.so``.
"""

from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
VALU_PEAK_TOPS = 78.6  # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one int32 lane-op per lane-clock


def algorithmic_bytes_per_shot(num_f: int, num_outputs: int) -> int:
    """packed f read + packed bits written."""
    return 8 * ((num_f + 63) // 64) + (num_outputs + 7) // 8


def algorithmic_ops_per_shot(program) -> int:
    """sum_levels sum_g [(T_A+T_B+2T_C+2T_D) * W * 2 + (T_A+T_D+5) * 16 + 40]."""
    total = 0
    for comp in program.components:
        for lv in comp.compiled_scalar_graphs:
            W = max(1, (lv.n_params + 63) // 64)
            G = lv.num_graphs
            if G == 0:
                continue
            nA = np.asarray(lv.node_phases.counts, dtype=np.int64)
            nD = np.asarray(lv.phase_pairs.counts, dtype=np.int64)
            nB = (np.asarray(lv.halfpi_phases.coeffs) != 0).sum(axis=1) if lv.halfpi_phases.coeffs.size else np.zeros(G, np.int64)
            pc = lv.pi_products
            if pc.psi_const.size:
                live = (pc.psi_const != 0) | pc.psi_params.any(axis=2)
                live &= (pc.phi_const != 0) | pc.phi_params.any(axis=2)
                nC = live.sum(axis=1)
            else:
                nC = np.zeros(G, np.int64)
            total += int(((nA + nB + 2 * nC + 2 * nD) * W * 2 + (nA + nD + 5) * 16 + 40).sum())
    return total


def valu_block(ref_ops_per_shot: int, shots: int, kernel_s: float, config: str) -> dict:
    """The bound that actually binds: integer VALU issue.

    `reference_algorithm_*` prices the reference's un-reduced algorithm (SURVEY 8(d) op count); the
    kernel executes far fewer instructions than that because the packer reduces the program
    algebraically.  When a rocprofv3 PMC summary of this workload is committed (profiles/), the
    executed VALU wave-instructions per launch are taken from it and turned into issue-cycles per
    instruction per SIMD (gfx950 issues the int ops used here at ~4 cycles per wave-instruction).
    """
    out = {
        "bound": "valu_issue",
        "reference_algorithm_ops_per_shot": ref_ops_per_shot,
        "reference_algorithm_equiv_Tlaneops": ref_ops_per_shot * shots / kernel_s / 1e12,
        "peak_Tlaneops_at_2cyc_issue": VALU_PEAK_TOPS,
    }
    pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        d = json.load(open(pmc))
        if d.get("_config") == config and d.get("_shots") == shots:
            insts = float(d["SQ_INSTS_VALU"])
            simd_cycles = kernel_s * 2.4e9 * 1024
            out.update({
                "executed_valu_wave_insts_per_launch": insts,
                "executed_valu_wave_insts_per_64_shots": insts / (shots / 64.0),
                "cycles_per_valu_inst_per_simd_at_2p4GHz": simd_cycles / insts,
                "valu_issue_busy_frac_at_4cyc": min(1.0, 4.0 * insts / simd_cycles),
                "pmc_source": "profiles/latest_pmc.json (rocprofv3 --pmc SQ_INSTS_VALU, same workload)",
            })
    except Exception:
        pass
    return out


def pmc_traffic_bytes(config: str, shots: int):
    """HBM bytes per launch from the committed PMC summary of this workload (None if absent).

    FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a coalesced stream
    (guides/MI355X_MICROARCH.md, HBM section), hence the factor 2 on the read side.
    """
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        if d.get("_config") == config and d.get("_shots") == shots:
            return (2.0 * float(d["FETCH_SIZE"]) + float(d["WRITE_SIZE"])) * 1024.0
    except Exception:
        pass
    return None


def host_cpu_budget() -> tuple[int, str]:
    """CPUs this process may actually use: min(affinity, cgroup v2 quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} hardware threads visible"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(round(int(quota) / int(period))))
            if q < n:
                note += f", cgroup cpu.max limits the container to {q} CPUs"
                n = q
    except Exception:
        pass
    return n, note


def cpu_baseline(program, cfg, seconds: float = 15.0) -> dict:
    """Time the C oracle (kind "port") on a bounded sample of the same workload on the host cores."""
    from oracle import oracle_c
    from tsim_amd import synth

    oracle_c.build()
    op = oracle_c.OracleProgram(program)
    cores, note = host_cpu_budget()
    f = synth.synth_f(40_000, cfg["num_f"], cfg["p_bit"], seed=cfg["seed"])
    best_rate, threads = 0.0, cores
    for th in sorted({cores, 2 * cores}):  # SMT siblings may or may not help under a quota
        t0 = time.perf_counter()
        op.sample_program(f, (1, 2), threads=th)
        rate = len(f) / (time.perf_counter() - t0)
        if rate > best_rate:
            best_rate, threads = rate, th
    n = int(min(max(best_rate * seconds, 40_000), 8_000_000))
    f = synth.synth_f(n, cfg["num_f"], cfg["p_bit"], seed=cfg["seed"])
    t0 = time.perf_counter()
    op.sample_program(f, (1, 2), threads=threads)
    dt = time.perf_counter() - t0
    # also the single-threaded rate (the reference's ChannelSampler and Python driver are
    # single-threaded) and the numpy restatement that keeps the reference's data movement (byte-per-bit
    # float32 GEMM % 2, materialised lookups, sequential scans) - small samples, a few seconds each
    f1 = synth.synth_f(20_000, cfg["num_f"], cfg["p_bit"], seed=cfg["seed"])
    t1 = time.perf_counter()
    op.sample_program(f1, (1, 2), threads=1)
    single = len(f1) / (time.perf_counter() - t1)
    from oracle import oracle_np

    f2 = synth.synth_f(2048, cfg["num_f"], cfg["p_bit"], seed=cfg["seed"])
    t2 = time.perf_counter()
    oracle_np.sample_program(program, f2, (1, 2))
    faithful = len(f2) / (time.perf_counter() - t2)
    return {
        "value": n / dt,
        "unit": "shots/s",
        "cores": cores,
        "single_thread_value": single,
        "numpy_reference_faithful_value": faithful,
        "kind": "port",
        "sample": f"{n} shots of the same C2 program and f distribution, C oracle (oracle/oracle.c, "
        f"OpenMP over shots, {threads} threads on {cores} CPUs; {note}), {dt:.1f} s wall",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--shots", type=int, default=1_000_000, help="shots per step per GPU")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--p-bit", type=float, default=None, help="override the per-bit fire probability of the synthetic f batch (experiments only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-leg", action="store_true", help="skip the extra timing of the full kernel alone (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    N = max(world, 1)
    use_dist = N > 1 or os.environ.get("TSIM_BENCH_FORCE_DIST") == "1"

    # torch first: its bundled HIP runtime must be the one the process shares
    import torch

    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=N, device_id=torch.device("cuda", local_rank))

    from tsim_amd import backend, prng, synth

    program, cfg = synth.config_program(args.config)
    if args.p_bit is not None:
        cfg = dict(cfg, p_bit=float(args.p_bit))
    hp = backend.HipProgram(program, device=local_rank)
    info = hp.info()
    num_f, n_out = cfg["num_f"], program.num_outputs
    B = int(args.shots)
    WF, WO = (num_f + 63) // 64, (n_out + 63) // 64

    # synthetic packed f batch of this rank's shard, resident in HBM before timing
    f = synth.synth_f(B, num_f, cfg["p_bit"], seed=cfg["seed"] + 1000 * rank)
    f_packed = np.packbits(f, axis=1, bitorder="little")
    pad = WF * 8 - f_packed.shape[1]
    if pad:
        f_packed = np.pad(f_packed, ((0, 0), (0, pad)))
    d_f = hp.malloc(B * WF * 8)
    hp.h2d(d_f, f_packed)
    del f

    # Pipeline of NSLOT lanes (tsim_sample_batch_device_begin/_end): step i runs entirely on the stream of
    # slot i % NSLOT, so its second pass (hard rows, latency-bound) overlaps the first pass of the next
    # steps on the other lanes; one output buffer per slot, no cross-stream event in the steady state.
    # N > 1: when a step is joined its rows are compacted to the reference's bit_packed layout
    # (ceil(n_out/8) bytes per shot instead of the padded 8-byte words) into a group buffer in HBM; every
    # GATHER_EVERY steps ONE asynchronous RCCL gather sends the whole group to rank 0 (fewer, larger
    # collectives: a gather per step would cost more host time than the step itself), double-buffered
    # so that it overlaps the kernels of the next group.
    # Hardware queues: the handle's own stream plus two more sit on three distinct ones; more than 4 busy
    # hardware queues is pathological on this stack (3x slower with GPU_MAX_HW_QUEUES=5 / 8), so lanes +
    # RCCL's stream stay <= 4.  Slots: 16 (two batches of up to 8 launches in flight).  With short hard-row lists the library runs the first passes of
    # consecutive launches on two of the lanes and the hard rows of four launches at a time as ONE grid on
    # the third (deferred second pass); a slot is reused only after its batch is done, so the number of
    # slots - not of lanes - covers the batch latency (3 slots: 46 us per step, 6: 33 us, 8: 22-23 us, 12: 21 us).
    default_slots = 16
    NSLOT = max(1, min(backend.HipProgram.PIPELINE_SLOTS, int(os.environ.get("TSIM_BENCH_SLOTS", str(default_slots)))))
    # batches per collective: a torch.distributed call costs ~0.15 ms of host time, several steps' worth -
    # smaller groups for short runs were measured and are worse (20 steps: 63 us per step with groups of 5,
    # 36 us with one collective at the end)
    GATHER_EVERY = max(1, int(os.environ.get("TSIM_BENCH_GATHER_EVERY", "64")))
    # How the bit-packed rows are collected (N > 1).  "alltoall" (default): every group of GATHER_EVERY
    # batches is split by batch index into N chunks and chunk j of every rank is assembled on rank j with ONE
    # RCCL all-to-all - a gather whose roots are spread over the node.  xGMI is a full mesh of point-to-point
    # links (~77 GB/s per direction and pair): at 4e10 shots/s a rank produces 120 GB/s of rows, and a gather
    # to rank 0 would push all of it through the single link to rank 0 (and everything through rank 0's one
    # PCIe link afterwards); the all-to-all puts 1/N of it on each of the N-1 links and leaves complete batches
    # on every rank, next to all 8 PCIe links.  "root0": the classic gather to rank 0, for comparison.
    GATHER_MODE = os.environ.get("TSIM_BENCH_GATHER", "alltoall")
    if GATHER_MODE == "alltoall":
        GATHER_EVERY = (GATHER_EVERY + N - 1) // N * N
    if use_dist:
        dev = torch.device("cuda", local_rank)
        # the gathers are queued on the lane where the hard-row batches run (results complete there): the
        # first-pass lanes - the handle's own stream is one of them - never wait for a gather or a join
        import ctypes as _C
        _sp = _C.c_void_p()
        if hp._lib.tsim_pipeline_lane_stream(hp._h, 2, _C.byref(_sp)) < 0:
            raise RuntimeError("tsim_pipeline_lane_stream failed")
        join_ptr = int(_sp.value)
        ext = torch.cuda.ExternalStream(join_ptr, device=dev)
        ext_main = torch.cuda.ExternalStream(hp.stream_ptr(), device=dev)  # first-pass lane 0
        out_bufs = [torch.zeros((B, WO * 8), dtype=torch.uint8, device=dev) for _ in range(NSLOT)]
        out_ptrs = [t.data_ptr() for t in out_bufs]
        RB = (n_out + 7) // 8
        grp_bufs = [torch.zeros((GATHER_EVERY, B, RB), dtype=torch.uint8, device=dev) for _ in range(2)]
        if GATHER_MODE == "alltoall":
            grp_recv = [torch.empty_like(grp_bufs[0]) for _ in range(2)]  # [N chunks of GATHER_EVERY/N batches, B, RB]
            grp_lists = [None, None]
        else:
            grp_lists = [[torch.empty_like(grp_bufs[0]) for _ in range(N)] if rank == 0 else None for _ in range(2)]
        grp_ptrs = [t.data_ptr() for t in grp_bufs]
        grp_pending = [None, None]  # gather handle of the group buffer's previous use
        series_fn = hp._lib.tsim_pipeline_set_compact_series
        wait_fn = hp._lib.tsim_pipeline_wait_stream
    else:
        d_outs = [hp.malloc(B * WO * 8) for _ in range(NSLOT)]
        out_ptrs = [d.ptr for d in d_outs]
        d_out = d_outs[0]
    inflight = []  # (slot, step id) begun, not yet joined (N == 1 bookkeeping)

    key = prng.key(cfg["seed"])
    shot_offset = rank * B
    step_no = [0]

    begin_fn = hp._lib.tsim_sample_batch_device_begin
    end_fn = hp._lib.tsim_sample_batch_device_end
    d_f_ptr = d_f.ptr

    gathered = [0]  # groups whose gather has been issued (N > 1)

    def gather_next(count=None):
        """Issue the gather of the oldest un-gathered group (all of its steps must be joined)."""
        k = gathered[0]
        g = k & 1
        count = GATHER_EVERY if count is None else count
        src = grp_bufs[g] if count == GATHER_EVERY else grp_bufs[g][:count]
        dst = None
        if rank == 0 and GATHER_MODE != "alltoall":
            dst = grp_lists[g] if count == GATHER_EVERY else [t[:count] for t in grp_lists[g]]
        with torch.cuda.stream(ext):
            if GATHER_MODE == "alltoall":  # equal chunks: a partial last group is rounded up to a multiple of N batches
                cnt = (count + N - 1) // N * N
                grp_pending[g] = dist.all_to_all_single(grp_recv[g][:cnt], grp_bufs[g][:cnt], async_op=True)
            else:
                grp_pending[g] = dist.gather(src, dst, dst=0, async_op=True)
        gathered[0] = k + 1

    import ctypes as _ct
    key_state = (_ct.c_uint32 * 2)(key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF)  # split in place by the library
    begin_split = hp._lib.tsim_sample_batch_device_begin_split
    h_prog = hp._h

    def step():
        # one host split per batch, key, subkey = split(key) (sampler.py:399), inside the launch call
        j = step_no[0]
        b = j % NSLOT
        step_no[0] = j + 1
        if not use_dist:
            # _begin on a slot whose previous step was not joined is ordered after that step's second
            # pass by the library (include/tsim_hip.h), so one call per step is enough here
            rc = begin_split(h_prog, b, d_f_ptr, B, num_f, key_state, shot_offset, out_ptrs[b], None, None, 1)  # inputs ready
            if rc < 0:
                raise RuntimeError(f"tsim_sample_batch_device_begin failed ({rc})")
            if not inflight or len(inflight) < NSLOT:
                inflight.append((b, j))
            return
        g, pos = (j // GATHER_EVERY) & 1, j % GATHER_EVERY
        if pos == 0 and grp_pending[g] is not None:
            # the gather that last read this group buffer (two groups ago, long finished) must be done before
            # the kernels overwrite it: first-pass lane 0 (the handle's stream) waits for it, the other lanes
            # are ordered after lane 0 below.  (Waiting on the batch lane instead would park the dependency
            # behind every queued hard-row batch and drain the pipeline once per group.)
            with torch.cuda.stream(ext_main):
                grp_pending[g].wait()
            grp_pending[g] = None
        # At the start of a group every lane waits ONCE for the engine's stream, i.e. for the gather that
        # last read this group buffer (tsim_pipeline_wait_stream); the launches themselves then need no
        # cross-stream dependency.  The kernels write the rows a second time in the reference's bit_packed
        # layout, into the group buffer.
        rc = 0
        if pos == 0 or j < NSLOT:  # (the first launches create the lanes)
            rc = wait_fn(hp._h, None)
        if rc >= 0 and pos == 0:  # the launches of this group also write bit_packed rows, one slice each
            rc = series_fn(hp._h, grp_ptrs[g], B * RB, GATHER_EVERY)
        if rc >= 0:
            rc = begin_split(h_prog, b, d_f_ptr, B, num_f, key_state, shot_offset, out_ptrs[b], None, None,
                             0 if j < NSLOT else 1)
        if rc < 0:
            raise RuntimeError(f"pipelined launch failed ({rc})")
        if pos == GATHER_EVERY - 1:  # group complete: join every lane on the engine's stream, then gather
            for k in range(NSLOT):
                end_fn(hp._h, k, join_ptr)
            gather_next()

    def drain():
        if not use_dist:
            while inflight:
                hp.sample_batch_device_end(inflight.pop(0)[0])
            return
        for b in range(NSLOT):
            end_fn(hp._h, b, join_ptr)
        n_steps = step_no[0]
        while gathered[0] * GATHER_EVERY < n_steps:
            gather_next(min(GATHER_EVERY, n_steps - gathered[0] * GATHER_EVERY))
        with torch.cuda.stream(ext):
            for g in range(2):
                if grp_pending[g] is not None:
                    grp_pending[g].wait()
                    grp_pending[g] = None
        gathered[0] = 0
        step_no[0] = 0

    def fence():
        drain()
        hp.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # initialisation (untimed, not part of --warmup): the launch plan of the library follows the hard-row
    # counts of earlier launches (deferred batches need one launch of feedback), lanes and per-slot buffers
    # are created by the first pipelined launches
    INIT_STEPS = 16
    for _ in range(INIT_STEPS):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    # HIP events around the dominant kernel only (level 2): timing events drain the queue they are
    # recorded on, and bracketing every side-stream kernel costs ~10 us per pipelined step
    PROF_EVERY = 25  # bracket one launch in 25: a timing event costs a queue drain (~5 us; 1 in 8 cost 6 % of the rate)
    hp.profile_set_sampling(PROF_EVERY)
    hp.profile_enable(0 if os.environ.get("TSIM_BENCH_NO_PROFILE") == "1" else 2)
    hp.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue_s = time.perf_counter() - t0  # host time to enqueue all steps (before draining)
    drain()
    hp.synchronize()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stages = hp.profile_read_stages()
    kern_ms, launches = hp.profile_read(reset=True)
    hp.profile_enable(False)

    # untimed: a few more steps with every kernel bracketed, for the per-kernel split and the
    # first-kernel-start -> last-kernel-end latency of one launch
    detail = None
    hp.profile_set_sampling(1)
    if info.get("pattern_tables"):
        hp.profile_enable(1)
        for _ in range(8):
            step()
        drain()
        hp.synchronize()
        dst = hp.profile_read_stages()
        dms, dl = hp.profile_read(reset=True)
        hp.profile_enable(False)
        detail = {"stage_avg_ms": {k: v / max(dl, 1) for k, v in dst.items()}, "launch_latency_ms": dms / max(dl, 1),
                  "launches": dl, "note": "separate untimed steps with every kernel bracketed by HIP events"}

    # untimed: the dominant kernel with the GPU to itself (serial launches on the handle's stream): its own
    # duration, the figure rocprofv3's serial kernel trace gives (profiles/r01/*_kernel_stats_serial.csv).
    # In the timed region two or three launches are in flight and each first pass shares the CUs with the
    # others, so its bracketed duration there is longer than its cost.
    serial_ms = None
    if info.get("pattern_tables") and not use_dist:
        for _ in range(2):
            hp.sample_batch_device(d_f.ptr, B, num_f, key, d_out.ptr, shot_offset=shot_offset)
        hp.synchronize()
        hp.profile_enable(1)
        hp.profile_read(reset=True)
        for _ in range(8):
            hp.sample_batch_device(d_f.ptr, B, num_f, key, d_out.ptr, shot_offset=shot_offset)
        hp.synchronize()
        sst = hp.profile_read_stages()
        _, sl_n = hp.profile_read(reset=True)
        hp.profile_enable(False)
        if sl_n and sst["pattern_pass"] >= sst["full_kernel"]:
            serial_ms = sst["pattern_pass"] / sl_n

    # the full kernel alone (pattern tables off), a few steps: the rate on inputs where no shot is
    # tabulated, and the quantity earlier rounds reported
    full_only = None
    if N == 1 and info.get("pattern_tables") and not use_dist and not args.no_full_leg:
        hp_full = backend.HipProgram(program, device=local_rank, pattern_tables=False)
        for _ in range(2):
            hp_full.sample_batch_device(d_f.ptr, B, num_f, key, d_out.ptr, shot_offset=shot_offset)
        hp_full.synchronize()
        hp_full.profile_enable(True)
        hp_full.profile_read(reset=True)
        for _ in range(5):
            hp_full.sample_batch_device(d_f.ptr, B, num_f, key, d_out.ptr, shot_offset=shot_offset)
        fms, fl = hp_full.profile_read(reset=True)
        hp_full.profile_enable(False)
        full_only = {"kernel": "tsimk::k_sample4" if hp_full.info().get("chunk_table_kernel") else "tsimk::k_sample",
                     "kernel_avg_ms": fms / max(fl, 1), "shots_per_s": B / (fms / max(fl, 1) * 1e-3), "launches": fl}

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_shots = float(B) * N * args.steps
        value = total_shots / elapsed
        launch_s = (kern_ms / max(launches, 1)) * 1e-3  # first kernel start -> last kernel end of one launch
        # dominant kernel: with pattern tables the first pass (k_sample_lw) touches every row and moves
        # all the algorithmic bytes; the hard-row kernels of a launch overlap the first pass of the
        # following launches (pipelined slots).  Without tables: the one full kernel.
        tables_dominant = bool(info.get("pattern_tables")) and stages["pattern_pass"] >= stages["full_kernel"]
        if tables_dominant:
            avg_kernel_s = stages["pattern_pass"] / max(launches, 1) * 1e-3
        else:  # no tables, or dense error patterns: the launch plan went back to the full kernel
            avg_kernel_s = launch_s
        if launches == 0:  # profiling events disabled (experiments): fall back to the step time
            avg_kernel_s = launch_s = elapsed / args.steps
        bytes_per_shot = algorithmic_bytes_per_shot(num_f, n_out)
        ops_per_shot = algorithmic_ops_per_shot(program)
        achieved_gbs = bytes_per_shot * B / avg_kernel_s / 1e9
        achieved_tops = ops_per_shot * B / avg_kernel_s / 1e12
        if tables_dominant:
            kernel_name = "tsimk::k_sample_lw (pattern-table pass; hard rows: k_sample4h_multi batches / k_sample4h + k_sample4, on the third lane)" \
                if info.get("chunk_table_kernel") else "tsimk::k_sample_lw (pattern-table pass; hard rows: k_sample, overlapped on the other lanes)"
        else:
            kernel_name = "tsimk::k_sample4 (LDS chunk tables)" if info.get("chunk_table_kernel") else "tsimk::k_sample"
        res = {
            "metric": "detector shots/sec, 35-qubit distillation circuit, 1/2/4/8 MI355X",
            "value": value,
            "unit": "shots/s",
            "n_gpus": N,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {cfg.get('name', 'shape of SURVEY 8d')}, synthetic seeded program: "
                f"{info['total_graphs']} stabiliser terms, {info['total_rows']} GF(2) rows, n_out={n_out}, "
                f"num_f={num_f}, p_bit={cfg['p_bit']}",
                "shots_per_step_per_gpu": B,
                "global_batch": B * N,
                "sharding": (f"shots x{N}, bit-packed rows ({(n_out + 7) // 8} B/shot) collected every {GATHER_EVERY} batches by "
                             + ("one RCCL all-to-all (batch j of all ranks assembled on rank j mod N)" if GATHER_MODE == "alltoall"
                                else "an RCCL gather to rank 0")) if use_dist else "single GPU",
                "f_resident_in_hbm": True,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic_bytes(args.config, B),
                "traffic_unit": "bytes per launch, rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)",
                "achieved_at_step_rate": bytes_per_shot * B / (elapsed / args.steps) / 1e9,
                "frac_at_step_rate": bytes_per_shot * B / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "kernel": kernel_name,
                "kernel_avg_ms": avg_kernel_s * 1e3,
                "launches": launches,
                "hip_event_sampling": f"1 launch in {PROF_EVERY} bracketed",
                "kernel_serial_avg_ms": serial_ms,
                "achieved_serial": (bytes_per_shot * B / (serial_ms * 1e-3) / 1e9) if serial_ms else None,
                "all_kernels": detail,
                "pipeline_slots": NSLOT,
                "init_steps_untimed": INIT_STEPS,
                "host_enqueue_ms_per_step": host_enqueue_s / args.steps * 1e3,
                "algorithmic_bytes_per_shot": bytes_per_shot,
                "note": "achieved = algorithmic bytes of one launch / HIP-event time of the dominant kernel inside the "
                "timed region; there two first passes are in flight (two lanes) plus the hard-row batch on a third, so "
                "each first pass shares the CUs and its own duration is about twice the step time - "
                "achieved_at_step_rate uses the step time instead (= shots/s x bytes/shot, the accounting BASELINE.md "
                "section 4 fixes), achieved_serial the kernel's duration with the GPU "
                "to itself (kernel_serial_avg_ms, measured after the timed region; = rocprofv3's serial kernel "
                "trace). (k_sample_lw when pattern tables are active: it reads every f row and writes every "
                "tabulated row.) Integer-VALU / latency bound, not HBM bound (DESIGN.md section 3.5); see `valu` "
                "and `stage_avg_ms`",
            },
            "valu": valu_block(ops_per_shot, B, avg_kernel_s, args.config),
        }
        if full_only is not None:
            res["full_kernel_only"] = full_only
        if N == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(program, cfg, args.cpu_seconds)
    if use_dist:
        dist.destroy_process_group()
    # RCCL prints its banner through C stdio: drain that buffer first so that the JSON line is
    # the LAST line on stdout.
    import ctypes

    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
