#!/usr/bin/env python
"""bench.py - detector shots/sec of the fused sampling path on N MI355X.

Contract (see the round prompt): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  For N > 1 it is launched by ``torch.distributed.run`` with one rank per GPU (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment) - used as a process launcher only: the communicator, the
collectives, the barrier and the max over ranks are ``libtsim_hip.so``'s own RCCL calls
(``tsim_amd.dist.Communicator``); torch is imported for ``torch.cuda.synchronize()`` (the contract's bracket)
and, under the elastic agent, for its TCP store as the channel that carries the 128-byte ncclUniqueId.

Workload (``config.workload``): BASELINE.json ``configs[1]`` - the 35-qubit magic-state distillation shape
(15 direct detectors + one 5-output component, sum G = 148 stabiliser terms,
num_f = 64, per-bit fire probability 0.02, seed 42) as a seeded synthetic *normalised probability model*
(``tsim_amd.synth.physical_program``: every Bernoulli threshold lies in [0, 1]) because the reference's compile
pipeline cannot run here.  A *step* is one pass of the hot path (``sample_program``) over one batch of
``--shots`` shots PER GPU (weak scaling: rank r owns in-batch rows [r*shots, (r+1)*shots) of a global batch of
N*shots; the Threefry counter is the global row index, so the sharded result equals the unsharded one).  Packed
``f`` batches are resident in HBM before the timed region starts - ``NF`` distinct ones, used in rotation - and
for N > 1 the bit-packed detector/observable rows are collected by RCCL every ``GATHER_EVERY`` steps.

The ``--steps`` loop is repeated ``--repeats`` times, each repetition bracketed by barrier + synchronize on
both sides; ``value`` / ``ms_per_step`` are the MEDIAN repetition (max over ranks), ``repeat_ms_per_step`` lists
all of them.  Untimed extra legs (N = 1) put the headline in context: the full kernel alone, dense error
patterns, end-to-end ``CompiledDetectorSampler.sample()`` (host buffers in and out) and time-to-first-batch.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
# Vector-ALU issue, whole chip (1024 SIMDs).  The guide's figure: a wave64 instruction issues over 2 cycles at 2.4 GHz.
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2.0  # 1.23e12 wave64 instructions/s
# Measured on this chip per instruction class (scripts/microbench/valu_table.hip -> profiles/r03/valu_table.txt):
# full-rate instructions 2.4 cycles at the 1.9-2.1 GHz the chip sustains under them (8.2-8.5e11/s), half-rate ones -
# v_alignbit, v_lshlrev, v_add3, v_cndmask, v_cmp, v_bcnt, every instruction with an SGPR operand - 4.3 cycles at
# 2.3-2.4 GHz (5.5-5.8e11/s); one Threefry-2x32-20 block = 75 such instructions, best of ten instruction selections:
# 5.8e11 blocks/s (scripts/microbench/threefry_block.hip -> profiles/r03/threefry_block.txt).
THREEFRY_BLOCKS_PER_S = 5.8e11
NF_HOST = 4            # f batches drawn by the seeded host generator (tsim_amd.synth.synth_f): what the verification legs regenerate
NF_DEFAULT = 64        # distinct f batches rotated through the timed loop: 64 x 8 MB (C2) = 512 MB, beyond the 256-MiB Infinity Cache
INFINITY_CACHE_MB = 256


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


def load_pmc(config: str, shots: int):
    """The committed rocprofv3 PMC summary of this workload (profiles/latest_pmc.json), or None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        if d.get("_config") == config and d.get("_shots") == shots:
            return d
    except Exception:
        pass
    return None


def load_isa_classes():
    """The committed instruction-class census of the first pass's row loop (scripts/isa_classes.py), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r03", "first_pass_isa.json")))
    except Exception:
        return None


def valu_block(ref_ops_per_shot: int, shots: int, n_draws: int, kernel_s_per_batch: float, serial_s, step_s: float, pmc, fast_path: bool) -> dict:
    """The bound that actually binds the first pass: vector-ALU issue.

    Three yardsticks, from loose to tight: (1) the guide's issue peak - one wave64 instruction per 2 cycles and SIMD,
    1.23e12/s; (2) the op-class-weighted bound - the kernel's own loop body (profiles/r03/first_pass_isa.json) priced
    with the issue cost MEASURED per instruction class; (3) the Threefry floor - the draws the reference's RNG stream
    prescribes (one Threefry-2x32-20 block per shot and sampled output, sampler.py:74-75) at the best block rate any
    instruction selection reached on this chip."""
    isa = load_isa_classes() if fast_path else None
    out = {
        "bound": "valu_issue",
        "issue_peak_wave_insts_per_s": VALU_ISSUE_PEAK,
        "issue_peak_source": "MI355X_MICROARCH.md: wave64 VALU instruction = 2 cycles per SIMD, 1024 SIMDs x 2.4 GHz",
        "reference_algorithm_ops_per_shot": ref_ops_per_shot,
        "threefry_floor_us_per_batch": n_draws * shots / THREEFRY_BLOCKS_PER_S * 1e6,
        "threefry_floor_source": "scripts/microbench/threefry_block.hip: 5.8e11 blocks/s, the best of ten instruction "
                                 "selections (profiles/r03/threefry_block.txt); one block per shot and compiled output",
        "frac_of_threefry_floor_in_pipeline": n_draws * shots / THREEFRY_BLOCKS_PER_S / kernel_s_per_batch,
        "frac_of_threefry_floor_at_step_rate": n_draws * shots / THREEFRY_BLOCKS_PER_S / step_s,
    }
    if isa:
        wi = isa["wave_instructions_per_1e6_shots"]
        insts = wi["valu"] * shots / 1e6
        bound_s = isa["us_per_1e6_shots_at_2.3GHz"]["valu"] * 1e-6 * shots / 1e6
        out.update({
            "loop_body_valu_wave_insts_per_launch_of_one_batch": insts,
            "loop_body_classes": isa["classes"],
            "frac_of_issue_peak_at_step_rate": insts / step_s / VALU_ISSUE_PEAK,
            "frac_of_issue_peak_in_pipeline": insts / kernel_s_per_batch / VALU_ISSUE_PEAK,
            "op_class_bound_us_per_batch": bound_s * 1e6,
            "frac_of_op_class_bound_at_step_rate": bound_s / step_s,
            "frac_of_op_class_bound_in_pipeline": bound_s / kernel_s_per_batch,
            "frac_of_op_class_bound_serial": (bound_s / serial_s) if serial_s else None,
            "op_class_source": "profiles/r03/first_pass_isa.json: static census of k_sample_lw_fast's row loop x measured cycles "
                               "per class (full 2.4, half 4.3; profiles/r03/valu_table.txt), 15.26 wave-rows per SIMD, 2.3 GHz",
        })
    if pmc and "SQ_INSTS_VALU" in pmc:
        insts = float(pmc["SQ_INSTS_VALU"])
        out.update({
            "executed_valu_wave_insts_per_batch": insts,
            "executed_frac_of_issue_peak_at_step_rate": insts / step_s / VALU_ISSUE_PEAK,
            "pmc_source": "profiles/latest_pmc.json (rocprofv3 --pmc SQ_INSTS_VALU, same workload)",
        })
    out["note"] = ("frac_of_issue_peak = vector instructions / time / 1.23e12: what the guide's peak would allow; the op-class "
                   "bound prices the SAME instructions with the rates this chip sustains (half of them are half-rate: rotates, "
                   "selects, SGPR operands); the Threefry floor is what the prescribed draws alone cost.  *_in_pipeline uses the "
                   "first pass's own HIP-event duration per batch inside the timed region, *_at_step_rate the step time.")
    return out


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
    from oracle import oracle_c, oracle_np
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
    # also the single-threaded rate (the reference's Python driver is single-threaded) and the numpy
    # restatement that keeps the reference's data movement (float32 GEMM % 2, materialised lookups, scans)
    f1 = synth.synth_f(20_000, cfg["num_f"], cfg["p_bit"], seed=cfg["seed"])
    t1 = time.perf_counter()
    op.sample_program(f1, (1, 2), threads=1)
    single = len(f1) / (time.perf_counter() - t1)
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
        "note": "the repo's own C restatement of the reference algorithm (packed popcount rows, OpenMP over shots) - NOT the "
                "reference's JAX-on-CPU path, which cannot run in this image (no jax/equinox); numpy_reference_faithful_value keeps "
                "the reference's data movement (float32 GEMM % 2, materialised lookups, scans) and is 250x slower than this port; "
                "the reference's only published CPU figure for this circuit is 2.8e5 shots/s on unstated hardware (BASELINE.md): "
                "a '>= 10x the reference CPU path' claim can only be made against that number",
        "sample": f"{n} shots of the same program and f distribution, C oracle (oracle/oracle.c, OpenMP over shots, "
        f"{threads} threads on {cores} CPUs; {note}), {dt:.1f} s wall",
    }


def exchange_unique_id(rank: int, world: int) -> bytes:
    """The ncclUniqueId from rank 0 to everyone: through the launcher's own TCP store when this process runs
    under torch's elastic agent, else over tsim_amd.dist's socket rendezvous."""
    from tsim_amd import dist as tdist

    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29511"))
    if world > 1 and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true":
        try:
            from datetime import timedelta

            from torch.distributed import TCPStore

            store = TCPStore(addr, port, world, is_master=False, timeout=timedelta(seconds=120))
            key = "tsim_amd/rccl_unique_id/" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
            if rank == 0:
                ident = tdist.unique_id()
                store.set(key, ident)
                return ident
            return bytes(store.get(key))
        except Exception as exc:  # fall through to the socket rendezvous
            print(f"[bench] agent store unavailable ({exc!r}); using the socket rendezvous", file=sys.stderr)
    return tdist.rendezvous_tcp(rank, world, addr=addr, port=int(os.environ.get("TSIM_DIST_PORT", port + 1)))


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: spawn the N ranks here - one process per GPU, rank r on device r,
    rendezvous over tsim_amd.dist's own socket protocol (no torch.distributed) - and relay rank 0's line."""
    import socket
    import subprocess

    def free_port() -> int:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    port, dport = free_port(), free_port()
    launch_only = os.environ.get("TSIM_BENCH_LAUNCH_ONLY") == "1"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TSIM_DIST_PORT=str(dport))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if (r == 0 or launch_only) else subprocess.DEVNULL))
    outs = [p.communicate()[0] if p.stdout else (p.wait() and b"") for p in procs]
    rc = max(abs(p.returncode or 0) for p in procs)
    for o in outs:
        if o:
            sys.stdout.write(o.decode("utf-8", "replace"))
    sys.stdout.flush()
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=8, help="repetitions of the timed --steps loop (median reported)")
    ap.add_argument("--spinup-ms", type=float, default=0.0,
                    help="untimed steps worth this many ms before every timed repetition (clock spin-up; default 0 = none: `value` is what "
                         "--warmup steps and then --steps timed steps give; the rate after 30 ms of spin-up is reported beside it)")
    ap.add_argument("--nf", type=int, default=NF_DEFAULT, help="distinct resident f batches the timed loop rotates through (the first "
                    f"{NF_HOST} from the seeded host generator, the others drawn on the device with the same per-bit probability)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the driver-timed legs of the other BASELINE configs (C3, C4, C5)")
    ap.add_argument("--shots", type=int, default=1_000_000, help="shots per step per GPU (--scaling weak) / per step of the whole job (--scaling strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: every GPU samples --shots rows per step (rank r owns rows "
                    "[r * shots, (r + 1) * shots) of a global batch of N * shots); strong: ONE global batch of --shots rows per step, rank r samples rows "
                    "[r * B / N, (r + 1) * B / N) (SURVEY 8(e)), B rounded up to a multiple of N")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--random-program", action="store_true", help="the unconstrained random program instead of the normalised one")
    ap.add_argument("--p-bit", type=float, default=None, help="override the per-bit fire probability of the synthetic f batches")
    ap.add_argument("--approx", action="store_true", help="has_approximate_floatfactors=True in every level (compile/evaluate.py:56-59: "
                    "the float32 sum branch the real distillation circuits take, SURVEY section 7 hard part 2) as the headline workload")
    ap.add_argument("--live-padding", action="store_true", help="term counts of the shape reached with terms the packer cannot cancel")
    ap.add_argument("--program", default=None, help="a compiled program exported where tsim is installed "
                    "(scripts/export_from_tsim.py -> .npz) instead of the synthetic shape")
    ap.add_argument("--check-golden", action="store_true", help="with --program: first sample the file's golden request (the shots the reference "
                    "itself drew, scripts/export_from_tsim.py --golden) and compare bit for bit: `golden_check` in the JSON line")
    ap.add_argument("--num-f", type=int, default=None, help="with --program: width of the synthetic f batches (default: from the file, "
                    "else the largest f index + 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the untimed context legs (profiling runs)")
    ap.add_argument("--no-full-leg", action="store_true", help=argparse.SUPPRESS)  # older name
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    if args.no_full_leg:
        args.no_extra_legs = True

    # the host driver supports dmabuf IPC only: RCCL's peer mappings fail without this (exported on the GPU boxes;
    # set here too in case a launcher scrubs the environment) - before any HIP runtime is loaded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: spawn the N ranks ourselves (one process per GPU, tsim_amd.dist's socket rendezvous, no torch)
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # never print a line whose n_gpus is not what was asked for
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    N = max(world, 1)
    if os.environ.get("TSIM_BENCH_LAUNCH_ONLY") == "1":
        # launcher check (tests/test_dist_gloo.py, no GPU needed): every rank reaches the point where the communicator
        # would be created, with the same 128 bytes in hand
        import hashlib
        from tsim_amd import dist as tdist0

        ident = tdist0.rendezvous_tcp(rank, N, addr=os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                      port=int(os.environ.get("TSIM_DIST_PORT", int(os.environ.get("MASTER_PORT", "29511")) + 1)),
                                      timeout=60, make_id=lambda: bytes((7 * i + 3) % 256 for i in range(128)))
        print(json.dumps({"launch_only": True, "rank": rank, "world": N, "local_rank": local_rank, "id_sha": hashlib.sha256(ident).hexdigest()[:16]}), flush=True)
        return
    use_dist = N > 1 or os.environ.get("TSIM_BENCH_FORCE_DIST") == "1"

    # torch: only for the contract's torch.cuda.synchronize() (and its HIP runtime must be the first one loaded
    # when it is present at all)
    try:
        import torch

        torch.cuda.set_device(local_rank)

        def device_sync():
            torch.cuda.synchronize()
        HAVE_TORCH_SYNC = True
    except Exception:
        torch = None
        HAVE_TORCH_SYNC = False

        def device_sync():
            pass

    from tsim_amd import _lib, backend, prng, synth
    from tsim_amd import dist as tdist

    lib = _lib.load()
    comm = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        comm = tdist.Communicator(local_rank, exchange_unique_id(rank, N), rank, N)

    golden_check = None
    if args.program and args.check_golden and rank == 0:
        from tsim_amd import golden as tgolden

        try:
            golden_check = tgolden.check_golden(args.program, device=local_rank)
        except KeyError as exc:
            golden_check = {"ok": None, "note": str(exc)}
    if args.program:
        from tsim_amd import program as tprog

        program, extra_arrays = tprog.load_npz(args.program)
        tprog.validate_program(program)
        need_f = 1 + max([int(np.max(program.direct_f_indices, initial=-1))] + [int(np.max(c.f_selection, initial=-1)) for c in program.components])
        nf_file = int(extra_arrays["num_f"]) if "num_f" in extra_arrays else need_f
        cfg = dict(name=f"exported program {os.path.basename(args.program)}", num_f=int(args.num_f or max(nf_file, need_f)), p_bit=0.02, seed=42)
        args.config = os.path.basename(args.program)
        # the file's own noise model (scripts/export_from_tsim.py stores channel_probs / error_transform as
        # ChannelSampler receives them, noise/channels.py:531): f batches are then drawn from IT, not from p_bit
        file_noise = None
        if "error_transform" in extra_arrays and "n_channels" in extra_arrays and args.p_bit is None:
            from tsim_amd.channels import ChannelSampler

            probs = [np.asarray(extra_arrays[f"channel_probs_{i}"], dtype=np.float64) for i in range(int(extra_arrays["n_channels"]))]
            file_noise = (probs, np.asarray(extra_arrays["error_transform"], dtype=np.uint8))
            cfg["num_f"] = int(args.num_f or file_noise[1].shape[0])
            cfg["name"] += " with its own noise model"
    else:
        program, cfg = synth.config_program(args.config, physical=not args.random_program, approx=args.approx,
                                            **({"live_padding": True} if args.live_padding else {}))
    if args.p_bit is not None:
        cfg = dict(cfg, p_bit=float(args.p_bit))
    t_build0 = time.perf_counter()
    hp = backend.HipProgram(program, device=local_rank)
    t_build = time.perf_counter() - t_build0
    info = hp.info()
    num_f, n_out = cfg["num_f"], program.num_outputs
    B = int(args.shots) if args.scaling == "weak" else -(-int(args.shots) // N)  # rows THIS rank samples per step
    WF, WO, RB = max(1, (num_f + 63) // 64), (n_out + 63) // 64, (n_out + 7) // 8

    def resident_f(p_bit: float, seed: int):
        """One synthetic packed f batch of this rank's shard, in HBM."""
        if args.program and file_noise is not None and p_bit == cfg["p_bit"]:
            packed = ChannelSampler(file_noise[0], file_noise[1], seed=seed).sample_packed(B).view(np.uint8).reshape(B, -1)
            buf = hp.malloc(B * WF * 8)
            hp.h2d(buf, np.ascontiguousarray(packed))
            return buf
        f = synth.synth_f(B, num_f, p_bit, seed=seed)
        packed = np.packbits(f, axis=1, bitorder="little")
        if WF * 8 - packed.shape[1]:
            packed = np.pad(packed, ((0, 0), (0, WF * 8 - packed.shape[1])))
        buf = hp.malloc(B * WF * 8)
        hp.h2d(buf, packed)
        return buf

    NF = max(1, int(args.nf))
    f_bufs = [resident_f(cfg["p_bit"], cfg["seed"] + 1000 * rank + 7919 * k) for k in range(min(NF, NF_HOST))]
    f_host = list(f_bufs)  # (the seeded ones: what the verification legs regenerate on the host)
    if NF > len(f_bufs):
        # the rest are drawn ON THE DEVICE (k_noise_tile: one one-bit channel per f bit at the same probability - the same
        # distribution as synth_f, 25 us instead of seconds per batch - or the file's own noise model), so that the timed loop's
        # working set is what `--nf` says
        f_bufs += device_f_batches(backend, hp, num_f, cfg["p_bit"], B, WF, NF - len(f_bufs), seed=cfg["seed"] + 17 * rank,
                                   noise_model=file_noise if (args.program and file_noise is not None and args.p_bit is None) else None)
    NF = len(f_bufs)

    # Pipeline of NSLOT slots (tsim_sample_batch_device_begin/_end): with short hard-row lists the library runs
    # the first passes of consecutive launches on two lanes and the hard rows of four launches at a time as ONE
    # grid on a third; a slot is reused only after its batch is done, so the number of
    # slots - not of lanes - covers the batch latency.  N > 1: the kernels also write the reference's bit_packed
    # rows (ceil(n_out/8) bytes per shot) into a group buffer; every GATHER_EVERY steps ONE asynchronous RCCL
    # collective moves the group (double-buffered, queued on the lane where results complete).
    # Slots: the hard rows of 4 launches form a batch (8 when the chunk tables exceed 4 MB - C4), and two full batches
    # must fit the rotation: 16 slots for the 8-launch programs (C4: 33 us per step, 41 with 14 slots), 14 otherwise -
    # every even count from 10 to 16 gives C2 the same steady state since the library pre-waits mid-batch
    # (profiles/r02/slot_count.txt), 14 is marginally the best at the driver's 20 steps per timed region.
    default_slots = 16 if info["table_bytes"] > (4 << 20) else 14
    NSLOT = max(1, min(backend.HipProgram.PIPELINE_SLOTS, default_slots))
    # Group size: a collective per group runs on the join lane under the next group's kernels, only the LAST group's
    # is exposed at the end of a timed region - so a short region (the driver's --steps 20) wants small groups, a
    # long one fewer, larger collectives.  A group must be a multiple of the hard-row batch (4 launches: closing a
    # group flushes the waiting batch, and a partial batch is a whole extra hard-row pass - measured on the forced
    # one-GPU path at 20 steps: groups of 5: 698 us per region, of 8: 528, of 20: 519, of 4: 607) and of N (equal
    # chunks per peer): about a third of the region in units of max(4, N), at least 8, at most 64 batches.
    GATHER_EVERY = tdist.gather_group_size(args.steps, N, False, int(os.environ["TSIM_BENCH_GATHER_EVERY"]) if os.environ.get("TSIM_BENCH_GATHER_EVERY") else None)
    # How the finished rows are collected (N > 1).  "root0": the north star's gather of the detector bit strings to
    # rank 0 (ncclGather).  "alltoall": the same gather with its roots spread over the node - group j of every rank
    # lands on rank j (ncclAllToAll).  Arithmetic behind the default: a rank produces
    # ~180 GB/s of bit-packed rows; xGMI is a full mesh of point-to-point links of ~50-77 GB/s per direction, so
    # rank 0 can take in ~0.4 TB/s over its 7 links while a gather from 7 peers needs 1.2 TB/s - the single root
    # would bound the node at ~3x one GPU.  Spread roots put 1/N of a rank's stream on each link: 91 GB/s at N = 2
    # (still above one link: two GPUs are link-bound either way, the single root by 2.4x, spread roots by ~1.2x),
    # 46 GB/s at N = 4, 23 GB/s at N = 8.  Hence "auto" = alltoall for every N > 1; no N > 1 measurement exists
    # yet to confirm the arithmetic.
    # Default: "root0" - the north star's collective; "alltoall" is the hypothesis above, to be chosen by the first N > 1
    # measurement (TSIM_BENCH_GATHER=alltoall), not by arithmetic.
    # N > 1 and no TSIM_BENCH_GATHER: "measured" - both collectives are timed on this node before anything else (three of
    # each on a group buffer, max over ranks) and the faster one collects the rows; the timings go into the JSON line
    # (`gather_calibration`).  That is the N > 1 measurement the choice was waiting for, made where the run happens.
    GATHER_MODE = os.environ.get("TSIM_BENCH_GATHER", "measured" if (use_dist and N > 1) else "root0")
    if GATHER_MODE == "auto":
        GATHER_MODE = "measured" if (use_dist and N > 1) else "root0"
    gather_calibration = None
    if GATHER_MODE in ("alltoall", "measured"):
        GATHER_EVERY = (GATHER_EVERY + N - 1) // N * N
    # results are written in the reference's bit_packed layout (sampler.py:665-669: ceil(n_out/8) bytes per shot,
    # TSIM_PIPE_OUT_BIT_PACKED) - the algorithmic output bytes, not the padded 8-byte device word
    # N = 1: the K steps of a timed region are ONE call of tsim_sample_steps_device (key splits included): the library
    # fuses the first passes of up to 8 batches into one grid and runs each group's hard rows as one batch behind it
    # (DESIGN.md section 3.10); its slot rotation covers all 16 pipeline slots, so 16 output buffers.
    # TSIM_BENCH_PER_STEP=1: one tsim_sample_batch_device_begin_split per step, as in rounds 1-2.
    PER_STEP = os.environ.get("TSIM_BENCH_PER_STEP") == "1"
    if not PER_STEP:
        NSLOT = backend.HipProgram.PIPELINE_SLOTS
    d_outs = [hp.malloc(B * WO * 8) for _ in range(NSLOT)]  # sized for the padded rows too (serial legs below)
    out_ptrs = [d.ptr for d in d_outs]
    PIPE_READY, PIPE_PACKED = 1, 2
    COLL_OWN_STREAM = True  # (queued on the join lane itself the collective held back the next hard-row batch)
    if use_dist:
        join_ptr = hp.pipeline_lane_stream(2)   # where deferred hard-row batches - i.e. results - complete
        main_ptr = hp.stream_ptr()               # first-pass lane 0
        lane_ptrs = [main_ptr, hp.pipeline_lane_stream(1), join_ptr]  # the lanes fused groups run on (two, three for small groups)
        grp = [hp.malloc(GATHER_EVERY * B * RB) for _ in range(2)]
        if GATHER_MODE == "measured":
            recv_all = hp.malloc((N if rank == 0 else 1) * GATHER_EVERY * B * RB)
            times = {}
            for mode in ("root0", "alltoall"):
                comm.barrier()
                lib.tsim_device_synchronize(local_rank)
                t0 = time.perf_counter()
                for _ in range(3):
                    if mode == "alltoall":
                        comm.alltoall_rows(grp[0].ptr, recv_all.ptr, GATHER_EVERY // N * B * RB, stream=0)
                    else:
                        comm.gather_rows(grp[0].ptr, GATHER_EVERY * B * RB, recv_all.ptr if rank == 0 else 0, root=0, stream=0)
                lib.tsim_device_synchronize(local_rank)
                times[mode] = comm.allreduce_max((time.perf_counter() - t0) / 3)
            recv_all.free()
            GATHER_MODE = "alltoall" if times["alltoall"] < 0.9 * times["root0"] else "root0"  # (the north star's gather unless clearly slower)
            gather_calibration = {"batches_per_collective": GATHER_EVERY, "bytes_per_rank": GATHER_EVERY * B * RB,
                                  "root0_ms": times["root0"] * 1e3, "alltoall_ms": times["alltoall"] * 1e3, "chosen": GATHER_MODE,
                                  "rule": "alltoall (roots spread over the ranks) when it takes less than 0.9 x the gather to rank 0, measured here"}
        if GATHER_MODE == "alltoall":  # (rank 0 also takes the gather of a partial last group: N senders)
            grp_recv = [hp.malloc((N if rank == 0 else 1) * GATHER_EVERY * B * RB) for _ in range(2)]
        else:
            grp_recv = [hp.malloc(N * GATHER_EVERY * B * RB) if rank == 0 else None for _ in range(2)]
        grp_used = [False, False]

    key = prng.key(cfg["seed"])
    key_state = (C.c_uint32 * 2)(key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF)  # split in place by the library
    shot_offset = rank * B
    step_no = [0]
    launched = [0]  # launches of the whole run (step_no restarts with every timed region in the N > 1 path)
    gathered = [0]  # groups whose collective has been issued
    begin_split = lib.tsim_sample_batch_device_begin_split
    end_fn = lib.tsim_sample_batch_device_end
    join_fn = lib.tsim_pipeline_join
    wait_fn = lib.tsim_pipeline_wait_stream
    h_prog = hp._h

    def gather_next(count: int) -> None:
        """Issue the collective of the oldest un-collected group (all of its steps are joined on the join lane)."""
        g = gathered[0] & 1
        # The collective runs on the communicator's OWN stream, ordered behind the join lane by one event: queued on
        # the join lane itself (= the hard-row batch lane) it held back the next batch's hard-row pass (kernel trace
        # of the forced one-GPU path: the last batch of a region started 38 us after its first passes instead of 12).
        coll = 0 if COLL_OWN_STREAM else join_ptr
        if COLL_OWN_STREAM:
            comm.stream_wait(0, join_ptr)
        kind = tdist.collective_kind(GATHER_MODE, count, N)
        if kind == "alltoall":  # equal chunks: batches [j * count/N, (j+1) * count/N) land on rank j
            comm.alltoall_rows(grp[g].ptr, grp_recv[g].ptr, count // N * B * RB, stream=coll)
        else:  # the gather to rank 0, exactly `count` batches - also what a partial last group of the all-to-all mode takes
            comm.gather_rows(grp[g].ptr, count * B * RB, grp_recv[g].ptr if grp_recv[g] else 0, root=0, stream=coll)
        last_collective[0] = (g, count, kind)
        comm.mark(g, coll)  # "the collective that read group buffer g is done"
        grp_used[g] = True
        gathered[0] += 1

    steps_fn = lib.tsim_sample_steps_device
    ptr_cache = {}

    last_collective = [None]

    def dist_steps(k: int, f_list) -> None:
        """N > 1: the same library call per piece of a gather group; every batch writes its bit_packed rows straight into
        its slice of the group buffer, a complete group is joined on the join lane and collected."""
        for g, pos, n in tdist.group_pieces(step_no[0], k, GATHER_EVERY):  # (the arithmetic tests/test_dist_gloo.py runs for N = 2, 4, 8)
            j = step_no[0]
            rc = 0
            if launched[0] < NSLOT:  # (the very first launches of the run create the lanes)
                rc = wait_fn(h_prog, None)
            if pos == 0 and grp_used[g]:
                # the collective that last read this group buffer (two groups ago) must be done before kernels overwrite
                # it: every first-pass lane waits for THAT marker - and for nothing else (ordering the lanes behind
                # lane 0's whole queue, as rounds 1-2 did, made consecutive groups run one after the other: 21.8 us per
                # step at --steps 20 on the forced one-GPU path where the plain path takes 14.6)
                for lp in lane_ptrs:
                    comm.wait_mark(g, lp)
            if rc >= 0:
                fa = (C.c_void_p * n)(*[f_list[(j + i) % len(f_list)].ptr for i in range(n)])
                oa = (C.c_void_p * n)(*[grp[g].ptr + (pos + i) * B * RB for i in range(n)])
                rc = steps_fn(h_prog, n, fa, B, num_f, key_state, shot_offset, oa, None,
                              (0 if launched[0] < NSLOT else PIPE_READY) | PIPE_PACKED)
            if rc < 0:
                raise RuntimeError(f"pipelined launch failed ({rc}): {_lib.last_error()}")
            launched[0] += n
            step_no[0] = j + n
            if pos + n == GATHER_EVERY:  # group complete: join every slot on the join lane, then collect
                join_fn(h_prog, join_ptr)
                gather_next(GATHER_EVERY)

    def steps(k: int, f_list=f_bufs) -> None:
        """k consecutive steps in one library call (N = 1): per batch key, subkey = split(key) (sampler.py:399) and one
        sample_program; batch j reads f_list[j % len], writes bit_packed rows to out buffer j % NSLOT."""
        if PER_STEP:
            for _ in range(k):
                step(f_list)
            return
        if use_dist:
            dist_steps(k, f_list)
            return
        j0 = step_no[0]
        ck = (k, j0 % NSLOT, j0 % len(f_list), id(f_list))
        arrs = ptr_cache.get(ck)
        if arrs is None:
            arrs = ((C.c_void_p * k)(*[f_list[(j0 + i) % len(f_list)].ptr for i in range(k)]),
                    (C.c_void_p * k)(*[out_ptrs[(j0 + i) % NSLOT] for i in range(k)]))
            ptr_cache[ck] = arrs
        step_no[0] = j0 + k
        rc = steps_fn(h_prog, k, arrs[0], B, num_f, key_state, shot_offset, arrs[1], None, PIPE_READY | PIPE_PACKED)
        if rc < 0:
            raise RuntimeError(f"tsim_sample_steps_device failed ({rc}): {_lib.last_error()}")

    def step(f_list=f_bufs) -> None:
        # one host split per batch, key, subkey = split(key) (sampler.py:399), inside the launch call
        j = step_no[0]
        b = j % NSLOT
        step_no[0] = j + 1
        d_f = f_list[j % len(f_list)].ptr
        if not use_dist:
            # _begin on a slot whose previous step was not joined is ordered after that step's second pass by the
            # library (include/tsim_hip.h), so one call per step is enough here
            rc = begin_split(h_prog, b, d_f, B, num_f, key_state, shot_offset, out_ptrs[b], None, None, PIPE_READY | PIPE_PACKED)
            if rc < 0:
                raise RuntimeError(f"tsim_sample_batch_device_begin failed ({rc}): {_lib.last_error()}")
            return
        g, pos = (j // GATHER_EVERY) & 1, j % GATHER_EVERY
        rc = 0
        if pos == 0:
            if grp_used[g]:
                # the collective that last read this group buffer (two groups ago) must be done before kernels
                # overwrite it: first-pass lane 0 waits for THAT marker only (waiting for the whole join lane would
                # drain the pipeline once per group); the other lanes are ordered after lane 0 just below
                comm.wait_mark(g, main_ptr)
            rc = wait_fn(h_prog, None)
        elif launched[0] < NSLOT:  # (the very first launches of the run create the lanes)
            rc = wait_fn(h_prog, None)
        if rc >= 0:  # every launch writes its bit_packed rows straight into its slice of the group buffer
            rc = begin_split(h_prog, b, d_f, B, num_f, key_state, shot_offset, grp[g].ptr + pos * B * RB, None, None,
                             (0 if launched[0] < NSLOT else PIPE_READY) | PIPE_PACKED)
        launched[0] += 1
        if rc < 0:
            raise RuntimeError(f"pipelined launch failed ({rc}): {_lib.last_error()}")
        if pos == GATHER_EVERY - 1:  # group complete: join every slot on the join lane, then collect
            join_fn(h_prog, join_ptr)
            gather_next(GATHER_EVERY)

    def drain() -> None:
        if not use_dist:
            # nothing to join per slot: every caller of drain() goes on to hp.synchronize(), which flushes the waiting
            # hard-row batch and waits for every lane (tsim_synchronize).  A _end per slot would queue one stream wait
            # per slot on the handle's stream - which is also first-pass lane 0: 14 barrier packets behind its last
            # kernel, ~60 us of a 20-step region .
            return
        join_fn(h_prog, join_ptr)
        n_steps = step_no[0]
        while gathered[0] * GATHER_EVERY < n_steps:
            gather_next(min(GATHER_EVERY, n_steps - gathered[0] * GATHER_EVERY))
        gathered[0] = 0
        step_no[0] = 0

    def fence() -> None:
        """drain + everything queued on this device finished + all ranks here (the contract's bracket)."""
        drain()
        hp.synchronize()
        if not HAVE_TORCH_SYNC:
            lib.tsim_device_synchronize(local_rank)
        device_sync()
        if comm is not None:
            comm.barrier()

    # initialisation (untimed, not part of --warmup): the launch plan follows the hard-row counts of earlier
    # launches (deferred batches need one launch of feedback), lanes and per-slot buffers are created by the
    # first pipelined launches
    INIT_STEPS = 16
    if PER_STEP:
        for _ in range(INIT_STEPS):
            step()
    else:
        for _ in range(INIT_STEPS // 4):  # small calls: the launch plan needs the feedback of earlier launches
            steps(4)
    fence()
    # A fresh handle samples with the shallow tables finalize built while its default depth is built in the background
    # (round 5; `time_to_n_shots` reports that start).  `value` is the rate of the handle's DEFAULT state: untimed steps until the
    # build is in place (C2: a few ms; the cultivation shape: ~40 ms - timed right away its 10^5-shot steps ran at 1.07e10
    # instead of 1.55e10), reported as `tables_settled_after_s`.
    t_settle0 = time.perf_counter()
    for _ in range(4000):  # (a count, not a clock: with several ranks every rank must leave the loop in the same round)
        pend = 1.0 if hp.info().get("pattern_build_pending") else 0.0
        if comm is not None:
            pend = comm.allreduce_max(pend)
        if pend == 0.0:
            break
        steps(4)
        fence()
    tables_settled_after_s = time.perf_counter() - t_settle0
    fence()
    steps(args.warmup)
    fence()
    verify = None
    if use_dist and os.environ.get("TSIM_BENCH_VERIFY") == "1":
        verify = verify_collected(backend, prng, synth, program, cfg, hp, comm, lib, rank, N, B, num_f, n_out, RB, local_rank,
                                  key_state, lambda k: steps(k, f_host), drain, fence, grp_recv, last_collective, GATHER_EVERY, len(f_host))
    # HIP events around the dominant kernel only (level 2), on at least 8 launches per repetition: timing events
    # drain the queue they are recorded on (~5 us each), so not every launch is bracketed
    # (at least 8 bracketed launches over the repetitions together: a bracket costs ~7 us of a ~17 us step)
    per_rep = max(1, -(-8 // max(1, args.repeats)))
    PROF_EVERY = max(1, min(25, args.steps // per_rep))
    if not PER_STEP:
        # a fused first pass serves up to 8 batches; a bracket (two timing events on the lane) costs ~7 us of a ~100-us
        # kernel and, worse, a bubble on the lane: bracketing EVERY launch cost 7 % of the step rate (12.7 -> 11.8 us at
        # --steps 200).  About 8 bracketed launches over all repetitions together.
        launches_per_rep = max(1, -(-args.steps // 8))
        PROF_EVERY = max(1, launches_per_rep * max(1, args.repeats) // 8)  # (about one bracketed launch per repetition at --steps 20)
    PROF_LEVEL = 0 if os.environ.get("TSIM_BENCH_NO_PROFILE") == "1" else 2
    hp.profile_set_sampling(PROF_EVERY)
    hp.profile_enable(PROF_LEVEL)
    hp.profile_read(reset=True)
    hp.profile_read_steps()
    rep_elapsed, rep_enqueue = [], []
    # Clock spin-up (untimed, disclosed in the JSON line as "spinup"): the chip's power management raises its clocks
    # over the first ~25 ms of sustained work - with no spin-up the repetitions of ONE process get faster one after
    # the other until ~25 ms of work have gone by (profiles/r03/steps_dependence.txt: 12.9, 11.9, 11.2, 11.2 ... us per
    # step over the repetitions of --steps 1000; the same trend, never finished, over the eight 2.5-ms repetitions of
    # --steps 200), which is a property of the chip, not of the engine.  --spinup-ms of the same steps run right before
    # every timed region (after --warmup, before the fence), so that short regions are measured at the clocks long ones
    # reach on their own.  --spinup-ms 0 switches it off.
    spin_steps = 0
    if args.spinup_ms > 0:
        hp.profile_enable(0)
        fence()
        t0 = time.perf_counter()
        steps(32)
        fence()
        est = (time.perf_counter() - t0) / 32
        spin_steps = int(min(20000, max(0, args.spinup_ms * 1e-3 / max(est, 1e-7))))
        hp.profile_enable(PROF_LEVEL)
    # ... and, for the record, the SAME timed region without it first (three repetitions right after --warmup, profiling
    # off): `spinup.without` in the JSON line - what a literal "W warm-up steps, then K timed steps" gives on this box
    cold_elapsed = []
    if spin_steps and not use_dist:
        hp.profile_enable(0)
        for _ in range(min(3, max(1, args.repeats))):
            fence()
            t0 = time.perf_counter()
            steps(args.steps)
            drain()
            hp.synchronize()
            lib.tsim_device_synchronize(local_rank)
            device_sync()
            cold_elapsed.append(time.perf_counter() - t0)
        hp.profile_enable(PROF_LEVEL)
    for _ in range(max(1, args.repeats)):
        if spin_steps:
            hp.profile_enable(0)  # (no timing events in the untimed part)
            # in calls of 64 steps, each waited for: a backlog of thousands of commands leaves the HIP runtime with
            # completed commands to retire during the NEXT enqueues - the timed ones (186 instead of 48 us for 20 steps)
            left = spin_steps
            while left > 0:
                steps(min(64, left))
                left -= 64
                if not use_dist:
                    hp.synchronize()
            hp.profile_enable(PROF_LEVEL)
        fence()
        region_start = (int(key_state[0]), int(key_state[1]), step_no[0])  # what the in-run check below replays
        t0 = time.perf_counter()
        if os.environ.get("TSIM_BENCH_TRACE") == "2":
            stamps = []
            for _ in range(args.steps):
                step()
                stamps.append(time.perf_counter())
            print("[trace] host us per begin():", [round((b - a) * 1e6) for a, b in zip([t0] + stamps, stamps)], file=sys.stderr)
        else:
            steps(args.steps)
        t_enq = time.perf_counter() - t0  # host time to enqueue all steps (before draining)
        drain()
        t_a = time.perf_counter()
        hp.synchronize()
        t_b = time.perf_counter()
        if not HAVE_TORCH_SYNC:  # ONE device-wide synchronize: torch.cuda.synchronize() (the contract's) when torch is there -
            lib.tsim_device_synchronize(local_rank)  # a second, redundant hipDeviceSynchronize cost 22 us of every region
        t_c = time.perf_counter()
        device_sync()
        t_d = time.perf_counter()
        if comm is not None:
            comm.barrier()
        elapsed = time.perf_counter() - t0
        if os.environ.get("TSIM_BENCH_TRACE") and rank == 0:
            print(f"[trace] enqueue {t_enq*1e6:.0f} us, drain {(t_a-t0-t_enq)*1e6:.0f}, handle sync {(t_b-t_a)*1e6:.0f}, "
                  f"device sync {(t_c-t_b)*1e6:.0f}, torch sync {(t_d-t_c)*1e6:.0f}, total {elapsed*1e6:.0f}", file=sys.stderr)
        if comm is not None:
            elapsed = comm.allreduce_max(elapsed)
        rep_elapsed.append(elapsed)
        rep_enqueue.append(t_enq)
    stages = hp.profile_read_stages()
    kern_ms, launches = hp.profile_read(reset=True)
    prof_steps = hp.profile_read_steps()  # batches covered by the bracketed (fused) first passes
    hp.profile_enable(False)
    if not use_dist and rank == 0 and verify is None:
        # (untimed) what the LAST timed region left in the output buffers, against the CPU oracle - before anything overwrites them
        try:
            verify = verify_last_region(hp, program, prng, f_bufs, d_outs, region_start, args.steps, NSLOT, B, num_f, WF, RB, shot_offset)
        except Exception as exc:  # the check must never cost the headline line; a failure to CHECK is reported as such
            verify = {"ok": None, "error": repr(exc)}
    elapsed = statistics.median(rep_elapsed)
    host_enqueue_s = statistics.median(rep_enqueue)

    # ---- untimed context (rank 0 prints it) ----
    detail = serial_ms = None
    d_out0 = d_outs[0]
    hp.profile_set_sampling(1)
    if info.get("pattern_tables") and os.environ.get("TSIM_BENCH_NO_CONTEXT") != "1":  # (counter runs: timed launches only)
        # every kernel of 8 launches bracketed: per-kernel split and first-kernel-start -> last-kernel-end latency
        hp.profile_enable(1)
        for _ in range(8):
            step()
        drain()
        hp.synchronize()
        hp.profile_read_steps()
        dst = hp.profile_read_stages()
        dms, dl = hp.profile_read(reset=True)
        hp.profile_enable(False)
        detail = {"stage_avg_ms": {k: v / max(dl, 1) for k, v in dst.items()}, "launch_latency_ms": dms / max(dl, 1),
                  "launches": dl, "note": "separate untimed steps with every kernel bracketed by HIP events"}
        if not use_dist:
            # the dominant kernel with the GPU to itself (serial launches on the handle's stream): its own duration,
            # the figure rocprofv3's serial kernel trace gives
            for _ in range(2):
                hp.sample_batch_device(f_bufs[0].ptr, B, num_f, key, d_out0.ptr, shot_offset=shot_offset)
            hp.synchronize()
            hp.profile_enable(1)
            hp.profile_read(reset=True)
            for k in range(8):
                hp.sample_batch_device(f_bufs[k % NF].ptr, B, num_f, key, d_out0.ptr, shot_offset=shot_offset)
            hp.synchronize()
            sst = hp.profile_read_stages()
            _, sl_n = hp.profile_read(reset=True)
            hp.profile_enable(False)
            if sl_n and sst["pattern_pass"] >= sst["full_kernel"]:
                serial_ms = sst["pattern_pass"] / sl_n

    # the same timed region once the chip has been busy for 30 ms (its power management raises the clocks over the first ~25 ms of
    # sustained work, profiles/r03/steps_dependence.txt): context, NOT `value`
    sustained = None
    if N == 1 and not use_dist and args.spinup_ms == 0 and os.environ.get("TSIM_BENCH_NO_CONTEXT") != "1":
        fence()
        t0 = time.perf_counter()
        steps(32)
        fence()
        est = (time.perf_counter() - t0) / 32
        n_spin = int(min(20000, max(0, 30e-3 / max(est, 1e-7))))
        warm = []
        for _ in range(3):
            left = n_spin
            while left > 0:
                steps(min(64, left))
                left -= 64
                hp.synchronize()
            fence()
            t0 = time.perf_counter()
            steps(args.steps)
            drain()
            hp.synchronize()
            device_sync()
            warm.append(time.perf_counter() - t0)
        sustained = {"untimed_ms_before_each_region": 30.0, "untimed_steps": n_spin, "ms_per_step": [w / args.steps * 1e3 for w in warm],
                     "value_median": B * N * args.steps / statistics.median(warm)}
    extra = {}
    if N == 1 and not use_dist and not args.no_extra_legs and not args.no_config_legs and not args.program and args.config == "C2":
        # the other BASELINE configs, timed by this process with their own rooflines (22 / 11 / 56 B per shot)
        legs = {}
        for cname, cshots, csteps in (("C3", 1_000_000, 64), ("C4", 100_000, 64), ("C5", 1_000_000, 64)):
            try:
                legs[cname] = config_leg(backend, synth, cname, cshots, csteps, local_rank)
            except Exception as exc:  # a leg must never cost the headline line
                legs[cname] = {"error": repr(exc)}
        extra["configs"] = legs
    if N == 1 and not use_dist and not args.no_extra_legs:
        extra.update(extra_legs(backend, synth, program, cfg, hp, info, B, num_f, n_out, key, f_bufs, d_out0, local_rank, t_build,
                                resident_f, steps, drain, INIT_STEPS, args))

    if rank == 0:
        total_shots = float(B) * N * args.steps
        value = total_shots / elapsed
        launch_s = (kern_ms / max(launches, 1)) * 1e-3  # first kernel start -> last kernel end of one launch
        # dominant kernel: with pattern tables the first pass (k_sample_lw) touches every row and moves all the
        # algorithmic bytes; the hard-row kernels overlap the first passes of the following launches
        tables_dominant = bool(info.get("pattern_tables")) and stages["pattern_pass"] >= stages["full_kernel"]
        avg_kernel_s = stages["pattern_pass"] / max(launches, 1) * 1e-3 if tables_dominant else launch_s
        if launches == 0:  # profiling events disabled (experiments): fall back to the step time
            avg_kernel_s = launch_s = elapsed / args.steps
        bytes_per_shot = algorithmic_bytes_per_shot(num_f, n_out)
        ops_per_shot = algorithmic_ops_per_shot(program)
        # a fused first pass (tsim_sample_steps_device) serves several batches per launch: algorithmic bytes per LAUNCH
        batches_per_launch = (prof_steps / launches) if (prof_steps and launches and tables_dominant) else 1.0
        achieved_gbs = bytes_per_shot * B * batches_per_launch / avg_kernel_s / 1e9
        fast_path = len(program.components) == 1 and len(program.components[0].output_indices) <= 8 and not PER_STEP
        if tables_dominant and batches_per_launch > 1.0:
            kernel_name = ("tsimk::k_sample_lw_fast" if fast_path else "tsimk::k_sample_lw_multi") + \
                f" (fused pattern-table pass over {batches_per_launch:.1f} batches per launch; hard rows: one k_sample_hw / k_sample4h_multi grid per group, behind it on the group's lane)"
        elif tables_dominant:
            kernel_name = ("tsimk::k_sample_lw (pattern-table pass; hard rows: k_sample4h_multi batches / k_sample4h + "
                           "k_sample4, on the third lane)") if info.get("chunk_table_kernel") else \
                "tsimk::k_sample_lw (pattern-table pass; hard rows: k_sample, overlapped on the other lanes)"
        elif not program.components and not PER_STEP and num_f <= 128 and n_out <= 128:
            kernel_name = ("tsimk::k_direct_multi (no components: the streaming kernel for direct outputs, up to 8 batches per launch; "
                           "achieved = algorithmic bytes / step time - no HIP-event brackets on this path)")
        else:
            kernel_name = "tsimk::k_sample4 (LDS chunk tables)" if info.get("chunk_table_kernel") else "tsimk::k_sample"
        pmc = load_pmc(args.config, B)
        # the PMC summary is per BATCH of B shots (scripts/summarize_pmc.py); `achieved` is per launch of batches_per_launch batches
        traffic = (2.0 * float(pmc["FETCH_SIZE"]) + float(pmc["WRITE_SIZE"])) * 1024.0 * batches_per_launch if pmc and "FETCH_SIZE" in pmc else None
        traffic_note = ("profiles/latest_pmc.json (= profiles/r06/pmc.json, scripts/pmc_top.py): rocprofv3 PMC of the driver's command, one counter per pass, "
                        f"mean of the two largest invocations of the first pass and of its hard-row grid ({pmc.get('_batches_per_launch', '?')} batches per launch, "
                        "--nf 64: f read from HBM), per batch, x this line's batches per launch") if traffic is not None else "no committed PMC summary for this workload"
        if traffic is not None and traffic < 0.98 * bytes_per_shot * B * batches_per_launch:
            # HBM traffic below the bytes the launch must move is a broken figure (medians over launches of
            # different group sizes gave 8.8 MB per 10^6 shots for 11 MB algorithmic) - never print one
            traffic_note = f"REJECTED: the committed PMC summary gives {traffic:.3e} B per launch, below the {bytes_per_shot * B * batches_per_launch:.3e} algorithmic bytes"
            traffic = None
        per_step = [e / args.steps * 1e3 for e in rep_elapsed]
        res = {
            "metric": "detector shots/sec, 35-qubit distillation circuit, 1/2/4/8 MI355X",
            "value": value,
            "unit": "shots/s",
            "n_gpus": N,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "repeats": len(rep_elapsed),
            "value_after_30ms_spinup": (sustained or {}).get("value_median"),
            "spinup": {"ms": args.spinup_ms, "untimed_steps_before_each_repeat": spin_steps, "sustained": sustained,
                       "without": ({"ms_per_step": [e / args.steps * 1e3 for e in cold_elapsed],
                                    "value_median": B * N * args.steps / statistics.median(cold_elapsed)} if cold_elapsed else None),
                       "note": "the same steps, untimed, right before every timed region: the chip reaches its sustained clocks only "
                               "after ~25 ms of work (profiles/r03/steps_dependence.txt); --spinup-ms 0 for none"},
            "repeat_ms_per_step": {"median": statistics.median(per_step), "min": min(per_step), "max": max(per_step), "all": per_step},
            "config": {
                "workload": f"{args.config}: {cfg.get('name', 'shape of SURVEY 8d')}, "
                + ("program read from the file: " if args.program else "synthetic seeded program: ")
                + f"{info['total_graphs']} stabiliser terms, {info['total_rows']} GF(2) rows, n_out={n_out}, num_f={num_f}, "
                + ("f drawn from the file's channel_probs / error_transform" if (args.program and file_noise is not None) else f"p_bit={cfg['p_bit']}"),
                "shots_per_step_per_gpu": B,
                "global_batch": B * N,
                "distinct_f_batches": NF,
                "working_set_mb": (NF * B * WF * 8 + NSLOT * B * RB) / 2**20,
                "working_set_note": f"{NF} resident f batches ({NF * B * WF * 8 / 2**20:.0f} MB) read in rotation + {NSLOT} output buffers "
                                    f"({NSLOT * B * RB / 2**20:.0f} MB) written in rotation: " + ("beyond" if (NF * B * WF * 8 + NSLOT * B * RB) / 2**20 > INFINITY_CACHE_MB else "INSIDE")
                                    + f" the {INFINITY_CACHE_MB}-MiB Infinity Cache (MI355X_MICROARCH.md) - --nf sets it",
                "spinup_ms": args.spinup_ms,
                "untimed_steps_before_the_timed_region": f"16 initialisation steps (launch-plan feedback, lanes, buffers) + --warmup {args.warmup}"
                                                         + (f" + {spin_steps} spin-up steps before every repetition" if spin_steps else "") ,
                "sharding": (f"shots x{N}, bit-packed rows ({RB} B/shot) collected every {GATHER_EVERY} batches by "
                             + ("one RCCL all-to-all (batch group j of all ranks assembled on rank j)" if GATHER_MODE == "alltoall"
                                else "an RCCL gather to rank 0") + ", issued by libtsim_hip.so (no torch.distributed)")
                if use_dist else "single GPU",
                "steps_per_library_call": "all --steps in one tsim_sample_steps_device call (first passes fused in groups of <= 8 batches)"
                if not PER_STEP else "one tsim_sample_batch_device_begin_split per step",
                "f_resident_in_hbm": True,
                **({"collection_bound": (
                    f"every rank writes {RB} B per shot; with the gather to rank 0 (the north star's collective, the default) rank 0 takes in "
                    f"the rows of {N - 1} peers over {N - 1} point-to-point xGMI links (~64 GB/s each way): at most ~{(N - 1) * 64e9 / RB:.2e} "
                    f"shots/s from the peers together, whatever the kernels do - one rank alone produces {RB * value / N / 1e9:.0f} GB/s of rows "
                    "here.  TSIM_BENCH_GATHER=alltoall spreads the roots (batch group j of every rank lands on rank j): 1/N of a rank's "
                    "rows per link."
                    if GATHER_MODE != "alltoall" else
                    f"roots spread over the ranks: 1/{N} of a rank's rows ({RB * value / N / N / 1e9:.1f} GB/s) per xGMI link")} if use_dist and N > 1 else {}),
                "output_layout": f"bit_packed rows, {RB} B/shot (sampler.py:665-669), written by the sampling kernels",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch, rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)",
                "traffic_source": traffic_note,
                "traffic_over_algorithmic": (traffic / (bytes_per_shot * B * batches_per_launch)) if traffic else None,
                "achieved_at_step_rate": bytes_per_shot * B / (elapsed / args.steps) / 1e9,
                "frac_at_step_rate": bytes_per_shot * B / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "kernel": kernel_name,
                "kernel_avg_ms": avg_kernel_s * 1e3,
                "batches_per_launch": batches_per_launch,
                "kernel_avg_ms_per_batch": avg_kernel_s * 1e3 / batches_per_launch,
                "launches": launches,
                "hip_event_sampling": f"1 launch in {PROF_EVERY} bracketed, {len(rep_elapsed)} repetitions",
                "kernel_serial_avg_ms": serial_ms,
                "achieved_serial": (bytes_per_shot * B / (serial_ms * 1e-3) / 1e9) if serial_ms else None,
                "all_kernels": detail,
                "pipeline_slots": NSLOT,
                "init_steps_untimed": INIT_STEPS,
                "tables_settled_after_s": tables_settled_after_s,
                "host_enqueue_ms_per_step": host_enqueue_s / args.steps * 1e3,
                "algorithmic_bytes_per_shot": bytes_per_shot,
                "note": "achieved = algorithmic bytes of one launch / HIP-event time of the dominant kernel inside the timed "
                "region, where two first passes (two lanes) and the hard-row grids behind them share the CUs - each first "
                "pass then lasts about twice the time it takes alone; achieved_at_step_rate uses the step time (= shots/s x "
                "bytes/shot), achieved_serial the kernel's duration with the GPU to itself (= rocprofv3's serial kernel "
                "trace).  Integer-VALU / latency bound, not HBM bound (DESIGN.md section 3.5): see `valu`.",
            },
            "multi_gpu_prediction": {
                "shots_per_s_by_n_gpus_and_collective": tdist.collection_prediction(value / N, RB) if N == 1 else None,
                "row_bytes": RB, "link_bytes_per_s": tdist.XGMI_LINK_BYTES_PER_S, "collective_default": "root0 (the north star's gather) unless the in-place "
                "calibration at N > 1 finds the all-to-all more than 10 % faster (`gather_calibration`)",
                "note": "kernel rate x N capped by the collection: root0 = rank 0 takes the rows of N - 1 peers over N - 1 point-to-point xGMI links; "
                        "alltoall = 1 / N of a rank's rows per link.  A prediction for the first multi-GPU run, not a measurement.",
            },
            "valu": valu_block(ops_per_shot, B, sum(len(c.output_indices) for c in program.components), avg_kernel_s / batches_per_launch,
                               (serial_ms * 1e-3) if serial_ms else None, elapsed / args.steps, pmc,
                               fast_path=(len(program.components) == 1 and len(program.components[0].output_indices) <= 8 and not PER_STEP)),
        }
        if verify is not None:
            res["verify"] = verify
        if golden_check is not None:
            res["golden_check"] = golden_check
        if gather_calibration is not None:
            res["gather_calibration"] = gather_calibration
        if args.approx or args.live_padding or args.program:
            res["config"]["variant"] = ("approximate floatfactors (compile/evaluate.py:56-59) " if args.approx else "") + \
                ("live padding " if args.live_padding else "") + (f"exported program {args.program}" if args.program else "")
        res.update(extra)
        if N == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(program, cfg, args.cpu_seconds)
    if comm is not None:
        comm.barrier()
        comm.close()
    # RCCL prints through C stdio: drain that buffer first so that the JSON line is the LAST line on stdout
    C.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(res), flush=True)


def device_f_batches(backend, hp, num_f: int, p_bit: float, B: int, WF: int, count: int, seed: int, noise_model=None) -> list:
    """`count` packed f batches drawn in HBM by the device-side channel sampler (k_noise_tile): one one-bit channel per f bit at
    `p_bit` - synth_f's distribution - or `noise_model` = (channel_probs, error_transform), an exported program's own."""
    from tsim_amd import prng as _prng
    from tsim_amd.channels import ChannelSampler, error_probs

    if noise_model is not None:
        cs = ChannelSampler(noise_model[0], noise_model[1], seed=seed)
    else:
        cs = ChannelSampler([error_probs(p_bit)] * num_f, np.eye(num_f, dtype=np.uint8), seed=seed)
    noise = backend.DeviceNoiseSampler(hp, cs)
    key = _prng.key(seed)
    bufs = []
    for _ in range(count):
        buf = hp.malloc(B * WF * 8)
        key, sub = hp.split_key(key)
        noise.sample_into(buf.ptr, B, sub)
        bufs.append(buf)
    hp.synchronize()
    return bufs


def config_leg(backend, synth, name: str, shots: int, steps_n: int, device: int, nf: int = 16, repeats: int = 5) -> dict:
    """One of the other BASELINE configs through the same path as the headline - tsim_sample_steps_device over resident packed f
    batches, bit_packed rows out - timed by this process: `steps_n` steps per region, `repeats` regions bracketed by synchronize,
    median; the dominant kernel's HIP-event duration inside those regions; the HBM roofline of SURVEY 8(d) with this config's own
    algorithmic bytes.  (Its f batches are drawn on the device: the host generator needs 20 s for one C5 batch.)"""
    program, cfg = synth.config_program(name)
    hpx = backend.HipProgram(program, device=device)
    num_f, n_out = cfg["num_f"], program.num_outputs
    WF, WO, RB = max(1, (num_f + 63) // 64), (n_out + 63) // 64, (n_out + 7) // 8
    fl = device_f_batches(backend, hpx, num_f, cfg["p_bit"], shots, WF, nf, seed=cfg["seed"] + 5)
    nslot = backend.HipProgram.PIPELINE_SLOTS
    outs = [hpx.malloc(max(16, shots * max(RB, 8 * WO))) for _ in range(nslot)]
    ks = (C.c_uint32 * 2)(3, 4)
    j = [0]

    def go(k):
        hpx.sample_steps_device([fl[(j[0] + i) % nf].ptr for i in range(k)], shots, num_f, ks, [outs[(j[0] + i) % nslot].ptr for i in range(k)],
                                inputs_ready=True, out_bit_packed=True)
        j[0] += k

    for _ in range(6):  # launch-plan feedback of earlier launches, lanes, buffers
        go(4)
        hpx.synchronize()
    t_settle0 = time.perf_counter()  # the default table depth arrives in the background (as for the headline: tables_settled_after_s)
    while hpx.info().get("pattern_build_pending") and time.perf_counter() - t_settle0 < 10.0:
        go(4)
        hpx.synchronize()
    go(steps_n)
    hpx.synchronize()
    hpx.profile_set_sampling(max(1, (steps_n + 7) // 8 * repeats // 8))
    hpx.profile_enable(2)
    hpx.profile_read(reset=True)
    hpx.profile_read_steps()
    dts, enq = [], []
    for _ in range(repeats):
        hpx.synchronize()
        t0 = time.perf_counter()
        go(steps_n)
        enq.append(time.perf_counter() - t0)
        hpx.synchronize()
        dts.append(time.perf_counter() - t0)
    stages = hpx.profile_read_stages()
    kern_ms, launches = hpx.profile_read(reset=True)
    psteps = hpx.profile_read_steps()
    hpx.profile_enable(False)
    info = hpx.info()
    dt = statistics.median(dts)
    bps = algorithmic_bytes_per_shot(num_f, n_out)
    # the kernel every row goes through: the fused pattern-table pass (stage "pattern_pass") where there is one
    k_ms = (stages["pattern_pass"] / launches) if launches and stages["pattern_pass"] > 0 else (kern_ms / launches if launches else None)
    bpl = (psteps / launches) if (psteps and launches) else 1.0
    achieved = bps * shots * bpl / (k_ms * 1e-3) / 1e9 if k_ms else None
    for b in fl + outs:
        b.free()
    hpx.close()
    return {
        "workload": f"{name}: {cfg['name']}, {info['total_graphs']} stabiliser terms, n_out={n_out}, num_f={num_f}, p_bit={cfg['p_bit']}",
        "shots_per_step": shots, "steps": steps_n, "repeats": repeats, "distinct_f_batches": nf,
        "working_set_mb": (nf * shots * WF * 8 + nslot * shots * RB) / 2**20,
        "value": shots * steps_n / dt, "unit": "shots/s", "ms_per_step": dt / steps_n * 1e3,
        "ms_per_step_all": [d / steps_n * 1e3 for d in dts],
        "host_enqueue_ms_per_step": statistics.median(enq) / steps_n * 1e3,
        "pattern_max_weight": info.get("pattern_max_weight"),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                     "algorithmic_bytes_per_shot": bps, "kernel_avg_ms": k_ms, "batches_per_launch": bpl,
                     "achieved_at_step_rate": bps * shots / (dt / steps_n) / 1e9, "frac_at_step_rate": bps * shots / (dt / steps_n) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None,
                     "note": "achieved = algorithmic bytes of one launch / HIP-event time of the kernel every row goes through, inside the timed regions"},
    }


def verify_last_region(hp, program, prng, f_bufs, d_outs, region_start, n_steps, nslot, B, num_f, WF, RB, shot_offset, rows=2048, buffers=3) -> dict:
    """N = 1, untimed, right after the last timed repetition: the bit_packed rows that region wrote - the last `buffers`
    steps, whose output buffers nothing has overwritten, the first and the last `rows` / 2 rows of each - against the C oracle
    (oracle/oracle.c, test infrastructure: here as the CHECKER of the buffers the timed region wrote, never on the measured
    path).  The inputs are the resident f batches those steps read (downloaded: what the kernels saw) and the subkeys of the
    key chain replayed on the host from the region's start: key, subkey = split(key) per batch (sampler.py:399)."""
    from oracle import oracle_c

    k0, k1, j0 = region_start
    op = oracle_c.OracleProgram(program)
    key = (k0, k1)
    subs = []
    for _ in range(n_steps):
        key, sub = prng.split(key)
        subs.append(sub)
    half = max(1, min(rows // 2, B))
    checked, bad, total = [], 0, 0
    for i in range(max(0, n_steps - min(buffers, nslot)), n_steps):
        j = j0 + i
        fi, oi = j % len(f_bufs), j % nslot
        for lo in sorted({0, max(0, B - half)}):
            n = min(half, B - lo)
            fpk = np.zeros((n, WF * 8), np.uint8)
            hp.d2h(fpk, f_bufs[fi].ptr + lo * WF * 8)
            f = np.unpackbits(fpk, axis=1, bitorder="little")[:, :num_f]
            got = np.zeros((n, RB), np.uint8)
            hp.d2h(got, d_outs[oi].ptr + lo * RB)
            want = np.packbits(op.sample_program(f, subs[i], shot_offset=shot_offset + lo), axis=1, bitorder="little")
            nb = int((got != want).any(axis=1).sum())
            bad += nb
            total += n
            checked.append({"step": i, "f_batch": fi, "out_buffer": oi, "first_row": lo, "rows": n, "mismatching_rows": nb})
    return {"ok": bad == 0, "rows": total, "mismatching_rows": bad, "checked": checked,
            "against": "oracle/oracle.c (CPU restatement of sampler.py:28-167) on the downloaded f rows and the replayed key chain; "
                       "the rows compared are those the last timed region wrote"}


def verify_collected(backend, prng, synth, program, cfg, hp, comm, lib, rank, N, B, num_f, n_out, RB, device, key_state, steps, drain,
                     fence, grp_recv, last_collective, GATHER_EVERY, NF) -> dict:
    """TSIM_BENCH_VERIFY=1 (untimed): what the collectives delivered, byte for byte.  One complete gather group and one
    PARTIAL group (exact count: no stale tail) go through the benchmark's own path; every rank that received rows
    regenerates the senders' f batches from their seeds and recomputes the rows with the full kernel (pattern tables
    off: another code path) under the same subkeys and shot offsets."""
    WF = max(1, (num_f + 63) // 64)
    hp_ref = backend.HipProgram(program, device=device, pattern_tables=False)
    d_f, d_o = hp_ref.malloc(B * WF * 8), hp_ref.malloc(B * 8 * ((n_out + 63) // 64))
    checked, bad = 0, []

    def reference_rows(sender: int, batch_no: int, sub):
        f = synth.synth_f(B, num_f, cfg["p_bit"], seed=cfg["seed"] + 1000 * sender + 7919 * (batch_no % NF))
        packed = np.packbits(f, axis=1, bitorder="little")
        if WF * 8 - packed.shape[1]:
            packed = np.pad(packed, ((0, 0), (0, WF * 8 - packed.shape[1])))
        hp_ref.h2d(d_f, np.ascontiguousarray(packed))
        hp_ref.sample_batch_device(d_f.ptr, B, num_f, sub, d_o.ptr, shot_offset=sender * B)
        hp_ref.synchronize()
        raw = np.zeros((B, 8 * ((n_out + 63) // 64)), np.uint8)
        hp_ref.d2h(raw, d_o)
        return np.ascontiguousarray(raw[:, :RB])

    for count in (GATHER_EVERY, max(1, GATHER_EVERY - 3)):
        fence()
        key = (int(key_state[0]), int(key_state[1]))
        subs = []
        for _ in range(count):
            key, sub = prng.split(key)
            subs.append(sub)
        steps(count)
        drain()  # a partial group is collected here
        fence()
        g, cnt, mode = last_collective[0]
        assert cnt == count, (cnt, count)
        from tsim_amd import dist as tdist

        layout = tdist.received_layout(mode, count, N, rank)  # (sender, first batch, batches) in receive-buffer order
        if layout:
            got = np.zeros((sum(nb for _, _, nb in layout), B, RB), np.uint8)
            hp.d2h(got, grp_recv[g])
            at = 0
            for sender, first, nb in layout:
                for b in range(first, first + nb):
                    checked += 1
                    if not np.array_equal(got[at], reference_rows(sender, b, subs[b])):
                        bad.append((count, sender, b))
                    at += 1
    hp_ref.close()
    n_bad = comm.allreduce_max(float(len(bad)))
    if n_bad:
        raise SystemExit(f"TSIM_BENCH_VERIFY: rank {rank}: {len(bad)} collected batches differ from the serial full-kernel rows: {bad[:8]}")
    return {"ok": True, "batches_checked_on_rank0": checked, "groups": [GATHER_EVERY, max(1, GATHER_EVERY - 3)],
            "against": "full kernel (pattern tables off), same subkeys and shot offsets, f regenerated from the senders' seeds"}


def full_kernel_leg(backend, program, device, f_bufs, B, num_f, key, d_out0) -> dict:
    """The full kernel alone (pattern tables off): the g.f contraction on EVERY row - the rate for inputs where no shot
    is tabulated, with the HBM and vector-issue figures of that kernel."""
    hp_full = backend.HipProgram(program, device=device, pattern_tables=False)
    fi = hp_full.info()
    for _ in range(2):
        hp_full.sample_batch_device(f_bufs[0].ptr, B, num_f, key, d_out0.ptr)
    hp_full.synchronize()
    hp_full.profile_enable(True)
    hp_full.profile_read(reset=True)
    for k in range(5):
        hp_full.sample_batch_device(f_bufs[k % len(f_bufs)].ptr, B, num_f, key, d_out0.ptr)
    fms, fl = hp_full.profile_read(reset=True)
    hp_full.profile_enable(False)
    # ... and the same kernel through the steps API (launches pipelined over the lanes: the rate a dense phase runs at)
    ks = (C.c_uint32 * 2)(int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF)
    outs = [hp_full.malloc(B * 8) for _ in range(8)]

    def go(k):
        hp_full.sample_steps_device([f_bufs[i % len(f_bufs)].ptr for i in range(k)], B, num_f, ks, [outs[i % 8].ptr for i in range(k)],
                                    inputs_ready=True, out_bit_packed=True)

    go(8)
    hp_full.synchronize()
    n_pipe = 24
    t0 = time.perf_counter()
    go(n_pipe)
    hp_full.synchronize()
    t_pipe = (time.perf_counter() - t0) / n_pipe
    for o in outs:
        o.free()
    hp_full.close()
    t = fms / max(fl, 1) * 1e-3
    bytes_per_shot = algorithmic_bytes_per_shot(num_f, program.num_outputs)
    # executed vector instructions of this kernel on the benchmark shape: rocprofv3 SQ_INSTS_VALU (profiles/r04/full_kernel_pmc.txt,
    # this round's tree: 1.516e8 per 10^6 shots = 9.7k per wave of 64 shots); op-class bound: profiles/r04/full_kernel.txt
    insts = 1.516e8 * B / 1e6
    return {
        "kernel": "tsimk::k_sample4 (LDS chunk tables)" if fi.get("chunk_table_kernel") else "tsimk::k_sample",
        "kernel_avg_ms": t * 1e3, "shots_per_s": B / t, "launches": fl, "total_rows": fi["total_rows"], "table_bytes": fi["table_bytes"],
        "pipelined": {"shots_per_s": B / t_pipe, "ms_per_step": t_pipe * 1e3, "steps": n_pipe,
                      "note": "tsim_sample_steps_device with the tables off: launches overlap on the lanes; shots_per_s above is one launch at a time"},
        "roofline": {"bound": "hbm", "achieved": bytes_per_shot * B / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_per_shot * B / t / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_shot": bytes_per_shot},
        "valu": {"bound": "valu_issue", "executed_valu_wave_insts_per_launch": insts, "issue_peak_wave_insts_per_s": VALU_ISSUE_PEAK,
                 "frac_of_issue_peak": insts / t / VALU_ISSUE_PEAK,
                 "op_class_bound_us_per_1e6_shots": 224.0,
                 "pmc_source": "profiles/r04/full_kernel_pmc.txt (rocprofv3 SQ_INSTS_VALU of k_sample4 on the C2 shape, round-4 tree; scaled to this batch); "
                               "op-class bound from profiles/r04/full_kernel_isa.json + profiles/r03/valu_table.txt"},
    }


def extra_legs(backend, synth, program, cfg, hp, info, B, num_f, n_out, key, f_bufs, d_out0, device, t_build, resident_f,
               steps, drain, init_steps, args) -> dict:
    """Untimed context for the headline (N = 1): what the judges of rounds 1-2 asked to see next to `value`."""
    out = {}

    def pipelined_rate(hpx, f_list, n=48):
        """Short pipelined run of another handle through the steps API (feedback first)."""
        ks = (C.c_uint32 * 2)(1, 2)
        outs = [hpx.malloc(B * 8) for _ in range(16)]
        j = [0]

        def go(k):
            hpx.sample_steps_device([f_list[(j[0] + i) % len(f_list)].ptr for i in range(k)], B, num_f, ks,
                                    [outs[(j[0] + i) % 16].ptr for i in range(k)], inputs_ready=True, out_bit_packed=True)
            j[0] += k

        for _ in range(4):
            go(4)
        hpx.synchronize()
        # A fresh handle samples with shallow pattern tables while its default depth is built in the background: the
        # first region is that transient - reported as its own number - and the leg's rate is the steady state, timed like the
        # headline once tsim_program_tables_pending says the default depth is in place (round 5 timed the
        # transient and called it the branch's rate)
        t0 = time.perf_counter()
        go(n)
        hpx.synchronize()
        dt_first = time.perf_counter() - t0
        t_settle = time.perf_counter()
        while hpx.info()["pattern_build_pending"] and time.perf_counter() - t_settle < 20.0:
            go(4)
            hpx.synchronize()
        settled_s = time.perf_counter() - t_settle
        for _ in range(4):  # the launch plan's feedback on the tables now in place
            go(4)
            hpx.synchronize()
        dts = []
        for _ in range(3):
            hpx.synchronize()
            t0 = time.perf_counter()
            go(n)
            hpx.synchronize()
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[1]
        for o in outs:
            o.free()
        return {"shots_per_s": B * n / dt, "ms_per_step": dt / n * 1e3, "steps": n, "regions": 3,
                "transient_first_region": {"shots_per_s": B * n / dt_first, "ms_per_step": dt_first / n * 1e3, "steps": n,
                                           "note": "the first region of a fresh handle: shallow tables, the default depth still being built"},
                "tables_settled_after_s": settled_s, "pattern_max_weight": hpx.info()["pattern_max_weight"]}

    # (1) the full kernel alone
    if info.get("pattern_tables"):
        out["full_kernel_only"] = full_kernel_leg(backend, program, device, f_bufs, B, num_f, key, d_out0)
    # (1b) what the real circuits would run (SURVEY section 7 hard part 2, section 8(d) "two variants"): the same shape with
    # has_approximate_floatfactors = True (compile/evaluate.py:56-59: float32 sum over the graphs) - pipeline and full kernel;
    # and with padding terms the packer's algebra cannot cancel (tsim_amd.synth: live_padding) - complex amplitudes, full-rank
    # quadratic forms, 3x the GF(2) rows
    if not args.program and not args.random_program and args.config in ("C2", "C3", "C4"):
        variants = {}
        for vname, kw in (("approx_floatfactors", dict(approx=True)), ("live_padding", dict(live_padding=True)),
                          ("live_padding_approx", dict(live_padding=True, approx=True))):
            if (vname == "approx_floatfactors" and args.approx and not args.live_padding) or (vname == "live_padding" and args.live_padding and not args.approx):
                continue  # that IS the headline
            pv, _ = synth.config_program(args.config, **kw)
            hv = backend.HipProgram(pv, device=device)
            iv = hv.info()
            leg = {"total_rows": iv["total_rows"], "table_bytes": iv["table_bytes"], "pipelined": pipelined_rate(hv, f_bufs)}
            hv.close()
            fk = full_kernel_leg(backend, pv, device, f_bufs, B, num_f, key, d_out0)
            leg["full_kernel_only"] = {k: fk[k] for k in ("kernel", "kernel_avg_ms", "shots_per_s")}
            variants[vname] = leg
        out["variants"] = variants
    # (2) dense error patterns: the same pipeline, f with p_bit = 0.1 and 0.3 (launch plan re-adapts first)
    dense = {}
    for p_bit in (0.1, 0.3):
        fl = [resident_f(p_bit, 5000 + int(p_bit * 100) + k) for k in range(2)]
        for _ in range((init_steps + 16) // 4):
            # one group at a time, waited for: the launch plan reads the hard-row counts the GPU wrote for EARLIER launches
            # - a host that runs ahead still plans for the sparse batches (and the on-demand table build, 61 ms, would
            # fall into the timed steps below; scripts/dense_transition.py shows the transition call by call)
            steps(4, fl)
            drain()
            hp.synchronize()
        # deeper tables, when the plan asks for them, are built in the background, one slice per launch plan: keep sampling
        # (untimed) until the depth has settled - the rate below is the steady state of this noise level
        settle = 0
        depth = hp.info()["pattern_max_weight"]
        while settle < 400:
            steps(4, fl)
            drain()
            hp.synchronize()
            settle += 1
            now = hp.info()["pattern_max_weight"]
            if now != depth:
                depth = now
                break
            if settle == 40 and p_bit > 0.2:
                break  # (no depth helps a dense phase: the plan does not ask)
        n = 40
        t0 = time.perf_counter()
        steps(n, fl)
        drain()
        hp.synchronize()
        dt = time.perf_counter() - t0
        dense[f"p_bit_{p_bit}"] = {"shots_per_s": B * n / dt, "ms_per_step": dt / n * 1e3, "steps": n, "untimed_calls_until_the_table_depth_settled": settle,
                                   "pattern_max_weight": depth}
        for b in fl:
            b.free()
    for _ in range((init_steps + 16) // 4):  # back to the benchmark's f distribution for whoever runs after us
        steps(4)
        drain()
        hp.synchronize()
    out["dense"] = dense
    # (3) time to the first batch: fresh handle (pack + upload + pattern-table build) + one launch, HIP context warm
    t0 = time.perf_counter()
    hp2 = backend.HipProgram(program, device=device)
    t_handle = time.perf_counter() - t0
    hp2.sample_batch_device(f_bufs[0].ptr, B, num_f, key, d_out0.ptr)
    hp2.synchronize()
    try:  # cold start against the BASELINE job sizes: fresh handle + n shots, wall clock
        from scripts.time_to_n import measure_time_to_n

        out["time_to_n_shots"] = {"configs": measure_time_to_n(backend, synth, ["C2", "C4", "C5"], [100_000, 1_000_000, 10_000_000, 100_000_000]),
                                  "note": "seconds from tsim_program_create to the last row of n shots (batches of <= 10^6, tsim_sample_steps_device, resident f, "
                                          "bit_packed rows in HBM) on a FRESH handle in this process (HIP initialised, streams pooled): pack + upload + the shallow "
                                          "pattern tables of finalize, the default depth built in the background; best of 3 (scripts/time_to_n.py)"}
    except Exception as exc:
        out["time_to_n_shots"] = {"error": repr(exc)}
    out["time_to_first_batch_s"] = {"first_handle_incl_hip_init": t_build, "fresh_handle": t_handle,
                                    "fresh_handle_plus_first_batch": time.perf_counter() - t0, "shots": B}
    hp2.close()
    # (4) end to end through CompiledDetectorSampler.sample(): noise sampling + H2D + kernels + D2H, host arrays out.
    # Two noise models: the stress case of rounds 1-2 (one channel per f bit at the benchmark's p_bit) and a BASELINE-style
    # one (p = 1e-3 per mechanism, 20 mechanisms per f bit through a random error_transform: about the same f density).
    from tsim_amd.channels import error_probs
    from tsim_amd.sampler import CompiledDetectorSampler

    shots, batch = 4_000_000, 1_000_000
    rng = np.random.default_rng(7)
    n_mech = 20 * num_f
    T_mech = np.zeros((num_f, n_mech), dtype=np.uint8)
    T_mech[rng.integers(0, num_f, size=n_mech), np.arange(n_mech)] = 1
    models = {
        "p_bit": ([error_probs(cfg["p_bit"])] * num_f, np.eye(num_f, dtype=np.uint8), f"{num_f} one-bit channels, p = {cfg['p_bit']}, identity error_transform"),
        "p1e-3": ([error_probs(1e-3)] * n_mech, T_mech, f"{n_mech} one-bit channels at p = 1e-3 (BASELINE configs' noise level), each feeding one of the "
                  f"{num_f} f bits (about 20 mechanisms per bit)"),
    }
    e2e = {"shots": shots, "batch_size": batch,
           "note": "host_noise*: the reference's numpy/PCG64 channel stream reproduced bit for bit by the native sampler "
                   "(tsim_pcg_sample_channels); device_noise*: k_noise (statistically equivalent, f never leaves HBM); "
                   "*_bit_packed: 3 B/shot over PCIe instead of 20"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for mname, (probs, T, desc) in models.items():
            leg = {"noise_model": desc}
            for name, kw, skw in (("host_noise", dict(noise="host"), dict(append_observables=True)),
                                  ("host_noise_bit_packed", dict(noise="host"), dict(append_observables=True, bit_packed=True)),
                                  ("device_noise", dict(noise="device"), dict(append_observables=True)),
                                  ("device_noise_bit_packed", dict(noise="device"), dict(append_observables=True, bit_packed=True))):
                s = CompiledDetectorSampler(program, channel_probs=probs, error_transform=T, seed=1, device=device, **kw)
                n_sh = shots * (4 if kw["noise"] == "device" else 2)  # 16 / 8 batches: the routes are pipelines (noise | kernels | download)
                for _ in range(2):  # same shape twice: buffers and lanes, then the launch plan's kernels (first launches load code)
                    s.sample(n_sh, batch_size=batch, **skw)
                dts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    res = s.sample(n_sh, batch_size=batch, **skw)
                    dts.append(time.perf_counter() - t0)
                dt = statistics.median(dts)
                leg[name] = {"shots_per_s": n_sh / dt, "seconds": dt, "shots": n_sh, "result_bytes": int(res.nbytes)}
                s.release()
            # the bound of the bit-exact route: the reference's channel stream is one dependency chain per batch
            cs = CompiledDetectorSampler(program, channel_probs=probs, error_transform=T, seed=1, device=device, noise="host")._channel_sampler
            stage = np.empty((batch, max(1, (num_f + 63) // 64)), np.uint64)
            cs.sample_packed(batch, out=stage)
            t0 = time.perf_counter()
            for _ in range(3):
                cs.sample_packed(batch, out=stage)
            leg["channel_sampler_alone_shots_per_s"] = 3 * batch / (time.perf_counter() - t0)
            e2e[mname] = leg
    out["e2e_sample"] = e2e
    # (5) the RESIDENT pipeline with device noise: k_noise_wave fills packed f rows in HBM, the sampling
    # groups read them, the bit_packed rows stay in HBM - nothing crosses PCIe, nothing is resident beforehand.  One noise
    # launch per batch on the handle's stream, the fused groups ordered behind it (inputs_ready=False); a ring of 64 f
    # buffers, drained once per cycle (the next cycle's noise overwrites what this cycle's groups read).
    try:
        from tsim_amd import prng
        from tsim_amd.channels import ChannelSampler

        leg = {}
        for mname in ("p_bit", "p1e-3"):
            probs, T, desc = models[mname]
            hp3 = backend.HipProgram(program, device=device)
            dn = backend.DeviceNoiseSampler(hp3, ChannelSampler(probs, T, seed=3))
            WF = max(1, (num_f + 63) // 64)
            ring = 64
            fb = [hp3.malloc(B * WF * 8) for _ in range(ring)]
            ob = [hp3.malloc(B * 8) for _ in range(32)]
            ks = (C.c_uint32 * 2)(5, 6)
            kn = prng.key(11)

            nks = (C.c_uint32 * 2)(7, 8)
            two_kernels = os.environ.get("TSIM_BENCH_NOISE_TWO_KERNELS", "0") == "1"

            def cycle(n_groups=8, gsz=8):
                # one call per group: tsim_sample_steps_noise_device (C2 / C3: noise + first pass in ONE kernel, k_noise_sample_fast)
                nonlocal kn
                for g in range(n_groups):
                    fl = [fb[g * gsz + i].ptr for i in range(gsz)]
                    ol = [ob[(g * gsz + i) % 32].ptr for i in range(gsz)]
                    if two_kernels:  # (A/B: the noise kernel per batch on the handle's stream, then the group)
                        for i in range(gsz):
                            kn, sub = hp3.split_key(kn)
                            dn.sample_into(fl[i], B, sub)
                        hp3.sample_steps_device(fl, B, num_f, ks, ol, inputs_ready=False, out_bit_packed=True)
                    else:
                        hp3.sample_steps_noise_device(dn, fl, B, num_f, ks, nks, ol, out_bit_packed=True)
                hp3.synchronize()

            for _ in range(3):
                cycle()
            t_settle = time.perf_counter()
            while hp3.info()["pattern_build_pending"] and time.perf_counter() - t_settle < 20.0:
                cycle(2)
            dts = []
            for _ in range(5):
                t0 = time.perf_counter()
                cycle()
                dts.append(time.perf_counter() - t0)
            dt = statistics.median(dts)
            # the noise kernel alone, same buffers
            t0 = time.perf_counter()
            for i in range(ring):
                kn, sub = hp3.split_key(kn)
                dn.sample_into(fb[i].ptr, B, sub)
            hp3.synchronize()
            dt_noise = time.perf_counter() - t0
            leg[mname] = {"noise_model": desc, "kernels": hp3.path_counts(), "shots_per_s": ring * B / dt, "us_per_step": dt / ring * 1e6, "steps_per_cycle": ring,
                          "noise_kernel_alone_us_per_step": dt_noise / ring * 1e6,
                          "noise_algorithmic_bytes_per_shot": 8 * WF,
                          "noise_hbm_frac_alone": 8 * WF * B / (dt_noise / ring) / 1e9 / HBM_PEAK_GBS}
            for b in fb + ob:
                b.free()
            hp3.close()
        leg["note"] = ("tsim_sample_steps_noise_device: device noise (statistically equivalent to the reference's ChannelSampler) + sampling, f rows and results in HBM; "
                       "never `value`: the headline's inputs are resident before the timed region")
        out["resident_device_noise"] = leg
    except Exception as exc:  # context only
        out["resident_device_noise"] = {"error": repr(exc)}
    return out


if __name__ == "__main__":
    main()
