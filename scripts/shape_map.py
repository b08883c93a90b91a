#!/usr/bin/env python3
"""Shape map: which kernel family serves each class of program shape around the BASELINE estimates,
and at what rate, through the same call the headline uses (tsim_sample_steps_device over resident packed f batches,
bit_packed rows out).  The reference takes every shape through one code path (src/tsim/sampler.py:117-167,
compile/pipeline.py:55-102); here the shape picks the kernel - this script prints the cliffs.

    python scripts/shape_map.py [--classes a,b,...] [--shots 1000000] [--steps 40] [--out profiles/r05/shape_map.txt]

Per class: kernel families launched in the timed regions (tsim_program_path_counts), pattern-table depth, shots/s, us per
step, the rate of the nearest BASELINE configuration measured in the same run, and the ratio.  `--check` also compares
2000 rows of one batch with the C oracle (the parity test per class lives in tests/test_gpu_shape_classes.py).
"""

from __future__ import annotations

import argparse
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def device_f(backend, hp, num_f, p_bit, B, WF, count, seed):
    from tsim_amd import prng
    from tsim_amd.channels import ChannelSampler, error_probs

    cs = ChannelSampler([error_probs(p_bit)] * num_f, np.eye(num_f, dtype=np.uint8), seed=seed)
    noise = backend.DeviceNoiseSampler(hp, cs)
    key = prng.key(seed)
    bufs = []
    for _ in range(count):
        buf = hp.malloc(B * WF * 8)
        key, sub = hp.split_key(key)
        noise.sample_into(buf.ptr, B, sub)
        bufs.append(buf)
    hp.synchronize()
    return bufs


def measure(backend, program, num_f, p_bit, shots, steps_n, nf=12, repeats=3, check=False, shot_offset=0, pattern_tables=None):
    hp = backend.HipProgram(program, pattern_tables=pattern_tables)
    n_out = program.num_outputs
    WF, WO, RB = max(1, (num_f + 63) // 64), (n_out + 63) // 64, (n_out + 7) // 8
    fl = device_f(backend, hp, num_f, p_bit, shots, WF, nf, seed=47)
    nslot = backend.HipProgram.PIPELINE_SLOTS
    outs = [hp.malloc(max(16, shots * max(RB, 8 * WO))) for _ in range(nslot)]
    ks = (C.c_uint32 * 2)(3, 4)
    j = [0]

    def go(k):
        hp.sample_steps_device([fl[(j[0] + i) % nf].ptr for i in range(k)], shots, num_f, ks, [outs[(j[0] + i) % nslot].ptr for i in range(k)],
                               inputs_ready=True, out_bit_packed=True, shot_offset=shot_offset)
        j[0] += k

    for _ in range(4):  # launch-plan feedback
        go(4)
        hp.synchronize()
    go(min(steps_n, 16))
    hp.synchronize()
    # A fresh handle starts with shallow pattern tables and builds its default depth in the background:
    # the rate right now is the transient's (`first`); the map's figure is the steady state, once nothing is pending.
    t0 = time.perf_counter()
    go(steps_n)
    hp.synchronize()
    first = shots * steps_n / (time.perf_counter() - t0)
    t_settle = time.perf_counter()
    while hp.info()["pattern_build_pending"] and time.perf_counter() - t_settle < 20.0:
        go(4)
        hp.synchronize()
    settle_s = time.perf_counter() - t_settle
    for _ in range(4):  # the plan's feedback on the tables now in place
        go(4)
        hp.synchronize()
    hp.path_counts(reset=True)
    dts = []
    for _ in range(repeats):
        hp.synchronize()
        t0 = time.perf_counter()
        go(steps_n)
        hp.synchronize()
        dts.append(time.perf_counter() - t0)
    paths = hp.path_counts(reset=True)
    info = hp.info()
    ok = None
    if check:
        from oracle import oracle_c as OC
        from tsim_amd import prng

        m = 2000
        f = (np.random.default_rng(5).random((m, num_f)) < p_bit).astype(np.uint8)
        got, _ = hp.sample_batch(f, (9, 10))
        want = OC.OracleProgram(program).sample_program(f, (9, 10))
        ok = bool(np.array_equal(got, want))
    for b in fl + outs:
        b.free()
    hp.close()
    dt = statistics.median(dts)
    return dict(rate=shots * steps_n / dt, us=dt / steps_n * 1e6, paths=paths, depth=info.get("pattern_max_weight"), ok=ok, first=first, settle_s=settle_s,
                table_mb=info.get("pattern_table_bytes", 0) / 2**20)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", default="")
    ap.add_argument("--shots", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-base", action="store_true", help="skip the BASELINE configurations (profiling one class)")
    ap.add_argument("--shot-offset", type=int, default=0, help="diagnostic: a shard that does not hold shot 0 runs no normalisation check")
    a = ap.parse_args()
    from tsim_amd import backend, synth

    names = [n for n in a.classes.split(",") if n] or list(synth.SHAPE_CLASSES)
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit(f"# shape map: {a.shots} shots per step, {a.steps} steps per region, median of 3 regions; p_bit 0.02; TSIM_AMD_TUNE={os.environ.get('TSIM_AMD_TUNE', '')!r}")
    base = {}
    for cn in (() if a.no_base else ("C2", "C3", "C4", "C5")):
        prog, cfg = synth.config_program(cn)
        r = measure(backend, prog, cfg["num_f"], cfg["p_bit"], a.shots, a.steps)
        base[cn] = r["rate"]
        emit(f"{cn:14s} num_f {cfg['num_f']:4d} outputs {prog.num_outputs:4d} comps {[(len(c.output_indices), len(c.f_selection)) for c in prog.components]}"
             f" depth {r['depth']} tables {r['table_mb']:.0f} MB  {r['rate']:.3e} shots/s  {r['us']:.1f} us/step  (first {r['first']:.2e}, settled after {r['settle_s'] * 1e3:.0f} ms)  paths {r['paths']}")
    emit("# class: (n_out, F) per component | kernel families | rate | ratio to the nearest configuration")
    worst = None
    for n in names:
        prog, c = synth.shape_class_program(n)
        try:
            r = measure(backend, prog, c["num_f"], c["p_bit"], a.shots, a.steps, check=a.check, shot_offset=a.shot_offset)
        except Exception as e:  # a class the library refuses is a cliff of its own
            emit(f"{n:14s} FAILED: {e}")
            continue
        ratio = r["rate"] / base.get(c["near"], float("nan"))
        if worst is None or ratio < worst[1]:
            worst = (n, ratio)
        emit(f"{n:14s} num_f {c['num_f']:4d} outputs {prog.num_outputs:4d} comps {[(len(x.output_indices), len(x.f_selection)) for x in prog.components]}"
             f" depth {r['depth']} tables {r['table_mb']:.0f} MB  {r['rate']:.3e} shots/s  {r['us']:.1f} us/step  {ratio:.2f} x {c['near']}  (first region after the shallow start {r['first']:.2e}, settled after {r['settle_s'] * 1e3:.0f} ms)"
             f"{'' if r['ok'] is None else ('  oracle ok' if r['ok'] else '  ORACLE MISMATCH')}  paths {r['paths']}")
    if worst:
        emit(f"# worst class: {worst[0]} at {worst[1]:.2f} of its nearest configuration (target: none below 1/3)")
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
