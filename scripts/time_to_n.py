#!/usr/bin/env python3
"""Cold start against the BASELINE job sizes: a FRESH handle (tsim_program_create .. finalize: pack,
upload, the pattern tables it builds at once) + n shots through tsim_sample_steps_device, wall clock, for n in 1e5 .. 1e8 -
the reference's counterpart is compile_detector_sampler() + sample(n) (src/tsim/sampler.py:340-420).  The f batches are resident
(drawn by a helper handle before the clock starts); HIP itself is initialised before.

    python scripts/time_to_n.py [--configs C2,C4,C5] [--ns 100000,1000000,10000000,100000000]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure_time_to_n(backend, synth, configs, ns, repeats=3, verbose=False) -> dict:
    """{config: {n: {"seconds": best of `repeats`, "fresh_handle_s", "batches", "shots_per_batch", "table_depth_at_end"}}}"""
    from scripts.shape_map import device_f

    res = {}
    for name in configs:
        program, cfg = synth.config_program(name)
        num_f, n_out = cfg["num_f"], program.num_outputs
        WF, RB = max(1, (num_f + 63) // 64), (n_out + 7) // 8
        helper = backend.HipProgram(program)
        res[name] = {}
        for n in ns:
            B = min(n, 1_000_000)
            k = (n + B - 1) // B
            nf = min(k, 8)
            fl = device_f(backend, helper, num_f, cfg["p_bit"], B, WF, nf, seed=11)
            outs = [helper.malloc(max(16, B * RB)) for _ in range(min(k, backend.HipProgram.PIPELINE_SLOTS))]
            helper.synchronize()
            dts, fin = [], []
            depth = None
            for _ in range(repeats):
                t0 = time.perf_counter()
                hp = backend.HipProgram(program)
                t1 = time.perf_counter()
                ks = (C.c_uint32 * 2)(1, 2)
                done = 0
                while done < k:
                    m = min(64, k - done)
                    hp.sample_steps_device([fl[(done + i) % nf].ptr for i in range(m)], B, num_f, ks, [outs[(done + i) % len(outs)].ptr for i in range(m)],
                                           inputs_ready=True, out_bit_packed=True)
                    done += m
                hp.synchronize()
                dts.append(time.perf_counter() - t0)
                fin.append(t1 - t0)
                depth = hp.info().get("pattern_max_weight")
                hp.close()
            for b in fl + outs:
                b.free()
            res[name][str(n)] = {"seconds": min(dts), "fresh_handle_s": min(fin), "batches": k, "shots_per_batch": B, "table_depth_at_end": depth}
            if verbose:
                print(f"{name} n={n:>11d}: {min(dts) * 1e3:8.2f} ms (handle {min(fin) * 1e3:6.2f} ms), {k} x {B}, table depth at the end {depth}, all {[round(d * 1e3, 2) for d in dts]}", flush=True)
        helper.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C4,C5")
    ap.add_argument("--ns", default="100000,1000000,10000000,100000000")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    from tsim_amd import backend, synth

    res = measure_time_to_n(backend, synth, a.configs.split(","), [int(x) for x in a.ns.split(",")], verbose=True)
    if a.json:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
