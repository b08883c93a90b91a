"""Where a wave of k_sample_wide spends its cycles (a library built with -DTSIMK_WIDE_TRACE:
scripts/build_variant.sh WORK scripts/_ab_wtrace.so -DTSIMK_WIDE_TRACE; TSIM_AMD_ALLOW_STALE=1 TSIM_AMD_LIB=scripts/_ab_wtrace.so).
usage: python scripts/wide_trace.py [config] [batches per call] [p_bit]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prog, cfg = synth.config_program(name)
p_bit = float(sys.argv[3]) if len(sys.argv) > 3 else cfg["p_bit"]
distinct = int(sys.argv[4]) if len(sys.argv) > 4 else nb
nf = cfg["num_f"]
hp = backend.HipProgram(prog)
B = 1_000_000
wf, rb = (nf + 63) // 64, (prog.num_outputs + 7) // 8
fs, outs = [], []
for i in range(nb):
    if i >= distinct:
        fs.append(fs[i % distinct]); outs.append(hp.malloc(B * rb + 16)); continue
    f = synth.synth_f(B, nf, p_bit, seed=1 + i)
    pk = np.packbits(f, axis=1, bitorder="little")
    pk = np.ascontiguousarray(np.pad(pk, ((0, 0), (0, wf * 8 - pk.shape[1]))))
    d = hp.malloc(pk.nbytes); hp.h2d(d, pk); fs.append(d)
    outs.append(hp.malloc(B * rb + 16))
ks = (C.c_uint32 * 2)(1, 2)
def call():
    hp.sample_steps_device([d.ptr for d in fs], B, nf, ks, [d.ptr for d in outs], inputs_ready=True, out_bit_packed=True)
for _ in range(6):
    call()
hp.synchronize()
lib = C.CDLL(os.environ["TSIM_AMD_LIB"]) if os.environ.get("TSIM_AMD_LIB") else None
buf = (C.c_ulonglong * 24)()
if lib is not None and hasattr(lib, "tsim_debug_wide_trace"):
    lib.tsim_debug_wide_trace(buf)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    call()
hp.synchronize()
dt = (time.perf_counter() - t0) / reps
if os.environ.get("WIDE_TRACE_BLOCKS"):
    for blk in range(int(os.environ["WIDE_TRACE_BLOCKS"])):
        t1 = time.perf_counter()
        for _ in range(10):
            call()
        hp.synchronize()
        print(f"  block {blk}: {(time.perf_counter() - t1) / 10 / nb * 1e6:.1f} us per batch; tables {hp.info()['pattern_max_weight']}")
ts = []
for _ in range(reps):
    hp.synchronize(); time.sleep(0.002)
    t1 = time.perf_counter(); call(); hp.synchronize(); ts.append(time.perf_counter() - t1)
print(f"  one call at a time (2 ms idle before each, host launch + synchronize included): median {sorted(ts)[len(ts)//2] * 1e6:.1f} us, min {min(ts) * 1e6:.1f} us")
print(f"{name} p_bit {p_bit}: {nb} batches of {B} per call: {dt * 1e6:.1f} us per call = {dt / nb * 1e6:.2f} us per batch = {nb * B / dt:.3e} shots/s; tables {hp.info()['pattern_max_weight']}")
if lib is not None and hasattr(lib, "tsim_debug_wide_trace"):
    lib.tsim_debug_wide_trace(buf)
    t = np.array(buf, dtype=np.float64)
    waves, chunks, passes = t[12], t[8], t[9]
    names = {0: "loop control / staging issue", 1: "wait for the chunk's rows (vmcnt 0)", 2: "direct outputs", 3: "set bits -> positions", 4: "rank, draws, table walk",
             5: "placement, store, queue push", 6: "dense passes (rest)", 7: "generic passes (heavy rows, normalisation check)",
             13: "dense: queue entry -> columns", 14: "dense: levels, Y_g and term sum", 15: "dense: |amp|, division, draw", 16: "dense: store"}
    print(f"  waves {waves / reps:.0f} per call, chunks per wave {chunks / waves:.1f}, dense passes per wave {passes / waves:.1f}; cycles per wave {t[11] / waves:.0f}")
    for k, nm in names.items():
        print(f"  {nm:52s} {t[k] / waves:10.0f} cycles per wave  {100 * t[k] / t[11]:5.1f} %   per chunk {t[k] / chunks:8.0f}" + (f"   per pass {t[k] / max(passes, 1):8.0f}" if k == 6 else ""))
