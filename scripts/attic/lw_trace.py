"""Timeline of the register first pass: s_memtime stamps of wave 0 of the first 16 blocks (a library built with
-DTSIMK_LW_TRACE: scripts/build_variant.sh WORK scripts/_ab_trace.so -DTSIMK_LW_TRACE; TSIM_AMD_LIB=scripts/_ab_trace.so)."""
import ctypes as C, os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import _lib, backend, synth
prog, cfg = synth.config_program("C2")
hp = backend.HipProgram(prog)
B = 1_000_000
f = synth.synth_f(B, 64, 0.02, seed=1)
pk = np.packbits(f, axis=1, bitorder="little")
d_f = hp.malloc(pk.nbytes); hp.h2d(d_f, pk)
d_o = hp.malloc(B * 8)
for _ in range(6):
    hp.sample_batch_device(d_f.ptr, B, 64, (1, 2), d_o.ptr)
hp.synchronize()
lib = C.CDLL(os.environ["TSIM_AMD_LIB"])
buf = (C.c_ulonglong * (16 * 32))()
lib.tsim_debug_lw_trace(buf)
t = np.array(buf, dtype=np.int64).reshape(16, 32)
names = ["top", "f in", "direct", "rank", "-", "3 draws", "3 bits", "2 bits", "stores", "append"]
for b in range(16):
    t0 = t[b, 24]
    line = [f"blk{b:2d} pre {t[b,25]-t0:5d}"]
    for it in range(2):
        base = it * 12
        prev = t[b, base + 0]
        if prev == 0: continue
        line.append(f"| it{it} top@{prev - t0:6d}")
        for k in (1, 2, 3, 5, 6, 7, 8, 9):
            v = t[b, base + k]
            if v == 0: continue
            line.append(f"{names[k]} +{v - prev}")
            prev = v
    line.append(f"| end@{t[b,26]-t0}")
    print(" ".join(line))
