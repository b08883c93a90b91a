"""Soak test of tsim_sample_batch_device_begin/_end: random batch sizes, noise levels and slot orders,
several launches in flight; every result is compared with the full kernel on a second handle.

usage: fuzz_pipeline.py [iterations per config] [C2 C3 ... (default: C2 C4 C3 C5)] [log]   ("log" prints every launch)"""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, warnings
warnings.simplefilter("ignore")
from tsim_amd import backend as hip, synth, prng

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2026)
bad = 0
for cfg_name in ([a for a in sys.argv[2:] if a != "log"] or ("C2", "C4", "C3", "C5")):
    prog, cfg = synth.config_program(cfg_name)
    hp = hip.HipProgram(prog)                       # tables + pipelining + adaptive plan
    ref = hip.HipProgram(prog, pattern_tables=False)  # the full kernel, serial
    nf = cfg["num_f"]; wf, wo = (nf + 63) // 64, (prog.num_outputs + 63) // 64
    BMAX = 60000
    d_f = [hp.malloc(BMAX * wf * 8) for _ in range(hp.PIPELINE_SLOTS)]
    d_o = [hp.malloc(BMAX * wo * 8) for _ in range(hp.PIPELINE_SLOTS)]
    r_f, r_o = ref.malloc(BMAX * wf * 8), ref.malloc(BMAX * wo * 8)
    pending = {}
    def check(slot):
        global bad
        B, f_packed, key, off = pending.pop(slot)
        hp.sample_batch_device_end(slot); hp.synchronize()
        got = np.zeros((B, wo * 8), np.uint8); hp.d2h(got, d_o[slot])
        ref.h2d(r_f, f_packed); ref.sample_batch_device(r_f.ptr, B, nf, key, r_o.ptr, shot_offset=off); ref.synchronize()
        want = np.zeros((B, wo * 8), np.uint8); ref.d2h(want, r_o)
        if not np.array_equal(got, want):
            bad += 1; print("MISMATCH", cfg_name, B, slot, off, int((got != want).any(axis=1).sum()), "rows differ")
    for it in range(n_iter):
        slot = int(rng.integers(0, hp.PIPELINE_SLOTS))
        if slot in pending:
            check(slot)
        B = int(rng.choice([1, 63, 64, 65, 1000, 4097, 20000, 59999]))
        p = float(rng.choice([0.0, 0.005, 0.02, 0.06, 0.3]))
        f = synth.synth_f(B, nf, p, seed=int(rng.integers(0, 1 << 30)))
        fp = np.packbits(f, axis=1, bitorder="little")
        fp = np.ascontiguousarray(np.pad(fp, ((0, 0), (0, wf * 8 - fp.shape[1]))))
        key = prng.key(int(rng.integers(0, 1 << 40)))
        off = int(rng.choice([0, 0, 12345]))
        hp.h2d(d_f[slot], fp)
        hp.sample_batch_device_begin(slot, d_f[slot].ptr, B, nf, key, d_o[slot].ptr, shot_offset=off)
        if len(sys.argv) > 2 and sys.argv[-1] == "log":
            print("launch", it, "slot", slot, "B", B, "p", p, "off", off, "tables", hp.info()["pattern_max_weight"], flush=True)
        pending[slot] = (B, fp, key, off)
        if rng.random() < 0.3 and pending:
            check(int(rng.choice(list(pending))))
    for slot in list(pending):
        check(slot)
    print(cfg_name, "done")
print("iterations per config", n_iter, "mismatches", bad)
