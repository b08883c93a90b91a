"""Fused first pass alone: 8 batches per launch, serial launches, HIP-event time per batch (scripts/build_variant.sh
variants with -DTSIMK_LWM_SKIP=mask leave parts of the pass out: 1 Threefry, 2 direct outputs, 4 rank, 8 stores)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
prog, cfg = synth.config_program(name)
hp = backend.HipProgram(prog)
B, nf = 1_000_000, cfg["num_f"]
wf = (nf + 63) // 64
bufs = []
for k in range(4):
    f = synth.synth_f(B, nf, cfg["p_bit"], seed=k)
    pk = np.packbits(f, axis=1, bitorder="little")
    pk = np.ascontiguousarray(np.pad(pk, ((0, 0), (0, wf * 8 - pk.shape[1]))))
    d = hp.malloc(pk.nbytes); hp.h2d(d, pk); bufs.append(d)
outs = [hp.malloc(B * 8) for _ in range(16)]
ks = (C.c_uint32 * 2)(1, 2)
n = 8
fa = [bufs[i % 4].ptr for i in range(n)]
for rep in range(4):  # feedback + warm-up
    hp.sample_steps_device(fa, B, nf, ks, [outs[(rep * n + i) % 16].ptr for i in range(n)], inputs_ready=True, out_bit_packed=True)
    hp.synchronize()
hp.profile_set_sampling(1); hp.profile_enable(2); hp.profile_read(reset=True); hp.profile_read_steps()
for rep in range(10):
    hp.sample_steps_device(fa, B, nf, ks, [outs[(rep * n + i) % 16].ptr for i in range(n)], inputs_ready=True, out_bit_packed=True)
    hp.synchronize()
st = hp.profile_read_stages(); ms, ln = hp.profile_read(reset=True); steps = hp.profile_read_steps()
print(name, "fused first pass: %.2f us per launch of %.1f batches = %.2f us per 1e6-shot batch" % (st["pattern_pass"] / ln * 1e3, steps / ln, st["pattern_pass"] / steps * 1e3))
# the serial API (tsim_sample_batch_device: one batch per call, first pass + hard rows on the handle's stream)
for rep in range(12):
    hp.sample_batch_device(bufs[rep % 4].ptr, B, nf, (3, 4 + rep), outs[rep % 16].ptr)
    hp.synchronize()
