"""First-pass kernel timing probe: serial launches of one program at a chosen num_f (row width) and noise level."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import backend, synth
name, num_f, p_bit = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
prog, cfg = synth.config_program(name)
hp = backend.HipProgram(prog)
B = 1_000_000
f = synth.synth_f(B, num_f, p_bit, seed=1)
wf = (num_f + 63) // 64
pk = np.zeros((B, wf * 8), np.uint8)
q = np.packbits(f, axis=1, bitorder="little"); pk[:, :q.shape[1]] = q
d_f = hp.malloc(pk.nbytes); hp.h2d(d_f, pk)
d_o = hp.malloc(B * 8 * ((prog.num_outputs + 63) // 64))
for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 5):
    hp.sample_batch_device(d_f.ptr, B, num_f, (1, 2), d_o.ptr)
hp.synchronize()
hp.profile_enable(1); hp.profile_read(reset=True)
for _ in range(10):
    hp.sample_batch_device(d_f.ptr, B, num_f, (1, 2), d_o.ptr)
hp.synchronize()
st = hp.profile_read_stages(); ms, n = hp.profile_read(reset=True)
print(name, "num_f", num_f, "p_bit", p_bit, {k: round(v / n * 1e3, 1) for k, v in st.items()}, "us per launch")
