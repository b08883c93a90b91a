"""Where the end-to-end device pipeline of CompiledDetectorSampler.sample(noise="device") spends its time."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import warnings
import numpy as np
from tsim_amd import synth, prng
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler

warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
nf = cfg["num_f"]
model = sys.argv[1] if len(sys.argv) > 1 else "p_bit"
if model == "p_bit":
    probs, T = [error_probs(cfg["p_bit"])] * nf, np.eye(nf, dtype=np.uint8)
else:
    rng = np.random.default_rng(7); n_mech = 20 * nf
    T = np.zeros((nf, n_mech), np.uint8); T[rng.integers(0, nf, size=n_mech), np.arange(n_mech)] = 1
    probs = [error_probs(1e-3)] * n_mech
s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=1, noise="device")
shots, batch = 4_000_000, 1_000_000
s.sample(shots, batch_size=batch, append_observables=True, bit_packed=True)
hp = s._hip()
noise = s._device_noise_sampler(hp)
d_f = hp.malloc(batch * 8); d_o = hp.malloc(shots * 8)
def T_(name, fn, n=5):
    hp.synchronize(); fn(); hp.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    hp.synchronize(); print(f"{name:44s} {(time.perf_counter()-t0)/n*1e3:8.3f} ms"); return r
k = prng.key(5)
T_("noise x4 (memset + k_noise, 1e6 shots each)", lambda: [noise.sample_into(d_f.ptr, batch, k) for _ in range(4)])
T_("sample_batch_device x4 (serial)", lambda: [hp.sample_batch_device(d_f.ptr, batch, nf, k, d_o.ptr) for _ in range(4)])
pg = np.empty((shots, 3), np.uint8)
T_("d2h 12 MB into pageable", lambda: hp.d2h(pg, d_o.ptr))
T_("np.empty 12 MB + d2h", lambda: hp.d2h(np.empty((shots, 3), np.uint8), d_o.ptr))
T_("whole sample(bit_packed)", lambda: s.sample(shots, batch_size=batch, append_observables=True, bit_packed=True), n=3)
T_("whole sample(bool)", lambda: s.sample(shots, batch_size=batch, append_observables=True), n=3)
