"""Where the end-to-end device pipeline of CompiledDetectorSampler.sample() spends its time."""
import sys, time
sys.path.insert(0, ".")
import warnings
import numpy as np
from tsim_amd import synth, prng
from tsim_amd.backend import alloc_pinned_numpy
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler

warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
probs = [error_probs(cfg["p_bit"])] * cfg["num_f"]
T = np.eye(cfg["num_f"], dtype=np.uint8)
s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=1, noise="device")
shots, batch = 4_000_000, 1_000_000
s.sample(shots, batch_size=batch)
st = s._device_state; hp = st["hp"]; b = st["bufs"]
def T_(name, fn, n=3):
    hp.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    hp.synchronize(); print(f"{name:34s} {(time.perf_counter()-t0)/n*1e3:8.2f} ms"); return r
k = prng.key(5)
T_("noise x4 (device)", lambda: [st["noise"].sample_into(b["f"].ptr, batch, k) for _ in range(4)])
T_("sample x4", lambda: [hp.sample_batch_device(b["f"].ptr, batch, 64, k, b["out"].ptr) for _ in range(4)])
T_("unpack 4e6 x 20", lambda: hp.unpack_bits_device(b["out"].ptr, shots, 20, b["u8"].ptr))
res = T_("alloc_pinned 80 MB", lambda: alloc_pinned_numpy(shots * 20, np.uint8, (shots, 20)))
T_("d2h 80 MB into pinned", lambda: hp.d2h(res, b["u8"]))
pg = np.empty((shots, 20), np.uint8)
T_("d2h 80 MB into pageable", lambda: hp.d2h(pg, b["u8"]))
T_("np.copy 80 MB", lambda: res.copy())
T_("whole sample()", lambda: s.sample(shots, batch_size=batch, append_observables=True), n=2)
