import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tsim_amd.channels import ChannelSampler
nf = 64
probs = [np.array([0.98, 0.02]) for _ in range(nf)]
et = np.eye(nf, dtype=np.uint8)
cs = ChannelSampler(probs, et, seed=5)
print("native:", cs._native is not None)
out = np.empty((1_000_000, 1), np.uint64)
for r in range(4):
    t = time.perf_counter(); rows = cs.sample_packed(1_000_000, out=out); dt = time.perf_counter() - t
    print(f"sample_packed 1e6 rows: {dt*1e3:.2f} ms   checksum {int(np.bitwise_xor.reduce(rows.ravel())):x}")
