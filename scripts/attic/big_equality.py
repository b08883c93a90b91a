"""Full-size equality: the pipelined two-pass path vs the full kernel on every row, 5e7 shots each,
same seeds, device noise, bit-packed outputs - the byte streams must be identical."""
import hashlib, os, sys, time, warnings
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
shots, batch = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000, 2_000_000
digests = []
for tables in ("1", "0"):
    os.environ["TSIM_AMD_PATTERN_TABLES"] = tables
    for name in ("C2", "C3", "C4"):
        prog, cfg = synth.config_program(name)
        probs = [error_probs(cfg["p_bit"])] * cfg["num_f"]
        T = np.eye(cfg["num_f"], dtype=np.uint8)
        s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=123, noise="device")
        n = shots if name != "C4" else shots // 10
        t0 = time.perf_counter()
        out = s.sample(n, batch_size=batch, bit_packed=True, append_observables=True)
        dt = time.perf_counter() - t0
        h = hashlib.sha256(out.tobytes()).hexdigest()[:16]
        digests.append((name, tables, h))
        print(f"{name} tables={tables}: {n:.1e} shots in {dt:.2f} s ({n/dt:.2e}/s), ones={int(np.unpackbits(out).sum())}, sha={h}")
ok = all(digests[i][2] == digests[i + 3][2] for i in range(3))
print("IDENTICAL" if ok else "MISMATCH")
