import sys, cProfile, pstats, warnings
sys.path.insert(0, ".")
import numpy as np
from tsim_amd import synth
from tsim_amd.channels import error_probs
from tsim_amd.sampler import CompiledDetectorSampler
warnings.simplefilter("ignore")
prog, cfg = synth.config_program("C2")
probs = [error_probs(cfg["p_bit"])] * cfg["num_f"]
T = np.eye(cfg["num_f"], dtype=np.uint8)
s = CompiledDetectorSampler(prog, channel_probs=probs, error_transform=T, seed=1, noise="device")
s.sample(1_000_000, batch_size=1_000_000, bit_packed=True, append_observables=True)
pr = cProfile.Profile(); pr.enable()
s.sample(4_000_000, batch_size=1_000_000, bit_packed=True, append_observables=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
