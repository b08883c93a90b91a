"""Host-side cost of building a handle (pack + upload + pattern-table build) per shape."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tsim_amd import backend, synth
for name in sys.argv[1:] or ["C2", "C3", "C4", "C5"]:
    prog, cfg = synth.config_program(name)
    t0 = time.perf_counter(); hp = backend.HipProgram(prog); t1 = time.perf_counter()
    hp2 = backend.HipProgram(prog, pattern_tables=False); t2 = time.perf_counter()
    i = hp.info()
    print(f"{name}: HipProgram() {1e3*(t1-t0):8.1f} ms (tables off: {1e3*(t2-t1):8.1f} ms)  image {i['image_bytes']/1e6:.1f} MB, pattern tables {i['pattern_table_bytes']/1e6:.2f} MB, weights {i['pattern_max_weight']}")
