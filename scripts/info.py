import sys; sys.path.insert(0, '.')
from tsim_amd import backend, synth
for name in sys.argv[1:] or ["C2"]:
    prog, cfg = synth.config_program(name)
    print(name, backend.HipProgram(prog).info())
