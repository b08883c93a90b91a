"""Recover the three 256-entry tables of numpy's ziggurat exponential sampler by black-box probing.

``Generator.geometric`` (p < 1/3) is ``ceil(-standard_exponential() / log1p(-p))`` and the reference's noise
sampler (src/tsim/noise/channels.py:641-655) draws its geometric gaps from it, so a native sampler that must
reproduce the numpy stream needs numpy's exact table constants: ``ke`` (uint64 acceptance thresholds), ``we``
(strip widths / 2^53) and ``fe`` (``exp(-x_i)``).  They were generated in extended precision upstream and cannot
be re-derived to the last bit from the published algorithm (Marsaglia & Tsang 2000; r = 7.69711747013104972), so
they are *measured* here from the installed numpy, which is a pinned dependency of the reference (uv.lock):

* PCG64's next outputs can be forced: for chosen 64-bit values v1, v2 pick states S1 = v1 (high half 0: the
  XSL-RR output is then the low half), S2 with output v2, and the increment inc = S2 - S1 * MULT; numpy accepts
  the state through ``bit_generator.state``.
* first-try acceptance (``ri < ke[idx]``) is visible in how far the state advanced (1 step) -> binary search
  gives ``ke[idx]`` exactly; an accepted draw returns ``ri * we[idx]`` -> ``we[idx]`` exactly (ri a power of two);
* the wedge test ``(fe[idx-1] - fe[idx]) * u + fe[idx] < exp(-x)`` (2 steps when it passes) is probed with
  x just below the strip edge and u = k * 2^-53: candidates for fe[idx] (a few ulps around exp(-we[idx] * 2^53))
  are eliminated until one is consistent with every observed decision, by induction from fe[0] = 1.

Output: tsim_amd/csrc/tsim_zig_tables.h.  tests/test_pcg_native.py re-probes the boundaries against the committed
tables, and tsim_amd.channels cross-checks the native stream against numpy at run time before trusting it.
"""
from __future__ import annotations

import math
import struct
import sys

import numpy as np

MULT = 0x2360ED051FC65DA44385DF649FCCF645
M128 = (1 << 128) - 1
M64 = (1 << 64) - 1
MINV = pow(MULT, -1, 1 << 128)


def step(state, inc):
    return (state * MULT + inc) & M128


def output(state):
    hi, lo = state >> 64, state & M64
    x, r = hi ^ lo, state >> 122
    return ((x >> r) | (x << ((-r) & 63))) & M64


class Forced:
    """A numpy Generator whose next two raw outputs are chosen by the caller."""

    def __init__(self):
        self.bg = np.random.PCG64(0)
        self.gen = np.random.Generator(self.bg)

    def arm(self, v1, v2=None):
        s1 = v1 & M64  # high half 0 -> rotation 0 -> output = low half
        if v2 is None:
            inc = 1
        else:
            s2 = v2 & M64
            inc = (s2 - s1 * MULT) & M128
            if not inc & 1:  # keep the increment odd: another preimage of v2 (high half 1 flips the low bit)
                s2 = (1 << 64) | ((v2 ^ 1) & M64)
                assert output(s2) == v2
                inc = (s2 - s1 * MULT) & M128
        s0 = ((s1 - inc) * MINV) & M128
        self.bg.state = {"bit_generator": "PCG64", "state": {"state": s0, "inc": inc}, "has_uint32": 0, "uinteger": 0}
        self.s0, self.inc = s0, inc

    def consumed(self, limit=64):
        cur, s = self.bg.state["state"]["state"], self.s0
        for k in range(limit + 1):
            if s == cur:
                return k
            s = step(s, self.inc)
        raise RuntimeError("state not found")


def probe_ke_we(fz):
    ke, we = [0] * 256, [0.0] * 256

    def first_try(idx, ri):
        fz.arm((ri << 11) | (idx << 3))
        x = fz.gen.standard_exponential()
        return fz.consumed() == 1, x

    for idx in range(256):
        if not first_try(idx, 0)[0]:
            continue  # ke[idx] == 0 (idx 1)
        lo, hi = 0, 1 << 53
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if first_try(idx, mid)[0]:
                lo = mid
            else:
                hi = mid
        ke[idx] = hi
        ri = 1 << (hi.bit_length() - 2)
        ok, x = first_try(idx, ri)
        assert ok
        we[idx] = x / ri
    return ke, we


def ulp_neighbours(v, n):
    out = [v]
    a = b = v
    for _ in range(n):
        a, b = math.nextafter(a, 0.0), math.nextafter(b, math.inf)
        out += [a, b]
    return out


def probe_we1(fz, fe1_guess):
    """idx 1 is never accepted on the first try (ke[1] = 0): read we[1] off a draw that passes the wedge test."""
    for ri in (1 << 20, 1 << 30, 1 << 40):
        fz.arm((ri << 11) | (1 << 3), 0)  # u = 0: passes iff fe[1] < exp(-x), true for small x
        x = fz.gen.standard_exponential()
        if fz.consumed() == 2:
            return x / ri
    raise RuntimeError("could not probe we[1]")


def probe_fe(fz, ke, we):
    fe = [0.0] * 256
    fe[0] = 1.0
    for idx in range(1, 256):
        X = we[idx] * 2.0**53
        cands = ulp_neighbours(math.exp(-X), 6)
        prev = fe[idx - 1]
        tried = 0
        # x close to the strip edge X: exp(-x) is only a few ulps above fe[idx]; sweep u = k * 2^-53
        for back in (1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 1025, 4097):
            ri = (1 << 53) - back
            if ri < ke[idx]:
                break
            x = ri * we[idx]
            e = math.exp(-x)
            # decisions change where fe + D*u crosses e: scan k geometrically, then refine around the flip
            def decide(k):
                fz.arm((ri << 11) | (idx << 3), k << 11)
                fz.gen.standard_exponential()
                return fz.consumed() == 2  # True: wedge test passed
            def model(c, k):
                return (prev - c) * (k * 2.0**-53) + c < e
            lo, hi = 0, (1 << 53) - 1
            if not decide(lo):
                ks = [0, 1, 2]
            elif decide(hi):
                ks = [hi, hi - 1]
            else:
                while hi - lo > 1:
                    mid = (lo + hi) // 2
                    if decide(mid):
                        lo = mid
                    else:
                        hi = mid
                ks = [max(0, lo - 2), max(0, lo - 1), lo, hi, hi + 1, hi + 2]
            for k in ks:
                k = min(k, (1 << 53) - 1)
                got = decide(k)
                cands = [c for c in cands if model(c, k) == got]
                tried += 1
            if len(cands) == 1 and tried >= 12:
                break
        if len(cands) != 1:
            raise RuntimeError(f"fe[{idx}]: {len(cands)} candidates left: {cands}")
        fe[idx] = cands[0]
    return fe


def main(out_path):
    fz = Forced()
    fz.arm(0x0123456789ABCDEF, 0xFEDCBA9876543210)
    raws = fz.bg.random_raw(2)
    assert int(raws[0]) == 0x0123456789ABCDEF and int(raws[1]) == 0xFEDCBA9876543210, "forcing PCG64 outputs failed"
    ke, we = probe_ke_we(fz)
    we[1] = probe_we1(fz, None)
    fe = probe_fe(fz, ke, we)
    r = we[255] * 2.0**53
    with open(out_path, "w") as fh:
        fh.write("// tsim_zig_tables.h - the ziggurat tables of numpy's standard_exponential (256 strips), MEASURED from numpy "
                 f"{np.__version__}\n// by scripts/numpy_ziggurat_tables.py (forced PCG64 states; see that file).  Do not edit.\n"
                 "#pragma once\n#include <stdint.h>\n\n")
        fh.write(f"static const double kZigExpR = {r!r};  // = we[255] * 2^53\n\n")
        fh.write("static const uint64_t kZigKe[256] = {\n")
        for i in range(0, 256, 4):
            fh.write("    " + ", ".join(f"0x{v:016X}ull" for v in ke[i:i + 4]) + ",\n")
        fh.write("};\n\nstatic const double kZigWe[256] = {\n")
        for i in range(0, 256, 4):
            fh.write("    " + ", ".join(float(v).hex() for v in we[i:i + 4]) + ",\n")
        fh.write("};\n\nstatic const double kZigFe[256] = {\n")
        for i in range(0, 256, 4):
            fh.write("    " + ", ".join(float(v).hex() for v in fe[i:i + 4]) + ",\n")
        fh.write("};\n")
    print("wrote", out_path, "r =", repr(r))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tsim_amd/csrc/tsim_zig_tables.h")
