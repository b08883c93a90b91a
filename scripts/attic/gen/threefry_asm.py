"""Generate the hand-scheduled Threefry-2x32-20 blocks of tsim_amd/csrc/tsim_threefry.hip.h.

`python scripts/gen/threefry_asm.py N` prints a __device__ function that advances N draws (N key pairs, ONE counter -
the shot index) together, instruction by instruction, so that the dependent add -> rotate -> xor chain of one draw
is covered by the other draws' instructions inside the same wave.  Same arithmetic as threefry2x32 in
tsim_kernels.hip.h (jax.random's Threefry, /root/reference/src/tsim/sampler.py:74-75 draws through it): the
generator only fixes instruction selection and order.
"""
import sys

ROT = [[13, 15, 26, 6], [17, 29, 16, 24]]


def emit(n: int, name: str) -> str:
    L = []
    a = L.append
    args = ", ".join(f"uint32_t k0_{i}, uint32_t k1_{i}" for i in range(n))
    outs = ", ".join(f"uint32_t &b{i}" for i in range(n))
    a(f"__device__ __forceinline__ void {name}({args}, unsigned long long s, {outs}) {{")
    a("  const uint32_t lo = (uint32_t)s, hi = (uint32_t)(s >> 32);")
    for i in range(n):
        a(f"  const uint32_t k2_{i} = k0_{i} ^ k1_{i} ^ 0x1BD11BDAu;")
    a("  uint32_t " + ", ".join(f"x0_{i}, x1_{i}" for i in range(n)) + ", t;")
    asm = []

    def each(fmt):
        for i in range(n):
            asm.append(fmt.replace("#", str(i)))

    # x1 = lo + k1 ; x0 = hi + k0 + x1 (round 1's add folded)
    each("v_add_u32 %[x1_#], %[k1_#], %[lo]")
    each("v_add3_u32 %[x0_#], %[hi], %[x1_#], %[k0_#]")
    ks = ["k0", "k1", "k2"]
    for g in range(5):
        rots = ROT[g & 1]
        for r_i, r in enumerate(rots):
            if r_i > 0:
                each("v_add_u32 %[x0_#], %[x0_#], %[x1_#]")
            each(f"v_alignbit_b32 %[x1_#], %[x1_#], %[x1_#], {32 - r}")
            each("v_xor_b32 %[x1_#], %[x1_#], %[x0_#]")
        ka, kb = ks[(g + 1) % 3], ks[(g + 2) % 3]
        if g < 4:
            # x1 += kb + (g+1) [scalar sum in t]; x0 = x0 + x1 + ka (next round's add folded)
            for i in range(n):
                asm.append(f"s_add_i32 %[t], %[{kb}_{i}], {g + 1}")
                asm.append(f"v_add_u32 %[x1_{i}], %[t], %[x1_{i}]")
            each(f"v_add3_u32 %[x0_#], %[x0_#], %[x1_#], %[{ka}_#]")
        else:
            for i in range(n):
                asm.append(f"s_add_i32 %[t], %[{kb}_{i}], {g + 1}")
                asm.append(f"v_add_u32 %[x1_{i}], %[t], %[x1_{i}]")
            each(f"v_add_u32 %[x0_#], %[{ka}_#], %[x0_#]")
            each("v_xor_b32 %[x0_#], %[x0_#], %[x1_#]")
    a('  asm(' + "\n      ".join(f'"{x}\\n"' for x in asm))
    a("      : " + ", ".join(f'[x0_{i}] "=&v"(x0_{i}), [x1_{i}] "=&v"(x1_{i})' for i in range(n)) + ', [t] "=&s"(t)')
    a('      : [lo] "v"(lo), [hi] "v"(hi), ' + ", ".join(f'[k0_{i}] "s"(k0_{i}), [k1_{i}] "s"(k1_{i}), [k2_{i}] "s"(k2_{i})' for i in range(n)))
    a('      : "scc");')
    for i in range(n):
        a(f"  b{i} = x0_{i};")
    a("}")
    return "\n".join(L)


if __name__ == "__main__":
    n = int(sys.argv[1])
    print(emit(n, sys.argv[2] if len(sys.argv) > 2 else f"threefry_bits32_x{n}"))
