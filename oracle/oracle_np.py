"""ORACLE (numpy) - TEST INFRASTRUCTURE ONLY, never shipped, never measured as product.

A CPU restatement, in plain numpy, of the reference's hot path
(`tsim.sampler.sample_program` and everything below it).  It deliberately
keeps the reference's *data movement*: one byte per bit, float32 GEMM ``% 2``
for the GF(2) contraction, materialised ``[B, G, T, 4]`` lookup tensors and
sequential exact-scalar scans - i.e. it shares no layout, packing or control
flow with the HIP kernels it checks.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

Reference anchors (all under /root/reference/src/tsim/):
  * sampler.py:28-167            _sample_component / sample_program
  * compile/evaluate.py:15-59    evaluate
  * compile/terms.py:20-187      phase tables + the four term families
  * core/exact_scalar.py:19-222  Z[omega]*2^k arithmetic
  * utils/linalg.py:81-102       matmul_gf2
  * jax.random (jax 0.6.2/0.9.2, threefry2x32, jax_threefry_partitionable=True)
    - third-party, absent from /root/reference; semantics restated from the
    published Threefry-2x32-20 algorithm (Salmon et al., SC'11) and JAX's
    documented partitionable key derivation, and pinned by the reference's own
    seeded count tests (tests/test_oracle_kats.py).

PARITY PINNED by: test/unit/test_sampler.py:223-233 (48,53,52,50),
test/integration/test_sampler_circuits.py:10-109 (48, 9, 48, 7/4/0),
test/unit/core/test_exact_scalar.py:67-84, test/unit/compile/test_terms.py
closed forms, test/unit/utils/test_linalg.py:102-135,
test/unit/compile/test_compile.py:31-46.

PARITY UNPINNED (no reference test constrains it; stated in DESIGN.md):
  * float32 ulp-level details of ``to_complex`` (FMA contraction), complex
    ``abs`` and ``2.0**power`` inside XLA;
  * the reduction order of the approximate-floatfactor branch
    (evaluate.py:56-59) - taken here as sequential in graph order;
  * the key layout for seeds >= 2**32.
"""

from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# Threefry-2x32 (20 rounds) and the jax.random subset the hot path uses
# --------------------------------------------------------------------------

_ROT_A = (13, 15, 26, 6)
_ROT_B = (17, 29, 16, 24)
_PARITY = np.uint32(0x1BD11BDA)


def _rotl(x: np.ndarray, r: int) -> np.ndarray:
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


def threefry2x32(k0, k1, c0, c1):
    """Threefry-2x32-20 block function on uint32 arrays (broadcasting)."""
    k0 = np.asarray(k0, dtype=np.uint32)
    k1 = np.asarray(k1, dtype=np.uint32)
    x0 = np.asarray(c0, dtype=np.uint32).copy()
    x1 = np.asarray(c1, dtype=np.uint32).copy()
    ks = (k0, k1, (k0 ^ k1 ^ _PARITY).astype(np.uint32))
    with np.errstate(over="ignore"):
        x0 = (x0 + ks[0]).astype(np.uint32)
        x1 = (x1 + ks[1]).astype(np.uint32)
        for blk in range(5):
            rots = _ROT_A if blk % 2 == 0 else _ROT_B
            for r in rots:
                x0 = (x0 + x1).astype(np.uint32)
                x1 = _rotl(x1, r)
                x1 = (x1 ^ x0).astype(np.uint32)
            x0 = (x0 + ks[(blk + 1) % 3]).astype(np.uint32)
            x1 = (x1 + ks[(blk + 2) % 3] + np.uint32(blk + 1)).astype(np.uint32)
    return x0, x1


def key(seed: int) -> tuple[int, int]:
    """``jax.random.key(seed)`` for the threefry2x32 impl: ``(seed >> 32, seed & 0xffffffff)``."""
    seed = int(seed)
    return ((seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF)


def split(k: tuple[int, int]) -> tuple[tuple[int, int], tuple[int, int]]:
    """``jax.random.split(key)`` (partitionable): child j = threefry(key, (0, j))."""
    x0, x1 = threefry2x32(k[0], k[1], np.zeros(2, np.uint32), np.arange(2, dtype=np.uint32))
    return (int(x0[0]), int(x1[0])), (int(x0[1]), int(x1[1]))


def random_bits32(k: tuple[int, int], n: int) -> np.ndarray:
    """``jax.random.bits(key, (n,), uint32)`` (partitionable): x0 ^ x1 of threefry(key, (0, s))."""
    x0, x1 = threefry2x32(k[0], k[1], np.zeros(n, np.uint32), np.arange(n, dtype=np.uint32))
    return (x0 ^ x1).astype(np.uint32)


def uniform01(k: tuple[int, int], n: int) -> np.ndarray:
    """``jax.random.uniform(key, (n,), float32)``: mantissa trick, in [0, 1)."""
    bits = random_bits32(k, n)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    return np.maximum(np.float32(0.0), f)


def bernoulli(k: tuple[int, int], p: np.ndarray) -> np.ndarray:
    """``jax.random.bernoulli(key, p)`` for a 1-D float32 ``p``: ``uniform < p``."""
    p = np.asarray(p, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        return uniform01(k, p.shape[0]) < p


# --------------------------------------------------------------------------
# utils/linalg.py:81-102
# --------------------------------------------------------------------------


def matmul_gf2(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``a[G,T,P] x b[B,P] -> uint8[B,G,T]`` via float32 GEMM then ``% 2``."""
    G, T, _ = a.shape
    if G * T == 0:
        return np.zeros((b.shape[0], G, T), dtype=np.uint8)
    s = b.astype(np.float32) @ a.astype(np.float32).reshape(G * T, -1).T
    return (s.reshape(-1, G, T) % 2).astype(np.uint8)


# --------------------------------------------------------------------------
# compile/terms.py:20-39 phase tables
# --------------------------------------------------------------------------

UNIT_PHASES = np.array(
    [
        [1, 0, 0, 0],
        [0, 1, 0, 0],
        [0, 0, 1, 0],
        [0, 0, 0, -1],
        [-1, 0, 0, 0],
        [0, -1, 0, 0],
        [0, 0, -1, 0],
        [0, 0, 0, 1],
    ],
    dtype=np.int32,
)
ONE_PLUS_PHASES = UNIT_PHASES.copy()
ONE_PLUS_PHASES[:, 0] += 1
IDENTITY = np.array([1, 0, 0, 0], dtype=np.int32)

# --------------------------------------------------------------------------
# core/exact_scalar.py - (coeffs int32[...,4], power int32[...])
# --------------------------------------------------------------------------


def scalar_mul(d1: np.ndarray, d2: np.ndarray) -> np.ndarray:
    """exact_scalar.py:19-39 (int32, silent wrap-around like XLA)."""
    d1 = d1.astype(np.int32)
    d2 = d2.astype(np.int32)
    a1, b1, c1, e1 = d1[..., 0], d1[..., 1], d1[..., 2], d1[..., 3]
    a2, b2, c2, e2 = d2[..., 0], d2[..., 1], d2[..., 2], d2[..., 3]
    with np.errstate(over="ignore"):
        A = a1 * a2 + b1 * e2 - c1 * c2 + e1 * b2
        Bc = a1 * b2 + b1 * a2 + c1 * e2 + e1 * c2
        C = a1 * c2 + b1 * b2 + c1 * a2 - e1 * e2
        D = a1 * e2 - b1 * c2 - c1 * b2 + e1 * a2
    return np.stack([A, Bc, C, D], axis=-1).astype(np.int32)


def reduce_step(power: np.ndarray, coeffs: np.ndarray):
    """exact_scalar.py:42-49: divide by 2 once where all four are even and not all zero."""
    red = np.all(coeffs % 2 == 0, axis=-1) & np.any(coeffs != 0, axis=-1)
    coeffs = np.where(red[..., None], coeffs // 2, coeffs).astype(np.int32)
    power = np.where(red, power + 1, power).astype(np.int32)
    return power, coeffs


def mul_with_power(x, y):
    """exact_scalar.py:52-71."""
    p1, c1 = x
    p2, c2 = y
    with np.errstate(over="ignore"):
        p = (p1 + p2).astype(np.int32)
    return reduce_step(p, scalar_mul(c1, c2))


def _shl_one(shift: np.ndarray) -> np.ndarray:
    """``jnp.left_shift(1, shift)`` on int32 with XLA semantics (shift >= 32 -> 0)."""
    shift = shift.astype(np.int64)
    ok = shift < 32
    val = np.left_shift(np.int64(1), np.where(ok, shift, 0))
    return np.where(ok, val, 0).astype(np.int64).astype(np.int32)  # bit 31 wraps to INT_MIN


def add_with_power(x, y):
    """exact_scalar.py:74-84."""
    p1, c1 = x
    p2, c2 = y
    s1 = _shl_one(np.maximum(p1 - p2, 0))[..., None]
    s2 = _shl_one(np.maximum(p2 - p1, 0))[..., None]
    p = np.minimum(p1, p2).astype(np.int32)
    with np.errstate(over="ignore"):
        c = (c1.astype(np.int32) * s1 + c2.astype(np.int32) * s2).astype(np.int32)
    return reduce_step(p, c)


def reduce_along_scan(power: np.ndarray, coeffs: np.ndarray, op, axis: int):
    """exact_scalar.py:98-137: sequential carry + final fix-point."""
    if axis < 0:
        axis += power.ndim
    power_t = np.moveaxis(power, axis, 0)
    coeffs_t = np.moveaxis(coeffs, axis, 0)
    carry = (power_t[0].astype(np.int32), coeffs_t[0].astype(np.int32))
    for i in range(1, power_t.shape[0]):
        carry = op(carry, (power_t[i].astype(np.int32), coeffs_t[i].astype(np.int32)))
    p, c = carry
    while True:
        new_p, new_c = reduce_step(p, c)
        changed = bool(np.any(new_p != p))
        p, c = new_p, new_c
        if not changed:
            break
    return p, c


class ExactScalarArray:
    """exact_scalar.py:140-222."""

    def __init__(self, coeffs: np.ndarray, power: np.ndarray | None = None):
        self.coeffs = np.asarray(coeffs, dtype=np.int32)
        if power is None:
            self.power = np.zeros(self.coeffs.shape[:-1], dtype=np.int32)
        else:
            self.power = np.asarray(power, dtype=np.int32)

    def __mul__(self, other: "ExactScalarArray") -> "ExactScalarArray":
        c1, c2 = np.broadcast_arrays(self.coeffs, other.coeffs)
        p1, p2 = np.broadcast_arrays(self.power, other.power)
        with np.errstate(over="ignore"):
            return ExactScalarArray(scalar_mul(c1, c2), (p1 + p2).astype(np.int32))

    def sum(self, axis: int = -1) -> "ExactScalarArray":
        p, c = reduce_along_scan(self.power, self.coeffs, add_with_power, axis)
        return ExactScalarArray(c, p)

    def prod(self, axis: int = -1) -> "ExactScalarArray":
        if axis < 0:
            axis += self.power.ndim
        if self.coeffs.shape[axis] == 0:
            shape = self.coeffs.shape[:axis] + self.coeffs.shape[axis + 1 :]
            c = np.zeros(shape, dtype=np.int32)
            c[..., 0] = 1
            return ExactScalarArray(c)
        p, c = reduce_along_scan(self.power, self.coeffs, mul_with_power, axis)
        return ExactScalarArray(c, p)

    def to_complex(self) -> np.ndarray:
        return to_complex(self.coeffs, self.power)


# float32 constants of exact_scalar.py:15-16 (exp(+-i*pi/4) evaluated in complex64)
E4_RE = np.float32(0.70710677)  # 0x3F3504F3
E4_IM = np.float32(0.70710677)


def pow2_f32(power: np.ndarray) -> np.ndarray:
    """``jnp.pow(2.0, power)`` as an exact float32 power of two (gradual underflow)."""
    with np.errstate(over="ignore", under="ignore"):
        return np.ldexp(np.float32(1.0), np.asarray(power, dtype=np.int32)).astype(np.float32)


def to_complex(coeffs: np.ndarray, power: np.ndarray) -> np.ndarray:
    """exact_scalar.py:87-89,218-222 in float32, one rounding per operation (no FMA)."""
    a = coeffs[..., 0].astype(np.float32)
    b = coeffs[..., 1].astype(np.float32)
    c = coeffs[..., 2].astype(np.float32)
    d = coeffs[..., 3].astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        re = (a + b * E4_RE) + d * E4_RE
        im = (b * E4_IM + c) + d * (-E4_IM)
        s = pow2_f32(power)
        out = np.empty(re.shape, dtype=np.complex64)
        out.real = re * s
        out.imag = im * s
    return out


def complex_abs(z: np.ndarray) -> np.ndarray:
    """``jnp.abs(complex64)`` -> float32: ``max * sqrt(1 + (min/max)^2)`` (0 when max == 0)."""
    re = np.abs(z.real.astype(np.float32))
    im = np.abs(z.imag.astype(np.float32))
    mx = np.maximum(re, im)
    mn = np.minimum(re, im)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        r = mn / mx
        out = mx * np.sqrt(np.float32(1.0) + r * r)
    out = np.where(mx == 0, np.float32(0.0), out)
    out = np.where(np.isinf(mx), np.float32(np.inf), out)
    out = np.where(np.isnan(re) | np.isnan(im), np.float32(np.nan), out)
    return out.astype(np.float32)


# --------------------------------------------------------------------------
# compile/terms.py:42-187 - the four families
# --------------------------------------------------------------------------


def node_phases_evaluate(np_, param_vals: np.ndarray) -> ExactScalarArray:
    """terms.py:56-73."""
    phases = np.asarray(np_.phases)
    rowsum = matmul_gf2(np.asarray(np_.params), param_vals)
    idx = (4 * rowsum.astype(np.int64) + phases) % 8
    term_vals = ONE_PLUS_PHASES[idx]
    mask = np.arange(phases.shape[1])[None, :] < np.asarray(np_.counts)[:, None]
    term_vals = np.where(mask[..., None], term_vals, IDENTITY)
    return ExactScalarArray(term_vals).prod(axis=-1)


def halfpi_phases_evaluate(hp, param_vals: np.ndarray) -> ExactScalarArray:
    """terms.py:94-107."""
    rowsum = matmul_gf2(np.asarray(hp.params), param_vals)
    idx = (rowsum.astype(np.uint8) * np.asarray(hp.coeffs).astype(np.uint8)) % 8
    total = np.sum(idx.astype(np.int64), axis=-1) % 8
    return ExactScalarArray(UNIT_PHASES[total])


def pi_products_evaluate(pp, param_vals: np.ndarray) -> ExactScalarArray:
    """terms.py:125-144."""
    psi = (np.asarray(pp.psi_const) + matmul_gf2(np.asarray(pp.psi_params), param_vals)) % 2
    phi = (np.asarray(pp.phi_const) + matmul_gf2(np.asarray(pp.phi_params), param_vals)) % 2
    exponent = (psi * phi) % 2
    s = np.sum(exponent.astype(np.int64), axis=-1) % 2
    return ExactScalarArray(((1 - 2 * s)[..., None] * IDENTITY).astype(np.int32))


def phase_pairs_evaluate(pp, param_vals: np.ndarray) -> ExactScalarArray:
    """terms.py:164-187."""
    ra = matmul_gf2(np.asarray(pp.alpha_params), param_vals).astype(np.int64)
    rb = matmul_gf2(np.asarray(pp.beta_params), param_vals).astype(np.int64)
    alpha = (np.asarray(pp.alpha) + ra * 4) % 8
    beta = (np.asarray(pp.beta) + rb * 4) % 8
    gamma = (alpha + beta) % 8
    term_vals = IDENTITY + UNIT_PHASES[alpha] + UNIT_PHASES[beta] - UNIT_PHASES[gamma]
    mask = np.arange(np.asarray(pp.alpha).shape[1])[None, :] < np.asarray(pp.counts)[:, None]
    term_vals = np.where(mask[..., None], term_vals, IDENTITY)
    return ExactScalarArray(term_vals.astype(np.int32)).prod(axis=-1)


# --------------------------------------------------------------------------
# compile/evaluate.py:15-59
# --------------------------------------------------------------------------


def evaluate_exact(circuit, param_vals: np.ndarray):
    """The exact branch of ``evaluate`` up to (but excluding) ``to_complex``.

    Returns ``(coeffs int32[B,4], power int32[B])`` of the summed amplitude, or
    ``None`` for an empty circuit.
    """
    pre = circuit.prefactor
    if np.asarray(pre.phase_indices).shape[0] == 0:
        return None
    total = _family_product(circuit, param_vals)
    with np.errstate(over="ignore"):
        total = ExactScalarArray(
            total.coeffs, (total.power + np.asarray(pre.power2, dtype=np.int32)).astype(np.int32)
        )
    s = total.sum()
    return s.coeffs, s.power


def _family_product(circuit, param_vals: np.ndarray) -> ExactScalarArray:
    pre = circuit.prefactor
    static = ExactScalarArray(UNIT_PHASES[np.asarray(pre.phase_indices)])
    ff = ExactScalarArray(np.asarray(pre.floatfactor, dtype=np.int32))
    total = node_phases_evaluate(circuit.node_phases, param_vals)
    for fac in (
        halfpi_phases_evaluate(circuit.halfpi_phases, param_vals),
        pi_products_evaluate(circuit.pi_products, param_vals),
        phase_pairs_evaluate(circuit.phase_pairs, param_vals),
        static,
        ff,
    ):
        total = total * fac
    return total


def evaluate(circuit, param_vals: np.ndarray) -> np.ndarray:
    """``evaluate(circuit, param_vals) -> complex64[B]`` (evaluate.py:15-59)."""
    param_vals = np.asarray(param_vals).astype(np.uint8)
    pre = circuit.prefactor
    B = param_vals.shape[0]
    if np.asarray(pre.phase_indices).shape[0] == 0:
        return np.zeros(B, dtype=np.complex64)
    if not pre.has_approximate_floatfactors:
        coeffs, power = evaluate_exact(circuit, param_vals)
        return to_complex(coeffs, power)
    total = _family_product(circuit, param_vals)
    z = total.to_complex()  # [B, G]
    approx = np.asarray(pre.approximate_floatfactors, dtype=np.complex64)
    scale = pow2_f32(np.asarray(pre.power2))
    acc_re = np.zeros(B, dtype=np.float32)
    acc_im = np.zeros(B, dtype=np.float32)
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        for g in range(z.shape[1]):  # sequential graph order (unpinned in the reference)
            zr, zi = z[:, g].real, z[:, g].imag
            ar, ai = approx[g].real, approx[g].imag
            tr = zr * ar - zi * ai
            ti = zr * ai + zi * ar
            acc_re = acc_re + tr * scale[g]
            acc_im = acc_im + ti * scale[g]
    out = np.empty(B, dtype=np.complex64)
    out.real, out.imag = acc_re, acc_im
    return out


# --------------------------------------------------------------------------
# sampler.py:28-167
# --------------------------------------------------------------------------


def sample_component(component, f_params: np.ndarray, k: tuple[int, int]):
    """sampler.py:28-81 -> (samples bool[B,n], next_key, max_norm_deviation float32)."""
    B = f_params.shape[0]
    levels = component.compiled_scalar_graphs
    n = len(levels) - 1
    f_sel = f_params[:, np.asarray(component.f_selection, dtype=np.int64)].astype(np.bool_)
    m = np.zeros((B, n), dtype=np.bool_)
    prev = complex_abs(evaluate(levels[0], f_sel))
    ones = np.ones((B, 1), dtype=np.bool_)
    zero = np.zeros((1, 1), dtype=np.bool_)
    max_dev = np.float32(0.0)
    for i, circuit in enumerate(levels[1:]):
        params = np.hstack([f_sel, m[:, :i], ones])
        check = np.hstack([f_sel[:1], m[:1, :i], zero])
        probs = complex_abs(evaluate(circuit, np.vstack([params, check])))
        p1 = probs[:B]
        p0 = probs[-1]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            norm = np.float32(np.float32(p0 + p1[0]) / prev[0])
            dev = np.abs(np.float32(norm - np.float32(1.0)))
            # jnp.maximum propagates NaN
            if np.isnan(dev) or np.isnan(max_dev):
                max_dev = np.float32(np.nan)
            else:
                max_dev = np.float32(max(max_dev, dev))
            k, sub = split(k)
            bits = bernoulli(sub, (p1 / prev).astype(np.float32))
            m[:, i] = bits
            prev = np.where(bits, p1, (prev - p1).astype(np.float32)).astype(np.float32)
    return m, k, max_dev


def sample_program(program, f_params: np.ndarray, k: tuple[int, int], return_devs: bool = False):
    """sampler.py:117-167 (the error/warning policy is left to the caller)."""
    f_params = np.asarray(f_params)
    B = f_params.shape[0]
    results = []
    devs = []
    if program.num_outputs == 0:
        out = np.zeros((B, 0), dtype=np.bool_)
        return (out, devs) if return_devs else out
    dfi = np.asarray(program.direct_f_indices, dtype=np.int64)
    if len(dfi) > 0:
        results.append(f_params[:, dfi].astype(np.bool_) ^ np.asarray(program.direct_flips, dtype=np.bool_))
    for comp in program.components:
        s, k, dev = sample_component(comp, f_params, k)
        devs.append(dev)
        results.append(s)
    combined = np.concatenate(results, axis=1)
    if program.output_reindex is not None:
        combined = combined[:, np.asarray(program.output_reindex, dtype=np.int64)]
    return (combined, devs) if return_devs else combined
