"""Clifford-only front-end (tsim_amd/clifford.py): Pauli-frame analysis checked against an explicit
stabilizer-tableau replay of every single error, the reference's seeded known-answer test, and the
GF(2) basis conventions of the reference (find_basis / transform_error_basis / classify_direct)."""

import numpy as np
import pytest

from tsim_amd.clifford import CliffordCircuit, find_basis, pauli_channel_2_probs


def test_find_basis_reconstructs_rows_and_is_greedy():
    rng = np.random.default_rng(0)
    for _ in range(50):
        n, d = int(rng.integers(1, 12)), int(rng.integers(1, 10))
        rows = [int(rng.integers(0, 2**d)) for _ in range(n)]
        basis_idx, combos = find_basis(rows)
        # reconstruction: row i = XOR of basis rows selected by combos[i]
        for i, r in enumerate(rows):
            acc = 0
            for pos, bi in enumerate(basis_idx):
                if (combos[i] >> pos) & 1:
                    acc ^= rows[bi]
            assert acc == r
        # greedy: a row is in the basis iff it is independent of the rows before it
        mat = np.array([[(r >> k) & 1 for k in range(d)] for r in rows], dtype=np.uint8)

        def rank(m):
            m = m.copy() % 2
            rk = 0
            for c in range(m.shape[1]):
                piv = [i for i in range(rk, m.shape[0]) if m[i, c]]
                if not piv:
                    continue
                m[[rk, piv[0]]] = m[[piv[0], rk]]
                for i in range(m.shape[0]):
                    if i != rk and m[i, c]:
                        m[i] ^= m[rk]
                rk += 1
            return rk

        expect = [i for i in range(n) if rank(mat[: i + 1]) > (rank(mat[:i]) if i else 0)]
        assert basis_idx == expect


def test_pauli_channel_2_layout():
    """index = z_i + 2 x_i + 4 z_j + 8 x_j (reference noise/channels.py:114-167)."""
    args = [0.001 * (k + 1) for k in range(15)]  # IX, IY, IZ, XI, ... ZZ
    t = pauli_channel_2_probs(*args)
    assert t.shape == (16,) and abs(t.sum() - 1) < 1e-12
    name = dict(zip(("IX", "IY", "IZ", "XI", "XX", "XY", "XZ", "YI", "YX", "YY", "YZ", "ZI", "ZX", "ZY", "ZZ"), args))
    assert t[1] == name["ZI"] and t[2] == name["XI"] and t[3] == name["YI"]
    assert t[4] == name["IZ"] and t[8] == name["IX"] and t[12] == name["IY"]
    assert t[5] == name["ZZ"] and t[10] == name["XX"] and t[15] == name["YY"] and t[6] == name["XZ"]


def test_reference_kat_bell_pair_with_x_error():
    """test/integration/test_sampler_circuits.py:25-37 of the reference: 4 of 10, rows 1, 6, 8, 9."""
    c = CliffordCircuit("""
        R 0 1
        H 0
        CNOT 0 1
        X_ERROR(0.3) 0
        M 0 1
        DETECTOR rec[-1] rec[-2]
        """)
    program, probs, et = c.compile()
    assert program.num_outputs == 1 and program.num_detectors == 1 and not program.components
    assert et.tolist() == [[1]] and np.allclose(probs[0], [0.7, 0.3])
    d = c.compile_detector_sampler(seed=1).sample(10)
    assert d.shape == (10, 1) and np.count_nonzero(d) == 4
    assert np.nonzero(d[:, 0])[0].tolist() == [1, 6, 8, 9]


_INVERSE = {"H": "H", "X": "X", "Y": "Y", "Z": "Z", "S": "S_DAG", "S_DAG": "S", "SQRT_X": "SQRT_X_DAG",
            "SQRT_X_DAG": "SQRT_X", "SQRT_Y": "SQRT_Y_DAG", "SQRT_Y_DAG": "SQRT_Y", "H_YZ": "H_YZ",
            "H_XY": "H_XY", "H_NXY": "H_NXY", "H_NXZ": "H_NXZ", "H_NYZ": "H_NYZ",
            "C_XYZ": "C_ZYX", "C_ZYX": "C_XYZ", "C_NXYZ": "C_ZYNX", "C_ZYNX": "C_NXYZ",
            "C_XNYZ": "C_ZNYX", "C_ZNYX": "C_XNYZ", "C_XYNZ": "C_NZYX", "C_NZYX": "C_XYNZ",
            "CX": "CX", "CZ": "CZ", "CY": "CY", "SWAP": "SWAP", "XCZ": "XCZ",
            "XCX": "XCX", "XCY": "XCY", "YCX": "YCX", "YCY": "YCY", "YCZ": "YCZ",
            "ISWAP": "ISWAP_DAG", "ISWAP_DAG": "ISWAP", "SQRT_XX": "SQRT_XX_DAG", "SQRT_XX_DAG": "SQRT_XX",
            "SQRT_YY": "SQRT_YY_DAG", "SQRT_YY_DAG": "SQRT_YY", "SQRT_ZZ": "SQRT_ZZ_DAG", "SQRT_ZZ_DAG": "SQRT_ZZ",
            "CZSWAP": "CZSWAP"}
_TWO = {"CX", "CZ", "CY", "SWAP", "XCZ", "XCX", "XCY", "YCX", "YCY", "YCZ", "ISWAP", "ISWAP_DAG", "SQRT_XX",
        "SQRT_XX_DAG", "SQRT_YY", "SQRT_YY_DAG", "SQRT_ZZ", "SQRT_ZZ_DAG", "CZSWAP"}


def _random_echo_circuit(rng, n, depth):
    """U, then U^-1, then measure everything: every measurement is deterministic (0) without noise.
    Returns the gate list; noise is inserted by the caller."""
    gates = []
    for _ in range(depth):
        g = str(rng.choice(list(_INVERSE)))
        if g in _TWO:
            a, b = rng.choice(n, size=2, replace=False)
            gates.append((g, (int(a), int(b))))
        else:
            gates.append((g, (int(rng.integers(0, n)),)))
    inv = [(_INVERSE[g], q) for g, q in reversed(gates)]
    return gates + inv


@pytest.mark.parametrize("seed", range(12))
def test_frame_propagation_matches_explicit_pauli_replay(seed):
    """For every error bit: replace the noise channel by the explicit Pauli gate and replay the
    noiseless tableau; the detectors that flip must be exactly those whose error set holds the bit."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 6))
    gates = _random_echo_circuit(rng, n, int(rng.integers(4, 14)))
    # noise sites: (position in the gate list, kind, qubits)
    sites = []
    for _ in range(int(rng.integers(1, 6))):
        kind = str(rng.choice(["X_ERROR", "Y_ERROR", "Z_ERROR", "DEPOLARIZE1", "DEPOLARIZE2"]))
        pos = int(rng.integers(0, len(gates) + 1))
        qs = tuple(int(q) for q in rng.choice(n, size=2 if kind == "DEPOLARIZE2" else 1, replace=False))
        sites.append((pos, kind, qs))
    sites.sort(key=lambda s: s[0])
    basis = str(rng.choice(["M", "MX", "MR"]))

    def text(explicit=None):
        """explicit = (site index, [pauli per qubit]) replaces that site by Pauli gates."""
        lines, si = [f"R {' '.join(map(str, range(n)))}"], 0
        if basis == "MX":
            lines.append(f"H {' '.join(map(str, range(n)))}")
        for pos in range(len(gates) + 1):
            while si < len(sites) and sites[si][0] == pos:
                _, kind, qs = sites[si]
                if explicit is None:
                    lines.append(f"{kind}(0.01) {' '.join(map(str, qs))}")
                elif explicit[0] == si:
                    for q, pl in zip(qs, explicit[1]):
                        if pl != "I":
                            lines.append(f"{pl} {q}")
                si += 1
            if pos < len(gates):
                g, q = gates[pos]
                lines.append(f"{g} {' '.join(map(str, q))}")
        lines.append(f"{basis} {' '.join(map(str, range(n)))}")
        for k in range(n):
            lines.append(f"DETECTOR rec[-{k + 1}]")
        lines.append(f"OBSERVABLE_INCLUDE(0) rec[-1] rec[-{n}]")
        return "\n".join(lines)

    an = CliffordCircuit(text()).analyze()
    base = [v for _, v in an.detectors] + [an.observables[0][1]]
    sets = [s for s, _ in an.detectors] + [an.observables[0][0]]
    bit = 0
    for si, (_, kind, qs) in enumerate(sites):
        if kind == "DEPOLARIZE1":
            variants = [("Z",), ("X",)]          # bit order: Z component, X component
        elif kind == "DEPOLARIZE2":
            variants = [("Z", "I"), ("X", "I"), ("I", "Z"), ("I", "X")]
        else:
            variants = [(kind[0],)]
        for paulis in variants:
            rep = CliffordCircuit(text((si, paulis))).analyze()
            got = [v for _, v in rep.detectors] + [rep.observables[0][1]]
            want = [b ^ ((s >> bit) & 1) for b, s in zip(base, sets)]
            assert got == want, (kind, paulis, bit)
            bit += 1
    assert bit == an.num_e


def _repetition_code(distance, rounds, p_data, p_meas):
    data = list(range(0, 2 * distance, 2))
    anc = list(range(1, 2 * distance - 1, 2))
    L = [f"R {' '.join(map(str, data + anc))}"]
    body = [f"X_ERROR({p_data}) {' '.join(map(str, data))}",
            f"CX {' '.join(f'{d} {a}' for d, a in zip(data[:-1], anc))}",
            f"CX {' '.join(f'{d} {a}' for d, a in zip(data[1:], anc))}",
            f"MR({p_meas}) {' '.join(map(str, anc))}"]
    k = len(anc)
    L += body + [f"DETECTOR rec[-{k - i}]" for i in range(k)]
    L += [f"REPEAT {rounds - 1} {{"] + body + [f"DETECTOR rec[-{k - i}] rec[-{2 * k - i}]" for i in range(k)] + ["}"]
    L += [f"M {' '.join(map(str, data))}"]
    d = len(data)
    L += [f"DETECTOR rec[-{d - i}] rec[-{d - i - 1}] rec[-{d + k - i}]" for i in range(k)]
    L += [f"OBSERVABLE_INCLUDE(0) rec[-1]"]
    return "\n".join(L)


def test_repetition_code_structure_and_statistics():
    dist, rounds, p, q = 5, 4, 0.02, 0.01
    c = CliffordCircuit(_repetition_code(dist, rounds, p, q))
    program, probs, et = c.compile()
    k = dist - 1
    assert program.num_detectors == k * (rounds + 1) and program.num_outputs == program.num_detectors + 1
    assert len(probs) == rounds * (dist + k)          # one channel per X_ERROR target and per noisy MR
    assert not program.components                     # every detector is its own f bit (or a basis row)
    assert et.shape[1] == len(probs)
    shots = 200_000
    d = c.compile_detector_sampler(seed=7).sample(shots, batch_size=50_000)
    assert d.shape == (shots, program.num_detectors)
    # first-round detector of an interior ancilla fires iff an odd number of {two data errors, its meas error}
    def odd(ps):
        r = 0.0
        for x in ps:
            r = r * (1 - x) + (1 - r) * x
        return r
    rate = d[:, 1].mean()
    assert abs(rate - odd([p, p, q])) < 4 * np.sqrt(rate / shots) + 1e-4
    # a middle-round detector compares two noisy measurements and sees one round of data errors
    rate2 = d[:, k + 1].mean()
    assert abs(rate2 - odd([p, p, q, q])) < 4 * np.sqrt(rate2 / shots) + 1e-4


def test_flips_and_unsupported_instructions():
    c = CliffordCircuit("X 0\nM 0 !0\nDETECTOR rec[-2]\nDETECTOR rec[-1]\nDETECTOR rec[-1] rec[-2]")
    an = c.analyze()
    assert [v for _, v in an.detectors] == [1, 0, 1] and all(s == 0 for s, _ in an.detectors)
    with pytest.raises(NotImplementedError):
        CliffordCircuit("T 0\nM 0").analyze()
    with pytest.raises(NotImplementedError):
        CliffordCircuit("M 0\nCX sweep[0] 1").analyze()
    with pytest.raises(ValueError):
        CliffordCircuit("M 0\nDETECTOR rec[-2]").analyze()


def test_mpp_and_basis_measurements_are_consistent():
    """Bell pair: XX and ZZ products are +1 deterministically; a Z error on one qubit flips XX only."""
    c = CliffordCircuit("""
        R 0 1
        H 0
        CX 0 1
        Z_ERROR(0.1) 0
        X_ERROR(0.2) 1
        MPP X0*X1 Z0*Z1
        DETECTOR rec[-2]
        DETECTOR rec[-1]
        MX 0 1
        DETECTOR rec[-1] rec[-2]
    """)
    an = c.analyze()
    assert [(s, v) for s, v in an.detectors] == [(0b01, 0), (0b10, 0), (0b01, 0)]


def test_nondeterministic_detector_is_rejected_and_repeated_outcomes_cancel():
    with pytest.raises(ValueError, match="not deterministic"):
        CliffordCircuit("H 0\nM 0\nDETECTOR rec[-1]").analyze()
    # the same random outcome twice cancels; a reset after a random outcome is deterministic again
    an = CliffordCircuit("H 0\nM 0\nM 0\nDETECTOR rec[-1] rec[-2]\nH 1\nMR 1\nM 1\nDETECTOR rec[-1]").analyze()
    assert [(s, v) for s, v in an.detectors] == [(0, 0), (0, 0)]
    # Bell pair: the two Z outcomes are random but equal
    an = CliffordCircuit("H 0\nCX 0 1\nM 0 1\nDETECTOR rec[-1] rec[-2]").analyze()
    assert an.detectors == [(0, 0)]
    with pytest.raises(ValueError, match="OBSERVABLE 0"):
        CliffordCircuit("H 0\nM 0\nOBSERVABLE_INCLUDE(0) rec[-1]").compile()


@pytest.mark.parametrize("basis", ["Z", "X"])
@pytest.mark.parametrize("distance", [3, 5])
def test_rotated_surface_code_compiles_to_a_direct_program(basis, distance):
    """BASELINE.json configs[0] for real: d = 3, 3 rounds -> 24 detectors, all on the direct path."""
    from tsim_amd.circuits import rotated_surface_code_memory

    rounds = 3
    c = CliffordCircuit(rotated_surface_code_memory(
        distance, rounds, basis=basis, after_clifford_depolarization=0.001,
        before_round_data_depolarization=0.002, before_measure_flip_probability=0.003,
        after_reset_flip_probability=0.004))
    program, probs, et = c.compile()
    checks = distance * distance - 1
    assert program.num_detectors == checks // 2 * 2 + checks * (rounds - 1)
    assert program.num_outputs == program.num_detectors + 1
    assert not program.components and not np.asarray(program.direct_flips).any()
    assert et.shape == (program.num_outputs, sum(int(np.log2(len(p))) for p in probs))
    # every detector is a basis row: f_j = XOR of its own error set, in output order
    assert np.asarray(program.direct_f_indices).tolist() == list(range(program.num_outputs))
    an = c.analyze()
    weights = [sum((s >> e) & 1 for s, _ in an.detectors) for e in range(an.num_e)]
    assert max(weights) <= 4 and sum(w > 0 for w in weights) > 0.6 * len(weights)
    if distance == 3:
        d = c.compile_detector_sampler(seed=3).sample(20000, batch_size=5000, append_observables=True)
        assert d.shape == (20000, program.num_outputs)
        assert 0.002 < d[:, : program.num_detectors].mean() < 0.08   # detection events are rare but present


def test_noiseless_surface_code_has_silent_detectors():
    from tsim_amd.circuits import rotated_surface_code_memory

    c = CliffordCircuit(rotated_surface_code_memory(3, 2))
    program, probs, et = c.compile()
    assert probs == [] and et.shape == (1, 0)  # no error bits: every output reads the always-zero column
    d = c.compile_detector_sampler(seed=0).sample(50, append_observables=True)
    assert d.shape == (50, program.num_outputs) and not d.any()


# ---------------------------------------------------------------------------
# measurement sampling: programs from circuit text reproduce the reference's seeded counts
# (through the CPU oracle here; through the GPU in tests/test_gpu_clifford.py)
# ---------------------------------------------------------------------------
def _oracle_counts(text, seed, shots_list):
    from conftest import run_batches
    from oracle import oracle_np as O

    program, probs, et = CliffordCircuit(text).compile_measurements()
    assert probs == []

    def np_sample(prog, f, key):
        return O.sample_program(prog, f, key)

    return program, run_batches(np_sample, program, seed, shots_list, num_f=et.shape[0])


def test_measurement_kats_from_text():
    # test/unit/test_sampler.py:223-233
    _, outs = _oracle_counts("H 0\nM 0", 0, [100] * 4)
    assert [int(o.sum()) for o in outs] == [48, 53, 52, 50]
    # test/integration/test_sampler_circuits.py:10-22
    prog, (o,) = _oracle_counts("R 0 1\nH 0\nCNOT 0 1\nM 0 1", 0, [100])
    assert len(prog.components) == 1 and prog.components[0].output_indices == (0, 1)
    assert np.array_equal(o[:, 0], o[:, 1]) and int(o[:, 0].sum()) == 48
    # test/integration/test_sampler_circuits.py:90-109: three single-output components, the last one constant
    prog, (o,) = _oracle_counts("RX 0\nRX 0\nM 0\nRX 0\nM 0\nR 0\nM 0", 0, [10])
    assert len(prog.components) == 3
    assert o.sum(axis=0).tolist() == [7, 4, 0]


def test_measurement_program_structure_with_noise():
    c = CliffordCircuit("""
        R 0 1 2
        X_ERROR(0.1) 0
        H 1
        CX 1 2
        X_ERROR(0.2) 2
        M 0 1 2 !2
    """)
    program, probs, et = c.compile_measurements()
    # record 0 = e0 (direct), records 1, 2, 3 share the random bit of the Bell pair; 2 and 3 carry e1
    assert program.num_outputs == 4 and len(probs) == 2 and et.tolist() == [[1, 0], [0, 1]]
    assert np.asarray(program.direct_f_indices).tolist() == [0]
    (comp,) = program.components
    assert comp.output_indices == (1, 2, 3) and np.asarray(comp.f_selection).tolist() == [1]
    from oracle import oracle_np as O
    from tsim_amd import prng

    f = np.array([[0, 0], [1, 0], [0, 1], [1, 1]] * 50, dtype=np.uint8)
    out = O.sample_program(program, f, prng.key(4))
    np.testing.assert_array_equal(out[:, 0], f[:, 0].astype(bool))
    np.testing.assert_array_equal(out[:, 2], out[:, 1] ^ f[:, 1].astype(bool))   # m2 = m1 ^ e1
    np.testing.assert_array_equal(out[:, 3], ~out[:, 2])                          # inverted record
    assert 60 < out[:, 1].sum() < 140


# ---------------------------------------------------------------------------
# statistics: the compiled sampler vs an independent vectorised Pauli-frame Monte Carlo of the same
# circuit (the reference's own integration test compares with Stim the same way,
# test/integration/test_sampler.py:212-257)
# ---------------------------------------------------------------------------
def _frame_monte_carlo(text, shots, seed):
    """Per-shot Pauli frames as uint8 arrays; errors are drawn per instruction, not per channel table."""
    from tsim_amd.clifford import _parse

    rng = np.random.default_rng(seed)
    ins = _parse(text)
    nq = 1 + max(int(t) for i in ins if i.name not in ("DETECTOR", "OBSERVABLE_INCLUDE", "TICK") for t in i.targets)
    fx = np.zeros((shots, nq), np.uint8)
    fz = np.zeros((shots, nq), np.uint8)
    recs, dets, obs = [], [], {}
    for i in ins:
        n, a, tg = i.name, i.args, [int(t) for t in i.targets] if i.name not in ("DETECTOR", "OBSERVABLE_INCLUDE") else i.targets
        if n == "TICK":
            continue
        if n == "H":
            for q in tg:
                fx[:, q], fz[:, q] = fz[:, q].copy(), fx[:, q].copy()
        elif n == "CX":
            for c, t in zip(tg[::2], tg[1::2]):
                fx[:, t] ^= fx[:, c]
                fz[:, c] ^= fz[:, t]
        elif n in ("R", "RX"):
            for q in tg:
                fx[:, q] = 0
                fz[:, q] = 0
        elif n in ("M", "MR", "MX"):
            for q in tg:
                recs.append((fz if n == "MX" else fx)[:, q].copy())
                if n == "MR":
                    fx[:, q] = 0
                    fz[:, q] = 0
        elif n in ("X_ERROR", "Z_ERROR"):
            for q in tg:
                (fx if n == "X_ERROR" else fz)[:, q] ^= (rng.random(shots) < a[0]).astype(np.uint8)
        elif n == "DEPOLARIZE1":
            for q in tg:
                r = rng.random(shots)
                k = np.where(r < a[0], rng.integers(1, 4, shots), 0)   # 1 = X, 2 = Y, 3 = Z
                fx[:, q] ^= ((k == 1) | (k == 2)).astype(np.uint8)
                fz[:, q] ^= ((k == 2) | (k == 3)).astype(np.uint8)
        elif n == "DEPOLARIZE2":
            for qa, qb in zip(tg[::2], tg[1::2]):
                r = rng.random(shots)
                k = np.where(r < a[0], rng.integers(1, 16, shots), 0)  # two base-4 digits, not both I
                pa, pb = k % 4, k // 4                                  # 0 I, 1 X, 2 Y, 3 Z
                fx[:, qa] ^= ((pa == 1) | (pa == 2)).astype(np.uint8)
                fz[:, qa] ^= ((pa == 2) | (pa == 3)).astype(np.uint8)
                fx[:, qb] ^= ((pb == 1) | (pb == 2)).astype(np.uint8)
                fz[:, qb] ^= ((pb == 2) | (pb == 3)).astype(np.uint8)
        elif n == "DETECTOR":
            v = np.zeros(shots, np.uint8)
            for t in tg:
                v ^= recs[len(recs) - int(t[5:-1])]
            dets.append(v)
        elif n == "OBSERVABLE_INCLUDE":
            v = obs.setdefault(int(a[0]), np.zeros(shots, np.uint8))
            for t in tg:
                v ^= recs[len(recs) - int(t[5:-1])]
        else:
            raise AssertionError(n)
    return np.stack(dets + [obs[k] for k in sorted(obs)], axis=1)


@pytest.mark.parametrize("channel", ["after_clifford_depolarization", "after_reset_flip_probability",
                                     "before_measure_flip_probability", "before_round_data_depolarization"])
def test_surface_code_single_noise_channel_matches_frame_monte_carlo(channel):
    from tsim_amd.circuits import rotated_surface_code_memory

    kw = {channel: 0.01}
    text = rotated_surface_code_memory(3, 3, basis="X", **kw)
    shots = 120_000
    c = CliffordCircuit(text)
    got = c.compile_detector_sampler(seed=42).sample(shots, batch_size=shots // 4, append_observables=True)
    mc = _frame_monte_carlo(text, shots, seed=7)
    assert got.shape == mc.shape
    tot_a, tot_b = int(got.sum()), int(mc.sum())
    assert tot_a > 1000
    # two independent samples of the same distribution: 5 sigma of the difference of the totals
    # (events within a shot are positively correlated, hence the generous variance factor)
    assert abs(tot_a - tot_b) < 5 * np.sqrt(6 * (tot_a + tot_b))
    # per-output rates too
    ra, rb = got.mean(axis=0), mc.mean(axis=0)
    assert np.all(np.abs(ra - rb) < 6 * np.sqrt((ra + rb) / shots) + 2e-4)


def test_stim_generated_dialect_parses():
    """Coordinates on QUBIT_COORDS / DETECTOR / SHIFT_COORDS, tags, comments and REPEAT blocks in the
    style of Stim's circuit generators."""
    text = """
        QUBIT_COORDS(1, 1) 1
        QUBIT_COORDS(2, 0) 2
        QUBIT_COORDS(3, 1) 3
        R 1 2 3   # reset everything
        X_ERROR(0.01) 1 2 3
        TICK
        CX 1 2
        DEPOLARIZE2(0.01) 1 2
        TICK
        CX 3 2
        DEPOLARIZE2(0.01) 3 2
        TICK
        X_ERROR(0.02) 2
        MR 2
        X_ERROR(0.01) 2
        DETECTOR(2, 0, 0) rec[-1]
        REPEAT 2 {
            TICK
            DEPOLARIZE1(0.01) 1 3
            CX 1 2
            CX 3 2
            MR(0.02) 2
            SHIFT_COORDS(0, 0, 1)
            DETECTOR(2, 0, 0) rec[-1] rec[-2]
        }
        M[final] 1 3
        DETECTOR(2, 0, 1) rec[-1] rec[-2] rec[-3]
        OBSERVABLE_INCLUDE(0) rec[-1]
    """
    c = CliffordCircuit(text)
    program, probs, et = c.compile()
    assert program.num_detectors == 4 and program.num_outputs == 5
    # 3 + 1 + 1 one-bit channels, 2 DEPOLARIZE2 (4 bits), then per round 2 DEPOLARIZE1 (2 bits) + 1 noisy MR
    assert [len(p) for p in probs] == [2, 2, 2, 16, 16, 2, 2] + [4, 4, 2] * 2
    d = c.compile_detector_sampler(seed=5).sample(5000, append_observables=True)
    assert d.shape == (5000, 5) and 0 < d.mean() < 0.2


def test_correlated_error_chain_numbering_and_sampling():
    """E / ELSE_CORRELATED_ERROR: one channel with one bit per alternative, at most one fires; the chain
    is numbered when it is closed (reference core/instructions.py:759-816), i.e. AFTER channels that
    appear while it is open."""
    c = CliffordCircuit("""
        R 0 1 2
        E(0.2) X0 X1
        ELSE_CORRELATED_ERROR(0.5) X2
        X_ERROR(0.1) 0
        M 0 1 2
        DETECTOR rec[-3]
        DETECTOR rec[-2]
        DETECTOR rec[-1]
    """)
    an = c.analyze()
    # e0 = the X_ERROR (numbered first), e1 = first alternative (X0 X1), e2 = second alternative (X2)
    assert [s for s, _ in an.detectors] == [0b011, 0b010, 0b100]
    assert [len(p) for p in an.channel_probs] == [2, 4]
    np.testing.assert_allclose(an.channel_probs[1], [0.8 * 0.5, 0.2, 0.8 * 0.5, 0.0])
    d = c.compile_detector_sampler(seed=9).sample(200_000, batch_size=50_000)
    assert not (d[:, 1] & d[:, 2]).any()                      # the alternatives exclude each other
    assert abs(d[:, 1].mean() - 0.2) < 0.005 and abs(d[:, 2].mean() - 0.4) < 0.005
    assert abs(d[:, 0].mean() - (0.2 * 0.9 + 0.8 * 0.1)) < 0.005
    # a second E closes the first chain
    an = CliffordCircuit("R 0\nE(0.1) X0\nE(0.2) X0\nM 0\nDETECTOR rec[-1]").analyze()
    assert [len(p) for p in an.channel_probs] == [2, 2] and an.detectors[0][0] == 0b11


def test_correlated_error_chain_among_many_error_bits():
    """Regression (round-1 advisor): the not-yet-numbered chain bits must never collide with numbered error
    bits, however many there are - before the chain opens, while it is open, and after it closes."""
    n = 90  # qubits; > 64 error bits before, during and after the chain
    lines = ["R " + " ".join(map(str, range(n)))]
    lines += [f"X_ERROR(0.01) {q}" for q in range(n)]            # e0 .. e89
    lines += [f"E(0.25) X{n - 1}"]                                  # chain opens (numbered at its close)
    lines += [f"X_ERROR(0.01) {q}" for q in range(n)]            # e90 .. e179, while the chain is open
    lines += [f"ELSE_CORRELATED_ERROR(0.5) X0 X{n - 2}"]
    lines += [f"E(0.125) X1"]                                      # closes the first chain: e180, e181; opens a second
    lines += [f"X_ERROR(0.01) {q}" for q in range(n)]            # e182 .. e271
    lines += ["M " + " ".join(map(str, range(n)))]                  # end of circuit closes the second chain: e272
    lines += [f"DETECTOR rec[-{n - q}]" for q in range(n)]
    an = CliffordCircuit("\n".join(lines)).analyze()
    assert an.num_e == 3 * n + 3
    want = [(1 << q) | (1 << (n + q)) | (1 << (2 * n + 2 + q)) for q in range(n)]
    want[n - 1] |= 1 << (2 * n)          # first alternative of chain 1
    want[0] |= 1 << (2 * n + 1)          # second alternative
    want[n - 2] |= 1 << (2 * n + 1)
    want[1] |= 1 << (3 * n + 2)          # chain 2
    assert [s for s, _ in an.detectors] == want
    assert [len(p) for p in an.channel_probs] == [2] * (2 * n) + [4] + [2] * n + [2]


def test_pair_measurements_and_mpad():
    an = CliffordCircuit("""
        R 0 1
        H 0
        CX 0 1
        Z_ERROR(0.1) 1
        MXX 0 1
        MZZ 0 1
        MPAD 1 0
        DETECTOR rec[-4]
        DETECTOR rec[-3]
        DETECTOR rec[-2] rec[-1]
    """).analyze()
    assert an.detectors == [(0b1, 0), (0, 0), (0, 1)]


def test_heralded_erase_records_a_herald_bit():
    c = CliffordCircuit("""
        R 0
        HERALDED_ERASE(0.4) 0
        M 0
        DETECTOR rec[-2]
        DETECTOR rec[-1]
    """)
    an = c.analyze()
    # bits: herald, Z component, X component; the herald is its own record, X flips the Z measurement
    assert an.detectors == [(0b001, 0), (0b100, 0)] and [len(p) for p in an.channel_probs] == [8]
    d = c.compile_detector_sampler(seed=4).sample(200_000, batch_size=50_000)
    assert abs(d[:, 0].mean() - 0.4) < 0.005            # heralded with probability p
    assert abs(d[:, 1].mean() - 0.2) < 0.005            # X or Y: half of the fired cases
    assert not (d[:, 1] & ~d[:, 0]).any()               # a flip never comes without its herald


def test_circuit_properties_and_without_noise(tmp_path):
    from tsim_amd.circuits import rotated_surface_code_memory

    text = rotated_surface_code_memory(3, 2, after_clifford_depolarization=0.01, before_measure_flip_probability=0.01)
    f = tmp_path / "sc.stim"
    f.write_text(text)
    c = CliffordCircuit.from_file(str(f))
    assert str(c) == text and c.is_clifford and c.num_qubits == 17
    assert c.num_measurements == 8 * 2 + 9 and c.num_detectors == 4 + 8 + 4 and c.num_observables == 1
    quiet = c.without_noise()
    program, probs, et = quiet.compile()
    assert probs == [] and program.num_detectors == c.num_detectors
    assert not quiet.compile_detector_sampler(seed=1).sample(20).any()


def test_compiled_form_matches_the_committed_fixture():
    """tests/golden/clifford_golden.npz (self-generated, see gen_clifford_golden.py): conventions that
    decide stream-level agreement with the reference must not drift."""
    import importlib.util
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("gen_clifford_golden", os.path.join(here, "golden", "gen_clifford_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = np.load(os.path.join(here, "golden", "clifford_golden.npz"))
    for name, text in gen.CASES.items():
        for k, v in gen.compiled(text).items():
            np.testing.assert_array_equal(v, gold[f"{name}.{k}"], err_msg=f"{name}.{k}")


# ---------------------------------------------------------------------------
# every Clifford gate of the reference's gate table (core/instructions.py GATE_TABLE): the tableau's
# conjugation action against a unitary written down from the gate's definition
# ---------------------------------------------------------------------------
_I2 = np.eye(2, dtype=complex)
_PX = np.array([[0, 1], [1, 0]], dtype=complex)
_PY = np.array([[0, -1j], [1j, 0]], dtype=complex)
_PZ = np.diag([1, -1]).astype(complex)
_PAULI = {"I": _I2, "X": _PX, "Y": _PY, "Z": _PZ}


def _matrix_images(U, n):
    """U P U^dagger for P = X_0, Z_0, X_1, Z_1...: list of (sign, pauli string)."""
    out = []
    names = ["".join(s) for s in __import__("itertools").product("IXYZ", repeat=n)]
    for q in range(n):
        for g in "XZ":
            P = np.array([[1]], dtype=complex)
            for k in range(n):
                P = np.kron(P, _PAULI[g] if k == q else _I2)
            Q = U @ P @ U.conj().T
            hit = None
            for nm in names:
                M = np.array([[1]], dtype=complex)
                for ch in nm:
                    M = np.kron(M, _PAULI[ch])
                c = np.trace(M.conj().T @ Q) / 2**n
                if abs(abs(c) - 1) < 1e-9:
                    assert abs(c.imag) < 1e-9
                    hit = (int(round(c.real)), nm)
            assert hit is not None, "not a Clifford"
            out.append(hit)
    return out


def _tableau_images(apply, n):
    from tsim_amd.clifford import _Sim

    sim = _Sim(n)
    apply(sim)
    t = sim.tab
    out = []
    for q in range(n):
        for row in (q, t.n + q):  # destabilizer row q started as X_q, stabilizer row as Z_q
            nm = "".join("IXZY"[int(t.x[row, k]) + 2 * int(t.z[row, k])] for k in range(n))
            out.append((-1 if t.r[row] else 1, nm))
    return out


def _one_qubit_unitaries():
    r = 1 / np.sqrt(2)
    S = np.diag([1, 1j])
    H = (_PX + _PZ) * r
    sx = H @ S @ H
    sy = np.array([[1 + 1j, -1 - 1j], [1 + 1j, 1 + 1j]]) / 2
    U = {"I": _I2, "X": _PX, "Y": _PY, "Z": _PZ, "H": H, "H_XZ": H, "H_XY": (_PX + _PY) * r, "H_YZ": (_PY + _PZ) * r,
         "H_NXY": (_PY - _PX) * r, "H_NXZ": (_PZ - _PX) * r, "H_NYZ": (_PZ - _PY) * r,
         "S": S, "SQRT_Z": S, "S_DAG": S.conj().T, "SQRT_Z_DAG": S.conj().T,
         "SQRT_X": sx, "SQRT_X_DAG": sx.conj().T, "SQRT_Y": sy, "SQRT_Y_DAG": sy.conj().T}
    return U


def _cycle_unitary(name):
    """The 120-degree rotation realising the axis cycle in the gate's name: its axis is the fixed
    vector of the cycle, its sense the one whose conjugation action is the cycle."""
    (sx, ax), (sz, az) = _cycle_images(name)
    col = {"X": 0, "Y": 1, "Z": 2}
    R = np.zeros((3, 3))
    R[col[ax], 0] = sx
    R[col[az], 2] = sz
    R[:, 1] = np.cross(R[:, 2], R[:, 0])  # Y = i X Z keeps the frame right-handed
    w, v = np.linalg.eig(R)
    axis = np.real(v[:, np.argmin(abs(w - 1))])
    axis = axis / abs(axis).max()
    n = axis[0] * _PX + axis[1] * _PY + axis[2] * _PZ
    for sign in (1, -1):
        u = (_I2 + sign * 1j * n) / 2
        if _matrix_images(u, 1) == [(sx, ax), (sz, az)]:
            return u
    raise AssertionError(name)



def _cycle_images(name):
    """C_NXYZ: -X -> Y -> Z -> -X, read off the name; returns the expected images of X and Z."""
    body, axes, k = name[2:], [], 0
    while k < len(body):
        sgn = 1
        if body[k] == "N":
            sgn, k = -1, k + 1
        axes.append((sgn, body[k]))
        k += 1
    img = {}
    for i, (s, a) in enumerate(axes):
        s2, b = axes[(i + 1) % 3]
        img[a] = (s * s2, b)
    return [img["X"], img["Z"]]


def test_one_qubit_cliffords_match_their_unitaries():
    from tsim_amd.clifford import _ACTION_1Q

    U = _one_qubit_unitaries()
    for name in _ACTION_1Q:
        if name.startswith("C_"):
            U[name] = _cycle_unitary(name)
    assert set(U) == set(_ACTION_1Q)
    assert np.allclose(U["C_XYZ"], (_I2 - 1j * (_PX + _PY + _PZ)) / 2)  # right-handed about (1,1,1)
    for name, u in U.items():
        assert np.allclose(u @ u.conj().T, _I2), name
        got = _tableau_images(lambda sim: sim.gate1(name, 0), 1)
        assert got == _matrix_images(u, 1), name
        if name.startswith("C_"):
            assert got == _cycle_images(name), name


def _two_qubit_unitaries():
    def proj(P, sign):
        return (np.eye(P.shape[0]) + sign * P) / 2

    CX = np.kron(proj(_PZ, 1), _I2) + np.kron(proj(_PZ, -1), _PX)
    CZ = np.diag([1, 1, 1, -1]).astype(complex)
    SWAP = np.eye(4, dtype=complex)[[0, 2, 1, 3]]
    ISWAP = np.array([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]], dtype=complex)
    U = {"SWAP": SWAP, "ISWAP": ISWAP, "ISWAP_DAG": ISWAP.conj().T, "CXSWAP": SWAP @ CX, "SWAPCX": CX @ SWAP,
         "CZSWAP": SWAP @ CZ, "SWAPCZ": CZ @ SWAP}
    for a in "XYZ":      # A-controlled-B: B on the second qubit when the first is in the -1 eigenstate of A
        for b in "XYZ":
            u = np.kron(proj(_PAULI[a], 1), _I2) + np.kron(proj(_PAULI[a], -1), _PAULI[b])
            U[f"{a}C{b}"] = u
            if a == "Z":
                U[f"C{b}"] = u
    U["CNOT"] = U["ZCX"]
    for a in "XYZ":      # phase the -1 eigenspace of AA by i
        PP = np.kron(_PAULI[a], _PAULI[a])
        U[f"SQRT_{a}{a}"] = proj(PP, 1) + 1j * proj(PP, -1)
        U[f"SQRT_{a}{a}_DAG"] = proj(PP, 1) - 1j * proj(PP, -1)
    return U


def test_two_qubit_cliffords_match_their_unitaries():
    from tsim_amd.clifford import _TWO_QUBIT

    U = _two_qubit_unitaries()
    assert set(U) == set(_TWO_QUBIT)
    for name, u in U.items():
        got = _tableau_images(lambda sim: sim.gate2(name, 0, 1), 2)
        assert got == _matrix_images(u, 2), name
    # and with the operands the other way round (second listed qubit is the control / first operand)
    SW = U["SWAP"]
    for name in ("CX", "XCY", "YCZ", "CXSWAP", "SQRT_XX", "ISWAP_DAG"):
        got = _tableau_images(lambda sim: sim.gate2(name, 1, 0), 2)
        assert got == _matrix_images(SW @ U[name] @ SW, 2), name


def test_spp_is_the_square_root_of_a_pauli_product():
    rng = np.random.default_rng(5)
    for _ in range(20):
        n = int(rng.integers(1, 4))
        qs = [int(q) for q in rng.permutation(3)[:n]]
        kinds = [str(rng.choice(list("XYZ"))) for _ in qs]
        P = np.array([[1]], dtype=complex)
        for q in range(3):
            P = np.kron(P, _PAULI[kinds[qs.index(q)]] if q in qs else _I2)
        for dag in (False, True):
            u = (np.eye(8) + P) / 2 + (-1j if dag else 1j) * (np.eye(8) - P) / 2
            got = _tableau_images(lambda sim: sim.spp(list(zip(kinds, qs)), dag=dag), 3)
            assert got == _matrix_images(u, 3), (kinds, qs, dag)


def test_new_gates_through_the_text_front_end():
    """Deterministic outcomes of small circuits using the added gates; tags and no-ops."""
    def records(text):
        an = CliffordCircuit(text).analyze()
        assert not any(an.rec_syms), text
        return an.rec_vals

    assert records("H_XY 0\nM 0") == [1]                 # Z -> -Z
    assert records("C_XYZ 0\nMX 0") == [0]               # Z -> X
    assert records("C_NXYZ 0\nMX 0") == [1]              # Z -> -X
    assert records("C_ZYX 0\nMY 0") == [0]               # Z -> Y
    assert records("C_ZNYX 0\nMY 0") == [1]              # Z -> -Y
    assert records("H_NXZ 0\nMX 0") == [1]               # Z -> -X
    assert records("X 0\nISWAP 0 1\nM 0 1") == [0, 1]
    assert records("X 0\nCXSWAP 0 1\nM 0 1") == [1, 1]   # CX then SWAP
    assert records("X 0\nSWAPCX 0 1\nM 0 1") == [0, 1]   # SWAP then CX: control 0 is |0>
    assert records("RX 0\nX 0\nZ 0\nXCX 0 1\nM 1") == [1]   # control in |->: X on the target
    assert records("RY 0\nZ 0\nYCZ 0 1\nRX 1\nYCZ 0 1\nMX 1") == [1]  # control in Y = -1: Z on |+> gives |->
    assert records("SQRT_XX 0 1\nSQRT_XX 0 1\nM 0 1") == [1, 1]       # XX
    assert records("SQRT_YY 0 1\nSQRT_YY_DAG 0 1\nM 0 1") == [0, 0]
    assert records("SPP X0*X1\nSPP X0*X1\nM 0 1") == [1, 1]
    assert records("SPP Z0\nSPP !Z0\nRX 0\nSPP Z0\nSPP_DAG Z0\nMX 0") == [0]
    assert records("RX 0\nSPP Z0\nSPP Z0\nMX 0") == [1]              # S S = Z on |+>
    assert records("I_ERROR(0.1) 0\nII 0 1\nII_ERROR(0.1) 0 1\nS[note] 0\nM 0") == [0]
    for bad in ("S[T] 0", "S_DAG[T] 0", "T 0", "T_DAG 0", "I[R_Z(theta=0.25*pi)] 0", "I[U3(theta=0.1*pi, phi=0, lambda=0)] 0",
                "SPP[T] X0", "SPP[R_PAULI(theta=0.1*pi)] X0*Z1"):
        with pytest.raises(NotImplementedError):
            CliffordCircuit(bad + "\nM 0")
    with pytest.raises(ValueError):
        CliffordCircuit("CX 0 0\nM 0").analyze()


# ---------------------------------------------------------------------------
# classically controlled Paulis (feedback)
# ---------------------------------------------------------------------------
_TELEPORT = """
{prep}
R 1 2
H 1
CX 1 2
CX 0 1
H 0
M{noise} 0 1
{cx} 
{cz}
{meas} 2
DETECTOR rec[-1]
"""


@pytest.mark.parametrize("prep,meas,expect", [("R 0", "M", 0), ("R 0\nX 0", "M", 1), ("RX 0", "MX", 0),
                                              ("RX 0\nZ 0", "MX", 1), ("RY 0", "MY", 0), ("RY 0\nX 0", "MY", 1)])
@pytest.mark.parametrize("spelling", [("CX rec[-1] 2", "CZ rec[-2] 2"), ("XCZ 2 rec[-1]", "CZ 2 rec[-2]"),
                                      ("CX rec[-1] 2 rec[-2] 1", "CZ rec[-2] 2")])
def test_teleportation_with_feedback_is_deterministic(prep, meas, expect, spelling):
    """The Bell measurement's two random outcomes cancel against the corrections: the teleported state
    is measured deterministically - without the feedback the detector is rejected as random."""
    c = CliffordCircuit(_TELEPORT.format(prep=prep, noise="", cx=spelling[0], cz=spelling[1], meas=meas))
    an = c.analyze()
    assert an.rec_syms[0] and an.rec_syms[1] and an.rec_syms[0] != an.rec_syms[1]
    assert an.detectors == [(0, expect)]
    if meas != "M":  # the Z correction matters for X / Y inputs
        with pytest.raises(ValueError, match="not deterministic"):
            CliffordCircuit(_TELEPORT.format(prep=prep, noise="", cx=spelling[0], cz="", meas=meas)).analyze()
    if meas != "MX":  # the X correction matters for Z / Y inputs
        with pytest.raises(ValueError, match="not deterministic"):
            CliffordCircuit(_TELEPORT.format(prep=prep, noise="", cx="", cz=spelling[1], meas=meas)).analyze()


def test_feedback_carries_measurement_errors_into_the_frame():
    """A flipped Bell-measurement record applies the wrong correction: the record error of qubit 1
    (X correction) flips a Z-basis readout, that of qubit 0 (Z correction) an X-basis one."""
    z = CliffordCircuit(_TELEPORT.format(prep="R 0", noise="(0.125)", cx="CX rec[-1] 2", cz="CZ rec[-2] 2", meas="M"))
    an = z.analyze()
    assert an.num_e == 2 and an.rec_sets[0] == 0b01 and an.rec_sets[1] == 0b10
    assert an.detectors == [(0b10, 0)]
    x = CliffordCircuit(_TELEPORT.format(prep="RX 0", noise="(0.125)", cx="CX rec[-1] 2", cz="CZ rec[-2] 2", meas="MX"))
    assert x.analyze().detectors == [(0b01, 0)]
    y = CliffordCircuit(_TELEPORT.format(prep="RY 0", noise="(0.125)", cx="CY rec[-1] 2", cz="", meas="MX"))
    with pytest.raises(ValueError):  # Y-basis input, X readout: random whatever the feedback
        y.analyze()
    # sampling: the detector fires exactly when record 1's error fires -> rate 1/8
    d = z.compile_detector_sampler(seed=3).sample(20000)
    assert abs(d.mean() - 0.125) < 0.01


def test_feedback_on_a_constant_record_and_bad_operands():
    an = CliffordCircuit("X 0\nM 0\nCX rec[-1] 1\nCY rec[-1] 2\nYCZ 3 rec[-1]\nM 1 2 3\nH 4\nCZ rec[-4] 4\nMX 4").analyze()
    assert an.rec_vals == [1, 1, 1, 1, 1] and not any(an.rec_syms)   # last: Z on |+> gives |->
    an = CliffordCircuit("X 0\nM !0\nCX rec[-1] 1\nM 1").analyze()
    assert an.rec_vals == [0, 0]                                      # the inverted record is what controls
    for bad in ("M 0\nCX 1 rec[-1]", "M 0\nXCZ rec[-1] 1", "M 0\nCX rec[-1] rec[-1]"):
        with pytest.raises(ValueError):
            CliffordCircuit(bad).analyze()
    with pytest.raises(NotImplementedError):
        CliffordCircuit("M 0\nSWAP rec[-1] 1").analyze()
    with pytest.raises(NotImplementedError):
        CliffordCircuit("CX sweep[0] 1\nM 1").analyze()
