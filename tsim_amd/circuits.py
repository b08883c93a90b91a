"""Circuit text generators for the Clifford front-end (`tsim_amd.clifford`).

The reference's own tests build their circuits with ``stim.Circuit.generated("surface_code:
rotated_memory_x", ...)`` (test/integration/test_sampler.py:212-257); Stim is not available here, so
these are independent constructions of the same families from their geometry.  They are examples and
test inputs, not part of the hot path.
"""

from __future__ import annotations


def repetition_code_memory(distance: int, rounds: int, *, before_round_data_flip: float = 0.0,
                           measure_flip: float = 0.0) -> str:
    """Bit-flip repetition code: data on even qubits, parity ancillas on the odd ones between them."""
    data = list(range(0, 2 * distance, 2))
    anc = list(range(1, 2 * distance - 1, 2))
    k = len(anc)
    j = lambda qs: " ".join(map(str, qs))  # noqa: E731
    body = []
    if before_round_data_flip > 0:
        body.append(f"X_ERROR({before_round_data_flip}) {j(data)}")
    body += [f"CX {' '.join(f'{d} {a}' for d, a in zip(data[:-1], anc))}",
             f"CX {' '.join(f'{d} {a}' for d, a in zip(data[1:], anc))}",
             f"MR({measure_flip}) {j(anc)}" if measure_flip > 0 else f"MR {j(anc)}"]
    L = [f"R {j(data + anc)}"] + body + [f"DETECTOR rec[-{k - i}]" for i in range(k)]
    if rounds > 1:
        L += [f"REPEAT {rounds - 1} {{"] + body + [f"DETECTOR rec[-{k - i}] rec[-{2 * k - i}]" for i in range(k)] + ["}"]
    L.append(f"M {j(data)}")
    d = len(data)
    L += [f"DETECTOR rec[-{d - i}] rec[-{d - i - 1}] rec[-{d + k - i}]" for i in range(k)]
    L.append("OBSERVABLE_INCLUDE(0) rec[-1]")
    return "\n".join(L)


def rotated_surface_code_memory(distance: int, rounds: int, *, basis: str = "Z",
                                after_clifford_depolarization: float = 0.0,
                                before_round_data_depolarization: float = 0.0,
                                before_measure_flip_probability: float = 0.0,
                                after_reset_flip_probability: float = 0.0) -> str:
    """Rotated surface code memory experiment in the Z or X basis.

    Data qubits sit at odd coordinates (2i+1, 2j+1), measure qubits at even coordinates; a measure
    qubit is X-type when its column and row indices have different parity.  X-type checks on the
    left/right boundary columns and Z-type checks on the top/bottom boundary rows are dropped, which
    leaves d^2 - 1 checks.  The four CX layers visit the neighbours in an order whose two middle
    steps are swapped between X- and Z-type checks, so that all checks commute through each other.
    """
    d = distance
    basis = basis.upper()
    if basis not in ("X", "Z"):
        raise ValueError("basis must be 'X' or 'Z'")
    data = [(2 * i + 1, 2 * j + 1) for j in range(d) for i in range(d)]
    xm, zm = [], []
    for cx in range(d + 1):
        for cy in range(d + 1):
            x_type = (cx % 2) != (cy % 2)
            if (cx == 0 or cx == d) and x_type:
                continue
            if (cy == 0 or cy == d) and not x_type:
                continue
            (xm if x_type else zm).append((2 * cx, 2 * cy))
    index = {c: k for k, c in enumerate(sorted(data + xm + zm, key=lambda c: (c[1], c[0])))}
    dq = [index[c] for c in data]
    xq = [index[c] for c in xm]
    zq = [index[c] for c in zm]
    mq = sorted(xq + zq)
    j = lambda qs: " ".join(map(str, qs))  # noqa: E731

    order_x = [(1, 1), (-1, 1), (1, -1), (-1, -1)]
    order_z = [(1, 1), (1, -1), (-1, 1), (-1, -1)]
    layers = []
    for step in range(4):
        pairs = []
        for (mx, my) in xm:
            nb = (mx + order_x[step][0], my + order_x[step][1])
            if nb in index and nb in data:
                pairs += [index[(mx, my)], index[nb]]      # X check: measure qubit controls
        for (mx, my) in zm:
            nb = (mx + order_z[step][0], my + order_z[step][1])
            if nb in index and nb in data:
                pairs += [index[nb], index[(mx, my)]]      # Z check: data qubit controls
        layers.append(pairs)

    def noisy2(pairs):
        out = [f"CX {j(pairs)}"]
        if after_clifford_depolarization > 0:
            out.append(f"DEPOLARIZE2({after_clifford_depolarization}) {j(pairs)}")
        return out

    def noisy1(name, qs):
        out = [f"{name} {j(qs)}"]
        if after_clifford_depolarization > 0:
            out.append(f"DEPOLARIZE1({after_clifford_depolarization}) {j(qs)}")
        return out

    def cycle():
        out = ["TICK"]
        if before_round_data_depolarization > 0:
            out.append(f"DEPOLARIZE1({before_round_data_depolarization}) {j(dq)}")
        out += noisy1("H", xq)
        for pairs in layers:
            out += ["TICK"] + noisy2(pairs)
        out += ["TICK"] + noisy1("H", xq) + ["TICK"]
        if before_measure_flip_probability > 0:
            out.append(f"X_ERROR({before_measure_flip_probability}) {j(mq)}")
        out.append(f"MR {j(mq)}")
        if after_reset_flip_probability > 0:
            out.append(f"X_ERROR({after_reset_flip_probability}) {j(mq)}")
        return out

    nm = len(mq)
    pos = {q: k for k, q in enumerate(mq)}          # position of a measure qubit inside one MR
    chosen = zq if basis == "Z" else xq             # checks that are deterministic in the first round
    L = [f"R{'X' if basis == 'X' else ''} {j(dq)}", f"R {j(mq)}"]
    if after_reset_flip_probability > 0:
        L.append(f"{'Z' if basis == 'X' else 'X'}_ERROR({after_reset_flip_probability}) {j(dq)}")
        L.append(f"X_ERROR({after_reset_flip_probability}) {j(mq)}")
    L += cycle()
    L += [f"DETECTOR rec[-{nm - pos[q]}]" for q in chosen]
    if rounds > 1:
        L += [f"REPEAT {rounds - 1} {{"] + cycle()
        L += [f"DETECTOR rec[-{nm - pos[q]}] rec[-{2 * nm - pos[q]}]" for q in mq] + ["}"]
    if before_measure_flip_probability > 0:
        L.append(f"{'Z' if basis == 'X' else 'X'}_ERROR({before_measure_flip_probability}) {j(dq)}")
    L.append(f"M{'X' if basis == 'X' else ''} {j(dq)}")
    nd = len(dq)
    dpos = {q: k for k, q in enumerate(dq)}
    coords = {index[c]: c for c in index}
    for q in chosen:
        mx, my = coords[q]
        nbs = [index[(mx + dx, my + dy)] for dx in (-1, 1) for dy in (-1, 1) if (mx + dx, my + dy) in index
               and (mx + dx, my + dy) in data]
        recs = [f"rec[-{nd - dpos[n]}]" for n in nbs] + [f"rec[-{nd + nm - pos[q]}]"]
        L.append("DETECTOR " + " ".join(recs))
    # logical operator: a row (Z basis: Z along the top row) or column (X basis) of data qubits
    line = [index[(2 * i + 1, 1)] for i in range(d)] if basis == "Z" else [index[(1, 2 * k + 1)] for k in range(d)]
    L.append("OBSERVABLE_INCLUDE(0) " + " ".join(f"rec[-{nd - dpos[q]}]" for q in line))
    return "\n".join(L)
