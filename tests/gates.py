"""Conditioning gates of the parity tests.

The stated bar for R, t, f is 1e-8 relative.  The reference's algorithms (a 78x78 LU and an
eigen-decomposition for P4Pf, a closed-form quartic with sqrt(1 - x^2) back-substitutions for
P3P) lose more than that on ill-conditioned samples all by themselves: the double-precision
oracle then differs from the SAME oracle evaluated in long double by e_ref > 1e-8.  A device
result is accepted within max(1e-8, 20 e_ref) of the long-double oracle; samples on which the
two oracles do not even agree on the number of solutions are reported as ungated.  For P4Pf
e_ref is taken over the sample and a few copies perturbed in the last place (see p4pf_gate)."""
import numpy as np

import oracle_lib as O

RTOL = 1e-8


def relerr(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)) if a.size else 0.0


PERTURBED = 3      # extra runs of the double oracle on inputs moved by less than one unit in the last place


def p4pf_gate(m, M):
    """-> (n, f, R, t of the long-double oracle, e_ref), or None when the double and the
    long-double oracle disagree on the solution count.  e_ref is the largest discrepancy between
    the long-double result and the double oracle run on the sample itself and on PERTURBED copies
    whose inputs are moved by at most 2^-52 relative: one run alone is one draw of the rounding
    noise and underestimates what the sample's conditioning does to a different but equally valid
    order of operations."""
    n2, f2, R2, t2 = O.p4pf(m, M, long_double=True)
    rng = np.random.default_rng(20240229)
    e_ref = 0.0
    for k in range(PERTURBED + 1):
        mm, MM = (m, M) if k == 0 else (m * (1.0 + rng.uniform(-1, 1, np.shape(m)) * 2.0 ** -52),
                                        M * (1.0 + rng.uniform(-1, 1, np.shape(M)) * 2.0 ** -52))
        n, f, R, t = O.p4pf(mm, MM)
        if n != n2:
            return None
        if n > 0:
            e_ref = max(e_ref, relerr(f, f2), relerr(R, R2, 1e-3), relerr(t, t2, 1e-3))
    return n2, f2, R2, t2, e_ref


def p3p_gate(f, P):
    """-> (status, C[4,3], Rc[4,3,3] of the long-double oracle, finite[4], e_ref), or None when the
    two oracles disagree on the status or on which of the four blocks are finite (a root beyond
    |cos| = 1 turns sqrt(1 - x^2) into NaN in one precision and not in the other)."""
    s0, C0, R0 = O.p3p(f, P)
    s1, C1, R1 = O.p3p(f, P, long_double=True)
    if s0 != s1:
        return None
    if s0 != 0:
        return s1, C1, R1, np.zeros(4, bool), 0.0
    fin0 = np.isfinite(R0).all(axis=(1, 2)) & np.isfinite(C0).all(axis=1)
    fin1 = np.isfinite(R1).all(axis=(1, 2)) & np.isfinite(C1).all(axis=1)
    if not np.array_equal(fin0, fin1):
        return None
    e_ref = max(relerr(C0[fin1], C1[fin1], 1e-3), relerr(R0[fin1], R1[fin1], 1e-3)) if fin1.any() else 0.0
    return s1, C1, R1, fin1, e_ref


def sample_problem(q, sample):
    """The solver inputs the RANSAC step builds from a sample tuple (basic.cpp:162-183 / 234-255)."""
    s = np.asarray(sample)[:4 if q["focal"] == 0 else 3].astype(np.int64)
    u = -(q["kx"][s].astype(np.float64) - q["width"] / 2.0)
    v = -(q["ky"][s].astype(np.float64) - q["height"] / 2.0)
    W = np.stack([q["X"][s], q["Y"][s], q["Z"][s]], 1).astype(np.float64)
    if q["focal"] == 0:
        return np.stack([u, v], 1), W
    b = np.stack([u, v, np.full(3, float(q["focal"]))], 1)
    nb = np.sqrt(b[:, 0] * b[:, 0] + b[:, 1] * b[:, 1] + b[:, 2] * b[:, 2])
    return b / nb[:, None], W


def pose_tolerance(q, sample):
    """Relative tolerance for the pose of one RANSAC step: max(1e-8, 20 e_ref), or None (ungated)."""
    a, W = sample_problem(q, sample)
    g = p4pf_gate(a, W) if q["focal"] == 0 else p3p_gate(a, W)
    if g is None:
        return None
    return max(RTOL, 20.0 * g[-1])
