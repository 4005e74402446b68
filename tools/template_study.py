#!/usr/bin/env python3
"""Sizing study: how small can an elimination template for the four P4Pf equations be?

The reference eliminates a 78 x 88 template (P4pf.cpp:959-1109) that an automatic generator produced
from the four distance-ratio equations in (zb, zc, zd, w = f^2) (P4pf.cpp:1435-1519).  This script
derives templates from those equations from scratch, in exact arithmetic modulo a prime, the way
such generators work:

  1. a random geometric instance mod p (image points, depths and f^2 drawn at random, the six squared
     distances computed from them), its 42 coefficients by the reference's formulas;
  2. Groebner basis (grevlex) -> the quotient-ring basis B (10 standard monomials) ;
  3. for an action variable x: the monomials R = x B \\ B must be reduced.  All monomial multiples of the
     four equations up to a degree bound form a matrix [E | R | B] (excess, reducible, basis
     columns); the template works iff rank[E | R] - rank[E] = |R|;
  4. rows are removed greedily while that holds, empty columns dropped.

It prints, per action variable and degree bound, the size of the pruned template, its non-zeros, how
many of its pivots can be template constants, and a structural multiply-add count of a sparse
elimination (minimum-degree order on the pattern), next to the same count for the reference's
78 x 88 table.  Nothing here is used by the library; it answers whether a smaller template would
leave the elimination kernel with substantially fewer multiply-adds.

Result (seed 1, profiles/r3_template_study.txt): with plain monomial multiples the action of zd needs
multiples up to total degree 8; greedy pruning leaves 133 x 158 (1 388 non-zeros, 4 333 structural
multiply-adds) where the reference's table has 78 x 88 (736 non-zeros, 3 861 in the same model).  The
reference's table is not a set of plain multiples: 39 of its 78 rows lack entries their polynomial
would have (monomials such as w^2 and w^2 zd have no column): columns of excess monomials may be
dropped as long as the square system stays regular, because the combination of rows that reduces a
monomial exactly also solves the system without those columns (checked modulo p on the reference's
table: the planted solution is an eigenvector of its action matrix).  With the reference's own basis
(1, zd, zc, zb, w, zd^2, zd zc, zd zb, zd w, zc^2) multiples up to degree 7 suffice; pruning leaves
107 rows, squared by dropping columns 107 x 117, 2 835 structural multiply-adds against 3 861: a
quarter fewer in this model, at the price of a new pivoting structure and unknown conditioning, and
no factor.  The kernel keeps the reference's table.

Usage: tools/template_study.py [seed]
"""
import itertools
import os
import sys
import time

import numpy as np
import sympy as sp

P = 30097
VARS = sp.symbols("zb zc zd w")
zb, zc, zd, w = VARS


def inv(a):
    return pow(int(a) % P, P - 2, P)


def instance(seed):
    """The 42 coefficients of P4pf.cpp:1435-1519 for a random consistent problem, mod P."""
    rng = np.random.default_rng(seed)
    a1, a2, b1, b2, c1, c2, d1, d2 = (int(x) for x in rng.integers(1, P, 8))
    tzb, tzc, tzd, tw = (int(x) for x in rng.integers(2, P, 4))

    def pt(x, y, z):          # camera-frame point z (x, y, f): squared norms only need w = f^2
        return (z * x % P, z * y % P, z)

    A, B, C, D = pt(a1, a2, 1), pt(b1, b2, tzb), pt(c1, c2, tzc), pt(d1, d2, tzd)

    def dist2(X, Y):
        return ((X[0] - Y[0]) ** 2 + (X[1] - Y[1]) ** 2 + tw * (X[2] - Y[2]) ** 2) % P

    glab, glac, glad, glbc, glbd, glcd = dist2(A, B), dist2(A, C), dist2(A, D), dist2(B, C), dist2(B, D), dist2(C, D)
    h = inv(2)
    t1 = inv(glad)
    t2, t3, t4 = t1 * glbc % P, glab * t1 % P, glac * t1 % P
    t5 = (t2 - t3 - t4) % P
    t9, t13 = d2 * d2 % P, d1 * d1 % P
    t11, t12, t14, t16 = t3 * t9 % P, t4 * t9 % P, t4 * t13 % P, t3 * t13 % P
    t24 = (-a2 * b2 - b1 * a1) % P
    t27 = (-c1 * a1 - c2 * a2) % P
    t28 = a1 * t1 % P
    t32 = t1 * d1 % P
    t33 = a1 * glac * t32 % P
    t34 = a2 * d2 % P
    t35 = t4 * t34 % P
    t37 = a1 * glab * t32 % P
    t39 = t3 * t34 % P
    t41, t42 = a2 * a2 % P, a1 * a1 % P
    t43 = t4 * t41 % P
    t46 = t42 * glac * t1 % P
    t51 = t42 * glab * t1 % P
    t53 = t42 * t1 % P
    t56 = t3 * t41 % P
    t59, t60 = c1 * c1 % P, c2 * c2 % P
    t67 = t1 * glbd % P
    t84 = glcd * t1 % P
    k = [0] * 42
    k[0] = 1; k[1] = t5 * h; k[2] = -1; k[3] = -1; k[4] = c2 * b2 + c1 * b1; k[5] = -t5
    k[6] = (t2 * t9 - t11 - t12 - t14 + t2 * t13 - t16) * h
    k[7] = 1 + (-t4 - t3 + t2) * h
    k[8] = t24; k[9] = t27
    k[10] = -t28 * glbc * d1 + t33 + t35 + t37 - t2 * t34 + t39
    k[11] = t41 + t42 + (-t43 - t46 + t2 * t41 - t51 + t53 * glbc - t56) * h
    k[12] = 1; k[13] = -t4; k[14] = -2; k[15] = t59 + t60; k[16] = 2 * t4; k[17] = -t14 - t12; k[18] = -t4 + 1
    k[19] = 2 * t27; k[20] = 2 * t33 + 2 * t35; k[21] = -t43 + t41 + t42 - t46
    k[22] = 1; k[23] = (t67 - 1 - t3) * h; k[24] = -1; k[25] = t3 - t67; k[26] = d2 * b2 + b1 * d1
    k[27] = (-t11 - t16 + t67 * t9 + t67 * t13 - t9 - t13) * h
    k[28] = (-t3 + t67 + 1) * h; k[29] = t24; k[30] = -t28 * glbd * d1 + t37 + t39 - t67 * t34
    k[31] = (t67 * t41 + t53 * glbd - t56 - t51 + t42 + t41) * h
    k[32] = 1; k[33] = (-t4 + t84 - 1) * h; k[34] = -1; k[35] = t4 - t84; k[36] = c1 * d1 + c2 * d2
    k[37] = (t84 * t13 - t12 - t14 - t13 - t9 + t84 * t9) * h
    k[38] = (-t4 + 1 + t84) * h; k[39] = t27; k[40] = t35 + t33 - t84 * t34 - glcd * a1 * t32
    k[41] = (t42 + t41 - t43 + glcd * t42 * t1 - t46 + t84 * t41) * h
    k = [int(x) % P for x in k]
    mono = [[w * zb * zc, w * zd ** 2, w * zb, w * zc, zb * zc, w * zd, zd ** 2, w, zb, zc, zd, 1],
            [w * zc ** 2, w * zd ** 2, w * zc, zc ** 2, w * zd, zd ** 2, w, zc, zd, 1],
            [w * zb * zd, w * zd ** 2, w * zb, w * zd, zb * zd, zd ** 2, w, zb, zd, 1],
            [w * zc * zd, w * zd ** 2, w * zc, w * zd, zc * zd, zd ** 2, w, zc, zd, 1]]
    eqs, pos = [], 0
    for ms in mono:
        eqs.append(sum(k[pos + i] * m for i, m in enumerate(ms)))
        pos += len(ms)
    sol = {zb: tzb, zc: tzc, zd: tzd, w: tw}
    assert all(int(e.subs(sol)) % P == 0 for e in eqs), "the planted solution must satisfy the equations"
    const = [i for i in range(42) if i in (0, 2, 3, 12, 14, 22, 24, 32, 34)]
    return eqs, const


def exps(m):
    return tuple(sp.Poly(m, *VARS).monoms()[0]) if m != 1 else (0, 0, 0, 0)


def poly_terms(e):
    return {mon: int(c) % P for mon, c in sp.Poly(e, *VARS, modulus=P).terms()}


def rank_mod(M):
    """Rank of an integer matrix mod P (Gaussian elimination, numpy rows)."""
    M = M.copy() % P
    r = 0
    rows, cols = M.shape
    for c in range(cols):
        piv = None
        for i in range(r, rows):
            if M[i, c]:
                piv = i
                break
        if piv is None:
            continue
        if piv != r:
            M[[r, piv]] = M[[piv, r]]
        M[r] = M[r] * inv(M[r, c]) % P
        nz = np.nonzero(M[:, c])[0]
        for i in nz:
            if i != r:
                M[i] = (M[i] - M[i, c] * M[r]) % P
        r += 1
        if r == rows:
            break
    return r


def build(eqs, action, B, deg):
    """Rows: all monomial multiples of the equations with total degree <= deg."""
    terms = [poly_terms(e) for e in eqs]
    edeg = [max(sum(m) for m in t) for t in terms]
    rows = []
    for ei, t in enumerate(terms):
        for d in range(deg - edeg[ei] + 1):
            for mult in itertools.product(range(d + 1), repeat=4):
                if sum(mult) != d:
                    continue
                rows.append((ei, mult, {tuple(a + b for a, b in zip(m, mult)): c for m, c in t.items()}))
    Bset = set(B)
    ai = VARS.index(action)
    R = sorted({tuple(x + (1 if i == ai else 0) for i, x in enumerate(b)) for b in B} - Bset)
    monos = set()
    for _, _, r in rows:
        monos |= set(r)
    E = sorted(monos - Bset - set(R), key=lambda m: (-sum(m), m))
    cols = E + R + list(B)
    index = {m: i for i, m in enumerate(cols)}
    M = np.zeros((len(rows), len(cols)), dtype=np.int64)
    for i, (_, _, r) in enumerate(rows):
        for m, c in r.items():
            M[i, index[m]] = c
    return rows, cols, M, len(E), len(R)


def works(M, nE, nR):
    return rank_mod(M[:, :nE + nR]) - rank_mod(M[:, :nE]) == nR


def prune(rows, M, nE, nR):
    """Greedy removal of rows while the template still works: blocks of rows first (halved when a block
    cannot go as a whole), highest-degree multiples first -- they bring the most excess monomials."""
    keep = sorted(range(len(rows)), key=lambda i: -sum(rows[i][1]))

    def shrink(block):
        nonlocal keep
        if not block:
            return
        trial = [j for j in keep if j not in set(block)]
        if works(M[trial], nE, nR):
            keep = trial
            return
        if len(block) == 1:
            return
        shrink(block[:len(block) // 2])
        shrink(block[len(block) // 2:])

    order = list(keep)
    for i in range(0, len(order), 32):
        shrink(order[i:i + 32])
    return sorted(keep)


def structural_cost(pattern, ncols_elim):
    """Multiply-adds of a sparse elimination of the first ncols_elim columns on a 0/1 pattern (rows x
    cols): at every step the (row, column) with the smallest Markowitz count among the remaining
    columns; fill-in is tracked on the pattern."""
    A = [set(np.nonzero(r)[0]) for r in pattern]
    live_rows = set(range(len(A)))
    cols_left = set(range(ncols_elim))
    mads = 0
    while cols_left:
        best = None
        colrows = {c: [i for i in live_rows if c in A[i]] for c in cols_left}
        for c, rs in colrows.items():
            if not rs:
                continue
            for i in rs:
                cost = (len(rs) - 1) * (len(A[i]) - 1)
                if best is None or cost < best[0]:
                    best = (cost, i, c)
        if best is None:
            break
        _, pi, pc = best
        for i in colrows[pc]:
            if i == pi:
                continue
            mads += len(A[pi]) - 1
            A[i] = (A[i] | A[pi]) - {pc}
        live_rows.discard(pi)
        cols_left.discard(pc)
    return mads


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    eqs, const = instance(seed)
    t0 = time.time()
    G = sp.groebner(eqs, *VARS, modulus=P, order="grevlex")
    lead = [sp.Poly(g, *VARS, modulus=P).monoms(order="grevlex")[0] for g in G.exprs]
    # standard monomials: not divisible by any leading monomial
    B = [m for m in itertools.product(range(8), repeat=4) if sum(m) <= 7 and not any(all(a >= b for a, b in zip(m, l)) for l in lead)]
    print("Groebner basis: %d polynomials, %.1f s; quotient ring dimension %d" % (len(G.exprs), time.time() - t0, len(B)))
    print("standard monomials (zb, zc, zd, w exponents):", B)
    for action in (zd, w, zb, zc):
        for deg in (5, 6, 7, 8):
            rows, cols, M, nE, nR = build(eqs, action, B, deg)
            if not works(M, nE, nR):
                print("action %s, multiples up to degree %d: %d x %d, does not reduce x B" % (action, deg, M.shape[0], M.shape[1]))
                continue
            keep = prune(rows, M, nE, nR)
            Mk = M[keep]
            used = np.nonzero(Mk.any(axis=0))[0]
            Mk = Mk[:, used]
            ne = int((used < nE).sum())
            nnz = int((Mk != 0).sum())
            per_eq = [sum(1 for i in keep if rows[i][0] == e) for e in range(4)]
            mads = structural_cost(Mk != 0, Mk.shape[1] - len(B))
            print("action %s, multiples up to degree %d: all %d x %d -> pruned %d x %d (excess %d, reducible %d, basis %d), nnz %d, rows per equation %s, "
                  "structural multiply-adds of the elimination %d" % (action, deg, M.shape[0], M.shape[1], Mk.shape[0], Mk.shape[1], ne, nR, len(B), nnz, per_eq, mads))
            break
    # the reference's table for comparison, same cost model
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    try:
        import gen_p4pf_elim as gen  # noqa: F401
        tab = gen.load_template() if hasattr(gen, "load_template") else None
    except Exception:
        tab = None
    if tab is not None:
        pat = np.zeros((78, 88), dtype=bool)
        for r, c, _ in tab:
            pat[r, c] = True
        print("reference template 78 x 88, nnz %d: structural multiply-adds of the elimination (same model) %d" % (pat.sum(), structural_cost(pat, 78)))


if __name__ == "__main__":
    main()
