#!/usr/bin/env python3
"""Generates pose_estimation_amd/csrc/p4pf_elim_gen.h: the wave-cooperative, register-resident
elimination of the 78x88 P4Pf template (reference P4pf.cpp:959-1109).

What the reference computes: Mr = M_left^-1 M_right (78x78 partial-pivot LU, 10 right-hand
sides), of which only rows 70..74 feed the action matrix.  Equivalent elimination used here
(TWO systems per 64-lane wave, one per 32-lane half; one matrix ROW per lane, columns = named
registers):

  stage 1  49 eliminations whose pivot is one of the template's constant entries (1, -1, -2).
           The order is chosen so that every pivot row is still in its original form when it
           is used: its entries are plain template coefficients, the same for all lanes of a
           half, so no cross-lane traffic is needed and the pivot rows never occupy a lane.
           Only the 29 rows that survive stage 1 are materialised (lanes 0..28 of each half).
  stage 2  the remaining 24 columns, dynamic partial pivoting over the 29 surviving rows
           (arg-max inside each half, pivot row broadcast with ds_bpermute).
  tail     the 5 rows left over hold a 5 x 15 system in columns 70..74 | 78..87; it is written
           out and solved per lane by the consumer (p4pf_reduce_tail), where it costs no
           cross-lane traffic.

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
LDS_SHARE = int(os.environ.get("P4PF_ELIM_LDS_SHARE", "5"))   # of every 8 pivot-row broadcasts, how many go through the LDS crossbar
KEEP = list(range(70, 75))          # columns whose rows of Mr are needed
RHS = list(range(78, 88))
QUAD_CHUNK = int(os.environ.get("P4PF_ELIM_QUAD_CHUNK", "8"))   # pivot-row entries in flight per group of broadcasts (four-system layout)


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

    def harm(targets, pc):
        # number of targets that are still unmodified and could serve as a constant pivot later
        n = 0
        for t in targets:
            if t in modified:
                continue
            if any(e[0] == "c" and cc != pc and cc not in done and cc not in KEEP and cc < 78 for cc, e in rows[t].items()):
                n += 1
        return n

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
                key = (r in modified, harm([x for x in rs if x != r], c), cost, c, r)
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
    assert not any(st["materialised"] for st in steps), "a pivot row was modified before use"
    lanes_rows = sorted(set(mat_rows) | set(survivors))
    assert lanes_rows == survivors and len(lanes_rows) <= 32
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
    w("// Wave-cooperative elimination of the 78x88 P4Pf template: two systems per wave (one per")
    w("// 32-lane half), one matrix row per lane, one named register per column.  See the generator")
    w("// for the schedule.")
    w("#pragma once")
    w("namespace pose_amd {")
    w(f"constexpr int P4PF_ELIM_LANES = {nl};   // lanes of each half that hold a matrix row")
    # init table: for each lane of a half and column the coefficient id (42 = structural zero)
    tbl = np.full((32, 88), 42, dtype=np.int64)
    for r in lanes_rows:
        for c, g in sched["orig"][r].items():
            tbl[lane_of[r], c] = g
    w("// coefficient id of entry (lane & 31, column) of the initial matrix; 42 = structural zero")
    w("__constant__ unsigned char c_elim_init[88][32] = {")
    for c in range(88):
        w("  {" + ",".join(str(int(v)) for v in tbl[:, c]) + "},")
    w("};")
    w("")
    w("// cf: LDS array of 43 doubles holding the 42 template coefficients of this lane's system and")
    w("// 0.0 at index 42 (the pointer differs between the two halves of the wave).")
    w("// Returns 0, or -1 when a pivot is exactly zero (same value in all lanes of a half).")
    w("// tail5x15: the 5 rows that survive stage 2, columns 70..74 | 78..87; entry (r, j) goes to")
    w("// tail5x15[(r * 15 + j) * stride], written by single lanes of each half when wr is set.")
    w("__device__ __forceinline__ int p4pf_elim_wave(const double *cf, double *tail5x15, size_t stride, bool wr)")
    w("{")
    w("    const int lane = (int)(threadIdx.x & 63);")
    w("    const int row = lane & 31;")
    w("    const bool upper = lane >= 32;")
    w("    int bad = 0;")
    initialised = set()
    kdone = set()

    def need(c):
        if c not in initialised:
            initialised.add(c)
            w(f"    double r{c} = cf[c_elim_init[{c}][row]];")

    def kexpr(g):
        if g in CONST:
            return repr(CONST[g])
        if g not in kdone:
            kdone.add(g)
            w(f"    const double k{g} = cf[{g}];")
        return f"k{g}"

    # Pivot-row broadcast, per 32-lane half.  Two routes share the work so that neither pipe is the
    # bottleneck: the LDS crossbar (ds_bpermute, both halves in one instruction) and the vector
    # ALU (v_readlane of each half's pivot lane into scalar registers, then a half-masked fma).
    def bcast_fma(cols, chunk_size):
        for i0 in range(0, len(cols), chunk_size):
            chunk = cols[i0:i0 + chunk_size]
            nl_ = (len(chunk) * LDS_SHARE + 7) // 8 if LDS_SHARE < 8 else len(chunk)
            via_lds, via_alu = chunk[:nl_], chunk[nl_:]
            for c2 in via_lds:
                w(f"        const double e{c2} = half_bcast(r{c2}, L);")
            for c2 in via_alu:
                w(f"        const double a{c2} = wave_bcast(r{c2}, L0), b{c2} = wave_bcast(r{c2}, L1);")
            for c2 in via_lds:
                w(f"        r{c2} = fma(m, e{c2}, r{c2});")
            if via_alu:
                # two half-masked FMAs per column with the scalar operand read directly (the
                # compiler would select between the halves with moves + v_cndmask instead)
                lines = ['"s_mov_b32 exec_hi, 0\\n\\t"']
                lines += [f'"v_fmac_f64 %[r{c2}], %[a{c2}], %[m]\\n\\t"' for c2 in via_alu]
                lines += ['"s_mov_b32 exec_lo, 0\\n\\t"', '"s_mov_b32 exec_hi, -1\\n\\t"']
                lines += [f'"v_fmac_f64 %[r{c2}], %[b{c2}], %[m]\\n\\t"' for c2 in via_alu]
                lines += ['"s_mov_b32 exec_lo, -1"']
                w("        asm volatile(" + "\n                     ".join(lines))
                w("                     : " + ", ".join(f'[r{c2}] "+v"(r{c2})' for c2 in via_alu))
                w("                     : [m] \"v\"(m), " + ", ".join(f'[a{c2}] "s"(a{c2}), [b{c2}] "s"(b{c2})' for c2 in via_alu) + ");")

    # ---- stage 1
    for si, st in enumerate(sched["steps"]):
        pc, pv = st["col"], st["pv"]
        need(pc)
        for c2 in st["pattern"]:
            need(c2)
        ops = [(c2, kexpr(sched["orig"][st["row"]][c2])) for c2 in st["pattern"]]
        w(f"    {{ // stage 1 step {si}: column {pc}, pivot row {st['row']} (original form)")
        w(f"        const double m = r{pc} * {repr(-1.0 / pv)};")
        for c2, ke in ops:
            w(f"        r{c2} = fma(m, {ke}, r{c2});")
        w("    }")
    # ---- stage 2
    surv_lanes = sorted(lane_of[r] for r in sched["survivors"])
    mask = 0
    for L in surv_lanes:
        mask |= 1 << L
    w(f"    bool active = ((0x{mask:x}u >> row) & 1u) != 0;   // the {len(surv_lanes)} surviving rows")
    for si, st in enumerate(sched["stage2"]):
        c = st["col"]
        need(c)
        for c2 in st["pattern"]:
            need(c2)
        w(f"    {{ // stage 2 step {si}: column {c}")
        w(f"        const int L = half_argmax_abs(r{c}, active);")
        w("        const int L0 = __builtin_amdgcn_readfirstlane(L), L1 = __builtin_amdgcn_readlane(L, 32);")
        w(f"        const double pv = half_bcast(r{c}, L);")
        w("        bad |= pv == 0.0;")
        w(f"        const double m = (active && lane != L) ? r{c} * neg_rcp(pv) : 0.0;")
        bcast_fma(st["pattern"], 8)
        w("        active = active && lane != L;")
        w("    }")
    # ---- tail: the five rows still active hold the reduced 5 x 15 system (columns KEEP | RHS);
    # its Gauss-Jordan solve needs no cross-lane traffic when done per lane by the consumer.
    for c in KEEP + RHS:
        need(c)
    w("    {")
    w("        const unsigned long long act = __ballot(active);")
    w("        const unsigned mine = upper ? (unsigned)(act >> 32) : (unsigned)act;")
    w("        bad |= __popc(mine) != 5;")
    w("        const int rank = __popc(mine & ((1u << row) - 1u));")
    w("        if (wr && active && rank < 5) {")
    w("            double *o = tail5x15 + (size_t)(rank * 15) * stride;")
    for j, c in enumerate(KEEP + RHS):
        w(f"            o[{j} * stride] = r{c};")
    w("        }")
    w("    }")
    w("    return bad ? -1 : 0;")
    w("}")
    w("} // namespace pose_amd")
    txt = "\n".join(o) + "\n"
    return txt


def emit_quad(sched):
    """Four systems per wave: one per row of 16 lanes, two matrix rows per lane (row position
    p = 16 * slot + lane % 16), one named register per (column, slot).  Every cross-lane
    instruction then serves four systems instead of two.

    Initial entries: register (column, slot) of a lane starts as coefficient id(position, column)
    of its system (42 = structural zero).  The ids come from a byte table kept in LDS and laid out
    [slot][position][order of first use], so that one 16-byte read gives a lane the ids of the
    next 16 registers it initialises."""
    o = []
    w = o.append
    order = [[], []]
    w("// Same elimination with FOUR systems per wave (one per row of 16 lanes, two matrix rows per")
    w("// lane).  cf points at the coefficients of this lane's system, tail5x15 at its output column,")
    w("// tab at the LDS copy of c_elim_quad_tab.")
    w("__device__ __forceinline__ int p4pf_elim_quad(const double *cf, const uint4 *tab, double *tail5x15, size_t stride, bool wr)")
    w("{")
    w("    const int lane = (int)(threadIdx.x & 63);")
    w("    const int rl = lane & 15;")
    w("    const int sh = lane & 48;")
    w("    int bad = 0;")

    def need(c, s):
        if c in order[s]:
            return
        i = len(order[s])
        order[s].append(c)
        if i % 16 == 0:
            w(f"    const uint4 t{s}_{i // 16} = tab[({16 * s} + rl) * QUAD_TAB_VEC + {i // 16}];")
        comp = "xyzw"[(i % 16) // 4]
        w(f"    double r{c}_{s} = cf[__builtin_amdgcn_ubfe(t{s}_{i // 16}.{comp}, {8 * (i % 4)}, 8)];")

    for s in (0, 1):
        kdone = set()

        def kexpr(g):
            if g in CONST:
                return repr(CONST[g])
            if g not in kdone:
                kdone.add(g)
                w(f"    const double k{g}_{s} = cf[{g}];")
            return f"k{g}_{s}"

        w(f"    // ---- stage 1, rows {16 * s}..{16 * s + 15}")
        for si, st in enumerate(sched["steps"]):
            pc, pv = st["col"], st["pv"]
            need(pc, s)
            for c2 in st["pattern"]:
                need(c2, s)
            ops = [(c2, kexpr(sched["orig"][st["row"]][c2])) for c2 in st["pattern"]]
            w(f"    {{ // step {si}: column {pc}")
            w(f"        const double m = r{pc}_{s} * {repr(-1.0 / pv)};")
            for c2, ke in ops:
                w(f"        r{c2}_{s} = fma(m, {ke}, r{c2}_{s});")
            w("    }")
        if s == 0:
            w('    asm volatile("" ::: "memory");')
    lane_of = sched["lane_of"]
    surv = sorted(lane_of[r] for r in sched["survivors"])
    m0 = sum(1 << L for L in surv if L < 16)
    m1 = sum(1 << (L - 16) for L in surv if L >= 16)
    w(f"    bool act0 = ((0x{m0:x}u >> rl) & 1u) != 0, act1 = ((0x{m1:x}u >> rl) & 1u) != 0;")
    for si, st in enumerate(sched["stage2"]):
        c = st["col"]
        for s in (0, 1):
            need(c, s)
            for c2 in st["pattern"]:
                need(c2, s)
        w(f"    {{ // stage 2 step {si}: column {c}")
        w("        bool sl;")
        w(f"        const int L = row16_argmax_abs(r{c}_0, act0, r{c}_1, act1, sl);")
        w(f"        const double pv = half_bcast(sl ? r{c}_1 : r{c}_0, L);")
        w("        bad |= pv == 0.0;")
        w("        const double nr = neg_rcp(pv);")
        w("        const bool p0 = lane == L && !sl, p1 = lane == L && sl;")
        w(f"        const double m0 = (act0 && !p0) ? r{c}_0 * nr : 0.0;")
        w(f"        const double m1 = (act1 && !p1) ? r{c}_1 * nr : 0.0;")
        pat = st["pattern"]
        for i0 in range(0, len(pat), QUAD_CHUNK):
            chunk = pat[i0:i0 + QUAD_CHUNK]
            for c2 in chunk:
                w(f"        const double e{c2} = half_bcast(sl ? r{c2}_1 : r{c2}_0, L);")
            for c2 in chunk:
                w(f"        r{c2}_0 = fma(m0, e{c2}, r{c2}_0);")
                w(f"        r{c2}_1 = fma(m1, e{c2}, r{c2}_1);")
        w("        act0 = act0 && !p0;")
        w("        act1 = act1 && !p1;")
        w("    }")
    for c in KEEP + RHS:
        need(c, 0)
        need(c, 1)
    w("    {")
    w("        const unsigned a0 = (unsigned)(__ballot(act0) >> sh) & 0xffffu, a1 = (unsigned)(__ballot(act1) >> sh) & 0xffffu;")
    w("        const int n0 = __popc(a0);")
    w("        bad |= n0 + __popc(a1) != 5;")
    w("        const unsigned below = (1u << rl) - 1u;")
    w("        const int rank0 = __popc(a0 & below), rank1 = n0 + __popc(a1 & below);")
    for s in (0, 1):
        w(f"        if (wr && act{s} && rank{s} < 5) {{")
        w(f"            double *o = tail5x15 + (size_t)(rank{s} * 15) * stride;")
        for j, c in enumerate(KEEP + RHS):
            w(f"            o[{j} * stride] = r{c}_{s};")
        w("        }")
    w("    }")
    w("    return bad ? -1 : 0;")
    w("}")
    w("} // namespace pose_amd")
    # the id table, in the order of first use found above
    tbl = np.full((32, 88), 42, dtype=np.int64)
    for r in sched["lanes_rows"]:
        for c, g in sched["orig"][r].items():
            tbl[lane_of[r], c] = g
    nvec = (max(len(order[0]), len(order[1])) + 15) // 16
    head = ["", "namespace pose_amd {",
            f"constexpr int QUAD_TAB_VEC = {nvec};   // 16-byte vectors of coefficient ids per (slot, position)",
            "// coefficient ids [slot * 16 + position][order of first use] of the initial matrix; 42 = structural zero",
            f"__constant__ __attribute__((aligned(16))) unsigned char c_elim_quad_tab[32 * {nvec * 16}] = {{"]
    for s in (0, 1):
        for pos in range(16):
            ids = [int(tbl[16 * s + pos, c]) for c in order[s]]
            ids += [42] * (nvec * 16 - len(ids))
            head.append("  " + ",".join(str(v) for v in ids) + ",")
    head.append("};")
    return "\n".join(head + o) + "\n"


if __name__ == "__main__":
    s = build_schedule()
    n1 = len(s["steps"])
    print("stage 1: %d constant-pivot steps, %d rows in lanes" % (n1, len(s["lanes_rows"])))
    print("stage 1 fma count:", sum(len(st["pattern"]) for st in s["steps"]))
    print("stage 2: %d columns, broadcast+fma count %d" % (len(s["stage2"]), sum(len(st["pattern"]) for st in s["stage2"])))
    out = os.path.join(ROOT, "pose_estimation_amd", "csrc", "p4pf_elim_gen.h")
    open(out, "w").write(emit(s) + emit_quad(s))
    print("wrote", out)
