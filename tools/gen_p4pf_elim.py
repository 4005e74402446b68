#!/usr/bin/env python3
"""Generates pose_estimation_amd/csrc/p4pf_elim_gen.h: the wave-cooperative, register-resident
elimination of the 78x88 P4Pf template (reference P4pf.cpp:959-1109).

What the reference computes: Mr = M_left^-1 M_right (78x78 partial-pivot LU, 10 right-hand
sides), of which only rows 70..74 feed the action matrix.  Equivalent elimination used here
(FOUR systems per 64-lane wave, one per row of 16 lanes; TWO matrix rows per lane, row position
p = 16 * slot + lane % 16; one named register per (column, slot)):

  stage 1  49 eliminations whose pivot is one of the template's constant entries (1, -1, -2).
           The order is chosen so that every pivot row is still in its original form when it
           is used: its entries are plain template coefficients, the same for all lanes of a
           system, so no cross-lane traffic is needed and the pivot rows never occupy a lane.
           Only the 29 rows that survive stage 1 are materialised.
  stage 2  the remaining 24 columns, dynamic partial pivoting over the 29 surviving rows
           (arg-max inside each row of 16 lanes with DPP; the pivot lane publishes its row
           through LDS memory, everybody reads it back as a broadcast).
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
KEEP = list(range(70, 75))          # columns whose rows of Mr are needed
RHS = list(range(78, 88))
QUAD_BCAST = os.environ.get("P4PF_ELIM_QUAD_BCAST", "lds")   # pivot-row broadcast: "lds" (write + broadcast read) or "bpermute"
SKIP = os.environ.get("P4PF_ELIM_SKIP", "")    # timing experiments only: "stage1" or "stage2" leaves that stage out
PREFETCH = int(os.environ.get("P4PF_ELIM_PREFETCH", "0"))     # stage-1 steps between an LDS read and its first use (0..8 measured the same)
ROW_BYTES = 320          # LDS bytes per published pivot row of a block (40 doubles); 320 mod 256 = 64 keeps the rows of a block on different banks
TWO_ROUNDS = int(os.environ.get("P4PF_ELIM_TWO_ROUNDS", "0"))   # publish slot-0 pivots and slot-1 pivots in two rounds of stores without selects
SPEC_RCP = int(os.environ.get("P4PF_ELIM_SPEC_RCP", "0"))   # every lane inverts its own candidates while the pivot search runs; the pivot lane's reciprocal is broadcast
BLOCK = int(os.environ.get("P4PF_ELIM_BLOCK", "4"))      # stage-2 steps per block of delayed updates (0: one pivot row published per step)
QUAD_CHUNK = int(os.environ.get("P4PF_ELIM_QUAD_CHUNK", "16"))   # pivot-row entries in flight per group of broadcasts


def load_template():
    txt = open(os.path.join(ROOT, "pose_estimation_amd", "csrc", "p4pf_template.h")).read()
    out = {}
    for name in ("ROW", "COL", "COEF"):
        m = re.search(r"P4PF_TPL_%s\[\d+\] = \{([^}]*)\}" % name, txt)
        out[name] = [int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    return list(zip(out["ROW"], out["COL"], out["COEF"]))


def _build_schedule(forced=None):
    """forced: optional list of (pivot row, column) pairs that replaces the greedy choice of
    stage-1 pivots (used to re-order the pivots the greedy pass found)."""
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
        if forced is not None:
            if len(steps) == len(forced):
                break
            pr, pc = forced[len(steps)]
            assert rows[pr][pc][0] == "c" and pr not in modified
            best = (None, pr, pc)
        for c in ([] if forced is not None else elim_cols):
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
    # Positions (p = 16 * slot + lane % 16): a stage-2 step whose only two candidate rows sit in
    # the two slots of ONE lane needs no cross-lane traffic at all, so the leading two-candidate
    # steps get their rows co-located by swapping positions.
    used = set()
    for st in stage2:
        if len(st["cand"]) != 2:
            break
        ra, rb = st["cand"]
        pa, pb = lane_of[ra], lane_of[rb]
        if {pa, pb} & used:
            break
        if abs(pa - pb) != 16:
            lo_row, hi_row = (ra, rb) if pa < pb else (rb, ra)
            # move hi_row's partner into the other slot of hi_row's lane
            target = lane_of[hi_row] - 16 if lane_of[hi_row] >= 16 else lane_of[hi_row] + 16
            if target >= len(lanes_rows) or target in used:
                break
            other = next(r for r, p in lane_of.items() if p == target)
            lane_of[other], lane_of[lo_row] = lane_of[lo_row], target
        used |= {lane_of[ra], lane_of[rb]}
        st["in_lane"] = True
    return dict(orig=orig, steps=steps, survivors=survivors, lanes_rows=lanes_rows, lane_of=lane_of, stage2=stage2)


# Order of the stage-1 pivots, as indices into the order the greedy pass finds them in.  Any
# order is valid in which a pivot row is used before a column it has an entry in is eliminated
# (it must still be in its original form); all of them do the same 395 multiply-adds.  This
# one takes the pivot rows point by point (the template's coefficients come in four groups, one
# per correspondence) and was picked by a random search over such orders for the smallest peak
# of live matrix columns + live coefficients: 73 doubles per lane instead of the greedy
# order's 89, which is what lets the kernel keep two rows per lane without spilling.
STAGE1_ORDER = [3, 2, 8, 22, 19, 4, 5, 11, 9, 12, 17, 26, 20, 21, 36, 40, 16, 25, 30, 31, 33, 29, 18, 41, 37, 44, 47, 43, 45,
                1, 23, 10, 24, 6, 7, 27, 28, 13, 34, 0, 38, 15, 39, 35, 42, 14, 32, 46, 48]


def build_schedule():
    greedy = _build_schedule()["steps"]
    assert sorted(STAGE1_ORDER) == list(range(len(greedy)))
    return _build_schedule([(greedy[i]["row"], greedy[i]["col"]) for i in STAGE1_ORDER])


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


def stage2_blocks(sched, block):
    """Partition of the stage-2 steps: the leading in-lane steps stay single; the others are
    taken `block` at a time.  Per block: its steps, its own columns and the trailing columns
    (every column one of its pivot rows can hold, minus the block's own)."""
    st2 = sched["stage2"]
    lead = [i for i, st in enumerate(st2) if st.get("in_lane")]
    rest = [i for i in range(len(st2)) if i not in lead]
    blocks = []
    for b0 in range(0, len(rest), block):
        idx = rest[b0:b0 + block]
        cols = [st2[i]["col"] for i in idx]
        trail = sorted(set().union(*[set(st2[i]["pattern"]) for i in idx]) - set(cols))
        blocks.append(dict(steps=idx, cols=cols, trail=trail))
    return lead, blocks


def simulate_blocked(sched, k, block=4):
    """NumPy model of the blocked stage 2 the kernel runs: inside a block only the block's own
    columns are kept current (panel); the pivot rows' trailing entries are published raw at the
    end of the block, every row folds the block's multipliers into effective ones
    (m' = m L^-1, L = unit lower triangle of the multipliers the pivot rows received inside the
    block) and applies all pivots of the block to its trailing entries at once."""
    lanes_rows = sched["lanes_rows"]
    nl = len(lanes_rows)
    R = np.zeros((nl, 88))
    for li, r in enumerate(lanes_rows):
        for c, g in sched["orig"][r].items():
            R[li, c] = k[g]
    for st in sched["steps"]:
        pc, pv = st["col"], st["pv"]
        m = R[:, pc] / pv
        vals = {c2: k[sched["orig"][st["row"]][c2]] for c2 in st["pattern"]}
        for c2 in st["pattern"]:
            R[:, c2] -= m * vals[c2]
        R[:, pc] = 0.0
    active = np.zeros(nl, bool)
    for r in sched["survivors"]:
        active[sched["lane_of"][r]] = True
    lead, blocks = stage2_blocks(sched, block)
    for i in lead:                                   # single steps, as in simulate()
        st = sched["stage2"][i]
        c = st["col"]
        cand = np.where(active, np.abs(R[:, c]), -1.0)
        L = int(np.argmax(cand))
        if cand[L] <= 0.0:
            return -1, None
        m = np.where(active, R[:, c] / R[L, c], 0.0)
        m[L] = 0.0
        for c2 in st["pattern"]:
            R[:, c2] -= m * R[L, c2]
        R[:, c] = 0.0
        active[L] = False
    for blk in blocks:
        nb = len(blk["cols"])
        M = np.zeros((nl, nb))                      # multipliers (update is  row += m * pivot row)
        piv = []
        for kk, c in enumerate(blk["cols"]):
            cand = np.where(active, np.abs(R[:, c]), -1.0)
            L = int(np.argmax(cand))
            if cand[L] <= 0.0:
                return -1, None
            M[:, kk] = np.where(active, -R[:, c] / R[L, c], 0.0)
            M[L, kk] = 0.0
            for c2 in blk["cols"][kk + 1:]:         # panel: the block's later columns stay current
                R[:, c2] += M[:, kk] * R[L, c2]
            active[L] = False
            piv.append(L)
        P = R[piv][:, blk["trail"]].copy()          # raw trailing rows of the pivots, as published
        Lm = np.array([[M[piv[j], kk] if kk < j else 0.0 for kk in range(nb)] for j in range(nb)])
        Me = M.copy()                               # effective multipliers: back substitution of m' L = m
        for kk in range(nb - 2, -1, -1):
            for j in range(kk + 1, nb):
                Me[:, kk] += Me[:, j] * Lm[j, kk]
        upd = Me @ P
        mask = np.ones(nl, bool)
        mask[piv] = False                           # retired rows are never read again
        R[np.ix_(mask, blk["trail"])] += upd[mask]
        for c in blk["cols"]:
            R[:, c] = 0.0
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
    w("// Wave-cooperative elimination of the 78x88 P4Pf template, FOUR systems per wave (one per row")
    w("// of 16 lanes, two matrix rows per lane); see the generator for the schedule.")
    w("// cf points at the 42 coefficients (+ 0.0 at index 42) of this lane's system in LDS, tail5x15")
    w("// at its output column: entry (r, j) of the 5 x 15 tail system goes to tail5x15[(r * 15 + j)")
    w("// * stride], written when wr is set.  Returns 0, or -1 when a pivot column had no usable entry.")
    w("// tab at the LDS copy of c_elim_tab, pbuf at 16 double2 of LDS scratch private to the")
    w("// lane's row of 16 lanes.")
    w("__device__ __forceinline__ int p4pf_elim_wave(const double *cf, const uint4 *tab, double2 *pbuf, double *tail5x15, size_t stride, bool wr)")
    w("{")
    w("    const int lane = (int)(threadIdx.x & 63);")
    w("    const int rl = lane & 15;")
    w("    const int sh = lane & 48;")
    w("    int bad = 0;")
    w("    const unsigned pbuf_addr = (unsigned)(size_t)pbuf;     // LDS byte address of the row's scratch")

    def need(c, s):
        if c in order[s]:
            return
        i = len(order[s])
        order[s].append(c)
        # id vectors are fetched one vector ahead of their first use
        if i == 0:
            w(f"    const uint4 t{s}_0 = tab[({16 * s} + rl) * ELIM_TAB_VEC];")
        if i % 16 == 0:
            w(f"    [[maybe_unused]] const uint4 t{s}_{i // 16 + 1} = tab[({16 * s} + rl) * ELIM_TAB_VEC + {i // 16 + 1}];")
        comp = "xyzw"[(i % 16) // 4]
        w(f"    double r{c}_{s} = cf[__builtin_amdgcn_ubfe(t{s}_{i // 16}.{comp}, {8 * (i % 4)}, 8)];")

    # Stage 1 is a chain of short dependent steps; its LDS reads (initial entries, pivot-row
    # coefficients) can be issued PREFETCH steps ahead of the step that needs them.
    steps = [] if SKIP in ("stage1", "both") else sched["steps"]
    for s in (0, 1):
        kdone = set()

        def fetch(st):
            need(st["col"], s)
            for c2 in st["pattern"]:
                need(c2, s)
                g = sched["orig"][st["row"]][c2]
                if g not in CONST and g not in kdone:
                    kdone.add(g)
                    w(f"    const double k{g}_{s} = cf[{g}];")

        def kexpr(g):
            return repr(CONST[g]) if g in CONST else f"k{g}_{s}"

        w(f"    // ---- stage 1, rows {16 * s}..{16 * s + 15}")
        if SKIP == "stage1b" and s == 1:
            break
        for st in steps[:PREFETCH]:
            fetch(st)
        for si, st in enumerate(steps):
            if si + PREFETCH < len(steps):
                fetch(steps[si + PREFETCH])
            pc, pv = st["col"], st["pv"]
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
    def emit_single(si, st):
        c = st["col"]
        for s in (0, 1):
            need(c, s)
            for c2 in st["pattern"]:
                need(c2, s)
        w(f"    {{ // stage 2 step {si}: column {c}")
        w("        bool sl, none;")
        w(f"        const int L = row16_argmax_abs(r{c}_0, act0, r{c}_1, act1, sl, none);")
        w(f"        const double pv = lane_bcast(sl ? r{c}_1 : r{c}_0, L);")
        w("        bad |= none;")
        w("        const double nr = neg_rcp(pv);")
        # rows that are retired (the pivot row included, from here on) are never read again, so
        # their multipliers need no masking: whatever they accumulate stays in their own lane
        w(f"        const double m0 = r{c}_0 * nr, m1 = r{c}_1 * nr;")
        pat = st["pattern"]
        if st.get("in_lane"):
            # both candidates live in the two slots of one lane: the pivot row is the lane's own
            # other slot; every other lane has exact zeros in this column, so its m0 = m1 = 0
            for c2 in pat:
                w(f"        const double e{c2} = sl ? r{c2}_1 : r{c2}_0;")
                w(f"        r{c2}_0 = fma(m0, e{c2}, r{c2}_0);")
                w(f"        r{c2}_1 = fma(m1, e{c2}, r{c2}_1);")
        elif QUAD_BCAST == "bpermute":
            for i0 in range(0, len(pat), QUAD_CHUNK):
                chunk = pat[i0:i0 + QUAD_CHUNK]
                for c2 in chunk:
                    w(f"        const double e{c2} = lane_bcast(sl ? r{c2}_1 : r{c2}_0, L);")
                for c2 in chunk:
                    w(f"        r{c2}_0 = fma(m0, e{c2}, r{c2}_0);")
                    w(f"        r{c2}_1 = fma(m1, e{c2}, r{c2}_1);")
        else:
            # the pivot lane of each system publishes its row through LDS memory (16-byte writes by
            # the four pivot lanes, 16-byte broadcast reads by everybody): half as many LDS
            # instructions as ds_bpermute needs, and cheaper ones.  LDS instructions of one wave
            # execute in order, so no wait is needed between the writes and the reads.
            pairs = [pat[i:i + 2] for i in range(0, len(pat), 2)]
            # exec-masked ds_write2_b64 in inline assembly: a divergent `if` would split the kernel's
            # single basic block and wreck register allocation
            def masked_writes(mask, name):
                for g0 in range(0, len(pairs), 6):
                    grp = pairs[g0:g0 + 6]
                    lines = ['"s_mov_b64 exec, %[mask]\\n\\t"']
                    for j, pr in enumerate(grp):
                        lines.append(f'"ds_write_b128 %[addr], %[a{j}] offset:{16 * (g0 + j)}\\n\\t"')
                    lines.append('"s_mov_b64 exec, -1"')
                    w("        asm volatile(" + "\n                     ".join(lines))
                    ops = ", ".join(f'[a{j}] "v"({name(pr)})' for j, pr in enumerate(grp))
                    w(f'                     :: [mask] "s"({mask}), [addr] "v"(pbuf_addr), {ops} : "memory");')
            for pr in pairs:
                a_, b_ = pr[0], pr[-1]
                w(f"        const v2f64 s{a_} = {{sl ? r{a_}_1 : r{a_}_0, sl ? r{b_}_1 : r{b_}_0}};")
            w("        const unsigned long long wm = __builtin_amdgcn_ballot_w64(lane == L);")
            masked_writes("wm", lambda pr: f"s{pr[0]}")
            w("        __builtin_amdgcn_wave_barrier();")
            for i0 in range(0, len(pairs), QUAD_CHUNK // 2):
                chunk = pairs[i0:i0 + QUAD_CHUNK // 2]
                for j, pr in enumerate(chunk):
                    w(f"        const double2 e{pr[0]} = pbuf[{i0 + j}];")
                for pr in chunk:
                    for c2, comp in zip(pr, "xy"):
                        w(f"        r{c2}_0 = fma(m0, e{pr[0]}.{comp}, r{c2}_0);")
                        w(f"        r{c2}_1 = fma(m1, e{pr[0]}.{comp}, r{c2}_1);")
            w("        __builtin_amdgcn_wave_barrier();")
        w("        act0 = act0 && !(lane == L && !sl);")
        w("        act1 = act1 && !(lane == L && sl);")
        w("    }")

    def emit_block(bi, blk):
        """A block of delayed updates: panel steps on the block's own columns, then every pivot
        row's raw trailing entries (and the multipliers it received inside the block) go to LDS in
        one round of stores, and all rows apply the whole block at once."""
        st2 = sched["stage2"]
        cols, trail, nb = blk["cols"], blk["trail"], len(blk["cols"])
        for c in cols + trail:
            need(c, 0)
            need(c, 1)
        npadL = 0 if nb == 1 else 2 * ((nb - 1 + 1) // 2)          # doubles reserved for the L entries at the head of a row
        w(f"    {{ // stage 2 block {bi}: columns {cols}, {len(trail)} trailing columns")
        if TWO_ROUNDS:
            w("        int pk0 = -1, pk1 = -1;           // index inside the block at which this lane's slot-0 / slot-1 row became the pivot")
        else:
            w("        int pk = -1, pk2 = -1;            // this lane's first / second pivot inside the block")
            w("        bool psl = false;                 // slot of the first one")
        for k, c in enumerate(cols):
            w(f"        // step {blk['steps'][k]}: column {c}")
            w(f"        bool sl{k}, none{k};")
            w(f"        const int L{k} = row16_argmax_abs(r{c}_0, act0, r{c}_1, act1, sl{k}, none{k});")
            w(f"        bad |= none{k};")
            if SPEC_RCP:
                w(f"        const double q{k}_0 = neg_rcp(r{c}_0), q{k}_1 = neg_rcp(r{c}_1);")
                w(f"        const double nr{k} = lane_bcast(sl{k} ? q{k}_1 : q{k}_0, L{k});")
            else:
                w(f"        const double nr{k} = neg_rcp(lane_bcast(sl{k} ? r{c}_1 : r{c}_0, L{k}));")
            # retired rows are never read again: no masking of their multipliers
            w(f"        double m{k}_0 = r{c}_0 * nr{k}, m{k}_1 = r{c}_1 * nr{k};")
            for c2 in cols[k + 1:]:
                w(f"        {{ const double e = lane_bcast(sl{k} ? r{c2}_1 : r{c2}_0, L{k});")
                w(f"          r{c2}_0 = fma(m{k}_0, e, r{c2}_0); r{c2}_1 = fma(m{k}_1, e, r{c2}_1); }}")
            w(f"        {{ const bool me = lane == L{k};")
            if TWO_ROUNDS:
                w(f"          pk0 = me && !sl{k} ? {k} : pk0; pk1 = me && sl{k} ? {k} : pk1;")
            else:
                w(f"          pk2 = me && pk >= 0 ? {k} : pk2; psl = me && pk < 0 ? sl{k} : psl; pk = me && pk < 0 ? {k} : pk;")
            w(f"          act0 = act0 && !(me && !sl{k}); act1 = act1 && !(me && sl{k}); }}")
        # ---- publish: one masked round of stores for every lane's first pivot of the block
        items = []                                  # doubles of a row, in order: L entries, padding, trailing entries
        for k in range(nb - 1):
            items.append(("m", k))
        while len(items) < npadL:
            items.append(("z", 0))
        for c in trail:
            items.append(("r", c))
        if len(items) % 2:
            items.append(("z", 0))
        npairs = len(items) // 2
        assert npairs * 16 <= ROW_BYTES, (npairs, ROW_BYTES)

        def val(it, other):
            kind, x = it
            if TWO_ROUNDS:
                return "0.0" if kind == "z" else f"{'mr'[kind == 'r']}{x}_{int(other)}"
            a, b = ("0", "1") if other else ("1", "0")
            if kind == "z":
                return "0.0"
            if kind == "m":
                return f"(psl ? m{x}_{a} : m{x}_{b})"
            return f"(psl ? r{x}_{a} : r{x}_{b})"

        def round_of_stores(tag, other, pkvar):
            for j in range(npairs):
                w(f"        const v2f64 {tag}{j} = {{{val(items[2 * j], other)}, {val(items[2 * j + 1], other)}}};")
            w(f"        const unsigned long long wm{tag} = __builtin_amdgcn_ballot_w64({pkvar} >= 0);")
            w(f"        const unsigned addr{tag} = pbuf_addr + (unsigned)({pkvar} < 0 ? 0 : {pkvar}) * {ROW_BYTES}u;")
            for g0 in range(0, npairs, 6):
                grp = list(range(g0, min(g0 + 6, npairs)))
                lines = ['"s_mov_b64 exec, %[mask]\\n\\t"']
                for j in grp:
                    lines.append(f'"ds_write_b128 %[addr], %[a{j - g0}] offset:{16 * j}\\n\\t"')
                lines.append('"s_mov_b64 exec, -1"')
                w("        asm volatile(" + "\n                     ".join(lines))
                ops = ", ".join(f'[a{j - g0}] "v"({tag}{j})' for j in grp)
                w(f'                     :: [mask] "s"(wm{tag}), [addr] "v"(addr{tag}), {ops} : "memory");')

        if TWO_ROUNDS:
            round_of_stores("pa", False, "pk0")
            round_of_stores("pb", True, "pk1")
        else:
            round_of_stores("pa", False, "pk")
            # a lane whose two rows both became pivots inside this block publishes the second one in a
            # round of its own; taken only when some lane of the wave needs it
            w("        if (__builtin_amdgcn_ballot_w64(pk2 >= 0) != 0ull) {")
            round_of_stores("pb", True, "pk2")
            w("        }")
        w("        __builtin_amdgcn_wave_barrier();")
        # ---- effective multipliers m' = m L^-1 (back substitution), L from the pivot rows' heads
        if nb > 1:
            for j in range(1, nb):
                for h in range((j + 1) // 2):
                    w(f"        const double2 l{j}_{h} = pbuf[{(j * ROW_BYTES) // 16 + h}];")

            def lref(j, k):
                return f"l{j}_{k // 2}.{'xy'[k % 2]}"
            for k in range(nb - 2, -1, -1):
                for j in range(k + 1, nb):
                    w(f"        m{k}_0 = fma(m{j}_0, {lref(j, k)}, m{k}_0); m{k}_1 = fma(m{j}_1, {lref(j, k)}, m{k}_1);")
        # ---- apply the block to the trailing columns
        first = npadL // 2                          # pair index of the first trailing pair
        tpairs = [trail[i:i + 2] for i in range(0, len(trail), 2)]
        CH = 2
        for i0 in range(0, len(tpairs), CH):
            chunk = list(range(i0, min(i0 + CH, len(tpairs))))
            for i in chunk:
                for k in range(nb):
                    w(f"        const double2 p{k}_{i} = pbuf[{(k * ROW_BYTES) // 16 + first + i}];")
            for i in chunk:
                for c2, comp in zip(tpairs[i], "xy"):
                    for k in range(nb):
                        w(f"        r{c2}_0 = fma(m{k}_0, p{k}_{i}.{comp}, r{c2}_0);")
                    for k in range(nb):
                        w(f"        r{c2}_1 = fma(m{k}_1, p{k}_{i}.{comp}, r{c2}_1);")
        w("        __builtin_amdgcn_wave_barrier();")
        w("    }")

    if SKIP not in ("stage2", "both"):
        if BLOCK > 0:
            lead, blocks = stage2_blocks(sched, BLOCK)
            for i in lead:
                emit_single(i, sched["stage2"][i])
            for bi, blk in enumerate(blocks):
                emit_block(bi, blk)
        else:
            for si, st in enumerate(sched["stage2"]):
                emit_single(si, st)
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
    nvec = (max(len(order[0]), len(order[1])) + 15) // 16 + 1     # + 1: the look-ahead fetch of the id vectors
    head = ["// GENERATED by tools/gen_p4pf_elim.py -- do not edit.", "#pragma once", "namespace pose_amd {",
            f"constexpr int ELIM_TAB_VEC = {nvec};   // 16-byte vectors of coefficient ids per (slot, position)",
            f"constexpr int ELIM_PBUF_VEC = {max(BLOCK, 1) * ROW_BYTES // 16 + 1};   // 16-byte vectors of LDS scratch per system: the published pivot rows of a block, + 1 so that the four systems of a wave sit on different banks",
            "// coefficient ids [slot * 16 + position][order of first use] of the initial matrix; 42 = structural zero",
            f"__constant__ __attribute__((aligned(16))) unsigned char c_elim_tab[32 * {nvec * 16}] = {{"]
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
    open(out, "w").write(emit(s))
    print("wrote", out)
