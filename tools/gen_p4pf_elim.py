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
           Only the 29 rows that survive stage 1 are materialised.  A slot runs a step only when
           one of its rows has an entry in the step's column.
  stage 2  the remaining 24 columns, dynamic partial pivoting over the surviving rows (arg-max
           inside each row of 16 lanes with DPP; the pivot lanes publish their rows through LDS
           memory, everybody reads them back as broadcasts), in blocks of delayed updates.
           split phase: the candidate rows of the leading 12 columns fall into two groups that
           never mix (6 rows / 5 columns and 13 rows / 7 columns).  Each group lives in one slot,
           so the two groups are eliminated side by side: every slot searches its own pivot,
           publishes from its own registers (no selection between the slots) and updates only
           its own rows, over its own set of trailing columns.
           joint phase: the last 12 columns have candidates in both slots.  The first of them is
           eliminated over both slots; 16 rows are active after it, slot 0 then has as many retired
           positions as slot 1 has active rows, the latter move into the former (through LDS, under
           EXEC masks) and the other 11 columns are eliminated in slot 0 alone.
           The masked stores write EXEC inside inline assembly and restore it to all ones: the
           generated function must be entered with all 64 lanes active (stated in its header).
  tail     the 5 rows left over hold a 5 x 15 system in columns 70..74 | 78..87; it is written
           out and solved per lane by the consumer (p4pf_reduce_tail), where it costs no
           cross-lane traffic.

The script also contains NumPy interpreters of the schedule (used by
tests/test_p4pf_elim_schedule.py) so that the schedule is validated without a GPU: simulate()
is the plain row-by-row elimination, simulate_kernel() performs exactly the updates the
generated code performs (slots that skip a step, per-slot pivots and trailing columns, blocks).
"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONST = {0: 1.0, 2: -1.0, 3: -1.0, 12: 1.0, 14: -2.0, 22: 1.0, 24: -1.0, 32: 1.0, 34: -1.0}
KEEP = list(range(70, 75))          # columns whose rows of Mr are needed
RHS = list(range(78, 88))
ROW_BYTES = 272          # LDS bytes per published pivot row of a block (at most 34 doubles are used); 68 dwords = 4 modulo the 32 banks of a store, so the up to six rows one round of masked stores writes lie on different banks
BLOCK = int(os.environ.get("P4PF_ELIM_BLOCK", "4"))      # pivot rows published per block of delayed updates
COMPACT = int(os.environ.get("P4PF_ELIM_COMPACT", "1"))  # after the first joint step the 16 active rows are gathered in slot 0 and the other steps run there alone
PBUF_VEC = 112           # 16-byte vectors of LDS scratch per system; every read of a published row is checked against it


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
    assert not any(st["materialised"] for st in steps), "a pivot row was modified before use"
    assert set(modified) <= set(survivors) and len(survivors) <= 32
    # stage 2: static column order with worst-case (any candidate may be pivot) fill
    pat = {r: set(rows[r].keys()) for r in survivors}
    rem_cols = sorted(c for c in set().union(*pat.values()) if c < 78 and c not in KEEP)
    stage2 = []
    cols_left = set(rem_cols)
    while cols_left:
        best = None
        for c in sorted(cols_left):
            cand = [r for r in survivors if c in pat[r]]
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
    sched = dict(orig=orig, steps=steps, survivors=survivors, lanes_rows=survivors, stage2=stage2)
    plan_slots(sched)
    return sched


def plan_slots(sched):
    """Splits stage 2 into the leading steps whose candidate rows form groups that never mix and
    the rest, puts each group into one slot, and marks which slot runs which stage-1 step.

    Adds to sched: slot_of / lane_of (row -> slot, row -> position 16 * slot + index),
    split = [steps of slot 0, steps of slot 1] (indices into stage2, original order),
    joint = the remaining step indices, and per stage-1 step st["slots"] = (bool, bool)."""
    st2 = sched["stage2"]
    group = {}                                   # row -> group id
    members = {}
    n_split = 0
    for i, st in enumerate(st2):
        ids = {group[r] for r in st["cand"] if r in group}
        rows = set(st["cand"]).union(*[members[g] for g in ids]) if ids else set(st["cand"])
        if len(rows) > 16:
            break
        g = min(ids) if ids else i
        for old in ids:
            members.pop(old)
        members[g] = rows
        for r in rows:
            group[r] = g
        n_split = i + 1
    groups = sorted(members.values(), key=lambda m: (len(m), sorted(m)))
    assert len(groups) == 2, "the split phase expects two independent groups of candidate rows"
    slot_of = {}
    for s, m in enumerate(groups):               # smaller group -> slot 0
        for r in m:
            slot_of[r] = s
    # rows outside both groups take part in no split step; they fill slot 0 first (wherever the three
    # spare positions of slot 1 are spent, the slots run the same stage-1 steps)
    free = [r for r in sched["survivors"] if r not in slot_of]
    for r in free:
        s = 0 if sum(1 for v in slot_of.values() if v == 0) < 16 else 1
        slot_of[r] = s
    lane_of = {}
    for s in (0, 1):
        rs = sorted(r for r in sched["survivors"] if slot_of[r] == s)
        assert len(rs) <= 16
        for i, r in enumerate(rs):
            lane_of[r] = 16 * s + i
    sched["slot_of"], sched["lane_of"] = slot_of, lane_of
    sched["split"] = [[i for i in range(n_split) if slot_of[st2[i]["cand"][0]] == s] for s in (0, 1)]
    for s in (0, 1):
        for i in sched["split"][s]:
            assert all(slot_of[r] == s for r in st2[i]["cand"])
    sched["joint"] = list(range(n_split, len(st2)))
    for st in sched["steps"]:
        st["slots"] = tuple(any(slot_of[t] == s for t in st["targets"]) for s in (0, 1))


# Order of the stage-1 pivots, as indices into the order the greedy pass finds them in.  Any
# order is valid in which a pivot row is used before a column it has an entry in is eliminated
# (it must still be in its original form); all of them do the same 395 multiply-adds.  This
# one takes the pivot rows point by point (the template's coefficients come in four groups, one
# per correspondence) and was picked by a random search over such orders for the smallest peak
# of live matrix columns + live coefficients: 73 doubles per lane instead of the greedy
# order's 89.
STAGE1_ORDER = [3, 2, 8, 22, 19, 4, 5, 11, 9, 12, 17, 26, 20, 21, 36, 40, 16, 25, 30, 31, 33, 29, 18, 41, 37, 44, 47, 43, 45,
                1, 23, 10, 24, 6, 7, 27, 28, 13, 34, 0, 38, 15, 39, 35, 42, 14, 32, 46, 48]


def build_schedule():
    greedy = _build_schedule()["steps"]
    assert sorted(STAGE1_ORDER) == list(range(len(greedy)))
    return _build_schedule([(greedy[i]["row"], greedy[i]["col"]) for i in STAGE1_ORDER])


def _chunks(steps, fits):
    """steps cut into the fewest runs that all fit, sizes as equal as possible, the smaller ones first."""
    n = 1
    while True:
        sizes = [len(steps) // n + (1 if i >= n - len(steps) % n else 0) for i in range(n)]
        cut = [steps[sum(sizes[:i]):sum(sizes[:i + 1])] for i in range(n)]
        if all(fits(c) for c in cut):
            return cut
        n += 1


def _fits(st2, idx):
    """Whether the pivot rows of the steps idx (one slot) fit the LDS scratch as one block."""
    cols = [st2[i]["col"] for i in idx]
    trail = set().union(*[set(st2[i]["pattern"]) for i in idx]) - set(cols)
    nb = len(idx)
    n_items = 2 * ((nb - 1 + 1) // 2) + len(trail)
    return n_items * 8 <= ROW_BYTES and ((nb - 1) * ROW_BYTES) // 16 + (32 if n_items > 32 else 16) <= PBUF_VEC


def split_blocks(sched, block):
    """Blocks of the split phase.  Every block serves ONE slot: each round of masked stores costs
    the same whatever the number of rows it publishes (2 + 2, 2 + 2, 1 + 3 pivots in three blocks
    are six rounds and measured 7 % slower than three blocks of one slot each), so a slot's steps
    are cut into as few, nearly equal blocks as the scratch allows (a row of more than 32 doubles
    takes a second read that reaches further) and the blocks of the two slots alternate.
    Per block and slot: step indices, own columns, trailing columns (every column one of the slot's
    pivot rows of the block can hold, minus the block's own), first LDS row."""
    st2 = sched["stage2"]

    def part(idx, base):
        cols = [st2[i]["col"] for i in idx]
        trail = sorted(set().union(*[set(st2[i]["pattern"]) for i in idx]) - set(cols)) if idx else []
        return dict(steps=idx, cols=cols, trail=trail, base=base)

    def fits(idx):
        return _fits(st2, idx)

    chunks = [_chunks(list(sched["split"][s]), fits) for s in (0, 1)]
    first = 0 if len(chunks[0]) > len(chunks[1]) else 1
    blocks = []
    while chunks[0] or chunks[1]:
        for s in (first, 1 - first):
            if chunks[s]:
                idx = chunks[s].pop(0)
                blocks.append([part(idx, 0) if t == s else part([], len(idx) if t == 1 else 0) for t in (0, 1)])
    return blocks


def joint_blocks(sched, block):
    """Blocks of the joint phase: steps, own columns, trailing columns.  With COMPACT only the
    first step is taken over both slots (a block of its own): 16 rows are active after it, and
    they are gathered in slot 0 for the steps that follow (single_blocks)."""
    st2 = sched["stage2"]
    rest = sched["joint"][:1] if COMPACT else sched["joint"]
    blocks = []
    for b0 in range(0, len(rest), block):
        idx = rest[b0:b0 + block]
        cols = [st2[i]["col"] for i in idx]
        trail = sorted(set().union(*[set(st2[i]["pattern"]) for i in idx]) - set(cols))
        blocks.append(dict(steps=idx, cols=cols, trail=trail))
    return blocks


def single_blocks(sched, block):
    """With COMPACT: the joint steps after the first, `block` at a time, all rows in slot 0.  Same
    form as the blocks of the split phase, with nothing for slot 1."""
    st2 = sched["stage2"]
    rest = sched["joint"][1:] if COMPACT else []
    blocks = []
    for idx in (_chunks(list(rest), lambda c: _fits(st2, c)) if rest else []):
        cols = [st2[i]["col"] for i in idx]
        trail = sorted(set().union(*[set(st2[i]["pattern"]) for i in idx]) - set(cols))
        blocks.append([dict(steps=idx, cols=cols, trail=trail, base=0), dict(steps=[], cols=[], trail=[], base=len(idx))])
    return blocks


def compact_cols(sched):
    """Columns a row still holds when the rows are gathered in slot 0."""
    st2 = sched["stage2"]
    rest = sched["joint"][1:]
    return sorted(set().union(*[set(st2[i]["pattern"]) | {st2[i]["col"]} for i in rest]))


def row_items(nb, trail):
    """Doubles of a published pivot row, in order: the multipliers it received from the earlier
    pivots of its block, padding to an even count, its raw trailing entries, padding."""
    items = [("m", k) for k in range(nb - 1)]
    if len(items) % 2:
        items.append(("z", 0))
    first = len(items) // 2                       # 16-byte piece that holds the first trailing pair
    items += [("r", c) for c in trail]
    if len(items) % 2:
        items.append(("z", 0))
    assert len(items) * 8 <= ROW_BYTES, (len(items), ROW_BYTES)
    return items, first


# ---------------------------------------------------------------------------------------------
# NumPy interpreters of the schedule (rows of R = row positions), for validation
# ---------------------------------------------------------------------------------------------
def _initial(sched, k):
    R = np.zeros((32, 88))
    for r in sched["lanes_rows"]:
        for c, g in sched["orig"][r].items():
            R[sched["lane_of"][r], c] = k[g]
    active = np.zeros(32, bool)
    for r in sched["survivors"]:
        active[sched["lane_of"][r]] = True
    return R, active


def _tail(R, active):
    """Gauss-Jordan on the needed columns of the five rows left -> rows 5..9 of the action matrix."""
    nl = len(active)
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


def simulate(sched, k):
    """k[42] -> (status, A5x10) where A5x10[i-5][j] = A(i,j), i = 5..9 of the action matrix.
    Plain elimination: every step updates every row, one pivot at a time, candidates everywhere."""
    R, active = _initial(sched, k)
    for st in sched["steps"]:
        pc, pv = st["col"], st["pv"]
        m = R[:, pc] / pv
        vals = {c2: k[sched["orig"][st["row"]][c2]] for c2 in st["pattern"]}
        for c2 in st["pattern"]:
            R[:, c2] -= m * vals[c2]
        R[:, pc] = 0.0
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
    return _tail(R, active)


def _block_update(R, active, rows, cols, trail):
    """One block of delayed updates over the row positions `rows` (a boolean mask): panel steps on
    the block's own columns, the pivot rows' raw trailing entries as published, effective
    multipliers m' = m L^-1 (L = unit lower triangle of the multipliers the pivot rows received
    inside the block), one update of the trailing columns.  Returns False when a column has no
    usable candidate."""
    nl = len(active)
    nb = len(cols)
    M = np.zeros((nl, nb))                      # multipliers (update is  row += m * pivot row)
    piv = []
    for kk, c in enumerate(cols):
        cand = np.where(active & rows, np.abs(R[:, c]), -1.0)
        L = int(np.argmax(cand))
        if cand[L] <= 0.0:
            return False
        M[:, kk] = np.where(active & rows, -R[:, c] / R[L, c], 0.0)
        M[L, kk] = 0.0
        for c2 in cols[kk + 1:]:                # panel: the block's later columns stay current
            R[:, c2] += M[:, kk] * R[L, c2]
        active[L] = False
        piv.append(L)
    P = R[piv][:, trail].copy()                 # raw trailing rows of the pivots, as published
    Lm = np.array([[M[piv[j], kk] if kk < j else 0.0 for kk in range(nb)] for j in range(nb)])
    Me = M.copy()                               # effective multipliers: back substitution of m' L = m
    for kk in range(nb - 2, -1, -1):
        for j in range(kk + 1, nb):
            Me[:, kk] += Me[:, j] * Lm[j, kk]
    upd = Me @ P
    mask = rows.copy()
    mask[piv] = False                           # retired rows are never read again
    R[np.ix_(mask, trail)] += upd[mask]
    for c in cols:
        R[rows, c] = 0.0
    return True


def simulate_kernel(sched, k, block=BLOCK):
    """NumPy model of exactly the updates the generated code performs: a slot skips the stage-1
    steps none of its rows takes part in; in the split phase every slot has its own pivots,
    columns and trailing columns; the joint phase runs in blocks over both slots."""
    R, active = _initial(sched, k)
    slot = [np.arange(32) < 16, np.arange(32) >= 16]
    for st in sched["steps"]:
        pc, pv = st["col"], st["pv"]
        vals = {c2: k[sched["orig"][st["row"]][c2]] for c2 in st["pattern"]}
        for s in (0, 1):
            if not st["slots"][s]:
                continue
            m = np.where(slot[s], R[:, pc] / pv, 0.0)
            for c2 in st["pattern"]:
                R[:, c2] -= m * vals[c2]
            R[slot[s], pc] = 0.0
    for parts in split_blocks(sched, block):
        for s in (0, 1):
            if parts[s]["steps"] and not _block_update(R, active, slot[s], parts[s]["cols"], parts[s]["trail"]):
                return -1, None
    for blk in joint_blocks(sched, block):
        if not _block_update(R, active, slot[0] | slot[1], blk["cols"], blk["trail"]):
            return -1, None
    if COMPACT:
        # the d-th retired position of slot 0 takes the d-th active row of slot 1
        dead0 = [p for p in range(16) if not active[p]]
        act1 = [p for p in range(16, 32) if active[p]]
        for d, p in enumerate(dead0):
            if d < len(act1):
                R[p] = R[act1[d]]
                active[p] = True
        active[16:] = False
        for parts in single_blocks(sched, block):
            if not _block_update(R, active, slot[0], parts[0]["cols"], parts[0]["trail"]):
                return -1, None
    return _tail(R, active)


def fma_counts(sched, block=BLOCK):
    """Wave-level multiply-adds of the generated code per stage (both slots added up)."""
    s1 = sum((len(st["pattern"]) + 1) * sum(st["slots"]) for st in sched["steps"])
    sp = 0
    for parts in split_blocks(sched, block):
        for p in parts:
            nb = len(p["cols"])
            sp += nb * len(p["trail"]) + nb * (nb - 1)      # trailing + panel and back substitution
    jt = 0
    for blk in joint_blocks(sched, block):
        nb = len(blk["cols"])
        jt += 2 * (nb * len(blk["trail"]) + nb * (nb - 1))
    for parts in single_blocks(sched, block):
        nb = len(parts[0]["cols"])
        jt += nb * len(parts[0]["trail"]) + nb * (nb - 1)
    return dict(stage1=s1, split=sp, joint=jt)


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
    next 16 registers it initialises.  A (column, slot) register exists only if the code of that
    slot ever touches the column."""
    o = []
    w = o.append
    order = [[], []]
    st2 = sched["stage2"]
    w("// Wave-cooperative elimination of the 78x88 P4Pf template, FOUR systems per wave (one per row")
    w("// of 16 lanes, two matrix rows per lane); see the generator for the schedule.")
    w("// cf points at the 42 coefficients (+ 0.0 at index 42) of this lane's system in LDS, tail5x15")
    w("// at its output column: entry (r, j) of the 5 x 15 tail system goes to tail5x15[(r * 15 + j)")
    w("// * stride], written when wr is set.  Returns 0, or -1 when a pivot column had no usable entry.")
    w("// tab at the LDS copy of c_elim_tab, pbuf at ELIM_PBUF_VEC double2 of LDS scratch private to the")
    w("// lane's row of 16 lanes.")
    w("// PRECONDITION: all 64 lanes of the wave are active at the call.  The masked LDS stores set EXEC")
    w("// inside inline assembly and restore it to all ones, not to its value at entry, so a call from")
    w("// divergent code would re-enable lanes; the callers branch on block-uniform values only.")
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

    # ---- stage 1: a chain of short dependent steps per slot
    for s in (0, 1):
        kdone = set()
        w(f"    // ---- stage 1, rows {16 * s}..{16 * s + 15}")
        for si, st in enumerate(sched["steps"]):
            if not st["slots"][s]:
                continue
            need(st["col"], s)
            for c2 in st["pattern"]:
                need(c2, s)
                g = sched["orig"][st["row"]][c2]
                if g not in CONST and g not in kdone:
                    kdone.add(g)
                    w(f"    const double k{g}_{s} = cf[{g}];")
            pc, pv = st["col"], st["pv"]
            w(f"    {{ // step {si}: column {pc}")
            w(f"        const double m = r{pc}_{s} * {repr(-1.0 / pv)};")
            for c2 in st["pattern"]:
                g = sched["orig"][st["row"]][c2]
                ke = repr(CONST[g]) if g in CONST else f"k{g}_{s}"
                w(f"        r{c2}_{s} = fma(m, {ke}, r{c2}_{s});")
            w("    }")
        if s == 0:
            w('    asm volatile("" ::: "memory");')
    lane_of = sched["lane_of"]
    surv = sorted(lane_of[r] for r in sched["survivors"])
    m0 = sum(1 << L for L in surv if L < 16)
    m1 = sum(1 << (L - 16) for L in surv if L >= 16)
    w(f"    bool act0 = ((0x{m0:x}u >> rl) & 1u) != 0, act1 = ((0x{m1:x}u >> rl) & 1u) != 0;")

    def masked_stores_direct(tag, items, s, pkvar):
        """The lanes with pkvar >= 0 store their slot-s row into LDS row pkvar of the block.  The
        doubles go out as they sit in their registers, two per ds_write2_b64 (two separate data
        operands, so nothing has to be packed into consecutive registers first); padding entries are
        never read and are not written.  exec-masked stores in inline assembly: a divergent `if`
        would split the kernel's single basic block and wreck register allocation."""
        real = [(t, (f"m{x}_{s}" if kind == "m" else f"r{x}_{s}")) for t, (kind, x) in enumerate(items) if kind != "z"]
        w(f"        const unsigned long long wm{tag} = __builtin_amdgcn_ballot_w64({pkvar} >= 0);")
        w(f"        const unsigned addr{tag} = pbuf_addr + (unsigned)({pkvar} < 0 ? 0 : {pkvar}) * {ROW_BYTES}u;")
        for g0 in range(0, len(real), 12):
            grp = real[g0:g0 + 12]
            lines = ['"s_mov_b64 exec, %[mask]\\n\\t"']
            for j in range(0, len(grp), 2):
                if j + 1 < len(grp):
                    lines.append(f'"ds_write2_b64 %[addr], %[a{j}], %[a{j + 1}] offset0:{grp[j][0]} offset1:{grp[j + 1][0]}\\n\\t"')
                else:
                    lines.append(f'"ds_write_b64 %[addr], %[a{j}] offset:{8 * grp[j][0]}\\n\\t"')
            lines.append('"s_mov_b64 exec, -1"')
            w("        asm volatile(" + "\n                     ".join(lines))
            ops = ", ".join(f'[a{j}] "v"({e})' for j, (_, e) in enumerate(grp))
            w(f'                     :: [mask] "s"(wm{tag}), [addr] "v"(addr{tag}), {ops} : "memory");')

    def emit_split_block(bi, parts):
        """Both slots take their next pivots side by side; a lane's row becomes a pivot at most once
        per slot, so every slot needs one round of stores and no selection between the slots."""
        for s in (0, 1):
            p = parts[s]
            for c in p["cols"] + p["trail"]:
                need(c, s)
        w(f"    {{ // stage 2 split block {bi}: slot 0 columns {parts[0]['cols']} ({len(parts[0]['trail'])} trailing), "
          f"slot 1 columns {parts[1]['cols']} ({len(parts[1]['trail'])} trailing)")
        for s in (0, 1):
            p = parts[s]
            cols, nb = p["cols"], len(p["cols"])
            if not nb:
                continue
            w(f"        int pk{s} = -1;                     // LDS row this lane's slot-{s} row is published to")
            for k, c in enumerate(cols):
                w(f"        // slot {s}, step {p['steps'][k]}: column {c}")
                w(f"        bool none{s}_{k};")
                w(f"        const int L{s}_{k} = row16_argmax_abs1(r{c}_{s}, act{s}, none{s}_{k});")
                w(f"        bad |= none{s}_{k};")
                w(f"        const double nr{s}_{k} = neg_rcp(lane_bcast(r{c}_{s}, L{s}_{k}));")
                # retired rows are never read again: no masking of their multipliers
                w(f"        double m{k}_{s} = r{c}_{s} * nr{s}_{k};")
                for c2 in cols[k + 1:]:
                    w(f"        r{c2}_{s} = fma(m{k}_{s}, lane_bcast(r{c2}_{s}, L{s}_{k}), r{c2}_{s});")
                w(f"        {{ const bool me = lane == L{s}_{k}; pk{s} = me ? {p['base'] + k} : pk{s}; act{s} = act{s} && !me; }}")
        firsts = [0, 0]
        for s in (0, 1):
            p = parts[s]
            nb = len(p["cols"])
            if not nb:
                continue
            items, firsts[s] = row_items(nb, p["trail"])
            masked_stores_direct("ps" + str(s), items, s, f"pk{s}")
        w("        __builtin_amdgcn_wave_barrier();")
        for s in (0, 1):
            p = parts[s]
            apply_block(s, p["base"], len(p["cols"]), p["trail"], firsts[s], (s,))
        w("        __builtin_amdgcn_wave_barrier();")
        w("    }")

    def apply_block(tag, base, nb, trail, first, slots):
        """Effective multipliers m' = m L^-1 (back substitution, L from the heads of the published
        rows) and the update of the trailing columns of `slots` with the block's nb pivot rows,
        which sit in the LDS rows base .. base + nb - 1.  One or two LDS reads per pivot row: lane j
        of a row of 16 lanes reads the 16-byte pieces j and 16 + j of the published row, and
        every multiply-add takes its entry from the lane that holds it (v_fmac_f64 with the DPP
        control row_newbcast: the first factor comes from a fixed lane of the own row of 16; a
        broadcast read per pair of entries kept the LDS pipe busy 70 % of the time).  Per element
        the pivots are applied in ascending order."""
        if not nb:
            return
        npieces = first + (len(trail) + 1) // 2
        assert ((base + nb - 1) * ROW_BYTES) // 16 + (32 if npieces > 16 else 16) <= PBUF_VEC, "a reader of the block's last row would leave the scratch"
        for k in range(nb):
            w(f"        const double2 q{tag}_{k}_0 = pbuf[{((base + k) * ROW_BYTES) // 16} + rl];")
            if npieces > 16:
                w(f"        const double2 q{tag}_{k}_1 = pbuf[{((base + k) * ROW_BYTES) // 16 + 16} + rl];")

        def fmac(acc, k, t, mult):
            piece = t // 2
            w(f'        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:{piece % 16} row_mask:0xf bank_mask:0xf" : "+v"({acc}) : '
              f'"v"(q{tag}_{k}_{piece // 16}.{"xy"[t % 2]}), "v"({mult}));')
        for k in range(nb - 2, -1, -1):
            for j in range(k + 1, nb):
                for s in slots:
                    fmac(f"m{k}_{s}", j, k, f"m{j}_{s}")
        for k in range(nb):
            for i, c2 in enumerate(trail):
                for s in slots:
                    fmac(f"r{c2}_{s}", k, 2 * first + i, f"m{k}_{s}")

    def emit_joint_block(bi, blk):
        """A block of delayed updates with candidates in both slots: panel steps on the block's own
        columns, then every pivot row's raw trailing entries (and the multipliers it received inside
        the block) go to LDS, and all rows apply the whole block at once.  A row becomes a pivot at
        most once, so the pivots that sit in slot 0 go out in one round of stores straight from the
        slot-0 registers and those in slot 1 in a second round: no selection between the slots."""
        cols, trail, nb = blk["cols"], blk["trail"], len(blk["cols"])
        for c in cols + trail:
            need(c, 0)
            need(c, 1)
        w(f"    {{ // stage 2 joint block {bi}: columns {cols}, {len(trail)} trailing columns")
        w("        int pk0 = -1, pk1 = -1;           // LDS row (= step inside the block) this lane's slot-0 / slot-1 row is published to")
        for k, c in enumerate(cols):
            w(f"        // step {blk['steps'][k]}: column {c}")
            w(f"        bool sl{k}, none{k};")
            w(f"        const int L{k} = row16_argmax_abs(r{c}_0, act0, r{c}_1, act1, sl{k}, none{k});")
            w(f"        bad |= none{k};")
            w(f"        const double nr{k} = neg_rcp(lane_bcast(sl{k} ? r{c}_1 : r{c}_0, L{k}));")
            # retired rows are never read again: no masking of their multipliers
            w(f"        double m{k}_0 = r{c}_0 * nr{k}, m{k}_1 = r{c}_1 * nr{k};")
            for c2 in cols[k + 1:]:
                w(f"        {{ const double e = lane_bcast(sl{k} ? r{c2}_1 : r{c2}_0, L{k});")
                w(f"          r{c2}_0 = fma(m{k}_0, e, r{c2}_0); r{c2}_1 = fma(m{k}_1, e, r{c2}_1); }}")
            w(f"        {{ const bool me0 = lane == L{k} && !sl{k}, me1 = lane == L{k} && sl{k};")
            w(f"          pk0 = me0 ? {k} : pk0; pk1 = me1 ? {k} : pk1; act0 = act0 && !me0; act1 = act1 && !me1; }}")
        items, first = row_items(nb, trail)
        if nb == 1:
            # one pivot row per system: the slot it sits in is chosen with selects (two per double) and
            # the row goes out in ONE round of stores -- a round costs more than the selects
            for kind, x in items:
                if kind == "r":
                    w(f"        const double rsel{x}_0 = sl0 ? r{x}_1 : r{x}_0;")
            w("        const int pks = lane == L0 ? 0 : -1;")
            masked_stores_direct("pjs", [(kind, f"sel{x}") if kind == "r" else (kind, x) for kind, x in items], 0, "pks")
        else:
            masked_stores_direct("pj0", items, 0, "pk0")
            masked_stores_direct("pj1", items, 1, "pk1")
        w("        __builtin_amdgcn_wave_barrier();")
        apply_block("j", 0, nb, trail, first, (0, 1))
        w("        __builtin_amdgcn_wave_barrier();")
        w("    }")

    for bi, parts in enumerate(split_blocks(sched, BLOCK)):
        emit_split_block(bi, parts)
    for bi, blk in enumerate(joint_blocks(sched, BLOCK)):
        emit_joint_block(bi, blk)
    if COMPACT:
        # 29 rows, 13 pivots so far: 16 rows are active, and slot 0 has exactly as many retired
        # positions as slot 1 has active rows.  The d-th retired position of slot 0 takes the d-th
        # active row of slot 1; the remaining steps then run in slot 0 alone: half the multiply-adds,
        # one round of stores per block, no selection between the slots.
        cc = compact_cols(sched)
        for c in cc:
            need(c, 0)
            need(c, 1)
        nmove = 16 - sum(1 for r in sched["survivors"] if sched["slot_of"][r] == 0) + len(sched["split"][0]) + 1
        gstride = len(cc) * 8 + 8           # bytes between the moved rows in LDS: not a multiple of 256, so that the rows sit on different banks
        assert nmove * gstride <= PBUF_VEC * 16
        w("    { // gather the active rows in slot 0: through LDS, stores and loads under EXEC masks, no selects")
        w("        const unsigned a1m = (unsigned)(__ballot(act1) >> sh) & 0xffffu;      // active rows of slot 1, this system")
        w("        const unsigned d0m = ~(unsigned)(__ballot(act0) >> sh) & 0xffffu;     // retired positions of slot 0")
        w("        const unsigned below = (1u << rl) - 1u;")
        w("        const int j1 = __popc(a1m & below), d = __popc(d0m & below);           // rank of this lane's slot-1 row / slot-0 position")
        w(f"        const bool give = act1 && j1 < {nmove}, take = !act0 && d < __popc(a1m) && d < {nmove};")
        w("        const unsigned long long gm = __builtin_amdgcn_ballot_w64(give), tm = __builtin_amdgcn_ballot_w64(take);")
        w(f"        const unsigned gaddr = pbuf_addr + (unsigned)(give ? j1 : 0) * {gstride}u, taddr = pbuf_addr + (unsigned)(take ? d : 0) * {gstride}u;")
        for g0 in range(0, len(cc), 12):
            grp = cc[g0:g0 + 12]
            lines = ['"s_mov_b64 exec, %[mask]\\n\\t"']
            for j in range(0, len(grp), 2):
                if j + 1 < len(grp):
                    lines.append(f'"ds_write2_b64 %[addr], %[a{j}], %[a{j + 1}] offset0:{g0 + j} offset1:{g0 + j + 1}\\n\\t"')
                else:
                    lines.append(f'"ds_write_b64 %[addr], %[a{j}] offset:{8 * (g0 + j)}\\n\\t"')
            lines.append('"s_mov_b64 exec, -1"')
            w("        asm volatile(" + "\n                     ".join(lines))
            ops = ", ".join(f'[a{j}] "v"(r{c}_1)' for j, c in enumerate(grp))
            w(f'                     :: [mask] "s"(gm), [addr] "v"(gaddr), {ops} : "memory");')
        # the loads land in the registers of the positions that take a row; every other lane keeps
        # what it has.  LDS operations of a wave execute in order, so the loads see the stores; the
        # wait for the loaded values is part of the statement (the compiler does not know of them).
        for g0 in range(0, len(cc), 14):
            grp = cc[g0:g0 + 14]
            lines = ['"s_mov_b64 exec, %[mask]\\n\\t"']
            for j, c in enumerate(grp):
                lines.append(f'"ds_read_b64 %[a{j}], %[addr] offset:{8 * (g0 + j)}\\n\\t"')
            lines.append('"s_waitcnt lgkmcnt(0)\\n\\t"')
            lines.append('"s_mov_b64 exec, -1"')
            w("        asm volatile(" + "\n                     ".join(lines))
            ops = ", ".join(f'[a{j}] "+v"(r{c}_0)' for j, c in enumerate(grp))
            w(f'                     : {ops} : [mask] "s"(tm), [addr] "v"(taddr) : "memory");')
        w("        act0 = act0 || take; act1 = false;")
        w("        __builtin_amdgcn_wave_barrier();")
        w("    }")
        for bi, parts in enumerate(single_blocks(sched, BLOCK)):
            emit_split_block(100 + bi, parts)
    for c in KEEP + RHS:
        need(c, 0)
        if not COMPACT:
            need(c, 1)
    w("    {")
    w("        const unsigned a0 = (unsigned)(__ballot(act0) >> sh) & 0xffffu, a1 = (unsigned)(__ballot(act1) >> sh) & 0xffffu;")
    w("        const int n0 = __popc(a0);")
    w("        bad |= n0 + __popc(a1) != 5;")
    w("        const unsigned below = (1u << rl) - 1u;")
    w("        [[maybe_unused]] const int rank0 = __popc(a0 & below), rank1 = n0 + __popc(a1 & below);")
    for s in ((0,) if COMPACT else (0, 1)):
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
    # a register that was never created must not hide an entry of the initial matrix
    for r in sched["lanes_rows"]:
        assert set(sched["orig"][r]) <= set(order[sched["slot_of"][r]]), "an initial entry has no register"
    # 16-byte vectors per position.  An odd number: the 16 positions of a slot then start on 16 different
    # groups of four banks, which is what the lane groups of a ds_read_b128 need.  The look-ahead fetch
    # of a position's last vector reads the first one of the next position (one vector of padding
    # at the end of the table).
    nvec = (max(len(order[0]), len(order[1])) + 15) // 16
    nvec += 1 - nvec % 2
    head = ["// GENERATED by tools/gen_p4pf_elim.py -- do not edit.", "#pragma once", "namespace pose_amd {",
            f"constexpr int ELIM_TAB_VEC = {nvec};   // 16-byte vectors of coefficient ids per (slot, position)",
            f"constexpr int ELIM_PBUF_VEC = {PBUF_VEC};   // 16-byte vectors of LDS scratch per system: the published pivot rows of a block and the overrun of the readers' second piece; a multiple of 256 bytes, so that the lane groups a ds_read_b128 is served in (lanes 0-3, 12-15 and 20-27 together, ...) see 16 different pieces of 16 bytes when every lane reads piece lane % 16 of its own system's row",
            "// coefficient ids [slot * 16 + position][order of first use] of the initial matrix; 42 = structural zero",
            f"__constant__ __attribute__((aligned(16))) unsigned char c_elim_tab[32 * {nvec * 16} + 16] = {{"]
    for s in (0, 1):
        for pos in range(16):
            ids = [int(tbl[16 * s + pos, c]) for c in order[s]]
            ids += [42] * (nvec * 16 - len(ids))
            head.append("  " + ",".join(str(v) for v in ids) + ",")
    head.append("  " + ",".join(["42"] * 16))
    head.append("};")
    return "\n".join(head + o) + "\n"


if __name__ == "__main__":
    s = build_schedule()
    print("stage 1: %d constant-pivot steps, %d rows in lanes; steps run by slot 0 / slot 1: %d / %d" % (
        len(s["steps"]), len(s["lanes_rows"]), sum(st["slots"][0] for st in s["steps"]), sum(st["slots"][1] for st in s["steps"])))
    print("stage 2: split phase, columns of slot 0 / slot 1:", [[s["stage2"][i]["col"] for i in s["split"][t]] for t in (0, 1)],
          "joint phase:", [s["stage2"][i]["col"] for i in s["joint"]])
    print("wave-level multiply-adds:", fma_counts(s))
    out = os.path.join(ROOT, "pose_estimation_amd", "csrc", "p4pf_elim_gen.h")
    open(out, "w").write(emit(s))
    print("wrote", out)
