#!/usr/bin/env python3
"""Checks the device assembly of the library for the one hazard the compiler does not see inside
inline assembly: a v_fmac_f64_dpp that reads, through DPP, a register written by a vector ALU
instruction less than two wait states earlier.  A label inside that window is a branch target whose
other predecessors are not followed: it is reported as a hazard too (conservatively).
Usage: check_dpp_hazards.py FILE.s"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


lines = [l.strip() for l in open(sys.argv[1]) if re.match(r"\s+[a-z]", l) or re.match(r"\.?[A-Za-z_][\w.$]*:", l)]
bad = n = 0
for i, l in enumerate(lines):
    if not l.startswith("v_fmac_f64_dpp"):
        continue
    n += 1
    src0 = regs(l.split()[2].rstrip(","))
    states = 0
    for p in reversed(lines[max(0, i - 3):i]):
        op = p.split()[0]
        if op.endswith(":"):
            bad += 1
            print("label inside the hazard window:", p, "->", l)
            break
        if op == "s_nop":
            states += int(p.split()[1]) + 1
        elif op.startswith("v_") and not op.startswith("v_cmp"):
            if regs(p.split()[1].rstrip(",")) & src0 and states < 2:
                bad += 1
                print("hazard:", p, "->", l)
            states += 1
        else:
            states += 1
        if states >= 2:
            break
print(f"{n} DPP multiply-adds checked, {bad} hazards")
sys.exit(1 if bad else 0)
