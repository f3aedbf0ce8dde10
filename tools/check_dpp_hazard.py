#!/usr/bin/env python3
"""Static check of the hand-written DPP instructions in the gfx950 assembly of csrc/imsim_hip.hip.

The compiler's hazard recogniser does not look inside inline asm.  gfx9 needs two wait states between a VALU instruction
that writes a VGPR and a DPP instruction that reads that VGPR as its DPP source (src0), and five between a VALU write of
EXEC and a DPP instruction.  This scans the .s file of a -save-temps build: for every *_dpp instruction it walks back over
the preceding instructions (labels and branch targets count as unknown = fine only if the distance is already covered)
and reports a violation.  Usage: python tools/check_dpp_hazard.py file.s   (exit code 1 on a hazard)"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    lines = open(path).read().splitlines()
    ins = []                                    # (line number, text) of instructions and labels
    for n, l in enumerate(lines, 1):
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") and not t.endswith(":") or t.startswith(";;"):
            continue
        ins.append((n, t))
    bad = []
    n_dpp = 0
    for k, (n, t) in enumerate(ins):
        if "_dpp" not in t.split()[0]:
            continue
        n_dpp += 1
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        src0 = regs(ops[1].split()[0])
        wait = 0
        for back in range(1, 6):
            if k - back < 0:
                break
            pn, pt = ins[k - back]
            if pt.endswith(":"):
                break                            # a label: predecessors unknown, conservatively stop (branch costs wait states)
            op = pt.split()[0]
            if op.startswith("s_nop"):
                wait += int(pt.split()[1], 0) + 1
                continue
            if op.startswith("v_") and "_dpp" not in op:
                pops = [o.strip() for o in pt.split(None, 1)[1].split(",")] if " " in pt else []
                dst = regs(pops[0]) if pops else set()
                if dst & src0 and wait < 2:
                    bad.append((n, t, pn, pt, "VALU write -> DPP source needs 2 wait states"))
                if ("exec" in (pops[0] if pops else "") or op.startswith("v_cmpx")) and wait < 5:
                    bad.append((n, t, pn, pt, "VALU write of EXEC -> DPP needs 5 wait states"))
            if op.startswith("v_") and "_dpp" in op:
                pops = [o.strip() for o in pt.split(None, 1)[1].split(",")]
                if regs(pops[0]) & src0 and wait < 2:
                    bad.append((n, t, pn, pt, "VALU write -> DPP source needs 2 wait states"))
            wait += 1
    return n_dpp, bad


if __name__ == "__main__":
    n_dpp, bad = check(sys.argv[1])
    print(f"{n_dpp} DPP instructions checked, {len(bad)} hazards")
    for b in bad[:20]:
        print(b)
    sys.exit(1 if bad else 0)
