#!/usr/bin/env python3
"""Static check of the hand-written DPP instructions in the gfx950 assembly of csrc/imsim_hip.hip.

The compiler's hazard recogniser does not look inside inline asm.  gfx9 needs two wait states between a VALU instruction
that writes a VGPR and a DPP instruction that reads that VGPR as its DPP source (src0), and five between a VALU write of
EXEC and a DPP instruction.  This scans the .s file of a -save-temps build: for every *_dpp instruction it walks back over
the preceding instructions along EVERY predecessor path -- the fall-through above a label and the branches that name it --
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
    # control flow: a label is reached by falling through from the line above (unless that is an unconditional branch) AND
    # from every branch that names it -- all of these are predecessors of the instruction behind the label
    label_at = {t[:-1]: k for k, (n, t) in enumerate(ins) if t.endswith(":")}
    jumps_to = {}
    for k, (n, t) in enumerate(ins):
        op = t.split()[0]
        if op.startswith("s_branch") or op.startswith("s_cbranch"):
            target = t.split()[-1]
            if target in label_at:
                jumps_to.setdefault(label_at[target], []).append(k)

    def hazards(k_dpp, src0, k, wait, seen):
        """walk back from instruction k (exclusive) along every predecessor path until five wait states are covered"""
        out = []
        j = k - 1
        while j >= 0 and wait < 5:
            pn, pt = ins[j]
            if pt.endswith(":"):
                for src in jumps_to.get(j, []):                     # the branches into this label: the branch itself issues (one state)
                    if (src, wait) not in seen:
                        seen.add((src, wait))
                        out += hazards(k_dpp, src0, src, wait + 1, seen)
                j -= 1                                              # and the fall-through path goes on above the label
                continue
            op = pt.split()[0]
            if op.startswith("s_branch") and j != k - 1 and ins[j + 1][1].endswith(":"):
                break                                               # an unconditional branch: nothing falls through into the label below it
            if op.startswith("s_nop"):
                wait += int(pt.split()[1], 0) + 1
                j -= 1
                continue
            if op.startswith("v_"):
                pops = [o.strip() for o in pt.split(None, 1)[1].split(",")] if " " in pt else []
                dst = regs(pops[0]) if pops else set()
                if dst & src0 and wait < 2:
                    out.append((ins[k_dpp][0], ins[k_dpp][1], pn, pt, "VALU write -> DPP source needs 2 wait states"))
                if "_dpp" not in op and ("exec" in (pops[0] if pops else "") or op.startswith("v_cmpx")) and wait < 5:
                    out.append((ins[k_dpp][0], ins[k_dpp][1], pn, pt, "VALU write of EXEC -> DPP needs 5 wait states"))
            wait += 1
            j -= 1
        return out

    for k, (n, t) in enumerate(ins):
        if "_dpp" not in t.split()[0]:
            continue
        n_dpp += 1
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        src0 = regs(ops[1].split()[0])
        bad += hazards(k, src0, k, 0, set())
    return n_dpp, bad


if __name__ == "__main__":
    n_dpp, bad = check(sys.argv[1])
    print(f"{n_dpp} DPP instructions checked, {len(bad)} hazards")
    for b in bad[:20]:
        print(b)
    sys.exit(1 if bad else 0)
