#!/usr/bin/env python3
"""Static check of the inline-asm loads against their waits, on the compiler's assembly output.

The streaming kernels issue LDS reads (ds_read*) and global loads (buffer_load*) from inline asm and wait for them with
explicit s_waitcnt lgkmcnt(N) / vmcnt(N); the compiler does not know that the asm outputs are written LATER, so it is free
to copy, move or reuse such a register between the asm and the wait (register coalescing decides, per instantiation).
This tool walks every kernel of a .s file and reports any instruction that reads or writes the destination registers of a
load that no wait has covered yet (returns in order: a wait for N leaves the N youngest pending).  Only loads issued from
inline asm (between ;;#ASMSTART and ;;#ASMEND) are checked: the compiler waits for its own (they stay in the list so that
the counting is right).  The walk is linear; by default the pending lists are dropped at labels that cannot be reached by falling through
(after s_branch / s_endpgm); --linear keeps them across every label (more places to LOOK at, some of them
on paths that never follow each other).  Stores and scalar loads are ignored (stores make the vmcnt
reading conservative; the kernels have no scalar loads inside their pipelined loops).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S csrc/assx_widem.hip -o /tmp/widem.s
    python tools/asm_wait_check.py /tmp/widem.s [kernel-name substring]"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out |= {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def check(lines, name, linear=False):
    pend = {"lgkm": [], "vm": []}
    bad = []
    in_asm = False
    last_op = ""
    for ln, raw in lines:
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        l = raw.split(";")[0].strip()
        if l.endswith(":") and not linear and last_op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pend = {"lgkm": [], "vm": []}  # not reachable by falling through: what is pending here came from elsewhere
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        last_op = op
        parts = [p.strip() for p in l[len(op):].split(",")]
        if op.startswith("s_waitcnt"):
            for key, pat in (("lgkm", r"lgkmcnt\((\d+)\)"), ("vm", r"vmcnt\((\d+)\)")):
                m = re.search(pat, l)
                if m:
                    n = int(m.group(1))
                    pend[key] = pend[key][len(pend[key]) - n:] if n > 0 else []
            continue
        if op == "s_barrier" or op.startswith("s_"):
            continue
        touched = set()
        for p_ in parts:
            touched |= regs(p_)
        for key in ("lgkm", "vm"):
            for dst, src, from_asm in pend[key]:
                if from_asm and dst & touched:
                    bad.append((ln, l, src))
        if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle") or op.startswith("ds_permute"):
            pend["lgkm"].append((regs(parts[0]), l, in_asm))
        elif op.startswith("ds_"):
            pend["lgkm"].append((set(), l, in_asm))
        elif op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load"):
            pend["vm"].append((set() if " lds" in l else regs(parts[0]), l, in_asm))
    return bad


def main():
    linear = "--linear" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--linear"]
    txt = open(args[0]).read().split("\n")
    want = args[1] if len(args) > 1 else ""
    total = 0
    i = 0
    while i < len(txt):
        m = re.match(r"^(_Z\w+):\s", txt[i] + " ")
        if m and ".amdhsa" not in txt[i]:
            name = m.group(1)
            j = i + 1
            while j < len(txt) and ".Lfunc_end" not in txt[j]:
                j += 1
            if want in name:
                bad = check([(k - i, txt[k]) for k in range(i, j)], name, linear)
                if bad:
                    print("%s: %d touches of a pending load's registers" % (name[:100], len(bad)))
                    for ln, l, src in bad[:6]:
                        print("    +%d  %s    <- pending: %s" % (ln, l, src))
                total += len(bad)
            i = j
        i += 1
    print("total", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
