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

LDS-DIRECT loads (buffer_load ... lds: the destination is LDS memory, not a register) are followed by COUNT (round 4): their
landing addresses are run-time values (M0 from a rotating ring slot), which a static walk cannot compare with the address
of a later ds_read, but every ring of these kernels keeps one invariant that can be counted: when an inline-asm LDS read
executes, the wave has at most  Nmax + B  LDS-direct loads in flight, Nmax = the largest vmcnt immediate the kernel waits
with, B = the LDS-direct loads one trip of the enclosing loop issues (0 outside loops).  A missing counted wait -- in the
prologue, or at the end of a trip: the walk goes round every loop three times, so what a trip leaves behind adds up -- breaks
it and is reported ("N LDS-direct loads in flight at an asm LDS read").  What this does NOT see: a wait whose immediate is
too large by less than a trip's loads, or a read of the wrong slot; the full-size co-residency tests are the guard there.

M0 (round 6): an LDS-direct load takes its LDS address from M0, which the compiler manages for its own LDS-direct loads and
for everything else that uses M0.  An inline-asm block that WRITES M0 (s_mov_b32 m0, ... inside ;;#ASMSTART .. ;;#ASMEND) sits
next to that bookkeeping without being part of it (listing M0 as a clobber is answered with "reserved registers on the
clobber list may not be preserved ... undefined behaviour"): every such write is reported.  The kernels hand the address in
through an input operand pinned to the register ("{m0}"(addr)), so the write is the compiler's own, outside the block.

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


LOOP_WALKS = 3  # times the walk goes round a loop (what a trip leaves in flight has to show up as an excess)


def check(lines, name, linear=False):
    pend = {"lgkm": [], "vm": []}
    bad = []
    seen = set()
    in_asm = False
    last_op = ""
    # ---- pre-pass: labels, the largest vmcnt immediate, the loops (backward branches) and their LDS-direct loads per trip
    code = [raw.split(";")[0].strip() for _, raw in lines]
    label_at = {c[:-1]: k for k, c in enumerate(code) if c.endswith(":")}
    nmax = 0
    for c in code:
        if c.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", c)
            if m:
                nmax = max(nmax, int(m.group(1)))
    is_ldsdirect = [(c.startswith("buffer_load") or c.startswith("global_load")) and " lds" in c for c in code]
    loops = []  # (head index, branch index, LDS-direct loads in between)
    for k, c in enumerate(code):
        if c.startswith("s_cbranch") or c.startswith("s_branch"):
            tgt = c.split()[-1]
            t = label_at.get(tgt)
            if t is not None and t < k:
                loops.append((t, k, sum(is_ldsdirect[t:k])))
    # the walk goes back at the BOTTOM-most branch to a head only: an earlier one (a conditional skip to the latch of a rotated
    # loop) falls through, so that the blocks between it and the bottom are walked as well
    back_edge = {}
    for t, k, _ in loops:
        if all(not (t2 == t and k2 > k) for t2, k2, _ in loops):
            back_edge[k] = t
    walks = {}
    incoming = {}  # label -> what an unconditional FORWARD branch to it had pending (the walk itself goes on below the branch)

    def per_trip(k):  # B of the loops around instruction k (the widest one: an outer trip contains the inner ones)
        return max([b for t, j, b in loops if t <= k <= j] or [0])

    k = 0
    while k < len(lines):
        ln, raw = lines[k]
        k += 1
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        l = raw.split(";")[0].strip()
        if l.endswith(":") and not linear and last_op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pend = {"lgkm": [], "vm": []}  # not reachable by falling through: what is pending here came from elsewhere
        if l.endswith(":") and l[:-1] in incoming:  # ... e.g. from a forward branch over this point: the longer lists count
            inc = incoming.pop(l[:-1])              # (once: the next time round the loop the branch saves its state again)
            for key in ("lgkm", "vm"):
                if len(inc[key]) > len(pend[key]):
                    pend[key] = inc[key]
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
        if op.startswith("s_cbranch") or op == "s_branch":
            t = back_edge.get(k - 1)
            if t is not None and walks.get(k - 1, 0) < LOOP_WALKS - 1:
                walks[k - 1] = walks.get(k - 1, 0) + 1
                k = t  # once more round the loop, with the LDS-direct loads this trip left in flight.  Register loads are not
                last_op = ""  # carried: the linear walk conflates the paths of a loop body, and what it believes pending at the
                pend = {"lgkm": [], "vm": [e for e in pend["vm"] if e[3]]}  # bottom would meet the top's register writes
            elif op == "s_branch" and label_at.get(l.split()[-1], -1) >= k:
                incoming[l.split()[-1]] = {key: list(v) for key, v in pend.items()}
            continue
        if in_asm and op.startswith("s_") and parts and re.match(r"^m0\b", parts[0]) and (ln, "m0") not in seen:
            seen.add((ln, "m0"))
            bad.append((ln, l, "inline asm writes M0 (pass the LDS address as an input operand pinned to the register: \"{m0}\"(addr))"))
        if op == "s_barrier" or op.startswith("s_"):
            continue
        touched = set()
        for p_ in parts:
            touched |= regs(p_)
        for key in ("lgkm", "vm"):
            for dst, src, from_asm, _ in pend[key]:
                if from_asm and dst & touched and (ln, src) not in seen:
                    seen.add((ln, src))
                    bad.append((ln, l, src))
        if in_asm and op.startswith("ds_read"):
            inflight = sum(1 for e in pend["vm"] if e[3])
            if inflight > nmax + per_trip(k - 1) and (ln, "lds-direct") not in seen:
                seen.add((ln, "lds-direct"))
                bad.append((ln, l, "%d LDS-direct loads in flight at an asm LDS read (largest counted wait %d + %d per trip)"
                            % (inflight, nmax, per_trip(k - 1))))
        if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle") or op.startswith("ds_permute"):
            pend["lgkm"].append((regs(parts[0]), l, in_asm, False))
        elif op.startswith("ds_"):
            pend["lgkm"].append((set(), l, in_asm, False))
        elif op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load"):
            direct = " lds" in l
            pend["vm"].append((set() if direct else regs(parts[0]), l, in_asm, direct))
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
                    print("%s: %d reports (a pending load's registers touched / LDS-direct loads in flight at an asm LDS read / M0 written by inline asm)" % (name[:100], len(bad)))
                    for ln, l, src in bad[:6]:
                        print("    +%d  %s    <- pending: %s" % (ln, l, src))
                total += len(bad)
            i = j
        i += 1
    print("total", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
