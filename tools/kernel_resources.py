#!/usr/bin/env python3
"""VGPR / SGPR / spill / scratch of every kernel in an object or shared library whose name contains a pattern.

    python tools/kernel_resources.py audio_source_separation_amd/csrc/assx_nmf.o valu

Scratch (private segment) > 0 on a hot kernel is the first thing to look for: the reservation alone throttles the
resident waves (the IP kernel)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    path, pat = os.path.abspath(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "")
    with tempfile.TemporaryDirectory() as d:
        link = os.path.join(d, "in.o")
        os.symlink(path, link)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], cwd=d, capture_output=True, check=True)
        dev = [f for f in os.listdir(d) if "gfx950" in f]
        if not dev:
            sys.exit("no gfx950 code object found")
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(d, dev[0])], capture_output=True,
                               text=True, check=True).stdout
    rows = []
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        if pat in name:
            rows.append((name, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                         g("group_segment_fixed_size")))
    print(f"{'vgpr':>5} {'sgpr':>5} {'spill':>5} {'scratch':>7} {'lds':>6}  kernel")
    for name, v, s, sp, ps, lds in rows:
        short = re.sub(r"^void ", "", name)
        short = re.sub(r"\(.*$", "", short)
        print(f"{v:>5} {s:>5} {sp:>5} {ps:>7} {lds:>6}  {short}")


if __name__ == "__main__":
    main()
