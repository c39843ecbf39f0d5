#!/usr/bin/env python3
"""Instruction histogram (or full listing with -l) of the kernels in an object whose demangled name contains a pattern.

    python tools/kernel_isa.py audio_source_separation_amd/csrc/assx_nmf.o "nmf_act_valu_kernel<double, 12, 2>" [-l]
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    path, pat = os.path.abspath(sys.argv[1]), sys.argv[2]
    listing = "-l" in sys.argv[3:]
    with tempfile.TemporaryDirectory() as d:
        link = os.path.join(d, "in.o")
        os.symlink(path, link)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], cwd=d, capture_output=True, check=True)
        dev = [f for f in os.listdir(d) if "gfx950" in f][0]
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(d, dev)], capture_output=True, text=True,
                              check=True).stdout
    for part in re.split(r"\n(?=[0-9a-f]{16} <)", text):
        m = re.match(r"[0-9a-f]{16} <(\S+)>:", part)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pat not in name:
            continue
        lines = [ln for ln in part.split("\n")[1:] if ln.strip()]
        print(re.sub(r"\(.*$", "", name), len(lines), "instructions")
        if listing:
            for ln in lines:
                print("   ", ln.split("//")[0].rstrip())
        else:
            c = Counter(ln.split()[0] for ln in lines)
            for k, v in c.most_common(30):
                print(f"    {v:5d} {k}")


if __name__ == "__main__":
    main()
