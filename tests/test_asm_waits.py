"""Inline-asm loads against their waits, checked on the compiler's assembly (tools/asm_wait_check.py).

The wide-channel covariance kernels read LDS through inline asm and wait with explicit s_waitcnt; the compiler does not
know that such an asm's outputs are written later and may copy them before the wait.  That happened (round 3:
lds_read_row4 unpacked its registers right after the asm; in four float64 instantiations the unpacking became v_mov_b64
instructions ahead of the wait, harmless until two workgroups shared a CU and the LDS answered later -- garbage weights in
a third of the bins, differently on every run).  No GPU needed: hipcc cross-compiles.  Compiled here: the streaming
covariance kernels of csrc/assx_widem_cov.hpp in the instantiations of tests/asm_wait_probe.hip (under a minute) and the
M <= 4 streaming kernels in the instantiations of the BASELINE configs (tests/asm_wait_probe_bss.hip, seconds); the whole of
csrc/assx_widem.hip and csrc/assx_bss.hip take 4-5 minutes each and are checked by hand with the same tool (0 reports at
the end of round 3)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("probe", ["asm_wait_probe.hip", "asm_wait_probe_bss.hip"])
def test_kernels_do_not_touch_pending_inline_asm_loads(tmp_path, probe):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = tmp_path / "widem.s"
    src = os.path.join(ROOT, "tests", probe)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-w", "--cuda-device-only",
                    "-S", src, "-o", str(asm)], check=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_wait_check.py"), str(asm)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("total 0"), r.stdout[-2000:]
    assert open(asm).read().count("#ASMSTART") > 100  # the kernels were really emitted
