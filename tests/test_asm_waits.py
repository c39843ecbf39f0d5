"""Inline-asm loads against their waits, checked on the compiler's assembly (tools/asm_wait_check.py).

The wide-channel covariance kernels read LDS through inline asm and wait with explicit s_waitcnt; the compiler does not
know that such an asm's outputs are written later and may copy them before the wait.  That happened (round 3:
lds_read_row4 unpacked its registers right after the asm; in four float64 instantiations the unpacking became v_mov_b64
instructions ahead of the wait, harmless until two workgroups shared a CU and the LDS answered later -- garbage weights in
a third of the bins, differently on every run).  No GPU needed: hipcc cross-compiles.  Compiled here: the streaming
covariance kernels of csrc/assx_widem_cov.hpp in the instantiations of tests/asm_wait_probe.hip (under a minute) and the
M <= 4 streaming kernels in the instantiations of the BASELINE configs (tests/asm_wait_probe_bss.hip, seconds); the whole of
csrc/assx_widem.hip and csrc/assx_bss.hip take 4-5 minutes each and are checked by csrc/build.sh itself (ASSX_CHECK, round 4).
Round 4 also: LDS-direct loads counted at inline-asm LDS reads (tests/asm_wait_ring.hip)."""
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


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    return hipcc


def test_checker_reports_a_register_consumed_before_its_wait(tmp_path):
    """tests/asm_wait_bad.hip has the round-3 bug on purpose: the checker must name it."""
    asm = tmp_path / "bad.s"
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "--cuda-device-only", "-S",
                    os.path.join(ROOT, "tests", "asm_wait_bad.hip"), "-o", str(asm)], check=True, timeout=300)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_wait_check.py"), str(asm)], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1 and "asm_wait_bad_kernel" in r.stdout and "ds_read_b128" in r.stdout, r.stdout


def test_build_fails_on_a_kernel_that_touches_a_pending_asm_load(tmp_path):
    """csrc/build.sh walks the device assembly of every translation unit it compiles (ASSX_CHECK, on by default) and a
    report fails the build: fed the bad kernel (ASSX_SRCS) it must exit non-zero and say why; fed a clean unit, zero."""
    _hipcc()
    build = os.path.join(ROOT, "audio_source_separation_amd", "csrc", "build.sh")
    env = dict(os.environ, ASSX_OBJ=str(tmp_path), ASSX_OUT=str(tmp_path / "lib.so"), ASSX_SRCS="../../tests/asm_wait_bad")
    r = subprocess.run(["bash", build], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "asm_wait_check FAILED" in r.stderr, (r.returncode, r.stderr[-1500:])
    env["ASSX_SRCS"] = "assx_api"
    r = subprocess.run(["bash", build], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "asm_wait_check assx_api: total 0" in r.stdout, (r.returncode, r.stderr[-1500:])


def test_checker_counts_lds_direct_loads_in_flight_at_asm_lds_reads(tmp_path):
    """LDS-direct loads (buffer_load ... lds) land in LDS, not in a register the walk could follow, at run-time addresses.
    What can be counted: at an inline-asm LDS read a ring has at most (largest counted wait + one trip's loads) of them in
    flight.  tests/asm_wait_ring.hip holds a small ring twice: with its per-trip wait (clean) and without it (reported,
    on the third time round the loop: the walk carries what a trip leaves pending)."""
    asm = tmp_path / "ring.s"
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "--cuda-device-only", "-S",
                    os.path.join(ROOT, "tests", "asm_wait_ring.hip"), "-o", str(asm)], check=True, timeout=300)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_wait_check.py"), str(asm)], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1 and "ring_bad_kernel" in r.stdout and "LDS-direct loads in flight" in r.stdout, r.stdout
    assert "ring_good_kernel" not in r.stdout and r.stdout.strip().endswith("total 1"), r.stdout
    assert open(asm).read().count(" lds") >= 10  # both kernels really use LDS-direct loads


def test_checker_reports_inline_asm_that_writes_m0(tmp_path):
    """Round 5's review (weak #9): an LDS-direct load written as `s_mov_b32 m0, ...; buffer_load ... lds` with "m0" on the
    clobber list sits next to the compiler's own M0 bookkeeping without being part of it.  tests/asm_wait_m0.hip holds that form
    (reported) and the form the library uses since round 6 -- the address as an input operand pinned to M0, so the write is the
    compiler's (clean); and the build fails on the bad one."""
    asm = tmp_path / "m0.s"
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "tests", "asm_wait_m0.hip"), "-o", str(asm)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "reserved registers" in r.stderr  # the backend's own warning, from the bad kernel only:
    assert r.stderr.count("inline asm clobber list contains reserved registers") == 1, r.stderr[-1500:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_wait_check.py"), str(asm)], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1 and "m0_bad_kernel" in r.stdout and "writes M0" in r.stdout, r.stdout
    assert "m0_good_kernel" not in r.stdout and r.stdout.strip().endswith("total 1"), r.stdout
    txt = open(asm).read()
    good = txt[txt.index("_Z14m0_good_kernel"):]
    assert "s_mov_b32 m0" in good or "s_movk_i32 m0" in good or "m0," in good  # the compiler's write is there, outside the asm block
    build = os.path.join(ROOT, "audio_source_separation_amd", "csrc", "build.sh")
    env = dict(os.environ, ASSX_OBJ=str(tmp_path), ASSX_OUT=str(tmp_path / "lib.so"), ASSX_SRCS="../../tests/asm_wait_m0")
    r = subprocess.run(["bash", build], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "asm_wait_check FAILED" in r.stderr, (r.returncode, r.stderr[-1500:])
