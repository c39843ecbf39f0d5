"""Host-side sanitizer runs of the library's threaded code (tools/sanitize/run.sh; round 5's review, weak #11).

csrc/assx_api.hip + csrc/assx_xfer.hip are compiled host-only under AddressSanitizer + UndefinedBehaviorSanitizer and
under ThreadSanitizer, linked against a host emulation of the HIP calls they make (tools/sanitize/hip_stub.cpp:
asynchronous in-order streams, events) and driven through uploads / downloads of ragged sizes with 1, 3 and 8 pool
threads and through the ticket slots under more streams than slots.  No GPU; ~15 s.  The second test seeds the bug the ring
exists to prevent -- a staging buffer refilled while its DMA is still reading it -- and expects the harness to say so."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tools", "sanitize", "run.sh")


def _need_toolchain():
    if not os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("hipcc / clang++ not available")


def test_host_side_code_is_clean_under_asan_ubsan_and_tsan(tmp_path):
    _need_toolchain()
    log = tmp_path / "san.log"
    r = subprocess.run(["bash", RUN, str(log)], capture_output=True, text=True, timeout=900)
    txt = log.read_text()
    assert r.returncode == 0, txt[-3000:]
    assert txt.count("ticket slots: ok") == 2 and txt.count("transfers with 8 host thread(s): ok") == 2
    assert "sanitizer runs: CLEAN" in txt and "ThreadSanitizer" not in txt and "AddressSanitizer" not in txt
    assert "runtime error" not in txt


def test_harness_reports_a_staging_buffer_reused_too_early(tmp_path):
    _need_toolchain()
    src = os.path.join(ROOT, "audio_source_separation_amd", "csrc")
    mut = tmp_path / "a" / "b"  # the sources include "../../include/assx.h"
    mut.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(ROOT, "include", "assx.h"), tmp_path / "include" / "assx.h")
    for f in os.listdir(src):
        if f.endswith((".hip", ".hpp")):
            shutil.copy(os.path.join(src, f), mut / f)
    p = mut / "assx_xfer.hip"
    s = p.read_text()
    wait = '    if (x->busy[s]) XF_HIP(ctx, hipEventSynchronize(x->ev[s]), "assx_upload: hipEventSynchronize");\n'
    assert s.count(wait) == 1
    p.write_text(s.replace(wait, ""))
    log = tmp_path / "mut.log"
    r = subprocess.run(["bash", RUN, str(log)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASSX_SAN_CSRC=str(mut)))
    txt = log.read_text()
    assert r.returncode != 0 and "sanitizer runs: FAILED" in txt
    assert "ThreadSanitizer: data race" in txt or "back[i] == want" in txt, txt[-3000:]
