"""CPU (needs hipcc only): the scatter loops of the partition / dedup kernels must not wait between their stores.

A value loaded (or an atomic issued) on a path that only some lanes take and consumed inside the conditional blocks of a
store loop makes the compiler put `s_waitcnt vmcnt(0)` in front of every store -- each store then waits for the one
before.  tools/isa_store_wait_audit.py finds that pattern in the ISA; this test keeps it out."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["msd.hip", "superk.hip"])
def test_no_waits_between_scattered_stores(tmp_path, src):
    asm = str(tmp_path / (src + ".s"))
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fopenmp", "-x", "hip", "--cuda-device-only",
                        "-S", os.path.join(ROOT, "spades_for_blackbird_amd", "csrc", src), "-o", asm],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_wait_audit.py"), asm], capture_output=True,
                       text=True, timeout=300)
    assert a.returncode == 0, a.stderr[-2000:]
    assert a.stdout.strip() == "", "kernels that wait between stores:\n" + a.stdout
