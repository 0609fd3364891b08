"""CPU: randomized test of the device allocator's arena bookkeeping (csrc/arena.h: best-fit free list with coalescing
inside a range that grows and shrinks in chunks), compiled for the host with AddressSanitizer + UBSan: allocated and free
blocks must tile the mapped range exactly after every operation."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arena_bookkeeping_fuzz(tmp_path):
    exe = str(tmp_path / "arena_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "arena_fuzz.cpp")])
    for seed in (1, 2, 3, 4):
        r = subprocess.run([exe, str(seed)], capture_output=True, text=True)
        assert r.returncode == 0 and "ARENA-FUZZ-OK" in r.stdout, r.stderr[-2000:]
