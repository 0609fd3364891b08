"""Builds the C++ host programs (the CLI drop-ins) against libbbk.so with g++ (in-tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "host")
BIN = os.path.join(HERE, "bin")
PROGRAMS = {"spades-kmercount": "kmercount_main.cpp", "spades-gbuilder": "gbuilder_main.cpp",
            "spades-kmer-estimating": "kmer_estimating_main.cpp", "spades-read-filter": "read_filter_main.cpp",
            "bbk-fastx-dump": "fastx_dump_main.cpp"}
HEADERS = ["common.hpp", "dataset.hpp", "fastx.hpp", "ingest.hpp", "multi.hpp"]


def build(force=False, verbose=False):
    os.makedirs(BIN, exist_ok=True)
    out = []
    deps = [os.path.join(HOST, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "bbk.h"),
                                                      os.path.join(HERE, "libbbk.so")]
    for name, src in PROGRAMS.items():
        exe = os.path.join(BIN, name)
        srcp = os.path.join(HOST, src)
        stale = not os.path.exists(exe) or any(os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(exe)
                                               for d in deps + [srcp])
        if force or stale:
            cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-fopenmp", "-o", exe, srcp, "-L" + HERE, "-lbbk", "-lz",
                   "-Wl,-rpath,$ORIGIN/.."]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        out.append(exe)
    return out


def build_sanitized(force=False):
    """Host sanitizer build (the reference's SPADES_ENABLE_ASAN option, cmake/options.cmake:18-23): the host-only
    parser tool compiled with AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU ASAN is not available on
    this pool).  tests/test_ingest.py runs the parser corner cases through it."""
    os.makedirs(BIN, exist_ok=True)
    exe = os.path.join(BIN, "bbk-fastx-dump-asan")
    srcp = os.path.join(HOST, "fastx_dump_main.cpp")
    deps = [os.path.join(HOST, h) for h in HEADERS] + [srcp]
    stale = not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps if os.path.exists(d))
    if force or stale:
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fopenmp", "-fsanitize=address,undefined",
                               "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-o", exe, srcp, "-lz"])
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
