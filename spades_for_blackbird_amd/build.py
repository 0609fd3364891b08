"""Builds libbbk.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m spades_for_blackbird_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbbk.so")
SOURCES = ["primitives.hip", "msd.hip", "reads.hip", "count.hip", "extindex.hip", "tipclip.hip", "readfilter.hip", "unitigs.hip"]
HEADERS = ["bbk_internal.h", "kmer_ops.h", "msd.h", "accum.h", os.path.join("..", "..", "include", "bbk.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fopenmp", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, extra_flags=None, suffix=""):
    """extra_flags / suffix build an experimental variant (objects and library get the suffix)."""
    global LIB
    hipcc = _hipcc()
    extra_flags = list(extra_flags or [])
    lib = LIB.replace(".so", suffix + ".so") if suffix else LIB
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", suffix + ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + extra_flags + ["-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-fopenmp", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    suffix = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--suffix=")), "")
    print(build(force="--force" in sys.argv, verbose=True, extra_flags=extra, suffix=suffix))
