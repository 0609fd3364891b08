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
SOURCES = ["primitives.hip", "msd.hip", "superk.hip", "reads.hip", "count.hip", "extindex.hip", "tipclip.hip", "readfilter.hip", "unitigs.hip", "group.hip"]
HEADERS = ["bbk_internal.h", "kmer_ops.h", "msd.h", "accum.h", "arena.h", os.path.join("..", "..", "include", "bbk.h")]
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


def _llvm_tool(name):
    """llvm-objdump / llvm-readelf: next to the hipcc in use, in the ROCm tree, or on PATH; None if absent."""
    import shutil
    cands = []
    hipcc = shutil.which(_hipcc()) or _hipcc()
    if os.path.isabs(hipcc):
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", name), os.path.join(root, "llvm", "bin", name)]
    cands += [os.path.join("/opt/rocm/lib/llvm/bin", name), shutil.which(name)]
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


def kernel_resources(objs=None):
    """Per-kernel scratch and spill figures from the code-object metadata of the built objects
    (llvm-objdump --offloading + llvm-readelf --notes): [(object, kernel, scratch_bytes, sgpr_spills, vgpr_spills)].
    Returns None when the llvm tools are not installed."""
    import re
    import shutil
    import tempfile
    objdump, readelf = _llvm_tool("llvm-objdump"), _llvm_tool("llvm-readelf")
    if not objdump or not readelf:
        return None
    objs = objs or [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    out = []
    for obj in objs:
        if not os.path.exists(obj):
            continue
        d = tempfile.mkdtemp(prefix="bbk_res_")
        try:
            local = os.path.join(d, os.path.basename(obj))
            shutil.copy(obj, local)
            r = subprocess.run([objdump, "--offloading", local], capture_output=True, text=True, cwd=d)
            if r.returncode != 0:
                raise RuntimeError("llvm-objdump --offloading %s failed (%d): %s" % (obj, r.returncode, r.stderr[-500:]))
            for f in os.listdir(d):
                if ARCH not in f:
                    continue
                rn = subprocess.run([readelf, "--notes", os.path.join(d, f)], capture_output=True, text=True)
                if rn.returncode != 0:
                    raise RuntimeError("llvm-readelf --notes failed (%d): %s" % (rn.returncode, rn.stderr[-500:]))
                notes = rn.stdout
                name = None
                vals = {}
                for line in notes.splitlines():
                    m = re.match(r"\s*\.name:\s+(\S+)", line)
                    if m:
                        if name:
                            out.append((os.path.basename(obj), name, vals.get("private_segment_fixed_size", 0),
                                        vals.get("sgpr_spill_count", 0), vals.get("vgpr_spill_count", 0)))
                        name, vals = m.group(1), {}
                        continue
                    m = re.match(r"\s*\.(private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):\s+(\d+)", line)
                    if m and name:
                        vals[m.group(1)] = int(m.group(2))
                if name:
                    out.append((os.path.basename(obj), name, vals.get("private_segment_fixed_size", 0),
                                vals.get("sgpr_spill_count", 0), vals.get("vgpr_spill_count", 0)))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # .name also appears for kernel ARGUMENTS in the metadata; kernels are the entries with a mangled / plain symbol
    return [r for r in out if r[1].startswith("_Z") or r[1].startswith("k_")]


def check_resources():
    """A build hygiene rule, not a correctness one (scratch by itself computes correctly on this pool): no
    kernel of the product library may use scratch (private segment) or spill vector registers -- either one costs a
    kernel of this path its occupancy.  Raises with the offending kernels; warns and returns None when the llvm tools
    that read the code-object metadata are not installed."""
    res = kernel_resources()
    if res is None:
        print("warning: llvm-objdump / llvm-readelf not found: kernel resource check skipped", file=sys.stderr)
        return None
    if not res:
        raise RuntimeError("no kernel metadata found in the built objects (objects not built?)")
    bad = [r for r in res if r[2] != 0 or r[4] != 0]
    if bad:
        raise RuntimeError("kernels using scratch / spilling VGPRs:\n" +
                           "\n".join("  %s %s scratch=%d sgpr_spill=%d vgpr_spill=%d" % r for r in bad))
    return res


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    suffix = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--suffix=")), "")
    print(build(force="--force" in sys.argv, verbose=True, extra_flags=extra, suffix=suffix))
    if not suffix:
        r = check_resources()
        if r is not None:
            print("%d kernels: no scratch, no VGPR spills (SGPR spills in %d)" % (len(r), sum(1 for x in r if x[3])))
