"""Build the gfx950 hot-path library (libpengk.so) in-tree with hipcc.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with the repo
snapshot to the GPU box.  Usage:  python peng-motif_amd/build.py [--force] [--verbose]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("PENGK_BUILD_OUT") or os.path.join(HERE, "libpengk.so")  # PENGK_BUILD_OUT: variant builds for tools/ab.sh
SOURCES = ["api.hip", "count.hip", "stats.hip", "iupac.hip", "em.hip", "em_fused.hip", "em_legacy.hip", "similarity.hip", "seqsum.hip", "comm.hip", "pack.cpp"]
HEADERS = [os.path.join(CSRC, "pengk_internal.h"), os.path.join(CSRC, "em_common.h"), os.path.join(CSRC, "em_serial.h"), os.path.join(CSRC, "seqsum.h"), os.path.join(ROOT, "include", "pengk.h")]
# -ffp-contract=off: float32 arithmetic must round exactly like the reference's scalar x86 code.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


EXTRA = os.environ.get("PENGK_EXTRA_FLAGS", "").split()  # experiments, e.g. -DPENGK_ABLATE=8


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    objdir = os.path.join(HERE, "build") if not os.environ.get("PENGK_BUILD_OUT") else LIB + ".obj"
    os.makedirs(objdir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(objdir, s + ".o")
        cmd = [hipcc()] + FLAGS + EXTRA + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(CSRC, s), "-o", o]
        if s.endswith(".cpp"):
            cmd.insert(1, "-xhip")
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(out.decode(errors="replace"))
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
