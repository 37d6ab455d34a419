"""In-tree build of libamplihip.so with hipcc for gfx950 (cross-compiles without a GPU) and of the
host-side BAM codec libampbam.so (g++, zlib)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# translation units of libamplihip.so: the kernels + C ABI, and the insertion-event aggregation (its sort headers triple the
# compile time of whatever includes them, so it is built -- and cached -- on its own)
UNITS = ["amplihip.hip", "amp_ins.hip"]
SRC = os.path.join(CSRC, "amplihip.hip")
HEADERS = [os.path.join(_HERE, "..", "include", "amplihip.h")] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
UNIT_DEPS = {"amplihip.hip": HEADERS, "amp_ins.hip": [os.path.join(_HERE, "..", "include", "amplihip.h"), os.path.join(CSRC, "amp_ins.hpp")]}
DEPS = [os.path.join(CSRC, u) for u in UNITS] + HEADERS
OUT = os.path.join(_HERE, "libamplihip.so")
OBJ_DIR = os.path.join(_HERE, "build")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    for u in UNITS:
        src = os.path.join(CSRC, u)
        obj = os.path.join(OBJ_DIR, u.replace(".hip", ".o"))
        stale = force or extra_flags or not os.path.isfile(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in [src] + UNIT_DEPS[u])
        if stale:
            cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


BAM_SRC = os.path.join(_HERE, "csrc", "ampbam.cpp")
BAM_DEPS = [BAM_SRC, os.path.join(_HERE, "csrc", "amp_inflate.hpp"), os.path.join(_HERE, "..", "include", "ampbam.h")]
BAM_OUT = os.path.join(_HERE, "libampbam.so")


def build_bam(force=False, verbose=False):
    if not force and os.path.isfile(BAM_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(BAM_OUT) for d in BAM_DEPS):
        return BAM_OUT
    cmd = [shutil.which("g++") or "g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", BAM_OUT, BAM_SRC, "-lz", "-ldl", "-pthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return BAM_OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_bam(force=True, verbose=True)
