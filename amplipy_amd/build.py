"""In-tree build of libamplihip.so with hipcc for gfx950 (cross-compiles without a GPU) and of the
host-side BAM codec libampbam.so (g++, zlib)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "amplihip.hip")
DEPS = [SRC, os.path.join(_HERE, "..", "include", "amplihip.h")] + sorted(
    os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith(".hpp"))
OUT = os.path.join(_HERE, "libamplihip.so")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-o", OUT, SRC, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


BAM_SRC = os.path.join(_HERE, "csrc", "ampbam.cpp")
BAM_DEPS = [BAM_SRC, os.path.join(_HERE, "..", "include", "ampbam.h")]
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
