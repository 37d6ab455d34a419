"""CPU-side checks of the C-ABI library: it builds, loads and exports every declared symbol."""
import ctypes as C
import os
import re
import shutil

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.isfile(
    os.path.join(ROOT, "amplipy_amd", "libamplihip.so")), reason="no hipcc and no prebuilt library")


def test_library_exports_every_declared_symbol():
    from amplipy_amd import build, lib
    if shutil.which("hipcc"):
        build.build()
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "amplihip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(amp_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libamplihip.so does not export %s" % name
    assert set(lib.EXPORTS) == declared
    assert L.amp_version() == 1
    assert L.amp_strerror(-4) == b"no usable GPU device"
    assert L.amp_read_status_exception(4) == b"KeyError"


def test_no_gpu_means_loud_failure():
    """There is no CPU fallback: without a device the engine refuses to start."""
    import torch
    from amplipy_amd import lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.AmpliHipError):
        lib.Engine(1000)


def test_primer_tables_host_entry_point_matches_golden():
    from amplipy_amd import lib
    from tests import helpers as H
    meta = H.load_json("primer_tables.json")
    tabs = np.load(H.GOLDEN + "/primer_tables.npz")
    bed = [l.rstrip("\r\n").split("\t") for l in open(H.GOLDEN + "/data/example_primers.bed") if l.strip()]
    primers = [(int(f[1]), int(f[2])) for f in bed]
    for off in (0, 5):
        mn, mx, mpl = lib.find_overlapping_primers(meta["example"]["ref_len"], primers, off)
        assert mpl == 30
        assert np.array_equal(mn, tabs["example_off%d_min_start" % off])
        assert np.array_equal(mx, tabs["example_off%d_max_end" % off])
    for k, s in enumerate(meta["random_sets"]):
        mn, mx, mpl = lib.find_overlapping_primers(s["ref_len"], s["primers"], s["offset"])
        assert mpl == s["max_primer_len"]
        assert np.array_equal(mn, tabs["rand%d_min_start" % k])
        assert np.array_equal(mx, tabs["rand%d_max_end" % k])


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under amplipy_amd/ may reference it."""
    pkg = os.path.join(ROOT, "amplipy_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dp, fn)).read()
                assert "oracle" not in text.replace("# oracle-free", ""), "%s mentions the oracle" % fn
