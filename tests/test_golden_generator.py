"""The committed fixture generator still reproduces tests/golden from the reference (runs only where the
reference is mounted: never on the GPU box), and it loads the reference from its file, not by name."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/AmpliPy.py"

pytestmark = pytest.mark.skipif(not os.path.isfile(REF), reason="reference not mounted here")


def test_generator_loads_the_reference_by_path():
    # a root-level AmpliPy.py (the drop-in entry point) exists and must not shadow the reference
    assert os.path.isfile(os.path.join(ROOT, "AmpliPy.py"))
    code = ("import sys; sys.argv=['x']; sys.path.insert(0, %r); import importlib.util as u;"
            "s=u.spec_from_file_location('mg', %r); m=u.module_from_spec(s); s.loader.exec_module(m);"
            "print(m.REF.__file__); print(m.REF.__name__)") % (ROOT, os.path.join(ROOT, "tools", "make_golden.py"))
    out = subprocess.run([sys.executable, "-B", "-c", code], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
    assert os.path.realpath(out[0]) == os.path.realpath(REF)
    assert out[1] == "amplipy_reference"


def test_fixtures_regenerate_identically():
    r = subprocess.run([sys.executable, "-B", os.path.join(ROOT, "tools", "make_golden.py"), "--check"], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical to tests/golden" in r.stdout
