"""Every `File.cs:line` citation in the oracle, the header, the kernels and the hosts points into the mounted reference tree at lines that
mention the symbol it is cited for (scripts/check_citations.py).  Build container only: the reference is not on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_citations_point_at_what_they_name():
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference tree not mounted")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_citations.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " 0 problem(s)" in p.stdout
