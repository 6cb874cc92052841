"""GPU: the driver hooks of __graft_entry__.py in the orders a caller may use them -- in particular build() and smoke() in ONE process,
and the library loaded before torch has touched the device (PyTorch-ROCm bundles its own libamdhip64: whichever HIP runtime copy opens
the device first decides whether the other one's launches work; egogaussian_amd/lib.py load() brings torch's up first)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("code", [
    "import __graft_entry__ as g; g.build(); g.smoke()",
    "import egogaussian_amd.lib as lib; lib.load(); import __graft_entry__ as g; g.smoke()",
    "import __graft_entry__ as g; g.smoke()",
], ids=["build-then-smoke", "library-loaded-before-any-device-use", "smoke-alone"])
def test_entry_points_in_one_process(code):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "smoke: R=" in r.stdout and "radii mismatches 0" in r.stdout
