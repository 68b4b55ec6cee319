"""The driver's entry points in ONE process, in the order build() -> smoke(): the library is loaded before anything
has imported torch (regression: two HIP runtimes in the process, `no ROCm-capable device` at the first copy)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "smoke ok" in p.stdout
