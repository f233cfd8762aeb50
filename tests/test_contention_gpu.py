"""Several processes on one GPU (how Pothos deployments and this suite's own -n runs share a device): every host-pointer
call must still return what a lone process gets.  tools/contention_probe.py creates handles and calls them at once in eight
processes; before the fixes recorded in profiles/r02/contention.md a new FreqDemod handle's first output was wrong in about
one creation in 30 under this load."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_processes_get_the_lone_process_answers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "contention_probe.py"), "8", "40"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("  ")]
    assert len(lines) >= 7, r.stdout[-2000:]
    assert all(l.split()[-1] == "0" for l in lines), r.stdout[-3000:]
