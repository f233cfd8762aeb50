"""The C ABI from a plain-C process (examples/c_abi_demo.c): no Python, no PyTorch, the system HIP runtime --
the situation of a Pothos plugin.  Builds the example with gcc and runs it."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_runs():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    r = subprocess.run([os.path.join(ROOT, "examples", "c_abi_demo")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")


def test_plain_c_client_drives_the_native_multi_device_path():
    """examples/c_shard_demo.c: pcx_shard_* over RCCL (one shard per visible device; bit-identical to the plain call on
    a one-GPU box) and two peer-copy shards on device 0 -- from C, against the system RCCL."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    r = subprocess.run([os.path.join(ROOT, "examples", "c_shard_demo")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
