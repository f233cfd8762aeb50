"""CPU suite: the -DPCX_WITH_POTHOS branch of the block sources is parsed and type-checked.

INTEGRATION.md 2 builds csrc/blocks/comms_blocks.cpp and fir_designer.cpp into a Pothos plugin module with -DPCX_WITH_POTHOS,
where `namespace pcxfw` IS Pothos (registration as /root/reference/filter/FIRFilter.cpp:385-389, buffer managers as :196-199 and
fft/FFT.cpp:54-59, registered calls as FIRFilter.cpp:113-124).  PothosCore is not installable in this image or on the GPU box, so
that branch used to meet no compiler at all ("written from memory", never syntax-checked).  tests/pothos_decl holds a
DECLARATION-ONLY header set of the PothosCore surface SURVEY.md 8b enumerates; this test runs g++ -fsyntax-only over the block
sources against it.  It proves nothing about PothosCore's behaviour -- no body, nothing links -- it keeps the branch from rotting:
a renamed member, a wrong override, a missing include fail here."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = os.path.join(ROOT, "pothoscomms_amd", "csrc", "blocks")
FLAGS = ["-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-DPCX_WITH_POTHOS",
         "-I" + os.path.join(ROOT, "tests", "pothos_decl"), "-I" + os.path.join(ROOT, "include"), "-I" + BLOCKS]

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


@pytest.mark.parametrize("src", ["comms_blocks.cpp", "fir_designer.cpp"])
def test_block_sources_type_check_against_the_pothos_surface(src):
    r = subprocess.run(["g++"] + FLAGS + [os.path.join(BLOCKS, src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_the_pothos_branch_is_what_was_checked():
    """the same command on a translation unit that uses a member the Pothos surface does not have must fail: the check above is
    not passing because the branch is compiled out"""
    probe = '#include "pcx_framework.hpp"\nvoid f(pcxfw::InputPort *p) { p->labels(); p->workInfoMutable(); }\n'
    r = subprocess.run(["g++"] + FLAGS + ["-x", "c++", "-"], input=probe, capture_output=True, text=True)
    assert r.returncode != 0 and "workInfoMutable" in r.stderr
    ok = '#include "pcx_framework.hpp"\nsize_t f(pcxfw::InputPort *p) { return p->elements(); }\nstatic_assert(sizeof(Pothos::BufferManagerArgs) > 0, "");\n'
    r = subprocess.run(["g++"] + FLAGS + ["-x", "c++", "-"], input=ok, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_the_runner_is_not_part_of_a_pothos_build():
    """runner.cpp plays the scheduler for the bundled runtime; inside Pothos the framework does that, and the file says so"""
    r = subprocess.run(["g++"] + FLAGS + [os.path.join(BLOCKS, "runner.cpp")], capture_output=True, text=True)
    assert r.returncode != 0 and "the runner drives the bundled runtime" in r.stderr
