"""GPU tier: the C ABI driven from a plain C99 program (``hirag-prod_amd/csrc/abi_example.c``, built by the
Makefile next to ``libhvs.so``): search on all three routes, range search and PageRank, each checked inside the
program against host arithmetic.  No Python between the caller and the library."""

import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_c_program_against_the_abi():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(here, "hirag-prod_amd", "csrc", "abi_example")
    if not os.path.exists(exe):
        pytest.skip("abi_example not built (`make -C hirag-prod_amd/csrc`, which build() runs)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, f"exit {out.returncode}: {out.stdout} {out.stderr}"
    assert "abi_example ok" in out.stdout
