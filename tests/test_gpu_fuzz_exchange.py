"""A slice of the two exchange fuzzers in the GPU suite (the long runs are profiles/r04_fuzz_exchange.txt, profiles/r05_group_one_phase.txt and, over the
three transports, profiles/r06_exchange_fuzz.txt): random lattices, cuts, formats, thermal lattice, solids on border columns and corner lines -- the
one-round exchange against the three phases, bit for bit, for the one-process-per-GPU host (one rank, its own neighbour) and for the one-process host (up
to eight domains on the one GPU, peer / staged / rccl, also against the CPU oracle)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases,seed", [("fuzz_exchange_gpu.py", 24, 3), ("fuzz_exchange_group_gpu.py", 40, 5)])
def test_exchange_fuzz_slice(luw, script, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), str(cases), str(seed)], capture_output=True, text=True, timeout=900,
        cwd=ROOT)
    assert r.returncode == 0 and "fuzz: %d cases, 0 different" % cases in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
