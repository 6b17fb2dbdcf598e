"""luw_create's placement search for the DDF array (csrc/luw_core.hip, tune_ddf_placement): with every candidate tried -- 1 GiB, 2 GiB, 512 MiB and
256 MiB chunks, hipMalloc -- the solver that keeps the fastest computes what the oracle computes, losers and all are released with it, and a second
large solver of the same process works (it does not search again)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", ["f32", "fp16c"])
def test_every_candidate_tried_and_results_equal_the_oracle(luw, dtype):
    env = dict(os.environ, LUW_TUNE_FAST="99", LUW_TUNE_VERBOSE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "placement_search_worker.py"), dtype], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "cycle 0 equal True" in r.stdout and "cycle 1 equal True" in r.stdout, r.stdout
    tried = [l for l in r.stderr.splitlines() if l.startswith("luw: placement candidate")]
    assert len(tried) == 7, r.stderr[-2000:]                            # the default mapping + six further draws, once per process
