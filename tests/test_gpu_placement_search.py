"""luw_create's placement search for the DDF array (csrc/luw_placement.hpp, tune_ddf_placement): with every candidate tried -- 1 GiB chunks, 1 GiB chunks
again, 4 GiB and 2 GiB chunks, hipMalloc and 512 MiB chunks, each a fresh draw of physical memory (the arrays tried before stay mapped until the search ends),
all of them released before luw_create returns -- the solver that keeps the fastest computes what the oracle computes, the device's free memory after luw_create
is what ONE DDF array costs, and a second large solver of the same process is allocated as the winner's kind without a search."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", ["f32", "fp16c"])
def test_every_candidate_tried_and_results_equal_the_oracle(luw, dtype):
    env = dict(os.environ, LUW_TUNE_FAST="99")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "placement_search_worker.py"), dtype], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "cycle 0 equal True" in r.stdout and "cycle 1 equal True" in r.stdout, r.stdout
    info = [eval(l.split("placement ", 1)[1]) for l in r.stdout.splitlines() if l.startswith("placement ")]
    # (six candidates: the default mapping + five further draws, once per process)
    assert info[0]["candidates_tried"] == 6 and info[0]["probe_TBps"] > 1.0 and info[0]["create_s"] > 0
    assert info[1]["candidates_tried"] == 0 and "first search kept" in info[1]["kept"] and info[1]["kept"].startswith(info[0]["kept"])
    used = [float(l.split()[-2]) for l in r.stdout.splitlines() if l.startswith("device memory used by the solver")]
    assert all(u < 1.35 * 2.6 + 1.0 for u in used), used               # GB: one DDF array (2.5 GB) + fields, nothing of the search left


def test_a_slow_first_draw_is_replaced_and_everything_else_released(luw):
    """the path a slow box takes: the array in place loses to a later draw (here: made to, luw_dev_inject_fault), the winner becomes the solver's DDF array,
    every other candidate -- the first one included -- is released before luw_create returns, and the results are the oracle's"""
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "placement_search_worker.py"), "f32", "replace"], capture_output=True, text=True,
        timeout=600,
        env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "cycle 0 equal True" in r.stdout and "cycle 1 equal True" in r.stdout, r.stdout
    info = [eval(l.split("placement ", 1)[1]) for l in r.stdout.splitlines() if l.startswith("placement ")]
    assert 2 <= info[0]["candidates_tried"] <= 6 and info[0]["probe_TBps"] > 1.0
    used = [float(l.split()[-2]) for l in r.stdout.splitlines() if l.startswith("device memory used by the solver")]
    assert all(u < 1.35 * 2.6 + 1.0 for u in used), used               # GB: one DDF array (2.5 GB) + fields, nothing of the search left
