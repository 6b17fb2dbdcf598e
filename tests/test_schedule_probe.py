"""DomainDecomposedLBM.choose_schedule -- the start-up probe `bench.py --gpus N` runs on x-split cuts: real steps under "shell first" and under "whole box,
then the exchange", the slowest rank decides, the default stays unless the whole box is more than 2 % faster.  Here: the decision rule and the bookkeeping on
a stub backend (no GPU); the values across schedule switches are held to the oracle on the GPU (tests/test_gpu_bench_workloads.py, the `switch` cases)."""
import types

import pytest

from latticeurbanwind_amd.distributed import DomainDecomposedLBM


class Stream:
    def synchronize(self): pass


class Backend:
    def __init__(self): self.comm, self.compute, self.configured = Stream(), Stream(), []
    def configure_step(self, overlap): self.configured.append(bool(overlap))


def sim_with(times, monkeypatch, can_overlap=True):
    """a DomainDecomposedLBM whose run() takes `times[overlap]` seconds per step on a fake clock"""
    import time
    import torch
    s = DomainDecomposedLBM.__new__(DomainDecomposedLBM)
    s.backend, s.overlap, s.initialized = Backend(), True, True
    s.layout = types.SimpleNamespace(can_overlap=lambda: can_overlap)
    clock = [0.0]
    s.run = lambda steps, **kw: clock.__setitem__(0, clock[0] + steps * times[s.overlap])
    monkeypatch.setattr(time, "perf_counter", lambda: clock[0])
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return s


@pytest.mark.parametrize("shell_first,whole,kept_overlap", [(3.60e-3, 3.59e-3, True), (3.60e-3, 3.50e-3, False), (3.50e-3, 3.60e-3, True)])
def test_the_default_stays_unless_the_whole_box_is_clearly_faster(monkeypatch, shell_first, whole, kept_overlap):
    s = sim_with({True: shell_first, False: whole}, monkeypatch)
    r = s.choose_schedule(steps=20)
    assert s.overlap is kept_overlap and r["kept"].startswith("shell first" if kept_overlap else "whole box")
    assert abs(r["shell_first_ms"] - shell_first * 1e3) < 1e-9 and abs(r["whole_box_ms"] - whole * 1e3) < 1e-9 and r["probe_steps"] == 20
    assert s.backend.configured[-1] is kept_overlap              # the library's step context follows the choice


def test_the_slowest_rank_decides_and_every_rank_calls_the_collective_once(monkeypatch):
    s = sim_with({True: 3.6e-3, False: 3.4e-3}, monkeypatch)     # locally the whole box wins by 6 % ...
    calls = []

    def reduce_max(v):
        calls.append(list(v)); return [3.6, 3.58]                  # ... but on another rank it does not
    r = s.choose_schedule(steps=10, reduce_max=reduce_max)
    assert len(calls) == 1 and s.overlap is True and r["whole_box_ms"] == 3.58


def test_a_schedule_that_fails_locally_is_never_chosen_and_the_collective_still_runs(monkeypatch):
    s = sim_with({True: 3.6e-3, False: 3.0e-3}, monkeypatch)
    good = s.run

    def run(steps, **kw):
        if not s.overlap: raise RuntimeError("whole-box launch failed")
        good(steps)
    s.run = run
    calls = []
    r = s.choose_schedule(steps=5, reduce_max=lambda v: (calls.append(v), v)[1])
    assert len(calls) == 1 and s.overlap is True and r["whole_box_ms"] == float("inf") and r["kept"].startswith("shell first")


def test_nothing_to_choose_without_an_interior(monkeypatch):
    assert sim_with({True: 1.0, False: 1.0}, monkeypatch, can_overlap=False).choose_schedule() is None
