"""bench.py's N > 1 self-check (digests of every rank's owned cells against the undivided CPU oracle) on CPU: world-2 gloo groups over the
oracle test double must pass it, and a single corrupted value on one rank must fail it.  The GPU run of the same code is bench.py
--gpus N itself (and tests/test_gpu_bench_distributed.py on the one-GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tmp_path, gN, D, fp16c, corrupt):
    world = D[0] * D[1] * D[2]
    out = str(tmp_path / "digests.json")
    port = 29500 + ((os.getpid() + 31 + corrupt) % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
        str(port),
           os.path.join(ROOT, "tests", "bench_parity_worker.py"), *map(str, gN), *map(str, D), str(int(fp16c)), str(corrupt), out]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.load(open(out))


@pytest.mark.parametrize("gN,D,fp16c", [((24, 16, 12), (2, 1, 1), False), ((16, 24, 12), (1, 2, 1), True), ((32, 16, 8), (4, 2, 1), False)])
def test_selfcheck_passes_on_a_correct_decomposed_run(tmp_path, gN, D, fp16c):
    # (the third case: eight ranks, BASELINE's literal cut)
    res = run(tmp_path, gN, D, fp16c, 0)
    assert res["bad"] == [[] for _ in range(D[0] * D[1] * D[2])] and res["max_abs_uy"] > 0.0


def test_selfcheck_sees_one_wrong_value(tmp_path):
    res = run(tmp_path, (24, 16, 12), (2, 1, 1), False, 1)
    assert res["bad"] == [[], ["u"]]


def test_parity_tile_shapes():
    sys.path.insert(0, ROOT)
    from benchmarks import multi
    from latticeurbanwind_amd.distributed import DomainLayout, choose_decomposition
    for world in (2, 4, 8):
        for D in (choose_decomposition(world, split_x=True), choose_decomposition(world)):
            gN = multi.parity_tile(world, D)
            lay = DomainLayout(gN, D, world - 1)
            assert lay.can_overlap()
            # zones thinner than every rank's block: a domain that does not own a face never lies inside that face's zone
            assert all(multi.PARITY_NUDGE_CELLS < g // d for g, d in zip(gN, D)) and multi.PARITY_SPONGE_CELLS < gN[2] // D[2]
            if D[0] > 1:
                for x_shell in (64, 128):                                          # FP32 / FP16C slabs
                    l2 = DomainLayout(gN, D, world - 1, x_shell=x_shell)
                    assert l2.interior_box()[1] - l2.interior_box()[0] >= 128      # an interior between the two x slabs, wide enough for the pair kernel


def test_group_host_variant_past_its_time_limit_costs_only_its_own_block(monkeypatch):
    """bench.py --gpus N measures the one-process host in child processes, one variant each, under a time limit: a child that does not answer
    (here: a limit of one second, which the interpreter start alone exceeds) is killed by its PID and leaves an error in its block -- the line
    of the RCCL measurement is still printed."""
    import argparse
    from benchmarks import multi
    monkeypatch.setattr(multi, "GROUP_HOST_TIMEOUT_S", 1)
    args = argparse.Namespace(dtype="f32", arith="native", kernel="auto", steps=4, warmup=2, coriolis=False, no_buildings=False, no_parity=True)
    out = multi.run_group_host(args, (2, 1, 1), (128, 64, 64), [0, 0])
    assert set(multi.GROUP_HOST_VARIANTS) <= set(out)
    for label in multi.GROUP_HOST_VARIANTS:
        assert "no result within 1 s" in out[label]["error"] and out[label]["process_wall_s"] < 10
