"""`bench.py --gpus N` on the one-GPU test box: the N > 1 code path end to end -- process group, self-check of a decomposed urban tile
against the CPU oracle (must pass, and the line must say what was compared), timing blocks, per-rank topology, the one-process
multi-domain host measured by rank 0 -- with all ranks on GPU 0 and the faces staged through gloo (--share-device).  What a node adds
is the RCCL wire; the code above it is this."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def full_record(stdout, path):
    """stdout holds ONE line of at most 4 KB (benchmarks/line.py) that names the file with everything measured; returns (line, full record)"""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    assert len(lines[0]) <= 4096
    line = json.loads(lines[0])
    assert line["secondary_file"] == path
    return line, json.load(open(path))


def bench(world, *extra, env=None, tmp_path=None):
    port = 29500 + ((os.getpid() + 97 + world) % 2000)
    path = str(tmp_path / "full.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
        str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2", "--share-device", "0", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, LUW_BENCH_FULL_JSON=path, **(env or {})))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    line, out = full_record(r.stdout, path)
    # the printed line: the contract's keys, the communicator, and per rank its PCI bus id and the link type to each halo neighbour
    assert line["n_gpus"] == world and line["value"] == out["value"] and line["rccl"]["world_size"] == world and line["roofline"]["frac"] > 0
    assert len(line["ranks"]) == world and all(r_["bus"] and r_["links"] for r_ in line["ranks"]) and line["parity"]["ok"] is True
    return out


def test_two_ranks_self_check_and_blocks(luw, tmp_path):
    out = bench(2, "--size", "384", "64", "64", tmp_path=tmp_path)
    assert out["n_gpus"] == 2 and out["config"]["n_gpu"] == [2, 1, 1] and out["value"] > 0
    # the start-up probe of the two step schedules (x is split): both timed on the slowest rank, one kept, and the line says which
    sp = out["config"]["schedule_probe"]
    assert sp["shell_first_ms"] > 0 and sp["whole_box_ms"] > 0 and sp["probe_steps"] == 20 and sp["kept"] in ("shell first, exchange beside the interior",
        "whole box, then the exchange")
    assert ("overlapped with the interior" in out["config"]["halo_exchange"]) == sp["kept"].startswith("shell first")
    par = out["parity"]
    assert par["ok"] and len(par["cases"]) == 4                         # literal cut and x-whole cut, FP32 and FP16C + Coriolis
    for c in par["cases"]:
        assert c["equal"] and c["mismatches"] == [] and c["steps"] == 8 and c["max_abs_uy"] > 0 and c["cells_compared"] == c["lattice"][0] * c["lattice"][
            1] * c["lattice"][2]
    assert {tuple(c["n_gpu"]) for c in par["cases"]} == {(2, 1, 1), (1, 2, 1)}
    for r in out["per_rank"]:
        assert r["pci_bus_id"] and r["halo_bytes_out_per_step"] == 2 * 5 * 64 * 64 * 4 and set(r["links"]) == {"x+", "x-"}
    sec = out["secondary"]
    assert sec["x_whole_n_gpu"]["n_gpu"] == [1, 2, 1] and sec["x_whole_n_gpu"]["value"] > 0
    gh = sec["group_host"]
    for label in ("peer", "peer_threads", "rccl"):
        assert gh[label]["parity"]["equal"] and gh[label]["value"] > 0, gh[label]
    assert gh["peer"]["direct_peer_stores"] and not gh["rccl"]["direct_peer_stores"]


def test_plain_command_starts_its_own_ranks(luw, tmp_path):
    """`python3 bench.py --gpus 2 ...` with NO launcher around it -- the form the driver's SCALE runs use: the process starts its ranks as a child
    torch.distributed.run (benchmarks/launch.py), their ONE line comes out of its stdout, its exit code is theirs"""
    path = str(tmp_path / "full.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "0", "--size", "384", "64", "64", "--steps", "4", "--warmup", "2",
        "--no-group-host"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, LUW_BENCH_FULL_JSON=path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    assert "without a launcher: starting -m torch.distributed.run" in r.stderr
    line, out = full_record(r.stdout, path)
    assert line["n_gpus"] == 2 and line["rccl"]["world_size"] == 2 and line["value"] > 0 and line["parity"]["ok"] is True
    assert out["config"]["n_gpu"] == [2, 1, 1] and len(out["per_rank"]) == 2


def test_four_ranks_fp16c_coriolis(luw, tmp_path):
    # (no one-process host here: four ranks, this process and a child of rank 0 would be the six processes a test box allows on its GPU)
    out = bench(4, "--size", "384", "64", "64", "--dtype", "fp16c", "--coriolis", "--no-group-host", tmp_path=tmp_path)
    assert out["config"]["n_gpu"] == [2, 2, 1] and out["parity"]["ok"] and out["value"] > 0


def test_one_rank_over_the_real_rccl_process_group(luw, tmp_path):
    """the N > 1 code path with ONE rank and no --share-device: the RCCL process group itself (high-priority stream option, device id), the CPU-side gloo
    group beside it, object collectives over RCCL, the self-check and the timing blocks -- everything of `bench.py --gpus N` that does not need a second GPU"""
    port = 29500 + ((os.getpid() + 311) % 2000)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-distributed", "--size", "256", "64", "64", "--steps", "4", "--warmup", "2"]
    path = str(tmp_path / "full.json")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
        LUW_BENCH_FULL_JSON=path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line, out = full_record(r.stdout, path)
    assert line["rccl"]["version"] and line["rccl"]["world_size"] == 1
    assert out["n_gpus"] == 1 and out["config"]["n_gpu"] == [1, 1, 1] and out["value"] > 0 and out["parity"]["ok"]
    assert "RCCL" in out["config"]["halo_exchange"] and out["config"]["rccl_version"]


@pytest.mark.parametrize("how,code", [("alt:raise", 3), ("alt:hang", 4)])
def test_a_failing_secondary_block_still_yields_the_line(luw, how, code, tmp_path):
    """first-contact insurance: the x-whole cut (a secondary block of the N > 1 line) raises on every rank, or never returns -- rank 0 still prints ONE
    parseable line with n_gpus, the headline of the literal cut and its per-rank blocks, an `error` where the block would be, and the job ends non-zero
    (no re-exec; the launcher takes the other ranks down)"""
    port = 29500 + ((os.getpid() + 411 + code) % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--share-device", "0", "--size", "384", "64", "64", "--no-parity",
           "--no-group-host"]
    path = str(tmp_path / "full.json")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, LUW_BENCH_INJECT=how, LUW_BENCH_BLOCK_TIMEOUT="8",
        LUW_BENCH_FULL_JSON=path))
    assert r.returncode != 0
    line, out = full_record(r.stdout, path)
    assert line["n_gpus"] == 2 and line["value"] > 0 and "error" in line["secondary"]["x_whole_n_gpu"]
    assert out["n_gpus"] == 2 and out["value"] > 0 and len(out["per_rank"]) == 2 and out["config"]["n_gpu"] == [2, 1, 1]
    assert "error" in out["secondary"]["x_whole_n_gpu"]
