"""`python3 bench.py --gpus N` -- the form the driver's SCALE runs use -- starts its own ranks (benchmarks/launch.py): one command drives all domains, as
`FluidX3D <deck>` does in the reference (FX/lbm.cpp:1057-1112).  CPU side: when the branch is taken, what it starts, and that it is taken before torch or
the HIP library are imported; the run itself on a GPU is tests/test_gpu_bench_distributed.py::test_plain_command_starts_its_own_ranks."""
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks import launch   # noqa: E402


def test_branch_is_taken_only_for_the_bare_multi_gpu_command():
    assert launch.needs_launcher(8, {})
    assert launch.needs_launcher(2, {"LOCAL_RANK": "0"})
    assert not launch.needs_launcher(1, {})                                         # the N = 1 line never launches
    assert not launch.needs_launcher(8, {"WORLD_SIZE": "8", "RANK": "3"})           # under torch.distributed.run: a rank
    assert not launch.needs_launcher(8, {"WORLD_SIZE": "1"})                        # a wrong launcher is an error of the caller, not a reason to launch again
    assert not launch.needs_launcher(8, {launch.LAUNCHED_MARK: "1"})                # a rank of a self-started run never launches


def test_child_command_line():
    cmd = launch.launcher_argv("bench.py", ["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, 29611, python="py")
    assert cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29611"]
    assert cmd[10] == os.path.abspath("bench.py") and cmd[11:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert 1024 < launch.free_port() < 65536


def test_self_launch_passes_the_exit_code_and_marks_the_ranks():
    seen = {}

    def run(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=seen["rc"])
    for rc, want in ((0, 0), (1, 1), (-9, 137)):
        seen["rc"] = rc
        assert launch.self_launch("bench.py", ["--gpus", "4"], 4, environ={"LUW_BENCH_MASTER_PORT": "29777", "PATH": "x"}, run=run,
            loaded=lambda: False) == want
    assert seen["cmd"][8:10] == ["--master-port", "29777"] and seen["cmd"][4:6] == ["--nproc-per-node", "4"]
    assert seen["env"][launch.LAUNCHED_MARK] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["PATH"] == "x"
    # a process that has imported torch or the HIP binding must never take the branch (it may have initialised the GPU): refused, nothing started
    import pytest
    seen.clear()
    with pytest.raises(RuntimeError):
        launch.self_launch("bench.py", ["--gpus", "4"], 4, environ={}, run=run, loaded=lambda: True)
    assert not seen


def test_branch_is_reached_before_any_gpu_runtime_is_imported():
    # bench.py as __main__ with the launcher replaced by a recorder: at the moment of the branch neither torch nor the HIP library's binding is loaded
    code = ("import sys, runpy; sys.path.insert(0, %r); import benchmarks.launch as L\n"
            "def rec(script, argv, gpus):\n"
            "    print('LAUNCH', gpus, argv, 'torch' in sys.modules, 'latticeurbanwind_amd.capi' in sys.modules, L.gpu_runtime_loaded()); return 7\n"
            "L.self_launch = rec; sys.argv = ['bench.py', '--gpus', '8', '--steps', '3']\n"
            "runpy.run_path(%r, run_name='__main__')\n") % (ROOT, os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", launch.LAUNCHED_MARK)}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 7, r.stderr[-2000:]
    assert r.stdout.strip() == "LAUNCH 8 ['--gpus', '8', '--steps', '3'] False False False"


def test_plain_command_really_starts_ranks_here():
    # no GPU in this container: the ranks come up under torch.distributed.run and say so (the launcher ends the second one when the first fails); the parent
    # relays the failure as ITS exit code and
    # prints no line.  (Before: "--gpus 2 but WORLD_SIZE=1" from the parent itself, no rank ever started.)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", launch.LAUNCHED_MARK)}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
        timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "without a launcher: starting -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1" in r.stderr
    assert r.stderr.count("no GPU visible; the hot path has no CPU fallback") >= 1 and "WORLD_SIZE=1" not in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
