"""bench.py --gpus N without a launcher: the process starts its own ranks.

The reference drives all its domains from ONE command (`FluidX3D <deck>`, FX/lbm.cpp:1057-1112), and so does the driver of this repo's SCALE runs:
`python3 bench.py --gpus N ...`.  One process per GPU needs a rendezvous, so a bench.py that finds no WORLD_SIZE in its environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process, lets the child's stdout / stderr through untouched
(rank 0's ONE JSON line is the only thing on stdout) and exits with the child's code.  This module imports neither torch nor the HIP library: the branch
is decided and taken before anything of this process can have initialised a GPU (a process that has must never be replaced by another program, and this
one is not: it waits for its child)."""
import os
import socket
import subprocess
import sys

LAUNCHED_MARK = "LUW_BENCH_SELF_LAUNCHED"      # set for the ranks of a self-started run: a rank never launches again, whatever its environment lacks


def needs_launcher(gpus, environ):
    """True when this process is the bare `bench.py --gpus N` (N > 1) command and has to start its ranks itself: no launcher's WORLD_SIZE / RANK in the
    environment and not itself a rank of a self-started run"""
    return gpus > 1 and "WORLD_SIZE" not in environ and LAUNCHED_MARK not in environ


def gpu_runtime_loaded():
    """has this process imported anything that could have initialised the GPU?  (the launcher branch refuses to run then)"""
    return any(m in sys.modules for m in ("torch", "latticeurbanwind_amd.capi"))


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_argv(script, argv, gpus, port, python=None):
    """the command line of the child: torch.distributed.run on one node, rendezvous on 127.0.0.1 (the container's host name may not resolve)"""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
        "--master-port", str(port), os.path.abspath(script), *argv]


def launcher_env(environ, gpus=1):
    env = dict(environ)
    env[LAUNCHED_MARK] = "1"
    if "OMP_NUM_THREADS" not in env:                           # (torch.distributed.run would pin it to 1: rank 0's self-check runs the OpenMP oracle)
        try:
            env["OMP_NUM_THREADS"] = str(max(1, len(os.sched_getaffinity(0)) // max(1, gpus)))
        except (AttributeError, OSError):
            pass
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    return env


def self_launch(script, argv, gpus, environ=None, run=subprocess.run, loaded=gpu_runtime_loaded):
    """start the ranks, wait, return the exit code for this process (non-zero when any rank failed)"""
    if loaded():
        raise RuntimeError("bench.py: the launcher branch was reached after torch / the HIP library were imported")
    environ = os.environ if environ is None else environ
    cmd = launcher_argv(script, argv, gpus, int(environ.get("LUW_BENCH_MASTER_PORT", 0)) or free_port())
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (gpus, " ".join(cmd[1:10]))); sys.stderr.flush()
    sys.stdout.flush()
    r = run(cmd, env=launcher_env(environ, gpus))                    # stdout / stderr inherited: the ranks' ONE line reaches this process's stdout as it is
    return r.returncode if r.returncode >= 0 else 128 - r.returncode
