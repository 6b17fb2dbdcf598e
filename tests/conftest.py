import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def usable_cores():
    """cores this process may really use: the affinity mask capped by the cgroup's CPU quota (os.cpu_count() reports the whole host inside a container)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


# The CPU oracle is OpenMP code, and libgomp sizes its team from the HOST's core count: on a GPU box whose container is allowed 16 of the 256 cores it sees
# (cgroup cpu.max) every parallel
# region of the oracle would start 256 threads that spin on 16 cores' worth of time (a 48x20x12 lattice then takes 125 ms per step; GPU suite of round 3: 13.5
# minutes,
# most of it this).  Set before the library is loaded: as many threads as cores we may use, and waiting threads that sleep instead of spinning.
os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def luw():
    """the product package with its HIP library built (hipcc cross-compiles on CPU boxes)"""
    import latticeurbanwind_amd as pkg
    pkg.build()
    pkg.load()
    return pkg


def pytest_runtest_setup(item):
    """LUW_TEST_MEMLOG=<file>: one line per test with the device's free memory, this process's resident set and the host's available memory -- the trail to read
    when a long GPU session dies in an allocation (tools/gpu_suite_memlog.sh)"""
    path = os.environ.get("LUW_TEST_MEMLOG")
    if not path:
        return
    free = total = -1
    try:
        import torch
        if torch.cuda.is_available():
            free, total = torch.cuda.mem_get_info(0)
    except Exception:                                        # noqa: BLE001 -- a diagnostic must never fail a test
        pass
    rss = avail = -1
    try:
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS:"): rss = int(line.split()[1]) // 1024
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"): avail = int(line.split()[1]) // 1024
        threads = len(os.listdir("/proc/self/task"))
    except OSError:
        threads = -1
    with open(path, "a") as f:
        f.write("%s device_free_MiB %d of %d rss_MiB %d host_available_MiB %d threads %d\n" % (item.nodeid, free >> 20, total >> 20, rss, avail, threads))
