"""Decomposed deck runs (latticeurbanwind_amd/run_deck.py): the launcher must write the files the single-GPU C++ driver
writes for the same deck -- with one rank, and with the deck's n_gpu = [2,1,1] / [1,2,2] on several ranks (here all ranks
share the one GPU of the test box and swap halos through gloo + host staging; on a node the same code runs over RCCL)."""
import glob
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

from vtkio import read_vtk

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")


def _case(tmp_path, case, n_gpu, tag):
    proj = str(tmp_path / (case + tag))
    shutil.copytree(os.path.join(GOLD, "refcases", case), proj)
    deck = glob.glob(os.path.join(proj, "conf.luw*"))[0]
    txt = re.sub(r"n_gpu = \[[^\]]*\]", "n_gpu = [%d, %d, %d]" % n_gpu, open(deck).read())
    open(deck, "w").write(txt)
    return proj, deck


def _files(proj):
    return {os.path.basename(p): p for p in glob.glob(os.path.join(proj, "RESULTS", "vtk", "*.vtk"))}


def test_runtime_decomposition_keeps_x_whole():
    from latticeurbanwind_amd.run_deck import runtime_decomposition as rd
    assert rd((751, 742, 174), (2, 1, 1)) == (1, 2, 1)                     # the reference's example deck
    assert rd((2048, 1024, 512), (4, 2, 1)) == (1, 4, 2)
    assert rd((48, 40, 24), (2, 2, 1)) == (1, 4, 1) and rd((48, 41, 23), (2, 1, 1)) == (2, 1, 1)   # least halo area; nothing divides: the deck's grid stays
    assert rd((100, 7, 64), (2, 1, 1)) == (1, 1, 2)


@gpu
@pytest.mark.parametrize("case,n_gpu", [("CaseA", (1, 1, 1)), ("CaseV", (1, 1, 1)), ("CaseA", (2, 1, 1)), ("CaseV", (1, 2, 2)), ("CaseN1", (1, 2, 1)), ("CaseP", (2, 2, 1)), ("CaseT1", (1, 1, 1)), ("CaseT1", (2, 2, 1)), ("CaseT3", (1, 2, 2))])
def test_run_deck_writes_the_drivers_files(luw, tmp_path, case, n_gpu):
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    ref_proj, ref_deck = _case(tmp_path, case, (1, 1, 1), "_ref")
    r = subprocess.run([DRIVER, ref_deck, "--ddf", "fp32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    proj, deck = _case(tmp_path, case, n_gpu, "_run")
    world = n_gpu[0] * n_gpu[1] * n_gpu[2]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if world == 1:
        cmd = [sys.executable, "-m", "latticeurbanwind_amd.run_deck", deck, "--ddf", "fp32"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", "29533",
               "-m", "latticeurbanwind_amd.run_deck", deck, "--ddf", "fp32", "--share-device", "0"] + (["--literal-n-gpu"] if case in ("CaseA", "CaseP") else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    want, got = _files(ref_proj), _files(proj)
    assert sorted(want) == sorted(got) and len(want) >= 3
    csv_w = sorted(glob.glob(os.path.join(ref_proj, "RESULTS", "*.csv"))); csv_g = sorted(glob.glob(os.path.join(proj, "RESULTS", "*.csv")))
    assert [os.path.basename(q) for q in csv_w] == [os.path.basename(q) for q in csv_g] and (case != "CaseP" or len(csv_w) == 5)
    for a, b in zip(csv_w, csv_g):                                  # probe CSVs: identical text
        assert open(a).read() == open(b).read(), os.path.basename(a)
    for name in sorted(want):
        hw, fw = read_vtk(want[name]); hg, fg = read_vtk(got[name])
        assert hw == hg, name
        for key in fw:
            if key in ("TI", "TLS"):     # numpy vs C++ evaluation of sqrt / division chains
                assert np.allclose(fg[key], fw[key], rtol=2e-6, atol=1e-12), (name, key)
            else:
                assert np.array_equal(fg[key], fw[key]), (name, key, int((fg[key] != fw[key]).sum()))
