"""Decks with n_gpu > 1 through the C++ driver itself: `luw_driver <deck>` builds all Dx*Dy*Dz domains in ONE process (the
reference's LBM object, FX/lbm.cpp:1057-1112), here all on the test box's single GPU (--devices 0,0,..).  Every file it writes
must be byte-identical to what the same deck with n_gpu = [1,1,1] writes: per-domain voxelisation, inlet points and probe cells
split over their owners, on-device statistics gathered from the domains, thermal lattice with its own halo swap."""
import filecmp
import glob
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from vtkio import read_vtk

pytestmark = pytest.mark.gpu
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
    return {os.path.relpath(p, proj): p for p in glob.glob(os.path.join(proj, "RESULTS", "**", "*.*"), recursive=True)}


# z splits: only cases whose geometry does not put a face exactly on a lattice plane below the cut (see the next test)
@pytest.mark.parametrize("case,n_gpu,ddf",
    [("CaseA", (2, 1, 1), "fp32"), ("CaseV", (1, 2, 1), "fp32"), ("CaseV", (2, 2, 1), "fp32"), ("CaseN1", (1, 2, 1), "fp32"), ("CaseP", (2, 2, 1), "fp32"),
                                            ("CaseT1", (2, 2, 1), "fp32"), ("CaseT3", (2, 1, 1), "fp16c"), ("CaseG", (2, 2, 2), "fp16c"),
                                                ("CaseL", (1, 1, 2), "fp32")])
def test_driver_with_n_gpu_writes_the_single_domain_files(luw, tmp_path, case, n_gpu, ddf):
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    ref_proj, ref_deck = _case(tmp_path, case, (1, 1, 1), "_one")
    r = subprocess.run([DRIVER, ref_deck, "--ddf", ddf], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    proj, deck = _case(tmp_path, case, n_gpu, "_dom")
    D = n_gpu[0] * n_gpu[1] * n_gpu[2]
    r = subprocess.run([DRIVER, deck, "--ddf", ddf, "--devices", ",".join(["0"] * D)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "domains" in r.stdout
    want, got = _files(ref_proj), _files(proj)
    assert sorted(want) == sorted(got) and len(want) >= 3
    for name in sorted(want):
        if filecmp.cmp(want[name], got[name], shallow=False):
            continue
        if name.endswith(".vtk"):                       # say which array differs
            hw, fw = read_vtk(want[name]); hg, fg = read_vtk(got[name])
            assert hw == hg, name
            for key in fw:
                assert np.array_equal(fg[key], fw[key]), (name, key, int((fg[key] != fw[key]).sum()))
        assert False, name + " differs"


@pytest.mark.parametrize("case,n_gpu,ddf",
    [("CaseA", (1, 1, 1), "fp32"), ("CaseA", (2, 2, 1), "fp32"), ("CaseT1", (2, 1, 1), "fp32"), ("CaseG", (2, 2, 2), "fp16c"), ("CaseP", (1, 2, 1), "fp32")])
def test_device_vtk_export_equals_the_host_conversion(luw, tmp_path, case, n_gpu, ddf):
    """the output path: every VTK the driver writes with the devices producing the payload (per-domain kernel: SoA -> AoS, SI units, big-endian;
    tke / TI / TLS from the statistics on the devices, the TLS stencil across domain cuts; slabs through pinned memory, pwrite from a writer
    thread) against the files of the host path (full download, gather, host conversion: LUW_HOST_VTK=1) -- byte for byte"""
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    D = n_gpu[0] * n_gpu[1] * n_gpu[2]
    dev = ["--devices", ",".join(["0"] * D)] if D > 1 else []
    out = {}
    for tag, env in (("_host", dict(os.environ, LUW_HOST_VTK="1")), ("_dev", dict(os.environ, LUW_HOST_VTK="0"))):
        proj, deck = _case(tmp_path, case, n_gpu, tag)
        r = subprocess.run([DRIVER, deck, "--ddf", ddf] + dev, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        out[tag] = _files(proj)
    want, got = out["_host"], out["_dev"]
    assert sorted(want) == sorted(got) and len(want) >= 3
    for name in sorted(want):
        if filecmp.cmp(want[name], got[name], shallow=False):
            continue
        if name.endswith(".vtk"):
            hw, fw = read_vtk(want[name]); hg, fg = read_vtk(got[name])
            assert hw == hg, name
            for key in fw:
                assert np.array_equal(fg[key].view(np.uint32), fw[key].view(np.uint32)), (name, key,
                    int((fg[key].view(np.uint32) != fw[key].view(np.uint32)).sum()))
        assert False, name + " differs"


def test_z_split_voxelisation_follows_the_reference_ray_origin(luw, tmp_path):
    """A domain voxelises its box with rays that start at ITS lowest layer -- for the lower domain of a z split that is the halo
    layer below z = 0 -- exactly like the reference kernel (r_origin = position(xyz) + offset with xyz.z = clamp((int)z0 - Oz, ..),
    FX/kernel.cpp:2392-2394).  Hit distances are truncated to whole cells, so a face lying exactly on a lattice plane can land one
    layer off compared with the undivided lattice: the masks agree everywhere but in one z layer of the
    building's footprint.  Nothing else may differ."""
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    masks = []
    for n_gpu in ((1, 1, 1), (1, 1, 2)):
        proj, deck = _case(tmp_path, "CaseB", n_gpu, "_z%d" % n_gpu[2])
        r = subprocess.run([DRIVER, deck, "--ddf", "fp32"] + (["--devices", "0,0"] if n_gpu[2] > 1 else []), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
        h, f = read_vtk(glob.glob(os.path.join(proj, "RESULTS", "vtk", "*_avg-*.vtk"))[0])
        masks.append(f["fluid"][..., 0] == 0.0)                     # (z, y, x) solid mask
    diff = masks[0] != masks[1]
    zs = np.unique(np.nonzero(diff)[0])
    assert len(zs) <= 1 and diff.sum() <= masks[0][1].sum()          # at most one layer, at most the building's footprint


def test_driver_refuses_more_domains_than_devices(luw, tmp_path):
    """one HIP device per domain unless --devices says otherwise (smart_device_selection errors alike, FX/lbm.cpp:961-979)"""
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    proj, deck = _case(tmp_path, "CaseB", (2, 1, 1), "_err")
    r = subprocess.run([DRIVER, deck, "--ddf", "fp32"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "fewer HIP devices than domains" in (r.stdout + r.stderr)
