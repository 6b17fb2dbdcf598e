"""Device voxeliser (luw_voxelize_mesh <-> LBM::voxelize_mesh_on_device, FX/lbm.cpp:1411, kernel FX/kernel.cpp:2381-2471)
against the TYPE_S masks the REAL reference produced on an MI355X (tests/golden/ref_*_Case*.npz, `solid`, recovered from the
`fluid` field of its _avg VTK).  Bar: bit-exact masks."""
import os

import numpy as np
import pytest

from oracle import setup_profile as sp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _case(name, fixture):
    ref = np.load(os.path.join(GOLD, fixture))
    deck = os.path.join(GOLD, "refcases", name, "conf.luwpf")
    su = sp.setup_profile_case(deck, solid_mask=ref["solid"].astype(bool))   # mask argument only skips the slow python voxeliser
    return su, ref


@pytest.mark.parametrize("name,fixture", [("CaseA", "ref_fp32_CaseA.npz"), ("CaseB", "ref_fp32_CaseB.npz"),
                                          ("CaseG", "ref_fp32_CaseG.npz"), ("CaseH", "ref_fp32_CaseH.npz"),
                                          ("CaseG", "ref_shipped_CaseG.npz")])
def test_voxelize_matches_reference_mask(luw, name, fixture):
    from latticeurbanwind_amd.lbm import LBM
    su, ref = _case(name, fixture)
    Nx, Ny, Nz = su["Nx"], su["Ny"], su["Nz"]
    lbm = LBM(Nx, Ny, Nz, nu=su["nu"])
    lbm.voxelize_mesh_on_device(su["tri_lattice"])
    got = (lbm.flags.data.reshape(Nz, Ny, Nx) & 1).astype(bool)
    # the reference's VTK shows the state after the BC fill (z = 0 plane and below-terrain rim cells become solid too):
    # compare the voxeliser output where the BC fill cannot have changed it, then the full mask after the same fill
    want = ref["solid"].astype(bool)
    core = np.zeros_like(got); core[1:want.shape[0], 1:-1, 1:-1] = True
    assert np.array_equal(got[:want.shape[0]][core[:want.shape[0]]], want[core[:want.shape[0]]]), \
        "voxeliser mask differs from the reference in %d interior cells" % int((got[:want.shape[0]] != want)[core[:want.shape[0]]].sum())
    assert int(got.sum()) == _console_solid(fixture)


def _console_solid(fixture):
    import re
    txt = open(os.path.join(GOLD, fixture.replace(".npz", ".console.txt"))).read()
    return int(re.search(r"solid = (\d+)", txt).group(1))


def test_voxelize_preserves_other_flags_and_clears_stale_solids(luw):
    """cells outside the mesh that were TYPE_S with zero velocity are cleared; TYPE_E bits and moving solids (u != 0) survive
    (FX/kernel.cpp:2447-2462)"""
    from latticeurbanwind_amd.lbm import LBM
    su, ref = _case("CaseB", "ref_fp32_CaseB.npz")
    Nx, Ny, Nz = su["Nx"], su["Ny"], su["Nz"]
    lbm = LBM(Nx, Ny, Nz, nu=su["nu"])
    f = lbm.flags.data.reshape(Nz, Ny, Nx); u = lbm.u.data.reshape(3, Nz, Ny, Nx)
    f[13, 5, 5] = 0x01                      # stale static solid in the air, inside the padded bounding box
    f[13, 6, 6] = 0x01; u[0, 13, 6, 6] = 0.01  # "moving" solid: kept
    f[13, 7, 7] = 0x02                      # TYPE_E in the air: kept
    f[1, 8, 8] = 0x02                       # TYPE_E inside the slab: becomes TYPE_S (TYPE_BO bits replaced)
    lbm.voxelize_mesh_on_device(su["tri_lattice"])
    f = lbm.flags.data.reshape(Nz, Ny, Nx)
    assert f[13, 5, 5] == 0 and f[13, 6, 6] == 0x01 and f[13, 7, 7] == 0x02 and f[1, 8, 8] == 0x01


def test_triangle_bins_do_not_change_the_mask(luw):
    """the kernel visits only the triangles binned to a 16x16 column tile (bounding box grown by one cell, triangle order
    kept); with LUW_TEST_AIDS=voxelize_all every tile sees every triangle, as the reference kernel does: same cells, on a
    mesh of 300 rotated boxes / pyramids / tetrahedra spread over a 400x320x64 lattice"""
    import sys, time
    sys.path.insert(0, GOLD)
    import make_refcases as mr
    from latticeurbanwind_amd.lbm import LBM
    rng = np.random.default_rng(5)
    tris = mr.box_tris(1.0, 399.0, 1.0, 319.0, 1.0, 3.0)
    for k in range(300):
        cx, cy = rng.uniform(20, 380), rng.uniform(20, 300)
        kind = k % 3
        if kind == 0:
            tris += mr.rot_box_tris(cx, cy, rng.uniform(4, 30), rng.uniform(4, 30), 3.0, 3.0 + rng.uniform(2, 50), rng.uniform(0, 90))
        elif kind == 1:
            w = rng.uniform(5, 20)
            tris += mr.hull_tris([(cx - w, cy - w, 3.0), (cx + w, cy - w * 0.8, 3.0), (cx + w * 0.9, cy + w, 3.0), (cx - w, cy + w * 0.7, 3.0)],
                [(cx, cy, 3.0 + rng.uniform(5, 40))])
        else:
            z = rng.uniform(10, 40)
            tris += mr.hull_tris([(cx - 6, cy - 5, z), (cx + 7, cy - 4, z + 1.5), (cx, cy + 8, z + 0.7)], [(cx + 0.5, cy, z + rng.uniform(4, 15))])
    tri = np.array(tris, np.float32)
    masks, secs = [], []
    for brute in (False, True):
        from latticeurbanwind_amd import capi
        if brute: os.environ["LUW_TEST_AIDS"] = "voxelize_all"
        else: os.environ.pop("LUW_TEST_AIDS", None)
        capi.reload_tuning()                                   # the library reads its environment once
        try:
            lbm = LBM(400, 320, 64, nu=0.01)
            t0 = time.perf_counter(); lbm.voxelize_mesh_on_device(tri); secs.append(time.perf_counter() - t0)
            masks.append((lbm.flags.data & 1).copy()); lbm.close()
        finally:
            os.environ.pop("LUW_TEST_AIDS", None)
            capi.reload_tuning()
    assert masks[0].sum() > 100000 and np.array_equal(masks[0], masks[1])
    print("voxelise %d triangles on 400x320x64: binned %.3f s, all triangles per column %.3f s" % (len(tris), secs[0], secs[1]))
