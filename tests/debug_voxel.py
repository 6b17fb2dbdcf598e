"""dump the device voxeliser's masks for the golden cases (debug aid; run on the GPU box, outputs under gpurun_out/)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import setup_profile as sp
from latticeurbanwind_amd.lbm import LBM

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for name in sys.argv[2:]:
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refcases", name, "conf.luwpf")
    su = sp.setup_profile_case(deck, solid_mask=np.zeros((1, 1, 1), bool) if False else None)
    Nx, Ny, Nz = su["Nx"], su["Ny"], su["Nz"]
    lbm = LBM(Nx, Ny, Nz, nu=su["nu"])
    lbm.voxelize_mesh_on_device(su["tri_lattice"])
    got = (lbm.flags.data.reshape(Nz, Ny, Nx) & 1).astype(np.uint8)
    py = (su["flags"].reshape(Nz, Ny, Nx) & 1).astype(np.uint8)
    np.savez_compressed(os.path.join(out, "vox_%s.npz" % name), hip=got, py_host=py)
    print(name, "hip solid", int(got.sum()), "python(host IEEE) solid", int(py.sum()))
