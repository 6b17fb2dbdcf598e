"""Runs one golden case through the single-GPU C++ driver and through latticeurbanwind_amd.run_deck on n_gpu = Dx Dy Dz ranks
(all on GPU 0, gloo + host staging) and lists which output arrays differ.  usage: compare_run_deck.py CaseX Dx Dy Dz"""
import glob, os, re, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from vtkio import read_vtk
case, D = sys.argv[1], tuple(int(v) for v in sys.argv[2:5])
tmp = tempfile.mkdtemp()
def mk(tag, n):
    proj = os.path.join(tmp, case + tag); shutil.copytree(os.path.join(ROOT, "tests/golden/refcases", case), proj)
    deck = glob.glob(os.path.join(proj, "conf.luw*"))[0]
    txt = re.sub(r"n_gpu = \[[^\]]*\]", "n_gpu = [%d, %d, %d]" % n, open(deck).read())
    open(deck, "w").write(txt)
    return proj, deck
rp, rd = mk("_ref", (1, 1, 1)); subprocess.run([os.path.join(ROOT, "latticeurbanwind_amd/host/luw_driver"), rd, "--ddf", "fp32"], capture_output=True)
p, d = mk("_run", D); w = D[0] * D[1] * D[2]
env = dict(os.environ, PYTHONPATH=ROOT)
r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(w), "--master-addr", "127.0.0.1", "--master-port", "29544", "-m", "latticeurbanwind_amd.run_deck", d, "--ddf", "fp32", "--share-device", "0"], capture_output=True, text=True, env=env, cwd=ROOT)
print("rc", r.returncode, r.stdout[-1500:]); print("\n".join(l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Traceback" in l or "File" in l or "run_deck" in l)[:3000])
for f in sorted(glob.glob(os.path.join(rp, "RESULTS/vtk/*.vtk"))):
    g = os.path.join(p, "RESULTS/vtk", os.path.basename(f))
    if not os.path.exists(g): print("missing", g); continue
    hw, fw = read_vtk(f); hg, fg = read_vtk(g)
    for k in fw:
        neq = (fw[k] != fg[k])
        if neq.any():
            idx = np.argwhere(neq.any(-1) if neq.ndim == 4 else neq)
            print(os.path.basename(f), k, "differ:", int(neq.sum()), "max abs", float(np.abs(fw[k] - fg[k]).max()), "z range", idx[:, 0].min(), idx[:, 0].max(), "y", idx[:, 1].min(), idx[:, 1].max(), "x", idx[:, 2].min(), idx[:, 2].max())
        else:
            print(os.path.basename(f), k, "equal")
shutil.rmtree(tmp)
