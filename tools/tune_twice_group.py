import sys, os
sys.path.insert(0, os.getcwd())
import latticeurbanwind_amd as luw
luw.load()
D = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 2, 2)
for k in range(3):
    g = luw.LBMGroup(512, 512, 512, *D, 1e-4, devices=[0] * (D[0] * D[1] * D[2]), global_arrays=False)
    print("cycle", k, "created", flush=True)
    g.close()
    print("cycle", k, "closed", flush=True)
