import sys, os
sys.path.insert(0, os.getcwd())
import latticeurbanwind_amd as luw
luw.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for k in range(n):
    g = luw.LBM(512, 512, 512, 1e-4)
    g.run(0); g.run(2)
    g.close()
    print("cycle", k, "ok", flush=True)
