"""GPU fuzz of the halo exchange of the one-process-per-GPU host: random local shapes (odd extents included), cuts, DDF formats, thermal lattice on / off,
solids anywhere (border columns, corner lines), random step counts -- everything in one batch with edge messages and the x faces left in their buffers (the
default) against the reference's three phases with unpack kernels (LUW_EXCHANGE=sequential) on one rank that is its own neighbour: rho, u, (T,) and the DDFs
bit for bit.   usage (GPU box): python3 tests/fuzz/fuzz_exchange_gpu.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    import latticeurbanwind_amd as luw
    from latticeurbanwind_amd import capi
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, PeerLoopbackTransport
    from helpers import synthetic_state, thermal_state, TYPE_S
    luw.load()
    bad = 0
    for case in range(cases):
        fp16c = bool(rng.integers(2)); thermal = bool(rng.integers(2))
        D = [(2, 1, 1), (2, 2, 1), (2, 1, 2), (2, 2, 2), (1, 2, 2), (1, 2, 1)][rng.integers(6)]
        own = (int(rng.choice([20, 37, 130, 257, 320]) if not fp16c else rng.choice([40, 130, 256, 257, 320])), int(rng.integers(5, 12)),
            int(rng.integers(5, 11)))
        gN = tuple(o * d for o, d in zip(own, D))
        steps = int(rng.integers(1, 9))
        os.environ["LUW_X_SHELL"] = str(int(rng.choice([16, 64, 128])))
        seed = int(rng.integers(1 << 30))
        jitter = int(rng.choice([0, 0, 100, 400]))            # schedule fuzzing (luw_dev_schedule_jitter): random delays in front of the library's kernels
        res = {}
        for exchange in ("batch", "sequential"):
            os.environ["LUW_EXCHANGE"] = exchange
            capi.schedule_jitter(seed + (exchange == "sequential"), jitter)
            lay = DomainLayout(gN, D, 0)
            if not lay.can_overlap(): break
            sim = DomainDecomposedLBM(gN, D, 0.01, rank=0, transport=PeerLoopbackTransport(lay), fp16c=fp16c, device=0,
                **(dict(alpha=0.004) if thermal else {}))
            lx, ly, lz = lay.lN
            st = synthetic_state(lx, ly, lz, seed=seed, shell=None)
            flags = st[0].reshape(lz, ly, lx).copy()
            r2 = np.random.default_rng(seed)
            flags[r2.random(flags.shape) < 0.04] = TYPE_S                        # solids anywhere, border columns and corner lines included
            if thermal:
                tflags, T = thermal_state(flags.ravel(), (lx, ly, lz))
                sim.set_fields(tflags, st[1], st[2], T)
            else:
                sim.set_fields(flags.ravel(), st[1], st[2])
            sim.run(steps)
            u, rho = sim.fields()
            out = [u.copy(), rho.copy(), np.asarray(sim.backend.lbm.download_fi()).copy()]
            if thermal: out += [sim.backend.download_T().copy(), np.asarray(sim.backend.lbm.download_gi()).copy()]
            res[exchange] = out
            sim.backend.close()
            capi.schedule_jitter(0, 0)
        if len(res) < 2: continue
        same = all(np.array_equal(a, b) for a, b in zip(res["batch"], res["sequential"]))
        bad += not same
        print("case %d: %s local %s n_gpu %s thermal %s x_shell %s jitter %d steps %d: %s" % (case, "fp16c" if fp16c else "f32", own, D, thermal,
            os.environ["LUW_X_SHELL"], jitter, steps,
            "equal" if same else "DIFFERENT"), flush=True)
    print("fuzz: %d cases, %d different" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
