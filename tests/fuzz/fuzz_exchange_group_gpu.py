"""GPU fuzz of the halo exchange of the one-process host (luw_group_*, what luw_driver runs for decks with n_gpu): random lattices (odd extents included),
cuts up to eight domains on the box's one GPU, DDF formats, thermal lattice on / off, solids anywhere (border columns, corner lines), nudging / sponge /
Coriolis on / off, random step counts in two run calls, one host thread or one per domain -- ONE pack / unpack round per step with edge messages and the
x faces read in place (the default) against the reference's three phases (LUW_GROUP_EXCHANGE=sequential) and, on small lattices, against the CPU oracle on
the undivided lattice: rho, u, (T) bit for bit.   usage (GPU box): python3 tests/fuzz/fuzz_exchange_group_gpu.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    import latticeurbanwind_amd as luw
    from latticeurbanwind_amd import capi
    from helpers import synthetic_state, thermal_state, TYPE_S
    from oracle import oracle
    luw.load()
    bad = 0
    for case in range(cases):
        fp16c = bool(rng.integers(2)); thermal = bool(rng.integers(4) == 0); forces = bool(rng.integers(2)); threads = bool(rng.integers(4) == 0)
        D = [(2, 1, 1), (2, 2, 1), (2, 1, 2), (2, 2, 2), (1, 2, 2), (1, 2, 1), (4, 2, 1), (1, 4, 2), (3, 1, 2)][rng.integers(9)]
        own = (int(rng.choice([6, 20, 37, 130, 257]) if not fp16c else rng.choice([6, 40, 130, 256, 257])), int(rng.integers(4, 10)), int(rng.integers(4, 9)))
        gN = tuple(o * d for o, d in zip(own, D))
        n = D[0] * D[1] * D[2]
        steps = (int(rng.integers(1, 7)), int(rng.integers(0, 6)))
        os.environ["LUW_X_SHELL"] = str(int(rng.choice([16, 64, 128])))
        os.environ["LUW_GROUP_THREADS"] = "1" if threads else "0"
        # (the draw round 5's cases spent on a removed option now picks the transport of half of them: the one-round exchange through send buffers + copies /
        # ONE grouped ncclSend / ncclRecv batch, round 6; RCCL's group calls are issued by one thread)
        transport = "peer" if rng.integers(2) == 0 else ("staged" if case % 2 == 0 else "rccl")
        if transport == "rccl": threads = False; os.environ["LUW_GROUP_THREADS"] = "0"
        os.environ["LUW_GROUP_TRANSPORT"] = transport
        seed = int(rng.integers(1 << 30))
        jitter = int(rng.choice([0, 0, 100, 400]))                                        # schedule fuzzing: random delays in front of the kernels (us at most)
        st = synthetic_state(*gN, seed=seed, shell="luw" if forces else None)
        flags = st[0].copy()
        r2 = np.random.default_rng(seed)
        flags[(r2.random(flags.shape) < 0.04) & ((flags & 3) == 0)] = TYPE_S        # solids anywhere, border columns and corner lines included
        T = None
        if thermal:
            flags, T = thermal_state(flags, gN)
            # boundary cells carry their temperature as a preset, as in every LUW deck (FX/setup.cpp:5945-5985): the top sponge then relaxes T towards an
            # INPUT.  (A top-layer cell that computes its T is read and rewritten by the same launch -- order-dependent in the reference kernel itself,
            # and the one place where the GPU kernels and the sequential oracle may differ)
            flags = flags.copy(); flags[(flags & 3) == 2] |= 4
        kw = dict(alpha=0.004) if thermal else {}
        nud = dict(n_cells=3, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1); spg = dict(n_cells=2, inv_tau=0.02)
        if forces: kw.update(buffer_nudging=nud, top_sponge=spg)
        res = {}
        for exchange in ("one_phase", "sequential"):
            if exchange == "sequential": os.environ["LUW_GROUP_EXCHANGE"] = "sequential"
            else: os.environ.pop("LUW_GROUP_EXCHANGE", None)
            capi.reload_tuning()
            capi.schedule_jitter(seed + (exchange == "sequential"), jitter)
            g = luw.LBMGroup(*gN, *D, 0.01, fp16c=fp16c, devices=[0] * n, **kw)
            assert g.one_phase() == (exchange == "one_phase")
            g.flags[:] = flags; g.u[:] = st[1]; g.rho[:] = st[2]
            if thermal: g.T[:] = T
            if forces: g.set_coriolis(0.0, 3e-5, 4e-5)
            g.run(0); g.run(steps[0]); g.run(steps[1])
            g.read_from_device(("u", "rho", "T") if thermal else ("u", "rho"))
            res[exchange] = [g.u.copy(), g.rho.copy()] + ([g.T.copy()] if thermal else [])
            g.close()
            capi.schedule_jitter(0, 0)
        same = all(np.array_equal(a, b) for a, b in zip(res["one_phase"], res["sequential"]))
        vs_oracle = ""
        if gN[0] * gN[1] * gN[2] <= 200000:
            o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c, **({"alpha": 0.004} if thermal else {}))
            o.flags[:] = flags; o.u[:] = st[1]; o.rho[:] = st[2]
            if thermal: o.T[:] = T
            if forces:
                o.set_coriolis(0.0, 3e-5, 4e-5); o.set_buffer_nudging(nud["n_cells"], nud["inv_tau"], nud["downstream_face"], nud["nudge_vertical"])
                o.set_top_sponge(spg["n_cells"], spg["inv_tau"])
            o.run(steps[0] + steps[1])
            ok = np.array_equal(res["one_phase"][0], o.u) and np.array_equal(res["one_phase"][1], o.rho) and (not thermal
                or np.array_equal(res["one_phase"][2], o.T))
            same = same and ok
            vs_oracle = ", oracle %s" % ("equal" if ok else "DIFFERENT")
        bad += not same
        print("case %d: %s global %s n_gpu %s thermal %s forces %s threads %s transport %s x_shell %s jitter %s steps %s: %s%s" % (case,
            "fp16c" if fp16c else "f32", gN, D, thermal, forces, threads, transport, os.environ["LUW_X_SHELL"], jitter, steps,
            "routes equal" if same else "DIFFERENT", vs_oracle), flush=True)
    print("fuzz: %d cases, %d different" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
