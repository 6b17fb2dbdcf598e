#!/usr/bin/env python3
"""First contact, stage 2b: pick the cross-device defaults of the one-process host by DATA.  Nothing of this repo has ever
crossed a device boundary, and three choices were made on one GPU where they cannot be judged:

  * x faces written by the step kernels as scattered 2-4-byte remote stores (default) -- or fetched by a coalesced pack kernel (LUW_GROUP_EXCHANGE=one_packed);
  * ONE exchange round per step with edge messages (default) -- or the reference's three phases (LUW_GROUP_EXCHANGE=sequential);
  * faces as peer stores (default) -- or through send buffers and copies / ONE grouped RCCL batch (LUW_GROUP_TRANSPORT=staged / rccl);
  * boundary shell first + exchange beside the interior (default) -- or the whole box as one launch, then the exchange (LUW_GROUP_OVERLAP=0);
  * one enqueueing host thread (default) -- or one per domain (LUW_GROUP_THREADS=1).

Every alternative is run in a FRESH child process (this script itself, --child), interleaved over --reps rounds, on the urban tile of BASELINE configs[3]
(FP32) and configs[4] (FP16C + Coriolis) cut as n_gpu over the given devices.  Each child reports ms per step and a digest of rho and u after the timed steps:
all alternatives MUST agree bit for bit (they are the same arithmetic), otherwise the script fails before it recommends anything.  Result:
<out>/defaults.json -- every measurement, the winner per question, and the RULE applied: the default is kept unless an alternative is more than 2 % faster in
every round ("fused x faces only if >= 0.98 x packed" in the verdict's words), because the default is the path every other test of the repo has run.

usage: first_contact_defaults.py [--devices 0,1,..] [--n-gpu Dx Dy Dz] [--size Nx Ny Nz per domain] [--steps K] [--reps R] [--out DIR] [--dry-run]
  --devices all on ONE device (e.g. 0,0,0,0,0,0,0,0): a REHEARSAL -- all variants must run and agree, the timings are labelled meaningless
  --dry-run: print the child command lines without running anything (CPU rehearsal of the control flow)"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# question -> (label -> environment); the first label of each question is the DEFAULT
QUESTIONS = {
    "x_faces": {"fused (step kernels store them)": {}, "packed (pack / insert kernels)": {"LUW_GROUP_EXCHANGE": "one_packed"}},
    "exchange": {"one round": {}, "three phases": {"LUW_GROUP_EXCHANGE": "sequential"}},
    "transport": {"peer stores": {}, "staged copies": {"LUW_GROUP_TRANSPORT": "staged"}, "rccl batch": {"LUW_GROUP_TRANSPORT": "rccl"}},
    "schedule": {"shell first, exchange beside the interior": {}, "whole box, then the exchange": {"LUW_GROUP_OVERLAP": "0"}},
    "host_threads": {"one enqueueing thread": {}, "one thread per domain": {"LUW_GROUP_THREADS": "1"}},
}
KEEP_DEFAULT_UNLESS_FASTER_BY = 0.02


def variants():
    """(question, label, env) for every alternative; the default appears once per question so that it is timed beside its alternatives in every round"""
    out = []
    for q, alts in QUESTIONS.items():
        for label, env in alts.items():
            out.append((q, label, env))
    return out


def child(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import latticeurbanwind_amd as luw
    from bench import fill_channel, NU, tile_forcing, coriolis_omega
    luw.load()
    D = tuple(args.n_gpu); n = D[0] * D[1] * D[2]
    gN = tuple(s * d for s, d in zip(args.size, D))
    nud, spg = tile_forcing()
    g = luw.LBMGroup(*gN, *D, NU, fp16c=args.fp16c, devices=args.devices[:n], buffer_nudging=nud, top_sponge=spg, native_arith=False)
    fill_channel(g.flags, g.u, g.rho, *gN, buildings=True)
    if args.fp16c: g.set_coriolis(*coriolis_omega())
    g.run(0); g.run(5)
    t0 = time.perf_counter(); g.run(args.steps); dt = (time.perf_counter() - t0) / args.steps
    g.read_from_device()
    dig = hashlib.blake2b(np.ascontiguousarray(g.u).tobytes() + np.ascontiguousarray(g.rho).tobytes(), digest_size=16).hexdigest()
    print(json.dumps({"ms_per_step": dt * 1e3, "digest": dig, "overlap": bool(g.overlaps()), "one_phase": bool(g.one_phase()), "transport": g.transport(),
        "direct_peer_stores": bool(g.direct_peer_stores())}), flush=True)
    g.close()


def decide(results, rehearsal):
    """winner per question under the rule; results: {(question, label): [ms per round]}"""
    verdict = {}
    for q, alts in QUESTIONS.items():
        labels = list(alts)
        default = labels[0]
        best = default
        for lab in labels[1:]:
            a, d = results.get((q, lab)), results.get((q, default))
            if not a or not d: continue
            # faster than the default by more than the margin in EVERY round, and faster than the best so far on average
            if all(x < y * (1.0 - KEEP_DEFAULT_UNLESS_FASTER_BY) for x, y in zip(a, d)) and sum(a) < sum(results[(q, best)]): best = lab
        verdict[q] = {"default": default, "recommended": default if rehearsal else best, "environment": {} if rehearsal else QUESTIONS[q][best],
            "mean_ms": {lab: (sum(results[(q, lab)]) / len(results[(q, lab)]) if results.get((q, lab)) else None) for lab in labels}}
    return verdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0,0,0,0,0,0,0,0")
    ap.add_argument("--n-gpu", type=int, nargs=3, default=[4, 2, 1])
    ap.add_argument("--size", type=int, nargs=3, default=None, help="owned cells per domain (default 512 512 512 across devices, 128 64 64 in a rehearsal)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "first_contact"))
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--fp16c", action="store_true")
    args = ap.parse_args()
    args.devices = [int(d) for d in args.devices.split(",")]
    n = args.n_gpu[0] * args.n_gpu[1] * args.n_gpu[2]
    if len(args.devices) < n: raise SystemExit("first_contact_defaults.py: %d devices for n_gpu %s" % (len(args.devices), args.n_gpu))
    rehearsal = len(set(args.devices[:n])) < n
    if args.size is None: args.size = [128, 64, 64] if rehearsal else [512, 512, 512]
    if args.steps is None: args.steps = 6 if rehearsal else 40
    if args.child:
        return child(args)
    os.makedirs(args.out, exist_ok=True)
    record = {"devices": args.devices[:n], "n_gpu": args.n_gpu, "owned_cells_per_domain": args.size, "steps": args.steps, "rounds": args.reps,
        "rehearsal": rehearsal,
            "rule": "the default of a question is kept unless an alternative is more than %d %% faster in EVERY round; all alternatives must "
            "give the same rho and u bit for bit" % int(KEEP_DEFAULT_UNLESS_FASTER_BY * 100), "workloads": {}}
    if rehearsal:
        record[
            "note"] = "REHEARSAL: all domains on one device -- every variant ran and agreed; the timings say nothing about a node and no recommendation is made"
    base = [sys.executable, os.path.abspath(__file__), "--child", "--devices", ",".join(str(d) for d in args.devices), "--n-gpu", *map(str, args.n_gpu),
        "--size",
        *map(str, args.size), "--steps", str(args.steps)]
    failed = False
    for wl, extra in (("configs[3] urban tile, FP32", []), ("configs[4] urban tile, FP16C + Coriolis", ["--fp16c"])):
        results, digests, runs = {}, set(), []
        for rnd in range(args.reps):
            seen = {}                                  # the all-defaults run is every question's first alternative: measured once per round
            for q, label, env in variants():
                cmd = base + extra
                key = tuple(sorted(env.items()))
                if args.dry_run:
                    if key not in seen: print("[dry run] %s %s" % (" ".join("%s=%s" % kv for kv in env.items()) or "(defaults)", " ".join(cmd[1:])))
                    seen[key] = None
                    continue
                if key not in seen:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **env), cwd=ROOT)
                    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    seen[key] = json.loads(lines[-1]) if r.returncode == 0 and lines else {
                        "error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
                if "error" in seen[key]:
                    runs.append({"question": q, "variant": label, "round": rnd, "error": seen[key]["error"]}); failed = True
                    print("%-13s %-44s FAILED: %s" % (q, label, seen[key]["error"][-200:]), flush=True)
                    continue
                m = seen[key]
                results.setdefault((q, label), []).append(m["ms_per_step"]); digests.add(m["digest"])
                runs.append(dict(m, question=q, variant=label, round=rnd, environment=env))
                print("%-13s %-44s %8.3f ms/step  digest %s" % (q, label, m["ms_per_step"], m["digest"][:12]), flush=True)
        if args.dry_run: continue
        agree = len(digests) == 1
        failed = failed or not agree
        record["workloads"][wl] = {"all_variants_bit_equal": agree, "digests": sorted(digests), "runs": runs,
            "decision": decide(results, rehearsal) if agree else "none: the variants do NOT agree -- a correctness failure, fix it before anything is timed"}
    if args.dry_run:
        print("[dry run] would write %s" % os.path.join(args.out, "defaults.json"))
        return 0
    with open(os.path.join(args.out, "defaults.json"), "w") as f:
        json.dump(record, f, indent=1); f.write("\n")
    print("first_contact_defaults: %s -> %s" % ("FAILED" if failed else ("rehearsal complete" if rehearsal else "decided"),
        os.path.join(args.out, "defaults.json")))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
