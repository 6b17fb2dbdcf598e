"""bench.py, the printed line: stdout carries ONE short JSON line (at most LINE_LIMIT bytes: the contract's keys, `roofline`, `cpu_baseline`, a parity
summary and two numbers per secondary block); everything else a run measured -- the blocks in full, per-rank topology, the self-check's cases -- goes to a file
whose path the line names (`secondary_file`).  Round 4's line had grown to 20 KB and the driver could not parse it any more (BENCH_r04.json)."""
import json
import math
import os

from .common import ROOT

LINE_LIMIT = 4096
FULL_DEFAULT = os.path.join("gpurun_out", "bench_secondary.json")


def strict(v):
    """the same value with nothing `json.loads` of a strict parser would refuse: NaN / infinities become null, numpy scalars become Python numbers"""
    if isinstance(v, dict):
        return {str(k): strict(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [strict(x) for x in v]
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return v if math.isfinite(v) else None
    if hasattr(v, "item"):
        return strict(v.item())
    return str(v)


def pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")     # always present


def block_summary(b):
    """two numbers per secondary block: wall time of a step and the fraction of the HBM roofline (wall clock for single-GPU blocks; for rank shapes the
    whole step -- shell, exchange, interior -- over the rank's owned cells)"""
    if not isinstance(b, dict):
        return None
    if "error" in b:
        return {"error": clip(str(b["error"]), 80)}
    s = {"ms": b.get("ms_per_step")}
    r = b.get("roofline") or {}
    if r.get("frac") is not None:
        s["frac"] = r["frac"]
    if "fp16c" in str(b.get("dtype", "")):
        s["arith"] = b.get("arith")
    for twin, name in (("exact", "exact_frac"), ("peer_loopback", "peer_frac")):
        if isinstance(b.get(twin), dict):
            s[name] = "error" if "error" in b[twin] else (b[twin].get("roofline") or {}).get("frac")
    return s


def parity_summary(p):
    """the parity block of the N = 1 line: FP32 against the reference's FP32 build and FP16C against its shipped build (K = 64 steps, RMSE of u in lattice
    units), the distance of the reference's two builds from EACH OTHER on the same deck, and the full-size planes of configs[0]"""
    if not isinstance(p, dict):
        return None
    if "error" in p:
        return {"error": clip(str(p["error"]), 120)}
    out = pick(p, "u_rmse_vs_reference", "unit", "steps", "tolerance", "case")
    sh = p.get("shipped") or {}
    k64 = {c: pick(sh[c].get("exact", {}), "K64").get("K64") for c in sh if isinstance(sh.get(c), dict) and "exact" in sh[c]}
    if k64:
        out["fp16c_K64"] = k64
    nat = {c: pick(sh[c].get("native", {}), "K64").get("K64") for c in sh if isinstance(sh.get(c), dict) and "native" in sh[c]}
    if any(v is not None for v in nat.values()):
        out["fp16c_native_K64"] = nat
    if p.get("fp32", {}).get("within_tolerance") is not None:
        out["fp32_within_tolerance"] = p["fp32"]["within_tolerance"]
    if "within_tolerance_at_K64" in sh:
        out["fp16c_within_tolerance_at_K64"] = sh["within_tolerance_at_K64"]
    if isinstance(p.get("reference_self_distance"), dict):     # the reference's FP32 build against its own FP16C build, K = 64
        out["reference_self_distance_K64"] = {c: v.get("K64") for c, v in p["reference_self_distance"].items() if isinstance(v, dict)}
    if isinstance(p.get("c1_planes"), dict):                   # configs[0] at full size (128^3, K = 100) against the real reference's planes
        out["c1_K100"] = pick(p["c1_planes"], "fp32", "fp16c_native_vs_shipped", "fp16c_exact_vs_shipped", "reference_fp32_vs_shipped", "error")
    return out


def compact_single(full):
    out = {k: full.get(k) for k in CONTRACT_KEYS}
    out.update(pick(full, "dtype", "data"))
    cfg = full.get("config", {})
    out["config"] = dict(pick(cfg, "global_lattice", "n_gpu", "bytes_per_lup", "arith", "kernel", "solid_fraction"),
        workload=clip(cfg.get("workload", ""), 260))
    if isinstance(cfg.get("placement"), dict):      # workgroup order of the step kernels (the line says which order was timed)
        out["config"]["rows_per_xcd"] = cfg["placement"].get("rows_per_xcd")
    roof = full.get("roofline", {})
    out["roofline"] = dict(pick(roof, "bound", "achieved", "peak", "unit", "frac", "kernel_ms", "algorithmic_bytes_per_launch", "whole_job_frac"),
        traffic=roof.get("traffic"))
    if roof.get("traffic_source"):
        out["roofline"]["traffic_source"] = clip(roof["traffic_source"], 120)
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = dict(pick(cb, "value", "unit", "cores", "kind", "cpu_model", "dram_GBps", "copy_bandwidth_GBps"),
            sample=clip(cb.get("sample", ""), 160))
        ref = (full.get("reference_context") or {}).get("reference_gpu_opencl_mlups")
        if ref:     # the reference's own OpenCL kernels on an MI355X (context, another session): one file holds both baselines
            out["cpu_baseline"]["reference_gpu_opencl_mlups"] = ref
    if "parity" in full:
        out["parity"] = parity_summary(full["parity"])
    dev = full.get("device", {})
    if dev:
        out["device"] = pick(dev, "name", "copy_GBps", "mclk", "fclk")
        if "step_probe" in dev:
            out["device"]["box_factor"] = dev["step_probe"].get("box_factor")
    if "timed_region_note" in full:
        out["timed_region_note"] = clip(full["timed_region_note"], 200)
    if "secondary" in full:
        out["secondary"] = {k: block_summary(v) for k, v in full["secondary"].items()}
    return out


def compact_multi(full):
    out = {k: full.get(k) for k in CONTRACT_KEYS}
    out.update(pick(full, "dtype", "data"))
    if full.get("error"):
        out["error"] = clip(str(full["error"]), 200)
    cfg = full.get("config", {})
    out["config"] = dict(pick(cfg, "global_lattice", "n_gpu", "cells_per_gpu", "bytes_per_lup", "kernel"), workload=clip(cfg.get("workload", ""), 260),
        halo_exchange=clip(cfg.get("halo_exchange", ""), 120))
    if isinstance(cfg.get("schedule_probe"), dict):        # the start-up probe of the two step schedules: what was measured on the slowest rank, what was kept
        sp = cfg["schedule_probe"]
        out["config"]["schedule_probe"] = {"shell_first_ms": sp.get("shell_first_ms"), "whole_box_ms": sp.get("whole_box_ms"),
            "kept": clip(sp.get("kept", ""), 48)}
    out["rccl"] = {"version": cfg.get("rccl_version"), "world_size": cfg.get("ranks_in_communicator")}
    roof = full.get("roofline", {})
    out["roofline"] = dict(pick(roof, "bound", "achieved", "peak", "unit", "frac", "kernel_ms", "whole_job_frac"), traffic=roof.get("traffic"))
    par = full.get("parity", {})
    if "cases" in par:
        out["parity"] = {"ok": par.get("ok"), "cases": len(par["cases"]), "equal": sum(1 for c in par["cases"] if c.get("equal")),
            "what": "rho, u, 19 DDF planes of every rank's owned cells against the CPU oracle on the undivided lattice, through the timed transport",
            "transport": clip(par.get("transport", ""), 60)}
    elif par:
        out["parity"] = {k: clip(v, 80) for k, v in par.items()}
    ranks = []
    for r in full.get("per_rank") or []:
        links = {k: (v.get("link") or ("error" if "error" in v else None)) for k, v in (r.get("links") or {}).items()}
        ranks.append(dict(pick(r, "rank", "kernel_ms", "exchange_ms"), bus=r.get("pci_bus_id"), ms=r.get("wall_ms_per_step"), links=links))
    if ranks:
        out["ranks"] = ranks
    sec = {}
    for k, v in (full.get("secondary") or {}).items():
        if k == "group_host" and isinstance(v, dict) and "error" not in v:
            sec[k] = {lab: (dict(pick(b, "value", "ms_per_step", "transport"), parity=(b.get("parity") or {}).get("equal")) if "error" not in b
                else {"error": clip(str(b["error"]), 80)}) for lab, b in v.items() if isinstance(b, dict)}
        elif isinstance(v, dict) and "error" in v:
            sec[k] = {"error": clip(str(v["error"]), 100)}
        elif isinstance(v, dict):
            sec[k] = pick(v, "value", "ms_per_step", "n_gpu", "roofline_frac_rank0_kernel")
    if sec:
        out["secondary"] = sec
    return out


# keys given up one after the other, least important first, if a line should still come out longer than the limit (it does not for anything bench.py
# measures today: tests/test_bench_line.py builds the worst case)
SHED_ORDER = ("timed_region_note", "device", "ranks", "secondary", "parity")


def render(full, full_path=None):
    """the line to print: compact form of `full`, strict JSON, at most LINE_LIMIT bytes"""
    full = strict(full)
    line = compact_multi(full) if (full.get("n_gpus", 1) > 1 or "per_rank" in full) else compact_single(full)
    if full_path:
        line["secondary_file"] = full_path
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for k in SHED_ORDER:
        if len(text) <= LINE_LIMIT:
            break
        if k in line:
            line[k] = {"see": "secondary_file"}
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        # last resort, never an exception (a line must come out, also from the watchdog's timer thread): the contract's keys, where the rest is, and why
        bare = {k: line.get(k) for k in CONTRACT_KEYS}
        bare.update(secondary_file=full_path, error=clip("line of %d bytes cut to the contract keys; %s" % (len(text), line.get("error") or ""), 200))
        text = json.dumps(bare, allow_nan=False, separators=(",", ":"))
    return text


def full_path_from_env():
    return os.environ.get("LUW_BENCH_FULL_JSON") or os.path.join(ROOT, FULL_DEFAULT)


def write_full(full, path=None):
    """everything that was measured, indented, in a file (never on stdout); returns the path written, relative to the repo where it lies inside it"""
    path = path or full_path_from_env()
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(strict(full), f, indent=1, allow_nan=False)
            f.write("\n")
    except OSError:
        return None
    ap = os.path.abspath(path)
    return os.path.relpath(ap, ROOT) if ap.startswith(ROOT + os.sep) else ap


def emit(fd, full, path=None):
    """write the full record to its file, then the short line to file descriptor `fd`"""
    p = write_full(full, path)
    os.write(fd, (render(full, p) + "\n").encode())
