"""The line bench.py prints (benchmarks/line.py): one strict-JSON line of at most 4 KB whatever was measured -- round 4's line had grown to 20 KB and the
driver stored `parsed: null` for it.  Worst cases are assembled here from blocks shaped like the real ones (every secondary block with both twins, eight
ranks with four links each, error texts, NaN)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchmarks import line as L      # noqa: E402

LONG = "x" * 1500


def roof(frac=0.7431):
    return {"bound": "hbm", "achieved": 5947.3, "peak": 8000.0, "unit": "GB/s", "frac": frac, "traffic": 42131234567, "kernel_ms": 6.7611,
        "algorithmic_bytes_per_launch": 40211234567, "note": LONG, "traffic_source": "profiles/r05_f32_1024x1024x256_bld_summary.json (" + LONG + ")",
        "whole_job_frac": 0.7391, "frac_of_device_copy": 1.1}


def block(err=False):
    if err:
        return {"error": "exit 1: " + LONG}
    b = {"value": 68853.5, "unit": "MLUPS", "ms_per_step": 1.9493, "steps": 200, "warmup": 20, "lattice": [1024, 1024, 256], "workload": LONG, "options": LONG,
        "roofline": roof(0.6587), "placement": {"create_s": 1.2, "text": LONG}}
    b["exact"] = {"value": 1.0, "ms_per_step": 1.9, "roofline": {"frac": 0.66, "kernel_ms": 1.9}}
    b["dtype"], b["arith"] = "fp16c-storage/f32-arithmetic", "native"
    b["peer_loopback"] = {"value": 1.0, "ms_per_step": 1.9, "transport": "peer-loopback", "roofline": {"frac": 0.66}}
    return b


def single_full():
    import bench
    keys = list(bench.SINGLE_BLOCKS) + list(bench.RANK_SHAPE_BLOCKS)
    return {"metric": "MLUPS (D3Q19) at 1/2/4/8 MI355X; % of HBM roofline; u-field RMSE vs ref", "value": 40226.1, "unit": "MLUPS", "n_gpus": 1, "steps": 20,
        "warmup": 5, "ms_per_step": 6.6731, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": LONG, "global_lattice": [1024, 1024, 256], "n_gpu": [1, 1, 1], "halo_exchange": None, "kernel": "auto", "bytes_per_lup": 153.0,
            "solid_fraction": 0.02118, "arith": "exact", "create_s": 2.5, "placement": {"text": LONG}},
        "roofline": roof(), "timed_region_note": LONG,
        "device": {"name": "AMD Instinct MI355X", "copy_GBps": 5102.4, "copy_frac_of_peak": 0.63, "mclk": "2000Mhz", "fclk": "2100Mhz", "sclk": "2400Mhz",
            "step_probe": {"workload": LONG, "ms_per_step": 3.29, "typical_ms_per_step": 3.3, "box_factor": 0.998, "note": LONG}},
        "secondary": {k: block(err=(i == 3)) for i, k in enumerate(keys)},
        "cpu_baseline": {"value": 513.8, "unit": "MLUPS", "cores": 16, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "dram_GBps": 86.8,
            "copy_bandwidth_GBps": 150.2, "dram_frac_of_copy": 0.58, "path": LONG, "sample": LONG},
        "parity": {"u_rmse_vs_reference": 1.2e-7, "unit": "lattice units", "steps": 64, "tolerance": 1e-5,
            "case": "tests/golden/refcases/CaseB (48x40x24, FP32 DDFs)",
            "fp32": {"K8": 1e-8, "K64": 1.2e-7, "u_avg": float("nan"), "within_tolerance": True},
            "shipped": {"CaseA": {"exact": {"K8": 1e-7, "K64": 2.65e-5, "u_avg": 1.5e-5}, "native": {"K8": 1e-7, "K64": 2.62e-5, "u_avg": 1.48e-5}},
                "CaseL": {"exact": {"K8": 1e-7, "K64": 4.0e-6, "u_avg": 2.4e-6}, "native": {"K8": 1e-7, "K64": 4.0e-6, "u_avg": 2.4e-6}},
                "precision": LONG, "within_tolerance_at_K8": True, "within_tolerance_at_K64": False, "note": LONG},
            "reference_self_distance": {"CaseA": {"K8": 1e-5, "K64": 2.2e-5, "u_avg": 1.4e-5}, "CaseL": {"K8": 2e-6, "K64": 3.2e-6, "u_avg": 2.3e-6},
                "what": "FP32 build vs shipped build of the reference, same deck"},
            "c1_planes": {"fp32": 2.6e-7, "fp16c_native_vs_shipped": 3.8e-5, "fp16c_exact_vs_shipped": 3.1e-5, "reference_fp32_vs_shipped": 3.2e-5,
                "steps": 100,
                "lattice": [128, 128, 128], "what": LONG}, "horizon": LONG}}


def multi_full(world=8):
    links = lambda: {k: {"peer": 1, "can_access": True, "performance_rank": 0, "native_atomics": 1, "link": "xgmi", "hops": 1, "rank": 3}
        for k in ("x+", "x-", "y+", "y-")}
    ranks = [{"rank": r, "device": r, "coord": [r % 4, r // 4, 0], "local_lattice": [514, 514, 512], "wall_ms_per_step": 3.6012, "kernel_ms": 1.7093,
        "kernel_cells": 67108864, "shell_ms": 1.8, "exchange_ms": 0.4512, "halo_bytes_out_per_step": 21053440, "exchange_GBps_out": 46.7,
        "wire": {"x": {"ms": 0.11, "GBps": 95.1}, "y": {"ms": 0.11, "GBps": 95.1}}, "device_copy_GBps": 5100.0, "mclk": "2000Mhz", "fclk": "2100Mhz",
        "pci_bus_id": "0000:%02x:00.0" % (5 + 16 * r), "links": links()} for r in range(world)]
    case = {"dtype": "f32", "coriolis": False, "lattice": [1536, 128, 64], "n_gpu": [4, 2, 1], "steps": 8, "forcing": LONG, "schedule": LONG, "equal": True,
        "mismatches": [], "cells_compared": 1536 * 128 * 64, "max_abs_uy": 0.01, "compared": LONG}
    gh = {"what": LONG, "devices": list(range(world)), "n_gpu": [4, 2, 1], "global_lattice": [2048, 1024, 512]}
    for lab in ("peer", "peer_threads", "rccl"):
        gh[lab] = {"parity": {"equal": True, "lattice": [1536, 128, 64], "compared": LONG}, "transport": "peer stores", "overlap": True, "value": 251234.5,
            "unit": "MLUPS", "ms_per_step": 4.2, "domain0_kernel_ms": 1.7, "direct_peer_stores": True, "host_threads": "one", "process_wall_s": 41.2}
    gh["rccl"] = {"error": LONG}
    return {"metric": "MLUPS (D3Q19) at 1/2/4/8 MI355X; % of HBM roofline; u-field RMSE vs ref", "value": 298123.4, "unit": "MLUPS", "n_gpus": world,
        "steps": 200,
        "warmup": 20, "ms_per_step": 3.6012, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": LONG, "global_lattice": [2048, 1024, 512], "n_gpu": [4, 2, 1], "cells_per_gpu": 512 ** 3, "halo_exchange": LONG,
            "kernel": "auto",
            "bytes_per_lup": 153.0, "rccl_version": "2.26.6", "ranks_in_communicator": world,
            "schedule_probe": {"shell_first_ms": 3.6123, "whole_box_ms": float("inf"), "kept": "shell first, exchange beside the interior", "probe_steps": 20,
                "rule": LONG}},
        "roofline": dict(roof(), note=LONG), "parity": {"transport": LONG, "ok": True, "cases": [case] * 4}, "per_rank": ranks,
        "secondary": {"x_whole_n_gpu": {"value": 300000.1, "unit": "MLUPS", "ms_per_step": 3.5, "n_gpu": [1, 4, 2], "global_lattice": [2048, 1024, 512],
            "what": LONG, "halo_exchange": LONG, "roofline_frac_rank0_kernel": 0.74, "per_rank": ranks}, "group_host": gh}}


def parse_strict(text):
    def refuse(name):
        raise ValueError(name)
    return json.loads(text, parse_constant=refuse)         # NaN / Infinity are not JSON


def test_single_gpu_line_worst_case_fits_and_is_strict_json():
    text = L.render(single_full(), "gpurun_out/bench_secondary.json")
    assert len(text) <= L.LINE_LIMIT < 8192 and "\n" not in text
    d = parse_strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline", "parity", "secondary", "secondary_file"):
        assert k in d, k
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes_per_launch"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "cpu_model", "dram_GBps"}
    assert d["config"]["workload"] and d["config"]["global_lattice"] == [1024, 1024, 256] and "model" not in d["config"]
    assert d["parity"]["u_rmse_vs_reference"] == 1.2e-7 and d["parity"]["fp16c_K64"]["CaseA"] == 2.65e-5 and d["parity"]["reference_self_distance_K64"] == {
        "CaseA": 2.2e-5, "CaseL": 3.2e-6}
    import bench
    assert set(d["secondary"]) == set(bench.SINGLE_BLOCKS) | set(bench.RANK_SHAPE_BLOCKS)          # nothing shed: two numbers per block fit
    ok = d["secondary"]["c2_f32"]
    assert ok == {"ms": 1.9493, "frac": 0.6587, "arith": "native", "exact_frac": 0.66, "peer_frac": 0.66}
    assert sum("error" in v for v in d["secondary"].values()) == 1


def test_default_blocks_are_known_and_few():
    import bench
    assert set(bench.DEFAULT_BLOCKS) <= set(bench.SINGLE_BLOCKS) | set(bench.RANK_SHAPE_BLOCKS) and len(bench.DEFAULT_BLOCKS) <= 9
    for must in ("c2_f32", "c3_fp16c", "c3_fp16c_thermal", "cube1024_f32", "cube1024_fp16c", "tile512_urban_fp16c_coriolis", "c4_rank_4x2x1_f32",
            "c5_rank_4x2x1_fp16c_coriolis"):
        assert must in bench.DEFAULT_BLOCKS


def test_rank_shape_blocks_cut_the_one_tile_the_gpu_tests_check():
    """round 4 built `[1,4,2]` blocks from a 512x2048x1024 lattice (ranks of 512x514x514) and labelled them as ranks of the 2048x1024x512 tile"""
    import bench
    from latticeurbanwind_amd.layout import DomainLayout
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_bench_workloads import RANK_CASES
    tested = {(D, tuple(b * d for b, d in zip(blk, D))) for _, blk, D, _, opts in RANK_CASES if "peer" not in opts}
    for key, spec in bench.RANK_SHAPE_BLOCKS.items():
        gN = bench.rank_shape_lattice(spec["D"])
        assert gN == (2048, 1024, 512), key
        lay = DomainLayout(gN, spec["D"], spec["rank"])
        assert tuple(lay.lN) == {(4, 2, 1): (514, 514, 512), (1, 4, 2): (2048, 258, 258)}[spec["D"]], key
        assert (spec["D"], gN) in tested, key


def test_multi_gpu_line_worst_case_fits_and_names_every_rank():
    text = L.render(multi_full(8), "gpurun_out/bench_secondary.json")
    assert len(text) <= L.LINE_LIMIT
    d = parse_strict(text)
    assert d["n_gpus"] == 8 and d["rccl"] == {"version": "2.26.6", "world_size": 8} and d["parity"] == dict(d["parity"], ok=True, cases=4, equal=4)
    assert [r["bus"] for r in d["ranks"]] == ["0000:%02x:00.0" % (5 + 16 * r) for r in range(8)]
    assert all(r["links"] == {"x+": "xgmi", "x-": "xgmi", "y+": "xgmi", "y-": "xgmi"} for r in d["ranks"])
    assert d["secondary"]["x_whole_n_gpu"]["n_gpu"] == [1, 4, 2] and d["secondary"]["group_host"]["peer"]["parity"] is True
    assert "error" in d["secondary"]["group_host"]["rccl"]


def test_nan_becomes_null_and_an_oversized_line_sheds_blocks_not_the_contract():
    full = single_full()
    full["value"] = float("nan")
    full["secondary"] = {"block_%03d" % i: block() for i in range(200)}        # far more than bench.py has
    d = parse_strict(L.render(full, None))
    assert d["value"] is None and d["secondary"] == {"see": "secondary_file"} and d["roofline"]["frac"] == 0.7431 and d["cpu_baseline"]["cores"] == 16


def test_full_record_goes_to_a_file_and_the_line_cites_it(tmp_path):
    path = str(tmp_path / "sub" / "full.json")
    r, w = os.pipe()
    L.emit(w, single_full(), path)
    os.close(w)
    text = os.read(r, 1 << 16).decode()
    os.close(r)
    assert text.endswith("\n") and text.count("\n") == 1
    assert json.loads(text)["secondary_file"] == path
    rec = json.load(open(path))
    assert rec["secondary"]["c2_f32"]["workload"] == LONG and rec["parity"]["fp32"]["u_avg"] is None


def test_render_never_raises_and_clips_the_error_text():
    # a line that cannot be brought under the limit by shedding (a config text nobody clipped): the contract keys still come out, with a short reason
    full = multi_full(8)
    full["error"] = "E" * 9000
    text = L.render(full, "gpurun_out/bench_secondary.json")
    assert len(text) <= L.LINE_LIMIT and len(json.loads(text)["error"]) <= 200
    full["metric"] = "m" * 200; full["unit"] = "MLUPS"
    huge = dict(full, n_gpus=1); huge.pop("per_rank", None)
    huge["config"] = {"workload": "w"}
    huge["roofline"] = dict(roof(), bound="b" * 5000)                 # (nothing sheds the roofline)
    text = L.render(huge, "gpurun_out/bench_secondary.json")
    d = json.loads(text)
    assert len(text) <= L.LINE_LIMIT and all(k in d for k in L.CONTRACT_KEYS) and d["secondary_file"] and "cut to the contract keys" in d["error"]


def test_reference_context_sits_beside_the_cpu_baseline():
    # a reader of ONE record sees both baselines -- the CPU restatement timed by this run and the reference's own OpenCL kernels on an
    # MI355X
    from benchmarks.common import REFERENCE_GPU_CONTEXT
    full = single_full(); full["reference_context"] = REFERENCE_GPU_CONTEXT
    d = json.loads(L.render(full, "gpurun_out/bench_secondary.json"))
    assert d["cpu_baseline"]["reference_gpu_opencl_mlups"] == {"fp32_build": 15189, "shipped_fp16c_thermal_build": 15673} and d["cpu_baseline"][
        "kind"] == "port"
    assert "profiles/r05_reference_perf" in REFERENCE_GPU_CONTEXT["source"] and os.path.exists(os.path.join(ROOT, REFERENCE_GPU_CONTEXT["source"]))


def test_the_schedule_probe_is_on_the_multi_gpu_line():
    # what the start-up probe measured and kept (an infinite time -- a schedule that could not run on some rank -- becomes null: strict JSON)
    d = parse_strict(L.render(multi_full(8), "gpurun_out/bench_secondary.json"))
    sp = d["config"]["schedule_probe"]
    assert sp["shell_first_ms"] == 3.6123 and sp["whole_box_ms"] is None and sp["kept"].startswith("shell first") and "rule" not in sp
