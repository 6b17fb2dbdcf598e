"""The solver process's progress channels: the GUI protocol ([[LUW_PROGRESS]]{json}, LUW_PROGRESS_MODE=gui) and the console's
running MLUPs | Bandwidth | Steps/s row.  The grammar (prefix, keys, stage -> label) is the committed fixture
tests/golden/progress_protocol.json, read off the reference's emitter and its Python / GUI consumers."""
import glob
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")
SPEC = json.load(open(os.path.join(GOLD, "progress_protocol.json")))


def run_driver(tmp_path, case, extra, env):
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    proj = str(tmp_path / case)
    shutil.copytree(os.path.join(GOLD, "refcases", case), proj)
    deck = glob.glob(os.path.join(proj, "conf.luw*"))[0]
    r = subprocess.run([DRIVER, deck] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    return r.stdout, proj


def events(stdout):
    out = []
    for line in stdout.splitlines():
        if SPEC["prefix"] in line:
            assert line.startswith(SPEC["prefix"]), line           # a protocol line stands alone on its line
            ev = json.loads(line[len(SPEC["prefix"]):])             # what core/luw_progress.py's consumer does
            assert list(ev.keys()) == SPEC["keys"], ev
            assert SPEC["labels"][ev["stage"]] == ev["label"], ev
            assert isinstance(ev["current"], int) and isinstance(ev["total"], int) and isinstance(ev["indeterminate"], bool) and isinstance(ev["detail"], str)
            out.append(ev)
    return out


def test_host_stage_events_without_a_gpu(tmp_path):
    """--dry-run: everything up to the solver runs on the host; the pre-solver stages are announced, in the reference's order"""
    out, _ = run_driver(tmp_path, "CaseN2", ["--dry-run"], {"LUW_PROGRESS_MODE": "gui"})
    ev = events(out)
    stages = [e["stage"] for e in ev]
    assert stages[:2] == ["load_stl", "load_stl"] and ev[0]["current"] == 0 and ev[1]["current"] == 1
    assert "interface_interpolation" in stages and "flux_correction" in stages
    assert stages.index("interface_interpolation") < stages.index("flux_correction")
    fc = [e for e in ev if e["stage"] == "flux_correction"]
    assert fc[0]["indeterminate"] is True and fc[-1]["indeterminate"] is False and "avg dU" in fc[-1]["detail"]
    quiet, _ = run_driver(tmp_path / "q", "CaseN2", ["--dry-run"], {"LUW_PROGRESS_MODE": ""})
    assert SPEC["prefix"] not in quiet                                   # the channel is off unless the parent asks for it


@pytest.mark.gpu
def test_gui_protocol_over_a_whole_run(tmp_path):
    out, proj = run_driver(tmp_path, "CaseV", ["--ddf", "fp32"], {"LUW_PROGRESS_MODE": "gui"})
    ev = events(out)
    first = {}
    for i, e in enumerate(ev):
        first.setdefault(e["stage"], i)
    order = [s for s in SPEC["stage_order"] if s in first]
    assert order == SPEC["stage_order"], first                          # every stage of a profile run appears ...
    assert [first[s] for s in order] == sorted(first[s] for s in order)  # ... in the reference's order
    solve = [e for e in ev if e["stage"] == "solve"]
    assert solve[-1]["current"] == solve[-1]["total"] > 0 and all(not e["indeterminate"] for e in solve)
    assert [e["current"] for e in solve] == sorted(e["current"] for e in solve)
    assert re.match(r"^\d+/\d+ steps \| [0-9.]+ Steps/s \| ETA ", solve[-1]["detail"])
    saved = [e for e in ev if e["stage"] == "save"]
    assert any(e["detail"].endswith(".vtk") or ".vtk" in e["detail"] for e in saved)
    assert SPEC["table_header"] not in out                               # the GUI gets protocol lines instead of the table (FX/info.cpp:225)
    log = glob.glob(os.path.join(proj, "proj_temp", "*_lbm.log"))
    assert log and SPEC["prefix"] not in open(log[0]).read()             # protocol lines are not logged


@pytest.mark.gpu
def test_console_running_row(tmp_path):
    out, proj = run_driver(tmp_path, "CaseV", ["--ddf", "fp32"], {"LUW_PROGRESS_MODE": ""})
    assert SPEC["table_header"] in out and SPEC["prefix"] not in out
    rows = re.findall(r"\|\s*(\d+)\s*\|\s*(\d+) GB/s\s*\|\s*(\d+)\s*\|\s*(\d+)\s+(\d+)%\s*\|\s*([0-9dhms ]+?)\s*\|", out)
    assert rows, out[-1500:]
    mlups, gbs, sps, t, pct, eta = rows[-1]
    assert int(pct) == 100 and int(mlups) > 0 and int(sps) > 0 and eta.strip() == "0s"
    assert abs(int(gbs) - int(mlups) * 153 / 1000) <= max(2, 0.02 * int(gbs))      # GB/s = MLUPs x bytes per cell (FP32: 153), FX/info.cpp:64-65
    log = open(glob.glob(os.path.join(proj, "proj_temp", "*_lbm.log"))[0]).read()
    assert log.count(" GB/s") == 1                                       # only the final row is logged
