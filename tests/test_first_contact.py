"""tools/first_contact.sh -- the stage-by-stage script for the first multi-GPU node (devices and links, the one-process host across devices, two ranks over
RCCL with the self-check first, all ranks): its control flow rehearsed without a GPU (--dry-run: every stage prints its command; an injected failure ends
the script with the stage's number, later stages do not start), and on the one-GPU test box with all ranks on device 0 (--share-device 0)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "first_contact.sh")


def test_dry_run_walks_all_stages_in_order(tmp_path):
    r = subprocess.run([SCRIPT, "--dry-run", "--out", str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    marks = [l for l in r.stdout.splitlines() if l.startswith("== stage")]
    assert [m.split(":")[0] for m in marks] == ["== stage %d" % n for n in range(1, 7)] and "first contact complete" in r.stdout
    assert "distinct_devices" in r.stdout and "bench_n 2" in r.stdout
    assert sorted(os.listdir(tmp_path)) == ["stage%d.log" % n for n in range(1, 7)] and r.stdout.count("LUW_SCHEDULE_JITTER") == 2


@pytest.mark.parametrize("fail", [1, 2, 3, 4, 5, 6])
def test_first_failure_ends_the_script_with_the_stage_number(tmp_path, fail):
    r = subprocess.run([SCRIPT, "--dry-run", "--out", str(tmp_path)], capture_output=True, text=True, timeout=60,
        env=dict(os.environ, FIRST_CONTACT_FAIL=str(fail)))
    assert r.returncode == fail
    started = [int(l.split()[2].rstrip(":")) for l in r.stdout.splitlines() if l.startswith("== stage") and "FAILED" not in l]
    assert started == list(range(1, fail + 1)) and "first contact complete" not in r.stdout


def test_no_stage_replaces_a_gpu_process_by_another_program():
    text = open(SCRIPT).read()
    code = [l for l in text.splitlines() if not l.lstrip().startswith("#")]
    assert not any(l.lstrip().startswith("exec ") or " exec " in l for l in code)
    assert "retry" not in text.lower().replace("retried", "") and "while true" not in text


@pytest.mark.gpu
def test_rehearsal_on_one_gpu(tmp_path):
    r = subprocess.run([SCRIPT, "--share-device", "0", "--out", str(tmp_path)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "2 ranks:" in r.stdout and "4 ranks:" in r.stdout and "parity ok" in r.stdout and "first contact complete" in r.stdout
