"""The deck reader of the C++ driver on malformed input (CPU, --dry-run): whatever a deck file holds, the process ends by itself --
exit code 0 or the driver's error exit, never a signal -- the way the reference's reader shrugs off bad lines (FX/setup.cpp:2775-3320:
unknown keys and unparsable values are skipped, missing essentials end in print_error + exit).  The same variants were run through an
AddressSanitizer / UBSan build of the driver while it was written (no report)."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, "tests", "golden", "refcases", "CaseA")
DRIVER = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")


@pytest.fixture(scope="module")
def driver(luw):
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    return DRIVER


def variants():
    src = open(os.path.join(CASE, "conf.luwpf")).read()
    out = {
        "empty": "",
        "garbage": "\x00\x01\x02 = = = \n[[[[\n" * 50,
        "comment_only": "// only a comment\n",
        "long_line": "casename = " + "x" * 200000 + "\n" + src,
        "unterminated_quote": src.replace('"', "", 1),
        "n_gpu_zero": src + "\nn_gpu = [0, 0, 0]\n",
        "n_gpu_text": src + "\nn_gpu = [a, b]\n",
        "duplicate_keys": src + src,
        "crlf": src.replace("\n", "\r\n"),
        "cell_size_nan": src + '\ncell_size = nan\nmesh_control = "cell_size"\n',
        "cell_size_negative": src + '\ncell_size = -5\nmesh_control = "cell_size"\n',
    }
    rng = random.Random(1)
    lines = src.splitlines()
    for i in range(8):
        l = lines[:]
        for _ in range(3):
            j = rng.randrange(len(l)); op = rng.randrange(4)
            if op == 0: del l[j]
            elif op == 1: l[j] = l[j][:rng.randrange(len(l[j]) + 1)]
            elif op == 2: l[j] = l[j] + l[j]
            else: l[j] = "".join(rng.choice('=[],"; /*') for _ in range(20))
        out["mutation_%d" % i] = "\n".join(l) + "\n"
    return out


VARIANTS = variants()


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_malformed_deck_ends_in_an_orderly_exit(driver, tmp_path, name):
    case = str(tmp_path / "case")
    shutil.copytree(CASE, case)
    deck = os.path.join(case, "conf.luwpf")
    with open(deck, "w") as f:
        f.write(VARIANTS[name])
    r = subprocess.run([driver, deck, "--dry-run"], capture_output=True, text=True, timeout=120)
    assert r.returncode >= 0, "killed by signal %d\n%s" % (-r.returncode, r.stderr[-2000:])
    assert r.returncode in (0, 1, 255), (r.returncode, r.stdout[-1500:])
    if r.returncode != 0:
        assert "Error" in r.stdout or "ERROR" in r.stdout or "error" in r.stdout.lower()
