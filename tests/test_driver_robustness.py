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


# ---------------------------------------------------------------- the deck's input files: STL, SurfData CSV, profile table
def _only(directory, ext):
    return os.path.join(directory, [f for f in sorted(os.listdir(directory)) if f.endswith(ext)][0])


def _rewrite(path, fn, binary=False):
    data = open(path, "rb" if binary else "r").read()
    open(path, "wb" if binary else "w").write(fn(data))


def _stl(fn):
    return lambda case: _rewrite(_only(os.path.join(case, "proj_temp"), ".stl"), fn, binary=True)


def _csv(fn):
    return lambda case: _rewrite(_only(os.path.join(case, "proj_temp"), ".csv"), fn)


def _profile(fn):
    return lambda case: _rewrite(os.path.join(case, "wind_bc", "profile.dat"), fn)


def _every(lines_fn, step):
    def f(text):
        l = text.splitlines()
        for i in range(1, len(l), step):
            l[i] = lines_fn(l[i])
        return "\n".join(l) + "\n"
    return f


def _nan_vertices(b):
    import struct
    b = bytearray(b)
    for off in range(84 + 12, min(len(b), 84 + 50 * 5), 50):
        b[off:off + 4] = struct.pack("<f", float("nan"))
    return bytes(b)


INPUT_VARIANTS = {
    "stl_truncated": ("CaseA", _stl(lambda b: b[:len(b) // 2 + 7])),
    "stl_triangle_count_2e9": ("CaseA", _stl(lambda b: b[:80] + b"\xff\xff\xff\x7f" + b[84:])),
    "stl_empty_file": ("CaseA", _stl(lambda b: b"")),
    "stl_header_only": ("CaseA", _stl(lambda b: b"\0" * 84)),
    "stl_nan_vertices": ("CaseA", _stl(_nan_vertices)),
    "stl_ascii": ("CaseA",
        _stl(lambda b: b"solid x\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nendloop\nendfacet\nendsolid x\n")),
    "stl_truncated_dataset_mode": ("CaseDG", _stl(lambda b: b[:len(b) // 2 + 7])),
    "csv_empty": ("CaseN1", _csv(lambda t: "")),
    "csv_header_only": ("CaseN1", _csv(lambda t: "X,Y,Z,u,v,w\n")),
    "csv_ragged_rows": ("CaseN1", _csv(_every(lambda l: ",".join(l.split(",")[:2]), 7))),
    "csv_nan_inf": ("CaseN1", _csv(_every(lambda l: ",".join(["1", "inf", "2", "nan"] + l.split(",")[4:]), 5))),
    "csv_text_rows": ("CaseN1", _csv(_every(lambda l: "a,b,c,d,e,f", 3))),
    "profile_empty": ("CaseA", _profile(lambda t: "")),
    "profile_one_point": ("CaseA", _profile(lambda t: "10 5\n")),
    "profile_junk_lines": ("CaseA", _profile(lambda t: "z u\nfoo bar\n10 nan\n-5 3\n10 4\n10 5\n")),
    "profile_unsorted": ("CaseA", _profile(lambda t: "\n".join(reversed(t.splitlines())) + "\n")),
}


@pytest.mark.parametrize("name", sorted(INPUT_VARIANTS))
def test_malformed_input_file_ends_in_an_orderly_exit(driver, tmp_path, name):
    base, damage = INPUT_VARIANTS[name]
    case = str(tmp_path / "case")
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "refcases", base), case)
    damage(case)
    deck = [os.path.join(case, f) for f in os.listdir(case) if f.startswith("conf.")][0]
    r = subprocess.run([driver, deck, "--dry-run"], capture_output=True, text=True, timeout=120)
    assert r.returncode >= 0, "killed by signal %d\n%s" % (-r.returncode, r.stderr[-2000:])
    assert r.returncode in (0, 1, 255), (r.returncode, r.stdout[-1500:])
