"""CPU checks of the product's C-ABI: the library builds for gfx950, loads, exports every symbol that
include/luw_core.h declares, and fails loudly (no fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="luw_core.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(luw_[a-z_A-Z0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(luw):
    from latticeurbanwind_amd import capi
    L = capi.load()
    names = declared_functions()
    assert len(names) >= 25
    for n in names + declared_functions("luw_core_dev.h"):
        assert hasattr(L, n), "missing export " + n
    assert sorted(capi.SYMBOLS) == names
    assert sorted(capi.DEV_SYMBOLS) == declared_functions("luw_core_dev.h")
    assert L.luw_abi_version() == 6


def test_product_header_is_the_boundary_only():
    """include/luw_core.h names the three product kernels and nothing of the lab: A/B kernel ids, timed runs, self-checks and raw DDF access live in
    include/luw_core_dev.h; the product sources carry no A/B kernel file"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "luw_core.h")).read(), flags=re.S)
    assert sorted(re.findall(r"#define (LUW_KERNEL_\w+)", hdr)) == ["LUW_KERNEL_AUTO", "LUW_KERNEL_PAIR", "LUW_KERNEL_SCALAR"]
    for lab in ("luw_run_timed", "luw_selfcheck", "luw_download_fi", "luw_dev_"):
        assert lab not in hdr, lab
    assert not os.path.exists(os.path.join(ROOT, "latticeurbanwind_amd", "csrc", "luw_kernels_vec.hpp"))


def test_environment_knobs_are_the_documented_ones(luw):
    """the library reads its environment in ONE place (the tuning table of luw_core.hip); the names in the binary, the names the table prints and the
    names INTEGRATION.md documents are the same list"""
    import subprocess
    from latticeurbanwind_amd import capi
    so = os.path.join(ROOT, "latticeurbanwind_amd", "csrc", "libluw_core.so")
    strings = subprocess.run(["strings", so], capture_output=True, text=True).stdout
    in_binary = sorted(set(re.findall(r"^(LUW_[A-Z_0-9]+)$", strings, flags=re.M)))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("## 6. Environment knobs"):doc.index("### Other hosts")]
    documented = sorted(set(re.findall(r"^\| `(LUW_[A-Z_0-9]+)", table, flags=re.M)))
    assert in_binary == documented, (in_binary, documented)
    src = "".join(open(os.path.join(ROOT, "latticeurbanwind_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "latticeurbanwind_amd", "csrc"))
        if f.endswith((".hip", ".hpp")))
    calls = re.findall(r"getenv\(", src)
    loader = src[src.index("static void tuning_load()"):src.index("static const Tuning& tuning()")]
    assert len(calls) == len(re.findall(r"getenv\(", loader)), "getenv outside the tuning table"
    try:
        printed = sorted(kv.split("=")[0] for kv in capi.tuning_text().split())
    except capi.LuwError:
        printed = None
    assert printed is None or printed == documented


def test_config_struct_layout_matches_header(luw):
    from latticeurbanwind_amd import capi
    hdr = open(os.path.join(ROOT, "include", "luw_core.h")).read()
    body = hdr[hdr.index("typedef struct luw_config {") + len("typedef struct luw_config {"):hdr.index("} luw_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        m = re.match(r"\s*(uint32_t|int32_t|float)\s+(.*)", decl.strip(), flags=re.S)
        if m:
            fields += [(n.strip(), m.group(1)) for n in m.group(2).split(",")]
    ctype = {"uint32_t": C.c_uint32, "int32_t": C.c_int32, "float": C.c_float}
    assert [(n, ctype[t]) for n, t in fields] == list(capi.Config._fields_)


def test_no_cpu_fallback_without_gpu(luw):
    from latticeurbanwind_amd import capi
    L = capi.load()
    n = C.c_int(0)
    rc = L.luw_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.LuwError):
        luw.LBM(8, 8, 8, 0.01)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "latticeurbanwind_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "luw_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_float_text_equals_the_restatement(luw):
    """luw_format_float9 (the product's 9-significant-digit float text, behind def_w / inv_tau constants and VTK headers) against the
    oracle's literal restatement of the reference's to_string(float) + strtof round trip, over edge cases and 200k random floats"""
    import numpy as np
    from latticeurbanwind_amd import capi
    from oracle import oracle
    L = capi.load()
    rng = np.random.default_rng(5)
    vals = np.concatenate([np.float32(
        [0.0, 1.0, 9.9999999, 0.99999999, 1e-7, 1.9999990, 0.57735027, 123456.789, 1e32, 3.4e38, 1.2e-38, 1e-45, 0.1, 0.0001, 99999999.0, 0.5000004]),
                           rng.standard_normal(50000).astype(np.float32), (10.0 ** rng.uniform(-38, 38, 100000)).astype(np.float32),
                               rng.integers(0, 2 ** 32, 50000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    buf = C.create_string_buffer(48)
    for v in vals:
        if not np.isfinite(v):
            continue
        assert L.luw_format_float9(float(v), buf, 48) == 0
        got = np.float32(float(buf.value.decode()))
        want = np.float32(oracle.literal(v))
        assert got == want or (got == 0 and want == 0), (float(v), buf.value)
        assert re.fullmatch(rb"-?\d\.\d{8}(E-?\d+)?", buf.value), buf.value
    assert L.luw_format_float9(float("nan"), buf, 48) == 0 and buf.value == b"NaN"
    assert L.luw_format_float9(-float("inf"), buf, 48) == 0 and buf.value == b"-Inf"
