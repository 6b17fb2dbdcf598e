"""Static check of the compiled gfx950 code (no GPU needed, hipcc cross-compiles): the FP16C kernels switch the wave's FP32
rounding mode to round-toward-zero for their tail encode (luw_device.hpp, fp16c_code_hi_in_rtz_mode).  That is only valid
if NO other floating-point instruction is scheduled behind the switch -- the C++ source orders them with asm fences; this
test reads the generated ISA and holds the compiler to it.  Also pins the properties the performance notes in DESIGN.md
rest on: no scratch spills in the product kernels, DDF stores in the saddr form."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "latticeurbanwind_amd", "csrc")


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    flags = re.search(r"^HIPFLAGS\s*=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1).split()
    flags = [f for f in flags if f not in ("-fPIC",) and not f.startswith("$(")]      # make variables (EXTRA) are empty in the product build
    out = str(tmp_path_factory.mktemp("isa") / "luw_core.s")
    subprocess.check_call([hipcc, *flags, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, "luw_core.hip")],
                          stderr=subprocess.DEVNULL)
    kernels, name, body = {}, None, []
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name and line.startswith(".Lfunc_end"):
            kernels[name] = body; name = None
        elif name:
            t = line.strip()
            if re.match(r"^\.LBB\w+:", t):
                body.append(t.split(":")[0] + ":")               # block labels stay: the RTZ test follows the control flow
            elif t and not t.startswith((";", ".")):
                body.append(t)
    return kernels


FP_ARITH = re.compile(r"^v_(add|sub|subrev|fma|fmac|mad|mac|div|rcp|rsq|sqrt|min|max|med3|cvt|exp|log|sin|cos|ldexp|frexp|trunc|ceil|floor|rndne|fract"
                      r"|pk_fma|pk_add)_?\w*f(16|32|64)")


def reachable_from(body, start):
    """instructions a wave can execute from position `start` on (fall-through and branches followed; blocks that merely
    sit behind the switch in the file but are entered from before it do not count)"""
    label = {t[:-1]: i for i, t in enumerate(body) if t.endswith(":")}
    seen, work = set(), [start]
    while work:
        i = work.pop()
        while i < len(body) and i not in seen:
            seen.add(i)
            t = body[i]
            if t.startswith("s_endpgm"):
                break
            m = re.match(r"^s_(c?branch\w*)\s+(\.LBB\w+)", t)
            if m:
                work.append(label[m.group(2)])
                if m.group(1) == "branch":
                    break
            assert not t.startswith(("s_setpc", "s_swappc")), "indirect jump behind the switch"
            i += 1
    return [body[i] for i in sorted(seen) if not body[i].endswith(":")]


def test_nothing_but_the_encode_follows_the_rounding_mode_switch(device_asm):
    seen = 0
    for name, body in device_asm.items():
        idx = [i for i, t in enumerate(body) if t.startswith("s_setreg")]
        if not idx or "k_codec_check" in name:
            continue
        assert len(idx) == 1, name
        seen += 1
        tail = reachable_from(body, idx[0] + 1)
        bad = [t for t in tail if FP_ARITH.match(t)]
        assert not bad, "%s: floating-point work behind the RTZ switch: %s" % (name, bad[:5])
        muls = [t for t in tail if t.startswith(("v_mul_f32", "v_pk_mul_f32"))]
        stores = [t for t in tail if t.startswith("global_store")]
        assert muls and len(stores) >= 19, name                      # the encode itself and the 19 DDF stores
    assert seen >= 6, "expected the FP16C scalar (both parities, thermal) and pair kernels to use the RTZ encode, found %d" % seen


def test_product_kernels_have_no_spills_and_store_through_saddr(device_asm):
    # scalar kernel: FP32 in both addressing forms (FLAT for planes within 32-bit byte offsets, row form beyond), FP16C in the row form,
    # general and force-free; pair kernel: force modes none / uniform / any in registers, any with the second cell parked in LDS (the product's general kernel),
    # and the three thermal variants (always parked); both time parities each: 8 + 8 + 6; the native-arithmetic instantiations (LUW_OPT_NATIVE_ARITH): the
    # one-cell
    # FP16C kernel, the pair kernel's three force modes and their three thermal variants: 2 + 6 + 6
    product = [n for n in device_asm if re.search(r"k_stream_collide_sI[tf]Li[01]ELi0ELi2ELb[01]ELb0ELb[01]E", n)
        or re.search(r"k_stream_collide_pILi[01]ELi0ELb0ELi[012]E", n)]
    # ... and the instantiations with the x-face output (luw_set_x_face_buffers): FP32 flat / row, the pair kernel's three force modes exact and native, plain
    # and with the thermal lattice: 4 + 12 + 12
    assert len(product) == 64, product
    flags_of = lambda n: re.findall(r"Lb([01])E", n.split("EvN3luw")[0])
    assert sum(1 for n in product if flags_of(n)[-2] == "1") == 14 + 12 and sum(1 for n in product if flags_of(n)[-1] == "1") == 28      # NATIVE, XFACE
    # the same kernels with the statistics epilogue (sampled steps): no spills either
    sampled = [n for n in device_asm if re.search(r"k_stream_collide_sI[tf]Li[01]ELi0ELi2ELb[01]ELb1ELb0E", n)
        or re.search(r"k_stream_collide_pILi[01]ELi0ELb1ELi2E", n)]
    # exact: FP32 flat / row, FP16C one-cell, pair; native: FP16C one-cell, pair; two parities
    assert len(sampled) == 12, sampled
    for name in sampled:
        assert not any(t.startswith(("scratch_", "buffer_store", "buffer_load")) for t in device_asm[name]), name + ": spills"
    for name in product:
        body = device_asm[name]
        assert not any(t.startswith(("scratch_", "buffer_store", "buffer_load")) for t in body), name + ": spills"
        # (the x-face output's five stores per border column go to a buffer through plain 64-bit addresses and carry no nt hint)
        ddf_stores = [t for t in body if re.match(r"global_store_(dword|short)", t) and ("nt" in t.split()[-1] or ("d16_hi" in t and not re.match(
            r"global_store_\w+ v\[", t)) or flags_of(name)[-1] == "0" and "d16_hi" in t)]
        vaddr = [t for t in ddf_stores if re.match(r"global_store_\w+ v\[", t)]
        assert len(ddf_stores) >= 14 and not vaddr, "%s: DDF stores with 64-bit VGPR addresses: %s" % (name, vaddr[:3])
