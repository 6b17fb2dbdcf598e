"""The workgroup order of the step kernels (xcd_row_order, csrc/luw_device.hpp; KParams::xcd_rows = G set by luw_create for lattices with large DDF planes,
LUW_XCD_ROWS): the hardware hands consecutive workgroups of a launch to the 8 XCDs in turn; the kernels permute (blockIdx.x, blockIdx.y) within each z layer
so that all blocks of one lattice row run on ONE XCD, G consecutive rows per XCD and turn.  Restated here from the header's formula and checked for what
the kernels rely on: a permutation of the launch's blocks (every cell updated exactly once), one XCD per row in the remapped part, the remainder rows
untouched.  The kernels themselves are held to the oracle with the order on (tests/test_gpu_parity.py)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def order(bx, by, nbx, ny, G):
    full = ny - ny % (8 * G) if G else 0
    if not G or by >= full:
        return bx, by
    pp = bx + nbx * by
    sq, r = pp >> 3, (pp >> 3) // nbx
    return sq % nbx, (r // G) * 8 * G + (pp & 7) * G + r % G


def test_the_restatement_is_the_headers_formula():
    src = open(os.path.join(ROOT, "latticeurbanwind_amd", "csrc", "luw_device.hpp")).read()
    body = src[src.index("void xcd_row_order("):]
    body = re.sub(r"\s+", "", body[:body.index("\n}\n")])
    assert "if(G&&blockIdx.y<gridDim.y-gridDim.y%(8u*G))" in body
    assert "pp=blockIdx.x+gridDim.x*blockIdx.y,sq=pp>>3,r=sq/gridDim.x;" in body and "biy=(r/G)*8u*G+(pp&7u)*G+r%G;bix=sq%gridDim.x;" in body


@pytest.mark.parametrize("G", [1, 2, 4, 16])
@pytest.mark.parametrize("nbx", [1, 2, 4, 5, 9])
@pytest.mark.parametrize("ny", [1, 7, 8, 31, 32, 33, 64, 100, 174, 256, 742])
def test_permutation_with_one_xcd_per_row(G, nbx, ny):
    full = ny - ny % (8 * G)
    seen, xcd_of_row = set(), {}
    for by in range(ny):
        for bx in range(nbx):
            lx, ly = order(bx, by, nbx, ny, G)
            assert 0 <= lx < nbx and 0 <= ly < ny
            assert (by < full) == (ly < full)                      # the remapped part maps onto itself, the remainder rows keep their blocks
            if by >= full: assert (lx, ly) == (bx, by)
            else: xcd_of_row.setdefault(ly, set()).add((bx + nbx * by) & 7)
            seen.add((lx, ly))
    assert len(seen) == nbx * ny
    assert all(len(x) == 1 for x in xcd_of_row.values()) and len(xcd_of_row) == full
    # G consecutive rows share an XCD, the next G rows have the next one
    for row, x in xcd_of_row.items():
        assert next(iter(x)) == (row // G) % 8


def test_off_is_the_dispatch_order():
    assert all(order(bx, by, 3, 40, 0) == (bx, by) for bx in range(3) for by in range(40))
