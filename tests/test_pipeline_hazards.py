"""Slot algebra behind the pipelined multi-GPU schedule (latticeurbanwind_amd/distributed.py, DomainDecomposedLBM.run): the interior
kernel of step t+1 may run while the halo pack / exchange / unpack of step t is still in flight, because the two never touch the
same (plane, cell) slot.  Checked here by brute force on small boxes from the index maps alone:
  * step kernel: Esoteric-Pull load_f/store_f, FX/kernel.cpp:1338-1351 (in place: the slots read are the slots written),
  * pack/unpack: transfer_extract_fi / transfer__insert_fi, FX/kernel.cpp:2223-2258 (restated in csrc/luw_kernels_aux.hpp),
  * thermal lattice: the first seven directions with the same algebra, FX/kernel.cpp:1322-1335,2338-2351.
Pure index arithmetic, no GPU and no library."""
import itertools

import pytest

from latticeurbanwind_amd.distributed import DomainLayout

C = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (-1, -1, 0), (1, 0, 1), (-1, 0, -1),
     (0, 1, 1), (0, -1, -1), (1, -1, 0), (-1, 1, 0), (1, 0, -1), (-1, 0, 1), (0, 1, -1), (0, -1, 1)]
FACE_SETS = {(0, 0): (1, 7, 13, 9, 15), (0, 1): (2, 8, 14, 10, 16), (1, 0): (3, 7, 14, 11, 17), (1, 1): (4, 8, 13, 12, 18),
             (2, 0): (5, 9, 16, 11, 18), (2, 1): (6, 10, 15, 12, 17)}


def neighbor(cell, i, lN):
    return tuple((cell[a] + C[i][a]) % lN[a] for a in range(3))


def step_slots(cell, t, lN, q):
    """slots (plane, cell) one cell's collide-stream touches at time t (reads == writes)"""
    out = {(0, cell)}
    for i in range(1, q, 2):
        a, b = (i, i + 1) if t % 2 else (i + 1, i)
        out.add((a, cell)); out.add((b, neighbor(cell, i, lN)))
    return out


def box_cells(box):
    return itertools.product(range(box[0], box[1]), range(box[2], box[3]), range(box[4], box[5]))


def face_cells(axis, layer, lN):
    rng = [range(n) for n in lN]
    rng[axis] = [layer]
    return itertools.product(*rng)


def transfer_slots(axis, t, lN, thermal):
    """(slots the pack kernel reads, slots the unpack kernel writes) for one axis at time t"""
    odd = t % 2
    reads, writes = set(), set()
    for pm in (0, 1):
        dirs = (2 * axis + pm + 1,) if thermal else FACE_SETS[(axis, pm)]
        for cell in face_cells(axis, lN[axis] - 2 if pm == 0 else 1, lN):
            for i in dirs:
                plane = (i + 1 if i % 2 else i - 1) if odd else i
                reads.add((plane, neighbor(cell, i, lN) if i % 2 else cell))
        for cell in face_cells(axis, lN[axis] - 1 if pm == 0 else 0, lN):
            for i in dirs:
                plane = i if odd else (i + 1 if i % 2 else i - 1)
                writes.add((plane, cell if i % 2 else neighbor(cell, i - 1, lN)))
    return reads, writes


def edge_slots(e, t, lN):
    """(slots the edge pack reads, slots the edge unpack writes) of edge message e = population 7 + e (csrc/luw_kernels_aux.hpp k_edges): an odd population
    leaves in slot B of the halo-halo line beyond the sender's corner and lands in slot B of the receiver's owned corner line; an even one leaves in slot A
    of the owned corner line and lands in the halo-halo line"""
    i = 7 + e
    odd_pop, io = i % 2 == 1, i if i % 2 else i - 1
    plane = (io + 1 if t % 2 else io) if odd_pop else (io if t % 2 else io + 1)
    def line(sender):
        rng = []
        for a in range(3):
            c, n = C[i][a], lN[a]
            if c == 0: rng.append(range(n))
            elif odd_pop: rng.append([(n - 1 if c > 0 else 0) if sender else (1 if c > 0 else n - 2)])
            else: rng.append([(n - 2 if c > 0 else 1) if sender else (0 if c > 0 else n - 1)])
        return {(plane, cell) for cell in itertools.product(*rng)}
    return line(True), line(False)


@pytest.mark.parametrize("D,gN", [((1, 2, 2), (6, 12, 10)), ((2, 2, 1), (20, 12, 4)), ((2, 2, 2), (16, 12, 12))])
def test_edge_messages_of_the_one_phase_exchange_keep_clear_of_the_interior(monkeypatch, D, gN):
    """the edge pack / unpack of step t against the interior of step t+1 (disjoint: pipelining holds for the one-phase exchange too); and every slot an edge
    message delivers is one the rims of the three-phase route's face inserts write as well -- the edge insert, coming last, replaces what those left there"""
    monkeypatch.setattr(DomainLayout, "X_SHELL", 2)
    lay = DomainLayout(gN, D, 0)
    lN = tuple(lay.lN)
    assert len(lay.edges()) == {2: 4, 3: 12}[len(lay.split_axes())]
    for t in (0, 1):
        interior = set()
        for cell in box_cells(lay.interior_box()):
            interior |= step_slots(cell, t + 1, lN, 19)
        face_writes = set()
        for axis in lay.split_axes():
            face_writes |= transfer_slots(axis, t, lN, False)[1]
        for e in lay.edges():
            reads, writes = edge_slots(e, t, lN)
            assert not (interior & reads) and not (interior & writes)
            assert writes <= face_writes


@pytest.mark.parametrize("thermal", [False, True])
@pytest.mark.parametrize("D,gN", [((1, 2, 2), (6, 12, 10)), ((1, 2, 1), (5, 14, 6)), ((1, 1, 2), (6, 6, 12)), ((1, 3, 2), (4, 18, 12))])
def test_interior_of_next_step_is_disjoint_from_the_exchange(D, gN, thermal):
    lay = DomainLayout(gN, D, 0)
    lN = tuple(lay.lN)
    q = 7 if thermal else 19
    for t in (0, 1):
        interior = set()
        for cell in box_cells(lay.interior_box()):
            interior |= step_slots(cell, t + 1, lN, q)
        shell = set()
        for box in lay.shell_boxes():
            for cell in box_cells(box):
                shell |= step_slots(cell, t + 1, lN, q)
        assert not (interior & shell)                     # what already lets shell and interior of ONE step overlap
        touched_by_exchange = set()
        for axis in lay.split_axes():
            reads, writes = transfer_slots(axis, t, lN, thermal)
            assert not (interior & reads), "interior(t+1) overwrites a slot the pack kernel of step t still reads"
            assert not (interior & writes), "interior(t+1) touches a slot the unpack kernel of step t writes"
            touched_by_exchange |= reads | writes
        # the shell of step t+1 does depend on the exchange of step t (that is what the halo is for): keep that ordering
        assert shell & touched_by_exchange


@pytest.mark.parametrize("x_shell", [1, 2])
@pytest.mark.parametrize("D,gN", [((2, 1, 1), (16, 5, 4)), ((2, 2, 1), (20, 12, 4)), ((2, 2, 2), (16, 12, 12))])
def test_x_split_layouts_are_disjoint_too(monkeypatch, D, gN, x_shell):
    # x split (a deck's literal n_gpu = [4, 2, 1]): the same property with x boundary slabs of any thickness
    monkeypatch.setattr(DomainLayout, "X_SHELL", x_shell)
    lay = DomainLayout(gN, D, 1)
    lN = tuple(lay.lN)
    assert lay.interior_box()[1] > lay.interior_box()[0]
    for t in (0, 1):
        interior = set()
        for cell in box_cells(lay.interior_box()):
            interior |= step_slots(cell, t + 1, lN, 19)
        for axis in lay.split_axes():
            reads, writes = transfer_slots(axis, t, lN, False)
            assert not (interior & reads) and not (interior & writes)


def test_shell_and_interior_tile_the_owned_cells():
    lay = DomainLayout((6, 12, 10), (1, 2, 2), 3)
    cells = list(box_cells(lay.interior_box()))
    for box in lay.shell_boxes():
        cells += list(box_cells(box))
    assert len(cells) == len(set(cells))
    owned = set(itertools.product(*[range(h, n - h) for h, n in zip(lay.H, lay.lN)]))
    assert set(cells) == owned
