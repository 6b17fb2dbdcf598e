"""Pure host logic of a block-decomposed lattice: the D3Q19 directions, which decomposition a world of N ranks takes, where a rank sits, what it owns and whom
it talks to (FX/lbm.cpp:1057-1073,1912-1931).  No GPU, no torch."""
import numpy as np

# D3Q19 directions c_i (FX/kernel.cpp:890-893); edge message e = 0..11 carries population 7 + e to the domain in direction c_(7+e)
C19 = ((0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (-1, -1, 0), (1, 0, 1), (-1, 0, -1), (0, 1, 1), (0, -1, -1),
       (1, -1, 0), (-1, 1, 0), (1, 0, -1), (-1, 0, 1), (0, 1, -1), (0, -1, 1))


def choose_decomposition(world, split_x=False):
    """n_gpu and per-GPU lattice shape factors for a weak-scaled tile of 512^3 cells per GPU.

    split_x=False (default): the memory-fastest axis is kept whole -- rows stay complete memory lines, the y/z boundary
    shells are whole rows and the halo traffic hides behind the interior: 8 GPUs cover the 2048x1024x512 tile of
    BASELINE configs[3] as n_gpu=[1,4,2] (local 2048x256x256: the least halo area among the x-whole grids, and the fastest
    rank step measured, 3.42 ms vs 3.55 ms for [1,2,4]; tools/bench_layouts.py).  split_x=True reproduces the deck's literal
    n_gpu=[4,2,1] (local 512^3); with x split the step runs the whole box first and exchanges afterwards (measured on
    MI355X, one rank with loopback halos: 3.77 ms sequential vs 4.2-5.2 ms with an x shell, vs 3.37 ms undivided).
    Returns (D, global_lattice)."""
    if split_x:
        table = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (4, 2, 1), 16: (4, 4, 1)}
    else:
        table = {1: (1, 1, 1), 2: (1, 2, 1), 4: (1, 2, 2), 8: (1, 4, 2), 16: (1, 4, 4)}
    if world in table:
        return table[world]
    d = [1, 1, 1]
    n, ax = world, 0
    for p in (2, 3, 5, 7):
        while n % p == 0:
            d[(ax % 2) + 1 if not split_x else ax % 3] *= p; ax += 1; n //= p
    if n != 1:
        d[1 if not split_x else 0] *= n
    return tuple(d)


def tile_lattice(world):
    """global lattice of the weak-scaled benchmark tile: 512^3 cells per GPU, growing x, then y, then x again
    (1: 512^3 = BASELINE configs[1]; 2: 1024x512x512; 4: 1024x1024x512; 8: 2048x1024x512 = configs[3])"""
    g = [512, 512, 512]
    n, ax = world, 0
    while n > 1 and n % 2 == 0:
        g[(0, 1)[ax % 2]] *= 2; ax += 1; n //= 2
    g[2] *= n
    return tuple(g)


class DomainLayout:
    """Pure host logic: where a rank sits, what it owns, whom it talks to (FX/lbm.cpp:1066-1073,1912-1931)."""

    X_SHELL = 128  # thickness of the x boundary slabs (see shell_boxes; csrc/luw_group.hpp group_x_shell)

    def __init__(self, global_N, D, rank, x_shell=None):
        if x_shell: self.X_SHELL = int(x_shell)
        self.gN = tuple(int(v) for v in global_N)
        self.D = tuple(int(v) for v in D)
        Dx, Dy, Dz = self.D
        if any(g % d for g, d in zip(self.gN, self.D)):
            raise ValueError("LBM grid %s is not equally divisible in domains %s" % (self.gN, self.D))  # FX/lbm.cpp:1058-1059 shrinks; we refuse
        if not 0 <= rank < Dx * Dy * Dz:
            raise ValueError("rank outside the domain grid")
        self.rank = rank
        self.coord = ((rank % (Dx * Dy)) % Dx, (rank % (Dx * Dy)) // Dx, rank // (Dx * Dy))
        self.H = tuple(int(d > 1) for d in self.D)                         # halo offsets
        self.lN = tuple(g // d + 2 * h for g, d, h in zip(self.gN, self.D, self.H))
        self.O = tuple(c * (g // d) - h for c, g, d, h in zip(self.coord, self.gN, self.D, self.H))

    def rank_of(self, coord):
        x, y, z = coord
        return x + (y + z * self.D[1]) * self.D[0]

    def neighbor(self, axis, sign):
        c = list(self.coord)
        c[axis] = (c[axis] + sign) % self.D[axis]
        return self.rank_of(c)

    def neighbor_dir(self, c):
        """rank of the domain in direction c = (cx, cy, cz), periodic"""
        return self.rank_of(tuple((k + d) % n for k, d, n in zip(self.coord, c, self.D)))

    def split_axes(self):
        return [a for a in range(3) if self.D[a] > 1]

    def edges(self):
        """edge messages this domain takes part in: population 7 + e crosses two cuts, both of them split"""
        return [e for e in range(12) if all(self.D[a] > 1 for a in range(3) if C19[7 + e][a])]

    def edge_length(self, e):
        """cells of edge e's line: the local extent of the axis its population does not move along"""
        return self.lN[[a for a in range(3) if C19[7 + e][a] == 0][0]]

    # ---- boxes (x0,x1,y0,y1,z0,z1) in local coordinates: ONE implementation, the library's (luw_step_boxes, csrc/luw_step.hpp: pure host arithmetic, also
    # what the one-process host luw_group_* cuts its domains with).  whole = the non-halo cells; interior + the disjoint shell slabs cover it exactly once;
    # y, z slabs are the one cell layer next to a halo (whole rows), x slabs whole blocks of X_SHELL cells from the first owned cell on.
    def _boxes(self):
        key = (self.lN, self.H, self.X_SHELL)
        if getattr(self, "_box_key", None) != key:
            import ctypes as C
            from . import capi
            u3 = C.c_uint32 * 3
            whole, inner, shell, n, ok = (C.c_uint32 * 6)(), (C.c_uint32 * 6)(), (C.c_uint32 * 36)(), C.c_uint32(0), C.c_int(0)
            capi.check(capi.load().luw_step_boxes(u3(*self.lN), u3(*self.H), int(self.X_SHELL), whole, inner, shell, C.byref(n), C.byref(ok)))
            self._box_key = key
            self._box_val = (tuple(whole), tuple(inner), [tuple(shell[6 * k:6 * k + 6]) for k in range(n.value)], bool(ok.value))
        return self._box_val

    def whole_box(self): return self._boxes()[0]
    def interior_box(self): return self._boxes()[1]
    def shell_boxes(self): return list(self._boxes()[2])
    def can_overlap(self): return self._boxes()[3]       # every split axis has at least four owned layers

    def local_slices(self):
        """slices of the GLOBAL (z,y,x) array that fill the local box incl. halos (periodic wrap), as index arrays"""
        idx = []
        for a in range(3):
            idx.append((np.arange(self.lN[a]) + self.O[a]) % self.gN[a])
        return idx  # x, y, z index arrays


