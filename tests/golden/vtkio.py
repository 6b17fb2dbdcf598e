"""Minimal reader for the legacy-VTK BINARY STRUCTURED_POINTS files the reference writes
(FX/lbm.hpp:307-356 and write_avg_vtk, FX/setup.cpp:2513-2683): big-endian, AoS per point."""
import numpy as np


def read_vtk(path):
    data = open(path, "rb").read()
    pos = 0
    def line():
        nonlocal pos
        e = data.index(b"\n", pos); s = data[pos:e].decode("ascii", "replace"); pos = e + 1
        return s
    hdr = {}
    fields = {}
    line(); line(); assert line() == "BINARY"; line()
    while pos < len(data):
        l = line()
        if not l:
            continue
        tok = l.split()
        if tok[0] == "DIMENSIONS": hdr["dims"] = tuple(int(t) for t in tok[1:4])
        elif tok[0] == "ORIGIN": hdr["origin"] = tuple(float(t) for t in tok[1:4])
        elif tok[0] == "SPACING": hdr["spacing"] = tuple(float(t) for t in tok[1:4])
        elif tok[0] == "POINT_DATA": hdr["points"] = int(tok[1])
        elif tok[0] == "SCALARS":
            name, typ, comp = tok[1], tok[2], int(tok[3]) if len(tok) > 3 else 1
            assert line().startswith("LOOKUP_TABLE")
            dt = {"float": ">f4", "unsigned_char": "u1", "double": ">f8"}[typ]
            n = hdr["points"] * comp
            arr = np.frombuffer(data, dtype=dt, count=n, offset=pos)
            pos += n * np.dtype(dt).itemsize
            nx, ny, nz = hdr["dims"]
            fields[name] = arr.reshape(nz, ny, nx, comp).astype(arr.dtype.newbyteorder("="))
    return hdr, fields
