#!/usr/bin/env python3
"""Generate the synthetic profile-mode (.luwpf) cases that are fed to the REAL reference binary
(oracle/_ref/FluidX3D, built by oracle/build_ref.sh) on the GPU box, and to this repo's driver.

Everything written here is our own synthetic input (deck text, profile table, a 2-box binary STL);
nothing is copied from /root/reference. Deck grammar follows FX/setup.cpp:61-178,2918-3305
(key = value, // comments, lists in [a, b]).

Cases (all 48x40xNz grids, cell 2 m, wind from 270 deg = +x, VK inlet off, single GPU):
  A  buffer nudging + top sponge ON (thin: 8 m buffer, 8 m sponge -> Nz = 24 + 4)
  B  nudging + sponge OFF                                      (Nz = 24)
  L  "laminar" micro-domain (cell 1e-5 m -> nu_lbm ~ 0.03), nudging + sponge OFF, uniform inflow
  V  case B + von-Karman synthetic-turbulence inlet (turb_inflow_enable = true, L = 20 m, 64 modes)
  M  case A-like profile deck with angle = [200, 45.5, 315]: three runs with ANG_<angle>_ prefixes
  DG dataset mode (*.luwdg), inflow = [3, 5.5] x angle = [0, 225]: four runs with DG_<inflow>_<angle>_ prefixes
  D  case B on a base slab at z = 0..4 m with proj_temp/interpolated_dem.csv (terrain hill): DEM ground plane, terrain clip,
     profile above local terrain, flux correction in profile mode
  T1..T3  N1 / N2 / N3-like decks with a T column in the CSV (temperature boundaries + thermal lattice + T outputs)
  P  case B with probe columns (deck key probes / probes_output): RESULTS/<lon>_<lat>[_offset].csv
  N1..N4  *.luw (NWP) decks on the case-B geometry with synthetic SurfData CSVs: N1 patch-driven 2-D mapping + flux
     correction + Coriolis; N2 KNN-HD (high_order) + flux correction; N3 nearest-sample + nudging + sponge + open
     downstream face; N4 patch mapping with an open downstream face
  C1 BASELINE configs[0] as a deck: 128^3 cells of 2 m, the 12-point profile of the reference's example deck, no buildings (ground slab only), nudging,
     sponge and VK inlet off, 100 steps (outputs at 50 and 100, last 4 averaged)
  G,H case B with a 'city' STL (aligned / off-grid / rotated boxes, roofs, pyramid, floating tetrahedron, overlapping
     boxes) at cell 2 m and 2.5 m: voxeliser goldens
"""
import os, struct, sys

def box_tris(x0, x1, y0, y1, z0, z1):
    v = [(x0,y0,z0),(x1,y0,z0),(x1,y1,z0),(x0,y1,z0),(x0,y0,z1),(x1,y0,z1),(x1,y1,z1),(x0,y1,z1)]
    q = [(0,3,2,1),(4,5,6,7),(0,1,5,4),(2,3,7,6),(1,2,6,5),(3,0,4,7)]  # outward-facing quads
    t = []
    for a,b,c,d in q:
        t.append((v[a],v[b],v[c])); t.append((v[a],v[c],v[d]))
    return t

def write_stl(path, tris):
    with open(path, "wb") as f:
        f.write(b"luw-mi355x synthetic case".ljust(80, b" "))
        f.write(struct.pack("<I", len(tris)))
        for p0,p1,p2 in tris:
            ux,uy,uz = (p1[0]-p0[0],p1[1]-p0[1],p1[2]-p0[2]); vx,vy,vz = (p2[0]-p0[0],p2[1]-p0[1],p2[2]-p0[2])
            n = (uy*vz-uz*vy, uz*vx-ux*vz, ux*vy-uy*vx)
            l = (n[0]**2+n[1]**2+n[2]**2)**0.5 or 1.0
            f.write(struct.pack("<12fH", n[0]/l,n[1]/l,n[2]/l, *p0, *p1, *p2, 0))

def rot_box_tris(cx, cy, lx, ly, z0, z1, deg):
    """box rotated about the vertical axis through (cx, cy)"""
    import math
    c, sn = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    out = []
    for tri in box_tris(-lx/2, lx/2, -ly/2, ly/2, z0, z1):
        out.append(tuple((cx + c*x - sn*y, cy + sn*x + c*y, z) for x, y, z in tri))
    return out

def hull_tris(base, apexes):
    """closed solid over a convex base polygon (counter-clockwise, z = base z) with a roof: apexes = 1 point (pyramid)
    or 2 points (ridge of a pitched roof over a 4-corner base)"""
    t = []
    n = len(base)
    for i in range(1, n-1): t.append((base[0], base[i+1], base[i]))          # floor, facing down
    if len(apexes) == 1:
        for i in range(n): t.append((base[i], base[(i+1) % n], apexes[0]))
    else:
        a, b = apexes                                                        # ridge a (near base[0],base[3]) .. b (near base[1],base[2])
        t += [(base[0], base[1], b), (base[0], b, a), (base[2], base[3], a), (base[2], a, b), (base[1], base[2], b), (base[3], base[0], a)]
    return t

def city_tris(s):
    """a small 'city' exercising the voxeliser: off-grid box, box with faces exactly ON lattice planes, box on half-cell
    planes, rotated box, pitched-roof house, pyramid, floating tetrahedron, two overlapping boxes"""
    t  = box_tris(10.3*s, 22.7*s, 8.6*s, 21.4*s, 0.0, 13.3*s)
    t += box_tris(30*s, 40*s, 10*s, 20*s, 0.0, 12*s)
    t += box_tris(47*s, 55*s, 9*s, 17*s, 0.0, 9*s)
    t += rot_box_tris(70*s, 18*s, 14.2*s, 9.1*s, 0.0, 17.7*s, 31.0)
    hb = [(12*s, 40*s, 0.0), (30*s, 40*s, 0.0), (30*s, 52*s, 0.0), (12*s, 52*s, 0.0)]
    t += box_tris(12*s, 30*s, 40*s, 52*s, 0.0, 8*s)
    t += hull_tris([(x, y, 8*s) for x, y, _ in hb], [(12*s, 46*s, 15.5*s), (30*s, 46*s, 15.5*s)])
    t += hull_tris([(42.2*s, 40.4*s, 0.0), (58.9*s, 41.1*s, 0.0), (57.3*s, 58.2*s, 0.0), (41.0*s, 56.6*s, 0.0)], [(50.1*s, 49.3*s, 21.4*s)])
    t += hull_tris([(68.3*s, 44.1*s, 22.2*s), (80.6*s, 46.3*s, 24.1*s), (73.4*s, 57.7*s, 23.3*s)], [(74.2*s, 49.9*s, 33.8*s)])
    t += box_tris(66.5*s, 78.5*s, 60.2*s, 70.9*s, 0.0, 11.1*s)
    t += box_tris(74.1*s, 86.3*s, 64.8*s, 75.6*s, 0.0, 15.2*s)
    return t

# BASELINE configs[0]: the 12-point profile of the reference's example_ProfileResearch_noDEM deck (height above ground in m, speed in m/s: data)
C1_PROFILE = [(1.25, 2.847), (2.5, 3.042), (5, 3.2604), (7.5, 3.4086), (12.5, 3.7674), (25, 4.3602), (50, 5.109), (75, 5.694), (100, 6.162), (150, 6.9654),
    (200, 7.3944), (250, 7.838)]

def write_case(root, name, s, extra, dims=(96, 80, 48), building=True, nstep=64, unsteady=8, purge=4, vk=False, cell=2.0, z0=0.0, dem=False, angles="[270]",
        profile=None):
    """s = length scale (metres per 'unit'); the box geometry is dims units (default 96 x 80 x 48), base slab 4 units."""
    d = os.path.join(root, name)
    os.makedirs(os.path.join(d, "proj_temp"), exist_ok=True)
    os.makedirs(os.path.join(d, "wind_bc"), exist_ok=True)
    tris  = box_tris(0, dims[0]*s, 0, dims[1]*s, z0-4*s, z0)         # ground slab (full footprint)
    if building == "city":
        tris += city_tris(s)
    elif building:
        tris += box_tris(30.3*s, 46.7*s, 28.6*s, 51.4*s, z0, z0+19.3*s)  # one building, off-grid faces
    if dem:   # proj_temp/interpolated_dem.csv: terrain elevation above the base slab, a smooth hill off-centre (own DEM frame: shifted + scaled xy)
        import math
        with open(os.path.join(d, "proj_temp", "interpolated_dem.csv"), "w") as f:
            f.write("x,y,elevation\n")
            for j in range(17):
                for i in range(21):
                    x, y = dims[0]*s*i/20.0, dims[1]*s*j/16.0
                    e = 5.2*s*math.exp(-((x-22.0*s)**2+(y-52.0*s)**2)/(260.0*s*s)) + 0.8*s*math.sin(0.05*x/s)**2
                    f.write("%.4f,%.4f,%.5f\n" % (1000.0+x, 500.0+y, e))
    write_stl(os.path.join(d, "proj_temp", name + "_PF.stl"), tris)
    with open(os.path.join(d, "wind_bc", "profile.dat"), "w") as f:
        f.write("z,U\n")
        for z, u in (profile or [(1,1.9),(2,2.4),(4,2.9),(8,3.4),(12,3.8),(20,4.3),(30,4.7),(44,5.0),(60,5.0)]):
            f.write("%g\t%g\n" % ((z*s if s >= 1 else z) if profile is None else z, u))
    deck = [
        "// LUW deck (synthetic, generated by tests/golden/make_refcases.py)",
        "casename = %s" % name,
        "datetime = 20260101120000",
        "si_x_cfd = [0.000000, %.9g]" % (dims[0]*s),
        "si_y_cfd = [0.000000, %.9g]" % (dims[1]*s),
        "si_z_cfd = [0.000000, %.9g]" % (dims[2]*s),
        "base_height = %.9g" % (4*s),
        "n_gpu = [1, 1, 1]",
        'mesh_control = "cell_size"',
        "cell_size = %.9g" % (cell*s),
        "validation = pass",
        "high_order = false",
        "flux_correction = false",
        "coriolis_term = false",
        "turb_inflow_enable = %s" % ("true" if vk else "false"),
        "run_nstep = %d" % nstep,
    ] + (["unsteady_output = %d" % unsteady] if unsteady else []) + (["purge_avg = %d" % purge] if purge else []) + [
        "angle = %s" % angles,
    ] + extra
    with open(os.path.join(d, "conf.luwpf"), "w") as f:
        f.write("\n".join(deck) + "\n")

def write_dataset_case(root, name):
    d = os.path.join(root, name)
    os.makedirs(os.path.join(d, "proj_temp"), exist_ok=True)
    tris = box_tris(0, 96.0, 0, 80.0, -4.0, 0.0) + box_tris(30.3, 46.7, 28.6, 51.4, 0.0, 19.3)
    write_stl(os.path.join(d, "proj_temp", name + "_DG.stl"), tris)
    deck = ["// LUW dataset-generation deck (synthetic)", "casename = %s" % name, "datetime = 20260101120000",
            "si_x_cfd = [0.000000, 96]", "si_y_cfd = [0.000000, 80]", "si_z_cfd = [0.000000, 48]", "base_height = 4", "n_gpu = [1, 1, 1]",
            'mesh_control = "cell_size"', "cell_size = 2", "high_order = true", "flux_correction = false", "coriolis_term = false", "turb_inflow_enable = false",
            "enable_buffer_nudging = true", "buffer_thickness_m = 8", "buffer_tau_s = 3", "enable_top_sponge = false",
            "run_nstep = 16", "unsteady_output = 8", "purge_avg = 4", "inflow = [3, 5.5]", "angle = [0, 225]"]
    with open(os.path.join(d, "conf.luwdg"), "w") as f:
        f.write("\n".join(deck) + "\n")

def wind(x, y, z):
    """smooth synthetic NWP-like field (SI): veering log-ish profile with horizontal variation and a weak vertical component"""
    import math
    zz = max(z - 4.0, 0.05)
    speed = 1.2 * math.log(1.0 + zz / 0.5) * (1.0 + 0.06 * math.sin(0.05 * x) + 0.04 * math.cos(0.07 * y))
    th = math.radians(12.0 + 0.25 * z + 2.0 * math.sin(0.03 * y))
    return speed * math.cos(th), speed * math.sin(th), 0.03 * math.sin(0.04 * x + 0.06 * y) * min(zz / 20.0, 1.0)

def air_temperature(x, y, z):
    """synthetic potential-temperature-like field in Kelvin: warm near the ground, horizontal variation"""
    import math
    return 287.5 + 4.0 * math.exp(-max(z - 4.0, 0.0) / 25.0) + 1.2 * math.sin(0.04 * x) * math.cos(0.03 * y)

def write_luw_case(root, name, kind, extra, nstep=16, unsteady=8, purge=4, with_T=False):
    """*.luw (NWP) case on the CaseB geometry: boundaries from proj_temp/SurfData_<datetime>.csv.
    kind = "patch" (X,Y,Z,u,v,w,patch: patch-driven 2-D mapping incl. a bottom patch with gentle terrain),
           "cloud" (X,Y,Z,u,v,w: nearest-sample or, with high_order = true, KNN-HD interpolation)"""
    import math
    d = os.path.join(root, name)
    os.makedirs(os.path.join(d, "proj_temp"), exist_ok=True)
    dims = (96, 80, 48); cell = 2.0
    tris = box_tris(0, dims[0], 0, dims[1], -4.0, 0.0) + box_tris(30.3, 46.7, 28.6, 51.4, 0.0, 19.3)
    write_stl(os.path.join(d, "proj_temp", name + "_DG.stl"), tris)
    Lx, Ly, Lz = (dims[0] - cell), (dims[1] - cell), (dims[2] - cell)     # cell centres span [0, (N-1)*cell] in SI sample coordinates
    def frange(a, b, h):
        n = int(round((b - a) / h)); return [a + (b - a) * i / n for i in range(n + 1)]
    rows = []
    xs, ys = frange(0.0, Lx, 3.1), frange(0.0, Ly, 2.9)
    def terrain(x, y): return 4.4 + 0.9 * math.exp(-((x - 20.0) ** 2 + (y - 60.0) ** 2) / 300.0)
    zs_side = lambda zg: [zg + (Lz - zg) * (i / 17.0) ** 1.3 for i in range(18)]    # stretched, starts at the terrain
    if kind == "patch":
        for x in xs:
            for y in ys: rows.append((x, y, terrain(x, y), 0.0, 0.0, 0.0, 0))
        for x in xs:
            for y in ys: rows.append((x, y, Lz) + wind(x, y, Lz) + (1,))
        for x in xs:
            for yy, pid in ((0.0, 2), (Ly, 3)):
                for z in zs_side(terrain(x, yy)): rows.append((x, yy, z) + wind(x, yy, z) + (pid,))
        for y in ys:
            for xx, pid in ((0.0, 4), (Lx, 5)):
                for z in zs_side(terrain(xx, y)): rows.append((xx, y, z) + wind(xx, y, z) + (pid,))
        header = "X,Y,Z,u,v,w,patch"
        if with_T:   # X,Y,Z,u,v,w,T,patch: ground temperature on the bottom patch, air temperature elsewhere
            rows = [r[:6] + ((291.0 + 2.5 * math.exp(-((r[0] - 60.0) ** 2 + (r[1] - 30.0) ** 2) / 500.0)) if r[6] == 0 else air_temperature(r[0], r[1], r[2]),) + (r[6],) for r in rows]
            header = "X,Y,Z,u,v,w,T,patch"
    else:
        zs = frange(0.0, Lz, 2.7)
        for x in xs:
            for y in ys: rows.append((x, y, Lz) + wind(x, y, Lz))
        for x in xs:
            for yy in (0.0, Ly):
                for z in zs: rows.append((x, yy, z) + wind(x, yy, z))
        for y in ys:
            for xx in (0.0, Lx):
                for z in zs: rows.append((xx, y, z) + wind(xx, y, z))
        header = "X,Y,Z,u,v,w"
        if with_T:
            rows = [r + (air_temperature(r[0], r[1], r[2]),) for r in rows]
            header = "X,Y,Z,u,v,w,T"
    with open(os.path.join(d, "proj_temp", "SurfData_20260101120000.csv"), "w") as f:
        f.write(header + "\n")
        for r in rows:
            last_is_patch = header.endswith("patch")
            f.write(",".join(("%d" % v) if (last_is_patch and i == len(r) - 1) else ("%.6f" % v) for i, v in enumerate(r)) + "\n")
    deck = [
        "// LUW deck (synthetic NWP-mode case, generated by tests/golden/make_refcases.py)",
        "casename = %s" % name, "datetime = 20260101120000",
        "cut_lon_manual = [121.30, 121.60]", "cut_lat_manual = [31.10, 31.40]",
        "si_x_cfd = [0.000000, %.9g]" % dims[0], "si_y_cfd = [0.000000, %.9g]" % dims[1], "si_z_cfd = [0.000000, %.9g]" % dims[2],
        "base_height = 4", "n_gpu = [1, 1, 1]", 'mesh_control = "cell_size"', "cell_size = %.9g" % cell,
        "validation = pass", 'downstream_bc = "+x"', "downstream_bc_yaw = 0", "turb_inflow_enable = false",
        "run_nstep = %d" % nstep, "unsteady_output = %d" % unsteady, "purge_avg = %d" % purge,
    ] + extra
    with open(os.path.join(d, "conf.luw"), "w") as f:
        f.write("\n".join(deck) + "\n")

if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(os.path.dirname(os.path.abspath(__file__)), "refcases")
    write_case(root, "CaseA", 1.0, ["enable_buffer_nudging = true", "buffer_thickness_m = 8", "buffer_tau_s = 3",
                                    "enable_top_sponge = true", "sponge_thickness_m = 8", "sponge_tau_s = 2"])
    write_case(root, "CaseB", 1.0, ["enable_buffer_nudging = false", "enable_top_sponge = false"])
    write_case(root, "CaseL", 1.0e-5, ["enable_buffer_nudging = false", "enable_top_sponge = false"])
    # V: case B with the von-Karman synthetic-turbulence inlet on (defaults TI = 5 %, seed 100; L = 20 m, 64 modes)
    write_case(root, "CaseV", 1.0, ["enable_buffer_nudging = false", "enable_top_sponge = false", "vk_inlet_l = 20", "vk_inlet_nmodes = 64"], vk=True)
    off = ["enable_buffer_nudging = false", "enable_top_sponge = false"]
    # G, H: voxeliser cases ("city" geometry) at cell 2 m (mesh scale 0.5, exact) and cell 2.5 m (scale 0.4, inexact)
    write_case(root, "CaseG", 1.0, off, building="city", nstep=16)
    write_case(root, "CaseH", 1.0, off, building="city", nstep=16, cell=2.5)
    # M: profile mode with several oblique angles (ANG_<angle>_ prefixes, downstream face by dominant axis, nudging on)
    write_case(root, "CaseM", 1.0, ["enable_buffer_nudging = true", "buffer_thickness_m = 8", "buffer_tau_s = 3", "enable_top_sponge = false"], nstep=16, angles="[200, 45.5, 315]")
    # DG: dataset mode (*.luwdg): uniform inflow for every (inflow, angle) pair, DG_<u>_<angle>_ output prefixes
    write_dataset_case(root, "CaseDG")
    # D: profile mode with a DEM ground plane (interpolated_dem.csv), STL base slab on z = 0..4 as luwvox writes it
    write_case(root, "CaseD", 1.0, off + ["flux_correction = true"], nstep=16, z0=4.0, dem=True)
    # P: probes (centre, grid-cell and metre offsets, lon:lat) sampled over the last 6 steps
    write_case(root, "CaseP", 1.0, off + ["cut_lon_manual = [121.4000, 121.4010]", "cut_lat_manual = [31.2000, 31.2007]",
                                          'probes = [center, "centre" NNE, center S10.5W20, 121.40031:31.20022, 121.40085:31.20051 E4, 125.0:31.2]', "probes_output = 6"], nstep=16)
    # N1..N4: *.luw (NWP) mode, boundaries from a synthetic SurfData CSV
    write_luw_case(root, "CaseN1", "patch", off + ["high_order = false", "flux_correction = true", "coriolis_term = true"])
    write_luw_case(root, "CaseN2", "cloud", off + ["high_order = true", "flux_correction = true", "coriolis_term = false"])
    write_luw_case(root, "CaseN3", "cloud", ["enable_buffer_nudging = true", "buffer_thickness_m = 8", "buffer_tau_s = 3", "enable_top_sponge = true",
                                             "sponge_thickness_m = 8", "sponge_tau_s = 2", "high_order = false", "flux_correction = false",
                                             "coriolis_term = false", "downstream_open_face = true"])
    write_luw_case(root, "CaseN4", "patch", off + ["high_order = true", "flux_correction = false", "coriolis_term = false", "downstream_open_face = true"])
    # T1..T3: NWP decks whose CSV carries a T column (buoyancy defaults to true): temperature boundaries, thermal lattice, T outputs
    write_luw_case(root, "CaseT1", "patch", off + ["high_order = false", "flux_correction = true", "coriolis_term = false"], with_T=True)
    write_luw_case(root, "CaseT2", "cloud", off + ["high_order = true", "flux_correction = false", "coriolis_term = false"], with_T=True)
    write_luw_case(root, "CaseT3", "cloud", ["enable_buffer_nudging = false", "enable_top_sponge = true", "sponge_thickness_m = 8", "sponge_tau_s = 2", "high_order = false",
                                             "flux_correction = false", "coriolis_term = false", "downstream_open_face = true"], with_T=True)
    # C1: BASELINE configs[0] as a deck (128^3 cells of 2 m, the reference's example profile, no buildings): the size-class fixture of tests/test_gpu_c1.py
    write_case(root, "CaseC1", 1.0, off, dims=(256, 256, 256), building=False, nstep=100, unsteady=50, purge=4, profile=C1_PROFILE)
    # performance decks for timing the reference itself on the GPU box (not fixtures: generated on demand)
    if "--perf" in sys.argv:
        write_case(root, "Perf512", 1.0, off, dims=(1024, 1024, 1024), building=False, nstep=300, unsteady=0, purge=0)
        write_case(root, "Perf1024x1024x256", 1.0, off, dims=(2048, 2048, 512), building=False, nstep=200, unsteady=0, purge=0)
    print("wrote cases under", root)
