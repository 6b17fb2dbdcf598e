"""ctypes binding of the C-ABI declared in include/luw_core.h (the drop-in boundary) and include/luw_core_dev.h (measurement / test entry points) of
the product library csrc/libluw_core.so.

There is deliberately no CPU fallback: if the HIP library is missing or no GPU is present, calls fail loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "csrc", "libluw_core.so")
_LIB = None

LUW_OK = 0
FIELD_RHO, FIELD_U, FIELD_FLAGS, FIELD_F, FIELD_FI, FIELD_T, FIELD_GI = 0, 1, 2, 3, 4, 5, 6
MASK_RHO, MASK_U, MASK_FLAGS, MASK_F, MASK_T = 1, 2, 4, 8, 32
DDF_FP32, DDF_FP16C = 0, 1
OPT_FORCE_FIELD, OPT_UPDATE_FIELDS_EVERY_STEP, OPT_NO_SUBGRID, OPT_TEMPERATURE, OPT_NATIVE_ARITH = 1, 2, 4, 8, 16
KERNEL_AUTO, KERNEL_SCALAR, KERNEL_PAIR = 0, 1, 7         # what the product library knows
# ids of the A/B and measurement-only variants: only the tools build (make -C csrc ab -> tools/libluw_core_ab.so) has them
KERNEL_VEC4, KERNEL_VEC2, KERNEL_SCALAR_CACHED, KERNEL_SCALAR_NT_ALL, KERNEL_VEC1, KERNEL_SCALAR_GENERAL = 2, 3, 4, 5, 6, 8
KERNEL_EXP_COPY, KERNEL_EXP_NOSHIFT = 100, 101
TYPE_S, TYPE_E, TYPE_T = 0x01, 0x02, 0x04
OK, ERR_INVALID, ERR_DEVICE, ERR_STATE, ERR_NOMEM = 0, -1, -2, -3, -4

SYMBOLS = [
    "luw_abi_version", "luw_last_error", "luw_device_count", "luw_format_float9", "luw_create", "luw_destroy", "luw_host_ptr", "luw_get_N",
    "luw_upload", "luw_download", "luw_initialize", "luw_run", "luw_get_t", "luw_set_f", "luw_set_coriolis", "luw_device_ptr", "luw_get_pitch",
    "luw_get_plane_stride", "luw_set_stream", "luw_enqueue_stream_collide", "luw_increment_time_step", "luw_reset_time_step", "luw_get_area",
    "luw_enqueue_extract_fi", "luw_set_x_face_buffers", "luw_set_x_face_inputs", "luw_get_edge_length", "luw_enqueue_extract_edges", "luw_enqueue_insert_edges",
        "luw_group_create",
        "luw_group_destroy", "luw_group_size", "luw_group_domain", "luw_group_domain_info",
    "luw_group_overlaps", "luw_group_direct_peer_stores", "luw_group_one_phase", "luw_group_scatter", "luw_group_gather", "luw_group_upload",
        "luw_group_download",
    "luw_group_initialize", "luw_group_run", "luw_group_run_sampled", "luw_group_get_t", "luw_group_set_f", "luw_group_set_coriolis",
    "luw_group_voxelize_mesh", "luw_group_vk_inlet_attach", "luw_group_gather_attach", "luw_group_gather_u", "luw_group_stats_reset",
    "luw_group_stats_download", "luw_group_transport", "luw_device_info", "luw_p2p_info", "luw_fields_every_step", "luw_step_boxes",
    "luw_domain_step_create", "luw_domain_step_destroy", "luw_domain_step_overlaps", "luw_domain_step_launch", "luw_domain_step_separate_stats",
    "luw_group_export_vtk", "luw_group_stats_count", "luw_enqueue_insert_fi", "luw_enqueue_extract_gi", "luw_enqueue_insert_gi", "luw_finish",
    "luw_stats_reset", "luw_stats_accumulate", "luw_run_sampled", "luw_stats_begin_sample", "luw_stats_download", "luw_stats_download_T",
    "luw_voxelize_mesh", "luw_voxelize_lattice", "luw_set_kernel", "luw_gather_attach", "luw_gather_u", "luw_vk_inlet_attach", "luw_vk_inlet_apply",
    "luw_vk_inlet_detach",
]
# include/luw_core_dev.h
DEV_SYMBOLS = [
    "luw_run_timed", "luw_group_run_timed", "luw_domain_step_timing", "luw_dev_placement_info", "luw_dev_workgroup_order", "luw_dev_reload_tuning",
    "luw_dev_tuning_text", "luw_dev_inject_fault", "luw_dev_schedule_jitter",
    "luw_download_fi", "luw_download_gi", "luw_upload_fi", "luw_selfcheck_fp16c_codec", "luw_selfcheck_arith",
]


class LuwError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("Nx", C.c_uint32), ("Ny", C.c_uint32), ("Nz", C.c_uint32),
        ("Dx", C.c_uint32), ("Dy", C.c_uint32), ("Dz", C.c_uint32),
        ("Ox", C.c_int32), ("Oy", C.c_int32), ("Oz", C.c_int32),
        ("nu", C.c_float),
        ("fx", C.c_float), ("fy", C.c_float), ("fz", C.c_float),
        ("omega_x", C.c_float), ("omega_y", C.c_float), ("omega_z", C.c_float),
        ("ddf_format", C.c_uint32), ("options", C.c_uint32),
        ("buffer_nudging_active", C.c_int32), ("buffer_n_cells", C.c_uint32), ("buffer_inv_tau_lbmu", C.c_float),
        ("buffer_nudge_vertical", C.c_int32), ("buffer_downstream_face_id", C.c_int32),
        ("top_sponge_active", C.c_int32), ("sponge_n_cells", C.c_uint32), ("sponge_inv_tau_lbmu", C.c_float),
        ("alpha", C.c_float), ("device", C.c_int32), ("kernel", C.c_uint32),
    ]


def build(force=False):
    """Compile csrc/luw_core.hip for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot)."""
    src_dir = os.path.join(_HERE, "csrc")
    if force and os.path.exists(_SO):
        os.remove(_SO)
    subprocess.check_call(["make", "-C", src_dir, "-s"])                          # make knows the sources (luw_core.hip + the kernel headers)
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "host"), "-s"])      # the deck driver (C++ host over the C-ABI)
    return _SO


def load(path=None):
    """dlopen the product library and declare the prototypes.  Raises if it has not been built.  `path` (first call only) names
    another build of the SAME sources -- the tools' A/B library -- and is never set by the package itself."""
    global _LIB, _SO
    if _LIB is not None:
        if path is not None and os.path.abspath(path) != os.path.abspath(_SO):
            raise LuwError("another libluw_core build is already loaded in this process")
        return _LIB
    if path is None and os.environ.get("LUW_LIB"):      # A/B scripts: another build of the same sources, named explicitly (never set by the package)
        path = os.environ["LUW_LIB"]
    if path is not None:
        _SO = path
    if not os.path.exists(_SO):
        raise LuwError("HIP library %s is missing: run latticeurbanwind_amd.build() (needs hipcc); there is no CPU fallback" % _SO)
    try:
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64 under the same SONAMEs as /opt/rocm; whichever is
        # loaded first serves the whole process.  Load torch's first so that torch tensors, streams and RCCL share
        # ONE HIP runtime with this library (loading ours first leaves torch without devices).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_SO)
    vp, u64, u32, i32, f32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float
    L.luw_abi_version.restype = i32
    L.luw_last_error.restype = C.c_char_p
    L.luw_device_count.argtypes = [C.POINTER(i32)]
    L.luw_format_float9.argtypes = [f32, C.c_char_p, u64]
    L.luw_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.luw_destroy.argtypes = [vp]; L.luw_destroy.restype = None
    L.luw_host_ptr.argtypes = [vp, i32]; L.luw_host_ptr.restype = vp
    L.luw_device_ptr.argtypes = [vp, i32]; L.luw_device_ptr.restype = vp
    L.luw_get_N.argtypes = [vp]; L.luw_get_N.restype = u64
    L.luw_get_t.argtypes = [vp]; L.luw_get_t.restype = u64
    L.luw_get_pitch.argtypes = [vp]; L.luw_get_pitch.restype = u32
    L.luw_get_plane_stride.argtypes = [vp]; L.luw_get_plane_stride.restype = u64
    L.luw_get_area.argtypes = [vp, u32]; L.luw_get_area.restype = u64
    L.luw_upload.argtypes = [vp, u32]
    L.luw_download.argtypes = [vp, u32]
    L.luw_initialize.argtypes = [vp]
    L.luw_run.argtypes = [vp, u64]
    L.luw_run_timed.argtypes = [vp, u64, C.POINTER(C.c_double)]
    L.luw_set_f.argtypes = [vp, f32, f32, f32]
    L.luw_set_coriolis.argtypes = [vp, f32, f32, f32]
    L.luw_set_stream.argtypes = [vp, vp]
    L.luw_enqueue_stream_collide.argtypes = [vp, u32, u32, u32, u32, u32, u32, i32]
    L.luw_increment_time_step.argtypes = [vp, u64]
    L.luw_reset_time_step.argtypes = [vp]
    L.luw_enqueue_extract_fi.argtypes = [vp, u32, vp, vp]
    L.luw_enqueue_insert_fi.argtypes = [vp, u32, vp, vp]
    L.luw_set_x_face_buffers.argtypes = [vp, vp, vp]
    L.luw_set_x_face_inputs.argtypes = [vp, vp, vp]
    L.luw_get_edge_length.argtypes = [vp, u32]; L.luw_get_edge_length.restype = u64
    L.luw_enqueue_extract_edges.argtypes = [vp, C.POINTER(vp)]; L.luw_enqueue_insert_edges.argtypes = [vp, C.POINTER(vp)]
    L.luw_finish.argtypes = [vp]
    L.luw_download_fi.argtypes = [vp, vp]
    L.luw_upload_fi.argtypes = [vp, vp]
    L.luw_selfcheck_fp16c_codec.argtypes = [i32, C.POINTER(u64)]
    L.luw_selfcheck_arith.argtypes = [i32, C.POINTER(u64)]
    L.luw_voxelize_mesh.argtypes = [vp, u32, vp, vp, vp, vp, C.c_uint8]
    L.luw_download_gi.argtypes = [vp, vp]
    L.luw_enqueue_extract_gi.argtypes = [vp, u32, vp, vp]; L.luw_enqueue_insert_gi.argtypes = [vp, u32, vp, vp]
    L.luw_set_kernel.argtypes = [vp, u32]
    L.luw_gather_attach.argtypes = [vp, u32, vp]; L.luw_gather_u.argtypes = [vp, vp]
    L.luw_voxelize_lattice.argtypes = [i32, u32, u32, u32, u32, vp, vp, vp, vp, C.c_uint8, vp]
    L.luw_vk_inlet_attach.argtypes = [vp, u64, u64, vp, vp, vp, vp, i32, i32]
    L.luw_vk_inlet_apply.argtypes = [vp]
    L.luw_vk_inlet_detach.argtypes = [vp]
    L.luw_stats_reset.argtypes = [vp]
    L.luw_stats_accumulate.argtypes = [vp]
    L.luw_run_sampled.argtypes = [vp, u64, u64, u64]
    L.luw_stats_begin_sample.argtypes = [vp, C.POINTER(i32)]
    L.luw_stats_download.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(u64)]
    L.luw_stats_download_T.argtypes = [vp, vp]
    i32p, u32p = C.POINTER(i32), C.POINTER(u32)
    L.luw_group_create.argtypes = [C.POINTER(Config), i32p, C.POINTER(vp)]
    L.luw_group_destroy.argtypes = [vp]; L.luw_group_destroy.restype = None
    L.luw_group_size.argtypes = [vp]; L.luw_group_size.restype = u32
    L.luw_group_domain.argtypes = [vp, u32]; L.luw_group_domain.restype = vp
    L.luw_group_domain_info.argtypes = [vp, u32, u32p, C.POINTER(C.c_int32), i32p]
    L.luw_group_overlaps.argtypes = [vp]; L.luw_group_direct_peer_stores.argtypes = [vp]; L.luw_group_one_phase.argtypes = [vp]
    L.luw_group_scatter.argtypes = [vp, i32, vp]; L.luw_group_gather.argtypes = [vp, i32, vp]
    L.luw_group_upload.argtypes = [vp, u32]; L.luw_group_download.argtypes = [vp, u32]
    L.luw_group_initialize.argtypes = [vp]
    L.luw_group_run.argtypes = [vp, u64]; L.luw_group_run_sampled.argtypes = [vp, u64, u64, u64]
    L.luw_group_run_timed.argtypes = [vp, u64, C.POINTER(C.c_double)]
    L.luw_group_get_t.argtypes = [vp]; L.luw_group_get_t.restype = u64
    L.luw_group_set_f.argtypes = [vp, f32, f32, f32]; L.luw_group_set_coriolis.argtypes = [vp, f32, f32, f32]
    L.luw_group_voxelize_mesh.argtypes = [vp, u32, vp, vp, vp, vp, C.c_uint8]
    L.luw_group_vk_inlet_attach.argtypes = [vp, u64, u64, vp, vp, vp, vp, i32, i32]
    L.luw_group_gather_attach.argtypes = [vp, u32, vp]; L.luw_group_gather_u.argtypes = [vp, vp]
    L.luw_group_stats_reset.argtypes = [vp]
    L.luw_group_stats_download.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(u64)]
    L.luw_group_transport.argtypes = [vp]
    L.luw_step_boxes.argtypes = [u32p, u32p, u32, u32p, u32p, u32p, u32p, i32p]
    L.luw_domain_step_create.argtypes = [vp, vp, vp, u32, i32, C.POINTER(vp)]
    L.luw_domain_step_destroy.argtypes = [vp]; L.luw_domain_step_destroy.restype = None
    L.luw_domain_step_overlaps.argtypes = [vp]
    L.luw_domain_step_launch.argtypes = [vp, i32, i32]
    L.luw_domain_step_separate_stats.argtypes = [vp]
    L.luw_domain_step_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.luw_fields_every_step.argtypes = [vp]
    L.luw_group_stats_count.argtypes = [vp]; L.luw_group_stats_count.restype = u64
    L.luw_group_export_vtk.argtypes = [vp, i32, vp, u32, i32, u64]
    L.luw_device_info.argtypes = [i32, C.c_char_p, u64, C.c_char_p, u64, C.POINTER(u64)]
    L.luw_p2p_info.argtypes = [i32, i32, i32p, i32p, i32p, u32p, u32p]
    L.luw_dev_placement_info.argtypes = [vp, i32p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p, u64]
    L.luw_dev_workgroup_order.argtypes = [vp]
    L.luw_dev_tuning_text.argtypes = [C.c_char_p, u64]
    L.luw_dev_inject_fault.argtypes = [u32]
    L.luw_dev_schedule_jitter.argtypes = [u64, u32]
    if L.luw_abi_version() != 6:
        raise LuwError("libluw_core.so ABI version mismatch")
    _LIB = L
    return L


TRANSPORT_NAMES = {0: "peer stores", 1: "staged (hipMemcpyPeerAsync)", 2: "rccl (grouped ncclSend/ncclRecv)"}
LINK_TYPES = {0: "HyperTransport", 1: "QPI", 2: "PCIe", 3: "InfiniBand", 4: "xGMI"}     # hsa_amd_link_info_type_t


def device_info(device):
    """name, PCI bus id and memory of a HIP device (luw_device_info)"""
    name, pci, mem = C.create_string_buffer(256), C.create_string_buffer(64), C.c_uint64(0)
    check(load().luw_device_info(int(device), name, 256, pci, 64, C.byref(mem)))
    return {"device": int(device), "name": name.value.decode(), "pci_bus_id": pci.value.decode(), "memory_GB": round(mem.value / 1e9, 1)}


def p2p_info(device, peer):
    """the link from `device` to `peer` as the HIP runtime reports it (luw_p2p_info)"""
    acc, rank, atom, lt, hops = C.c_int(0), C.c_int(0), C.c_int(0), C.c_uint32(0), C.c_uint32(0)
    check(load().luw_p2p_info(int(device), int(peer), C.byref(acc), C.byref(rank), C.byref(atom), C.byref(lt), C.byref(hops)))
    return {"peer": int(peer), "can_access": bool(acc.value), "performance_rank": rank.value, "native_atomics": atom.value,
            "link": LINK_TYPES.get(lt.value, "type %d" % lt.value) if lt.value != 0xFFFFFFFF else None, "hops": None if hops.value == 0xFFFFFFFF
                else hops.value}


def check(rc):
    if rc != LUW_OK:
        raise LuwError("luw_core error %d: %s" % (rc, load().luw_last_error().decode("utf-8", "replace")))


def reload_tuning():
    """the library reads its environment knobs once; tests and A/B tools that change os.environ between two solvers call this"""
    check(load().luw_dev_reload_tuning())


def tuning_text():
    buf = C.create_string_buffer(1024)
    check(load().luw_dev_tuning_text(buf, 1024))
    return buf.value.decode()


def placement_info(handle):
    """what luw_create's placement search did for a solver (luw_dev_placement_info)"""
    n, tb, sec, kept = C.c_int(0), C.c_double(0.0), C.c_double(0.0), C.create_string_buffer(64)
    check(load().luw_dev_placement_info(handle, C.byref(n), C.byref(tb), C.byref(sec), kept, 64))
    return {"candidates_tried": n.value, "kept": kept.value.decode(), "probe_TBps": round(tb.value, 3), "create_s": round(sec.value, 3),
        "rows_per_xcd": int(load().luw_dev_workgroup_order(handle))}


FAULT_NO_PEER_ODD_PAIRS, FAULT_RCCL_INIT, FAULT_SLOW_FIRST_PLACEMENT = 1, 2, 4


def schedule_jitter(seed, max_us):
    """luw_dev_schedule_jitter: random delays in front of every second kernel the library enqueues (max_us = 0 switches it off)"""
    check(load().luw_dev_schedule_jitter(int(seed), int(max_us)))


def inject_fault(mask):
    """luw_dev_inject_fault: test hook of the multi-domain host (0 clears)"""
    check(load().luw_dev_inject_fault(int(mask)))
