/*
 * luw_core.h -- C-ABI of the MI355X-native D3Q19 lattice-Boltzmann core that replaces the reference's
 * in-process solver API (class LBM / LBM_Domain of the LUW-modified FluidX3D).
 *
 * The reference has no FFI: its driver (FX/setup.cpp main_setup) talks to the solver through the C++ class
 * `LBM` (FX/lbm.hpp:223-633).  Every entry point below names the reference member it replaces; a
 * maintainer swaps `LBM` for the thin C++ mirror in latticeurbanwind_amd/host/lbm.hpp (same member names)
 * which forwards to this ABI -- see INTEGRATION.md.   ("FX/" = core/cfd_core/FluidX3D/src/ of the reference.)
 *
 * Conventions: plain C, no torch / HIP types in signatures (streams travel as void*), every function
 * returns LUW_OK (0) or a negative error code and records a message readable with luw_last_error().
 * A solver handle is not re-entrant; different handles may be driven from different host threads.
 * All arrays use the reference's global cell index n = x + (y + z*Ny)*Nx (FX/lbm.hpp:512-514) and its
 * SoA field layout: rho[N], u[3N] = ux[N] uy[N] uz[N], flags[N], F[3N] (FX/lbm.cpp:283-294).
 */
#ifndef LUW_CORE_H
#define LUW_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUW_ABI_VERSION 6

/* error codes */
#define LUW_OK 0
#define LUW_ERR_INVALID (-1)     /* bad argument / configuration (reference: print_error + exit(1)) */
#define LUW_ERR_DEVICE (-2)      /* HIP runtime error */
#define LUW_ERR_STATE (-3)       /* call order violated (e.g. run before initialize) */
#define LUW_ERR_NOMEM (-4)

/* cell flags, FX/defines.hpp:50-57 */
#define LUW_TYPE_S 0x01
#define LUW_TYPE_E 0x02
#define LUW_TYPE_T 0x04
#define LUW_TYPE_X 0x40
#define LUW_TYPE_Y 0x80

/* fields (luw_host_ptr / luw_device_ptr) and field masks (luw_upload / luw_download) */
#define LUW_FIELD_RHO 0
#define LUW_FIELD_U 1
#define LUW_FIELD_FLAGS 2
#define LUW_FIELD_F 3
#define LUW_FIELD_FI 4           /* device only: DDFs, 19 planes */
#define LUW_FIELD_T 5            /* temperature (LUW_OPT_TEMPERATURE), lbm.T of the reference */
#define LUW_FIELD_GI 6           /* device only: thermal DDFs, 7 planes */
#define LUW_MASK_RHO (1u<<LUW_FIELD_RHO)
#define LUW_MASK_U (1u<<LUW_FIELD_U)
#define LUW_MASK_FLAGS (1u<<LUW_FIELD_FLAGS)
#define LUW_MASK_F (1u<<LUW_FIELD_F)
#define LUW_MASK_T (1u<<LUW_FIELD_T)

/* DDF storage formats: compile-time `#define FP16C` in the reference (FX/defines.hpp:13-14), a run-time
 * choice here */
#define LUW_DDF_FP32 0
#define LUW_DDF_FP16C 1

/* option bits */
#define LUW_OPT_FORCE_FIELD 0x1u        /* allocate + read the per-cell force F (FORCE_FIELD, FX/kernel.cpp:1617-1623) */
#define LUW_OPT_UPDATE_FIELDS_EVERY_STEP 0x2u /* write rho,u (and T) in every step exactly like UPDATE_FIELDS (FX/kernel.cpp:1709-1716);
                                           without it rho,u,T are written by the last step of each luw_run() call and on
                                           luw_download(), which yields identical values whenever they are observed */
#define LUW_OPT_TEMPERATURE 0x8u        /* thermal D3Q7 lattice (TEMPERATURE, FX/kernel.cpp:1306-1335,1639-1684): T field, TYPE_T cells, cfg.alpha */
#define LUW_OPT_NO_SUBGRID 0x4u         /* disable the Smagorinsky-Lilly model (reference: always on, FX/defines.hpp:25) */
/* FP16C kernels in the hardware's own arithmetic: one v_rcp_f32 for the divisions by the density, v_sqrt_f32 / v_rcp_f32 for the Smagorinsky rate,
 * every fused multiply-add written out, moment / stress sums in trees, populations scaled by 2^-112 between decode and encode -- what the reference's own
 * build does in spirit (-cl-mad-enable, native division: FX/opencl.hpp:305, FX/kernel.cpp:1088-1100,1735).  Results then agree with the bit-exact
 * default of this interface (and with the CPU oracle) to rounding, within the gates of tests/test_gpu_native_arith.py, instead of bit for bit; they are
 * deterministic and the same in the pair and the one-cell kernel, i.e. independent of how a lattice is cut.  The deck driver and bench.py set this bit
 * for FP16C unless asked for `--arith exact`.  Ignored for FP32 DDFs. */
#define LUW_OPT_NATIVE_ARITH 0x10u

/* kernel selection (cfg.kernel, luw_set_kernel): all three compute the same values.  (The A/B and measurement-only variants of the tools build have
 * their ids in include/luw_core_dev.h; this library rejects them.) */
#define LUW_KERNEL_AUTO 0               /* FP32: SCALAR.  FP16C: PAIR for boxes of 128 cells and more in x (whole pairs from a 4-byte boundary), else SCALAR */
#define LUW_KERNEL_SCALAR 1             /* 1 cell / lane, one dword (FP32) per lane and plane; non-temporal on the 14 aligned planes */
#define LUW_KERNEL_PAIR 7               /* FP16C only: 2 cells / lane collided one after the other on packed FP32 pairs, one dword per lane and plane */

typedef struct luw_config {
	uint32_t struct_size;            /* = sizeof(luw_config), ABI check */
	uint32_t Nx, Ny, Nz;             /* LOCAL lattice of this domain incl. halo layers on split axes; LBM_Domain ctor FX/lbm.cpp:246-249 */
	uint32_t Dx, Dy, Dz;             /* domains per axis (n_gpu deck key), FX/lbm.hpp:444 */
	int32_t Ox, Oy, Oz;              /* global offset of local cell (0,0,0), FX/lbm.cpp:1072 */
	float nu;                        /* kinematic viscosity in lattice units; w = 1/(3 nu + 1/2) is derived exactly as FX/lbm.cpp:664 */
	float fx, fy, fz;                /* global volume force, LBM ctor / set_f(), FX/lbm.hpp:492-495 */
	float omega_x, omega_y, omega_z; /* LBM::set_coriolis, FX/lbm.hpp:496-498 */
	uint32_t ddf_format;             /* LUW_DDF_* */
	uint32_t options;                /* LUW_OPT_* */
	/* process-global solver configuration the reference bakes into the kernel source, FX/lbm.cpp:770-782 */
	int32_t buffer_nudging_active;   /* buffer_nudging_active */
	uint32_t buffer_n_cells;         /* buffer_n_cells */
	float buffer_inv_tau_lbmu;       /* buffer_inv_tau_lbmu */
	int32_t buffer_nudge_vertical;   /* buffer_nudge_vertical */
	int32_t buffer_downstream_face_id; /* 0 none, 1 west, 2 east, 3 south, 4 north */
	int32_t top_sponge_active;       /* top_sponge_active (sponge_ref_mode 0 only) */
	uint32_t sponge_n_cells;         /* sponge_n_cells */
	float sponge_inv_tau_lbmu;       /* sponge_inv_tau_lbmu */
	float alpha;                     /* thermal diffusivity in lattice units (LUW_OPT_TEMPERATURE), LBM(..., alpha, beta) FX/lbm.hpp:444 */
	int32_t device;                  /* HIP device ordinal */
	uint32_t kernel;                 /* LUW_KERNEL_* */
} luw_config;

typedef struct luw_solver luw_solver;

/* library */
int luw_abi_version(void);
const char* luw_last_error(void);
int luw_device_count(int* count);            /* smart_device_selection's enumeration, FX/lbm.cpp:947-979 */
/* What Device_Info prints per device (FX/opencl.hpp:25-107) plus where it sits: name, PCI bus id "dddd:bb:dd.f", memory in bytes.
 * Any output pointer may be NULL; name_size / pci_size are the capacities of the two text buffers. */
int luw_device_info(int device, char* name, uint64_t name_size, char* pci_bus_id, uint64_t pci_size, uint64_t* total_memory);
/* The link between two devices of this node as the HIP runtime reports it: can kernels of `device` access `peer`'s memory, the
 * runtime's performance rank of the link, whether atomics are native on it, the link type (HSA_AMD_LINK_INFO_TYPE_*: 4 = xGMI,
 * 2 = PCIe) and the number of hops.  The reference has no counterpart (its domains meet in host memory, FX/lbm.cpp:1895-1935). */
int luw_p2p_info(int device, int peer, int* can_access, int* performance_rank, int* native_atomics, uint32_t* link_type, uint32_t* hops);
/* to_string(float) of the reference (FX/utilities.hpp:2741-2750): 9 significant digits, "d.dddddddd[E<exp>]", "NaN", "Inf".  The
 * solver's constants and the VTK headers (ORIGIN, SPACING) are defined through this text; drivers use the same routine. */
int luw_format_float9(float x, char* text, uint64_t size);

/* life cycle: LBM::LBM (FX/lbm.cpp:1057-1112) / LBM::~LBM.
 * For DDF arrays of 1 GiB and more (planes under 1.5 GiB) luw_create times four steps of the real kernel on the freshly mapped array and, while that is
 * under the rate of the fast class of placements, makes up to FIVE further draws of physical memory (1 GiB chunks again, 4 GiB, 2 GiB, hipMalloc, 512 MiB
 * chunks), keeping the fastest (physical placement changes the step time by up to 10 % on MI355X).  Every array tried stays mapped
 * until the search ends: the PEAK device memory of luw_create is up to six DDF arrays, each further draw only if free memory exceeds one array + 40 B per
 * cell + 2 GiB -- a co-tenant process allocating on the same device in that window can run out of memory; LUW_TUNE_PLACEMENT=0 (or =<n> draws) bounds
 * it.  Nothing of the search outlives luw_create (one search per process and device); it is skipped on devices that several solvers share.
 * Environment knobs: INTEGRATION.md section 5; the library reads them once per process (luw_dev_reload_tuning reads them again). */
int luw_create(const luw_config* cfg, luw_solver** out);
void luw_destroy(luw_solver* s);

/* host mirrors owned by the solver: lbm.rho[n], lbm.u.x/y/z[n], lbm.flags[n], lbm.F (FX/lbm.hpp:428-433).
 * Returns NULL for fields that do not exist.  rho is pre-filled with 1.0f (FX/lbm.cpp:286). */
void* luw_host_ptr(luw_solver* s, int field);
uint64_t luw_get_N(const luw_solver* s);     /* LBM_Domain::get_N */

/* Memory_Container::write_to_device / read_from_device, FX/lbm.hpp:406-416 */
int luw_upload(luw_solver* s, uint32_t field_mask);
int luw_download(luw_solver* s, uint32_t field_mask);

/* LBM::run(0): upload rho,u,flags,F + initialize kernel, t = 0 (FX/lbm.cpp:1221-1260, FX/kernel.cpp:1370-1452) */
int luw_initialize(luw_solver* s);
/* LBM::run(steps): `steps` x { stream_collide; t++ } (FX/lbm.cpp:1262-1312), returns after the device finished */
int luw_run(luw_solver* s, uint64_t steps);
uint64_t luw_get_t(const luw_solver* s);     /* LBM::get_t */

/* run-time setters the reference allows between steps at no cost (kernel arguments, FX/lbm.cpp:345) */
/* 1 when every step of this solver writes rho,u like the reference's UPDATE_FIELDS build: LUW_OPT_UPDATE_FIELDS_EVERY_STEP, or -- decided by
 * luw_initialize -- a buffer-nudging / sponge reference cell (outer face the domain owns, FX/kernel.cpp:1543-1611) is a fluid cell, whose u those
 * terms read one step later, or (thermal lattice) a top-layer cell under the sponge is not a TYPE_T preset; with TYPE_E / solid faces and preset
 * boundary temperatures (every LUW deck) the fields are written by the last step of a run() call only. */
int luw_fields_every_step(const luw_solver* s);
int luw_set_f(luw_solver* s, float fx, float fy, float fz);               /* LBM::set_f */
int luw_set_coriolis(luw_solver* s, float ox, float oy, float oz);        /* LBM::set_coriolis */

/* ---- device-level interface for callers that own streams (multi-GPU driver, on-device inlet updaters) ---- */
/* device buffer handles, the analogue of lbm_domain[d]->u etc. (FX/setup.cpp:1085).  Device layout: x-pitch
 * luw_get_pitch() (>= Nx), index x + (y + z*Ny)*pitch, plane stride luw_get_plane_stride(). */
void* luw_device_ptr(luw_solver* s, int field);
uint32_t luw_get_pitch(const luw_solver* s);
uint64_t luw_get_plane_stride(const luw_solver* s);
int luw_set_stream(luw_solver* s, void* hip_stream);                      /* stream used by all enqueue calls; NULL = solver's own */
/* enqueue ONE stream_collide over the box [x0,x1) x [y0,y1) x [z0,z1) of local cells at the current t, without
 * incrementing t and without host synchronisation (interior / boundary-shell split of the multi-GPU driver).
 * write_fields bit 0: also store rho,u; bit 1 (LUW_WF_SAMPLE): this step is a statistics sample and the box carries the
 * Welford update of its cells itself -- call luw_stats_begin_sample once per sampled step before its boxes; when it reports
 * fused = 0 (thermal lattice, A/B kernels) write the fields instead and call luw_stats_accumulate after the step. */
#define LUW_WF_SAMPLE 2
int luw_stats_begin_sample(luw_solver* s, int* fused);
int luw_enqueue_stream_collide(luw_solver* s, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1, int write_fields);
int luw_increment_time_step(luw_solver* s, uint64_t steps);               /* LBM_Domain::increment_time_step */
int luw_reset_time_step(luw_solver* s);                                   /* LBM_Domain::reset_time_step */
/* halo transfer of the 5 outgoing DDFs per face cell: transfer_extract_fi / transfer__insert_fi
 * (FX/kernel.cpp:2241-2270).  direction 0/1/2 = x/y/z.  Buffers are DEVICE pointers holding 5*A elements of
 * the DDF storage type, A = luw_get_area(direction), element (b*A + a): population b of the face (reference order), face cell a with x running fastest
 * wherever the face contains x (a = y + z*Ny, x + z*Nx, x + y*Nx for direction 0, 1, 2; the reference's y faces run a = z + x*Nz -- the order inside a
 * buffer is private to extract and insert, the host only moves the bytes). */
uint64_t luw_get_area(const luw_solver* s, uint32_t direction);           /* LBM_Domain::get_area */
int luw_enqueue_extract_fi(luw_solver* s, uint32_t direction, void* dev_buffer_p, void* dev_buffer_m);
/* x-split domains: while a pair of face buffers is set here, every stream_collide launch whose box holds the first / last owned x column ALSO writes that
 * column's five outgoing populations into them (same elements as luw_enqueue_extract_fi(s, 0, p, m): the values sit in registers there, the extract kernel
 * fetches them back one element per 128-byte line); luw_enqueue_extract_fi with direction 0 and the same buffers then returns without a launch when the
 * launches of the current step have covered both columns, and does its own work otherwise (sampled steps, the one-cell FP16C kernel, boxes that cut
 * a column).  The buffers must stay valid, and must not be overwritten between a step's launches and the use of its faces.  NULL, NULL switches the
 * output off. */
int luw_set_x_face_buffers(luw_solver* s, void* dev_buffer_p, void* dev_buffer_m);
int luw_enqueue_insert_fi(luw_solver* s, uint32_t direction, const void* dev_buffer_p, const void* dev_buffer_m);
/* The x-face insert without its kernel: luw_set_x_face_inputs(p, m) stands for luw_enqueue_insert_fi(s, 0, p, m) at the current t, but leaves the values in the
 * buffers.  The launches of the NEXT step whose box holds the first / last owned x column read them there (the kernels with the x-face output: the loads
 * they replace are the very slots the insert kernel would have filled, one element per 128-byte line); whatever else needs the values in the lattice first
 * -- a launch of another kernel, a pack kernel, luw_download_fi, a change of t other than the step to t + 1 -- makes the library run the insert kernel then,
 * on the stream set at that moment.  The caller keeps the buffers unchanged until the next step's launches have run (stream order on one stream is enough;
 * a host whose neighbour writes them directly alternates two pairs), calls luw_enqueue_insert_edges AFTER this (the edges across the x cut then land in the
 * rims of these buffers), and does not mix, within one step, launches on ONE border column that can read the buffers with launches that cannot (error;
 * the two columns are tracked separately).  Either buffer may be NULL: that side is not handed over by the call. */
int luw_set_x_face_inputs(luw_solver* s, const void* dev_buffer_p, const void* dev_buffer_m);
/* The halo exchange in ONE phase instead of the reference's x -> y -> z sequence (FX/lbm.cpp:1908-1934, where a population that crosses two cuts at
 * once reaches the diagonal neighbour in two hops through the rims of the faces): in D3Q19 exactly one population crosses a given pair of cuts in a
 * given diagonal direction, along the line where the two faces meet.  Edge e = 0..11 carries population i = 7 + e (c_i: FX/kernel.cpp:890-893) to the
 * domain in direction c_i; its buffer holds luw_get_edge_length(e) elements (the local extent of the third axis; 0 when that pair of axes is not split
 * on this domain).  A host packs all faces and edges, moves everything in one batch, inserts the faces of all axes (any order) and the edges LAST: same
 * populations in the same slots as the three-phase route.  dev_buffers: 12 device pointers; a NULL entry: that edge is not moved by the call. */
uint64_t luw_get_edge_length(const luw_solver* s, uint32_t edge);
int luw_enqueue_extract_edges(luw_solver* s, void* const* dev_buffers);
int luw_enqueue_insert_edges(luw_solver* s, const void* const* dev_buffers);
/* the thermal lattice's halo swap: ONE population per face cell and side (buffers of get_area(direction) DDF elements),
 * transfer_extract_gi / transfer__insert_gi, FX/kernel.cpp:2338-2363.  (The reference also swaps T itself, for rendering only.) */
int luw_enqueue_extract_gi(luw_solver* s, uint32_t direction, void* buf_p, void* buf_m);
int luw_enqueue_insert_gi(luw_solver* s, uint32_t direction, const void* buf_p, const void* buf_m);
int luw_finish(luw_solver* s);                                            /* LBM_Domain::finish_queue */

/* switch the kernel variant of an existing solver (A/B on the same memory; values are identical for all product variants) */
int luw_set_kernel(luw_solver* s, uint32_t kernel);

/* LBM::voxelize_mesh_on_device(mesh, TYPE_S) for a static mesh (FX/lbm.cpp:1411-1645, kernel voxelize_mesh FX/kernel.cpp:2381-2471,
 * rays along z): p0/p1/p2 are the triangle corners (xyz triples, lattice index coordinates of the GLOBAL lattice, i.e. after the
 * driver's scale + translate, FX/setup.cpp:4084-4087); bounds = Mesh::pmin xyz, Mesh::pmax xyz, or NULL to run Mesh::find_bounds
 * (FX/utilities.hpp:4774-4785) on the corners.  Uploads the host flags/u mirrors, works on the device flags and leaves the
 * result in the host flags mirror. */
int luw_voxelize_mesh(luw_solver* s, uint32_t triangle_number, const float* p0, const float* p1, const float* p2, const float* bounds, uint8_t flag);

/* Probe columns (FX/setup.cpp:4495-4506 reads lbm.u at the probe cells after a full-field download every step): attach a
 * list of cells (reference-layout indices n = x+(y+z*Ny)*Nx) once, then luw_gather_u copies u at those cells, packed
 * [i][3], to the host -- a few hundred bytes per step instead of 12 bytes per lattice cell.  The last executed step must
 * have written the fields (luw_run does at the end of each call). */
int luw_gather_attach(luw_solver* s, uint32_t count, const uint64_t* cells);
int luw_gather_u(luw_solver* s, float* out);

/* The same voxelisation on a bare lattice, without a solver object: flags is a host array u8[Nx*Ny*Nz] (reference layout,
 * in/out); bounds = pmin xyz, pmax xyz.  Used by the set-up export of decomposed runs, where the GLOBAL lattice is voxelised
 * once (the reference voxelises per domain with the triangles that overlap it, FX/lbm.cpp:1455-1587: same cells). */
int luw_voxelize_lattice(int device, uint32_t Nx, uint32_t Ny, uint32_t Nz, uint32_t triangle_number, const float* p0, const float* p1,
	const float* p2, const float* bounds, uint8_t flag, uint8_t* flags);

/* von-Karman synthetic-turbulence inlet: the device half of the reference's VonKarmanInletUpdater (FX/setup.cpp:413-1149,
 * kernel vk_inlet_apply FX/kernel.cpp:2495-2571).  The caller builds the tables like build_gpu_runtime_ does
 * (latticeurbanwind_amd/host/vk_inlet.hpp): point_cell[P] = cell index n in the reference layout, point_face[P] = 0 west /
 * 1 east / 2 south / 3 north / 4 top, point_data[7*P] SoA (px, py, pz, base_u.xyz, sigma), mode_data[10*5*M] SoA over the five
 * faces (kx, ky, kz, omega, Ax, Ay, Az, phix, phiy, phiz).  Once attached, luw_run() rewrites u on those cells before every
 * step (the run loop's pre_step_update, FX/setup.cpp:4872) with update_stride / stride_interpolation as in
 * compute_time_params_ (FX/setup.cpp:1118-1140); callers that enqueue steps themselves call luw_vk_inlet_apply(). */
int luw_vk_inlet_attach(luw_solver* s, uint64_t point_count, uint64_t mode_count, const uint64_t* point_cell, const uint8_t* point_face,
                        const float* point_data, const float* mode_data, int update_stride, int stride_interpolation);
int luw_vk_inlet_apply(luw_solver* s);
int luw_vk_inlet_detach(luw_solver* s);

/* on-device time averaging, replaces the per-sample device->host copy + host Welford update of the reference
 * (process_post_step_samples / accumulate_from_buffers, FX/setup.cpp:4441-4542): running mean of u (3 comp.) and rho,
 * M2 of the three velocity components, identical arithmetic and operation order.
 * luw_stats_accumulate samples the device rho,u as they stand (they must have been written by the last step: every
 * luw_run() call ends with such a step).  luw_stats_download returns host arrays in the layout write_avg_vtk consumes
 * (FX/setup.cpp:2513-2683): avg_u AoS [3n+c], the others [n]; any pointer may be NULL. */
int luw_stats_reset(luw_solver* s);
int luw_stats_accumulate(luw_solver* s);
int luw_stats_download(luw_solver* s, float* avg_u, float* avg_rho, float* m2_u, float* m2_v, float* m2_w, uint64_t* count);
/* The sampling window of run_lbm in one call (FX/setup.cpp:4252-4268: after every step of the purge_avg window with
 * (t - avg_start) % purge_avg_stride == 0 the reference copies rho,u to the host and updates the statistics): runs `steps` steps
 * like luw_run; step number first_sample (counted from 1 within this call) and every stride-th step after it are samples.
 * Equivalent, value for value, to { luw_run(1); luw_stats_accumulate(); } at those steps, but nothing waits on the host in
 * between and, with the product kernels, a sampled step carries the Welford update in its own epilogue (the rho,u it would
 * have written and read back never travel: 56 instead of 88 B per cell and sample).  Cells the step never updates (TYPE_S) are
 * taken as the constants they are: mean = field value, M2 = 0 -- what Welford's update yields for a constant, bit for bit.
 * Needs luw_stats_reset first. */
int luw_run_sampled(luw_solver* s, uint64_t steps, uint64_t first_sample, uint64_t stride);
int luw_stats_download_T(luw_solver* s, float* avg_T);   /* running mean of T (LUW_OPT_TEMPERATURE), T_avg of FX/setup.cpp:4481-4484 */

/* ================================================================================================================
 * Several domains in ONE process: the reference's `LBM lbm(N, Dx, Dy, Dz, nu, ...)` for Dx*Dy*Dz > 1 (FX/lbm.hpp:444-450,
 * FX/lbm.cpp:1057-1112): it builds every LBM_Domain itself (one device each), steps them together (do_time_step,
 * FX/lbm.cpp:1262-1290) and swaps their halos (communicate_field, FX/lbm.cpp:1907-1935).  A luw_group is that object: one HIP
 * device and one compute + one communication stream per domain, halo faces written straight into the neighbour's receive buffer
 * by the pack kernels (peer stores over xGMI; hipMemcpyPeerAsync where there is no peer access), boundary shell / interior
 * overlap and pipelined steps as in the one-process-per-GPU driver.  Results equal the undivided run bit for bit.
 * cfg describes the GLOBAL lattice: Nx,Ny,Nz without halos (each divisible by Dx,Dy,Dz: the caller shrinks the grid like
 * FX/lbm.cpp:1058-1060 does), everything else as for luw_create.  devices[d] = HIP device of domain d = x + (y + z*Dy)*Dx, or
 * NULL for cfg->device + d (smart_device_selection's one device per domain, FX/lbm.cpp:947-979; fewer devices than domains is an
 * error unless an explicit list says which domains share a device). */
typedef struct luw_group luw_group;
int luw_group_create(const luw_config* cfg, const int* devices, luw_group** out);       /* LBM::LBM */
void luw_group_destroy(luw_group* g);                                                    /* LBM::~LBM */
uint32_t luw_group_size(const luw_group* g);                                             /* LBM::get_D */
/* lbm.lbm_domain[d] (FX/setup.cpp:1085): host mirrors, device buffers, per-domain calls */
luw_solver* luw_group_domain(luw_group* g, uint32_t d);
int luw_group_domain_info(const luw_group* g, uint32_t d, uint32_t* local_N, int32_t* offset, int* device); /* LBM_Domain::get_Nx.., Ox.. (FX/lbm.cpp:1072) */
int luw_group_overlaps(const luw_group* g);                 /* 1: shell / interior overlap in use (every split axis has >= 4 owned layers) */
int luw_group_direct_peer_stores(const luw_group* g);       /* 1: every face travels as peer stores of the pack kernel, none through a copy */
/* 1: ONE pack / unpack round per step -- the faces of all axes together, the populations that cross two cuts as twelve edge messages straight to the
 * diagonal neighbours, the x faces written by the step kernels and read by the next step's in place (the default: as peer stores where every pair of
 * trading domains has peer access, through send buffers and ONE batch of copies / ncclSend / ncclRecv with the staged and RCCL transports); 0: the
 * reference's three phases x, y, z with rims, FX/lbm.cpp:1907-1935 (LUW_GROUP_EXCHANGE=sequential; a peer transport where SOME pair lacks peer access) */
int luw_group_one_phase(const luw_group* g);
/* How the faces of communicate_field (FX/lbm.cpp:1907-1935) travel between the domains of this process.  Chosen at luw_group_create from
 * the environment variable LUW_GROUP_TRANSPORT = peer (default) | staged | rccl:
 *   PEER    the pack kernel of a domain stores straight into the neighbour's receive buffer (xGMI remote stores); pairs of devices
 *           without peer access fall back to STAGED
 *   STAGED  pack into a send buffer, hipMemcpyPeerAsync to the neighbour
 *   RCCL    pack into a send buffer, grouped ncclSend / ncclRecv on the domains' communication streams (librccl is loaded on demand) */
#define LUW_TRANSPORT_PEER 0
#define LUW_TRANSPORT_STAGED 1
#define LUW_TRANSPORT_RCCL 2
int luw_group_transport(const luw_group* g);                /* the transport in effect (LUW_TRANSPORT_*) */
/* Memory_Container's global index space (FX/lbm.hpp:274-297) over the domains' host mirrors: global arrays in the reference
 * layout n = x + (y + z*Ny)*Nx of the GLOBAL lattice, components SoA.  scatter also fills the halo layers (periodic wrap), i.e.
 * what communicate_rho_u_flags leaves there during LBM::initialize (FX/lbm.cpp:1243-1256); gather reads owned cells. */
int luw_group_scatter(luw_group* g, int field, const void* global_src);
int luw_group_gather(luw_group* g, int field, void* global_dst);
int luw_group_upload(luw_group* g, uint32_t field_mask);
int luw_group_download(luw_group* g, uint32_t field_mask);
/* LBM::initialize incl. the odd-t halo exchange, FX/lbm.cpp:1221-1260 */
int luw_group_initialize(luw_group* g);
int luw_group_run(luw_group* g, uint64_t steps);                                         /* LBM::run(steps): returns after all devices finished */
/* luw_run_sampled for every domain, no host sync inside the window */
int luw_group_run_sampled(luw_group* g, uint64_t steps, uint64_t first_sample, uint64_t stride);
uint64_t luw_group_get_t(const luw_group* g);
int luw_group_set_f(luw_group* g, float fx, float fy, float fz);
int luw_group_set_coriolis(luw_group* g, float ox, float oy, float oz);
/* per domain, FX/lbm.cpp:1455-1587 */
int luw_group_voxelize_mesh(luw_group* g, uint32_t triangle_number, const float* p0, const float* p1, const float* p2, const float* bounds, uint8_t flag);
/* global cell indices in, every domain gets the inlet points / probe cells it owns (VonKarmanInletUpdater::build_gpu_runtime_ does
 * the same per domain, FX/setup.cpp:1012-1057) */
int luw_group_vk_inlet_attach(luw_group* g, uint64_t point_count, uint64_t mode_count, const uint64_t* point_cell, const uint8_t* point_face,
                              const float* point_data, const float* mode_data, int update_stride, int stride_interpolation);
int luw_group_gather_attach(luw_group* g, uint32_t count, const uint64_t* cells);
int luw_group_gather_u(luw_group* g, float* out);
int luw_group_stats_reset(luw_group* g);
/* ---- One domain's share of a decomposed step: the schedule BOTH hosts run -- luw_group_* for all domains of this process, and a host that owns one domain
 * per process (latticeurbanwind_amd/distributed.py over RCCL), which does the exchange itself between luw_domain_step_launch calls.  Replaces, per domain,
 * the body of LBM::do_time_step (FX/lbm.cpp:1262-1290) up to communicate_fi: here the boundary shell runs first on the communication stream, the
 * interior on the compute stream, and consecutive steps are pipelined (interior(t) behind shell(t-1) only).
 * luw_step_boxes: the boxes alone, pure host arithmetic (local extents incl. halo layers, halo flags per axis, thickness of the x slabs): whole6 / interior6 =
 * x0,x1,y0,y1,z0,z1; shell_boxes = up to six such boxes. */
typedef struct luw_domain_step luw_domain_step;
int luw_step_boxes(const uint32_t* local_N, const uint32_t* halo, uint32_t x_shell, uint32_t* whole6, uint32_t* interior6, uint32_t* shell_boxes,
	uint32_t* shell_count, int* can_overlap);
int luw_domain_step_create(luw_solver* s, void* compute_stream, void* comm_stream, uint32_t x_shell /* 0: 128 cells */, int overlap, luw_domain_step** out);
void luw_domain_step_destroy(luw_domain_step* d);
int luw_domain_step_overlaps(const luw_domain_step* d);       /* 1: shell on the communication stream + interior on the compute stream, pipelined */
/* the kernels of one step (the von-Karman inlet update first); no exchange, no t++ */
int luw_domain_step_launch(luw_domain_step* d, int write_fields /* bit 0 | LUW_WF_SAMPLE */, int timed);
/* a sampled step without fused statistics: luw_stats_accumulate behind the step, ordered against the next shell */
int luw_domain_step_separate_stats(luw_domain_step* d);

/* VTK payloads straight from the devices: Memory_Container::write_vtk (FX/lbm.hpp:307-356) and the sections of write_avg_vtk (FX/setup.cpp:2513-2683)
 * without the full-field download.  Every domain converts its own cells on its device -- SoA -> AoS, SI scaling, big-endian -- in z slabs that a
 * writer thread puts into the open file `fd` with pwrite() from `file_offset` on (the caller has written the text header before it); planes
 * z < Nz_write are written (0: all).  Same bytes as the host conversion.  source:
 *   U, RHO       factor * value                         (raw_u / raw_rho files; rho,u must be current: the last step of a run writes them)
 *   T            value * factor + offset if affine      (units.si_T), else factor * value
 *   AVG_U, AVG_RHO, AVG_T     mean * factor + offset    (u_avg, rho_avg, T_avg of the averaged file)
 *   FLUID, TKE, TI, TLS       the derived fields of write_avg_vtk from the means and second moments on the devices, times factor; u_factor = SI
 *                velocity per lattice velocity, grid_dx = SI cell size, tls_cap = cap of the length scale; want_* = the deck's output switches */
#define LUW_EXPORT_U 0
#define LUW_EXPORT_RHO 1
#define LUW_EXPORT_T 2
#define LUW_EXPORT_AVG_U 3
#define LUW_EXPORT_AVG_RHO 4
#define LUW_EXPORT_AVG_T 5
#define LUW_EXPORT_FLUID 6
#define LUW_EXPORT_TKE 7
#define LUW_EXPORT_TI 8
#define LUW_EXPORT_TLS 9
typedef struct luw_export_params {
	uint32_t struct_size;
	float factor, offset; int32_t affine;
	float u_factor, grid_dx, tls_cap; int32_t want_tke, want_ti, want_tls;
} luw_export_params;
int luw_group_export_vtk(luw_group* g, int source, const luw_export_params* prm, uint32_t Nz_write, int fd, uint64_t file_offset);
uint64_t luw_group_stats_count(const luw_group* g);            /* samples accumulated since luw_group_stats_reset (avg_count of FX/setup.cpp:4441) */
int luw_group_stats_download(luw_group* g, float* avg_u, float* avg_rho, float* m2_u, float* m2_v, float* m2_w, float* avg_T, uint64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* LUW_CORE_H */
