// luw_core.hip -- HIP kernels (gfx950) + C-ABI of the MI355X-native D3Q19 core.  See include/luw_core.h.
//
// Device code (included below):
//   luw_device.hpp          per-cell arithmetic: f_eq, moments, forces, collision (scalar and packed), FP16C codec, thermal cell
//   luw_kernels_common.hpp  slot algebra, k_initialize (f_eq(rho,u) -> Esoteric-Pull store with t=1, FX/kernel.cpp:1370-1452)
//   luw_kernels_step.hpp    k_stream_collide_s (1 cell per lane: FP32 product kernel, FP16C fallback, thermal lattice) and
//                           k_stream_collide_p (FP16C product kernel: 2 cells per lane, packed FP32 collision)
//   luw_kernels_aux.hpp     k_extract_fi / k_insert_fi (halo pack/unpack, FX/kernel.cpp:2241-2270), voxeliser, probe gather,
//                           von-Karman inlet, statistics, codec self-check
// This file: the host runtime (allocation and placement, launches, kernel choice) and the C-ABI.
//
// Memory layout in HBM: SoA planes fi[q][z][y][x] with x-pitch Px (multiple of 64) and plane stride Np=Px*Ny*Nz, every
// array shifted by a lead pad so that the first owned cell of a row starts a 256-byte block (lead_alloc);
// rho[Np], u[3][Np], flags[Np], F[3][Np] share the pitch.  Host mirrors keep the reference layout (pitch Nx).
#include "luw_device.hpp"
#include "../../include/luw_core_dev.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <functional>
#include <atomic>
#include <algorithm>
#include <memory>
#include <dlfcn.h>
#include <chrono>

using namespace luw;

#include "luw_kernels_common.hpp"
#include "luw_kernels_step.hpp"
#ifdef LUW_AB_KERNELS   // tools build only (make ab; the header lives under tools/ab_kernels/): A/B and measurement-only kernel variants
#include LUW_AB_KERNELS
#endif
#include "luw_kernels_aux.hpp"

// =====================================================================================================
// host side
// =====================================================================================================
static thread_local std::string g_last_error;
static int fail(const int code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr) do { \
	const hipError_t e_ = (expr); \
	if(e_!=hipSuccess) return fail(LUW_ERR_DEVICE, std::string(#expr)+": "+hipGetErrorString(e_)); \
} while(0)

// ---------------------------------------------------------------- tuning table
// Every knob the library takes from the environment, read ONCE into this table at first use (luw_dev_reload_tuning() reads it again: tests and A/B tools
// that change the environment between two solvers of one process).  Nothing else in the library calls getenv, and no run / step path reads the
// environment.  INTEGRATION.md section 5 lists each knob with its default and purpose; tests/test_capi_library.py holds the two lists together.
struct Tuning {
	size_t alloc_chunk = 1024ull<<20; // LUW_ALLOC = vmm:<MiB> (physical chunk size of lattice-sized arrays) | vmm:one (~0: one piece) | malloc (0: hipMalloc)
	bool copy_staged = false;         // LUW_COPY_STAGED: host <-> device copies of every array through the staging buffer (test aid)
	bool addr_row = false;            // LUW_ADDR_ROW: FP32 kernel in the row addressing form also where the flat form would do (test aid, same values)
	bool pair_general = false;        // LUW_PAIR_GENERAL: FP16C kernels never take the force-free / uniform-force specialisations (test aid, same values)
	bool fuse_stats = true;           // LUW_FUSE_STATS=0: sampled steps use the separate statistics kernel (A/B and test aid, same values)
	uint64_t plane_skew = 0ull;       // LUW_PLANE_SKEW=<64-element blocks> behind each DDF plane (0: 513 for FP32, 33 for FP16C; study aid)
	int placement_candidates = -1;    // LUW_TUNE_PLACEMENT=<n>: allocations of the DDF array luw_create may try (0 / 1: no search; default 4)
	double placement_bar = 0.0;       // LUW_TUNE_FAST=<TB/s>: probe rate from which a placement is kept without further candidates (99: try all; test aid)
	bool placement_verbose = false;   // LUW_TUNE_VERBOSE: print every candidate's probe time to stderr
	bool vk_ahead = true;             // LUW_VK_AHEAD=0: von-Karman inlet evaluated in line instead of one step ahead on a side stream (A/B aid, same values)
	bool voxelize_all = false;        // LUW_VOXELIZE_ALL_TRIANGLES: every voxeliser tile tests every triangle (test aid for the bins)
	uint32_t x_shell = 0u;            // LUW_X_SHELL=<cells>: thickness of the x boundary slabs of a decomposed step (0: 128; A/B aid)
	int group_transport = LUW_TRANSPORT_PEER; bool group_transport_bad = false; // LUW_GROUP_TRANSPORT = peer | staged | rccl (luw_group_create)
	bool group_threads = false;       // LUW_GROUP_THREADS=1: one host thread per domain in luw_group_run
#ifdef LUW_AB_KERNELS                 // tools build only
	int ab_kernel = -1;               // LUW_KERNEL=<id>: overrides the kernel choice of callers that expose none
	bool ab_pair_copy = false;        // LUW_PAIR_COPY: the pair kernel's memory path alone (no physics)
#endif
};
static Tuning g_tuning;
static std::atomic<bool> g_tuning_loaded{false};
static void tuning_load() {
	Tuning t;
	auto on = [](const char* n) { return getenv(n)!=nullptr; };
	auto off0 = [](const char* n) { const char* e = getenv(n); return e&&e[0]=='0'; };
	if(const char* e = getenv("LUW_ALLOC")) {
		if(strncmp(e, "malloc", 6)==0) t.alloc_chunk = 0u;
		else if(strncmp(e, "vmm:one", 7)==0) t.alloc_chunk = ~(size_t)0u;
		else if(strncmp(e, "vmm:", 4)==0) { const size_t v = (size_t)strtoull(e+4, nullptr, 10); if(v) t.alloc_chunk = v<<20; }
	}
	t.copy_staged = on("LUW_COPY_STAGED"); t.addr_row = on("LUW_ADDR_ROW"); t.pair_general = on("LUW_PAIR_GENERAL");
	t.fuse_stats = !off0("LUW_FUSE_STATS");
	if(const char* e = getenv("LUW_PLANE_SKEW")) t.plane_skew = strtoull(e, nullptr, 10);
	if(const char* e = getenv("LUW_TUNE_PLACEMENT")) t.placement_candidates = atoi(e);
	if(const char* e = getenv("LUW_TUNE_FAST")) t.placement_bar = atof(e);
	t.placement_verbose = on("LUW_TUNE_VERBOSE");
	t.vk_ahead = !off0("LUW_VK_AHEAD");
	t.voxelize_all = on("LUW_VOXELIZE_ALL_TRIANGLES");
	if(const char* e = getenv("LUW_X_SHELL")) t.x_shell = (uint32_t)strtoul(e, nullptr, 10);
	if(const char* e = getenv("LUW_GROUP_TRANSPORT")) {
		if(strcmp(e, "rccl")==0) t.group_transport = LUW_TRANSPORT_RCCL;
		else if(strcmp(e, "staged")==0) t.group_transport = LUW_TRANSPORT_STAGED;
		else if(strcmp(e, "peer")!=0&&e[0]) t.group_transport_bad = true;
	}
	{ const char* e = getenv("LUW_GROUP_THREADS"); t.group_threads = e&&e[0]=='1'; }
#ifdef LUW_AB_KERNELS
	if(const char* e = getenv("LUW_KERNEL")) t.ab_kernel = atoi(e);
	t.ab_pair_copy = on("LUW_PAIR_COPY");
#endif
	g_tuning = t;
	g_tuning_loaded.store(true);
}
static const Tuning& tuning() { if(!g_tuning_loaded.load()) tuning_load(); return g_tuning; }

// ---- floats as 9-significant-digit text.  The reference bakes its kernel constants into OpenCL source as decimal text and writes
// VTK headers the same way (to_string(float), FX/utilities.hpp:2603-2634,2741-2750; used at FX/lbm.cpp:664,774,780): what the
// kernel computes with is the float READ BACK from that text, so the digits have to be the reference's, one float operation at a
// time.  Decimal exponent: a binary ladder of powers of ten, each rung applied at most once, from 10^32 down to 10^1 -- scaling
// down while the value is >= 10, scaling up while it is below 1 (the thresholds of the upward ladder sit one decade lower, so
// that the mantissa ends in [1, 10)).  Digits: the integer part, then eight decimals from one truncation of (x - int) * 10^8
// and one round-half-up whose carry may run into the integer part and the exponent.
struct Decimal9 { bool negative, special; uint32_t integral, decimals; int exponent; };
static Decimal9 split_decimal9(float x) {
	Decimal9 d = { x<0.0f, false, 0u, 0u, 0 };
	if(d.negative) x = -x;
	if(std::isnan(x)||std::isinf(x)) { d.special = true; return d; }
	static const struct {
		float at_least, times;
		int decades;
	} down[6] = { { 1E32f, 1E-32f, 32 }, { 1E16f, 1E-16f, 16 }, { 1E8f, 1E-8f, 8 }, { 1E4f, 1E-4f, 4 }, { 1E2f, 1E-2f, 2 }, { 1E1f, 1E-1f, 1 } };
	static const struct {
		float below, times;
		int decades;
	} up[6] = { { 1E-31f, 1E32f, 32 }, { 1E-15f, 1E16f, 16 }, { 1E-7f, 1E8f, 8 }, { 1E-3f, 1E4f, 4 }, { 1E-1f, 1E2f, 2 }, { 1E0f, 1E1f, 1 } };
	if(x>=10.0f) for(const auto& r : down) if(x>=r.at_least) { x *= r.times; d.exponent += r.decades; }
	if(x>0.0f&&x<=1.0f) for(const auto& r : up) if(x<r.below) { x *= r.times; d.exponent -= r.decades; }
	d.integral = (uint32_t)x;
	const float scaled = (x-(float)d.integral)*1E8f;
	d.decimals = (uint32_t)scaled;
	if(scaled-(float)d.decimals>=0.5f&&++d.decimals>=100000000u) { // half up; 0.99999999x carries
		d.decimals = 0u;
		if(++d.integral>=10u) { d.integral = 1u; d.exponent++; }
	}
	return d;
}
static void format_decimal9(const float x, char* text, const size_t size) {
	const Decimal9 d = split_decimal9(x);
	const char* sign = d.negative ? "-" : "";
	if(d.special) snprintf(text, size, "%s%s", sign, std::isnan(x) ? "NaN" : "Inf");
	else if(d.exponent!=0) snprintf(text, size, "%s%u.%08uE%d", sign, d.integral, d.decimals, d.exponent);
	else snprintf(text, size, "%s%u.%08u", sign, d.integral, d.decimals);
}
// the float the reference kernel sees after device_defines() printed it and the OpenCL compiler parsed it
static float literal_roundtrip(const float x) {
	if(std::isnan(x)||std::isinf(x)) return x;
	char text[48];
	format_decimal9(x, text, sizeof(text));
	return strtof(text, nullptr);
}

// One block of device memory.  Two ways to get it: hipMalloc, or the virtual memory management API: ONE address range backed by
// physical chunks of a chosen size (hipMemCreate / hipMemMap), which fixes the size of the physically contiguous pieces a lattice
// array is made of instead of leaving it to the state of the driver's heap (alloc_vmm_chunk below).
struct DevBlock {
	void* base = nullptr; size_t bytes = 0u;
	std::vector<hipMemGenericAllocationHandle_t> chunks; size_t chunk_bytes = 0u; // VMM only
};
// the mapped pieces of a block, in address order: (offset, length) of every hipMemMap call of dev_alloc
static void dev_block_pieces(const DevBlock& b, const size_t mapped, std::vector<std::pair<size_t, size_t>>& out) {
	out.clear();
	for(size_t off=0u; off<mapped&&out.size()<b.chunks.size(); off+=b.chunk_bytes) out.emplace_back(off, std::min(b.chunk_bytes, mapped-off));
}
// Tear-down of a mapped block: every piece is unmapped with ITS OWN (address, length) -- hipMemUnmap takes one mapping, not a range of them.  Round 3 unmapped
// the whole reserved range in one call (and ignored the return code): the pieces behind the first stayed mapped while their handles were released and the
// address range freed, which is what made a release in the middle of luw_create crash a later one (tools/vmm_unmap_repro.hip shows both sequences).
// The device is idle first: nothing in flight may still address the block.
// keep_range: the physical memory goes back now, the (empty) address range stays reserved until a later dev_free of the same block -- a candidate of the
// placement search is released this way, so that the next candidate never lands on addresses whose page-table entries were torn down a moment ago.
static void dev_free(DevBlock& b, const bool keep_range = false) {
	if(!b.base) return;
	(void)hipDeviceSynchronize();
	if(b.chunks.empty()&&b.chunk_bytes==0u) (void)hipFree(b.base);
	else {
		std::vector<std::pair<size_t, size_t>> pieces;
		dev_block_pieces(b, b.bytes, pieces);
		for(const auto& pc : pieces) (void)hipMemUnmap((char*)b.base+pc.first, pc.second);
		for(auto& h : b.chunks) (void)hipMemRelease(h);
		b.chunks.clear();
		if(keep_range) { (void)hipGetLastError(); return; }           // chunk_bytes != 0 marks the block as "a reserved range, nothing mapped"
		(void)hipMemAddressFree(b.base, b.bytes);
	}
	(void)hipGetLastError();
	b = DevBlock{};
}
static hipError_t dev_alloc(DevBlock& b, const size_t bytes, const int device, const size_t vmm_chunk, const bool short_last = false) {
	b = DevBlock{};
	if(vmm_chunk==0u) { const hipError_t e = hipMalloc(&b.base, bytes); if(e==hipSuccess) b.bytes = bytes; else b.base = nullptr; return e; }
	hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
	size_t gran = 0u;
	hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
	if(e!=hipSuccess) return e;
	const size_t want = vmm_chunk==~(size_t)0u ? bytes : vmm_chunk;   // ~0: the whole block as one physical allocation
	const size_t chunk = ((std::max(want, gran)+gran-1u)/gran)*gran;
	// short_last: the last chunk holds only the remainder (at the allocation granularity) instead of a whole chunk
	const size_t whole = (bytes/chunk)*chunk, rest = ((bytes-whole+gran-1u)/gran)*gran;
	const size_t total = short_last ? whole+rest : ((bytes+chunk-1u)/chunk)*chunk;
	if((e = hipMemAddressReserve(&b.base, total, chunk, nullptr, 0ull))!=hipSuccess) { b.base = nullptr; return e; }
	b.bytes = total; b.chunk_bytes = chunk;
	size_t mapped = 0u;
	for(size_t off=0u; off<total; off+=chunk) {
		const size_t len = std::min(chunk, total-off);
		hipMemGenericAllocationHandle_t h;
		if((e = hipMemCreate(&h, len, &prop, 0ull))!=hipSuccess) break;
		if((e = hipMemMap((char*)b.base+off, len, 0u, h, 0ull))!=hipSuccess) { (void)hipMemRelease(h); break; }
		b.chunks.push_back(h); mapped += len;
	}
	if(e==hipSuccess) {
		hipMemAccessDesc acc{};
		acc.location = prop.location;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		e = hipMemSetAccess(b.base, total, &acc, 1u);
	}
	if(e!=hipSuccess) { // undo what was mapped
		std::vector<std::pair<size_t, size_t>> pieces;
		dev_block_pieces(b, mapped, pieces);
		for(const auto& pc : pieces) (void)hipMemUnmap((char*)b.base+pc.first, pc.second);
		for(auto& h : b.chunks) (void)hipMemRelease(h);
		(void)hipMemAddressFree(b.base, total);
		b = DevBlock{};
	}
	return e;
}
// How lattice-sized arrays are allocated.  Measured on MI355X (tools/placement_study.sh, profiles/r02_placement_study.txt): the step
// kernel on a hipMalloc'ed DDF array runs in one of two classes (512^3 FP32: 3.37 or 3.65 ms, fixed for the life of the
// allocation), on an array mapped from 1 GiB physical chunks always in 3.26 ms; 1024x1024x256: 7.28 -> 6.69 ms.  Large
// physically contiguous pieces let the GPU's page tables use large fragments, and the 19 + 19 streams of a step stop
// missing in the TLBs.  So arrays of at least 64 MiB are mapped from chunks (LUW_ALLOC=vmm:<chunk MiB>, default 1024);
// LUW_ALLOC=malloc restores plain hipMalloc (and with it the placement search of tune_ddf_placement).
static size_t alloc_vmm_chunk() { return tuning().alloc_chunk; }

struct luw_solver {
	luw_config cfg;
	KParams kp;
	uint64_t N = 0;          // Nx*Ny*Nz
	uint64_t t = 0;
	bool initialized = false;
	bool counted = false;    // registered in g_live_solvers
	bool fields_current = true; // device rho,u reflect the state after the last executed step
	bool every_step_auto = false; // a nudging / sponge reference cell is a fluid cell: rho,u are written by every step (see reference_cells_are_inputs)
	size_t ddf_bytes = 4;
	void* d_fi = nullptr;
	float* d_rho = nullptr; float* d_u = nullptr; uint8_t* d_flags = nullptr; float* d_F = nullptr;
	float* d_wbuf = nullptr; float* d_sigma = nullptr;
	float* d_avg_u = nullptr; float* d_avg_rho = nullptr; float* d_m2 = nullptr; uint64_t avg_count = 0ull;
	uint32_t vk_P = 0u, vk_M = 0u; int vk_stride = 1; bool vk_interp = false, vk_active = false; uint64_t vk_last_t = ~0ull;
	uint32_t* d_vk_cell = nullptr; uint8_t* d_vk_face = nullptr; float* d_vk_point = nullptr; float* d_vk_mode = nullptr;
	// the inlet values of step t+1 are evaluated on a side stream while step t runs (vk_apply): two packed buffers, the step each holds
	float* d_vk_val[2] = { nullptr, nullptr };
	uint64_t vk_val_t[2] = { ~0ull, ~0ull };
	hipStream_t vk_stream = nullptr;
	hipEvent_t vk_ready[2] = { nullptr, nullptr }, vk_taken[2] = { nullptr, nullptr };
	float* h_rho = nullptr; float* h_u = nullptr; uint8_t* h_flags = nullptr; float* h_F = nullptr;
	void* d_gi = nullptr; float* d_T = nullptr; float* h_T = nullptr; float* d_avg_T = nullptr; // TEMPERATURE
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	uint32_t kernel = LUW_KERNEL_AUTO;
	std::vector<DevBlock> raw; // device blocks behind the lattice-sized arrays (lead_alloc)
	uint32_t gather_count = 0u; uint32_t* d_gather_cell = nullptr; float* d_gather_out = nullptr; // probe columns
	void* d_stage = nullptr; size_t stage_bytes = 0u; // copy_pitched's staging buffer (chunk-mapped arrays)
	// x-face output of the step kernels (luw_set_x_face_buffers): buffers, and which of the two border columns the launches of step xf_t have covered
	void* xf_p = nullptr; void* xf_m = nullptr; uint32_t xf_cover = 0u; uint64_t xf_t = ~0ull;
	// what luw_create's placement search did (luw_dev_placement_info): candidates probed, the kind kept, its probe rate, seconds spent in luw_create
	int placement_tried = 0; std::string placement_kept = "default (no search)"; double placement_tbps = 0.0, create_seconds = 0.0;
};

// Lattice-sized device arrays start LEAD elements into their allocation, LEAD = 64 - halo_x: with the x pitch a multiple of
// 64 elements, the first OWNED cell of every row (x = halo_x) then begins a 256-byte (FP32) / 128-byte (FP16C) block, so
// that a wave's 64 consecutive cells are exactly the lines it touches -- in a halo'ed domain (Nx = 512 + 2) just as in a
// single one.  Measured on MI355X before this: interior kernel of a 514x514x512 domain 7.3 ms vs 3.5 ms for 512^3.
static hipError_t lead_alloc(luw_solver* s, void** base, const size_t elems, const size_t elem_bytes, const size_t* chunk_override = nullptr) {
	DevBlock blk;
	const size_t lead = (size_t)(64u-s->kp.halo_x)*elem_bytes, total = elems*elem_bytes+64u*elem_bytes;
	// arrays under 64 MiB: hipMalloc.  Larger ones: chunks of EXACTLY the configured size (1 GiB) -- measured: the same lattice on chunks of 0.93 GiB
	// ("equal pieces, no waste") runs the step 8-15 % slower (1024x1024x256 FP32 7.30 vs 6.72 ms, FP16C 3.97 vs 3.45; profiles/r03_alloc_chunks_ab.txt) --
	// and only the LAST chunk cut to the remainder, so that an array costs at most one allocation granule more than its size instead of up to a
	// whole chunk (u, m2, avg_u of a 512^3 domain: 1.5 -> 2 GiB each before).
	size_t chunk = 0u;
	if(total>=(64ull<<20)) {
		const size_t cap = chunk_override ? *chunk_override : alloc_vmm_chunk(), mib2 = 2ull<<20;   // (override: tune_ddf_placement's candidates)
		chunk = (cap==0u||cap==~(size_t)0u) ? cap : std::min<size_t>(cap, ((total+mib2-1u)/mib2)*mib2);
	}
	hipError_t e = dev_alloc(blk, total, s->cfg.device, chunk, true);
	if(e!=hipSuccess&&chunk) { (void)hipGetLastError(); e = dev_alloc(blk, total, s->cfg.device, 0u); } // no VMM on this system: hipMalloc
	if(e!=hipSuccess) return e;
	e = hipMemsetAsync(blk.base, 0, total, s->stream); // padding / not-yet-uploaded memory must hold defined values
	*base = (char*)blk.base+lead;
	s->raw.push_back(std::move(blk));
	return e;
}

static int set_device(const luw_solver* s) { HIP_TRY(hipSetDevice(s->cfg.device)); return LUW_OK; }

// Host mirror (reference layout, pitch Nx) <-> device array (pitch Px), `planes` components: 1-D copies between the host and a
// contiguous device staging buffer of bounded size, and a kernel that moves the rows between staging and lattice (55 GB/s either
// way on the test box: PCIe-bound).  The runtime's 2-D copy serves the cases it handles well (see below).
template<typename E> __global__ __launch_bounds__(256) void k_rows_copy(E* __restrict__ lattice, const size_t lattice_pitch, E* __restrict__ staging,
	const uint32_t nx, const bool to_lattice) {
	const uint32_t x = blockIdx.x*blockDim.x+threadIdx.x;
	if(x>=nx) return;
	const size_t row = (size_t)blockIdx.y+(size_t)blockIdx.z*gridDim.y;
	E* l = lattice+row*lattice_pitch+x; E* t = staging+row*nx+x;
	if(to_lattice) *l = *t; else *t = *l;
}
static const DevBlock* block_of(const luw_solver* s, const void* p) {
	for(const DevBlock& b : s->raw) if((const char*)p>=(const char*)b.base&&(const char*)p<(const char*)b.base+b.bytes&&!(b.chunks.empty()&&b.chunk_bytes)) return &b;
	return nullptr;
}
static int copy_pitched(void* dst, const void* src, const size_t elem, luw_solver* s, const uint32_t planes, const bool to_device, hipStream_t st) {
	const size_t rows = (size_t)s->cfg.Ny*s->cfg.Nz;
	// hipMemcpy2DAsync is left to the one case it is good at -- arrays in ONE physical piece whose rows are whole dwords.  It rejects
	// ranges that span the chunks of a mapped array ("invalid argument"), and rows that are not a multiple of four bytes take a path
	// that moves 0.06 GB/s (flags of a 514-cell-wide domain: 2.1 s instead of 3 ms) -- both go through the staging buffer.
	const DevBlock* blk = block_of(s, to_device ? dst : src);
	const bool force_staged = tuning().copy_staged; // test aid: the staged path for every array
	if(!force_staged&&(!blk||blk->chunks.size()<=1u)&&((size_t)s->cfg.Nx*elem)%4u==0u) {
		for(uint32_t c=0u; c<planes; c++) {
			if(to_device) HIP_TRY(hipMemcpy2DAsync((char*)dst+(size_t)c*s->kp.Np*elem, (size_t)s->kp.Px*elem, (const char*)src+(size_t)c*s->N*elem,
				(size_t)s->cfg.Nx*elem, (size_t)s->cfg.Nx*elem, rows, hipMemcpyHostToDevice, st));
			else HIP_TRY(hipMemcpy2DAsync((char*)dst+(size_t)c*s->N*elem, (size_t)s->cfg.Nx*elem, (const char*)src+(size_t)c*s->kp.Np*elem,
				(size_t)s->kp.Px*elem, (size_t)s->cfg.Nx*elem, rows, hipMemcpyDeviceToHost, st));
		}
		return LUW_OK;
	}
	const size_t row_bytes = (size_t)s->cfg.Nx*elem;
	if(s->d_stage&&row_bytes>s->stage_bytes) { // sized by a narrower element type on a lattice of very few rows: too small for one row of this one
		HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(s->d_stage); s->d_stage = nullptr;
	}
	if(!s->d_stage) { // copies and kernels of successive batches are ordered by the stream, so one buffer serves them all
		s->stage_bytes = std::max<size_t>(row_bytes, std::min<size_t>(256ull<<20, rows*row_bytes));
		if(hipMalloc(&s->d_stage, s->stage_bytes)!=hipSuccess) { s->d_stage = nullptr; return fail(LUW_ERR_NOMEM, "copy: staging buffer"); }
	}
	const size_t batch_rows = std::max<size_t>(1u, s->stage_bytes/row_bytes);
	const uint32_t bx = s->cfg.Nx>=256u ? 256u : ((s->cfg.Nx+63u)/64u)*64u;
	for(uint32_t c=0u; c<planes; c++) for(size_t r0=0u; r0<rows; r0+=batch_rows) {
		const size_t nr = std::min(batch_rows, rows-r0);
		char* stage = (char*)s->d_stage;
		char* lat = (char*)(to_device ? dst : const_cast<void*>(src))+((size_t)c*s->kp.Np+r0*s->kp.Px)*elem;
		char* host = (char*)(to_device ? const_cast<void*>(src) : dst)+((size_t)c*s->N+r0*s->cfg.Nx)*elem;
		// rows as a (y, z)-shaped grid: gridDim.y <= 65535
		const uint32_t gy = (uint32_t)std::min<size_t>(nr, 4096u);
		// the kernel indexes row = y + z*gy and may run past nr in the last z slab: launch the full slabs and the remainder separately
		auto launch = [&](const size_t first, const uint32_t ny, const uint32_t nz) {
			if(ny==0u||nz==0u) return;
			const dim3 grid((s->cfg.Nx+bx-1u)/bx, ny, nz), block(bx);
			char* l = lat+first*s->kp.Px*elem; char* t = stage+first*row_bytes;
			if(elem==4u) hipLaunchKernelGGL(k_rows_copy<uint32_t>, grid, block, 0, st, (uint32_t*)l, (size_t)s->kp.Px, (uint32_t*)t, s->cfg.Nx, to_device);
			else if(elem==2u) hipLaunchKernelGGL(k_rows_copy<uint16_t>, grid, block, 0, st, (uint16_t*)l, (size_t)s->kp.Px, (uint16_t*)t, s->cfg.Nx, to_device);
			else hipLaunchKernelGGL(k_rows_copy<uint8_t>, grid, block, 0, st, (uint8_t*)l, (size_t)s->kp.Px, (uint8_t*)t, s->cfg.Nx, to_device);
		};
		const uint32_t full = (uint32_t)(nr/gy), rem = (uint32_t)(nr%gy);
		if(to_device) {
			HIP_TRY(hipMemcpyAsync(stage, host, nr*row_bytes, hipMemcpyHostToDevice, st));
			launch(0u, gy, full); launch((size_t)full*gy, rem, 1u);
		} else {
			launch(0u, gy, full); launch((size_t)full*gy, rem, 1u);
			HIP_TRY(hipMemcpyAsync(host, stage, nr*row_bytes, hipMemcpyDeviceToHost, st));
		}
		HIP_TRY(hipGetLastError());
	}
	return LUW_OK;
}

#ifdef LUW_AB_KERNELS
template<typename T, int V> static void launch_vec(luw_solver* s, const Box& b, const int write_fields) {
	T* fi = (T*)s->d_fi;
	const bool odd = (s->t&1ull)!=0ull;
	const uint32_t nvec = (b.x1-1u)/V-b.x0/V+1u;          // vectors overlapping [x0,x1)
	uint32_t vx = 1u; while(vx<nvec&&vx<256u) vx <<= 1;   // power of two
	const uint32_t ry = 256u/vx;
	const uint32_t nchunk = (nvec+vx-1u)/vx;
	const uint32_t rows = (b.y1-b.y0)*(b.z1-b.z0);
	const dim3 grid(((rows+ry-1u)/ry)*nchunk), block(vx, ry);
	if(odd)
		hipLaunchKernelGGL((k_stream_collide_v<T, V, 1>), grid, block, 0, s->stream, s->kp, b, nchunk, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
	else hipLaunchKernelGGL((k_stream_collide_v<T, V, 0>), grid, block, 0, s->stream, s->kp, b, nchunk, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
}
#endif
// threads per block for a row of nx lanes: whole waves, at most 256, chosen so that the blocks of a row carry the fewest idle
// lanes (a 375-lane row of the pair kernel: 3 x 128 instead of 2 x 256; ties go to the larger block)
static uint32_t row_block(const uint32_t nx) {
	if(nx<=256u) return ((nx+63u)/64u)*64u;
	uint32_t best = 256u, best_lanes = ((nx+255u)/256u)*256u;
	for(uint32_t bx : {192u, 128u, 64u}) { const uint32_t lanes = ((nx+bx-1u)/bx)*bx; if(lanes<best_lanes) { best = bx; best_lanes = lanes; } }
	return best;
}
// Where the position-dependent forces of this domain act, as cell ranges per face (the host's copy of in_force_zone, luw_device.hpp): buffer
// nudging within Nbuf cells of the lateral faces the domain owns (not the downstream one) and of the top, the sponge in the sponge_N layers
// under the top.  lo[a] / hi[a]: the first cell behind the zone at the low face of axis a / the first cell of the zone at its high face
// (0 / N when there is none): cells of [lo, hi) on all three axes are outside every zone.
static void force_free_core(const luw_solver* s, uint32_t lo[3], uint32_t hi[3]) {
	const KParams& k = s->kp;
	const int64_t N[3] = { (int64_t)k.Nx, (int64_t)k.Ny, (int64_t)k.Nz };
	int64_t l[3] = { 0, 0, 0 }, h[3] = { N[0], N[1], N[2] };
	if(k.buffer_active) {
		const int64_t nb = (int64_t)k.buffer_N;
		if(k.downstream_face!=1u&&k.has_w) l[0] = std::max<int64_t>(l[0], nb-k.Ox+1);
		if(k.downstream_face!=2u&&k.has_e) h[0] = std::min<int64_t>(h[0], (int64_t)k.Nxg-1-nb-k.Ox);
		if(k.downstream_face!=3u&&k.has_s) l[1] = std::max<int64_t>(l[1], nb-k.Oy+1);
		if(k.downstream_face!=4u&&k.has_n) h[1] = std::min<int64_t>(h[1], (int64_t)k.Nyg-1-nb-k.Oy);
		if(k.has_t) h[2] = std::min<int64_t>(h[2], (int64_t)k.Nzg-1-nb-k.Oz);
	}
	if(k.sponge_active&&k.has_t) h[2] = std::min<int64_t>(h[2], (int64_t)k.Nzg-1-(int64_t)k.sponge_N-k.Oz);
	for(int a=0; a<3; a++) {
		lo[a] = (uint32_t)std::min<int64_t>(std::max<int64_t>(l[a], 0), N[a]);
		hi[a] = (uint32_t)std::min<int64_t>(std::max<int64_t>(h[a], (int64_t)lo[a]), N[a]);
	}
}
// The nudging / sponge zones as cell ranges of this domain (KParams zw_lo ... zp_n): the conditions of FX/kernel.cpp:1537-1541,1598 -- the term is on,
// the domain owns the face, it is not the downstream one, 0 <= distance <= Nbuf (sponge: 0 <= layer < Nsponge) -- solved for the local coordinate
// and clipped to the domain.  n = 0: no cell.
static void set_zone_ranges(KParams& k) {
	auto range = [](const bool on, const int64_t lo, const int64_t hi, const int64_t N, uint32_t& zlo, uint32_t& zn) {
		const int64_t a = std::max<int64_t>(lo, 0), b = std::min<int64_t>(hi, N-1);
		if(on&&b>=a) { zlo = (uint32_t)a; zn = (uint32_t)(b-a+1); } else { zlo = 0u; zn = 0u; }
	};
	const int64_t nb = (int64_t)k.buffer_N;
	const bool buf = k.buffer_active!=0u;
	range(buf&&k.downstream_face!=1u&&k.has_w, k.west_x, k.west_x+nb, k.Nx, k.zw_lo, k.zw_n);
	range(buf&&k.downstream_face!=2u&&k.has_e, k.east_x-nb, k.east_x, k.Nx, k.ze_lo, k.ze_n);
	range(buf&&k.downstream_face!=3u&&k.has_s, k.south_y, k.south_y+nb, k.Ny, k.zs_lo, k.zs_n);
	range(buf&&k.downstream_face!=4u&&k.has_n, k.north_y-nb, k.north_y, k.Ny, k.zn_lo, k.zn_n);
	range(buf&&k.has_t, k.top_z-nb, k.top_z, k.Nz, k.zt_lo, k.zt_n);
	range(k.sponge_active&&k.has_t, (int64_t)k.top_z-(int64_t)k.sponge_N, (int64_t)k.top_z-1, k.Nz, k.zp_lo, k.zp_n);
}
// what can push the cells of box b (collide_cell_pk): re-evaluated per launch, so luw_set_f / luw_set_coriolis take effect at once.
// (Cutting a box that reaches into the zones along their boundaries -- specialised kernel on the zone-free core, general kernels on six slabs
// around it -- was built and measured on the 512^3 urban tile with its 80-cell nudging zones and 100-layer sponge: 2.36-2.38 ms in one
// launch, 2.41-2.46 ms cut; the core is a third of the cells there and the slabs cost more than it gains.  One launch per box it stays.)
static int box_force_mode(const luw_solver* s, const Box& b) {
	const KParams& k = s->kp;
	if(k.has_F) return PAIR_FORCE_ANY;
	uint32_t lo[3], hi[3];
	force_free_core(s, lo, hi);
	if(b.x0<lo[0]||b.x1>hi[0]||b.y0<lo[1]||b.y1>hi[1]||b.z0<lo[2]||b.z1>hi[2]) return PAIR_FORCE_ANY;
	return (k.coriolis||k.fx!=0.0f||k.fy!=0.0f||k.fz!=0.0f) ? PAIR_FORCE_UNIFORM : PAIR_FORCE_NONE;
}
// ---- x-face output of the step kernels (luw_set_x_face_buffers).  A launch takes the instantiation with the output when the buffers are set, x is split
// and its box holds the first or the last owned x column; it "covers" a column when the box also spans every non-halo (y, z) of it.  When the launches of
// a step have covered both columns, luw_enqueue_extract_fi(direction 0) on the same buffers has nothing left to do.
static bool xface_wanted(const luw_solver* s, const Box& b) {
	if(!s->xf_p||!s->xf_m||!s->kp.halo_x||s->cfg.Nx<4u) return false;
	return (b.x0<=1u&&b.x1>1u)||(b.x0<=s->cfg.Nx-2u&&b.x1>s->cfg.Nx-2u);
}
static void xface_covered(luw_solver* s, const Box& b) {
	if(s->xf_t!=s->t) { s->xf_t = s->t; s->xf_cover = 0u; }
	const bool spans = b.y0<=s->kp.halo_y&&b.y1>=s->cfg.Ny-s->kp.halo_y&&b.z0<=s->kp.halo_z&&b.z1>=s->cfg.Nz-s->kp.halo_z;
	if(!spans) return;
	if(b.x0<=1u&&b.x1>1u) s->xf_cover |= 2u;                            // first owned column: the face towards -x
	if(b.x0<=s->cfg.Nx-2u&&b.x1>s->cfg.Nx-2u) s->xf_cover |= 1u;        // last owned column: the face towards +x
}
// ---------------------------------------------------------------- the kernel instantiations, as tables
// Every stream_collide variant the library carries is one row: what it is for (the key the launchers look up) and the function that launches its two
// time-parity instances.  Nothing else instantiates the step kernels.
struct LaunchGeom { dim3 grid, block; int xa; uint32_t lds; };

// ---- k_stream_collide_s: one cell per lane
// mode 0: step, 4: step + thermal lattice; 1, 2, 3: A/B variants (tools build)
struct ScalarKey { uint8_t ddf_bytes; int mode; int nt; bool flat, stats, noforce, native, xface; };
typedef void (*ScalarLaunch)(luw_solver*, const Box&, const LaunchGeom&, int write_fields, const StatsArgs&);
template<typename T, int MODE, int NT, bool FLAT, bool STATS, bool NOFORCE, bool NATIVE=false,
	bool XFACE=false> static void scalar_instance(luw_solver* s, const Box& b, const LaunchGeom& g, const int wf, const StatsArgs& S) {
	T* const fi = (T*)s->d_fi; T* const gi = MODE==4 ? (T*)s->d_gi : nullptr; float* const Tf = MODE==4 ? s->d_T : nullptr;
	T* const xp = XFACE ? (T*)s->xf_p : nullptr; T* const xm = XFACE ? (T*)s->xf_m : nullptr;
	if(s->t&1ull) hipLaunchKernelGGL((k_stream_collide_s<T, 1, MODE, NT, FLAT, STATS, NOFORCE, NATIVE, XFACE>), g.grid, g.block, 0, s->stream, s->kp, b, g.xa,
		fi, s->d_rho, s->d_u, s->d_flags, s->d_F, wf, gi, Tf, S, xp, xm);
	else hipLaunchKernelGGL((k_stream_collide_s<T, 0, MODE, NT, FLAT, STATS, NOFORCE, NATIVE, XFACE>), g.grid, g.block, 0, s->stream, s->kp, b, g.xa, fi,
		s->d_rho, s->d_u, s->d_flags, s->d_F, wf, gi, Tf, S, xp, xm);
}
struct ScalarRow { ScalarKey key; ScalarLaunch launch; const char* what; };
static const ScalarRow scalar_table[] = {
	//  bytes mode nt flat   stats  noforce
	{ { 4u, 0, 2, true,  false, false }, scalar_instance<float, 0, 2, true, false, false>,
		"FP32 product kernel, flat addressing (planes within 32-bit byte offsets)" },
	{ { 4u, 0, 2, false, false, false }, scalar_instance<float, 0, 2, false, false, false>,      "FP32 product kernel, row addressing (any plane size)" },
	{ { 4u, 0, 2, true,  true,  false }, scalar_instance<float, 0, 2, true, true, false>,        "FP32, sampled step (fused Welford update)" },
	{ { 4u, 0, 2, false, true,  false }, scalar_instance<float, 0, 2, false, true, false>,       "FP32, sampled step, row addressing" },
	{ { 4u, 4, 2, true,  false, false }, scalar_instance<float, 4, 2, true, false, false>,       "FP32 + thermal lattice" },
	{ { 4u, 4, 2, false, false, false }, scalar_instance<float, 4, 2, false, false, false>,      "FP32 + thermal lattice, row addressing" },
	{ { 2u, 0, 2, false, false, false }, scalar_instance<uint16_t, 0, 2, false, false, false>,
		"FP16C one-cell kernel (rows too narrow / unaligned for the pair kernel)" },
	{ { 2u, 0, 2, false, false, true  }, scalar_instance<uint16_t, 0, 2, false, false, true>,    "FP16C one-cell kernel, force-free box: 7 waves per SIMD" },
	{ { 2u, 0, 2, false, true,  false }, scalar_instance<uint16_t, 0, 2, false, true, false>,    "FP16C one-cell kernel, sampled step" },
	{ { 2u, 4, 2, false, false, false }, scalar_instance<uint16_t, 4, 2, false, false, false>,   "FP16C one-cell kernel + thermal lattice" },
	{ { 2u, 4, 2, false, false, true  }, scalar_instance<uint16_t, 4, 2, false, false, true>,    "FP16C one-cell kernel + thermal lattice, force-free box" },
	// x-split domains, boxes that hold the first / last owned x column: the same kernels with the x-face output (luw_set_x_face_buffers)
	{ { 4u, 0, 2, true,  false, false, false, true }, scalar_instance<float, 0, 2, true, false, false, false, true>,  "FP32 + x-face output" },
	{ { 4u, 0, 2, false, false, false, false, true }, scalar_instance<float, 0, 2, false, false, false, false, true>, "FP32, row addressing + x-face output" },
	{ { 2u, 0, 2, false, false, false, true }, scalar_instance<uint16_t, 0, 2, false, false, false, true>, "FP16C one-cell kernel, native arithmetic" },
	{ { 2u, 4, 2, false, false, false, true }, scalar_instance<uint16_t, 4, 2, false, false, false, true>,
		"FP16C one-cell kernel + thermal lattice, native arithmetic" },
#ifdef LUW_AB_KERNELS   // tools build: measurement-only and A/B variants
	{ { 4u, 1, 1, true,  false, false }, scalar_instance<float, 1, 1, true, false, false>,       "A/B: no collision" },
	{ { 4u, 1, 1, false, false, false }, scalar_instance<float, 1, 1, false, false, false>,      "A/B: no collision, row addressing" },
	{ { 4u, 2, 1, true,  false, false }, scalar_instance<float, 2, 1, true, false, false>,       "A/B: x+1 neighbours replaced by x" },
	{ { 4u, 2, 1, false, false, false }, scalar_instance<float, 2, 1, false, false, false>,      "A/B: no shift, row addressing" },
	{ { 4u, 0, 0, true,  false, false }, scalar_instance<float, 0, 0, true, false, false>,       "A/B: default cache policy" },
	{ { 4u, 0, 0, false, false, false }, scalar_instance<float, 0, 0, false, false, false>,      "A/B: default cache policy, row addressing" },
	{ { 4u, 0, 1, true,  false, false }, scalar_instance<float, 0, 1, true, false, false>,       "A/B: non-temporal on all planes" },
	{ { 4u, 0, 1, false, false, false }, scalar_instance<float, 0, 1, false, false, false>,      "A/B: non-temporal on all planes, row addressing" },
	{ { 4u, 3, 2, true,  false, false }, scalar_instance<float, 3, 2, true, false, false>,       "A/B: general path only" },
	{ { 4u, 3, 2, false, false, false }, scalar_instance<float, 3, 2, false, false, false>,      "A/B: general path only, row addressing" },
	{ { 2u, 1, 1, false, false, false }, scalar_instance<uint16_t, 1, 1, false, false, false>,   "A/B: FP16C no collision" },
	{ { 2u, 2, 1, false, false, false }, scalar_instance<uint16_t, 2, 1, false, false, false>,   "A/B: FP16C no shift" },
	{ { 2u, 0, 0, false, false, false }, scalar_instance<uint16_t, 0, 0, false, false, false>,   "A/B: FP16C default cache policy" },
	{ { 2u, 0, 1, false, false, false }, scalar_instance<uint16_t, 0, 1, false, false, false>,   "A/B: FP16C non-temporal on all planes" },
	{ { 2u, 3, 2, false, false, false }, scalar_instance<uint16_t, 3, 2, false, false, false>,   "A/B: FP16C general path only" },
#endif
};
static int launch_scalar(luw_solver* s, const Box& b, const int write_fields, const StatsArgs* st = nullptr) {
	LaunchGeom g{};
	g.xa = (int)b.x0-(int)((b.x0+64u-s->kp.halo_x)&63u); // block start of the line that holds b.x0 (see lead_alloc)
	const uint32_t nx = (uint32_t)((int)b.x1-g.xa), bx = row_block(nx);
	g.grid = dim3((nx+bx-1u)/bx, b.y1-b.y0, b.z1-b.z0); g.block = dim3(bx);
	ScalarKey k{ (uint8_t)s->ddf_bytes, 0, 2, false, st!=nullptr, false, false, false };
	// FLAT addressing (one 32-bit byte offset per neighbour within a plane) for FP32 lattices whose planes fit it; the row form otherwise.  In-plane
	// offsets span the lattice part Px*Ny*Nz of a plane only -- the skew behind it is stride, never addressed -- so 1024^3 with its 2^32-byte planes
	// still qualifies (largest offset 2^32 - 4).  LUW_ADDR_ROW: the row form also where the flat form would do (both are product code, same values)
	const bool force_row = tuning().addr_row;
	k.flat = s->ddf_bytes==4u && (uint64_t)s->kp.Px*s->cfg.Ny*s->cfg.Nz*4ull<=(1ull<<32) && !force_row;
	// FP16C, nothing can push the cells of this box: the instantiation without the force assembly (69 / 76 VGPRs).  LUW_PAIR_GENERAL: never (test aid)
	const bool general_only = tuning().pair_general;
	k.noforce = s->ddf_bytes==2u && !st && !general_only && box_force_mode(s, b)==PAIR_FORCE_NONE;
	if(s->d_gi&&!st) k.mode = 4; // thermal lattice on: the product kernel plus the D3Q7 cell update
	// native arithmetic (FP16C, plain steps): one instantiation for every box
	if(s->ddf_bytes==2u&&!st&&(s->cfg.options&LUW_OPT_NATIVE_ARITH)!=0u) { k.native = true; k.noforce = false; }
	// x-face output: FP32 plain steps on a box that holds a border column
	k.xface = s->ddf_bytes==4u && k.mode==0 && !st && xface_wanted(s, b);
#ifdef LUW_AB_KERNELS
	if(!st&&!s->d_gi) switch(s->kernel) {
		case LUW_KERNEL_EXP_COPY: k.mode = 1; k.nt = 1; k.noforce = false; break;
		case LUW_KERNEL_EXP_NOSHIFT: k.mode = 2; k.nt = 1; k.noforce = false; break;
		case LUW_KERNEL_SCALAR_CACHED: k.nt = 0; k.noforce = false; break;
		case LUW_KERNEL_SCALAR_NT_ALL: k.nt = 1; k.noforce = false; break;
		case LUW_KERNEL_SCALAR_GENERAL: k.mode = 3; k.noforce = false; break;
		default: break;
	}
#endif
	for(const ScalarRow& r : scalar_table) {
		const ScalarKey& q = r.key;
		if(q.ddf_bytes==k.ddf_bytes&&q.mode==k.mode&&q.nt==k.nt&&q.flat==k.flat&&q.stats==k.stats&&q.noforce==k.noforce&&q.native==k.native&&q.xface==k.xface) {
			r.launch(s, b, g, write_fields, st ? *st : StatsArgs{});
			if(k.xface) xface_covered(s, b);
			return LUW_OK;
		}
	}
	return fail(LUW_ERR_STATE, "stream_collide: this library carries no one-cell kernel for the requested combination");
}

// ---- k_stream_collide_p: FP16C, two cells per lane
struct PairKey { int mode; bool stats; int force; bool park, thermal, native, xface; };   // mode 1: memory path only (tools build)
typedef void (*PairLaunch)(luw_solver*, const Box&, const LaunchGeom&, int write_fields, const StatsArgs&);
template<int MODE, bool STATS, int FORCE, bool PARK, bool THERMAL, bool NATIVE=false,
	bool XFACE=false> static void pair_instance(luw_solver* s, const Box& b, const LaunchGeom& g, const int wf, const StatsArgs& S) {
	uint16_t* const fi = (uint16_t*)s->d_fi; uint16_t* const gi = THERMAL ? (uint16_t*)s->d_gi : nullptr; float* const Tf = THERMAL ? s->d_T : nullptr;
	uint16_t* const xp = XFACE ? (uint16_t*)s->xf_p : nullptr; uint16_t* const xm = XFACE ? (uint16_t*)s->xf_m : nullptr;
	if(s->t&1ull) hipLaunchKernelGGL((k_stream_collide_p<1, MODE, STATS, FORCE, PARK, THERMAL, NATIVE, XFACE>), g.grid, g.block, g.lds, s->stream, s->kp, b,
		fi, s->d_rho, s->d_u, s->d_flags, s->d_F, wf, S, gi, Tf, xp, xm);
	else hipLaunchKernelGGL((k_stream_collide_p<0, MODE, STATS, FORCE, PARK, THERMAL, NATIVE, XFACE>), g.grid, g.block, g.lds, s->stream, s->kp, b, fi,
		s->d_rho, s->d_u, s->d_flags, s->d_F, wf, S, gi, Tf, xp, xm);
}
struct PairRow { PairKey key; PairLaunch launch; const char* what; };
static const PairRow pair_table[] = {
	//  mode stats  force               park   thermal
	{ { 0, false, PAIR_FORCE_NONE,    false, false }, pair_instance<0, false, PAIR_FORCE_NONE, false, false>,
		"nothing can push the cells of the box: no force path, 5 waves per SIMD" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false>, "volume force / Coriolis only, 5 waves" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false }, pair_instance<0, false, PAIR_FORCE_ANY, true, false>,
		"general (zones, force field): second cell's values parked in LDS, 5 waves" },
	{ { 0, false, PAIR_FORCE_ANY,     false, false }, pair_instance<0, false, PAIR_FORCE_ANY, false, false>,
		"general, everything in registers, 4 waves (LUW_PAIR_PARK=0: A/B and test aid)" },
	{ { 0, true,  PAIR_FORCE_ANY,     false, false }, pair_instance<0, true, PAIR_FORCE_ANY, false, false>,      "sampled step (fused Welford update)" },
	{ { 0, false, PAIR_FORCE_NONE,    true,  true  }, pair_instance<0, false, PAIR_FORCE_NONE, true, true>,      "+ thermal lattice, force-free box" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  true  }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, true>,   "+ thermal lattice, uniform forces" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  true  }, pair_instance<0, false, PAIR_FORCE_ANY, true, true>,       "+ thermal lattice, general" },
	// LUW_OPT_NATIVE_ARITH: the same six in the hardware's own arithmetic (collide_cell_pk_native); sampled steps keep the exact kernel
	{ { 0, false, PAIR_FORCE_NONE,    false, false, true }, pair_instance<0, false, PAIR_FORCE_NONE, false, false, true>,      "native: force-free box" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false, true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false, true>,   "native: uniform forces" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false, true }, pair_instance<0, false, PAIR_FORCE_ANY, true, false, true>,        "native: general" },
	{ { 0, false, PAIR_FORCE_NONE,    true,  true,  true }, pair_instance<0, false, PAIR_FORCE_NONE, true, true, true>,        "native + thermal, force-free" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  true,  true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, true, true>,     "native + thermal, uniform" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  true,  true }, pair_instance<0, false, PAIR_FORCE_ANY, true, true, true>,         "native + thermal, general" },
	// x-split domains, boxes that hold the first / last owned x column: x-face output (luw_set_x_face_buffers), exact and native
	{ { 0, false, PAIR_FORCE_NONE,    false, false, false, true }, pair_instance<0, false, PAIR_FORCE_NONE, false, false, false, true>,
		"force-free + x-face" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false, false, true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false, false, true>, "uniform + x-face" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false, false, true }, pair_instance<0, false, PAIR_FORCE_ANY, true, false, false, true>,      "general + x-face" },
	{ { 0, false, PAIR_FORCE_NONE,    false, false, true,  true }, pair_instance<0, false, PAIR_FORCE_NONE, false, false, true, true>,     "native + x-face" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false, true,  true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false, true, true>,  "native + x-face" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false, true,  true }, pair_instance<0, false, PAIR_FORCE_ANY, true, false, true, true>,       "native + x-face" },
#ifdef LUW_AB_KERNELS
	{ { 1, false, PAIR_FORCE_ANY,     false, false }, pair_instance<1, false, PAIR_FORCE_ANY, false, false>,
		"A/B: the kernel's memory path alone (LUW_PAIR_COPY)" },
	{ { 0, false, PAIR_FORCE_NONE,    true,  false }, pair_instance<0, false, PAIR_FORCE_NONE, true, false>,
		"A/B: force-free with PARK (7 waves: no gain, profiles/r03_pair_park_ab.txt)" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  false }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, false>,
		"A/B: uniform forces with PARK (6 waves: slower)" },
#endif
};
static int launch_pair(luw_solver* s, const Box& b, const int write_fields, const StatsArgs* st = nullptr) {
	LaunchGeom g{};
	const uint32_t nx = (b.x1-b.x0+1u)/2u;                         // an odd count only when the box ends at an odd Nx: the last lane owns one cell
	const uint32_t bx = row_block(nx);
	g.grid = dim3((nx+bx-1u)/bx, b.y1-b.y0, b.z1-b.z0); g.block = dim3(bx);
	const bool general_only = tuning().pair_general;   // test aid: the general kernel also where a specialisation would do (same values)
	PairKey k{ 0, st!=nullptr, (st||general_only) ? PAIR_FORCE_ANY : box_force_mode(s, b), false, s->d_gi!=nullptr };
	// PARK (luw_kernels_step.hpp): the lane's second set of values waits in LDS instead of in registers.  Measured interleaved on MI355X
	// (profiles/r03_pair_park_ab.txt): it pays where the registers cost a wave of occupancy that matters -- the general kernel, 109 -> 91 VGPRs,
	// 4 -> 5 waves per SIMD: urban 512^3 tile 2.344 -> 2.276 ms, + Coriolis 2.465 -> 2.375 -- and not above five waves (force-free 86 -> 68
	// VGPRs, 7 waves: 3.50 -> 3.49 ms; uniform forces 96 -> 78, 6 waves: 3.78 -> 3.91 ms on 1024x1024x256); the thermal variants always park.
	constexpr unsigned park_modes = 1u<<PAIR_FORCE_ANY;
	k.park = k.thermal || (!st && (park_modes&(1u<<k.force))!=0u);
	// native arithmetic: plain steps of the product kernel (a sampled step runs the exact kernel: its values differ in rounding only)
	k.native = (s->cfg.options&LUW_OPT_NATIVE_ARITH)!=0u && !st;
	if(k.native&&!k.thermal) k.park = k.force==PAIR_FORCE_ANY;
	// x-face output: plain steps of the D3Q19 lattice with the product's park choice, on a box that holds a border column
	k.xface = !st && !k.thermal && k.mode==0 && k.park==(k.force==PAIR_FORCE_ANY) && xface_wanted(s, b);
#ifdef LUW_AB_KERNELS
	const bool copy_only = tuning().ab_pair_copy;   // tools build, measurement aid: the kernel's memory path alone (no physics)
	if(copy_only&&!st&&!k.thermal) k = PairKey{ 1, false, PAIR_FORCE_ANY, false, false };
#endif
	g.lds = k.park ? (bx/64u)*pair_park_bytes_per_wave(k.thermal, (k.mode==0&&!k.stats) ? k.force : PAIR_FORCE_NONE) : 0u;
	for(const PairRow& r : pair_table) {
		const PairKey& q = r.key;
		if(q.mode==k.mode&&q.stats==k.stats&&q.force==k.force&&q.park==k.park&&q.thermal==k.thermal&&q.native==k.native&&q.xface==k.xface) {
			r.launch(s, b, g, write_fields, st ? *st : StatsArgs{});
			if(k.xface) xface_covered(s, b);
			return LUW_OK;
		}
	}
	return fail(LUW_ERR_STATE, "stream_collide: this library carries no pair kernel for the requested combination");
}

// Kernel choice.  LUW_KERNEL_AUTO = the scalar kernel (FP32: 39.5k MLUPS at 512^3; vector kernels 20-29k) and, for FP16C rows
// wide enough, the pair kernel (profiles/r01_kernel_ab.md).  The other kernels stay selectable for A/B runs.
// can a sampled step carry the Welford update itself?  Product kernels only (scalar / pair, no thermal lattice: its T statistics
// stay with k_stats_accumulate); LUW_FUSE_STATS=0 keeps the separate kernel (A/B and test aid)
static bool can_fuse_stats(const luw_solver* s) {
	return tuning().fuse_stats && !s->d_gi && (s->kernel==LUW_KERNEL_AUTO||s->kernel==LUW_KERNEL_SCALAR||s->kernel==LUW_KERNEL_PAIR);
}
static int launch_stream_collide(luw_solver* s, const Box& b, const int write_fields, const StatsArgs* st = nullptr) {
	if(b.x0>=b.x1||b.y0>=b.y1||b.z0>=b.z1) return LUW_OK; // empty box
	if(b.x1>s->cfg.Nx||b.y1>s->cfg.Ny||b.z1>s->cfg.Nz) return fail(LUW_ERR_INVALID, "stream_collide: box exceeds the local lattice");
	if(b.y1-b.y0>65535u||b.z1-b.z0>65535u) return fail(LUW_ERR_INVALID, "stream_collide: box too large for the launch geometry");
	const bool fp16 = s->ddf_bytes==2u;
	uint32_t k = s->kernel;
	// AUTO: the scalar kernel, except FP16C rows of at least one wave of pairs, which take the pair kernel (dword accesses, packed
	// FP32 collision: 69.0k vs 67.2k MLUPS at 512^3, 63.1k vs 61.1k with Coriolis)
	constexpr uint32_t pair_min = 128u;   // (round 2: 256)
	if(k==LUW_KERNEL_AUTO) k = (fp16 && b.x1-b.x0>=pair_min) ? LUW_KERNEL_PAIR : LUW_KERNEL_SCALAR;
	if(s->d_gi&&(!fp16||st)) k = LUW_KERNEL_SCALAR; // FP32 / sampled steps: the thermal cell update of the one-cell kernel
#ifdef LUW_AB_KERNELS
	// the vector kernels assume rows that start on a 16-byte boundary at x = 0
	if(s->kp.halo_x&&(k==LUW_KERNEL_VEC4||k==LUW_KERNEL_VEC2||k==LUW_KERNEL_VEC1)) k = LUW_KERNEL_SCALAR;
#endif
	// pair kernel: FP16C; pairs start on a 4-byte boundary -- at even x, or at odd x when x is split (the row's lead pad then puts
	// x = 1 on a line start, lead_alloc); the range holds whole pairs, except that it may end at an odd Nx of an unsplit row (the
	// last cell then pairs with the row padding)
	if(k==LUW_KERNEL_PAIR) {
		const bool starts_aligned = ((b.x0+s->kp.halo_x)&1u)==0u;
		const bool whole_pairs = ((b.x1-b.x0)&1u)==0u || (!s->kp.halo_x && b.x1==s->cfg.Nx);
		if(!fp16||!starts_aligned||!whole_pairs) k = LUW_KERNEL_SCALAR;
	}
	if(st&&k!=LUW_KERNEL_PAIR&&k!=LUW_KERNEL_SCALAR) return fail(LUW_ERR_STATE, "stream_collide: this kernel has no fused statistics");
	if(k==LUW_KERNEL_PAIR) { if(int e = launch_pair(s, b, write_fields, st)) return e; }
	else if(st) { if(int e = launch_scalar(s, b, write_fields, st)) return e; }
#ifdef LUW_AB_KERNELS
	else if(k==LUW_KERNEL_VEC4) { if(fp16) launch_vec<uint16_t, 4>(s, b, write_fields); else launch_vec<float, 4>(s, b, write_fields); }
	else if(k==LUW_KERNEL_VEC2) { if(fp16) launch_vec<uint16_t, 2>(s, b, write_fields); else launch_vec<float, 2>(s, b, write_fields); }
	else if(k==LUW_KERNEL_VEC1) { if(fp16) launch_vec<uint16_t, 1>(s, b, write_fields); else launch_vec<float, 1>(s, b, write_fields); }
#endif
	else { if(int e = launch_scalar(s, b, write_fields)) return e; }
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}

// launch helper: picks the template instance for (storage type, lattice, direction)
template<bool G, bool INSERT> static void launch_transfer(luw_solver* s, const uint32_t direction, void* buf_p, void* buf_m) {
	const uint32_t A = (uint32_t)luw_get_area(s, direction);
	const dim3 grid((A+255u)/256u), block(256);
	const uint32_t odd = (uint32_t)(s->t&1ull);
	void* lat = G ? s->d_gi : s->d_fi;
	#define LUW_TR(TT, DD) do { \
		if constexpr(INSERT) hipLaunchKernelGGL((k_insert_fi<TT, G, DD>), grid, block, 0, s->stream, s->kp, A, odd, (const TT*)buf_p, (const TT*)buf_m, (TT*)lat); \
		else hipLaunchKernelGGL((k_extract_fi<TT, G, DD>), grid, block, 0, s->stream, s->kp, A, odd, (TT*)buf_p, (TT*)buf_m, (const TT*)lat); \
	} while(0)
	if(s->ddf_bytes==2u) { if(direction==0u) LUW_TR(uint16_t, 0); else if(direction==1u) LUW_TR(uint16_t, 1); else LUW_TR(uint16_t, 2); }
	else { if(direction==0u) LUW_TR(float, 0); else if(direction==1u) LUW_TR(float, 1); else LUW_TR(float, 2); }
	#undef LUW_TR
}

// Where the driver places the DDF array physically changes the step time of this 19-stream kernel by 10-14 % on MI355X: allocations of the same size come
// out in classes that last for the life of the allocation (512^3 FP32: 3.3 or 3.75 ms per step; tools/placement_probe.py, tools/chunk_study.sh), and WHICH
// kind of allocation is of the fast class depends on the box: 1 GiB chunks on most, 2 GiB chunks or a plain hipMalloc on others
// (profiles/r03_chunk_study_slow_box.txt).  Large solvers therefore time the real kernel on their DDF array -- the box the step launches (non-halo cells),
// zero DDFs = rest state, flags 0 = all fluid: a valid, full-cost step -- and, while the rate is under the bar of the fast class, try the OTHER kinds, once
// each: 2 GiB chunks, hipMalloc, 512 MiB chunks.  Bounded: at most three further candidates, ONE extra array alive at a time, every loser released before
// the next candidate is mapped and nothing of the search left when luw_create returns (placement_kept / placement_tbps / placement_tried say what
// happened; bench.py prints them). Skipped for small lattices, for planes of 2 GiB and more (1024^3 runs alike on every kind,
// profiles/r02_placement_study.txt),
// when the device has no room for a second DDF array, and when the device is shared: other solvers of this process live on it (g_live_solvers), or the
// caller says so (luw_group_create for devices that host several domains: thread-local g_device_is_shared; rank processes sharing one GPU set
// LUW_TUNE_PLACEMENT=0) -- concurrent probes would time each other.
static std::atomic<int> g_live_solvers[64];
static thread_local bool g_device_is_shared = false;
constexpr size_t PLACEMENT_UNSET = ~(size_t)0u-1u;     // g_placement_kind: chunk size the process's search kept for this device (0: hipMalloc)
static std::atomic<size_t> g_placement_kind[64];
static struct PlacementKindInit { PlacementKindInit() { for(auto& k : g_placement_kind) k.store(PLACEMENT_UNSET); } } g_placement_kind_init;
static const char* dev_block_kind(const DevBlock& b) {
	if(b.chunks.empty()) return "hipMalloc";
	return b.chunk_bytes>=(2048ull<<20) ? "2 GiB chunks" : b.chunk_bytes>=(1024ull<<20) ? "1 GiB chunks" : b.chunk_bytes>=(512ull<<20) ? "512 MiB chunks"
		: "chunks under 512 MiB";
}
static int tune_ddf_placement(luw_solver* s) {
	const Tuning& T = tuning();
	const size_t elems = 19ull*s->kp.Np, bytes = elems*s->ddf_bytes;
	const bool mapped = !s->raw.front().chunks.empty();
	if(s->placement_kept=="default (no search)") s->placement_kept = std::string(dev_block_kind(s->raw.front()))+" (no search)";
	constexpr int NALT = 3;
	const size_t alternatives[NALT] = { 2048ull<<20, 0u, 512ull<<20 };   // chunk sizes behind the default's 1 GiB (0: hipMalloc)
	const int candidates = std::min(T.placement_candidates>=0 ? T.placement_candidates : 1+NALT, 1+NALT);
	if(bytes<(1ull<<30)||candidates<2||!mapped) return LUW_OK;
	if(T.placement_candidates<0&&s->kp.Np*s->ddf_bytes>(3ull<<29)) return LUW_OK;
	if(g_device_is_shared||(s->cfg.device<64&&g_live_solvers[s->cfg.device].load()>1)) return LUW_OK;
	// ONE search per process and device: which kind of allocation is the fast one is a property of the machine (and of the process's allocation history),
	// not of the solver -- later large solvers of the process are allocated as the first one's winner straight away (luw_create, g_placement_kind)
	if(s->cfg.device<64&&g_placement_kind[s->cfg.device].load()!=PLACEMENT_UNSET) return LUW_OK;
	// the box a step launches: every non-halo cell (the FP16C pair kernel needs its pairs to start at the first owned cell of an x-split row)
	const Box box = { s->kp.halo_x, s->cfg.Nx-s->kp.halo_x, s->kp.halo_y, s->cfg.Ny-s->kp.halo_y, s->kp.halo_z, s->cfg.Nz-s->kp.halo_z };
	struct Events { hipEvent_t e0 = nullptr, e1 = nullptr; ~Events() { if(e0) (void)hipEventDestroy(e0); if(e1) (void)hipEventDestroy(e1); } } ev;
	HIP_TRY(hipEventCreate(&ev.e0)); HIP_TRY(hipEventCreate(&ev.e1));
	auto step_ms = [&](float& ms) -> int { // two steps (both parities) after one untimed
		struct Restore { luw_solver* s; ~Restore() { s->initialized = false; s->t = 0ull; } } restore{ s };
		s->initialized = true; s->t = 0ull;
		if(int e = launch_stream_collide(s, box, 0)) return e;
		s->t = 1ull;
		HIP_TRY(hipEventRecord(ev.e0, s->stream));
		if(int e = launch_stream_collide(s, box, 0)) return e;
		s->t = 2ull;
		if(int e = launch_stream_collide(s, box, 0)) return e;
		HIP_TRY(hipEventRecord(ev.e1, s->stream));
		HIP_TRY(hipEventSynchronize(ev.e1));
		HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
		return LUW_OK;
	};
	// a placement of the fast class moves this many algorithmic bytes per second through the probe (FP32 153, FP16C 77 B per update, + the thermal planes;
	// FP16C with zones: the general kernel is VALU-bound).  LUW_TUNE_FAST=<TB/s> overrides the bar (99: every candidate is tried)
	const double cells = (double)(box.x1-box.x0)*(double)(box.y1-box.y0)*(double)(box.z1-box.z0);
	const double probe_bytes = 2.0*((s->ddf_bytes==4u ? 153.0 : 77.0)+(s->d_gi ? 14.0*(double)s->ddf_bytes : 0.0))*cells;
	const double bar = T.placement_bar>0.0 ? T.placement_bar*1e12 : (s->ddf_bytes==4u ? 6.25e12 : (s->kp.buffer_active||s->kp.sponge_active) ? 5.0e12 : 6.1e12);
	auto rate = [&](const float ms) { return probe_bytes/((double)ms*1e-3); };
	float best_ms = 0.0f;
	if(int e = step_ms(best_ms)) return e;   // (the first probe of a process also ramps the GPU up: measured again)
	if(int e = step_ms(best_ms)) return e;
	s->placement_tried = 1;
	if(T.placement_verbose) fprintf(stderr, "luw: placement candidate 0 (%s): %.3f ms per 2 steps = %.2f TB/s\n", dev_block_kind(s->raw.front()), best_ms,
		rate(best_ms)*1e-12);
	for(int k=1; k<candidates&&rate(best_ms)<bar; k++) {
		size_t free_b = 0u, total_b = 0u;
		// room for ONE more array plus what the run may still allocate (statistics: 32 B per cell, staging, halo buffers)
		if(hipMemGetInfo(&free_b, &total_b)!=hipSuccess||free_b<bytes+40ull*s->kp.Np+(2ull<<30)) break;
		void* fi = nullptr;
		if(lead_alloc(s, &fi, elems, s->ddf_bytes, &alternatives[k-1])!=hipSuccess) { (void)hipGetLastError(); break; }
		DevBlock cand = std::move(s->raw.back()); s->raw.pop_back();
		void* const old_fi = s->d_fi;
		s->d_fi = fi;
		float ms = 0.0f;
		if(int e = step_ms(ms)) { s->d_fi = old_fi; dev_free(cand); return e; }
		s->placement_tried++;
		if(T.placement_verbose) fprintf(stderr, "luw: placement candidate %d (%s): %.3f ms per 2 steps = %.2f TB/s (best so far %.3f ms)\n", k,
			dev_block_kind(cand), ms, rate(ms)*1e-12, best_ms);
		if(ms<best_ms) { best_ms = ms; std::swap(s->raw.front(), cand); } // fi is the first lead_alloc of luw_create; cand now holds the loser
		else s->d_fi = old_fi;
		// the loser's memory goes before the next candidate comes; a mapped loser's (now empty) address range goes with the solver
		const bool was_mapped = !cand.chunks.empty();
		dev_free(cand, was_mapped);
		if(was_mapped) s->raw.push_back(std::move(cand));
	}
	s->placement_kept = dev_block_kind(s->raw.front()); s->placement_tbps = rate(best_ms)*1e-12;
	if(s->cfg.device<64) g_placement_kind[s->cfg.device].store(s->raw.front().chunks.empty() ? (size_t)0u : s->raw.front().chunk_bytes);
	// the probe steps left zeros, but be explicit
	HIP_TRY(hipMemsetAsync(s->raw.front().base, 0, std::min(s->raw.front().bytes, bytes+64u*s->ddf_bytes), s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

static bool kernel_selectable(const uint32_t k) {
#ifdef LUW_AB_KERNELS
	return k<=LUW_KERNEL_SCALAR_GENERAL||k==LUW_KERNEL_EXP_COPY||k==LUW_KERNEL_EXP_NOSHIFT;
#else
	return k==LUW_KERNEL_AUTO||k==LUW_KERNEL_SCALAR||k==LUW_KERNEL_PAIR;
#endif
}

extern "C" {

int luw_abi_version(void) { return LUW_ABI_VERSION; }
int luw_format_float9(float x, char* text, uint64_t size) {
	if(!text||size<24u) return fail(LUW_ERR_INVALID, "luw_format_float9: needs a buffer of at least 24 characters");
	format_decimal9(x, text, (size_t)size);
	return LUW_OK;
}
const char* luw_last_error(void) { return g_last_error.c_str(); }
int luw_device_count(int* count) {
	if(!count) return fail(LUW_ERR_INVALID, "luw_device_count: null argument");
	HIP_TRY(hipGetDeviceCount(count));
	return LUW_OK;
}

int luw_device_info(int device, char* name, uint64_t name_size, char* pci_bus_id, uint64_t pci_size, uint64_t* total_memory) {
	int ndev = 0;
	HIP_TRY(hipGetDeviceCount(&ndev));
	if(device<0||device>=ndev) return fail(LUW_ERR_INVALID, "luw_device_info: no such HIP device");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if(name&&name_size) snprintf(name, (size_t)name_size, "%s", prop.name);
	if(pci_bus_id&&pci_size) HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id, (int)std::min<uint64_t>(pci_size, 64u), device));
	if(total_memory) *total_memory = (uint64_t)prop.totalGlobalMem;
	return LUW_OK;
}
int luw_p2p_info(int device, int peer, int* can_access, int* performance_rank, int* native_atomics, uint32_t* link_type, uint32_t* hops) {
	int ndev = 0;
	HIP_TRY(hipGetDeviceCount(&ndev));
	if(device<0||device>=ndev||peer<0||peer>=ndev) return fail(LUW_ERR_INVALID, "luw_p2p_info: no such HIP device");
	int v = device==peer ? 1 : 0;
	if(can_access) { if(device!=peer) HIP_TRY(hipDeviceCanAccessPeer(&v, device, peer)); *can_access = v; }
	if(performance_rank) {
		v = 0;
		if(device!=peer&&hipDeviceGetP2PAttribute(&v, hipDevP2PAttrPerformanceRank, device, peer)!=hipSuccess) { (void)hipGetLastError(); v = -1; }
		*performance_rank = v;
	}
	if(native_atomics) {
		v = 1;
		if(device!=peer&&hipDeviceGetP2PAttribute(&v, hipDevP2PAttrNativeAtomicSupported, device, peer)!=hipSuccess) { (void)hipGetLastError(); v = -1; }
		*native_atomics = v;
	}
	uint32_t lt = 0u, hc = 0u;
	if(device!=peer&&hipExtGetLinkTypeAndHopCount(device, peer, &lt, &hc)!=hipSuccess) { (void)hipGetLastError(); lt = ~0u; hc = ~0u; }
	if(link_type) *link_type = lt;
	if(hops) *hops = hc;
	return LUW_OK;
}

void luw_destroy(luw_solver* s) {
	if(!s) return;
	(void)hipSetDevice(s->cfg.device);
	if(s->own_stream) (void)hipStreamSynchronize(s->own_stream);
	(void)luw_vk_inlet_detach(s); // side stream, its events, value buffers and tables
	for(DevBlock& r : s->raw) dev_free(r); // fi, rho, u, flags, F, statistics
	if(s->counted&&s->cfg.device>=0&&s->cfg.device<64) g_live_solvers[s->cfg.device]--;
	(void)hipFree(s->d_wbuf); (void)hipFree(s->d_sigma);
	(void)hipFree(s->d_gather_cell); (void)hipFree(s->d_gather_out);
	(void)hipFree(s->d_stage);
	(void)hipHostFree(s->h_rho); (void)hipHostFree(s->h_u); (void)hipHostFree(s->h_flags); (void)hipHostFree(s->h_F); (void)hipHostFree(s->h_T);
	if(s->own_stream) (void)hipStreamDestroy(s->own_stream);
	delete s;
}

int luw_create(const luw_config* cfg, luw_solver** out) {
	if(!cfg||!out) return fail(LUW_ERR_INVALID, "luw_create: null argument");
	*out = nullptr;
	const auto t_create = std::chrono::steady_clock::now();
	if(cfg->struct_size!=sizeof(luw_config)) return fail(LUW_ERR_INVALID, "luw_create: luw_config size mismatch (ABI)");
	if((uint64_t)cfg->Nx*cfg->Ny*cfg->Nz==0ull) return fail(LUW_ERR_INVALID, "Grid point number is 0."); // FX/lbm.cpp:1123
	if(cfg->Dx*cfg->Dy*cfg->Dz==0u) return fail(LUW_ERR_INVALID, "You specified 0 LBM grid domains."); // FX/lbm.cpp:1124
	if(cfg->nu==0.0f) return fail(LUW_ERR_INVALID, "Viscosity cannot be 0."); // FX/lbm.cpp:1141
	if(cfg->nu<0.0f) return fail(LUW_ERR_INVALID, "Viscosity cannot be negative."); // FX/lbm.cpp:1142
	if(cfg->ddf_format!=LUW_DDF_FP32&&cfg->ddf_format!=LUW_DDF_FP16C) return fail(LUW_ERR_INVALID, "luw_create: unknown ddf_format");
	if(!kernel_selectable(cfg->kernel))
		return fail(LUW_ERR_INVALID, "luw_create: this library has no such kernel (A/B and measurement-only variants exist in the tools build only)");
	if((cfg->Dx>1u&&cfg->Nx<3u)||(cfg->Dy>1u&&cfg->Ny<3u)||(cfg->Dz>1u&&cfg->Nz<3u))
		return fail(LUW_ERR_INVALID, "luw_create: split axes need at least one interior cell between the halo layers");
	if((cfg->options&LUW_OPT_TEMPERATURE)&&!(cfg->alpha>=0.0f)) return fail(LUW_ERR_INVALID, "luw_create: thermal diffusivity must not be negative");
	if(cfg->buffer_nudging_active&&cfg->buffer_n_cells==0u) return fail(LUW_ERR_INVALID, "luw_create: buffer_n_cells must be > 0");
	if(cfg->top_sponge_active&&cfg->sponge_n_cells==0u) return fail(LUW_ERR_INVALID, "luw_create: sponge_n_cells must be > 0");
	const uint32_t Px = (cfg->Nx+63u)&~63u; // rows are whole 256-byte blocks (see lead_alloc)
	// plane stride: the lattice plus a skew of an odd number of 64-element blocks.  With a bare power-of-two stride the 19 planes of a
	// cell sit at the same offset of 19 equally aligned regions (512^3 FP32: 3.84 ms against 3.29 ms).  How much skew is a matter of the
	// DRAM address mapping and was measured (tools/skew_study.sh, profiles/r02_skew_study.md): FP16C is flat from 33 blocks (4 KiB) up and
	// worse from 385 on some lattices; FP32 with 33 blocks (8 KiB, the round-1 value) depends on the GPU it lands on -- 512^3 3.30 / 3.45 /
	// 3.64 ms and 1024x512x256 3.48 / 3.76 ms on three boxes -- while 513 blocks (128 KiB + 256 B) gave 3.28-3.30 and 3.31-3.37 ms on all of
	// them (1024x1024x256: 6.61-6.75 ms either way).  LUW_PLANE_SKEW=<blocks> overrides (study aid).
	const uint64_t skew_env = tuning().plane_skew;
	const uint64_t skew_blocks = skew_env ? skew_env : cfg->ddf_format==LUW_DDF_FP16C ? 33ull : 513ull;
	const uint64_t Np = (uint64_t)Px*cfg->Ny*cfg->Nz+64ull*skew_blocks;
	if(Np>=(1ull<<32)) return fail(LUW_ERR_INVALID, "luw_create: more than 2^32 (padded) cells per domain are not supported (32-bit cell indices)");
	int ndev = 0;
	HIP_TRY(hipGetDeviceCount(&ndev));
	if(cfg->device<0||cfg->device>=ndev) return fail(LUW_ERR_INVALID, "luw_create: no such HIP device"); // FX/lbm.cpp:961-979
	HIP_TRY(hipSetDevice(cfg->device));

	luw_solver* s = new luw_solver();
	s->cfg = *cfg;
	if(cfg->device<64) { g_live_solvers[cfg->device]++; s->counted = true; }
	s->N = (uint64_t)cfg->Nx*cfg->Ny*cfg->Nz;
	s->ddf_bytes = cfg->ddf_format==LUW_DDF_FP16C ? 2u : 4u;
	s->kernel = cfg->kernel;
#ifdef LUW_AB_KERNELS
	// tools build: overrides the kernel choice of callers that expose none (the deck driver)
	if(tuning().ab_kernel>=0) s->kernel = (uint32_t)tuning().ab_kernel;
#endif
	KParams& k = s->kp;
	memset(&k, 0, sizeof(k));
	k.Nx = cfg->Nx; k.Ny = cfg->Ny; k.Nz = cfg->Nz; k.Px = Px; k.Np = (uint32_t)Np;
	k.halo_x = cfg->Dx>1u; k.halo_y = cfg->Dy>1u; k.halo_z = cfg->Dz>1u;
	k.Ox = cfg->Ox; k.Oy = cfg->Oy; k.Oz = cfg->Oz;
	k.w = literal_roundtrip(1.0f/(3.0f*cfg->nu+0.5f)); // FX/lbm.hpp:140, FX/lbm.cpp:664
	k.fx = cfg->fx; k.fy = cfg->fy; k.fz = cfg->fz;
	k.tau0 = 1.0f/k.w; k.tau0sq = k.tau0*k.tau0; k.half_tau0 = 0.5f*k.tau0;
	k.omx = cfg->omega_x; k.omy = cfg->omega_y; k.omz = cfg->omega_z; k.coriolis = k.omx!=0.0f||k.omy!=0.0f||k.omz!=0.0f;
	k.m2omx = -2.0f*k.omx; k.m2omy = -2.0f*k.omy; k.m2omz = -2.0f*k.omz;
	k.subgrid = (cfg->options&LUW_OPT_NO_SUBGRID) ? 0u : 1u;
	k.buffer_active = cfg->buffer_nudging_active ? 1u : 0u;
	k.buffer_N = cfg->buffer_n_cells; k.nudge_vertical = (uint32_t)cfg->buffer_nudge_vertical; k.downstream_face = (uint32_t)cfg->buffer_downstream_face_id;
	k.buffer_inv_tau = literal_roundtrip(cfg->buffer_inv_tau_lbmu);
	k.sponge_active = cfg->top_sponge_active ? 1u : 0u;
	k.sponge_N = cfg->sponge_n_cells;
	// FX/lbm.cpp:613-625
	k.Nxg = (cfg->Nx-2u*k.halo_x)*cfg->Dx; k.Nyg = (cfg->Ny-2u*k.halo_y)*cfg->Dy; k.Nzg = (cfg->Nz-2u*k.halo_z)*cfg->Dz;
	k.west_x = -cfg->Ox; k.east_x = (int)k.Nxg-1-cfg->Ox; k.south_y = -cfg->Oy; k.north_y = (int)k.Nyg-1-cfg->Oy; k.top_z = (int)k.Nzg-1-cfg->Oz;
	k.has_w = k.west_x>=0&&k.west_x<(int)cfg->Nx; k.has_e = k.east_x>=0&&k.east_x<(int)cfg->Nx;
	k.has_s = k.south_y>=0&&k.south_y<(int)cfg->Ny; k.has_n = k.north_y>=0&&k.north_y<(int)cfg->Ny;
	k.has_t = k.top_z>=0&&k.top_z<(int)cfg->Nz;
	set_zone_ranges(k);
	k.has_F = (cfg->options&LUW_OPT_FORCE_FIELD) ? 1u : 0u;
	k.w_T = (cfg->options&LUW_OPT_TEMPERATURE) ? literal_roundtrip(1.0f/(2.0f*cfg->alpha+0.5f)) : 0.0f; // FX/lbm.cpp:750

	auto oom = [&](const char* what) { luw_destroy(s); return fail(LUW_ERR_NOMEM, std::string("luw_create: allocation failed: ")+what); };
	if(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking)!=hipSuccess) return oom("stream");
	s->stream = s->own_stream;
	// (memset inside lead_alloc runs on the solver's own non-blocking stream: the legacy NULL stream does not order against it)
	{	// a DDF array of the size the placement search looks at, in a process whose search has already settled on a kind for this device: that kind
		const size_t kind = cfg->device<64 ? g_placement_kind[cfg->device].load() : PLACEMENT_UNSET;
		const bool reuse = kind!=PLACEMENT_UNSET && alloc_vmm_chunk()!=0u && 19ull*Np*s->ddf_bytes>=(1ull<<30) && Np*s->ddf_bytes<=(3ull<<29);
		if(lead_alloc(s, &s->d_fi, 19ull*Np, s->ddf_bytes, reuse ? &kind : nullptr)!=hipSuccess) return oom("fi");
		if(reuse) s->placement_kept = std::string(dev_block_kind(s->raw.front()))+" (the kind this process's first search kept)";
	}
	if(lead_alloc(s, (void**)&s->d_rho, Np, 4u)!=hipSuccess) return oom("rho");
	if(lead_alloc(s, (void**)&s->d_u, 3ull*Np, 4u)!=hipSuccess) return oom("u");
	if(lead_alloc(s, (void**)&s->d_flags, Np, 1u)!=hipSuccess) return oom("flags");
	if(k.has_F&&lead_alloc(s, (void**)&s->d_F, 3ull*Np, 4u)!=hipSuccess) return oom("F");
	if(cfg->options&LUW_OPT_TEMPERATURE) {
		if(lead_alloc(s, &s->d_gi, 7ull*Np, s->ddf_bytes)!=hipSuccess) return oom("gi");
		if(lead_alloc(s, (void**)&s->d_T, Np, 4u)!=hipSuccess) return oom("T");
		if(hipHostMalloc((void**)&s->h_T, s->N*4ull)!=hipSuccess) return oom("host T");
		for(uint64_t n=0ull; n<s->N; n++) s->h_T[n] = 1.0f; // T = Memory<float>(device, N, 1u, true, true, 1.0f), FX/lbm.cpp:304
	}
	if(hipHostMalloc((void**)&s->h_rho, s->N*4ull)!=hipSuccess) return oom("host rho");
	if(hipHostMalloc((void**)&s->h_u, 3ull*s->N*4ull)!=hipSuccess) return oom("host u");
	if(hipHostMalloc((void**)&s->h_flags, s->N)!=hipSuccess) return oom("host flags");
	if(k.has_F&&hipHostMalloc((void**)&s->h_F, 3ull*s->N*4ull)!=hipSuccess) return oom("host F");
	for(uint64_t n=0ull; n<s->N; n++) s->h_rho[n] = 1.0f; // Memory<float>(device, N, 1u, true, true, 1.0f), FX/lbm.cpp:286
	memset(s->h_u, 0, 3ull*s->N*4ull);
	memset(s->h_flags, 0, s->N);
	if(s->h_F) memset(s->h_F, 0, 3ull*s->N*4ull);
	if(hipStreamSynchronize(s->stream)!=hipSuccess) return oom("memset sync");
	// ramps of the nudging / sponge terms, evaluated on the host exactly like FX/kernel.cpp:1581-1583,1604-1606
	if(k.buffer_active) {
		std::vector<float> wb(k.buffer_N+2u);
		for(uint32_t d=0u; d<=k.buffer_N+1u; d++) {
			const float xi = 1.0f-(float)d/(float)k.buffer_N;
			float w_buf = sinf(1.5707963267948966f*xi);
			w_buf *= w_buf;
			wb[d] = w_buf;
		}
		if(hipMalloc((void**)&s->d_wbuf, wb.size()*4u)!=hipSuccess||hipMemcpy(s->d_wbuf, wb.data(), wb.size()*4u, hipMemcpyHostToDevice)!=hipSuccess)
			return oom("wbuf");
		k.wbuf = s->d_wbuf;
	}
	if(k.sponge_active) {
		const float inv_tau = literal_roundtrip(cfg->sponge_inv_tau_lbmu);
		const int Ns = (int)k.sponge_N;
		std::vector<float> sg(k.sponge_N);
		for(int d=0; d<Ns; d++) {
			const float xi = Ns>1 ? 1.0f-(float)d/(float)(Ns-1) : 1.0f;
			float sigma = sinf(1.5707963267948966f*xi);
			sigma = inv_tau*sigma*sigma;
			sg[d] = sigma;
		}
		if(hipMalloc((void**)&s->d_sigma, sg.size()*4u)!=hipSuccess||hipMemcpy(s->d_sigma, sg.data(), sg.size()*4u, hipMemcpyHostToDevice)!=hipSuccess)
			return oom("sigma");
		k.sigma = s->d_sigma;
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	if(int e = tune_ddf_placement(s)) { luw_destroy(s); return e; } // last: the probe steps run the complete kernel (nudging / sponge tables included)
	s->create_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now()-t_create).count();
	*out = s;
	return LUW_OK;
}

void* luw_host_ptr(luw_solver* s, int field) {
	if(!s) return nullptr;
	switch(field) {
		case LUW_FIELD_RHO: return s->h_rho;
		case LUW_FIELD_U: return s->h_u;
		case LUW_FIELD_FLAGS: return s->h_flags;
		case LUW_FIELD_F: return s->h_F;
		case LUW_FIELD_T: return s->h_T;
		default: return nullptr;
	}
}
void* luw_device_ptr(luw_solver* s, int field) {
	if(!s) return nullptr;
	switch(field) {
		case LUW_FIELD_RHO: return s->d_rho;
		case LUW_FIELD_U: return s->d_u;
		case LUW_FIELD_FLAGS: return s->d_flags;
		case LUW_FIELD_F: return s->d_F;
		case LUW_FIELD_FI: return s->d_fi;
		case LUW_FIELD_T: return s->d_T;
		case LUW_FIELD_GI: return s->d_gi;
		default: return nullptr;
	}
}
uint64_t luw_get_N(const luw_solver* s) { return s ? s->N : 0ull; }
uint64_t luw_get_t(const luw_solver* s) { return s ? s->t : 0ull; }
uint32_t luw_get_pitch(const luw_solver* s) { return s ? s->kp.Px : 0u; }
uint64_t luw_get_plane_stride(const luw_solver* s) { return s ? s->kp.Np : 0ull; }
uint64_t luw_get_area(const luw_solver* s, uint32_t direction) {
	if(!s||direction>2u) return 0ull;
	const uint64_t A[3] = { (uint64_t)s->cfg.Ny*s->cfg.Nz, (uint64_t)s->cfg.Nz*s->cfg.Nx, (uint64_t)s->cfg.Nx*s->cfg.Ny };
	return A[direction];
}

int luw_set_stream(luw_solver* s, void* hip_stream) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_set_stream: null solver");
	s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
	return LUW_OK;
}
int luw_finish(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_finish: null solver");
	if(int e = set_device(s)) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

int luw_upload(luw_solver* s, uint32_t mask) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_upload: null solver");
	if(int e = set_device(s)) return e;
	int e = LUW_OK;
	if(mask&LUW_MASK_RHO) if((e = copy_pitched(s->d_rho, s->h_rho, 4u, s, 1u, true, s->stream))) return e;
	if(mask&LUW_MASK_U) if((e = copy_pitched(s->d_u, s->h_u, 4u, s, 3u, true, s->stream))) return e;
	if(mask&LUW_MASK_FLAGS) if((e = copy_pitched(s->d_flags, s->h_flags, 1u, s, 1u, true, s->stream))) return e;
	if((mask&LUW_MASK_F)&&s->d_F) if((e = copy_pitched(s->d_F, s->h_F, 4u, s, 3u, true, s->stream))) return e;
	if((mask&LUW_MASK_T)&&s->d_T) if((e = copy_pitched(s->d_T, s->h_T, 4u, s, 1u, true, s->stream))) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

int luw_download(luw_solver* s, uint32_t mask) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_download: null solver");
	if(int e = set_device(s)) return e;
	int e = LUW_OK;
	if(mask&LUW_MASK_RHO) if((e = copy_pitched(s->h_rho, s->d_rho, 4u, s, 1u, false, s->stream))) return e;
	if(mask&LUW_MASK_U) if((e = copy_pitched(s->h_u, s->d_u, 4u, s, 3u, false, s->stream))) return e;
	if(mask&LUW_MASK_FLAGS) if((e = copy_pitched(s->h_flags, s->d_flags, 1u, s, 1u, false, s->stream))) return e;
	if((mask&LUW_MASK_F)&&s->d_F) if((e = copy_pitched(s->h_F, s->d_F, 4u, s, 3u, false, s->stream))) return e;
	if((mask&LUW_MASK_T)&&s->d_T) if((e = copy_pitched(s->h_T, s->d_T, 4u, s, 1u, false, s->stream))) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

int luw_download_fi(luw_solver* s, void* host_dst) {
	if(!s||!host_dst) return fail(LUW_ERR_INVALID, "luw_download_fi: bad argument");
	if(int e = set_device(s)) return e;
	if(int e = copy_pitched(host_dst, s->d_fi, s->ddf_bytes, s, 19u, false, s->stream)) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}
int luw_download_gi(luw_solver* s, void* host_dst) {
	if(!s||!host_dst) return fail(LUW_ERR_INVALID, "luw_download_gi: bad argument");
	if(!s->d_gi) return fail(LUW_ERR_STATE, "luw_download_gi: the solver was created without LUW_OPT_TEMPERATURE");
	if(int e = set_device(s)) return e;
	if(int e = copy_pitched(host_dst, s->d_gi, s->ddf_bytes, s, 7u, false, s->stream)) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}
int luw_upload_fi(luw_solver* s, const void* host_src) {
	if(!s||!host_src) return fail(LUW_ERR_INVALID, "luw_upload_fi: bad argument");
	if(int e = set_device(s)) return e;
	if(int e = copy_pitched(s->d_fi, host_src, s->ddf_bytes, s, 19u, true, s->stream)) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

int luw_run(luw_solver* s, uint64_t steps);
int luw_upload(luw_solver* s, uint32_t mask);
int luw_download(luw_solver* s, uint32_t mask);
static int vk_apply(luw_solver* s);
static int voxelize_launch(const VoxGrid& vg, uint8_t* d_flags, const float* d_u, const uint8_t flag, const uint32_t ntri, const float* p0, const float* p1,
	const float* p2, const float* const d[3], const float* pmin, const float* pmax, hipStream_t st);
int luw_voxelize_mesh(luw_solver* s, uint32_t triangle_number, const float* p0, const float* p1, const float* p2, const float* bounds, uint8_t flag) {
	if(!s||!p0||!p1||!p2||triangle_number==0u) return fail(LUW_ERR_INVALID, "luw_voxelize_mesh: bad argument");
	if(int e = set_device(s)) return e;
	float pmin[3], pmax[3]; // Mesh::find_bounds seeds with p0[0] only, FX/utilities.hpp:4774-4785
	if(bounds) for(int c=0; c<3; c++) { pmin[c] = bounds[c]; pmax[c] = bounds[3+c]; }
	else {
		for(int c=0; c<3; c++) pmin[c] = pmax[c] = p0[c];
		for(uint32_t i=1u; i<triangle_number; i++) for(int c=0; c<3; c++) {
			pmin[c] = fminf(fminf(fminf(p0[3u*i+c], p1[3u*i+c]), p2[3u*i+c]), pmin[c]);
			pmax[c] = fmaxf(fmaxf(fmaxf(p0[3u*i+c], p1[3u*i+c]), p2[3u*i+c]), pmax[c]);
		}
	}
	float* d[3] = { nullptr, nullptr, nullptr };
	const float* h[3] = { p0, p1, p2 };
	struct Release { float** d; ~Release() { for(int k=0; k<3; k++) (void)hipFree(d[k]); } } release{ d }; // the triangle arrays go on every path out
	for(int k=0; k<3; k++) {
		if(hipMalloc((void**)&d[k], 12ull*triangle_number)!=hipSuccess) { d[k] = nullptr; return fail(LUW_ERR_NOMEM, "luw_voxelize_mesh: allocation failed"); }
		HIP_TRY(hipMemcpy(d[k], h[k], 12ull*triangle_number, hipMemcpyHostToDevice));
	}
	if(int e = luw_upload(s, LUW_MASK_FLAGS|LUW_MASK_U)) return e; // the host mirror is authoritative before the first run
	const VoxGrid vg = { s->kp.Nx, s->kp.Ny, s->kp.Nz, s->kp.Px, s->kp.Ox, s->kp.Oy, s->kp.Oz, (uint64_t)s->kp.Np };
	if(int rc = voxelize_launch(vg, s->d_flags, s->d_u, flag, triangle_number, p0, p1, p2, d, pmin, pmax, s->stream)) return rc;
	return luw_download(s, LUW_MASK_FLAGS); // LBM::voxelize_mesh_on_device leaves the result in lbm.flags
}

// bins + launch shared by luw_voxelize_mesh (a solver's domain) and luw_voxelize_lattice (bare lattice)
static int voxelize_launch(const VoxGrid& vg, uint8_t* d_flags, const float* d_u, const uint8_t flag, const uint32_t ntri, const float* p0, const float* p1,
	const float* p2, const float* const d[3], const float* pmin, const float* pmax, hipStream_t st) {
	const uint32_t tx = (vg.Nx+VOX_TILE-1u)/VOX_TILE, ty = (vg.Ny+VOX_TILE-1u)/VOX_TILE;
	const bool brute = tuning().voxelize_all; // test aid: every tile sees every triangle
	std::vector<uint32_t> start((size_t)tx*ty+1u, 0u), tri;
	auto range = [&](const uint32_t i, int& a0, int& a1, int& b0, int& b1) {
		if(brute) { a0 = 0; a1 = (int)tx-1; b0 = 0; b1 = (int)ty-1; return; }
		const float xlo = fminf(fminf(p0[3u*i], p1[3u*i]), p2[3u*i]), xhi = fmaxf(fmaxf(p0[3u*i], p1[3u*i]), p2[3u*i]);
		const float ylo = fminf(fminf(p0[3u*i+1u], p1[3u*i+1u]), p2[3u*i+1u]), yhi = fmaxf(fmaxf(p0[3u*i+1u], p1[3u*i+1u]), p2[3u*i+1u]);
		const float pad = 1.0f+1.0e-4f; // overlap_pad + overlap_eps of the reference's subset test
		a0 = (int)floorf((xlo-pad-(float)vg.Ox)/(float)VOX_TILE); a1 = (int)floorf((xhi+pad-(float)vg.Ox)/(float)VOX_TILE);
		b0 = (int)floorf((ylo-pad-(float)vg.Oy)/(float)VOX_TILE); b1 = (int)floorf((yhi+pad-(float)vg.Oy)/(float)VOX_TILE);
		a0 = std::max(a0, 0); b0 = std::max(b0, 0); a1 = std::min(a1, (int)tx-1); b1 = std::min(b1, (int)ty-1);
	};
	for(uint32_t i=0u; i<ntri; i++) {
		int a0, a1, b0, b1;
		range(i, a0, a1, b0, b1);
		for(int b=b0; b<=b1; b++) for(int a=a0; a<=a1; a++) start[(size_t)a+(size_t)b*tx+1u]++;
	}
	for(size_t t=0u; t<(size_t)tx*ty; t++) {
		if((uint64_t)start[t]+start[t+1u]>0xFFFFFFFFull) return fail(LUW_ERR_INVALID, "voxelize: triangle bins exceed 2^32 entries");
		start[t+1u] += start[t];
	}
	tri.resize(std::max<size_t>(start.back(), 1u));
	{ std::vector<uint32_t> fill(start.begin(), start.end()-1);
	  for(uint32_t i=0u; i<ntri; i++) {
		int a0, a1, b0, b1;
		range(i, a0, a1, b0, b1);
		for(int b=b0; b<=b1; b++) for(int a=a0; a<=a1; a++) tri[fill[(size_t)a+(size_t)b*tx]++] = i;
	} }
	uint32_t* d_start = nullptr; uint32_t* d_tri = nullptr;
	if(hipMalloc((void**)&d_start, 4ull*start.size())!=hipSuccess||hipMalloc((void**)&d_tri, 4ull*tri.size())!=hipSuccess) {
		(void)hipFree(d_start);
		(void)hipFree(d_tri);
		return fail(LUW_ERR_NOMEM, "voxelize: allocation failed");
	}
	hipError_t e = hipMemcpy(d_start, start.data(), 4ull*start.size(), hipMemcpyHostToDevice);
	if(e==hipSuccess) e = hipMemcpy(d_tri, tri.data(), 4ull*tri.size(), hipMemcpyHostToDevice);
	if(e==hipSuccess) {
		hipLaunchKernelGGL(k_voxelize_z, dim3(tx, ty), dim3(256), 0, st, vg, d_flags, d_u, flag, d_start, d_tri, d[0], d[1], d[2],
			pmin[0]-2.0f, pmin[1]-2.0f, pmin[2]-2.0f, pmax[0]+2.0f, pmax[1]+2.0f, pmax[2]+2.0f); // bounding box + 2 cells, FX/lbm.cpp:498
		e = hipGetLastError();
	}
	if(e==hipSuccess) e = hipStreamSynchronize(st);
	(void)hipFree(d_start); (void)hipFree(d_tri);
	if(e!=hipSuccess) return fail(LUW_ERR_DEVICE, std::string("voxelize: ")+hipGetErrorString(e));
	return LUW_OK;
}

int luw_gather_attach(luw_solver* s, uint32_t count, const uint64_t* cells) {
	if(!s||(count>0u&&!cells)) return fail(LUW_ERR_INVALID, "luw_gather_attach: bad argument");
	if(int e = set_device(s)) return e;
	(void)hipFree(s->d_gather_cell); (void)hipFree(s->d_gather_out); s->d_gather_cell = nullptr; s->d_gather_out = nullptr; s->gather_count = 0u;
	if(count==0u) return LUW_OK;
	std::vector<uint32_t> c(count);
	const uint64_t A = (uint64_t)s->cfg.Nx*s->cfg.Ny;
	for(uint32_t i=0u; i<count; i++) {
		if(cells[i]>=s->N) return fail(LUW_ERR_INVALID, "luw_gather_attach: cell index outside the lattice");
		const uint32_t z = (uint32_t)(cells[i]/A), y = (uint32_t)((cells[i]%A)/s->cfg.Nx), x = (uint32_t)(cells[i]%s->cfg.Nx);
		c[i] = x+(y+z*s->cfg.Ny)*s->kp.Px;
	}
	auto drop = [&]() { (void)hipFree(s->d_gather_cell); (void)hipFree(s->d_gather_out); s->d_gather_cell = nullptr; s->d_gather_out = nullptr; };
	if(hipMalloc((void**)&s->d_gather_cell, 4ull*count)!=hipSuccess) {
		s->d_gather_cell = nullptr;
		return fail(LUW_ERR_NOMEM, "luw_gather_attach: allocation failed");
	}
	if(hipMalloc((void**)&s->d_gather_out, 12ull*count)!=hipSuccess) {
		s->d_gather_out = nullptr;
		drop();
		return fail(LUW_ERR_NOMEM, "luw_gather_attach: allocation failed");
	}
	if(hipMemcpy(s->d_gather_cell, c.data(), 4ull*count, hipMemcpyHostToDevice)!=hipSuccess) {
		drop();
		return fail(LUW_ERR_DEVICE, "luw_gather_attach: upload failed");
	}
	s->gather_count = count;
	return LUW_OK;
}
int luw_gather_u(luw_solver* s, float* out) {
	if(!s||!out) return fail(LUW_ERR_INVALID, "luw_gather_u: bad argument");
	if(s->gather_count==0u) return LUW_OK;
	if(!s->fields_current) return fail(LUW_ERR_STATE, "luw_gather_u: rho,u on the device are stale (the last step did not write fields)");
	if(int e = set_device(s)) return e;
	hipLaunchKernelGGL(k_gather_u, dim3((s->gather_count+255u)/256u), dim3(256), 0, s->stream, s->gather_count, s->d_gather_cell, s->d_u, (size_t)s->kp.Np,
		s->d_gather_out);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(out, s->d_gather_out, 12ull*s->gather_count, hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

int luw_voxelize_lattice(int device, uint32_t Nx, uint32_t Ny, uint32_t Nz, uint32_t triangle_number, const float* p0, const float* p1, const float* p2,
	const float* bounds, uint8_t flag, uint8_t* flags) {
	if(!p0||!p1||!p2||!bounds||!flags||triangle_number==0u||(uint64_t)Nx*Ny*Nz==0ull||(uint64_t)Nx*Ny>0xFFFFFF00ull)
		return fail(LUW_ERR_INVALID, "luw_voxelize_lattice: bad argument");
	HIP_TRY(hipSetDevice(device));
	const uint64_t N = (uint64_t)Nx*Ny*Nz;
	uint8_t* d_flags = nullptr; float* d[3] = { nullptr, nullptr, nullptr };
	auto cleanup = [&]() { (void)hipFree(d_flags); for(int k=0; k<3; k++) (void)hipFree(d[k]); };
	if(hipMalloc((void**)&d_flags, N)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_voxelize_lattice: allocation failed");
	const float* h[3] = { p0, p1, p2 };
	for(int k=0; k<3; k++)
		if(hipMalloc((void**)&d[k], 12ull*triangle_number)!=hipSuccess||hipMemcpy(d[k], h[k], 12ull*triangle_number, hipMemcpyHostToDevice)!=hipSuccess) {
		cleanup();
		return fail(LUW_ERR_NOMEM, "luw_voxelize_lattice: allocation failed");
	}
	if(hipMemcpy(d_flags, flags, N, hipMemcpyHostToDevice)!=hipSuccess) { cleanup(); return fail(LUW_ERR_DEVICE, "luw_voxelize_lattice: upload failed"); }
	const VoxGrid vg = { Nx, Ny, Nz, Nx, 0, 0, 0, N };
	const int rc = voxelize_launch(vg, d_flags, nullptr, flag, triangle_number, p0, p1, p2, d, bounds, bounds+3, (hipStream_t)0);
	hipError_t e = rc==LUW_OK ? hipMemcpy(flags, d_flags, N, hipMemcpyDeviceToHost) : hipSuccess;
	cleanup();
	if(rc!=LUW_OK) return rc;
	if(e!=hipSuccess) return fail(LUW_ERR_DEVICE, std::string("luw_voxelize_lattice: ")+hipGetErrorString(e));
	return LUW_OK;
}

int luw_vk_inlet_detach(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_vk_inlet_detach: null solver");
	(void)hipSetDevice(s->cfg.device);
	if(s->vk_stream) { (void)hipStreamSynchronize(s->vk_stream); (void)hipStreamDestroy(s->vk_stream); s->vk_stream = nullptr; }
	for(int b=0; b<2; b++) { // the side stream and its four events come and go together: a partial creation leaves nothing behind
		if(s->vk_ready[b]) { (void)hipEventDestroy(s->vk_ready[b]); s->vk_ready[b] = nullptr; }
		if(s->vk_taken[b]) { (void)hipEventDestroy(s->vk_taken[b]); s->vk_taken[b] = nullptr; }
	}
	for(int b=0; b<2; b++) { (void)hipFree(s->d_vk_val[b]); s->d_vk_val[b] = nullptr; s->vk_val_t[b] = ~0ull; }
	(void)hipFree(s->d_vk_cell); (void)hipFree(s->d_vk_face); (void)hipFree(s->d_vk_point); (void)hipFree(s->d_vk_mode);
	s->d_vk_cell = nullptr; s->d_vk_face = nullptr; s->d_vk_point = nullptr; s->d_vk_mode = nullptr;
	s->vk_active = false; s->vk_P = s->vk_M = 0u;
	return LUW_OK;
}
int luw_vk_inlet_attach(luw_solver* s, uint64_t point_count, uint64_t mode_count, const uint64_t* point_cell, const uint8_t* point_face,
	const float* point_data, const float* mode_data, int update_stride, int stride_interpolation) {
	if(!s||!point_cell||!point_face||!point_data||!mode_data) return fail(LUW_ERR_INVALID, "luw_vk_inlet_attach: null argument");
	if(point_count==0ull||mode_count==0ull||point_count>=(1ull<<31)||mode_count>65536ull) return fail(LUW_ERR_INVALID, "luw_vk_inlet_attach: bad table sizes");
	if(int e = set_device(s)) return e;
	(void)luw_vk_inlet_detach(s);
	std::vector<uint32_t> cell(point_count); // reference-layout cell index -> pitched device index
	const uint64_t NxNy = (uint64_t)s->cfg.Nx*s->cfg.Ny;
	for(uint64_t i=0ull; i<point_count; i++) {
		const uint64_t n = point_cell[i];
		if(n>=s->N) return fail(LUW_ERR_INVALID, "luw_vk_inlet_attach: point cell outside the lattice");
		const uint64_t t = n%NxNy; const uint32_t x = (uint32_t)(t%s->cfg.Nx), y = (uint32_t)(t/s->cfg.Nx), z = (uint32_t)(n/NxNy);
		cell[i] = x+(y+z*s->cfg.Ny)*s->kp.Px;
	}
	const size_t P = point_count, V = 5ull*mode_count;
	// all four tables or none: a failure half-way leaves the solver without an inlet (detach frees what was allocated)
	auto table = [&](void** dst, const void* src, const size_t bytes) {
		if(hipMalloc(dst, bytes)!=hipSuccess) { *dst = nullptr; return false; }
		return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice)==hipSuccess;
	};
	if(!table((void**)&s->d_vk_cell, cell.data(), P*4u)||!table((void**)&s->d_vk_face, point_face, P)||!table((void**)&s->d_vk_point, point_data, 7ull*P*4u)
		||!table((void**)&s->d_vk_mode, mode_data, 10ull*V*4u)) {
		(void)hipGetLastError(); (void)luw_vk_inlet_detach(s);
		return fail(LUW_ERR_NOMEM, "luw_vk_inlet_attach: allocating / uploading the inlet tables failed");
	}
	s->vk_P = (uint32_t)P; s->vk_M = (uint32_t)mode_count; s->vk_stride = update_stride>1 ? update_stride : 1; s->vk_interp = stride_interpolation!=0;
	const bool ahead = tuning().vk_ahead; // LUW_VK_AHEAD=0: evaluate in line before every step (A/B and test aid)
	if(ahead) {
		bool ok = hipMalloc((void**)&s->d_vk_val[0], 3ull*P*4u)==hipSuccess&&hipMalloc((void**)&s->d_vk_val[1], 3ull*P*4u)==hipSuccess;
		if(ok&&!s->vk_stream) {
			ok = hipStreamCreateWithFlags(&s->vk_stream, hipStreamNonBlocking)==hipSuccess;
			for(int b=0; b<2&&ok; b++) ok = hipEventCreateWithFlags(&s->vk_ready[b], hipEventDisableTiming)==hipSuccess
				&&hipEventCreateWithFlags(&s->vk_taken[b], hipEventDisableTiming)==hipSuccess;
		}
		if(!ok) { (void)hipGetLastError(); (void)luw_vk_inlet_detach(s); return fail(LUW_ERR_NOMEM, "luw_vk_inlet_attach: side stream / value buffers"); }
	}
	s->vk_active = true; s->vk_last_t = ~0ull;
	return LUW_OK;
}
int luw_vk_inlet_apply(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_vk_inlet_apply: null solver");
	if(!s->vk_active) return fail(LUW_ERR_STATE, "luw_vk_inlet_apply: no inlet attached");
	if(int e = set_device(s)) return e;
	return vk_apply(s);
}

int luw_stats_reset(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_stats_reset: null solver");
	if(int e = set_device(s)) return e;
	const size_t Np = s->kp.Np;
	if(!s->d_avg_u) {
		if(lead_alloc(s, (void**)&s->d_avg_u, 3ull*Np, 4u)!=hipSuccess||lead_alloc(s, (void**)&s->d_avg_rho, Np, 4u)!=hipSuccess
			||lead_alloc(s, (void**)&s->d_m2, 3ull*Np, 4u)!=hipSuccess)
			return fail(LUW_ERR_NOMEM, "luw_stats_reset: allocation failed");
	}
	if(s->d_T&&!s->d_avg_T) { if(lead_alloc(s, (void**)&s->d_avg_T, Np, 4u)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_stats_reset: allocation failed"); }
	if(s->d_avg_T) HIP_TRY(hipMemsetAsync(s->d_avg_T, 0, Np*4ull, s->stream));
	HIP_TRY(hipMemsetAsync(s->d_avg_u, 0, 3ull*Np*4ull, s->stream));
	HIP_TRY(hipMemsetAsync(s->d_avg_rho, 0, Np*4ull, s->stream));
	HIP_TRY(hipMemsetAsync(s->d_m2, 0, 3ull*Np*4ull, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->avg_count = 0ull;
	return LUW_OK;
}
int luw_stats_accumulate(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_stats_accumulate: null solver");
	if(!s->d_avg_u) return fail(LUW_ERR_STATE, "luw_stats_accumulate: call luw_stats_reset first");
	if(!s->fields_current) return fail(LUW_ERR_STATE, "luw_stats_accumulate: rho,u on the device are stale (the last step did not write fields)");
	if(int e = set_device(s)) return e;
	s->avg_count++;
	const float inv_n = 1.0f/(float)s->avg_count; // FX/setup.cpp:4442-4443
	const uint32_t bx = s->cfg.Nx>=256u ? 256u : ((s->cfg.Nx+63u)/64u)*64u;
	const dim3 grid((s->cfg.Nx+bx-1u)/bx, s->cfg.Ny, s->cfg.Nz), block(bx);
	hipLaunchKernelGGL(k_stats_accumulate, grid, block, 0, s->stream, s->kp, inv_n, s->d_rho, s->d_u, s->d_avg_u, s->d_avg_rho, s->d_m2, s->d_T, s->d_avg_T);
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
int luw_stats_download(luw_solver* s, float* avg_u, float* avg_rho, float* m2_u, float* m2_v, float* m2_w, uint64_t* count) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_stats_download: null solver");
	if(!s->d_avg_u) return fail(LUW_ERR_STATE, "luw_stats_download: no statistics have been accumulated");
	if(int e = set_device(s)) return e;
	const uint64_t N = s->N;
	int e = LUW_OK;
	if(avg_u) { // the reference keeps u_avg as AoS [3n+c] (FX/setup.cpp:4453-4477): interleave on the host
		std::unique_ptr<float[]> tmp(new float[3ull*N]); // fully written by the copy: no value-initialisation
		if((e = copy_pitched(tmp.get(), s->d_avg_u, 4u, s, 3u, false, s->stream))) return e;
		HIP_TRY(hipStreamSynchronize(s->stream));
		const unsigned T = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
		std::vector<std::thread> th;
		for(unsigned t=0u; t<T; t++) th.emplace_back([&, t]() {
			const float* src = tmp.get();
			for(uint64_t n=N*t/T; n<N*(t+1ull)/T; n++) { avg_u[3ull*n] = src[n]; avg_u[3ull*n+1ull] = src[N+n]; avg_u[3ull*n+2ull] = src[2ull*N+n]; }
		});
		for(auto& x : th) x.join();
	}
	if(avg_rho) if((e = copy_pitched(avg_rho, s->d_avg_rho, 4u, s, 1u, false, s->stream))) return e;
	float* m2h[3] = { m2_u, m2_v, m2_w };
	for(int c=0; c<3; c++) if(m2h[c]) if((e = copy_pitched(m2h[c], s->d_m2+(size_t)c*s->kp.Np, 4u, s, 1u, false, s->stream))) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	if(count) *count = s->avg_count;
	return LUW_OK;
}
int luw_stats_download_T(luw_solver* s, float* avg_T) {
	if(!s||!avg_T) return fail(LUW_ERR_INVALID, "luw_stats_download_T: bad argument");
	if(!s->d_avg_T) return fail(LUW_ERR_STATE, "luw_stats_download_T: no temperature statistics (LUW_OPT_TEMPERATURE + luw_stats_reset)");
	if(int e = set_device(s)) return e;
	if(int e = copy_pitched(avg_T, s->d_avg_T, 4u, s, 1u, false, s->stream)) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}
int luw_selfcheck_fp16c_codec(int device, uint64_t* mismatches) {
	if(!mismatches) return fail(LUW_ERR_INVALID, "luw_selfcheck_fp16c_codec: null argument");
	HIP_TRY(hipSetDevice(device));
	unsigned long long* d = nullptr;
	HIP_TRY(hipMalloc((void**)&d, 8));
	HIP_TRY(hipMemset(d, 0, 8));
	hipLaunchKernelGGL(k_codec_check, dim3(4096), dim3(256), 0, 0, d);
	HIP_TRY(hipGetLastError());
	unsigned long long h = 0ull;
	HIP_TRY(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
	(void)hipFree(d);
	*mismatches = h;
	return LUW_OK;
}

int luw_selfcheck_arith(int device, uint64_t* mismatches) {
	if(!mismatches) return fail(LUW_ERR_INVALID, "luw_selfcheck_arith: null argument");
	HIP_TRY(hipSetDevice(device));
	unsigned long long* d = nullptr;
	HIP_TRY(hipMalloc((void**)&d, 24));
	HIP_TRY(hipMemset(d, 0, 24));
	hipLaunchKernelGGL(k_arith_check, dim3(4096), dim3(256), 0, 0, d);
	HIP_TRY(hipGetLastError());
	unsigned long long h[3] = { 0ull, 0ull, 0ull };
	HIP_TRY(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
	(void)hipFree(d);
	for(int k=0; k<3; k++) mismatches[k] = h[k];
	return LUW_OK;
}

int luw_set_f(luw_solver* s, float fx, float fy, float fz) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_set_f: null solver");
	s->cfg.fx = s->kp.fx = fx; s->cfg.fy = s->kp.fy = fy; s->cfg.fz = s->kp.fz = fz;
	return LUW_OK;
}
int luw_set_coriolis(luw_solver* s, float ox, float oy, float oz) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_set_coriolis: null solver");
	s->cfg.omega_x = s->kp.omx = ox; s->cfg.omega_y = s->kp.omy = oy; s->cfg.omega_z = s->kp.omz = oz;
	s->kp.coriolis = ox!=0.0f||oy!=0.0f||oz!=0.0f;
	s->kp.m2omx = -2.0f*ox; s->kp.m2omy = -2.0f*oy; s->kp.m2omz = -2.0f*oz;
	return LUW_OK;
}

// Buffer nudging and the top sponge pull cells towards u of a REFERENCE cell on an outer face (FX/kernel.cpp:1543-1611).  In LUW's decks those
// faces are TYPE_E (or solid ground): their u is an input that no step rewrites, so it does not matter that this library writes rho,u only in the
// last step of a run() call while the reference (UPDATE_FIELDS) writes them in every step.  A caller who leaves FLUID cells on such a face would see
// the target velocity of the last written step instead of the previous step's: for such a lattice the solver writes the fields every step, like
// the reference, whatever the length of the run() calls.  (Bit-level expectations end there: the reference kernel then reads a neighbour's u while that
// neighbour's thread rewrites it in the same launch.)  Checked on the host mirror at initialisation.
static bool reference_cells_are_inputs(const luw_solver* s) {
	const KParams& k = s->kp;
	const uint32_t Nx = s->cfg.Nx, Ny = s->cfg.Ny, Nz = s->cfg.Nz;
	auto input_cell = [&](const uint32_t x, const uint32_t y, const uint32_t z) { return (s->h_flags[(size_t)x+((size_t)y+(size_t)z*Ny)*Nx]&TYPE_BO)!=0u; };
	bool ok = true;
	auto x_face = [&](const uint32_t x) { for(uint32_t z=0u; z<Nz&&ok; z++) for(uint32_t y=0u; y<Ny; y++) if(!input_cell(x, y, z)) { ok = false; break; } };
	auto y_face = [&](const uint32_t y) { for(uint32_t z=0u; z<Nz&&ok; z++) for(uint32_t x=0u; x<Nx; x++) if(!input_cell(x, y, z)) { ok = false; break; } };
	auto z_face = [&](const uint32_t z) { for(uint32_t y=0u; y<Ny&&ok; y++) for(uint32_t x=0u; x<Nx; x++) if(!input_cell(x, y, z)) { ok = false; break; } };
	if(k.zw_n) x_face((uint32_t)k.west_x);
	if(k.ze_n&&ok) x_face((uint32_t)k.east_x);
	if(k.zs_n&&ok) y_face((uint32_t)k.south_y);
	if(k.zn_n&&ok) y_face((uint32_t)k.north_y);
	if((k.zt_n||k.zp_n)&&ok) z_face((uint32_t)k.top_z);
	// thermal lattice: the sponge on T reads the top layer's temperature, which only a preset (TYPE_T) keeps between the steps that store T (thermal_cell)
	if(s->d_gi&&k.zp_n&&ok) {
		const uint32_t z = (uint32_t)k.top_z;
		for(uint32_t y=0u; y<Ny&&ok; y++) for(uint32_t x=0u; x<Nx; x++) if((s->h_flags[(size_t)x+((size_t)y+(size_t)z*Ny)*Nx]&TYPE_T)==0u) {
			ok = false;
			break;
		}
	}
	return ok;
}
int luw_fields_every_step(const luw_solver* s) { return (s&&((s->cfg.options&LUW_OPT_UPDATE_FIELDS_EVERY_STEP)!=0u||s->every_step_auto)) ? 1 : 0; }

int luw_initialize(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_initialize: null solver");
	s->every_step_auto = !reference_cells_are_inputs(s);
	if(int e = luw_upload(s, LUW_MASK_RHO|LUW_MASK_U|LUW_MASK_FLAGS|LUW_MASK_F|LUW_MASK_T)) return e;
	const uint32_t bx = s->cfg.Nx>=256u ? 256u : ((s->cfg.Nx+63u)/64u)*64u;
	const dim3 grid((s->cfg.Nx+bx-1u)/bx, s->cfg.Ny, s->cfg.Nz), block(bx);
	if(s->ddf_bytes==2u) hipLaunchKernelGGL((k_initialize<uint16_t>), grid, block, 0, s->stream, s->kp, (uint16_t*)s->d_fi, s->d_rho, s->d_u, s->d_flags,
		(uint16_t*)s->d_gi, s->d_T);
	else hipLaunchKernelGGL((k_initialize<float>), grid, block, 0, s->stream, s->kp, (float*)s->d_fi, s->d_rho, s->d_u, s->d_flags, (float*)s->d_gi, s->d_T);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s->stream));
	if(s->d_stage) { (void)hipFree(s->d_stage); s->d_stage = nullptr; s->stage_bytes = 0u; } // the bulk uploads are done; downloads allocate it again on demand
	s->t = 0ull;
	s->initialized = true;
	s->fields_current = true;
	return LUW_OK;
}

int luw_enqueue_stream_collide(luw_solver* s, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1, int write_fields) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_enqueue_stream_collide: null solver");
	if(!s->initialized) return fail(LUW_ERR_STATE, "luw_enqueue_stream_collide: call luw_initialize first");
	if(int e = set_device(s)) return e;
	const Box b = { x0, x1, y0, y1, z0, z1 };
	const int wf = write_fields&1;
	s->fields_current = wf!=0; // callers cover the lattice with boxes of one step using the same flag
	if(write_fields&LUW_WF_SAMPLE) { // a box of a sampled step (luw_stats_begin_sample counted it)
		if(!s->d_avg_u||!can_fuse_stats(s)||s->avg_count==0ull)
			return fail(LUW_ERR_STATE, "luw_enqueue_stream_collide: LUW_WF_SAMPLE needs luw_stats_begin_sample to have returned fused = 1");
		const StatsArgs st = { s->d_avg_u, s->d_avg_rho, s->d_m2, 1.0f/(float)s->avg_count };
		return launch_stream_collide(s, b, wf, &st);
	}
	return launch_stream_collide(s, b, wf);
}
int luw_stats_begin_sample(luw_solver* s, int* fused) {
	if(!s||!fused) return fail(LUW_ERR_INVALID, "luw_stats_begin_sample: bad argument");
	if(!s->d_avg_u) return fail(LUW_ERR_STATE, "luw_stats_begin_sample: call luw_stats_reset first");
	*fused = can_fuse_stats(s) ? 1 : 0;
	if(*fused) s->avg_count++;
	return LUW_OK;
}
int luw_set_kernel(luw_solver* s, uint32_t kernel) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_set_kernel: null solver");
	if(!kernel_selectable(kernel)) return fail(LUW_ERR_INVALID, "luw_set_kernel: this library has no such kernel");
	s->kernel = kernel;
	return LUW_OK;
}
int luw_increment_time_step(luw_solver* s, uint64_t steps) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_increment_time_step: null solver");
	s->t += steps;
	return LUW_OK;
}

int luw_reset_time_step(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_reset_time_step: null solver");
	s->t = 0ull;
	return LUW_OK;
}

// VonKarmanInletUpdater::update + compute_time_params_ (FX/setup.cpp:538-558,1118-1140): at most once per time step.
// The evaluation (253 k inlet points x 256 modes x 3 cosf on a 512x512x128 deck: 155 us of pure VALU work) does not sit in front of
// the step any more: the values of step t+1 are computed on a side stream into a packed buffer while step t -- HBM-bound -- runs,
// and the step's own stream only scatters them into u (a few us).  Same kernel arithmetic, same values.
static int vk_launch_eval(luw_solver* s, const uint64_t t, float* dst, const size_t dstride, const uint32_t* cell, hipStream_t st) {
	const uint64_t stride = s->vk_stride>1 ? (uint64_t)s->vk_stride : 1ull;
	uint32_t use_interp = 0u; float t0 = (float)t, t1 = (float)t, alpha = 0.0f;
	if(stride>1ull) {
		const uint64_t anchor = (t/stride)*stride;
		if(s->vk_interp) { use_interp = 1u; t0 = (float)anchor; t1 = (float)(anchor+stride); alpha = (float)(t-anchor)/(float)stride; }
		else { t0 = (float)anchor; t1 = t0; }
	}
	hipLaunchKernelGGL(k_vk_inlet_apply, dim3((s->vk_P+255u)/256u), dim3(256), 0, st, use_interp, t0, t1, alpha, s->vk_P, s->vk_M, 5u*s->vk_M, cell,
		s->d_vk_face, s->d_vk_point, s->d_vk_mode, dst, dstride);
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
static int vk_apply(luw_solver* s) {
	if(!s->vk_active||s->vk_last_t==s->t) return LUW_OK;
	s->vk_last_t = s->t;
	if(!s->d_vk_val[0]) return vk_launch_eval(s, s->t, s->d_u, (size_t)s->kp.Np, s->d_vk_cell, s->stream); // in line
	auto eval_into = [&](const int b, const uint64_t t) -> int { // on the side stream, once the scatter that last read buffer b is done
		HIP_TRY(hipStreamWaitEvent(s->vk_stream, s->vk_taken[b], 0));
		if(int e = vk_launch_eval(s, t, s->d_vk_val[b], (size_t)s->vk_P, nullptr, s->vk_stream)) return e;
		HIP_TRY(hipEventRecord(s->vk_ready[b], s->vk_stream));
		s->vk_val_t[b] = t;
		return LUW_OK;
	};
	int cur = s->vk_val_t[0]==s->t ? 0 : s->vk_val_t[1]==s->t ? 1 : -1;
	if(cur<0) { cur = 0; if(int e = eval_into(0, s->t)) return e; } // first step, or time was set from outside
	HIP_TRY(hipStreamWaitEvent(s->stream, s->vk_ready[cur], 0));
	hipLaunchKernelGGL(k_vk_scatter, dim3((s->vk_P+255u)/256u), dim3(256), 0, s->stream, s->vk_P, s->d_vk_cell, s->d_vk_val[cur], s->d_u, (size_t)s->kp.Np);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(s->vk_taken[cur], s->stream));
	return eval_into(1-cur, s->t+1ull); // next step's values, beside this step
}

static int run_steps(luw_solver* s, uint64_t steps, double* mean_kernel_ms, const uint64_t first_sample = 0ull, const uint64_t stride = 0ull) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_run: null solver");
	if(int e = set_device(s)) return e;
	if(!s->initialized) { if(int e = luw_initialize(s)) return e; } // LBM::run initialises on first use, FX/lbm.cpp:1294-1296
	const Box whole = { 0u, s->cfg.Nx, 0u, s->cfg.Ny, 0u, s->cfg.Nz };
	const bool every = luw_fields_every_step(s)!=0;
	std::vector<hipEvent_t> ev;
	// on every path out
	struct EventsFree { std::vector<hipEvent_t>& v; ~EventsFree() { for(hipEvent_t e : v) if(e) (void)hipEventDestroy(e); } } events_free{ ev };
	if(mean_kernel_ms) {
		ev.assign(2u*steps, nullptr);
		for(auto& e : ev) HIP_TRY(hipEventCreate(&e));
	}
	for(uint64_t i=0ull; i<steps; i++) {
		int wf = (every||i+1ull==steps) ? 1 : 0;
		// luw_run_sampled: step i+1 of this call is a statistics sample
		const bool sampled = stride>0ull && i+1ull>=first_sample && (i+1ull-first_sample)%stride==0ull;
		const bool fused = sampled && can_fuse_stats(s);
		StatsArgs st{};
		if(fused) { s->avg_count++; st = StatsArgs{ s->d_avg_u, s->d_avg_rho, s->d_m2, 1.0f/(float)s->avg_count }; } // FX/setup.cpp:4442-4443
		if(sampled&&!fused) wf = 1;
		if(int e = vk_apply(s)) return e; // pre_step_update of the reference's run loop, FX/setup.cpp:4872
		if(mean_kernel_ms) HIP_TRY(hipEventRecord(ev[2u*i], s->stream));
		if(int e = launch_stream_collide(s, whole, wf, fused ? &st : nullptr)) return e;
		if(mean_kernel_ms) HIP_TRY(hipEventRecord(ev[2u*i+1u], s->stream));
		s->t++;
		if(sampled&&!fused) { s->fields_current = true; if(int e = luw_stats_accumulate(s)) return e; s->fields_current = false; }
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	if(steps>0ull) s->fields_current = true;
	if(mean_kernel_ms) {
		double sum = 0.0;
		for(uint64_t i=0ull; i<steps; i++) { float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, ev[2u*i], ev[2u*i+1u])); sum += (double)ms; }
		*mean_kernel_ms = steps ? sum/(double)steps : 0.0;
	}
	return LUW_OK;
}
int luw_run(luw_solver* s, uint64_t steps) { return run_steps(s, steps, nullptr); }
int luw_run_sampled(luw_solver* s, uint64_t steps, uint64_t first_sample, uint64_t stride) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_run_sampled: null solver");
	if(first_sample==0ull||stride==0ull) return fail(LUW_ERR_INVALID, "luw_run_sampled: first_sample and stride count from 1");
	if(!s->d_avg_u) return fail(LUW_ERR_STATE, "luw_run_sampled: call luw_stats_reset first");
	return run_steps(s, steps, nullptr, first_sample, stride);
}
int luw_run_timed(luw_solver* s, uint64_t steps, double* mean_kernel_ms) {
	if(!mean_kernel_ms) return fail(LUW_ERR_INVALID, "luw_run_timed: null output");
	return run_steps(s, steps, mean_kernel_ms);
}

int luw_set_x_face_buffers(luw_solver* s, void* dev_buffer_p, void* dev_buffer_m) {
	if(!s||((dev_buffer_p==nullptr)!=(dev_buffer_m==nullptr))) return fail(LUW_ERR_INVALID, "luw_set_x_face_buffers: bad argument");
	s->xf_p = dev_buffer_p; s->xf_m = dev_buffer_m; s->xf_cover = 0u; s->xf_t = ~0ull;
	return LUW_OK;
}
int luw_enqueue_extract_fi(luw_solver* s, uint32_t direction, void* buf_p, void* buf_m) {
	if(!s||direction>2u||!buf_p||!buf_m) return fail(LUW_ERR_INVALID, "luw_enqueue_extract_fi: bad argument");
	if(int e = set_device(s)) return e;
	if(direction==0u) { // the step kernels of this step have written both x faces into these very buffers already
		const bool done = buf_p==s->xf_p&&buf_m==s->xf_m&&s->xf_t==s->t&&s->xf_cover==3u;
		s->xf_cover = 0u; s->xf_t = ~0ull;
		if(done) return LUW_OK;
	}
	launch_transfer<false, false>(s, direction, buf_p, buf_m);
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
int luw_enqueue_insert_fi(luw_solver* s, uint32_t direction, const void* buf_p, const void* buf_m) {
	if(!s||direction>2u||!buf_p||!buf_m) return fail(LUW_ERR_INVALID, "luw_enqueue_insert_fi: bad argument");
	if(int e = set_device(s)) return e;
	launch_transfer<false, true>(s, direction, const_cast<void*>(buf_p), const_cast<void*>(buf_m));
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
int luw_enqueue_extract_gi(luw_solver* s, uint32_t direction, void* buf_p, void* buf_m) {
	if(!s||direction>2u||!buf_p||!buf_m) return fail(LUW_ERR_INVALID, "luw_enqueue_extract_gi: bad argument");
	if(!s->d_gi) return fail(LUW_ERR_STATE, "luw_enqueue_extract_gi: the solver was created without LUW_OPT_TEMPERATURE");
	if(int e = set_device(s)) return e;
	launch_transfer<true, false>(s, direction, buf_p, buf_m);
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
int luw_enqueue_insert_gi(luw_solver* s, uint32_t direction, const void* buf_p, const void* buf_m) {
	if(!s||direction>2u||!buf_p||!buf_m) return fail(LUW_ERR_INVALID, "luw_enqueue_insert_gi: bad argument");
	if(!s->d_gi) return fail(LUW_ERR_STATE, "luw_enqueue_insert_gi: the solver was created without LUW_OPT_TEMPERATURE");
	if(int e = set_device(s)) return e;
	launch_transfer<true, true>(s, direction, const_cast<void*>(buf_p), const_cast<void*>(buf_m));
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}

// ---- include/luw_core_dev.h: measurement and test entry points
int luw_dev_reload_tuning(void) { tuning_load(); return LUW_OK; }
static std::atomic<uint32_t> g_injected_faults{0u};
int luw_dev_inject_fault(uint32_t mask) { g_injected_faults.store(mask); return LUW_OK; }
int luw_dev_tuning_text(char* text, uint64_t size) {
	if(!text||size<64u) return fail(LUW_ERR_INVALID, "luw_dev_tuning_text: needs a buffer");
	const Tuning& t = tuning();
	const std::string alloc = t.alloc_chunk==0u ? "malloc" : t.alloc_chunk==~(size_t)0u ? "vmm:one" : "vmm:"+std::to_string(t.alloc_chunk>>20);
	snprintf(text, (size_t)size,
		"LUW_ALLOC=%s LUW_COPY_STAGED=%d LUW_ADDR_ROW=%d LUW_PAIR_GENERAL=%d LUW_FUSE_STATS=%d LUW_PLANE_SKEW=%llu LUW_TUNE_PLACEMENT=%d "
		"LUW_TUNE_FAST=%g LUW_TUNE_VERBOSE=%d LUW_VK_AHEAD=%d LUW_VOXELIZE_ALL_TRIANGLES=%d LUW_X_SHELL=%u LUW_GROUP_TRANSPORT=%s LUW_GROUP_THREADS=%d",
		alloc.c_str(), (int)t.copy_staged, (int)t.addr_row, (int)t.pair_general, (int)t.fuse_stats, (unsigned long long)t.plane_skew, t.placement_candidates,
		t.placement_bar, (int)t.placement_verbose, (int)t.vk_ahead, (int)t.voxelize_all, t.x_shell,
		t.group_transport==LUW_TRANSPORT_RCCL ? "rccl" : t.group_transport==LUW_TRANSPORT_STAGED ? "staged" : "peer", (int)t.group_threads);
	return LUW_OK;
}
int luw_dev_placement_info(const luw_solver* s, int* candidates_tried, double* probe_TBps, double* create_seconds, char* kept, uint64_t kept_size) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_dev_placement_info: null solver");
	if(candidates_tried) *candidates_tried = s->placement_tried;
	if(probe_TBps) *probe_TBps = s->placement_tbps;
	if(create_seconds) *create_seconds = s->create_seconds;
	if(kept&&kept_size) snprintf(kept, (size_t)kept_size, "%s", s->placement_kept.c_str());
	return LUW_OK;
}

} // extern "C"

#include "luw_group.hpp"
#include "luw_export.hpp"
