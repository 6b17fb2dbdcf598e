// luw_memory.hpp -- device memory of a solver: blocks (hipMalloc or one address range mapped from physical chunks), the solver object, lattice arrays
// with their lead pad, host <-> device copies of pitched arrays.  Included by luw_core.hip only, after luw_host.hpp.
#pragma once

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
// An address range whose mappings are gone is not handed back to the runtime at once: the next reservation of the process would land on the very addresses
// whose translations were torn down a moment ago, and on this ROCm (7.2, gfx950) a NEW mapping at a recycled virtual address is not seen by all of the
// device: tools/vmm_group_cycle.hip, a bare HIP program, maps fifty ranges from 1 GiB chunks, zeroes them, writes a mark into every 2 MiB page with one
// kernel and reads it back with the next -- with hipMemAddressFree at once, from the second cycle on about 2 % of the pages do not read back what was
// written (translations of the previous mapping survive somewhere); with the ranges kept reserved, none (profiles/r06_vmm.txt).  Round 5 met the same thing
// as a memory access fault in the runtime's own memset of a just-mapped array, right after an eight-domain group had released fifty ranges
// (profiles/r05_vmm_range_reuse.txt).  Retired ranges wait here -- reserved, nothing mapped, their physical memory given back (checked:
// tools/vmm_product_cycle.py) -- and only the oldest go back once more than 32 TiB of address space or 2048 ranges are waiting: thousands of solvers later.
static std::mutex g_retired_mutex;
static std::deque<std::pair<void*, size_t>> g_retired_ranges;
static size_t g_retired_bytes = 0u;
static void retire_address_range(void* base, const size_t bytes) {
	std::lock_guard<std::mutex> lock(g_retired_mutex);
	g_retired_ranges.emplace_back(base, bytes); g_retired_bytes += bytes;
	while(g_retired_ranges.size()>2048u||g_retired_bytes>(32ull<<40)) {
		(void)hipMemAddressFree(g_retired_ranges.front().first, g_retired_ranges.front().second);
		g_retired_bytes -= g_retired_ranges.front().second; g_retired_ranges.pop_front();
	}
}
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
		retire_address_range(b.base, b.bytes);
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
	void* xf_p = nullptr; void* xf_m = nullptr; uint32_t xf_cover = 0u; uint64_t xf_t = ~0ull, xf_area[2] = { 0ull, 0ull };
	struct FaceRect { uint32_t y0, y1, z0, z1; };     // (y, z) extents of the launches of step xf_t that held a border column, per side (xface_covered)
	FaceRect xf_rect[2][8] = {}; uint32_t xf_rects[2] = { 0u, 0u }; bool xf_lost[2] = { false, false };
	// x-face input (luw_set_x_face_inputs): receive buffers whose insert is pending, side by side (bit 0: the face that came from +x, for the last owned
	// column; bit 1: from -x, for the first) -- xin_buf: still only in the buffer; xin_inplace: read there by a launch of step xin_for_t. Whatever else needs a
	// side in the lattice has the insert kernel run for it first (xin_settle).
	// Each side carries the step it is for and the time parity of its hand-over (a host may hand the two sides over at different moments of a step).
	const void* xin_p = nullptr; const void* xin_m = nullptr; uint32_t xin_buf = 0u, xin_inplace = 0u, xin_odd[2] = { 0u, 0u }; bool xin_use = false;
	uint64_t xin_for_t[2] = { 0ull, 0ull };
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
	for(const DevBlock& b : s->raw) if((const char*)p>=(const char*)b.base&&(const char*)p<(const char*)b.base+b.bytes&&!(b.chunks.empty()&&b.chunk_bytes))
		return &b;
	return nullptr;
}
static int copy_pitched(void* dst, const void* src, const size_t elem, luw_solver* s, const uint32_t planes, const bool to_device, hipStream_t st) {
	const size_t rows = (size_t)s->cfg.Ny*s->cfg.Nz;
	// hipMemcpy2DAsync is left to the one case it is good at -- arrays in ONE physical piece whose rows are whole dwords.  It rejects
	// ranges that span the chunks of a mapped array ("invalid argument"), and rows that are not a multiple of four bytes take a path
	// that moves 0.06 GB/s (flags of a 514-cell-wide domain: 2.1 s instead of 3 ms) -- both go through the staging buffer.
	const DevBlock* blk = block_of(s, to_device ? dst : src);
	if((!blk||blk->chunks.size()<=1u)&&((size_t)s->cfg.Nx*elem)%4u==0u) {
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
