// luw_placement.hpp -- luw_create's placement search for the DDF array.  Included by luw_core.hip only, after luw_launch.hpp.
#pragma once

// Where the driver places the DDF array physically changes the step time of this 19-stream kernel by 10-14 % on MI355X: allocations of the same size come
// out in classes that last for the life of the allocation (512^3 FP32: 3.3 or 3.75 ms per step; tools/placement_probe.py, tools/chunk_study.sh), and WHICH
// kind of allocation is of the fast class depends on the box: 1 GiB chunks on most, 2 GiB chunks or a plain hipMalloc on others
// (profiles/r03_chunk_study_slow_box.txt).  Large solvers therefore time the real kernel on their DDF array -- the box the step launches (non-halo cells),
// zero DDFs = rest state, flags 0 = all fluid: a valid, full-cost step -- and, while the rate is under the bar of the fast class, try the OTHER kinds, once
// each: 1 GiB chunks again, 4 GiB chunks, 2 GiB chunks, hipMalloc, 512 MiB chunks -- each a fresh draw of physical memory (the arrays tried before stay mapped
// until the search ends).  Bounded: at most five further draws, by the device's free memory, and nothing of the search left when luw_create returns
// (placement_kept / placement_tbps / placement_tried say what happened; bench.py prints them). Skipped for small lattices, for planes of 2 GiB and more
// (1024^3 runs alike on every kind, profiles/r02_placement_study.txt),
// and when the device is shared: other solvers of this process live on it (g_live_solvers), or the
// caller says so (luw_group_create for devices that host several domains: thread-local g_device_is_shared; rank processes sharing one GPU set
// LUW_TUNE_PLACEMENT=0) -- concurrent probes would time each other.
static std::atomic<int> g_live_solvers[64];
static thread_local bool g_device_is_shared = false;
constexpr size_t PLACEMENT_UNSET = ~(size_t)0u-1u;     // g_placement_kind: chunk size the process's search kept for this device (0: hipMalloc)
static std::atomic<size_t> g_placement_kind[64];
static struct PlacementKindInit { PlacementKindInit() { for(auto& k : g_placement_kind) k.store(PLACEMENT_UNSET); } } g_placement_kind_init;
static const char* dev_block_kind(const DevBlock& b) {
	if(b.chunks.empty()) return "hipMalloc";
	return b.chunk_bytes>=(4096ull<<20) ? "4 GiB chunks" : b.chunk_bytes>=(2048ull<<20) ? "2 GiB chunks" : b.chunk_bytes>=(1024ull<<20) ? "1 GiB chunks"
		: b.chunk_bytes>=(512ull<<20) ? "512 MiB chunks"
		: "chunks under 512 MiB";
}
static int tune_ddf_placement(luw_solver* s) {
	const Tuning& T = tuning();
	const size_t elems = 19ull*s->kp.Np, bytes = elems*s->ddf_bytes;
	const bool mapped = !s->raw.front().chunks.empty();
	if(s->placement_kept=="default (no search)") s->placement_kept = std::string(dev_block_kind(s->raw.front()))+" (no search)";
	constexpr int NALT = 5;
	// further draws: the default kind once more, then the other kinds (chunk sizes; 0: hipMalloc)
	const size_t alternatives[NALT] = { 1024ull<<20, 4096ull<<20, 2048ull<<20, 0u, 512ull<<20 };
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
	auto step_ms = [&](float& ms) -> int { // per two steps (both parities): four timed steps after one untimed
		struct Restore { luw_solver* s; ~Restore() { s->initialized = false; s->t = 0ull; } } restore{ s };
		s->initialized = true; s->t = 0ull;
		if(int e = launch_stream_collide(s, box, 0)) return e;
		HIP_TRY(hipEventRecord(ev.e0, s->stream));
		for(s->t=1ull; s->t<=4ull; s->t++) if(int e = launch_stream_collide(s, box, 0)) return e;
		HIP_TRY(hipEventRecord(ev.e1, s->stream));
		HIP_TRY(hipEventSynchronize(ev.e1));
		HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
		ms *= 0.5f;
		return LUW_OK;
	};
	// a placement of the fast class moves this many algorithmic bytes per second through the probe (FP32 153, FP16C 77 B per update, + the thermal planes;
	// FP16C with zones: the general kernel is VALU-bound).  LUW_TUNE_FAST=<TB/s> overrides the bar (99: every candidate is tried)
	const double cells = (double)(box.x1-box.x0)*(double)(box.y1-box.y0)*(double)(box.z1-box.z0);
	const double probe_bytes = 2.0*((s->ddf_bytes==4u ? 153.0 : 77.0)+(s->d_gi ? 14.0*(double)s->ddf_bytes : 0.0))*cells;
	// (zone cells cost the probe more: the FP32 urban tile probes 6.2 TB/s on a fast draw, 6.07-6.13 on a slow one)
	const bool zones = s->kp.buffer_active||s->kp.sponge_active;
	const double bar = T.placement_bar>0.0 ? T.placement_bar*1e12 : (s->ddf_bytes==4u ? (zones ? 6.2e12 : 6.27e12) : zones ? 5.0e12 : 6.1e12);
	auto rate = [&](const float ms) { return probe_bytes/((double)ms*1e-3); };
	float best_ms = 0.0f;
	if(int e = step_ms(best_ms)) return e;   // (the first probe of a process also ramps the GPU up: measured again)
	if(int e = step_ms(best_ms)) return e;
	if(g_injected_faults.load()&LUW_FAULT_SLOW_FIRST_PLACEMENT) best_ms *= 1.3f;   // test hook: the array in place has to be replaced by another draw
	s->placement_tried = 1;
	// Every candidate is a fresh draw of physical memory: the arrays tried before stay mapped until the search ends, so that a new one cannot be handed the
	// pages a slow one has just given back.  (What decides the class is WHERE the array lands, more than its kind: on a box in its slow state one kind, mapped
	// by fresh processes back to back -- each getting the pages its predecessor returned -- ran the workload at 7.0-7.6 ms a dozen times in a row, and kinds
	// mapped one after the other into the same freed space all probed alike; profiles/r04_placement_truth.txt.)  Bounded by the device's free memory and by
	// `candidates`; everything but the kept array is released before luw_create returns.
	std::vector<DevBlock> held;                                   // arrays tried and not kept (so far)
	for(int k=1; k<candidates&&rate(best_ms)<bar; k++) {
		size_t free_b = 0u, total_b = 0u;
		// room for one more array plus what the run may still allocate (statistics: 32 B per cell, staging, halo buffers)
		if(hipMemGetInfo(&free_b, &total_b)!=hipSuccess||free_b<bytes+40ull*s->kp.Np+(2ull<<30)) break;
		void* fi = nullptr;
		if(lead_alloc(s, &fi, elems, s->ddf_bytes, &alternatives[k-1])!=hipSuccess) { (void)hipGetLastError(); break; }
		DevBlock cand = std::move(s->raw.back()); s->raw.pop_back();
		void* const old_fi = s->d_fi;
		s->d_fi = fi;
		float ms = 0.0f;
		if(int e = step_ms(ms)) { s->d_fi = old_fi; dev_free(cand); for(DevBlock& h : held) dev_free(h); return e; }
		s->placement_tried++;
		// another draw has to be CLEARLY faster (3 %) to replace what is kept: a probe of a few steps resolves no less, and on a box where nothing reaches
		// the bar (all within 1 % of each other: profiles/r04_placement_10x.txt) every process then keeps the same one -- the default
		// (... or reach the bar where the kept one does not: on a box whose first draw probed 6.15 TB/s the second one's 6.30 was 2.4 % better -- and the
		// workload ran 7.08 ms on the first, 6.75 on such a draw)
		if(ms<0.97f*best_ms||(rate(ms)>=bar&&rate(best_ms)<bar)) { best_ms = ms; std::swap(s->raw.front(), cand); } // cand now holds the loser
		else s->d_fi = old_fi;
		held.push_back(std::move(cand));
	}
	// the losers' memory goes now; a mapped loser's (then empty) address range goes with the solver
	for(DevBlock& h : held) {
		const bool was_mapped = !h.chunks.empty();
		dev_free(h, was_mapped);
		if(was_mapped) s->raw.push_back(std::move(h));
	}
	s->placement_kept = dev_block_kind(s->raw.front()); s->placement_tbps = rate(best_ms)*1e-12;
	if(s->cfg.device<64) g_placement_kind[s->cfg.device].store(s->raw.front().chunks.empty() ? (size_t)0u : s->raw.front().chunk_bytes);
	// the probe steps left zeros, but be explicit
	HIP_TRY(hipMemsetAsync(s->raw.front().base, 0, std::min(s->raw.front().bytes, bytes+64u*s->ddf_bytes), s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}
