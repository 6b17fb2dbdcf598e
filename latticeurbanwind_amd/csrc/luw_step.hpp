// luw_step.hpp -- the schedule of ONE domain's share of a decomposed step, shared by both hosts: luw_group_* (all domains in one process, luw_group.hpp) and
// the one-process-per-GPU driver latticeurbanwind_amd/distributed.py (through luw_domain_step_*; the exchange between its calls is RCCL's).  Boxes (whole,
// interior, the disjoint shell slabs), the launches of a step on the two streams with the events that pipeline consecutive steps, and the C-ABI over them.
// Replaces, per domain, the body of LBM::do_time_step (FX/lbm.cpp:1262-1290) up to communicate_fi.  Included by luw_core.hip only, before luw_group.hpp.
#pragma once

#define GROUP_TRY(call) do { if(int e_ = (call)) return e_; } while(0)

// ---- Boxes: whole (non-halo cells), interior, and the disjoint shell slabs covering their difference (cells whose DDFs the pack kernels read).
static void step_axis_ranges(const uint32_t lN[3], const uint32_t H[3], const uint32_t x_shell, const int a, uint32_t nonhalo[2], uint32_t lo_slab[2],
	uint32_t hi_slab[2], uint32_t inner[2]) {
	nonhalo[0] = H[a]; nonhalo[1] = lN[a]-H[a];
	if(!H[a]) { lo_slab[0] = lo_slab[1] = hi_slab[0] = hi_slab[1] = 0u; inner[0] = 0u; inner[1] = lN[a]; return; }
	const uint32_t lo = nonhalo[0], hi = nonhalo[1];
	if(a!=0) { lo_slab[0] = lo; lo_slab[1] = lo+1u; hi_slab[0] = hi-1u; hi_slab[1] = hi; }   // y, z: the one cell layer next to the halo (whole rows)
	// x: whole blocks of x_shell cells from the first owned cell on (a one-cell x face would run one lane per wave and touch a full memory line per value)
	else {
		const uint32_t first_end = std::min(lo+x_shell, hi);
		const uint32_t last_start = std::max(lo+((hi-1u-lo)/x_shell)*x_shell, first_end);
		lo_slab[0] = lo; lo_slab[1] = first_end; hi_slab[0] = last_start; hi_slab[1] = hi;
	}
	inner[0] = lo_slab[1]; inner[1] = hi_slab[0];
}
static void step_boxes(const uint32_t lN[3], const uint32_t H[3], const uint32_t x_shell, Box& whole, Box& interior, std::vector<Box>& shell) {
	uint32_t nh[3][2], lo[3][2], hi[3][2], in[3][2];
	for(int a=0; a<3; a++) step_axis_ranges(lN, H, x_shell, a, nh[a], lo[a], hi[a], in[a]);
	whole = Box{ nh[0][0], nh[0][1], nh[1][0], nh[1][1], nh[2][0], nh[2][1] };
	interior = Box{ in[0][0], in[0][1], in[1][0], in[1][1], in[2][0], in[2][1] };
	shell.clear();
	uint32_t rng[3][2] = { { nh[0][0], nh[0][1] }, { nh[1][0], nh[1][1] }, { nh[2][0], nh[2][1] } };
	for(int a=0; a<3; a++) {
		if(!H[a]) continue;
		const uint32_t slabs[2][2] = { { lo[a][0], lo[a][1] }, { hi[a][0], hi[a][1] } };
		for(int k=0; k<2; k++) {
			uint32_t r[3][2] = { { rng[0][0], rng[0][1] }, { rng[1][0], rng[1][1] }, { rng[2][0], rng[2][1] } };
			r[a][0] = slabs[k][0]; r[a][1] = slabs[k][1];
			if(r[0][1]>r[0][0]&&r[1][1]>r[1][0]&&r[2][1]>r[2][0]) shell.push_back(Box{ r[0][0], r[0][1], r[1][0], r[1][1], r[2][0], r[2][1] });
		}
		rng[a][0] = in[a][0]; rng[a][1] = in[a][1]; // later axes exclude what this axis already covered
	}
}
// shell / interior overlap needs an interior: at least four owned layers on every split axis
static bool step_can_overlap(const uint32_t lN[3], const uint32_t H[3]) { for(int a=0; a<3; a++) if(H[a]&&lN[a]<6u) return false; return true; }

// what a step needs of one domain: its solver, its two streams, the events that order them, its boxes
struct StepCtx {
	luw_solver* s; hipStream_t compute, comm;
	hipEvent_t shell_done, interior_done, pre_done, stats_done; bool* stats_pending;
	bool overlap; const Box* whole; const Box* interior; const std::vector<Box>* shell;
};
// The kernels of one step.  Overlap: boundary shell on the communication stream (the exchange follows it there), interior on the compute stream, steps
// pipelined -- interior(t) waits for shell(t-1) only, shell(t) for interior(t-1) (and, by stream order, for the unpack of step t-1); the dependency
// analysis is tests/test_pipeline_hazards.py.  Otherwise the whole box on the compute stream.  t0/t1 (s0/s1): optional timing events around the
// interior or whole-box launch (the shell launches).
static int step_launch(const StepCtx& c, const int wf, hipEvent_t t0, hipEvent_t t1, hipEvent_t s0 = nullptr, hipEvent_t s1 = nullptr) {
	luw_solver* s = c.s;
	if(c.overlap) {
		// this step's shell rewrites the rho,u the last sample reads
		if((wf&1)&&*c.stats_pending) { HIP_TRY(hipStreamWaitEvent(c.comm, c.stats_done, 0)); *c.stats_pending = false; }
		HIP_TRY(hipStreamWaitEvent(c.compute, c.shell_done, 0));   // interior(t) needs shell(t-1) ...
		HIP_TRY(hipStreamWaitEvent(c.comm, c.interior_done, 0));   // ... shell(t) needs interior(t-1); both no-ops before the first record
		if(s->vk_active) { // pre_step_update (FX/setup.cpp:4872): rewrites u on TYPE_E inlet cells, read by shell and interior
			GROUP_TRY(luw_set_stream(s, c.compute)); GROUP_TRY(luw_vk_inlet_apply(s));
			HIP_TRY(hipEventRecord(c.pre_done, c.compute)); HIP_TRY(hipStreamWaitEvent(c.comm, c.pre_done, 0));
		}
		GROUP_TRY(luw_set_stream(s, c.comm));
		if(s0) HIP_TRY(hipEventRecord(s0, c.comm));
		for(const Box& b : *c.shell) GROUP_TRY(luw_enqueue_stream_collide(s, b.x0, b.x1, b.y0, b.y1, b.z0, b.z1, wf)); // boundary shell first ...
		if(s1) HIP_TRY(hipEventRecord(s1, c.comm));
		HIP_TRY(hipEventRecord(c.shell_done, c.comm));
		GROUP_TRY(luw_set_stream(s, c.compute));
		if(t0) HIP_TRY(hipEventRecord(t0, c.compute));
		const Box& in = *c.interior;
		GROUP_TRY(luw_enqueue_stream_collide(s, in.x0, in.x1, in.y0, in.y1, in.z0, in.z1, wf)); // ... interior overlaps the halo traffic
		if(t1) HIP_TRY(hipEventRecord(t1, c.compute));
		HIP_TRY(hipEventRecord(c.interior_done, c.compute));
	} else {
		GROUP_TRY(luw_set_stream(s, c.compute));
		if(s->vk_active) GROUP_TRY(luw_vk_inlet_apply(s));
		if(t0) HIP_TRY(hipEventRecord(t0, c.compute));
		const Box& w = *c.whole;
		GROUP_TRY(luw_enqueue_stream_collide(s, w.x0, w.x1, w.y0, w.y1, w.z0, w.z1, wf));
		if(t1) HIP_TRY(hipEventRecord(t1, c.compute));
	}
	return LUW_OK;
}
// thermal lattice / no fused statistics: the step wrote rho,u; the statistics kernel follows on the compute stream
static int step_separate_stats(const StepCtx& c) {
	if(c.overlap) HIP_TRY(hipStreamWaitEvent(c.compute, c.shell_done, 0));
	GROUP_TRY(luw_set_stream(c.s, c.compute));
	c.s->fields_current = true;
	GROUP_TRY(luw_stats_accumulate(c.s));
	HIP_TRY(hipEventRecord(c.stats_done, c.compute)); *c.stats_pending = true;
	return LUW_OK;
}

extern "C" {

// ---- the same schedule for a host that owns ONE domain per process (latticeurbanwind_amd/distributed.py): boxes, streams, events and the launches of a step
// live here; the caller does the exchange between luw_domain_step_launch calls (pack / RCCL / unpack on the communication stream it handed over).
struct luw_domain_step {
	luw_solver* s = nullptr; hipStream_t compute = nullptr, comm = nullptr;
	hipEvent_t shell_done = nullptr, interior_done = nullptr, pre_done = nullptr, stats_done = nullptr;
	bool stats_pending = false, overlap = false;
	Box whole{}, interior{}; std::vector<Box> shell;
	std::vector<hipEvent_t> timing; // per timed step: interior (or whole-box) start / end, shell start / end
};
static StepCtx domain_step_ctx(luw_domain_step* d) {
	return StepCtx{ d->s, d->compute, d->comm, d->shell_done, d->interior_done, d->pre_done, d->stats_done, &d->stats_pending, d->overlap, &d->whole,
		&d->interior, &d->shell };
}
static void domain_step_drop_timing(luw_domain_step* d) { for(hipEvent_t e : d->timing) if(e) (void)hipEventDestroy(e); d->timing.clear(); }
int luw_step_boxes(const uint32_t* local_N, const uint32_t* halo, uint32_t x_shell, uint32_t* whole6, uint32_t* interior6, uint32_t* shell_boxes,
	uint32_t* shell_count, int* can_overlap) {
	if(!local_N||!halo||x_shell==0u) return fail(LUW_ERR_INVALID, "luw_step_boxes: bad argument");
	Box w, in; std::vector<Box> sh;
	step_boxes(local_N, halo, x_shell, w, in, sh);
	auto put = [](uint32_t* o, const Box& b) { o[0] = b.x0; o[1] = b.x1; o[2] = b.y0; o[3] = b.y1; o[4] = b.z0; o[5] = b.z1; };
	if(whole6) put(whole6, w);
	if(interior6) put(interior6, in);
	if(shell_boxes) for(size_t k=0; k<sh.size(); k++) put(shell_boxes+6u*k, sh[k]);
	if(shell_count) *shell_count = (uint32_t)sh.size();
	if(can_overlap) *can_overlap = step_can_overlap(local_N, halo) ? 1 : 0;
	return LUW_OK;
}
void luw_domain_step_destroy(luw_domain_step* d) {
	if(!d) return;
	if(d->s) (void)hipSetDevice(d->s->cfg.device);
	domain_step_drop_timing(d);
	for(hipEvent_t e : { d->shell_done, d->interior_done, d->pre_done, d->stats_done }) if(e) (void)hipEventDestroy(e);
	delete d;
}
int luw_domain_step_create(luw_solver* s, void* compute_stream, void* comm_stream, uint32_t x_shell, int overlap, luw_domain_step** out) {
	if(!s||!out||!compute_stream) return fail(LUW_ERR_INVALID, "luw_domain_step_create: bad argument");
	*out = nullptr;
	if(int e = set_device(s)) return e;
	std::unique_ptr<luw_domain_step, void(*)(luw_domain_step*)> d(new luw_domain_step(), luw_domain_step_destroy);
	d->s = s; d->compute = (hipStream_t)compute_stream; d->comm = (hipStream_t)comm_stream;
	const uint32_t lN[3] = { s->cfg.Nx, s->cfg.Ny, s->cfg.Nz }, H[3] = { s->kp.halo_x, s->kp.halo_y, s->kp.halo_z };
	const uint32_t X = x_shell ? x_shell : 128u;
	step_boxes(lN, H, X, d->whole, d->interior, d->shell);
	d->overlap = overlap!=0 && comm_stream!=nullptr && step_can_overlap(lN, H);
	for(hipEvent_t* e : { &d->shell_done, &d->interior_done, &d->pre_done, &d->stats_done }) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
	*out = d.release();
	return LUW_OK;
}
int luw_domain_step_overlaps(const luw_domain_step* d) { return (d&&d->overlap) ? 1 : 0; }
int luw_domain_step_launch(luw_domain_step* d, int write_fields, int timed) {
	if(!d) return fail(LUW_ERR_INVALID, "luw_domain_step_launch: null argument");
	if(int e = set_device(d->s)) return e;
	hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
	if(timed) for(hipEvent_t& e : ev) { HIP_TRY(hipEventCreate(&e)); d->timing.push_back(e); }   // (owned by d->timing at once: a failure leaks none)
	return step_launch(domain_step_ctx(d), write_fields, ev[0], ev[1], d->overlap ? ev[2] : nullptr, d->overlap ? ev[3] : nullptr);
}
int luw_domain_step_separate_stats(luw_domain_step* d) {
	if(!d) return fail(LUW_ERR_INVALID, "luw_domain_step_separate_stats: null argument");
	if(int e = set_device(d->s)) return e;
	return step_separate_stats(domain_step_ctx(d));
}
// waits for both streams; 0 timed steps -> zeros.  shell_ms is -1 without shell / interior overlap
int luw_domain_step_timing(luw_domain_step* d, double* kernel_ms, double* shell_ms) {
	if(!d||!kernel_ms||!shell_ms) return fail(LUW_ERR_INVALID, "luw_domain_step_timing: null argument");
	if(int e = set_device(d->s)) return e;
	if(d->comm) HIP_TRY(hipStreamSynchronize(d->comm));
	HIP_TRY(hipStreamSynchronize(d->compute));
	double k = 0.0, sh = 0.0; const size_t n = d->timing.size()/4u;
	for(size_t i=0; i<n; i++) {
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, d->timing[4u*i], d->timing[4u*i+1u])); k += (double)ms;
		if(d->overlap) { HIP_TRY(hipEventElapsedTime(&ms, d->timing[4u*i+2u], d->timing[4u*i+3u])); sh += (double)ms; }
	}
	*kernel_ms = n ? k/(double)n : 0.0; *shell_ms = d->overlap ? (n ? sh/(double)n : 0.0) : -1.0;
	domain_step_drop_timing(d);
	return LUW_OK;
}

} // extern "C"
