// luw_group.hpp -- the multi-domain host runtime: ONE process, D = Dx*Dy*Dz domains, one HIP device + one compute / one
// communication stream per domain.  This is the reference's `LBM` object for D > 1 (FX/lbm.cpp:1057-1112 constructor,
// :1221-1260 initialize, :1262-1290 do_time_step, :1907-1935 communicate_field) behind the C-ABI (luw_group_* in luw_core.h).
// Host code of libluw_core.so; included by luw_core.hip only, after the single-domain entry points it is built on.
//
// What is the reference's: block decomposition, domain id d = x + (y + z*Dy)*Dx, local extents N/D + 2 on split axes, offsets,
// periodic neighbour (x+1)%Dx, per step and per axis x, y, z: extract the 5 outgoing DDFs of both faces (faces include the halo
// rims), swap with the two neighbours, insert; one odd-t exchange at initialisation.
// What is ours (MI355X): nothing is staged through the host.  With peer access (xGMI between the GPUs of a node, or several
// domains on one device) the pack kernel of a domain writes its face STRAIGHT into the neighbour's receive buffer -- a remote
// store stream over xGMI, no copy engine, no intermediate buffer; without peer access the face goes through
// hipMemcpyPeerAsync.  The boundary shell of a domain runs first on its communication stream, its faces travel while the
// interior box runs on the compute stream, and steps are pipelined: interior(t+1) waits for shell(t) only (the dependency
// analysis is the one of latticeurbanwind_amd/distributed.py, held by tests/test_pipeline_hazards.py).  All ordering is done
// with HIP events between streams; the host thread only enqueues and synchronises once per luw_group_run call.
#pragma once

struct GroupDomain {
	luw_solver* s = nullptr;
	int device = 0;
	uint32_t coord[3] = { 0u, 0u, 0u };
	uint32_t lN[3] = { 0u, 0u, 0u };
	int32_t O[3] = { 0, 0, 0 };
	uint32_t nbr[3][2] = {};                  // [axis][0: + neighbour, 1: - neighbour] (domain ids, periodic)
	hipStream_t compute = nullptr, comm = nullptr;
	void* recv[3][2] = {};                    // [axis][0: face coming from the + neighbour (insert's buf_p), 1: from the - neighbour (buf_m)], 5*A DDF elements
	void* send[3][2] = {};                    // staging, only for neighbours without peer access
	void* grecv[3][2] = {}; void* gsend[3][2] = {}; // thermal lattice: one population per face cell
	// one-phase exchange (group_exchange_one_phase): the x faces arrive in TWO sets of receive buffers used in turn (set 0 is recv[0]: a step's kernels
	// read one set in place while the neighbours' kernels of the same step fill the other); the twelve edge messages land in erecv[e] (population 7 + e,
	// sent by the domain in direction -c, enbr[e] being the domain in direction +c that takes OUR edge e); nbrs: every domain this one trades with
	void* recvx[2][2] = {};
	void* erecv[12] = {};
	void* esend[12] = {};                     // staging of the edge messages where they do not travel as peer stores (staged copies, RCCL)
	uint32_t enbr[12] = {};
	std::vector<uint32_t> nbrs;
	hipEvent_t packed_all = nullptr, unpacked_all = nullptr;
	hipEvent_t shell_done = nullptr, interior_done = nullptr, pre_done = nullptr, stats_done = nullptr;
	hipEvent_t packed[3] = {}, unpacked[3] = {}, gpacked[3] = {}, gunpacked[3] = {};
	bool stats_pending = false;
	Box whole{}, interior{};
	std::vector<Box> shell;
	std::vector<uint32_t> gather_src;         // probe gather: positions of this domain's cells in the caller's list
	// [DDFs / thermal][axis]: number of the last exchange whose pack / unpack this domain has enqueued (threaded runs)
	uint64_t packed_seq[2][3] = {}, unpacked_seq[2][3] = {};
};

struct luw_group {
	luw_config gcfg{};                        // the GLOBAL lattice in Nx,Ny,Nz
	uint32_t D[3] = { 1u, 1u, 1u }, gN[3] = { 1u, 1u, 1u }, H[3] = { 0u, 0u, 0u };
	std::vector<GroupDomain> dom;
	std::vector<std::vector<char>> peer;      // peer[a][b]: kernels of domain a's device may write domain b's memory
	bool overlap = false, initialized = false, thermal = false;
	size_t ddf_bytes = 4u;
	uint64_t t = 0ull;
	uint64_t exchanges = 0ull;                // halo exchanges done by luw_group_run calls (threaded runs number them)
	uint32_t gather_total = 0u;
	int transport = LUW_TRANSPORT_PEER;       // how faces travel between domains (luw_group_transport)
	std::vector<void*> rccl_comm;             // LUW_TRANSPORT_RCCL: one communicator per DISTINCT device ...
	std::vector<int> rccl_rank;               // ... and every domain's rank in it (domains sharing a device share the rank)
	bool failed = false;                      // a run stopped half-way: streams and sequence numbers are not trustworthy any more
	bool one_phase = false;                   // every face, the edge messages and the thermal faces in one pack / unpack round per step (peer stores)
	uint32_t xset = 0u;                       // one-phase: the set of x receive buffers the NEXT exchange fills
};

// thickness of the x boundary slabs: 128 cells (a one-cell x face would run one lane per wave; 128 FP16C cells are a full wave of the pair kernel,
// which narrower slabs would leave to the one-cell kernel; FP32: 16 / 32-cell slabs are slower, 64 / 128 / 256 equal within the spread between
// fresh processes, profiles/r03_ab_rank_shape_xshell.txt); LUW_X_SHELL overrides (A/B aid)
static uint32_t group_x_shell(const luw_group* g) {
	(void)g;
	return tuning().x_shell ? tuning().x_shell : 128u;
}

static void group_boxes(const luw_group* g, GroupDomain& d) { step_boxes(d.lN, g->H, group_x_shell(g), d.whole, d.interior, d.shell); }

static void group_rccl_teardown(luw_group* g);
// the x faces of domain i go straight into its neighbours' receive buffers (peer stores)
static bool group_x_direct(const luw_group* g, const size_t i) { return g->H[0]&&g->peer[i][g->dom[i].nbr[0][0]]&&g->peer[i][g->dom[i].nbr[0][1]]; }
// ... and then the step kernels of domain i write there (luw_set_x_face_buffers): the launch that holds them waits, like the pack kernel would, until both
// neighbours have consumed what the previous step put into those buffers
static int group_x_face_ready(luw_group* g, const size_t i, const uint32_t xs) { // xs: the set of x receive buffers this step fills (one-phase exchange)
	if(!group_x_direct(g, i)||tuning().group_x_packed) return LUW_OK;
	GroupDomain& d = g->dom[i];
	hipStream_t st = g->overlap ? d.comm : d.compute;
	if(g->one_phase) {
		// this step's kernels fill set xs of the x neighbours' receive buffers: the launches that READ that set -- the neighbours' previous step, which
		// their `packed_all` record follows on the same stream -- have to be through; the set those launches are reading now is the other one
		GroupDomain& P = g->dom[d.nbr[0][0]]; GroupDomain& M = g->dom[d.nbr[0][1]];
		HIP_TRY(hipStreamWaitEvent(st, P.packed_all, 0)); HIP_TRY(hipStreamWaitEvent(st, M.packed_all, 0));
		return luw_set_x_face_buffers(d.s, P.recvx[xs][1], M.recvx[xs][0]);
	}
	HIP_TRY(hipStreamWaitEvent(st, g->dom[d.nbr[0][0]].unpacked[0], 0));
	HIP_TRY(hipStreamWaitEvent(st, g->dom[d.nbr[0][1]].unpacked[0], 0));
	return LUW_OK;
}
static void group_free(luw_group* g) {
	if(!g) return;
	for(GroupDomain& d : g->dom) {
		(void)hipSetDevice(d.device);
		if(d.compute) (void)hipStreamSynchronize(d.compute);
		if(d.comm) (void)hipStreamSynchronize(d.comm);
	}
	group_rccl_teardown(g);
	for(GroupDomain& d : g->dom) {
		(void)hipSetDevice(d.device);
		if(d.s) { (void)luw_set_stream(d.s, nullptr); luw_destroy(d.s); }
		for(int k=0; k<2; k++) (void)hipFree(d.recvx[1][k]);         // (set 0 is recv[0])
		for(void* e : d.erecv) (void)hipFree(e);
		for(void* e : d.esend) (void)hipFree(e);
		for(hipEvent_t e : { d.packed_all, d.unpacked_all }) if(e) (void)hipEventDestroy(e);
		for(int a=0; a<3; a++) for(int k=0; k<2; k++) {
			(void)hipFree(d.recv[a][k]);
			(void)hipFree(d.send[a][k]);
			(void)hipFree(d.grecv[a][k]);
			(void)hipFree(d.gsend[a][k]);
		}
		for(hipEvent_t e : { d.shell_done, d.interior_done, d.pre_done, d.stats_done }) if(e) (void)hipEventDestroy(e);
		for(int a=0; a<3; a++) for(hipEvent_t e : { d.packed[a], d.unpacked[a], d.gpacked[a], d.gunpacked[a] }) if(e) (void)hipEventDestroy(e);
		if(d.compute) (void)hipStreamDestroy(d.compute);
		if(d.comm) (void)hipStreamDestroy(d.comm);
	}
	delete g;
}

static int group_set_device(const GroupDomain& d) { HIP_TRY(hipSetDevice(d.device)); return LUW_OK; }

// ---- the halo exchange of one field (thermal_pass false: 5 DDFs per face cell; true: the thermal lattice's single population), per
// domain and axis in two halves: pack (into the neighbours' receive buffers: peer stores, or staged through a copy) and unpack (once
// both neighbours have delivered).  On the domain's communication stream, or its compute stream (on_compute).
static int domain_pack(luw_group* g, const size_t i, const int a, const bool thermal_pass, const bool on_compute) {
	GroupDomain& d = g->dom[i];
	hipStream_t st = on_compute ? d.compute : d.comm;
	GroupDomain& P = g->dom[d.nbr[a][0]]; GroupDomain& M = g->dom[d.nbr[a][1]];
	hipEvent_t* unp = thermal_pass ? P.gunpacked : P.unpacked; hipEvent_t* unm = thermal_pass ? M.gunpacked : M.unpacked;
	HIP_TRY(hipStreamWaitEvent(st, unp[a], 0)); HIP_TRY(hipStreamWaitEvent(st, unm[a], 0)); // the neighbours have consumed what the previous step put there
	// my + face lands in the + neighbour's "from the - side" buffer, my - face in the - neighbour's "from the + side" buffer
	void* dst_p = (thermal_pass ? P.grecv : P.recv)[a][1]; void* dst_m = (thermal_pass ? M.grecv : M.recv)[a][0];
	const bool direct = g->peer[i][d.nbr[a][0]]&&g->peer[i][d.nbr[a][1]];
	void* out_p = direct ? dst_p : (thermal_pass ? d.gsend : d.send)[a][0]; void* out_m = direct ? dst_m : (thermal_pass ? d.gsend : d.send)[a][1];
	GROUP_TRY(luw_set_stream(d.s, st));
	GROUP_TRY(thermal_pass ? luw_enqueue_extract_gi(d.s, (uint32_t)a, out_p, out_m) : luw_enqueue_extract_fi(d.s, (uint32_t)a, out_p, out_m));
	if(!direct) {
		const size_t bytes = (size_t)luw_get_area(d.s, (uint32_t)a)*(thermal_pass ? 1u : 5u)*g->ddf_bytes;
		HIP_TRY(hipMemcpyPeerAsync(dst_p, P.device, out_p, d.device, bytes, st));
		HIP_TRY(hipMemcpyPeerAsync(dst_m, M.device, out_m, d.device, bytes, st));
	}
	HIP_TRY(hipEventRecord((thermal_pass ? d.gpacked : d.packed)[a], st));
	return LUW_OK;
}
static int domain_unpack(luw_group* g, const size_t i, const int a, const bool thermal_pass, const bool on_compute) {
	GroupDomain& d = g->dom[i];
	hipStream_t st = on_compute ? d.compute : d.comm;
	GroupDomain& P = g->dom[d.nbr[a][0]]; GroupDomain& M = g->dom[d.nbr[a][1]];
	if(!(g_injected_faults.load()&LUW_FAULT_UNPACK_WITHOUT_WAIT)) {   // (test hook: the dependency the schedule fuzz must miss when it is gone)
		HIP_TRY(hipStreamWaitEvent(st, (thermal_pass ? P.gpacked : P.packed)[a], 0));
		HIP_TRY(hipStreamWaitEvent(st, (thermal_pass ? M.gpacked : M.packed)[a], 0));
	}
	GROUP_TRY(luw_set_stream(d.s, st));
	GROUP_TRY(thermal_pass ? luw_enqueue_insert_gi(d.s, (uint32_t)a, d.grecv[a][0], d.grecv[a][1])
		: luw_enqueue_insert_fi(d.s, (uint32_t)a, d.recv[a][0], d.recv[a][1]));
	HIP_TRY(hipEventRecord((thermal_pass ? d.gunpacked : d.unpacked)[a], st));
	return LUW_OK;
}
// a communicator works on ONE stream per batch: the stream of the first domain on its device (on a node every domain is its own leader)
static size_t group_leader(const luw_group* g, const size_t i) { for(size_t k=0; k<i; k++) if(g->dom[k].device==g->dom[i].device) return k; return i; }
// one message of the one-round exchange where faces do not travel as peer stores: from a send buffer of domain src into a receive buffer of domain dst
struct GroupMsg { uint32_t src, dst; const void* from; void* into; size_t bytes; };
static void domain_messages(luw_group* g, const size_t i, const uint32_t xs, std::vector<GroupMsg>& out);
#include "luw_group_rccl.hpp"   // LUW_GROUP_TRANSPORT=rccl: grouped ncclSend / ncclRecv (librccl looked up at run time)

// ---- the exchange in ONE phase (default where every pair of trading domains has peer access): what latticeurbanwind_amd/distributed.py does per rank over
// RCCL (_communicate_one_phase), with peer stores.  Per step and domain ONE pack round -- the y / z faces by their pack kernels and the twelve edge
// populations (k_edges: the population that crosses two cuts at once goes straight to the diagonal neighbour instead of riding in the rims of two
// consecutive face exchanges, FX/lbm.cpp:1908-1934), all written into the receivers' buffers; the x faces are there already, written by the step kernels --
// and ONE unpack round: the x faces are handed to the next step's kernels where they lie (luw_set_x_face_inputs: no unpack kernel), y / z faces inserted,
// the edges last (they overwrite what the rims of the faces carried).  Same populations in the same slots as the three-phase route
// (tests/test_gpu_group.py, tests/fuzz/fuzz_exchange_gpu.py hold both to each other and to the oracle).
// Where the faces do not travel as peer stores (LUW_GROUP_TRANSPORT=staged / rccl) the same round packs into the domain's own send buffers -- the x faces
// are there already, written by the step kernels -- and every message (two faces per split axis, the edges, the thermal faces) is then moved into the
// receiver's buffer: hipMemcpyPeerAsync behind the pack kernels (staged), or ONE grouped batch of ncclSend / ncclRecv for the whole group (rccl,
// group_exchange_rccl_all below).
static void domain_messages(luw_group* g, const size_t i, const uint32_t xs, std::vector<GroupMsg>& out) {
	GroupDomain& d = g->dom[i];
	for(int a=0; a<3; a++) if(g->H[a]) for(int k=0; k<2; k++) {
		GroupDomain& to = g->dom[d.nbr[a][k]];
		const size_t A = (size_t)luw_get_area(d.s, (uint32_t)a);
		// my + face (k = 0) is what the + neighbour receives "from its - side" ([1]); my - face lands in the - neighbour's [0]
		out.push_back(GroupMsg{ (uint32_t)i, d.nbr[a][k], d.send[a][k], a==0 ? to.recvx[xs][1-k] : to.recv[a][1-k], 5u*A*g->ddf_bytes });
		if(g->thermal) out.push_back(GroupMsg{ (uint32_t)i, d.nbr[a][k], d.gsend[a][k], to.grecv[a][1-k], A*g->ddf_bytes });
	}
	for(uint32_t e=0u; e<12u; e++) if(const uint64_t L = luw_get_edge_length(d.s, e))
		out.push_back(GroupMsg{ (uint32_t)i, d.enbr[e], d.esend[e], g->dom[d.enbr[e]].erecv[e], (size_t)L*g->ddf_bytes });
}
static int domain_pack_all(luw_group* g, const size_t i, const bool on_compute, const uint32_t xs) {
	GroupDomain& d = g->dom[i];
	hipStream_t st = on_compute ? d.compute : d.comm;
	const bool direct = g->transport==LUW_TRANSPORT_PEER;   // (one round with peer stores: every pair of trading domains has peer access, luw_group_create)
	for(const uint32_t nb : d.nbrs) HIP_TRY(hipStreamWaitEvent(st, g->dom[nb].unpacked_all, 0)); // they have consumed what the previous step put there
	GROUP_TRY(luw_set_stream(d.s, st));
	for(int a=0; a<3; a++) {
		if(!g->H[a]) continue;
		GroupDomain& P = g->dom[d.nbr[a][0]]; GroupDomain& M = g->dom[d.nbr[a][1]];
		// (x: no launch where this step's kernels have written both faces -- into these very buffers, group_x_face_ready / luw_group_create)
		if(!direct) GROUP_TRY(luw_enqueue_extract_fi(d.s, (uint32_t)a, d.send[a][0], d.send[a][1]));
		else if(a==0) GROUP_TRY(luw_enqueue_extract_fi(d.s, 0u, P.recvx[xs][1], M.recvx[xs][0]));
		else GROUP_TRY(luw_enqueue_extract_fi(d.s, (uint32_t)a, P.recv[a][1], M.recv[a][0]));
	}
	void* out[12];
	for(uint32_t e=0u; e<12u; e++) out[e] = !luw_get_edge_length(d.s, e) ? nullptr : direct ? g->dom[d.enbr[e]].erecv[e] : d.esend[e];
	GROUP_TRY(luw_enqueue_extract_edges(d.s, out));
	if(g->thermal) for(int a=0; a<3; a++) if(g->H[a]) {
		if(direct) GROUP_TRY(luw_enqueue_extract_gi(d.s, (uint32_t)a, g->dom[d.nbr[a][0]].grecv[a][1], g->dom[d.nbr[a][1]].grecv[a][0]));
		else GROUP_TRY(luw_enqueue_extract_gi(d.s, (uint32_t)a, d.gsend[a][0], d.gsend[a][1]));
	}
	if(g->transport==LUW_TRANSPORT_STAGED) {
		std::vector<GroupMsg> msgs;
		domain_messages(g, i, xs, msgs);
		for(const GroupMsg& m : msgs) HIP_TRY(hipMemcpyPeerAsync(m.into, g->dom[m.dst].device, m.from, d.device, m.bytes, st));
	}
	HIP_TRY(hipEventRecord(d.packed_all, st));
	return LUW_OK;
}
// x_in_place: the x faces stay in their receive buffers for the next step's kernels (not at initialisation, whose exchange is followed by a reset of t)
static int domain_unpack_all(luw_group* g, const size_t i, const bool on_compute, const bool x_in_place, const uint32_t xs) {
	GroupDomain& d = g->dom[i];
	hipStream_t st = on_compute ? d.compute : d.comm;
	if(!(g_injected_faults.load()&LUW_FAULT_UNPACK_WITHOUT_WAIT)) {
		for(const uint32_t nb : d.nbrs) HIP_TRY(hipStreamWaitEvent(st, g->dom[nb].packed_all, 0));
		// rccl: what this domain receives arrives on its device's leader stream (group_exchange_rccl_all records packed[0] behind the batch there)
		if(g->transport==LUW_TRANSPORT_RCCL) HIP_TRY(hipStreamWaitEvent(st, g->dom[group_leader(g, i)].packed[0], 0));
	}
	GROUP_TRY(luw_set_stream(d.s, st));
	if(g->H[0]) {
		if(x_in_place) GROUP_TRY(luw_set_x_face_inputs(d.s, d.recvx[xs][0], d.recvx[xs][1]));
		else GROUP_TRY(luw_enqueue_insert_fi(d.s, 0u, d.recvx[xs][0], d.recvx[xs][1]));
	}
	for(int a=1; a<3; a++) if(g->H[a]) GROUP_TRY(luw_enqueue_insert_fi(d.s, (uint32_t)a, d.recv[a][0], d.recv[a][1]));
	GROUP_TRY(luw_enqueue_insert_edges(d.s, d.erecv));
	if(g->thermal) for(int a=0; a<3; a++) if(g->H[a]) GROUP_TRY(luw_enqueue_insert_gi(d.s, (uint32_t)a, d.grecv[a][0], d.grecv[a][1]));
	HIP_TRY(hipEventRecord(d.unpacked_all, st));
	return LUW_OK;
}
static int group_exchange_one_phase(luw_group* g, const bool on_compute, const bool x_in_place) {
	for(size_t i=0; i<g->dom.size(); i++) { GROUP_TRY(group_set_device(g->dom[i])); GROUP_TRY(domain_pack_all(g, i, on_compute, g->xset)); }
	if(g->transport==LUW_TRANSPORT_RCCL) GROUP_TRY(group_exchange_rccl_all(g, on_compute, g->xset));
	for(size_t i=0; i<g->dom.size(); i++) { GROUP_TRY(group_set_device(g->dom[i])); GROUP_TRY(domain_unpack_all(g, i, on_compute, x_in_place, g->xset)); }
	g->xset ^= 1u;
	return LUW_OK;
}

// one host thread for all domains: every domain packs, then every domain unpacks, axis by axis
static int group_exchange(luw_group* g, const bool thermal_pass, const bool on_compute) {
	for(int a=0; a<3; a++) {
		if(!g->H[a]) continue;
		if(g->transport==LUW_TRANSPORT_RCCL) { GROUP_TRY(group_exchange_rccl_axis(g, a, thermal_pass, on_compute)); continue; }
		for(size_t i=0; i<g->dom.size(); i++) { GROUP_TRY(group_set_device(g->dom[i])); GROUP_TRY(domain_pack(g, i, a, thermal_pass, on_compute)); }
		for(size_t i=0; i<g->dom.size(); i++) { GROUP_TRY(group_set_device(g->dom[i])); GROUP_TRY(domain_unpack(g, i, a, thermal_pass, on_compute)); }
	}
	return LUW_OK;
}
static int group_communicate(luw_group* g, const bool on_compute, const bool x_in_place = true) { // communicate_fi, then communicate_gi (FX/lbm.cpp:1266-1284)
	if(g->one_phase) return group_exchange_one_phase(g, on_compute, x_in_place);
	GROUP_TRY(group_exchange(g, false, on_compute));
	if(g->thermal) GROUP_TRY(group_exchange(g, true, on_compute));
	return LUW_OK;
}
static int group_join(luw_group* g) {
	for(GroupDomain& d : g->dom) {
		GROUP_TRY(group_set_device(d));
		HIP_TRY(hipStreamSynchronize(d.comm)); HIP_TRY(hipStreamSynchronize(d.compute));
	}
	return LUW_OK;
}

// what one step means for one domain, besides the exchange: kernels of step i of a luw_group_run call
struct GroupStepPlan { bool sampled, fused, separate; int wf; };
static int domain_plan_step(luw_group* g, const size_t k, const uint64_t i, const uint64_t steps, const uint64_t first_sample, const uint64_t stride,
	GroupStepPlan& pl) {
	const bool every = luw_fields_every_step(g->dom[k].s)!=0; // the option, or a fluid reference cell on one of this domain's faces (luw_initialize)
	pl.sampled = stride>0ull && i+1ull>=first_sample && (i+1ull-first_sample)%stride==0ull;
	pl.fused = false;
	if(pl.sampled) { int f = 0; GROUP_TRY(luw_stats_begin_sample(g->dom[k].s, &f)); pl.fused = f!=0; } // every domain answers alike (same kernels everywhere)
	pl.separate = pl.sampled&&!pl.fused;
	pl.wf = ((every||i+1ull==steps||pl.separate) ? 1 : 0)|(pl.fused ? LUW_WF_SAMPLE : 0);
	return LUW_OK;
}
static StepCtx group_step_ctx(luw_group* g, const size_t k) {
	GroupDomain& d = g->dom[k];
	return StepCtx{ d.s, d.compute, d.comm, d.shell_done, d.interior_done, d.pre_done, d.stats_done, &d.stats_pending, g->overlap, &d.whole, &d.interior,
		&d.shell };
}
static int domain_launch_step(luw_group* g, const size_t k, const GroupStepPlan& pl, hipEvent_t t0, hipEvent_t t1, const uint32_t xs) {
	GROUP_TRY(group_x_face_ready(g, k, xs));
	return step_launch(group_step_ctx(g, k), pl.wf, t0, t1);
}
static int domain_separate_stats(luw_group* g, const size_t k) { return step_separate_stats(group_step_ctx(g, k)); }

// ---- one host thread PER DOMAIN (opt-in, LUW_GROUP_THREADS=1; for hosts where one enqueueing thread -- ~25 runtime calls per domain and
// step -- would not keep eight devices busy).
// A stream can only wait for an event that has already been RECORDED (hipStreamWaitEvent on an unrecorded event is a no-op), and the
// records now happen in other threads: each domain publishes, per axis, the number of the last exchange whose pack / unpack it has
// enqueued (a release store behind the hipEventRecord), and a neighbour spins on that number (acquire) before it enqueues its wait.
// The same numbers keep an event from being re-recorded before every neighbour has enqueued its wait on the previous record.
struct GroupThreads {
	std::atomic<int> abort{0};
	std::vector<std::string> error;
};
static bool group_wait_seq(const uint64_t* seq, const uint64_t want, const std::atomic<int>& abort) {
	for(uint64_t spins=0ull; __atomic_load_n(seq, __ATOMIC_ACQUIRE)<want; spins++) {
		if(abort.load(std::memory_order_relaxed)) return false;
		if(spins>64ull) std::this_thread::yield();
	}
	return true;
}
static int domain_exchange_threaded(luw_group* g, const size_t k, const bool thermal_pass, const bool on_compute, const uint64_t X, GroupThreads& T) {
	GroupDomain& d = g->dom[k];
	const int f = thermal_pass ? 1 : 0;
	for(int a=0; a<3; a++) {
		if(!g->H[a]) continue;
		GroupDomain& P = g->dom[d.nbr[a][0]]; GroupDomain& M = g->dom[d.nbr[a][1]];
		// the neighbours have enqueued (and recorded) the unpack of the previous exchange: their buffers may be written again after it
		if(!group_wait_seq(&P.unpacked_seq[f][a], X-1ull, T.abort)||!group_wait_seq(&M.unpacked_seq[f][a], X-1ull, T.abort)) return LUW_ERR_STATE;
		GROUP_TRY(domain_pack(g, k, a, thermal_pass, on_compute));
		__atomic_store_n(&d.packed_seq[f][a], X, __ATOMIC_RELEASE);
		if(!group_wait_seq(&P.packed_seq[f][a], X, T.abort)||!group_wait_seq(&M.packed_seq[f][a], X, T.abort)) return LUW_ERR_STATE;
		GROUP_TRY(domain_unpack(g, k, a, thermal_pass, on_compute));
		__atomic_store_n(&d.unpacked_seq[f][a], X, __ATOMIC_RELEASE);
	}
	return LUW_OK;
}
// one-phase exchange number X of one domain's thread: the counters packed_seq[0][0] / unpacked_seq[0][0] stand for the whole round
static int domain_exchange_one_phase_threaded(luw_group* g, const size_t k, const bool on_compute, const uint64_t X, GroupThreads& T, const uint32_t xs) {
	GroupDomain& d = g->dom[k];
	for(const uint32_t nb : d.nbrs) if(!group_wait_seq(&g->dom[nb].unpacked_seq[0][0], X-1ull, T.abort)) return LUW_ERR_STATE;
	GROUP_TRY(domain_pack_all(g, k, on_compute, xs));
	__atomic_store_n(&d.packed_seq[0][0], X, __ATOMIC_RELEASE);
	for(const uint32_t nb : d.nbrs) if(!group_wait_seq(&g->dom[nb].packed_seq[0][0], X, T.abort)) return LUW_ERR_STATE;
	GROUP_TRY(domain_unpack_all(g, k, on_compute, true, xs));
	__atomic_store_n(&d.unpacked_seq[0][0], X, __ATOMIC_RELEASE);
	return LUW_OK;
}
static int domain_run_threaded(luw_group* g, const size_t k, const uint64_t steps, const uint64_t first_sample, const uint64_t stride,
	std::vector<hipEvent_t>* tev, const uint64_t X0, GroupThreads& T) {
	GroupDomain& d = g->dom[k];
	GROUP_TRY(group_set_device(d)); // per host thread
	const uint32_t xset0 = g->xset;   // (the group's set counter moves with the steps; threads derive it from the step number)
	for(uint64_t i=0ull; i<steps; i++) {
		GroupStepPlan pl;
		GROUP_TRY(domain_plan_step(g, k, i, steps, first_sample, stride, pl));
		if(g->one_phase) {
			// the x neighbours have enqueued (and recorded) the pack round of the previous exchange: group_x_face_ready waits for that record
			if(group_x_direct(g, k)&&(!group_wait_seq(&g->dom[d.nbr[0][0]].packed_seq[0][0], X0+i, T.abort)
				||!group_wait_seq(&g->dom[d.nbr[0][1]].packed_seq[0][0], X0+i, T.abort))) return LUW_ERR_STATE;
			GROUP_TRY(domain_launch_step(g, k, pl, tev ? (*tev)[2u*i] : nullptr, tev ? (*tev)[2u*i+1u] : nullptr, (xset0+(uint32_t)i)&1u));
			GROUP_TRY(domain_exchange_one_phase_threaded(g, k, !g->overlap, X0+i+1ull, T, (xset0+(uint32_t)i)&1u));
			if(pl.separate) GROUP_TRY(domain_separate_stats(g, k));
			GROUP_TRY(luw_increment_time_step(d.s, 1ull));
			continue;
		}
		if(group_x_direct(g, k)) { // the neighbours have enqueued (and recorded) the x unpack of the previous exchange: domain_launch_step may wait for it
			if(!group_wait_seq(&g->dom[d.nbr[0][0]].unpacked_seq[0][0], X0+i, T.abort)||!group_wait_seq(&g->dom[d.nbr[0][1]].unpacked_seq[0][0], X0+i, T.abort))
				return LUW_ERR_STATE;
		}
		GROUP_TRY(domain_launch_step(g, k, pl, tev ? (*tev)[2u*i] : nullptr, tev ? (*tev)[2u*i+1u] : nullptr, 0u));
		GROUP_TRY(domain_exchange_threaded(g, k, false, !g->overlap, X0+i+1ull, T));
		if(g->thermal) GROUP_TRY(domain_exchange_threaded(g, k, true, !g->overlap, X0+i+1ull, T));
		if(pl.separate) GROUP_TRY(domain_separate_stats(g, k));
		GROUP_TRY(luw_increment_time_step(d.s, 1ull));
	}
	HIP_TRY(hipStreamSynchronize(d.comm)); HIP_TRY(hipStreamSynchronize(d.compute));
	return LUW_OK;
}

static int group_run_steps(luw_group* g, const uint64_t steps, const uint64_t first_sample, const uint64_t stride, double* mean_kernel_ms);
// A run that stops half-way (a launch failed, a domain thread gave up) leaves work enqueued on some streams and the threaded runs' sequence
// numbers out of step: drain what is there, then refuse further runs instead of letting the next one wait for an exchange that never comes.
static int group_run(luw_group* g, const uint64_t steps, const uint64_t first_sample, const uint64_t stride, double* mean_kernel_ms) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_run: null group");
	if(g->failed) return fail(LUW_ERR_STATE, "luw_group_run: an earlier run of this group failed half-way; destroy it");
	const int rc = group_run_steps(g, steps, first_sample, stride, mean_kernel_ms);
	if(rc!=LUW_OK&&g->dom.size()>1u&&g->initialized) {
		const std::string msg = g_last_error;
		for(GroupDomain& d : g->dom) { (void)hipSetDevice(d.device); (void)hipStreamSynchronize(d.comm); (void)hipStreamSynchronize(d.compute); }
		(void)hipGetLastError();
		g->failed = true;
		g_last_error = msg;
	}
	return rc;
}
static int group_run_steps(luw_group* g, const uint64_t steps, const uint64_t first_sample, const uint64_t stride, double* mean_kernel_ms) {
	if(!g->initialized) GROUP_TRY(luw_group_initialize(g));
	if(g->dom.size()==1u) { // undivided lattice: the single-domain path, no events
		luw_solver* s = g->dom[0].s;
		GROUP_TRY(luw_set_stream(s, nullptr));
		int e = LUW_OK;
		if(mean_kernel_ms) e = luw_run_timed(s, steps, mean_kernel_ms);
		else if(stride) e = luw_run_sampled(s, steps, first_sample, stride);
		else e = luw_run(s, steps);
		g->t = luw_get_t(s);
		return e;
	}
	std::vector<hipEvent_t> tev; // timing of domain 0's interior / whole-box launch
	struct TevFree { std::vector<hipEvent_t>& v; ~TevFree() { for(hipEvent_t e : v) if(e) (void)hipEventDestroy(e); } } tev_free{ tev };
	if(mean_kernel_ms) { GROUP_TRY(group_set_device(g->dom[0])); tev.assign(2u*steps, nullptr); for(auto& e : tev) HIP_TRY(hipEventCreate(&e)); }
	// Default: ONE enqueueing thread (measured with eight domains on one device: 44 us per domain and step, i.e. 0.36 ms per step for eight --
	// well inside a 1.9 ms FP16C step; profiles/r02_group_one_gpu.txt).  LUW_GROUP_THREADS=1 gives every domain its own host thread.
	const bool threaded = tuning().group_threads;
	// a call of a step or two (probe windows) is not worth starting threads for; RCCL's group calls are issued by ONE thread
	if(threaded&&steps>=4ull&&g->transport!=LUW_TRANSPORT_RCCL) {
		GroupThreads T; T.error.assign(g->dom.size(), std::string());
		std::vector<int> rc(g->dom.size(), LUW_OK);
		const uint64_t X0 = g->exchanges;
		auto work = [&](const size_t k) {
			rc[k] = domain_run_threaded(g, k, steps, first_sample, stride, (mean_kernel_ms&&k==0u) ? &tev : nullptr, X0, T);
			if(rc[k]!=LUW_OK) { T.error[k] = g_last_error; T.abort.store(1); } // g_last_error is per thread: carry the message over
		};
		std::vector<std::thread> th;
		for(size_t k=1; k<g->dom.size(); k++) th.emplace_back(work, k);
		work(0u);
		for(auto& x : th) x.join();
		for(size_t k=0; k<g->dom.size(); k++) if(rc[k]!=LUW_OK&&!T.error[k].empty()) return fail(rc[k], T.error[k]);
		for(size_t k=0; k<g->dom.size(); k++) if(rc[k]!=LUW_OK) return fail(rc[k], "luw_group_run: stopped because another domain failed");
		g->exchanges = X0+steps; g->t += steps;
		if(g->one_phase) g->xset = (g->xset+(uint32_t)(steps&1ull))&1u;
		// axes / passes that never ran keep in step
		for(GroupDomain& d : g->dom) for(int f=0; f<2; f++) for(int a=0; a<3; a++) { d.packed_seq[f][a] = g->exchanges; d.unpacked_seq[f][a] = g->exchanges; }
	} else {
		for(uint64_t i=0ull; i<steps; i++) {
			GroupStepPlan pl{};
			for(size_t k=0; k<g->dom.size(); k++) {
				GROUP_TRY(group_set_device(g->dom[k]));
				GROUP_TRY(domain_plan_step(g, k, i, steps, first_sample, stride, pl));
				GROUP_TRY(domain_launch_step(g, k, pl, (mean_kernel_ms&&k==0u) ? tev[2u*i] : nullptr, (mean_kernel_ms&&k==0u) ? tev[2u*i+1u] : nullptr,
					g->xset));
			}
			GROUP_TRY(group_communicate(g, !g->overlap));
			if(pl.separate) for(size_t k=0; k<g->dom.size(); k++) { GROUP_TRY(group_set_device(g->dom[k])); GROUP_TRY(domain_separate_stats(g, k)); }
			for(GroupDomain& d : g->dom) GROUP_TRY(luw_increment_time_step(d.s, 1ull));
			g->t++;
		}
		GROUP_TRY(group_join(g));
	}
	for(GroupDomain& d : g->dom) { d.stats_pending = false; if(steps>0ull) d.s->fields_current = true; }
	if(mean_kernel_ms) {
		GROUP_TRY(group_set_device(g->dom[0]));
		double sum = 0.0;
		for(uint64_t i=0ull; i<steps; i++) { float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, tev[2u*i], tev[2u*i+1u])); sum += (double)ms; }
		*mean_kernel_ms = steps ? sum/(double)steps : 0.0;
	}
	return LUW_OK;
}
