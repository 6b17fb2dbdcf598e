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
	if(!group_x_direct(g, i)) return LUW_OK;
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
	HIP_TRY(hipStreamWaitEvent(st, (thermal_pass ? P.gpacked : P.packed)[a], 0));
	HIP_TRY(hipStreamWaitEvent(st, (thermal_pass ? M.gpacked : M.packed)[a], 0));
	GROUP_TRY(luw_set_stream(d.s, st));
	GROUP_TRY(thermal_pass ? luw_enqueue_insert_gi(d.s, (uint32_t)a, d.grecv[a][0], d.grecv[a][1])
		: luw_enqueue_insert_fi(d.s, (uint32_t)a, d.recv[a][0], d.recv[a][1]));
	HIP_TRY(hipEventRecord((thermal_pass ? d.gunpacked : d.unpacked)[a], st));
	return LUW_OK;
}
// ---- LUW_TRANSPORT_RCCL: the faces as RCCL point-to-point messages, the reference's communicate_field (FX/lbm.cpp:1907-1935) with
// ncclSend / ncclRecv in place of its PCIe staging.  librccl is looked up at run time (dlopen) only when this transport is asked for, so
// the library's link dependencies stay the HIP runtime alone.  One communicator per distinct device (ncclCommInitAll); per axis ONE group
// call carries every domain's two sends and two receives, each on that domain's communication stream, so pack -> send / recv -> unpack
// are ordered by the streams themselves and no event crosses a device.
struct RcclApi {
	void* lib = nullptr;
	int (*CommInitAll)(void**, int, const int*) = nullptr;
	int (*CommDestroy)(void*) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi* rccl_api() {
	static RcclApi api;
	static const bool ok = [] {
		// an RCCL that is already in the process (torch brings its own) is the one to use: two copies would not share their topology state
		const char* names[] = { "librccl.so", "librccl.so.1" };
		for(const char* n : names) if(!api.lib) api.lib = dlopen(n, RTLD_NOW|RTLD_NOLOAD);
		for(const char* n : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) if(!api.lib) api.lib = dlopen(n, RTLD_NOW|RTLD_GLOBAL);
		if(!api.lib) return false;
		auto sym = [](const char* n) { return dlsym(api.lib, n); };
		api.CommInitAll = (int(*)(void**, int, const int*))sym("ncclCommInitAll");
		api.CommDestroy = (int(*)(void*))sym("ncclCommDestroy");
		api.GroupStart = (int(*)())sym("ncclGroupStart");
		api.GroupEnd = (int(*)())sym("ncclGroupEnd");
		api.Send = (int(*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
		api.Recv = (int(*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
		api.GetErrorString = (const char*(*)(int))sym("ncclGetErrorString");
		return api.CommInitAll&&api.CommDestroy&&api.GroupStart&&api.GroupEnd&&api.Send&&api.Recv&&api.GetErrorString;
	}();
	return ok ? &api : nullptr;
}
#define RCCL_TRY(call) do { const int r_ = (call); if(r_!=0) return fail(LUW_ERR_DEVICE, std::string(#call)+": "+rccl_api()->GetErrorString(r_)); } while(0)
static const int RCCL_UINT8 = 1; // ncclUint8: faces travel as bytes, nothing interprets them

static int group_rccl_setup(luw_group* g) {
	RcclApi* R = rccl_api();
	if(!R) return fail(LUW_ERR_DEVICE, "luw_group_create: LUW_GROUP_TRANSPORT=rccl but librccl could not be loaded");
	std::vector<int> devs; // distinct devices in order of first use
	g->rccl_rank.assign(g->dom.size(), 0);
	for(size_t i=0; i<g->dom.size(); i++) {
		const auto it = std::find(devs.begin(), devs.end(), g->dom[i].device);
		g->rccl_rank[i] = (int)(it-devs.begin());
		if(it==devs.end()) devs.push_back(g->dom[i].device);
	}
	std::vector<void*> comms(devs.size(), nullptr);
	if(g_injected_faults.load()&LUW_FAULT_RCCL_INIT) return fail(LUW_ERR_DEVICE, "ncclCommInitAll: injected failure (luw_dev_inject_fault)");
	RCCL_TRY(R->CommInitAll(comms.data(), (int)devs.size(), devs.data()));
	g->rccl_comm = comms;
	return LUW_OK;
}
static void group_rccl_teardown(luw_group* g) {
	RcclApi* R = rccl_api();
	if(R) for(void* c : g->rccl_comm) if(c) (void)R->CommDestroy(c);
	g->rccl_comm.clear();
}
// one axis: every domain packs into its send buffers; one grouped batch of sends and receives; every domain unpacks.
// A communicator works on ONE stream per batch: the communication stream of the first domain on its device (the "leader").  Domains that
// share a device with their leader (test set-ups; on a node every domain is its own leader and the waits below are on the stream itself)
// hand over with events: leader waits for their pack, they wait for the leader's batch before they unpack.
static int group_exchange_rccl_axis(luw_group* g, const int a, const bool thermal_pass, const bool on_compute) {
	RcclApi* R = rccl_api();
	const size_t n = g->dom.size();
	auto stream_of = [&](GroupDomain& d) { return on_compute ? d.compute : d.comm; };
	auto leader = [&](const size_t i) { size_t l = i; for(size_t k=0; k<i; k++) if(g->dom[k].device==g->dom[i].device) { l = k; break; } return l; };
	for(size_t i=0; i<n; i++) {
		GroupDomain& d = g->dom[i];
		GROUP_TRY(group_set_device(d));
		GROUP_TRY(luw_set_stream(d.s, stream_of(d)));
		void** out = (thermal_pass ? d.gsend : d.send)[a];
		GROUP_TRY(thermal_pass ? luw_enqueue_extract_gi(d.s, (uint32_t)a, out[0], out[1]) : luw_enqueue_extract_fi(d.s, (uint32_t)a, out[0], out[1]));
		HIP_TRY(hipEventRecord((thermal_pass ? d.gpacked : d.packed)[a], stream_of(d)));
	}
	for(size_t i=0; i<n; i++) if(leader(i)!=i) {
		GroupDomain& L = g->dom[leader(i)];
		GROUP_TRY(group_set_device(L));
		HIP_TRY(hipStreamWaitEvent(stream_of(L), (thermal_pass ? g->dom[i].gpacked : g->dom[i].packed)[a], 0));
	}
	// Message list in ONE global order -- (domain i, its + face), (domain i, its - face) for i = 0, 1, ... -- walked once for the sends
	// and once for the receives: RCCL pairs the k-th send of rank s to rank r with the k-th receive of r from s, and any two
	// messages between the same pair of ranks keep their relative order in both walks (also when + and - neighbour coincide,
	// and when several domains live on one device and talk to themselves).
	RCCL_TRY(R->GroupStart());
	for(size_t i=0; i<n; i++) for(int k=0; k<2; k++) {
		GroupDomain& d = g->dom[i];
		const uint32_t to = d.nbr[a][k];
		const size_t bytes = (size_t)luw_get_area(d.s, (uint32_t)a)*(thermal_pass ? 1u : 5u)*g->ddf_bytes;
		RCCL_TRY(R->Send((thermal_pass ? d.gsend : d.send)[a][k], bytes, RCCL_UINT8, g->rccl_rank[to], g->rccl_comm[g->rccl_rank[i]],
			stream_of(g->dom[leader(i)])));
	}
	for(size_t i=0; i<n; i++) for(int k=0; k<2; k++) {
		GroupDomain& d = g->dom[i];
		const uint32_t to = d.nbr[a][k];
		GroupDomain& dst = g->dom[to];
		// my + face (k = 0) is what the + neighbour receives "from its - side" (recv[a][1]); my - face lands in the - neighbour's recv[a][0]
		void* into = (thermal_pass ? dst.grecv : dst.recv)[a][1-k];
		const size_t bytes = (size_t)luw_get_area(d.s, (uint32_t)a)*(thermal_pass ? 1u : 5u)*g->ddf_bytes;
		RCCL_TRY(R->Recv(into, bytes, RCCL_UINT8, g->rccl_rank[i], g->rccl_comm[g->rccl_rank[to]], stream_of(g->dom[leader(to)])));
	}
	RCCL_TRY(R->GroupEnd());
	for(size_t i=0; i<n; i++) if(leader(i)==i) {
		GroupDomain& L = g->dom[i];
		GROUP_TRY(group_set_device(L));
		HIP_TRY(hipEventRecord((thermal_pass ? L.gunpacked : L.unpacked)[a], stream_of(L))); // "this device's batch is done"
	}
	for(size_t i=0; i<n; i++) {
		GroupDomain& d = g->dom[i];
		GROUP_TRY(group_set_device(d));
		if(leader(i)!=i) HIP_TRY(hipStreamWaitEvent(stream_of(d), (thermal_pass ? g->dom[leader(i)].gunpacked : g->dom[leader(i)].unpacked)[a], 0));
		GROUP_TRY(luw_set_stream(d.s, stream_of(d)));
		GROUP_TRY(thermal_pass ? luw_enqueue_insert_gi(d.s, (uint32_t)a, d.grecv[a][0], d.grecv[a][1])
			: luw_enqueue_insert_fi(d.s, (uint32_t)a, d.recv[a][0], d.recv[a][1]));
	}
	return LUW_OK;
}

// ---- the exchange in ONE phase (default where every pair of trading domains has peer access): what latticeurbanwind_amd/distributed.py does per rank over
// RCCL (_communicate_one_phase), with peer stores.  Per step and domain ONE pack round -- the y / z faces by their pack kernels and the twelve edge
// populations (k_edges: the population that crosses two cuts at once goes straight to the diagonal neighbour instead of riding in the rims of two
// consecutive face exchanges, FX/lbm.cpp:1908-1934), all written into the receivers' buffers; the x faces are there already, written by the step kernels --
// and ONE unpack round: the x faces are handed to the next step's kernels where they lie (luw_set_x_face_inputs: no unpack kernel), y / z faces inserted,
// the edges last (they overwrite what the rims of the faces carried).  Same populations in the same slots as the three-phase route
// (tests/test_gpu_group.py, tests/fuzz/fuzz_exchange_gpu.py hold both to each other and to the oracle).
static int domain_pack_all(luw_group* g, const size_t i, const bool on_compute, const uint32_t xs) {
	GroupDomain& d = g->dom[i];
	hipStream_t st = on_compute ? d.compute : d.comm;
	for(const uint32_t nb : d.nbrs) HIP_TRY(hipStreamWaitEvent(st, g->dom[nb].unpacked_all, 0)); // they have consumed what the previous step put there
	GROUP_TRY(luw_set_stream(d.s, st));
	for(int a=0; a<3; a++) {
		if(!g->H[a]) continue;
		GroupDomain& P = g->dom[d.nbr[a][0]]; GroupDomain& M = g->dom[d.nbr[a][1]];
		// (x: no launch where this step's kernels have written both faces -- into these very buffers, group_x_face_ready)
		if(a==0) GROUP_TRY(luw_enqueue_extract_fi(d.s, 0u, P.recvx[xs][1], M.recvx[xs][0]));
		else GROUP_TRY(luw_enqueue_extract_fi(d.s, (uint32_t)a, P.recv[a][1], M.recv[a][0]));
	}
	void* out[12];
	for(uint32_t e=0u; e<12u; e++) out[e] = luw_get_edge_length(d.s, e) ? g->dom[d.enbr[e]].erecv[e] : nullptr;
	GROUP_TRY(luw_enqueue_extract_edges(d.s, out));
	if(g->thermal) for(int a=0; a<3; a++) if(g->H[a])
		GROUP_TRY(luw_enqueue_extract_gi(d.s, (uint32_t)a, g->dom[d.nbr[a][0]].grecv[a][1], g->dom[d.nbr[a][1]].grecv[a][0]));
	HIP_TRY(hipEventRecord(d.packed_all, st));
	return LUW_OK;
}
// x_in_place: the x faces stay in their receive buffers for the next step's kernels (not at initialisation, whose exchange is followed by a reset of t)
static int domain_unpack_all(luw_group* g, const size_t i, const bool on_compute, const bool x_in_place, const uint32_t xs) {
	GroupDomain& d = g->dom[i];
	hipStream_t st = on_compute ? d.compute : d.comm;
	for(const uint32_t nb : d.nbrs) HIP_TRY(hipStreamWaitEvent(st, g->dom[nb].packed_all, 0));
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

// host mirrors <-> one global array in the reference layout (n = x + (y + z*Ny)*Nx over the GLOBAL lattice, components SoA)
struct GroupField { int comps; size_t elem; };
static bool group_field_info(const int field, GroupField& f) {
	switch(field) {
		case LUW_FIELD_RHO: case LUW_FIELD_T: f = GroupField{ 1, 4u }; return true;
		case LUW_FIELD_U: case LUW_FIELD_F: f = GroupField{ 3, 4u }; return true;
		case LUW_FIELD_FLAGS: f = GroupField{ 1, 1u }; return true;
		default: return false;
	}
}
// rows of one domain handled by a pool of host threads (the copies are memory-bound: a few threads saturate a socket)
template<typename Fn> static void group_parallel_rows(const uint64_t rows, Fn fn) {
	const unsigned T = (unsigned)std::max<uint64_t>(1ull, std::min<uint64_t>(std::min<uint64_t>(16ull, std::thread::hardware_concurrency()), rows/64ull+1ull));
	if(T<=1u) { for(uint64_t r=0ull; r<rows; r++) fn(r); return; }
	std::vector<std::thread> th;
	for(unsigned t=0u; t<T; t++) th.emplace_back([=]() { for(uint64_t r=rows*t/T; r<rows*(t+1ull)/T; r++) fn(r); });
	for(auto& x : th) x.join();
}

extern "C" {

int luw_group_create(const luw_config* cfg, const int* devices, luw_group** out) {
	if(!cfg||!out) return fail(LUW_ERR_INVALID, "luw_group_create: null argument");
	*out = nullptr;
	if(cfg->struct_size!=sizeof(luw_config)) return fail(LUW_ERR_INVALID, "luw_group_create: luw_config size mismatch (ABI)");
	const uint32_t D[3] = { cfg->Dx, cfg->Dy, cfg->Dz }, gN[3] = { cfg->Nx, cfg->Ny, cfg->Nz };
	if(D[0]*D[1]*D[2]==0u) return fail(LUW_ERR_INVALID, "You specified 0 LBM grid domains."); // FX/lbm.cpp:1124
	if((uint64_t)gN[0]*gN[1]*gN[2]==0ull) return fail(LUW_ERR_INVALID, "Grid point number is 0."); // FX/lbm.cpp:1123
	for(int a=0; a<3; a++) if(gN[a]%D[a]!=0u)
		return fail(LUW_ERR_INVALID, "LBM grid is not equally divisible in domains (the caller shrinks it to a multiple first, FX/lbm.cpp:1058-1060)");
	const uint32_t n = D[0]*D[1]*D[2];
	int ndev = 0;
	HIP_TRY(hipGetDeviceCount(&ndev));
	// FX/lbm.cpp:961-979
	if(!devices&&(int)n+cfg->device>ndev&&n>1u)
		return fail(LUW_ERR_INVALID, "luw_group_create: fewer HIP devices than domains (pass an explicit device list to share devices)");
	std::unique_ptr<luw_group, void(*)(luw_group*)> g(new luw_group(), group_free);
	g->gcfg = *cfg;
	for(int a=0; a<3; a++) { g->D[a] = D[a]; g->gN[a] = gN[a]; g->H[a] = D[a]>1u ? 1u : 0u; }
	g->ddf_bytes = cfg->ddf_format==LUW_DDF_FP16C ? 2u : 4u;
	g->thermal = (cfg->options&LUW_OPT_TEMPERATURE)!=0u;
	g->dom.resize(n);
	for(uint32_t i=0u; i<n; i++) { // FX/lbm.cpp:1066-1073
		GroupDomain& d = g->dom[i];
		d.device = devices ? devices[i] : (n>1u ? cfg->device+(int)i : cfg->device);
		if(d.device<0||d.device>=ndev) return fail(LUW_ERR_INVALID, "luw_group_create: no such HIP device");
		d.coord[0] = (i%(D[0]*D[1]))%D[0]; d.coord[1] = (i%(D[0]*D[1]))/D[0]; d.coord[2] = i/(D[0]*D[1]);
		for(int a=0; a<3; a++) {
			d.lN[a] = gN[a]/D[a]+2u*g->H[a];
			d.O[a] = (int32_t)(d.coord[a]*(gN[a]/D[a]))-(int32_t)g->H[a];
			uint32_t c[3] = { d.coord[0], d.coord[1], d.coord[2] };
			c[a] = (d.coord[a]+1u)%D[a]; d.nbr[a][0] = c[0]+(c[1]+c[2]*D[1])*D[0];
			c[a] = (d.coord[a]+D[a]-1u)%D[a]; d.nbr[a][1] = c[0]+(c[1]+c[2]*D[1])*D[0];
		}
	}
	g->overlap = n>1u&&step_can_overlap(g->dom[0].lN, g->H);
	// peer access between the devices of neighbouring domains (xGMI): enables the direct remote stores of the pack kernels
	g->peer.assign(n, std::vector<char>(n, 0));
	for(uint32_t i=0u; i<n; i++) for(uint32_t j=0u; j<n; j++) {
		const int di = g->dom[i].device, dj = g->dom[j].device;
		if((g_injected_faults.load()&LUW_FAULT_NO_PEER_ODD_PAIRS)&&((i+j)&1u)) continue; // test hook: this pair has "no peer access" (staged copies)
		if(di==dj) { g->peer[i][j] = 1; continue; }
		int can = 0;
		if(hipDeviceCanAccessPeer(&can, di, dj)==hipSuccess&&can) {
			HIP_TRY(hipSetDevice(di));
			const hipError_t e = hipDeviceEnablePeerAccess(dj, 0u);
			if(e==hipSuccess||e==hipErrorPeerAccessAlreadyEnabled) g->peer[i][j] = 1;
			(void)hipGetLastError();
		}
	}
	// transport: peer stores where the devices allow them (default), LUW_GROUP_TRANSPORT=staged the copy path everywhere, LUW_GROUP_TRANSPORT=rccl grouped
	// ncclSend / ncclRecv (tuning table)
	if(tuning().group_transport_bad) return fail(LUW_ERR_INVALID, "luw_group_create: LUW_GROUP_TRANSPORT must be peer, staged or rccl");
	g->transport = tuning().group_transport;
	if(g->transport!=LUW_TRANSPORT_PEER) for(auto& row : g->peer) std::fill(row.begin(), row.end(), 0); // faces through send buffers
	if(g->transport==LUW_TRANSPORT_RCCL&&n>1u) GROUP_TRY(group_rccl_setup(g.get())); // connections before the lattices (cf. TorchDistTransport.warm_up)
	// the diagonal neighbours (edge e carries population 7 + e to the domain in direction c_(7+e), FX/kernel.cpp:890-893) and everybody a domain trades with
	static const int EC[12][3] = { { 1, 1, 0 }, { -1, -1, 0 }, { 1, 0, 1 }, { -1, 0, -1 }, { 0, 1, 1 }, { 0, -1, -1 }, { 1, -1, 0 }, { -1, 1, 0 }, { 1, 0, -1 },
		{ -1, 0, 1 }, { 0, 1, -1 }, { 0, -1, 1 } };
	auto edge_exists = [&](const int e) { for(int a=0; a<3; a++) if(EC[e][a]!=0&&!g->H[a]) return false; return true; };
	for(uint32_t i=0u; i<n; i++) {
		GroupDomain& d = g->dom[i];
		auto at = [&](const int sx, const int sy, const int sz) {
			const uint32_t c[3] = { (d.coord[0]+D[0]+(uint32_t)sx)%D[0], (d.coord[1]+D[1]+(uint32_t)sy)%D[1], (d.coord[2]+D[2]+(uint32_t)sz)%D[2] };
			return c[0]+(c[1]+c[2]*D[1])*D[0];
		};
		std::vector<uint32_t> all;
		for(int a=0; a<3; a++) if(g->H[a]) { all.push_back(d.nbr[a][0]); all.push_back(d.nbr[a][1]); }
		for(int e=0; e<12; e++) if(edge_exists(e)) {
			d.enbr[e] = at(EC[e][0], EC[e][1], EC[e][2]);
			all.push_back(d.enbr[e]); all.push_back(at(-EC[e][0], -EC[e][1], -EC[e][2]));
		}
		std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end());
		d.nbrs = all;
	}
	// ONE pack / unpack round per step where the faces travel as peer stores between every pair of trading domains (LUW_GROUP_EXCHANGE=sequential: the
	// reference's three phases, which the staged and RCCL transports always take)
	g->one_phase = n>1u && g->transport==LUW_TRANSPORT_PEER && !tuning().group_sequential;
	for(uint32_t i=0u; i<n&&g->one_phase; i++) for(const uint32_t j : g->dom[i].nbrs) if(!g->peer[i][j]) { g->one_phase = false; break; }
	// Streams and halo buffers BEFORE the lattices: every long-lived small allocation is in place before the large arrays and the
	// placement search of luw_create run
	for(uint32_t i=0u; i<n; i++) {
		GroupDomain& d = g->dom[i];
		HIP_TRY(hipSetDevice(d.device));
		int lo = 0, hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi); // hi = numerically lowest = highest priority
		HIP_TRY(hipStreamCreateWithFlags(&d.compute, hipStreamNonBlocking));
		// shell, pack / unpack and copies on a high-priority queue: they are not to be stuck behind the interior kernel's workgroups
		if(hipStreamCreateWithPriority(&d.comm, hipStreamNonBlocking, hi)!=hipSuccess) {
			(void)hipGetLastError();
			HIP_TRY(hipStreamCreateWithFlags(&d.comm, hipStreamNonBlocking));
		}
		for(hipEvent_t* e : { &d.shell_done, &d.interior_done, &d.pre_done, &d.stats_done, &d.packed_all, &d.unpacked_all })
			HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
		if(g->one_phase) for(int e=0; e<12; e++) if(edge_exists(e)) { // one element per cell of the axis the edge runs along
			const size_t L = (size_t)d.lN[EC[e][0]==0 ? 0 : EC[e][1]==0 ? 1 : 2];
			if(hipMalloc(&d.erecv[e], L*g->ddf_bytes)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_group_create: halo buffers");
			HIP_TRY(hipMemset(d.erecv[e], 0, L*g->ddf_bytes));
		}
		for(int a=0; a<3; a++) {
			for(hipEvent_t* e : { &d.packed[a], &d.unpacked[a], &d.gpacked[a], &d.gunpacked[a] }) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
			if(!g->H[a]) continue;
			const size_t A = (size_t)d.lN[(a+1)%3]*d.lN[(a+2)%3];
			const bool staged = !(g->peer[i][d.nbr[a][0]]&&g->peer[i][d.nbr[a][1]]);
			for(int k=0; k<2; k++) {
				if(hipMalloc(&d.recv[a][k], 5u*A*g->ddf_bytes)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_group_create: halo buffers");
				HIP_TRY(hipMemset(d.recv[a][k], 0, 5u*A*g->ddf_bytes));
				if(staged&&hipMalloc(&d.send[a][k], 5u*A*g->ddf_bytes)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_group_create: halo buffers");
				if(a==0) {
					d.recvx[0][k] = d.recv[0][k];
					if(g->one_phase) {
						if(hipMalloc(&d.recvx[1][k], 5u*A*g->ddf_bytes)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_group_create: halo buffers");
						HIP_TRY(hipMemset(d.recvx[1][k], 0, 5u*A*g->ddf_bytes));
					}
				}
				if(g->thermal) {
					if(hipMalloc(&d.grecv[a][k], A*g->ddf_bytes)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_group_create: halo buffers");
					HIP_TRY(hipMemset(d.grecv[a][k], 0, A*g->ddf_bytes));
					if(staged&&hipMalloc(&d.gsend[a][k], A*g->ddf_bytes)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_group_create: halo buffers");
				}
			}
		}
	}
	for(uint32_t i=0u; i<n; i++) {
		GroupDomain& d = g->dom[i];
		luw_config c = *cfg;
		c.Nx = d.lN[0]; c.Ny = d.lN[1]; c.Nz = d.lN[2];
		c.Ox = d.O[0]; c.Oy = d.O[1]; c.Oz = d.O[2];
		c.device = d.device;
		// a device that hosts several domains of this group (test set-ups: eight domains on one GPU) is shared: no placement search there -- the first domain
		// would search alone, keep the memory the later ones need and be timed against nothing they run beside
		size_t on_device = 0u;
		for(const GroupDomain& o : g->dom) if(o.device==d.device) on_device++;
		g_device_is_shared = on_device>1u;
		const int rc = luw_create(&c, &d.s);
		g_device_is_shared = false;
		GROUP_TRY(rc);
		group_boxes(g.get(), d);
		if(g->H[0]) { // the step kernels write the x faces themselves (luw_set_x_face_buffers): into the neighbours' receive buffers, or into the send buffers
			void* fp = group_x_direct(g.get(), i) ? g->dom[d.nbr[0][0]].recv[0][1] : d.send[0][0];
			void* fm = group_x_direct(g.get(), i) ? g->dom[d.nbr[0][1]].recv[0][0] : d.send[0][1];
			GROUP_TRY(luw_set_x_face_buffers(d.s, fp, fm));
		}
	}
	*out = g.release();
	return LUW_OK;
}
void luw_group_destroy(luw_group* g) { group_free(g); }
uint32_t luw_group_size(const luw_group* g) { return g ? (uint32_t)g->dom.size() : 0u; }
luw_solver* luw_group_domain(luw_group* g, uint32_t d) { return (g&&d<g->dom.size()) ? g->dom[d].s : nullptr; }
uint64_t luw_group_get_t(const luw_group* g) { return g ? g->t : 0ull; }
int luw_group_overlaps(const luw_group* g) { return (g&&g->overlap) ? 1 : 0; }
int luw_group_one_phase(const luw_group* g) { return (g&&g->one_phase) ? 1 : 0; }
int luw_group_transport(const luw_group* g) {
	if(!g) return -1;
	if(g->transport==LUW_TRANSPORT_PEER&&!luw_group_direct_peer_stores(g)) return LUW_TRANSPORT_STAGED; // some pair of devices has no peer access
	return g->transport;
}
int luw_group_direct_peer_stores(const luw_group* g) {
	if(!g) return 0;
	for(size_t i=0; i<g->dom.size(); i++) for(int a=0; a<3; a++) if(g->H[a]&&!(g->peer[i][g->dom[i].nbr[a][0]]&&g->peer[i][g->dom[i].nbr[a][1]])) return 0;
	return 1;
}
int luw_group_domain_info(const luw_group* g, uint32_t d, uint32_t* local_N, int32_t* offset, int* device) {
	if(!g||d>=g->dom.size()) return fail(LUW_ERR_INVALID, "luw_group_domain_info: bad argument");
	for(int a=0; a<3; a++) { if(local_N) local_N[a] = g->dom[d].lN[a]; if(offset) offset[a] = g->dom[d].O[a]; }
	if(device) *device = g->dom[d].device;
	return LUW_OK;
}

int luw_group_set_f(luw_group* g, float fx, float fy, float fz) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_set_f: null group");
	for(GroupDomain& d : g->dom) GROUP_TRY(luw_set_f(d.s, fx, fy, fz));
	return LUW_OK;
}
int luw_group_set_coriolis(luw_group* g, float ox, float oy, float oz) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_set_coriolis: null group");
	for(GroupDomain& d : g->dom) GROUP_TRY(luw_set_coriolis(d.s, ox, oy, oz));
	return LUW_OK;
}

// Memory_Container's global index space over the domains' host mirrors (FX/lbm.hpp:274-297): scatter fills every domain's
// mirror INCLUDING its halo layers (periodic wrap, what communicate_rho_u_flags leaves there at initialisation,
// FX/lbm.cpp:1243-1256); gather reads the owned cells back.
int luw_group_scatter(luw_group* g, int field, const void* global_src) {
	GroupField f;
	if(!g||!global_src||!group_field_info(field, f)) return fail(LUW_ERR_INVALID, "luw_group_scatter: bad argument");
	const uint64_t GN = (uint64_t)g->gN[0]*g->gN[1]*g->gN[2];
	for(GroupDomain& d : g->dom) {
		char* dst = (char*)luw_host_ptr(d.s, field);
		if(!dst) return fail(LUW_ERR_STATE, "luw_group_scatter: the solver has no such field");
		const uint64_t LN = (uint64_t)d.lN[0]*d.lN[1]*d.lN[2];
		const uint32_t gNx = g->gN[0], gNy = g->gN[1], gNz = g->gN[2];
		for(int c=0; c<f.comps; c++) {
			const char* src = (const char*)global_src+(size_t)c*GN*f.elem;
			char* out = dst+(size_t)c*LN*f.elem;
			const GroupDomain* dp = &d; const size_t elem = f.elem;
			group_parallel_rows((uint64_t)d.lN[1]*d.lN[2], [=](const uint64_t r) {
				const uint32_t y = (uint32_t)(r%dp->lN[1]), z = (uint32_t)(r/dp->lN[1]);
				const uint32_t gy = (uint32_t)(((int64_t)y+dp->O[1]+(int64_t)gNy)%gNy), gz = (uint32_t)(((int64_t)z+dp->O[2]+(int64_t)gNz)%gNz);
				const char* srow = src+((size_t)gy+(size_t)gz*gNy)*gNx*elem;
				char* drow = out+(size_t)r*dp->lN[0]*elem;
				// the local row is the global row from gx0 on, wrapping at the lattice edge: at most a few contiguous runs
				for(uint32_t x=0u; x<dp->lN[0]; ) {
					const uint32_t gx = (uint32_t)(((int64_t)x+dp->O[0]+(int64_t)gNx)%gNx);
					const uint32_t run = std::min(dp->lN[0]-x, gNx-gx);
					memcpy(drow+(size_t)x*elem, srow+(size_t)gx*elem, (size_t)run*elem);
					x += run;
				}
			});
		}
	}
	return LUW_OK;
}
static int group_gather_from(luw_group* g, const GroupField f, void* global_dst, const std::function<const char*(GroupDomain&)>& source) {
	const uint64_t GN = (uint64_t)g->gN[0]*g->gN[1]*g->gN[2];
	for(GroupDomain& d : g->dom) {
		const char* srcb = source(d);
		if(!srcb) return fail(LUW_ERR_STATE, "luw_group_gather: the solver has no such field");
		const uint64_t LN = (uint64_t)d.lN[0]*d.lN[1]*d.lN[2];
		const uint32_t gNx = g->gN[0], gNy = g->gN[1];
		const uint32_t H0 = g->H[0], H1 = g->H[1], H2 = g->H[2];
		const uint32_t ox = d.lN[0]-2u*H0, oy = d.lN[1]-2u*H1, oz = d.lN[2]-2u*H2; // owned extents
		for(int c=0; c<f.comps; c++) {
			const char* src = srcb+(size_t)c*LN*f.elem;
			char* out = (char*)global_dst+(size_t)c*GN*f.elem;
			const GroupDomain* dp = &d; const size_t elem = f.elem;
			group_parallel_rows((uint64_t)oy*oz, [=](const uint64_t r) {
				const uint32_t y = (uint32_t)(r%oy)+H1, z = (uint32_t)(r/oy)+H2;
				const uint32_t gy = (uint32_t)((int32_t)y+dp->O[1]), gz = (uint32_t)((int32_t)z+dp->O[2]), gx0 = (uint32_t)((int32_t)H0+dp->O[0]);
				memcpy(out+(((size_t)gy+(size_t)gz*gNy)*gNx+gx0)*elem, src+(((size_t)y+(size_t)z*dp->lN[1])*dp->lN[0]+H0)*elem, (size_t)ox*elem);
			});
		}
	}
	return LUW_OK;
}
int luw_group_gather(luw_group* g, int field, void* global_dst) {
	GroupField f;
	if(!g||!global_dst||!group_field_info(field, f)) return fail(LUW_ERR_INVALID, "luw_group_gather: bad argument");
	return group_gather_from(g, f, global_dst, [field](GroupDomain& d) { return (const char*)luw_host_ptr(d.s, field); });
}
int luw_group_upload(luw_group* g, uint32_t mask) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_upload: null group");
	for(GroupDomain& d : g->dom) { GROUP_TRY(luw_set_stream(d.s, nullptr)); GROUP_TRY(luw_upload(d.s, mask)); }
	return LUW_OK;
}
int luw_group_download(luw_group* g, uint32_t mask) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_download: null group");
	for(GroupDomain& d : g->dom) { GROUP_TRY(luw_set_stream(d.s, nullptr)); GROUP_TRY(luw_download(d.s, mask)); }
	return LUW_OK;
}

int luw_group_initialize(luw_group* g) { // LBM::initialize, FX/lbm.cpp:1221-1260
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_initialize: null group");
	for(GroupDomain& d : g->dom) { GROUP_TRY(luw_set_stream(d.s, nullptr)); GROUP_TRY(luw_initialize(d.s)); }
	if(g->dom.size()>1u) {
		// "the communicate calls at initialization need an odd time step", FX/lbm.cpp:1242
		for(GroupDomain& d : g->dom) GROUP_TRY(luw_increment_time_step(d.s, 1ull));
		GROUP_TRY(group_communicate(g, false, false));                // (one-phase: the x faces go into the lattice here, t is reset behind this exchange)
		GROUP_TRY(group_join(g));
		for(GroupDomain& d : g->dom) GROUP_TRY(luw_reset_time_step(d.s)); // FX/lbm.cpp:1258
	}
	g->t = 0ull; g->initialized = true;
	return LUW_OK;
}
int luw_group_run(luw_group* g, uint64_t steps) { return group_run(g, steps, 0ull, 0ull, nullptr); }
int luw_group_run_sampled(luw_group* g, uint64_t steps, uint64_t first_sample, uint64_t stride) {
	if(first_sample==0ull||stride==0ull) return fail(LUW_ERR_INVALID, "luw_group_run_sampled: first_sample and stride count from 1");
	return group_run(g, steps, first_sample, stride, nullptr);
}
int luw_group_run_timed(luw_group* g, uint64_t steps, double* mean_kernel_ms) {
	if(!mean_kernel_ms) return fail(LUW_ERR_INVALID, "luw_group_run_timed: null output");
	return group_run(g, steps, 0ull, 0ull, mean_kernel_ms);
}

int luw_group_voxelize_mesh(luw_group* g, uint32_t triangle_number, const float* p0, const float* p1, const float* p2, const float* bounds, uint8_t flag) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_voxelize_mesh: null group");
	// every domain voxelises its own box (halos included) against the triangles binned to its tiles: FX/lbm.cpp:1455-1587
	for(GroupDomain& d : g->dom) { GROUP_TRY(luw_set_stream(d.s, nullptr)); GROUP_TRY(luw_voxelize_mesh(d.s, triangle_number, p0, p1, p2, bounds, flag)); }
	return LUW_OK;
}

// global cell index -> (owning domain, local reference-layout index); cells are owned by exactly one domain (halos belong to the neighbour)
static void group_locate(const luw_group* g, const uint64_t n, uint32_t& dom, uint64_t& local) {
	const uint64_t A = (uint64_t)g->gN[0]*g->gN[1];
	const uint32_t z = (uint32_t)(n/A), y = (uint32_t)((n%A)/g->gN[0]), x = (uint32_t)(n%g->gN[0]);
	const uint32_t b[3] = { g->gN[0]/g->D[0], g->gN[1]/g->D[1], g->gN[2]/g->D[2] };
	const uint32_t c[3] = { x/b[0], y/b[1], z/b[2] };
	dom = c[0]+(c[1]+c[2]*g->D[1])*g->D[0];
	const GroupDomain& d = g->dom[dom];
	const uint32_t lx = (uint32_t)((int32_t)x-d.O[0]), ly = (uint32_t)((int32_t)y-d.O[1]), lz = (uint32_t)((int32_t)z-d.O[2]);
	local = (uint64_t)lx+((uint64_t)ly+(uint64_t)lz*d.lN[1])*d.lN[0];
}
int luw_group_vk_inlet_attach(luw_group* g, uint64_t point_count, uint64_t mode_count, const uint64_t* point_cell, const uint8_t* point_face,
	const float* point_data, const float* mode_data, int update_stride, int stride_interpolation) {
	if(!g||!point_cell||!point_face||!point_data||!mode_data) return fail(LUW_ERR_INVALID, "luw_group_vk_inlet_attach: null argument");
	const uint64_t GN = (uint64_t)g->gN[0]*g->gN[1]*g->gN[2];
	const size_t n = g->dom.size();
	std::vector<std::vector<uint64_t>> cell(n), src(n);
	for(uint64_t i=0ull; i<point_count; i++) {
		if(point_cell[i]>=GN) return fail(LUW_ERR_INVALID, "luw_group_vk_inlet_attach: point cell outside the lattice");
		uint32_t dm; uint64_t local; group_locate(g, point_cell[i], dm, local);
		cell[dm].push_back(local); src[dm].push_back(i);
	}
	for(size_t k=0; k<n; k++) { // each domain gets the points it owns; the mode table is shared
		GroupDomain& d = g->dom[k];
		if(cell[k].empty()) { GROUP_TRY(luw_vk_inlet_detach(d.s)); continue; }
		const size_t P = cell[k].size();
		std::vector<uint8_t> face(P); std::vector<float> data(7u*P);
		for(size_t i=0; i<P; i++) { face[i] = point_face[src[k][i]]; for(int c=0; c<7; c++) data[(size_t)c*P+i] = point_data[(size_t)c*point_count+src[k][i]]; }
		GROUP_TRY(luw_vk_inlet_attach(d.s, P, mode_count, cell[k].data(), face.data(), data.data(), mode_data, update_stride, stride_interpolation));
	}
	return LUW_OK;
}

int luw_group_gather_attach(luw_group* g, uint32_t count, const uint64_t* cells) {
	if(!g||(count>0u&&!cells)) return fail(LUW_ERR_INVALID, "luw_group_gather_attach: bad argument");
	const uint64_t GN = (uint64_t)g->gN[0]*g->gN[1]*g->gN[2];
	std::vector<std::vector<uint64_t>> local(g->dom.size());
	for(GroupDomain& d : g->dom) d.gather_src.clear();
	for(uint32_t i=0u; i<count; i++) {
		if(cells[i]>=GN) return fail(LUW_ERR_INVALID, "luw_group_gather_attach: cell index outside the lattice");
		uint32_t dm; uint64_t l; group_locate(g, cells[i], dm, l);
		local[dm].push_back(l); g->dom[dm].gather_src.push_back(i);
	}
	for(size_t k=0; k<g->dom.size(); k++) if(int e = luw_gather_attach(g->dom[k].s, (uint32_t)local[k].size(), local[k].data())) {
		for(GroupDomain& d : g->dom) { d.gather_src.clear(); (void)luw_gather_attach(d.s, 0u, nullptr); } // all or nothing
		g->gather_total = 0u;
		return e;
	}
	g->gather_total = count;
	return LUW_OK;
}
int luw_group_gather_u(luw_group* g, float* out) {
	if(!g||!out) return fail(LUW_ERR_INVALID, "luw_group_gather_u: bad argument");
	std::vector<float> tmp;
	for(GroupDomain& d : g->dom) {
		if(d.gather_src.empty()) continue;
		tmp.resize(3u*d.gather_src.size());
		GROUP_TRY(luw_set_stream(d.s, nullptr));
		GROUP_TRY(luw_gather_u(d.s, tmp.data()));
		for(size_t i=0; i<d.gather_src.size(); i++) for(int c=0; c<3; c++) out[3u*d.gather_src[i]+c] = tmp[3u*i+c];
	}
	return LUW_OK;
}

uint64_t luw_group_stats_count(const luw_group* g) { return (g&&!g->dom.empty()) ? g->dom[0].s->avg_count : 0ull; }
int luw_group_stats_reset(luw_group* g) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_stats_reset: null group");
	for(GroupDomain& d : g->dom) { GROUP_TRY(luw_set_stream(d.s, nullptr)); GROUP_TRY(luw_stats_reset(d.s)); }
	return LUW_OK;
}
// global arrays in the layout write_avg_vtk consumes: avg_u AoS [3n+c], the others [n]; any pointer may be NULL
int luw_group_stats_download(luw_group* g, float* avg_u, float* avg_rho, float* m2_u, float* m2_v, float* m2_w, float* avg_T, uint64_t* count) {
	if(!g) return fail(LUW_ERR_INVALID, "luw_group_stats_download: null group");
	const uint32_t gNx = g->gN[0], gNy = g->gN[1];
	uint64_t LNmax = 0ull;
	for(const GroupDomain& d : g->dom) LNmax = std::max<uint64_t>(LNmax, (uint64_t)d.lN[0]*d.lN[1]*d.lN[2]);
	std::unique_ptr<float[]> buf(new float[8ull*LNmax]); // avg_u (3, AoS), avg_rho, m2 x3, avg_T of ONE domain at a time: bounded by a block, faulted in once
	for(GroupDomain& d : g->dom) {
		GROUP_TRY(luw_set_stream(d.s, nullptr));
		const uint64_t LN = (uint64_t)d.lN[0]*d.lN[1]*d.lN[2];
		float* lu = buf.get(); float* lr = lu+3ull*LN; float* l2[3] = { lr+LN, lr+2ull*LN, lr+3ull*LN }; float* lT = lr+4ull*LN;
		uint64_t cnt = 0ull;
		GROUP_TRY(luw_stats_download(d.s, avg_u ? lu : nullptr, avg_rho ? lr : nullptr, m2_u ? l2[0] : nullptr, m2_v ? l2[1] : nullptr, m2_w ? l2[2] : nullptr,
			&cnt));
		if(avg_T) GROUP_TRY(luw_stats_download_T(d.s, lT));
		if(count) *count = cnt;
		const uint32_t H0 = g->H[0], H1 = g->H[1], H2 = g->H[2];
		const uint32_t ox = d.lN[0]-2u*H0, oy = d.lN[1]-2u*H1, oz = d.lN[2]-2u*H2;
		const GroupDomain* dp = &d;
		float* outs[5] = { avg_rho, m2_u, m2_v, m2_w, avg_T }; const float* ins[5] = { lr, l2[0], l2[1], l2[2], lT };
		group_parallel_rows((uint64_t)oy*oz, [=](const uint64_t r) {
			const uint32_t y = (uint32_t)(r%oy)+H1, z = (uint32_t)(r/oy)+H2;
			const size_t grow = (((size_t)((int32_t)y+dp->O[1]))+(size_t)((int32_t)z+dp->O[2])*gNy)*gNx+(size_t)((int32_t)H0+dp->O[0]);
			const size_t lrow = ((size_t)y+(size_t)z*dp->lN[1])*dp->lN[0]+H0;
			if(avg_u) memcpy(avg_u+3u*grow, lu+3u*lrow, (size_t)ox*12u);
			for(int k=0; k<5; k++) if(outs[k]) memcpy(outs[k]+grow, ins[k]+lrow, (size_t)ox*4u);
		});
	}
	return LUW_OK;
}

} // extern "C"
