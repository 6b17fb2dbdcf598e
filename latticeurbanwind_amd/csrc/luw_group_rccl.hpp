// luw_group_rccl.hpp -- the RCCL transport of the multi-domain host (LUW_GROUP_TRANSPORT=rccl): librccl looked up at run time, one communicator per
// distinct device, the faces of one axis as ONE grouped batch of ncclSend / ncclRecv.  Included by luw_group.hpp (after GroupDomain / luw_group).
#pragma once

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
	auto leader = [&](const size_t i) { return group_leader(g, i); };
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
// The one-round exchange (luw_group.hpp, group_exchange_one_phase) over RCCL: every domain has packed its faces, edges and thermal faces into its send
// buffers (domain_pack_all; the x faces were written there by the step kernels); ONE ncclGroupStart / End per step carries all of it -- what
// latticeurbanwind_amd/distributed.py does per rank with one batch_isend_irecv.  Same populations into the same slots as the three phases
// (FX/lbm.cpp:1907-1935), the twelve edge messages in place of the rims.  Leaders as in group_exchange_rccl_axis: the batch of a device runs on the stream
// of the first domain on it; behind the batch that stream records the leader's packed[0] ("this device's batch is done": the per-axis events are free in
// this mode), which domain_unpack_all of every domain on the device waits for.
static int group_exchange_rccl_all(luw_group* g, const bool on_compute, const uint32_t xs) {
	RcclApi* R = rccl_api();
	const size_t n = g->dom.size();
	auto stream_of = [&](GroupDomain& d) { return on_compute ? d.compute : d.comm; };
	for(size_t i=0; i<n; i++) {
		const size_t l = group_leader(g, i);
		GroupDomain& L = g->dom[l];
		GROUP_TRY(group_set_device(L));
		// the leader's batch reads this domain's send buffers (packed_all) and writes its receive buffers (consumed by its last unpack round)
		if(l!=i) { HIP_TRY(hipStreamWaitEvent(stream_of(L), g->dom[i].packed_all, 0)); HIP_TRY(hipStreamWaitEvent(stream_of(L), g->dom[i].unpacked_all, 0)); }
	}
	// ONE global order of messages, walked once for the sends and once for the receives: RCCL pairs the k-th send of rank s to rank r with the k-th receive
	// of r from s, and any two messages between the same pair of ranks keep their relative order in both walks
	std::vector<GroupMsg> msgs;
	for(size_t i=0; i<n; i++) domain_messages(g, i, xs, msgs);
	RCCL_TRY(R->GroupStart());
	for(const GroupMsg& m : msgs)
		RCCL_TRY(R->Send(m.from, m.bytes, RCCL_UINT8, g->rccl_rank[m.dst], g->rccl_comm[g->rccl_rank[m.src]], stream_of(g->dom[group_leader(g, m.src)])));
	for(const GroupMsg& m : msgs)
		RCCL_TRY(R->Recv(m.into, m.bytes, RCCL_UINT8, g->rccl_rank[m.src], g->rccl_comm[g->rccl_rank[m.dst]], stream_of(g->dom[group_leader(g, m.dst)])));
	RCCL_TRY(R->GroupEnd());
	for(size_t i=0; i<n; i++) if(group_leader(g, i)==i) {
		GroupDomain& L = g->dom[i];
		GROUP_TRY(group_set_device(L));
		HIP_TRY(hipEventRecord(L.packed[0], stream_of(L)));
	}
	return LUW_OK;
}

