// luw_group_api.hpp -- the C-ABI of the multi-domain host (luw_group_* in include/luw_core.h): create / destroy, Memory_Container's global index space
// over the domains' mirrors (scatter / gather), initialise, the run calls, and what is split over / collected from the owning domains (voxelisation,
// inlet points, probe cells, statistics).  Included by luw_core.hip only, behind luw_group.hpp (the exchange and the step loop).
#pragma once

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
	g->overlap = n>1u&&step_can_overlap(g->dom[0].lN, g->H)&&tuning().group_overlap;
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
	// ONE pack / unpack round per step: as peer stores where every pair of trading domains has peer access, through the send buffers and ONE batch of copies /
	// ncclSend / ncclRecv with the staged and RCCL transports (LUW_GROUP_EXCHANGE=sequential: the reference's three phases; also what a peer transport takes
	// on a node where SOME pair of trading devices has no peer access)
	g->one_phase = n>1u && !tuning().group_sequential;
	if(g->transport==LUW_TRANSPORT_PEER) for(uint32_t i=0u; i<n&&g->one_phase; i++) for(const uint32_t j : g->dom[i].nbrs)
		if(!g->peer[i][j]) { g->one_phase = false; break; }
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
			if(g->transport!=LUW_TRANSPORT_PEER) {
				if(hipMalloc(&d.esend[e], L*g->ddf_bytes)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_group_create: halo buffers");
				HIP_TRY(hipMemset(d.esend[e], 0, L*g->ddf_bytes));
			}
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
		// (LUW_GROUP_EXCHANGE=one_packed: not at all -- the pack kernel fetches the x faces and the insert kernel puts them: what first contact times the
		// kernels' scattered 2-4-byte remote stores against, tools/first_contact_defaults.py)
		// otherwise the step kernels write the x faces themselves (luw_set_x_face_buffers): into the neighbours' receive buffers, or into the send buffers
		if(g->H[0]&&!tuning().group_x_packed) {
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
