// luw_api.hpp -- C-ABI of one domain, part 1: library queries, life cycle (luw_create / luw_destroy), host mirrors, uploads and downloads.
// Included by luw_core.hip only, after luw_placement.hpp.
#pragma once

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
	// them (1024x1024x256: 6.61-6.75 ms either way).
	const uint64_t skew_blocks = cfg->ddf_format==LUW_DDF_FP16C ? 33ull : 513ull;
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
	// Workgroup order (xcd_row_order, luw_device.hpp): G = 4 consecutive lattice rows per XCD and turn for lattices with large DDF planes.  Measured
	// interleaved over lattice shapes and G = 0 / 1 / 2 / 4 / 16 (profiles/r05_xcd_rows_sweep.txt): FP32 1024^3, 2048x1024x512, 2048x2048x256 +3-6 %,
	// the 1024x1024x256 headline +1.0 % (four rounds of four), 512^3 and the 512^3 urban tile even -- so FP32 from 1 GiB planes on; FP16C gains only at
	// 1024^3 (2 GiB planes, +1.5 %) and is mixed below.  G = 1 and 2 lose on the smaller lattices (a single row per turn: -2.5 % at 512^3), 4 never did.
	const uint64_t plane_bytes = Np*(cfg->ddf_format==LUW_DDF_FP16C ? 2ull : 4ull);
	k.xcd_rows = tuning().xcd_rows>=0 ? (uint32_t)tuning().xcd_rows : plane_bytes>=((cfg->ddf_format==LUW_DDF_FP16C ? 2ull : 1ull)<<30) ? 4u : 0u;
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
	if(hipStreamSynchronize(s->stream)!=hipSuccess) { luw_destroy(s); return fail(LUW_ERR_DEVICE, "luw_create: the device did not finish the set-up copies"); }
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
	if(int e = xin_settle(s)) return e;
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
	xin_drop(s);   // a new lattice: nothing of an earlier exchange belongs to it
	if(int e = copy_pitched(s->d_fi, host_src, s->ddf_bytes, s, 19u, true, s->stream)) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

} // extern "C"
