// luw_api_aux.hpp -- C-ABI of one domain, part 2: device voxeliser, probe gather, von-Karman inlet tables, on-device statistics, device self-checks.
// Included by luw_core.hip only, after luw_api.hpp.
#pragma once

extern "C" {

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
	const bool ahead = tuning().vk_ahead; // LUW_TEST_AIDS=vk_inline: evaluate in line before every step
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

} // extern "C"
