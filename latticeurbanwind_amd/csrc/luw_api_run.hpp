// luw_api_run.hpp -- C-ABI of one domain, part 3: run-time setters, initialisation, box launches, the inlet update ahead of a step, luw_run and its
// sampled / timed forms, halo pack / unpack entry points, the entry points of include/luw_core_dev.h.  Included by luw_core.hip only, after luw_api_aux.hpp.
#pragma once

extern "C" {

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
	xin_drop(s);
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
	for(uint32_t side=0u; side<2u; side++) if((s->xin_buf&(1u<<side))&&s->t+steps!=s->xin_for_t[side]) {
		// a pending x face: only the step to the time it is for keeps it pending
		if(int e = set_device(s)) return e;
		if(int e = xin_settle(s, 1u<<side)) return e;
		s->xin_buf &= ~(1u<<side); s->xin_inplace &= ~(1u<<side);
	}
	s->t += steps;
	return LUW_OK;
}

int luw_reset_time_step(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_reset_time_step: null solver");
	if(s->xin_buf) {
		if(int e = set_device(s)) return e;
		if(int e = xin_settle(s)) return e;
		xin_drop(s);
	}
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
	// the step kernels of this step have written both x faces into these very buffers already (the record is of step xf_t: it lapses with the step)
	if(direction==0u&&buf_p==s->xf_p&&buf_m==s->xf_m&&s->xf_t==s->t&&s->xf_cover==3u) return LUW_OK;
	if(int e = xin_settle(s)) return e;   // a pack kernel reads the lattice (the rims of the y / z faces: the x halo columns)
	launch_transfer<false, false>(s, direction, buf_p, buf_m);
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
int luw_set_x_face_inputs(luw_solver* s, const void* buf_p, const void* buf_m) {
	if(!s||(!buf_p&&!buf_m)) return fail(LUW_ERR_INVALID, "luw_set_x_face_inputs: bad argument");
	if(!s->kp.halo_x) return fail(LUW_ERR_STATE, "luw_set_x_face_inputs: x is not split on this domain");
	if(int e = set_device(s)) return e;
	const uint32_t sides = (buf_p ? 1u : 0u)|(buf_m ? 2u : 0u);
	if(int e = xin_settle(s, sides)) return e;      // an earlier buffer of that side that nothing has taken yet
	if(buf_p) { s->xin_p = buf_p; s->xin_odd[0] = (uint32_t)(s->t&1ull); s->xin_for_t[0] = s->t+1ull; }
	if(buf_m) { s->xin_m = buf_m; s->xin_odd[1] = (uint32_t)(s->t&1ull); s->xin_for_t[1] = s->t+1ull; }
	s->xin_buf |= sides; s->xin_inplace &= ~sides;
	return LUW_OK;
}
int luw_enqueue_insert_fi(luw_solver* s, uint32_t direction, const void* buf_p, const void* buf_m) {
	if(!s||direction>2u||!buf_p||!buf_m) return fail(LUW_ERR_INVALID, "luw_enqueue_insert_fi: bad argument");
	if(int e = set_device(s)) return e;
	if(direction==0u) xin_drop(s);                                      // these values replace whatever was pending
	launch_transfer<false, true>(s, direction, const_cast<void*>(buf_p), const_cast<void*>(buf_m));
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
// edge e (population 7 + e) exists on this domain when both axes of its pair are split; its buffer holds one element per cell of the third axis
uint64_t luw_get_edge_length(const luw_solver* s, uint32_t edge) {
	if(!s||edge>=12u) return 0ull;
	const uint32_t pair = (edge%6u)/2u, H[3] = { s->kp.halo_x, s->kp.halo_y, s->kp.halo_z }, N[3] = { s->cfg.Nx, s->cfg.Ny, s->cfg.Nz };
	const uint32_t a = pair==2u ? 1u : 0u, b = pair==0u ? 1u : 2u;
	return (H[a]&&H[b]) ? (uint64_t)N[3u-a-b] : 0ull;
}
static int launch_edges(luw_solver* s, void* const* bufs, const bool insert, const char* who) {
	if(!s||!bufs) return fail(LUW_ERR_INVALID, std::string(who)+": bad argument");
	if(int e = set_device(s)) return e;
	EdgeBufs B{}; uint32_t Lmax = 0u;
	for(uint32_t e=0u; e<12u; e++) {
		const uint64_t L = luw_get_edge_length(s, e);
		B.p[e] = L ? bufs[e] : nullptr;             // (a null entry: that edge is not moved by this call)
		if(B.p[e]) Lmax = std::max(Lmax, (uint32_t)L);
	}
	if(!Lmax) return LUW_OK;
	const dim3 grid((Lmax+255u)/256u, 12u), block(256);
	const uint32_t odd = (uint32_t)(s->t&1ull);
	// x faces pending in their receive buffers: the edges across the x cut join them there (whoever takes the faces -- a launch in place, or the insert
	// kernel -- takes the edges with them)
	const uint32_t to_faces = insert ? (s->xin_buf&~s->xin_inplace) : 0u;
	void* const ip = (to_faces&1u) ? const_cast<void*>(s->xin_p) : nullptr; void* const im = (to_faces&2u) ? const_cast<void*>(s->xin_m) : nullptr;
	schedule_jitter(s->stream);
	if(s->ddf_bytes==2u) {
		if(insert) hipLaunchKernelGGL((k_edges<uint16_t, true>), grid, block, 0, s->stream, s->kp, odd, B, (uint16_t*)s->d_fi, (uint16_t*)ip, (uint16_t*)im);
		else hipLaunchKernelGGL((k_edges<uint16_t, false>), grid, block, 0, s->stream, s->kp, odd, B, (uint16_t*)s->d_fi, (uint16_t*)nullptr,
			(uint16_t*)nullptr);
	} else {
		if(insert) hipLaunchKernelGGL((k_edges<float, true>), grid, block, 0, s->stream, s->kp, odd, B, (float*)s->d_fi, (float*)ip, (float*)im);
		else hipLaunchKernelGGL((k_edges<float, false>), grid, block, 0, s->stream, s->kp, odd, B, (float*)s->d_fi, (float*)nullptr, (float*)nullptr);
	}
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
int luw_enqueue_extract_edges(luw_solver* s, void* const* dev_buffers) { return launch_edges(s, dev_buffers, false, "luw_enqueue_extract_edges"); }
int luw_enqueue_insert_edges(luw_solver* s, const void* const* dev_buffers) {
	return launch_edges(s, const_cast<void* const*>(reinterpret_cast<const void* const*>(dev_buffers)), true, "luw_enqueue_insert_edges");
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
int luw_dev_reload_tuning(void) { (void)tuning(); tuning_load(); return LUW_OK; } // (not while another thread is inside the library)
int luw_dev_inject_fault(uint32_t mask) { g_injected_faults.store(mask); return LUW_OK; }
int luw_dev_schedule_jitter(uint64_t seed, uint32_t max_us) { g_jitter_state.store(seed); g_jitter_max_us.store(std::min(max_us, 5000u)); return LUW_OK; }
int luw_dev_tuning_text(char* text, uint64_t size) {
	if(!text||size<64u) return fail(LUW_ERR_INVALID, "luw_dev_tuning_text: needs a buffer");
	const Tuning& t = tuning();
	const std::string alloc = t.alloc_chunk==0u ? "malloc" : t.alloc_chunk==~(size_t)0u ? "vmm:one" : "vmm:"+std::to_string(t.alloc_chunk>>20);
	snprintf(text, (size_t)size,
		"LUW_ALLOC=%s LUW_TEST_AIDS=%s%s%s%s LUW_TUNE_PLACEMENT=%d LUW_TUNE_FAST=%g LUW_X_SHELL=%u LUW_GROUP_TRANSPORT=%s LUW_GROUP_THREADS=%d "
		"LUW_GROUP_EXCHANGE=%s LUW_GROUP_OVERLAP=%d LUW_XCD_ROWS=%d LUW_SCHEDULE_JITTER=%llu:%u",
		alloc.c_str(), t.addr_row ? "addr_row," : "", t.fuse_stats ? "" : "separate_stats,", t.vk_ahead ? "" : "vk_inline,",
		t.voxelize_all ? "voxelize_all," : "",
		t.placement_candidates, t.placement_bar, t.x_shell,
		t.group_transport==LUW_TRANSPORT_RCCL ? "rccl" : t.group_transport==LUW_TRANSPORT_STAGED ? "staged" : "peer", (int)t.group_threads,
		t.group_sequential ? "sequential" : t.group_x_packed ? "one_packed" : "one-round", (int)t.group_overlap, t.xcd_rows,
		(unsigned long long)t.jitter_seed, t.jitter_us);
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
int luw_dev_workgroup_order(const luw_solver* s) { return s ? (int)s->kp.xcd_rows : -1; }

} // extern "C"
