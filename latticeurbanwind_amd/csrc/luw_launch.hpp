// luw_launch.hpp -- which kernel instantiation a box launch takes: force modes of a box, x-face output, the tables of every stream_collide variant the
// library carries and their launchers, the halo pack / unpack launcher.  Included by luw_core.hip only, after luw_memory.hpp.
#pragma once

#ifdef LUW_AB_KERNELS
template<typename T, int V> static void launch_vec(luw_solver* s, const Box& b, const int write_fields) {
	T* fi = (T*)s->d_fi;
	const bool odd = (s->t&1ull)!=0ull;
	const uint32_t nvec = (b.x1-1u)/V-b.x0/V+1u;          // vectors overlapping [x0,x1)
	uint32_t vx = 1u; while(vx<nvec&&vx<256u) vx <<= 1;   // power of two
	const uint32_t ry = 256u/vx;
	const uint32_t nchunk = (nvec+vx-1u)/vx;
	const uint32_t rows = (b.y1-b.y0)*(b.z1-b.z0);
	const dim3 grid(((rows+ry-1u)/ry)*nchunk), block(vx, ry);
	if(odd)
		hipLaunchKernelGGL((k_stream_collide_v<T, V, 1>), grid, block, 0, s->stream, s->kp, b, nchunk, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
	else hipLaunchKernelGGL((k_stream_collide_v<T, V, 0>), grid, block, 0, s->stream, s->kp, b, nchunk, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
}
#endif
// threads per block for a row of nx lanes: whole waves, at most 256, chosen so that the blocks of a row carry the fewest idle
// lanes (a 375-lane row of the pair kernel: 3 x 128 instead of 2 x 256; ties go to the larger block)
static uint32_t row_block(const uint32_t nx) {
	if(nx<=256u) return ((nx+63u)/64u)*64u;
	uint32_t best = 256u, best_lanes = ((nx+255u)/256u)*256u;
	for(uint32_t bx : {192u, 128u, 64u}) { const uint32_t lanes = ((nx+bx-1u)/bx)*bx; if(lanes<best_lanes) { best = bx; best_lanes = lanes; } }
	return best;
}
// Where the position-dependent forces of this domain act, as cell ranges per face (the host's copy of in_force_zone, luw_device.hpp): buffer
// nudging within Nbuf cells of the lateral faces the domain owns (not the downstream one) and of the top, the sponge in the sponge_N layers
// under the top.  lo[a] / hi[a]: the first cell behind the zone at the low face of axis a / the first cell of the zone at its high face
// (0 / N when there is none): cells of [lo, hi) on all three axes are outside every zone.
static void force_free_core(const luw_solver* s, uint32_t lo[3], uint32_t hi[3]) {
	const KParams& k = s->kp;
	const int64_t N[3] = { (int64_t)k.Nx, (int64_t)k.Ny, (int64_t)k.Nz };
	int64_t l[3] = { 0, 0, 0 }, h[3] = { N[0], N[1], N[2] };
	if(k.buffer_active) {
		const int64_t nb = (int64_t)k.buffer_N;
		if(k.downstream_face!=1u&&k.has_w) l[0] = std::max<int64_t>(l[0], nb-k.Ox+1);
		if(k.downstream_face!=2u&&k.has_e) h[0] = std::min<int64_t>(h[0], (int64_t)k.Nxg-1-nb-k.Ox);
		if(k.downstream_face!=3u&&k.has_s) l[1] = std::max<int64_t>(l[1], nb-k.Oy+1);
		if(k.downstream_face!=4u&&k.has_n) h[1] = std::min<int64_t>(h[1], (int64_t)k.Nyg-1-nb-k.Oy);
		if(k.has_t) h[2] = std::min<int64_t>(h[2], (int64_t)k.Nzg-1-nb-k.Oz);
	}
	if(k.sponge_active&&k.has_t) h[2] = std::min<int64_t>(h[2], (int64_t)k.Nzg-1-(int64_t)k.sponge_N-k.Oz);
	for(int a=0; a<3; a++) {
		lo[a] = (uint32_t)std::min<int64_t>(std::max<int64_t>(l[a], 0), N[a]);
		hi[a] = (uint32_t)std::min<int64_t>(std::max<int64_t>(h[a], (int64_t)lo[a]), N[a]);
	}
}
// The nudging / sponge zones as cell ranges of this domain (KParams zw_lo ... zp_n): the conditions of FX/kernel.cpp:1537-1541,1598 -- the term is on,
// the domain owns the face, it is not the downstream one, 0 <= distance <= Nbuf (sponge: 0 <= layer < Nsponge) -- solved for the local coordinate
// and clipped to the domain.  n = 0: no cell.
static void set_zone_ranges(KParams& k) {
	auto range = [](const bool on, const int64_t lo, const int64_t hi, const int64_t N, uint32_t& zlo, uint32_t& zn) {
		const int64_t a = std::max<int64_t>(lo, 0), b = std::min<int64_t>(hi, N-1);
		if(on&&b>=a) { zlo = (uint32_t)a; zn = (uint32_t)(b-a+1); } else { zlo = 0u; zn = 0u; }
	};
	const int64_t nb = (int64_t)k.buffer_N;
	const bool buf = k.buffer_active!=0u;
	range(buf&&k.downstream_face!=1u&&k.has_w, k.west_x, k.west_x+nb, k.Nx, k.zw_lo, k.zw_n);
	range(buf&&k.downstream_face!=2u&&k.has_e, k.east_x-nb, k.east_x, k.Nx, k.ze_lo, k.ze_n);
	range(buf&&k.downstream_face!=3u&&k.has_s, k.south_y, k.south_y+nb, k.Ny, k.zs_lo, k.zs_n);
	range(buf&&k.downstream_face!=4u&&k.has_n, k.north_y-nb, k.north_y, k.Ny, k.zn_lo, k.zn_n);
	range(buf&&k.has_t, k.top_z-nb, k.top_z, k.Nz, k.zt_lo, k.zt_n);
	range(k.sponge_active&&k.has_t, (int64_t)k.top_z-(int64_t)k.sponge_N, (int64_t)k.top_z-1, k.Nz, k.zp_lo, k.zp_n);
}
// what can push the cells of box b (collide_cell_pk): re-evaluated per launch, so luw_set_f / luw_set_coriolis take effect at once.
// (Cutting a box that reaches into the zones along their boundaries -- specialised kernel on the zone-free core, general kernels on six slabs
// around it -- was built and measured on the 512^3 urban tile with its 80-cell nudging zones and 100-layer sponge: 2.36-2.38 ms in one
// launch, 2.41-2.46 ms cut; the core is a third of the cells there and the slabs cost more than it gains.  One launch per box it stays.)
static int box_force_mode(const luw_solver* s, const Box& b) {
	const KParams& k = s->kp;
	if(k.has_F) return PAIR_FORCE_ANY;
	uint32_t lo[3], hi[3];
	force_free_core(s, lo, hi);
	if(b.x0<lo[0]||b.x1>hi[0]||b.y0<lo[1]||b.y1>hi[1]||b.z0<lo[2]||b.z1>hi[2]) return PAIR_FORCE_ANY;
	return (k.coriolis||k.fx!=0.0f||k.fy!=0.0f||k.fz!=0.0f) ? PAIR_FORCE_UNIFORM : PAIR_FORCE_NONE;
}
// ---- x-face output of the step kernels (luw_set_x_face_buffers).  A launch takes the instantiation with the output when the buffers are set, x is split
// and its box holds the first or the last owned x column; it "covers" a column when the box also spans every non-halo (y, z) of it.  When the launches of
// a step have covered both columns, luw_enqueue_extract_fi(direction 0) on the same buffers has nothing left to do.
static bool xface_wanted(const luw_solver* s, const Box& b) {
	if(!s->xf_p||!s->xf_m||!s->kp.halo_x||s->cfg.Nx<4u) return false;
	return (b.x0<=1u&&b.x1>1u)||(b.x0<=s->cfg.Nx-2u&&b.x1>s->cfg.Nx-2u);
}
// (a column is covered when the (y, z) extents of the step's launches that hold it are disjoint and add up to every non-halo (y, z) -- one whole box, or
// the shell boxes and the interior of luw_step.hpp.  Extents that overlap, e.g. a box launched twice, or more than eight of them: no claim for this step,
// the pack kernel runs)
static void xface_covered(luw_solver* s, const Box& b) {
	if(s->xf_t!=s->t) {
		s->xf_t = s->t; s->xf_cover = 0u;
		for(int k=0; k<2; k++) { s->xf_area[k] = 0ull; s->xf_rects[k] = 0u; s->xf_lost[k] = false; }
	}
	const luw_solver::FaceRect r{ std::max(b.y0, s->kp.halo_y), std::min(b.y1, s->cfg.Ny-s->kp.halo_y), std::max(b.z0, s->kp.halo_z),
		std::min(b.z1, s->cfg.Nz-s->kp.halo_z) };
	if(r.y1<=r.y0||r.z1<=r.z0) return;
	const uint64_t full = (uint64_t)(s->cfg.Ny-2u*s->kp.halo_y)*(s->cfg.Nz-2u*s->kp.halo_z);
	auto add = [&](const int k, const uint32_t bit) { // k = 1: first owned column, the face towards -x; 0: last owned column, towards +x
		if(s->xf_lost[k]) return;
		for(uint32_t q=0u; q<s->xf_rects[k]; q++) {
			const luw_solver::FaceRect& o = s->xf_rect[k][q];
			if(r.y0<o.y1&&o.y0<r.y1&&r.z0<o.z1&&o.z0<r.z1) { s->xf_lost[k] = true; s->xf_cover &= ~bit; return; }
		}
		if(s->xf_rects[k]==8u) { s->xf_lost[k] = true; s->xf_cover &= ~bit; return; }
		s->xf_rect[k][s->xf_rects[k]++] = r;
		s->xf_area[k] += (uint64_t)(r.y1-r.y0)*(r.z1-r.z0);
		if(s->xf_area[k]>=full) s->xf_cover |= bit;
	};
	if(b.x0<=1u&&b.x1>1u) add(1, 2u);
	if(b.x0<=s->cfg.Nx-2u&&b.x1>s->cfg.Nx-2u) add(0, 1u);
}
// ---- x-face input (luw_set_x_face_inputs): the insert of the x faces is pending, the values wait in the receive buffers.  A launch of the step they are for
// that holds a border column reads that column's side there when its instantiation can (the ones with the x-face output); everything else that wants a side
// in the lattice -- another instantiation, a pack kernel, a download, a change of t other than the step to xin_for_t -- has xin_settle run the insert kernel
// for it first, with the time parity of the moment the buffers were handed over.  Once a launch has read a side in place the lattice cannot take it any more
// (the cells' slots hold their new values): the launches of one step on one border column must agree on whether they can read the buffers.
static void launch_insert_x(luw_solver* s, const void* buf_p, const void* buf_m, uint32_t odd);
static int xin_settle(luw_solver* s, const uint32_t sides = 3u) {
	const uint32_t todo = sides & s->xin_buf & ~s->xin_inplace;
	if(!todo) return LUW_OK;
	if(todo==3u&&s->xin_odd[0]==s->xin_odd[1]) launch_insert_x(s, s->xin_p, s->xin_m, s->xin_odd[0]);
	else {
		if(todo&1u) launch_insert_x(s, s->xin_p, nullptr, s->xin_odd[0]);
		if(todo&2u) launch_insert_x(s, nullptr, s->xin_m, s->xin_odd[1]);
	}
	HIP_TRY(hipGetLastError());
	s->xin_buf &= ~todo;
	return LUW_OK;
}
static void xin_drop(luw_solver* s) { s->xin_buf = 0u; s->xin_inplace = 0u; }
// before a stream_collide launch: may it read the pending sides of its border columns in place (sets s->xin_use for the instance launcher), or must they go
// into the lattice first?
static int xin_before_launch(luw_solver* s, const Box& b, const bool instantiation_reads_them) {
	s->xin_use = false;
	if(!s->xin_buf||!s->kp.halo_x||s->cfg.Nx<4u) return LUW_OK;
	const uint32_t need = (((b.x0<=1u&&b.x1>1u) ? 2u : 0u)|((b.x0<=s->cfg.Nx-2u&&b.x1>s->cfg.Nx-2u) ? 1u : 0u)) & s->xin_buf;
	if(!need) return LUW_OK;
	const bool for_this_step = (!(need&1u)||s->xin_for_t[0]==s->t) && (!(need&2u)||s->xin_for_t[1]==s->t);
	// (a side the box does not hold, or that is in the lattice already, gets a null pointer: the kernel tests each side's lanes against its own buffer)
	// (rows of fewer than six cells: a pair-kernel lane could hold the first AND the last owned column, and it fetches one side's values -- such a domain
	// takes its faces through the insert kernel)
	if(instantiation_reads_them&&for_this_step&&s->cfg.Nx>=6u) { s->xin_use = true; s->xin_inplace |= need; return LUW_OK; }
	if(need&s->xin_inplace) return fail(LUW_ERR_STATE, "stream_collide: an earlier launch of this step read this border column's x face in its receive buffer; "
		"this launch cannot");
	return xin_settle(s, need);
}
// ---------------------------------------------------------------- the kernel instantiations, as tables
// Every stream_collide variant the library carries is one row: what it is for (the key the launchers look up) and the function that launches its two
// time-parity instances.  Nothing else instantiates the step kernels.
struct LaunchGeom { dim3 grid, block; int xa; uint32_t lds; };

// ---- k_stream_collide_s: one cell per lane
// mode 0: step, 4: step + thermal lattice; 1, 2, 3: A/B variants (tools build)
struct ScalarKey { uint8_t ddf_bytes; int mode; int nt; bool flat, stats, noforce, native, xface; };
typedef void (*ScalarLaunch)(luw_solver*, const Box&, const LaunchGeom&, int write_fields, const StatsArgs&);
template<typename T, int MODE, int NT, bool FLAT, bool STATS, bool NOFORCE, bool NATIVE=false,
	bool XFACE=false> static void scalar_instance(luw_solver* s, const Box& b, const LaunchGeom& g, const int wf, const StatsArgs& S) {
	T* const fi = (T*)s->d_fi; T* const gi = MODE==4 ? (T*)s->d_gi : nullptr; float* const Tf = MODE==4 ? s->d_T : nullptr;
	T* const xp = XFACE ? (T*)s->xf_p : nullptr; T* const xm = XFACE ? (T*)s->xf_m : nullptr;
	const T* const ip = (XFACE&&s->xin_use&&(s->xin_buf&1u)) ? (const T*)s->xin_p : nullptr;
	const T* const im = (XFACE&&s->xin_use&&(s->xin_buf&2u)) ? (const T*)s->xin_m : nullptr;
	if(s->t&1ull) hipLaunchKernelGGL((k_stream_collide_s<T, 1, MODE, NT, FLAT, STATS, NOFORCE, NATIVE, XFACE>), g.grid, g.block, 0, s->stream, s->kp, b, g.xa,
		fi, s->d_rho, s->d_u, s->d_flags, s->d_F, wf, gi, Tf, S, xp, xm, ip, im);
	else hipLaunchKernelGGL((k_stream_collide_s<T, 0, MODE, NT, FLAT, STATS, NOFORCE, NATIVE, XFACE>), g.grid, g.block, 0, s->stream, s->kp, b, g.xa, fi,
		s->d_rho, s->d_u, s->d_flags, s->d_F, wf, gi, Tf, S, xp, xm, ip, im);
}
struct ScalarRow { ScalarKey key; ScalarLaunch launch; const char* what; };
static const ScalarRow scalar_table[] = {
	//  bytes mode nt flat   stats  noforce
	{ { 4u, 0, 2, true,  false, false }, scalar_instance<float, 0, 2, true, false, false>,
		"FP32 product kernel, flat addressing (planes within 32-bit byte offsets)" },
	{ { 4u, 0, 2, false, false, false }, scalar_instance<float, 0, 2, false, false, false>,      "FP32 product kernel, row addressing (any plane size)" },
	{ { 4u, 0, 2, true,  true,  false }, scalar_instance<float, 0, 2, true, true, false>,        "FP32, sampled step (fused Welford update)" },
	{ { 4u, 0, 2, false, true,  false }, scalar_instance<float, 0, 2, false, true, false>,       "FP32, sampled step, row addressing" },
	{ { 4u, 4, 2, true,  false, false }, scalar_instance<float, 4, 2, true, false, false>,       "FP32 + thermal lattice" },
	{ { 4u, 4, 2, false, false, false }, scalar_instance<float, 4, 2, false, false, false>,      "FP32 + thermal lattice, row addressing" },
	{ { 2u, 0, 2, false, false, false }, scalar_instance<uint16_t, 0, 2, false, false, false>,
		"FP16C one-cell kernel (rows too narrow / unaligned for the pair kernel)" },
	{ { 2u, 0, 2, false, false, true  }, scalar_instance<uint16_t, 0, 2, false, false, true>,    "FP16C one-cell kernel, force-free box: 7 waves per SIMD" },
	{ { 2u, 0, 2, false, true,  false }, scalar_instance<uint16_t, 0, 2, false, true, false>,    "FP16C one-cell kernel, sampled step" },
	{ { 2u, 4, 2, false, false, false }, scalar_instance<uint16_t, 4, 2, false, false, false>,   "FP16C one-cell kernel + thermal lattice" },
	{ { 2u, 4, 2, false, false, true  }, scalar_instance<uint16_t, 4, 2, false, false, true>,    "FP16C one-cell kernel + thermal lattice, force-free box" },
	// x-split domains, boxes that hold the first / last owned x column: the same kernels with the x-face output (luw_set_x_face_buffers)
	{ { 4u, 0, 2, true,  false, false, false, true }, scalar_instance<float, 0, 2, true, false, false, false, true>,  "FP32 + x-face output" },
	{ { 4u, 0, 2, false, false, false, false, true }, scalar_instance<float, 0, 2, false, false, false, false, true>, "FP32, row addressing + x-face output" },
	{ { 4u, 4, 2, true,  false, false, false, true }, scalar_instance<float, 4, 2, true, false, false, false, true>,
		"FP32 + thermal lattice + x-face output" },
	{ { 4u, 4, 2, false, false, false, false, true }, scalar_instance<float, 4, 2, false, false, false, false, true>,
		"FP32 + thermal, row addressing + x-face" },
	{ { 2u, 0, 2, false, false, false, true }, scalar_instance<uint16_t, 0, 2, false, false, false, true>, "FP16C one-cell kernel, native arithmetic" },
	{ { 2u, 4, 2, false, false, false, true }, scalar_instance<uint16_t, 4, 2, false, false, false, true>,
		"FP16C one-cell kernel + thermal lattice, native arithmetic" },
	{ { 2u, 0, 2, false, true,  false, true }, scalar_instance<uint16_t, 0, 2, false, true, false, true>,
		"FP16C one-cell kernel, sampled step, native arithmetic" },
#ifdef LUW_AB_KERNELS   // tools build: measurement-only and A/B variants
	{ { 4u, 1, 1, true,  false, false }, scalar_instance<float, 1, 1, true, false, false>,       "A/B: no collision" },
	{ { 4u, 1, 1, false, false, false }, scalar_instance<float, 1, 1, false, false, false>,      "A/B: no collision, row addressing" },
	{ { 4u, 2, 1, true,  false, false }, scalar_instance<float, 2, 1, true, false, false>,       "A/B: x+1 neighbours replaced by x" },
	{ { 4u, 2, 1, false, false, false }, scalar_instance<float, 2, 1, false, false, false>,      "A/B: no shift, row addressing" },
	{ { 4u, 0, 0, true,  false, false }, scalar_instance<float, 0, 0, true, false, false>,       "A/B: default cache policy" },
	{ { 4u, 0, 0, false, false, false }, scalar_instance<float, 0, 0, false, false, false>,      "A/B: default cache policy, row addressing" },
	{ { 4u, 0, 1, true,  false, false }, scalar_instance<float, 0, 1, true, false, false>,       "A/B: non-temporal on all planes" },
	{ { 4u, 0, 1, false, false, false }, scalar_instance<float, 0, 1, false, false, false>,      "A/B: non-temporal on all planes, row addressing" },
	{ { 4u, 3, 2, true,  false, false }, scalar_instance<float, 3, 2, true, false, false>,       "A/B: general path only" },
	{ { 4u, 3, 2, false, false, false }, scalar_instance<float, 3, 2, false, false, false>,      "A/B: general path only, row addressing" },
	{ { 2u, 1, 1, false, false, false }, scalar_instance<uint16_t, 1, 1, false, false, false>,   "A/B: FP16C no collision" },
	{ { 2u, 2, 1, false, false, false }, scalar_instance<uint16_t, 2, 1, false, false, false>,   "A/B: FP16C no shift" },
	{ { 2u, 0, 0, false, false, false }, scalar_instance<uint16_t, 0, 0, false, false, false>,   "A/B: FP16C default cache policy" },
	{ { 2u, 0, 1, false, false, false }, scalar_instance<uint16_t, 0, 1, false, false, false>,   "A/B: FP16C non-temporal on all planes" },
	{ { 2u, 3, 2, false, false, false }, scalar_instance<uint16_t, 3, 2, false, false, false>,   "A/B: FP16C general path only" },
#endif
};
static int launch_scalar(luw_solver* s, const Box& b, const int write_fields, const StatsArgs* st = nullptr) {
	LaunchGeom g{};
	g.xa = (int)b.x0-(int)((b.x0+64u-s->kp.halo_x)&63u); // block start of the line that holds b.x0 (see lead_alloc)
	const uint32_t nx = (uint32_t)((int)b.x1-g.xa), bx = row_block(nx);
	g.grid = dim3((nx+bx-1u)/bx, b.y1-b.y0, b.z1-b.z0); g.block = dim3(bx);
	ScalarKey k{ (uint8_t)s->ddf_bytes, 0, 2, false, st!=nullptr, false, false, false };
	// FLAT addressing (one 32-bit byte offset per neighbour within a plane) for FP32 lattices whose planes fit it; the row form otherwise.  In-plane
	// offsets span the lattice part Px*Ny*Nz of a plane only -- the skew behind it is stride, never addressed -- so 1024^3 with its 2^32-byte planes
	// still qualifies (largest offset 2^32 - 4).  LUW_TEST_AIDS=addr_row: the row form also where the flat form would do (both are product code, same values)
	const bool force_row = tuning().addr_row;
	k.flat = s->ddf_bytes==4u && (uint64_t)s->kp.Px*s->cfg.Ny*s->cfg.Nz*4ull<=(1ull<<32) && !force_row;
	// FP16C, nothing can push the cells of this box: the instantiation without the force assembly (69 / 76 VGPRs)
	k.noforce = s->ddf_bytes==2u && !st && box_force_mode(s, b)==PAIR_FORCE_NONE;
	if(s->d_gi&&!st) k.mode = 4; // thermal lattice on: the product kernel plus the D3Q7 cell update
	// native arithmetic (FP16C; plain and sampled steps -- a sampled step with the thermal lattice never comes here: can_fuse_stats): one instantiation per kind
	if(s->ddf_bytes==2u&&(s->cfg.options&LUW_OPT_NATIVE_ARITH)!=0u) { k.native = true; k.noforce = false; }
	// x-face output: FP32 plain steps on a box that holds a border column
	k.xface = s->ddf_bytes==4u && (k.mode==0||k.mode==4) && !st && xface_wanted(s, b);
#ifdef LUW_AB_KERNELS
	if(!st&&!s->d_gi) switch(s->kernel) {
		case LUW_KERNEL_EXP_COPY: k.mode = 1; k.nt = 1; k.noforce = false; break;
		case LUW_KERNEL_EXP_NOSHIFT: k.mode = 2; k.nt = 1; k.noforce = false; break;
		case LUW_KERNEL_SCALAR_CACHED: k.nt = 0; k.noforce = false; break;
		case LUW_KERNEL_SCALAR_NT_ALL: k.nt = 1; k.noforce = false; break;
		case LUW_KERNEL_SCALAR_GENERAL: k.mode = 3; k.noforce = false; break;
		default: break;
	}
#endif
	for(const ScalarRow& r : scalar_table) {
		const ScalarKey& q = r.key;
		if(q.ddf_bytes==k.ddf_bytes&&q.mode==k.mode&&q.nt==k.nt&&q.flat==k.flat&&q.stats==k.stats&&q.noforce==k.noforce&&q.native==k.native&&q.xface==k.xface) {
			if(int e = xin_before_launch(s, b, k.xface)) return e;
			r.launch(s, b, g, write_fields, st ? *st : StatsArgs{});
			if(k.xface) xface_covered(s, b);
			return LUW_OK;
		}
	}
	return fail(LUW_ERR_STATE, "stream_collide: this library carries no one-cell kernel for the requested combination");
}

// ---- k_stream_collide_p: FP16C, two cells per lane
struct PairKey { int mode; bool stats; int force; bool park, thermal, native, xface; };   // mode 1: memory path only (tools build)
typedef void (*PairLaunch)(luw_solver*, const Box&, const LaunchGeom&, int write_fields, const StatsArgs&);
template<int MODE, bool STATS, int FORCE, bool PARK, bool THERMAL, bool NATIVE=false,
	bool XFACE=false> static void pair_instance(luw_solver* s, const Box& b, const LaunchGeom& g, const int wf, const StatsArgs& S) {
	uint16_t* const fi = (uint16_t*)s->d_fi; uint16_t* const gi = THERMAL ? (uint16_t*)s->d_gi : nullptr; float* const Tf = THERMAL ? s->d_T : nullptr;
	uint16_t* const xp = XFACE ? (uint16_t*)s->xf_p : nullptr; uint16_t* const xm = XFACE ? (uint16_t*)s->xf_m : nullptr;
	const uint16_t* const ip = (XFACE&&s->xin_use&&(s->xin_buf&1u)) ? (const uint16_t*)s->xin_p : nullptr;
	const uint16_t* const im = (XFACE&&s->xin_use&&(s->xin_buf&2u)) ? (const uint16_t*)s->xin_m : nullptr;
	if(s->t&1ull) hipLaunchKernelGGL((k_stream_collide_p<1, MODE, STATS, FORCE, PARK, THERMAL, NATIVE, XFACE>), g.grid, g.block, g.lds, s->stream, s->kp, b,
		fi, s->d_rho, s->d_u, s->d_flags, s->d_F, wf, S, gi, Tf, xp, xm, ip, im);
	else hipLaunchKernelGGL((k_stream_collide_p<0, MODE, STATS, FORCE, PARK, THERMAL, NATIVE, XFACE>), g.grid, g.block, g.lds, s->stream, s->kp, b, fi,
		s->d_rho, s->d_u, s->d_flags, s->d_F, wf, S, gi, Tf, xp, xm, ip, im);
}
struct PairRow { PairKey key; PairLaunch launch; const char* what; };
static const PairRow pair_table[] = {
	//  mode stats  force               park   thermal
	{ { 0, false, PAIR_FORCE_NONE,    false, false }, pair_instance<0, false, PAIR_FORCE_NONE, false, false>,
		"nothing can push the cells of the box: no force path, 5 waves per SIMD" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false>, "volume force / Coriolis only, 5 waves" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false }, pair_instance<0, false, PAIR_FORCE_ANY, true, false>,
		"general (zones, force field): second cell's values parked in LDS, 5 waves" },
	{ { 0, false, PAIR_FORCE_ANY,     false, false }, pair_instance<0, false, PAIR_FORCE_ANY, false, false>,
		"general, everything in registers, 4 waves (LUW_PAIR_PARK=0: A/B and test aid)" },
	{ { 0, true,  PAIR_FORCE_ANY,     false, false }, pair_instance<0, true, PAIR_FORCE_ANY, false, false>,      "sampled step (fused Welford update)" },
	{ { 0, false, PAIR_FORCE_NONE,    true,  true  }, pair_instance<0, false, PAIR_FORCE_NONE, true, true>,      "+ thermal lattice, force-free box" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  true  }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, true>,   "+ thermal lattice, uniform forces" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  true  }, pair_instance<0, false, PAIR_FORCE_ANY, true, true>,       "+ thermal lattice, general" },
	// LUW_OPT_NATIVE_ARITH: the same six in the hardware's own arithmetic (collide_cell_pk_native); sampled steps keep the exact kernel
	{ { 0, false, PAIR_FORCE_NONE,    false, false, true }, pair_instance<0, false, PAIR_FORCE_NONE, false, false, true>,      "native: force-free box" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false, true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false, true>,   "native: uniform forces" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false, true }, pair_instance<0, false, PAIR_FORCE_ANY, true, false, true>,        "native: general" },
	{ { 0, false, PAIR_FORCE_NONE,    true,  true,  true }, pair_instance<0, false, PAIR_FORCE_NONE, true, true, true>,        "native + thermal, force-free" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  true,  true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, true, true>,     "native + thermal, uniform" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  true,  true }, pair_instance<0, false, PAIR_FORCE_ANY, true, true, true>,         "native + thermal, general" },
	{ { 0, true,  PAIR_FORCE_ANY,     false, false, true }, pair_instance<0, true, PAIR_FORCE_ANY, false, false, true>,        "native: sampled step" },
	// x-split domains, boxes that hold the first / last owned x column: x-face output (luw_set_x_face_buffers), exact and native
	{ { 0, false, PAIR_FORCE_NONE,    false, false, false, true }, pair_instance<0, false, PAIR_FORCE_NONE, false, false, false, true>,
		"force-free + x-face" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false, false, true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false, false, true>, "uniform + x-face" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false, false, true }, pair_instance<0, false, PAIR_FORCE_ANY, true, false, false, true>,      "general + x-face" },
	{ { 0, false, PAIR_FORCE_NONE,    false, false, true,  true }, pair_instance<0, false, PAIR_FORCE_NONE, false, false, true, true>,     "native + x-face" },
	{ { 0, false, PAIR_FORCE_UNIFORM, false, false, true,  true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, false, false, true, true>,  "native + x-face" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  false, true,  true }, pair_instance<0, false, PAIR_FORCE_ANY, true, false, true, true>,       "native + x-face" },
	// ... and with the thermal lattice (what the reference's shipped build runs): the D3Q19 x faces out of / into the kernels, the D3Q7 faces by their kernels
	{ { 0, false, PAIR_FORCE_NONE,    true,  true,  false, true }, pair_instance<0, false, PAIR_FORCE_NONE, true, true, false, true>,      "thermal + x-face" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  true,  false, true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, true, false, true>,   "thermal + x-face" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  true,  false, true }, pair_instance<0, false, PAIR_FORCE_ANY, true, true, false, true>,       "thermal + x-face" },
	{ { 0, false, PAIR_FORCE_NONE,    true,  true,  true,  true }, pair_instance<0, false, PAIR_FORCE_NONE, true, true, true, true>,
		"thermal native + x-face" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  true,  true,  true }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, true, true, true>,
		"thermal native + x-face" },
	{ { 0, false, PAIR_FORCE_ANY,     true,  true,  true,  true }, pair_instance<0, false, PAIR_FORCE_ANY, true, true, true, true>,
		"thermal native + x-face" },
#ifdef LUW_AB_KERNELS
	{ { 1, false, PAIR_FORCE_ANY,     false, false }, pair_instance<1, false, PAIR_FORCE_ANY, false, false>,
		"A/B: the kernel's memory path alone (LUW_PAIR_COPY)" },
	{ { 0, false, PAIR_FORCE_NONE,    true,  false }, pair_instance<0, false, PAIR_FORCE_NONE, true, false>,
		"A/B: force-free with PARK (7 waves: no gain, profiles/r03_pair_park_ab.txt)" },
	{ { 0, false, PAIR_FORCE_UNIFORM, true,  false }, pair_instance<0, false, PAIR_FORCE_UNIFORM, true, false>,
		"A/B: uniform forces with PARK (6 waves: slower)" },
#endif
};
static int launch_pair(luw_solver* s, const Box& b, const int write_fields, const StatsArgs* st = nullptr) {
	LaunchGeom g{};
	const uint32_t nx = (b.x1-b.x0+1u)/2u;                         // an odd count only when the box ends at an odd Nx: the last lane owns one cell
	const uint32_t bx = row_block(nx);
	g.grid = dim3((nx+bx-1u)/bx, b.y1-b.y0, b.z1-b.z0); g.block = dim3(bx);
	PairKey k{ 0, st!=nullptr, st ? PAIR_FORCE_ANY : box_force_mode(s, b), false, s->d_gi!=nullptr };
	// PARK (luw_kernels_pair.hpp): the lane's second set of values waits in LDS instead of in registers.  Measured interleaved on MI355X
	// (profiles/r03_pair_park_ab.txt): it pays where the registers cost a wave of occupancy that matters -- the general kernel, 109 -> 91 VGPRs,
	// 4 -> 5 waves per SIMD: urban 512^3 tile 2.344 -> 2.276 ms, + Coriolis 2.465 -> 2.375 -- and not above five waves (force-free 86 -> 68
	// VGPRs, 7 waves: 3.50 -> 3.49 ms; uniform forces 96 -> 78, 6 waves: 3.78 -> 3.91 ms on 1024x1024x256); the thermal variants always park.
	constexpr unsigned park_modes = 1u<<PAIR_FORCE_ANY;
	k.park = k.thermal || (!st && (park_modes&(1u<<k.force))!=0u);
	// native arithmetic: plain and sampled steps of the product kernel (a run is native from its first step to its last)
	k.native = (s->cfg.options&LUW_OPT_NATIVE_ARITH)!=0u;
	if(k.native&&!k.thermal&&!st) k.park = k.force==PAIR_FORCE_ANY;
	// x-face output: plain steps of the D3Q19 lattice with the product's park choice, on a box that holds a border column
	k.xface = !st && k.mode==0 && (k.thermal || k.park==(k.force==PAIR_FORCE_ANY)) && xface_wanted(s, b);
#ifdef LUW_AB_KERNELS
	const bool copy_only = tuning().ab_pair_copy;   // tools build, measurement aid: the kernel's memory path alone (no physics)
	if(copy_only&&!st&&!k.thermal) k = PairKey{ 1, false, PAIR_FORCE_ANY, false, false };
#endif
	g.lds = k.park ? (bx/64u)*pair_park_bytes_per_wave(k.thermal, (k.mode==0&&!k.stats) ? k.force : PAIR_FORCE_NONE) : 0u;
	for(const PairRow& r : pair_table) {
		const PairKey& q = r.key;
		if(q.mode==k.mode&&q.stats==k.stats&&q.force==k.force&&q.park==k.park&&q.thermal==k.thermal&&q.native==k.native&&q.xface==k.xface) {
			// (the uniform-force instantiation sits at its 96 VGPRs without a register for the x-face INPUT: it writes its faces, and has the unpack kernel
			// run for what it receives -- pair_reads_x_face_inputs, luw_kernels_pair.hpp)
			if(int e = xin_before_launch(s, b, k.xface&&pair_reads_x_face_inputs(k.force, k.thermal))) return e;
			r.launch(s, b, g, write_fields, st ? *st : StatsArgs{});
			if(k.xface) xface_covered(s, b);
			return LUW_OK;
		}
	}
	return fail(LUW_ERR_STATE, "stream_collide: this library carries no pair kernel for the requested combination");
}

// Kernel choice.  LUW_KERNEL_AUTO = the scalar kernel (FP32: 39.5k MLUPS at 512^3; vector kernels 20-29k) and, for FP16C rows
// wide enough, the pair kernel (profiles/r01_kernel_ab.md).  The other kernels stay selectable for A/B runs.
// can a sampled step carry the Welford update itself?  Product kernels only (scalar / pair, no thermal lattice: its T statistics
// stay with k_stats_accumulate); LUW_TEST_AIDS=separate_stats keeps the separate kernel
static bool can_fuse_stats(const luw_solver* s) {
	return tuning().fuse_stats && !s->d_gi && (s->kernel==LUW_KERNEL_AUTO||s->kernel==LUW_KERNEL_SCALAR||s->kernel==LUW_KERNEL_PAIR);
}
static int launch_stream_collide(luw_solver* s, const Box& b, const int write_fields, const StatsArgs* st = nullptr) {
	if(b.x0>=b.x1||b.y0>=b.y1||b.z0>=b.z1) return LUW_OK; // empty box
	if(b.x1>s->cfg.Nx||b.y1>s->cfg.Ny||b.z1>s->cfg.Nz) return fail(LUW_ERR_INVALID, "stream_collide: box exceeds the local lattice");
	if(b.y1-b.y0>65535u||b.z1-b.z0>65535u) return fail(LUW_ERR_INVALID, "stream_collide: box too large for the launch geometry");
	const bool fp16 = s->ddf_bytes==2u;
	uint32_t k = s->kernel;
	// AUTO: the scalar kernel, except FP16C rows of at least one wave of pairs, which take the pair kernel (dword accesses, packed
	// FP32 collision: 69.0k vs 67.2k MLUPS at 512^3, 63.1k vs 61.1k with Coriolis)
	constexpr uint32_t pair_min = 128u;   // (round 2: 256)
	if(k==LUW_KERNEL_AUTO) k = (fp16 && b.x1-b.x0>=pair_min) ? LUW_KERNEL_PAIR : LUW_KERNEL_SCALAR;
	if(s->d_gi&&(!fp16||st)) k = LUW_KERNEL_SCALAR; // FP32 / sampled steps: the thermal cell update of the one-cell kernel
#ifdef LUW_AB_KERNELS
	// the vector kernels assume rows that start on a 16-byte boundary at x = 0
	if(s->kp.halo_x&&(k==LUW_KERNEL_VEC4||k==LUW_KERNEL_VEC2||k==LUW_KERNEL_VEC1)) k = LUW_KERNEL_SCALAR;
#endif
	// pair kernel: FP16C; pairs start on a 4-byte boundary -- at even x, or at odd x when x is split (the row's lead pad then puts
	// x = 1 on a line start, lead_alloc); the range holds whole pairs, except that it may end at an odd Nx of an unsplit row (the
	// last cell then pairs with the row padding)
	if(k==LUW_KERNEL_PAIR) {
		const bool starts_aligned = ((b.x0+s->kp.halo_x)&1u)==0u;
		const bool whole_pairs = ((b.x1-b.x0)&1u)==0u || (!s->kp.halo_x && b.x1==s->cfg.Nx);
		if(!fp16||!starts_aligned||!whole_pairs) k = LUW_KERNEL_SCALAR;
	}
	if(st&&k!=LUW_KERNEL_PAIR&&k!=LUW_KERNEL_SCALAR) return fail(LUW_ERR_STATE, "stream_collide: this kernel has no fused statistics");
	schedule_jitter(s->stream);
	// the workgroup order of large lattices (KParams::xcd_rows, luw_create's rule) is a property of launches over ALL non-halo rows -- what it was measured
	// on; a shell or slab box of a decomposed step keeps the dispatch order
	struct OrderGuard { luw_solver* s; uint32_t G; ~OrderGuard() { s->kp.xcd_rows = G; } } order{ s, s->kp.xcd_rows };
	if(!(b.y0<=s->kp.halo_y&&b.y1>=s->cfg.Ny-s->kp.halo_y&&b.z0<=s->kp.halo_z&&b.z1>=s->cfg.Nz-s->kp.halo_z)) s->kp.xcd_rows = 0u;
	if(k==LUW_KERNEL_PAIR) { if(int e = launch_pair(s, b, write_fields, st)) return e; }
	else if(st) { if(int e = launch_scalar(s, b, write_fields, st)) return e; }
#ifdef LUW_AB_KERNELS
	else if(k==LUW_KERNEL_VEC4) { if(fp16) launch_vec<uint16_t, 4>(s, b, write_fields); else launch_vec<float, 4>(s, b, write_fields); }
	else if(k==LUW_KERNEL_VEC2) { if(fp16) launch_vec<uint16_t, 2>(s, b, write_fields); else launch_vec<float, 2>(s, b, write_fields); }
	else if(k==LUW_KERNEL_VEC1) { if(fp16) launch_vec<uint16_t, 1>(s, b, write_fields); else launch_vec<float, 1>(s, b, write_fields); }
#endif
	else { if(int e = launch_scalar(s, b, write_fields)) return e; }
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}

// launch helper: picks the template instance for (storage type, lattice, direction)
template<bool G, bool INSERT> static void launch_transfer(luw_solver* s, const uint32_t direction, void* buf_p, void* buf_m) {
	const uint32_t A = (uint32_t)luw_get_area(s, direction);
	const dim3 grid((A+255u)/256u), block(256);
	const uint32_t odd = (uint32_t)(s->t&1ull);
	void* lat = G ? s->d_gi : s->d_fi;
	schedule_jitter(s->stream);
	#define LUW_TR(TT, DD) do { \
		if constexpr(INSERT) hipLaunchKernelGGL((k_insert_fi<TT, G, DD>), grid, block, 0, s->stream, s->kp, A, odd, (const TT*)buf_p, (const TT*)buf_m, (TT*)lat); \
		else hipLaunchKernelGGL((k_extract_fi<TT, G, DD>), grid, block, 0, s->stream, s->kp, A, odd, (TT*)buf_p, (TT*)buf_m, (const TT*)lat); \
	} while(0)
	if(s->ddf_bytes==2u) { if(direction==0u) LUW_TR(uint16_t, 0); else if(direction==1u) LUW_TR(uint16_t, 1); else LUW_TR(uint16_t, 2); }
	else { if(direction==0u) LUW_TR(float, 0); else if(direction==1u) LUW_TR(float, 1); else LUW_TR(float, 2); }
	#undef LUW_TR
}
// the x-face insert with an explicit time parity (xin_settle: the parity of the moment the buffers were handed over)
static void launch_insert_x(luw_solver* s, const void* buf_p, const void* buf_m, const uint32_t odd) {
	const uint32_t A = (uint32_t)luw_get_area(s, 0u);
	const dim3 grid((A+255u)/256u), block(256);
	schedule_jitter(s->stream);
	if(s->ddf_bytes==2u) hipLaunchKernelGGL((k_insert_fi<uint16_t, false, 0>), grid, block, 0, s->stream, s->kp, A, odd, (const uint16_t*)buf_p,
		(const uint16_t*)buf_m, (uint16_t*)s->d_fi);
	else hipLaunchKernelGGL((k_insert_fi<float, false, 0>), grid, block, 0, s->stream, s->kp, A, odd, (const float*)buf_p, (const float*)buf_m,
		(float*)s->d_fi);
}
