// luw_device.hpp -- device-side building blocks of the D3Q19 collide-stream step for gfx950 (wave64).
//
// What is computed follows the reference kernel `stream_collide` (FX/kernel.cpp:1475-1780, with the
// device functions it calls: f_eq :1016-1055, moments :1075-1100, Guo forcing :1103-1113, FP16C codec
// :864-875).  HOW it is computed is ours: coordinates come from the launch geometry (no div/mod per cell),
// the sin^2 ramps of the nudging/sponge terms come from host-built tables, DDFs are addressed as per-plane
// base pointer + 32-bit cell offset, and the vector kernels move 4 cells per lane.
//
// Arithmetic contract (shared with the CPU oracle used by the tests): FP32, fmaf() exactly where the
// reference writes fma(), every other operation individually rounded (this translation unit is compiled with
// -ffp-contract=off), IEEE-correct division and square root (hipcc default).  Zero-valued terms of the
// c_i-weighted sums are dropped: adding (+-0) never changes a value, so results are value-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace luw {

constexpr float DEF_W0 = 1.0f/3.0f;   // FX/lbm.cpp:674-676
constexpr float DEF_WS = 1.0f/18.0f;
constexpr float DEF_WE = 1.0f/36.0f;
constexpr float DEF_C = 0.57735027f;  // FX/lbm.cpp:663

constexpr uint8_t TYPE_S = 0x01, TYPE_E = 0x02, TYPE_BO = 0x03, TYPE_T = 0x04, TYPE_SU = 0x38, TYPE_G = 0x20;

// Everything the kernels need besides the big arrays; passed by value (lands in SGPRs).
struct KParams {
	uint32_t Nx, Ny, Nz;      // local lattice
	uint32_t Px;              // x pitch of every device array (multiple of 64, >= Nx)
	uint32_t Np;              // plane stride = Px*Ny*Nz (< 2^32, checked on the host)
	uint32_t halo_x, halo_y, halo_z; // 1 if that axis is split over domains (cells 0 and N-1 are halo: FX/kernel.cpp:856-859)
	int32_t Ox, Oy, Oz;
	float w;                  // def_w
	float tau0, tau0sq;       // 1/w and its square as the kernel would compute them (IEEE division / product, done once on the host)
	float fx, fy, fz;
	float omx, omy, omz;
	uint32_t coriolis;        // any component of omega nonzero
	uint32_t subgrid;
	// BUFFER_NUDGING / TOP_SPONGE, FX/lbm.cpp:613-625,770-782
	uint32_t buffer_active, buffer_N, nudge_vertical, downstream_face;
	float buffer_inv_tau;
	uint32_t sponge_active, sponge_N;
	uint32_t Nxg, Nyg, Nzg;   // global extents
	int32_t west_x, east_x, south_y, north_y, top_z;   // local coordinates of the global outer faces
	uint32_t has_w, has_e, has_s, has_n, has_t;
	const float* wbuf;        // wbuf[d] = sin^2(pi/2 (1 - d/Nbuf)),           d = 0..Nbuf   (FX/kernel.cpp:1581-1583)
	const float* sigma;       // sigma[d] = inv_tau sin^2(pi/2 (1 - d/(Ns-1))), d = 0..Ns-1  (FX/kernel.cpp:1604-1606)
	// The same zones as cell ranges of THIS domain, filled by the host (luw_core.hip, set_zone_ranges): a coordinate c lies in a zone when
	// (c - lo) < n as unsigned numbers; n = 0 where the zone does not exist here (term off, face not owned, downstream face).  One subtraction and
	// one compare per face instead of the four conditions of FX/kernel.cpp:1537-1541 each; same cells.
	uint32_t zw_lo, zw_n, ze_lo, ze_n;   // buffer nudging next to the west / east face (x)
	uint32_t zs_lo, zs_n, zn_lo, zn_n;   // south / north face (y)
	uint32_t zt_lo, zt_n;                // top (z)
	uint32_t zp_lo, zp_n;                // top sponge layers (z)
	uint32_t has_F;
	float w_T;                // TEMPERATURE: def_w_T = 1/(2 alpha + 1/2), FX/lbm.cpp:750 (0 when the thermal lattice is off)
	// native-arithmetic kernels (LUW_OPT_NATIVE_ARITH, collide_cell_pk_native): host-folded constants
	float half_tau0;          // tau0 / 2
	float m2omx, m2omy, m2omz; // -2 omega
	// workgroup order of the step kernels: 0 = as dispatched (consecutive workgroups go round robin to the 8 XCDs: the blocks of a row sit on different XCDs),
	// G > 0 = the blocks of one row on ONE XCD, G consecutive rows per XCD and turn (xcd_row_order below; which lattices get it: luw_create)
	uint32_t xcd_rows;
};
// blockIdx.x / blockIdx.y of the block this workgroup works on: a permutation of the launch's blocks within each z layer -- of its first rows, as many as
// make whole turns of 8 G rows; the rows behind them keep the dispatch order
__device__ __forceinline__ void xcd_row_order(const KParams& p, uint32_t& bix, uint32_t& biy) {
	bix = blockIdx.x; biy = blockIdx.y;
	const uint32_t G = p.xcd_rows;                     // G consecutive rows per XCD and turn
	if(G&&blockIdx.y<gridDim.y-gridDim.y%(8u*G)) {
		const uint32_t pp = blockIdx.x+gridDim.x*blockIdx.y, sq = pp>>3, r = sq/gridDim.x;
		biy = (r/G)*8u*G+(pp&7u)*G+r%G; bix = sq%gridDim.x;
	}
}

#include "luw_codec.hpp"   // FP16C <-> FP32, inside namespace luw

// ---------------------------------------------------------------- IEEE division and square root without the range handling
// `a/b` and `sqrtf(x)` compile to the correctly rounded sequences of the device library: for the division two v_div_scale (operands
// scaled out of the denormal / overflow ranges), v_rcp_f32, two Newton fmas, the quotient, two residual corrections, v_div_fmas
// (undoes the scaling) and v_div_fixup (zeros, infinities, NaNs): 11 instructions; for the square root a 2^32 pre-scaling of inputs
// below 2^-96, v_sqrt_f32, the two neighbours of its result, their residuals, two selects, the un-scaling and a class test: 15.
// Where the VALU is the limit (FP16C kernels) and the operands sit in the plain range, the SAME instruction sequences run without the
// scaling and the special-case tails -- bit-identical results by construction, since those parts are identities there:
//   division n/d, 2^-60 <= d < 2^60 (an "ordinary" density: density_is_ordinary, tested per lane; other lanes redo the quotient with `/`),
//     n = 0 or 2^-25 <= |n| < 64 (moment sums of FP16C populations: multiples of 2^-25 below 38; the constants 0.5 and 2): v_div_scale leaves
//     both operands alone (ISA: it scales for a denormal d, 1/d or n/d, for |n| < 2^-103, for exponents 96 or more apart) and VCC = 0, so
//     v_div_fmas is a plain fma; v_div_fixup only re-applies the sign, which matters for n = -0 alone (this sequence gives +0: the sign of a
//     zero, never a value).  The reciprocal refinement depends on d only: several numerators share it.
//   square root, x = 0 or x >= 2^-96: no pre-scaling, and the class test (zero / infinity pass through) is covered by v_sqrt_f32 itself:
//     for x = 0 both neighbour tests fail (NaN and +0 residuals) and 0 stays.
// Checked on the device against the library forms: luw_selfcheck_arith (every float of the square root's range; > 2^32 quotients).
#ifndef LUW_PLAIN_ARITH
#define LUW_PLAIN_ARITH 1   /* 0: the library's division / square root everywhere (A/B builds) */
#endif
struct Recip { float d, r; };
__device__ __forceinline__ Recip recip_prepare(const float d) {
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e = fmaf(-d, r0, 1.0f);
	return Recip{ d, fmaf(e, r0, r0) };
}
__device__ __forceinline__ float div_by(const float n, const Recip R) {
	float q = n*R.r;
	q = fmaf(fmaf(-R.d, q, n), R.r, q);
	return fmaf(fmaf(-R.d, q, n), R.r, q);
}
__device__ __forceinline__ float sqrt_in_range(const float x) {
	const float s = __builtin_amdgcn_sqrtf(x);
	const float down = __uint_as_float(__float_as_uint(s)-1u), up = __uint_as_float(__float_as_uint(s)+1u);
	const float r_down = fmaf(-down, s, x), r_up = fmaf(-up, s, x);
	const float t = r_down<=0.0f ? down : s;
	return r_up>0.0f ? up : t;
}
// the densities for which div_by is the library's division: finite, positive, exponent in [-60, 60)
__device__ __forceinline__ bool density_is_ordinary(const float rho) { return __float_as_uint(rho)-0x21800000u<0x3C000000u; }

// ---------------------------------------------------------------- direction algebra, FX/kernel.cpp:890-893
// c_I . (a, b, c) with the zero terms dropped, evaluated left to right like c(i)*a+c(19+i)*b+c(38+i)*c (FX/kernel.cpp:1111).  D3Q19 order: rest, the six
// axis directions (+x -x +y -y +z -z), then the diagonals (++0 --0, +0+ -0-, 0++ 0--, +-0 -+0, +0- -0+, 0+- 0-+): odd I and I + 1 are opposite.
template<int I> __device__ __forceinline__ float cdot(const float a, const float b, const float c) {
	if constexpr(I== 1) return  a; else if constexpr(I== 2) return -a;
	else if constexpr(I== 3) return  b; else if constexpr(I== 4) return -b;
	else if constexpr(I== 5) return  c; else if constexpr(I== 6) return -c;
	else if constexpr(I== 7) return  a+b; else if constexpr(I== 8) return -a-b;
	else if constexpr(I== 9) return  a+c; else if constexpr(I==10) return -a-c;
	else if constexpr(I==11) return  b+c; else if constexpr(I==12) return -b-c;
	else if constexpr(I==13) return  a-b; else if constexpr(I==14) return -a+b;
	else if constexpr(I==15) return  a-c; else if constexpr(I==16) return -a+c;
	else if constexpr(I==17) return  b-c; else return -b+c; // 18
}
template<int K, typename Fn> __device__ __forceinline__ void for_each_pair(Fn&& fn) { // fn(K), fn(K + 1), ..., fn(8) with K a compile-time constant
	fn(std::integral_constant<int, K>{});
	if constexpr(K<8) for_each_pair<K+1>(fn);
}

// ---------------------------------------------------------------- f_eq, FX/kernel.cpp:1016-1055
// Second-order equilibrium of the SHIFTED populations (stored value = f - w_i): per opposite pair k (directions 2k+1, 2k+2) with s = 3 c.u and
// q = s^2 - 3 u^2, f_eq = w_k rho (q / 2 +- s) + w_k (rho - 1).  The nesting of the fused multiply-adds is the reference's (three deep per population:
// fma(s, s, -3 u^2), fma(1/2, q, +-s), fma(w rho, ., w (rho - 1))), which is what makes the values bit-equal to its restatement; the pair loop and the
// two weight classes (k < 3: axis directions, 1/18; else diagonals, 1/36) are this file's.
struct EqCommon { float q0, lead[2], base[2], s3[3]; };            // -3 u^2; w rho and w (rho - 1) per weight class; 3 u
__device__ __forceinline__ EqCommon eq_common(const float rho, const float ux, const float uy, const float uz, float& feq_rest) {
	EqCommon e;
	const float dev = rho-1.0f;
	e.q0 = -3.0f*(sq(ux)+sq(uy)+sq(uz));
	e.s3[0] = ux*3.0f; e.s3[1] = uy*3.0f; e.s3[2] = uz*3.0f;
	feq_rest = DEF_W0*fmaf(rho, 0.5f*e.q0, dev);
	e.lead[0] = DEF_WS*rho; e.lead[1] = DEF_WE*rho; e.base[0] = DEF_WS*dev; e.base[1] = DEF_WE*dev;
	return e;
}
__device__ __forceinline__ void calculate_f_eq(const float rho, const float ux, const float uy, const float uz, float* feq) {
	const EqCommon e = eq_common(rho, ux, uy, uz, feq[0]);
	for_each_pair<0>([&](auto kc) {
		constexpr int k = decltype(kc)::value, cls = k<3 ? 0 : 1;
		const float s = cdot<2*k+1>(e.s3[0], e.s3[1], e.s3[2]);
		const float q = fmaf(s, s, e.q0);
		feq[2*k+1] = fmaf(e.lead[cls], fmaf(0.5f, q, s), e.base[cls]);
		feq[2*k+2] = fmaf(e.lead[cls], fmaf(0.5f, q, -s), e.base[cls]);
	});
}

// ---------------------------------------------------------------- moments, FX/kernel.cpp:1075-1100
// the sums: density and the momentum before the division by it
__device__ __forceinline__ void moment_sums(const float* f, float& rho, float& mx, float& my, float& mz) {
	float r = f[0];
	#pragma unroll
	for(int i=1; i<19; i++) r += f[i];
	rho = r+1.0f;
	mx = f[ 1]-f[ 2]+f[ 7]-f[ 8]+f[ 9]-f[10]+f[13]-f[14]+f[15]-f[16];
	my = f[ 3]-f[ 4]+f[ 7]-f[ 8]+f[11]-f[12]+f[14]-f[13]+f[17]-f[18];
	mz = f[ 5]-f[ 6]+f[ 9]-f[10]+f[11]-f[12]+f[16]-f[15]+f[18]-f[17];
}
__device__ __forceinline__ void calculate_rho_u(const float* f, float& rhon, float& uxn, float& uyn, float& uzn) {
	float rho, ux, uy, uz;
	moment_sums(f, rho, ux, uy, uz);
	rhon = rho;
	uxn = ux/rho;
	uyn = uy/rho;
	uzn = uz/rho;
}

template<int I> __device__ __forceinline__ void forcing_term(const float ux, const float uy, const float uz, const float fx, const float fy, const float fz,
	const float uF, float* Fin) {
	constexpr float w9 = 9.0f*(I<7 ? DEF_WS : DEF_WE);
	Fin[I] = w9*fmaf(cdot<I>(fx, fy, fz), cdot<I>(ux, uy, uz)+0.33333334f, uF);
	if constexpr(I<18) forcing_term<I+1>(ux, uy, uz, fx, fy, fz, uF, Fin);
}
// Guo forcing, FX/kernel.cpp:1103-1113
__device__ __forceinline__ void calculate_forcing_terms(const float ux, const float uy, const float uz, const float fx, const float fy, const float fz,
	float* Fin) {
	const float uF = -0.33333334f*fmaf(ux, fx, fmaf(uy, fy, uz*fz));
	Fin[0] = 9.0f*DEF_W0*uF;
	forcing_term<1>(ux, uy, uz, fx, fy, fz, uF, Fin);
}

// ---------------------------------------------------------------- LUW force assembly, FX/kernel.cpp:1516-1623
// Adds Coriolis, buffer nudging, top sponge and the per-cell force to (fxn,fyn,fzn).  n is the device index of
// the cell, (x,y,z) its local coordinates.  u is the device velocity field (3 planes of stride Np).
// ZONES = false: the caller knows that the cell lies outside the nudging / sponge zones and that there is no force field -- what is
// left is the volume force and Coriolis (the uniform part)
// The nudging / sponge references of one cell, fetched AHEAD of its collision (pair kernel, general instantiation): what assemble_force would read
// through n_ref once the moments are known -- the target velocity of the nearest owned face with its weight, the top layer's velocity with the
// sponge's sigma.  They depend on the position alone, so the loads go out with the DDF loads and their latency (four dependent trips to memory per
// lane pair otherwise, ~3400 of a wave's 21000 cycles on the urban tile: profiles/r03_stall_counters.md) is hidden behind decode and moments.
// A TYPE_E cell takes neither term; its four registers carry what it reads instead, its own rho and u (FX/kernel.cpp:1516-1523), fetched just as early.
struct ForceRefs { float tu[3], wb, su[3], sg; bool zn, zs; };
__device__ __forceinline__ void fetch_force_refs(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y, const uint32_t z, const bool live,
	const bool is_E,
		const float* __restrict__ rho, const float* __restrict__ u, ForceRefs& r) {
	// (values of lanes outside the zones are never read: defined as "whatever the register holds", at no instruction)
	asm volatile("" : "=v"(r.tu[0]), "=v"(r.tu[1]), "=v"(r.tu[2]), "=v"(r.wb), "=v"(r.su[0]), "=v"(r.su[1]), "=v"(r.su[2]), "=v"(r.sg));
	const bool own = live && is_E;                                  // reads its own fields
	const bool act = live && !is_E;                                 // the zone terms can act (a cell that is not collided: nothing)
	bool zn = false;
	uint32_t d_min = 0u, n_ref = n;
	if(p.buffer_active) { // the selection of assemble_force
		const uint32_t d_w = x-(uint32_t)p.west_x, d_e = (uint32_t)p.east_x-x, d_s = y-(uint32_t)p.south_y, d_n = (uint32_t)p.north_y-y,
			d_t = (uint32_t)p.top_z-z;
		const bool in_w = x-p.zw_lo<p.zw_n, in_e = x-p.ze_lo<p.ze_n, in_s = y-p.zs_lo<p.zs_n, in_n = y-p.zn_lo<p.zn_n, in_t = z-p.zt_lo<p.zt_n;
		zn = act && (in_w||in_e||in_s||in_n||in_t);
		if(zn) {
			// assemble_force walks w, e, s, n, t and keeps the first nearest face.  The three that depend on (y, z) alone -- the same for every lane
			// of a row, scalar work in the pair kernel -- are settled among themselves first (same order, same strict <); w and e, which come before
			// them, then only lose to a strictly nearer one: the same winner.
			uint32_t d_row = p.buffer_N+1u, base_row = 0u;
			if(in_s) { if(d_s<d_row) { d_row = d_s; base_row = ((uint32_t)p.south_y+z*p.Ny)*p.Px; } }
			if(in_n) { if(d_n<d_row) { d_row = d_n; base_row = ((uint32_t)p.north_y+z*p.Ny)*p.Px; } }
			if(in_t) { if(d_t<d_row) { d_row = d_t; base_row = (y+(uint32_t)p.top_z*p.Ny)*p.Px; } }
			d_min = p.buffer_N+1u;
			const uint32_t rowyz = (y+z*p.Ny)*p.Px;
			if(in_w) { if(d_w<d_min) { d_min = d_w; n_ref = (uint32_t)p.west_x+rowyz; } }
			if(in_e) { if(d_e<d_min) { d_min = d_e; n_ref = (uint32_t)p.east_x+rowyz; } }
			if(d_row<d_min) { d_min = d_row; n_ref = x+base_row; }
		}
	}
	r.zn = zn;
	// ONE run of four loads for both kinds of lane (two runs in two branches made the second wait for every load in flight before it could reuse
	// the first one's address registers): weight and target velocity of a zone lane, own density and velocity of a TYPE_E lane (n_ref = n there)
	if(zn||own) {
		const float* const first = own ? rho+n : p.wbuf+d_min;
		r.wb = *first;
		r.tu[0] = u[n_ref]; r.tu[1] = u[(size_t)p.Np+n_ref]; r.tu[2] = u[2ull*p.Np+n_ref];
	}
	r.zs = act && z-p.zp_lo<p.zp_n;
	if(r.zs) {
		r.sg = p.sigma[(uint32_t)p.top_z-1u-z];
		const uint32_t n_top = x+(y+(uint32_t)p.top_z*p.Ny)*p.Px;
		r.su[0] = u[n_top]; r.su[1] = u[(size_t)p.Np+n_top]; r.su[2] = u[2ull*p.Np+n_top];
	}
}
// refs (pair kernel, general instantiation only): the references above, already fetched -- same arithmetic on them
template<bool ZONES=true> __device__ __forceinline__ void assemble_force(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y,
	const uint32_t z, const bool is_E,
		const float rhon, const float uxn, const float uyn, const float uzn, const float* __restrict__ u, const float* __restrict__ F, float& fxn, float& fyn,
			float& fzn,
		const ForceRefs* refs = nullptr) {
	fxn = p.fx; fyn = p.fy; fzn = p.fz;
	if(p.coriolis) { // with omega = 0 the three terms are +-0: adding them changes no value
		const float cor_x = -2.0f*rhon*(p.omy*uzn-p.omz*uyn);
		const float cor_y = -2.0f*rhon*(p.omz*uxn-p.omx*uzn);
		const float cor_z = -2.0f*rhon*(p.omx*uyn-p.omy*uxn);
		fxn += cor_x;
		fyn += cor_y;
		fzn += cor_z;
	}
	if constexpr(!ZONES) return;
	if(refs) {
		if(refs->zn) {
			const float a_x = refs->wb*p.buffer_inv_tau*(refs->tu[0]-uxn);
			const float a_y = refs->wb*p.buffer_inv_tau*(refs->tu[1]-uyn);
			const float a_z = p.nudge_vertical==1u ? refs->wb*p.buffer_inv_tau*(refs->tu[2]-uzn) : 0.0f;
			fxn += rhon*a_x;
			fyn += rhon*a_y;
			fzn += rhon*a_z;
		}
		if(refs->zs) {
			fxn += rhon*refs->sg*(refs->su[0]-uxn);
			fyn += rhon*refs->sg*(refs->su[1]-uyn);
			fzn += rhon*refs->sg*(refs->su[2]-uzn);
		}
		if(p.has_F) {
			fxn += F[n];
			fyn += F[(size_t)p.Np+n];
			fzn += F[2ull*p.Np+n];
		}
		return;
	}
	if(p.buffer_active && !is_E) {
		// distance to each face (meaningful inside that face's zone, wrapped-around garbage outside, where it is not used)
		const uint32_t d_w = x-(uint32_t)p.west_x, d_e = (uint32_t)p.east_x-x, d_s = y-(uint32_t)p.south_y, d_n = (uint32_t)p.north_y-y,
			d_t = (uint32_t)p.top_z-z;
		const bool in_w = x-p.zw_lo<p.zw_n, in_e = x-p.ze_lo<p.ze_n, in_s = y-p.zs_lo<p.zs_n, in_n = y-p.zn_lo<p.zn_n, in_t = z-p.zt_lo<p.zt_n;
		if(in_w||in_e||in_s||in_n||in_t) {
			uint32_t d_min = p.buffer_N+1u;
			uint32_t n_ref = n;
			const uint32_t rowyz = (y+z*p.Ny)*p.Px;
			if(in_w) { if(d_w<d_min) { d_min = d_w; n_ref = (uint32_t)p.west_x+rowyz; } }
			if(in_e) { if(d_e<d_min) { d_min = d_e; n_ref = (uint32_t)p.east_x+rowyz; } }
			if(in_s) { if(d_s<d_min) { d_min = d_s; n_ref = x+((uint32_t)p.south_y+z*p.Ny)*p.Px; } }
			if(in_n) { if(d_n<d_min) { d_min = d_n; n_ref = x+((uint32_t)p.north_y+z*p.Ny)*p.Px; } }
			if(in_t) { if(d_t<d_min) { d_min = d_t; n_ref = x+(y+(uint32_t)p.top_z*p.Ny)*p.Px; } }
			const float w_buf = p.wbuf[d_min];
			const float u_target_x = u[n_ref];
			const float u_target_y = u[(size_t)p.Np+n_ref];
			const float u_target_z = u[2ull*p.Np+n_ref];
			const float a_x = w_buf*p.buffer_inv_tau*(u_target_x-uxn);
			const float a_y = w_buf*p.buffer_inv_tau*(u_target_y-uyn);
			const float a_z = p.nudge_vertical==1u ? w_buf*p.buffer_inv_tau*(u_target_z-uzn) : 0.0f;
			fxn += rhon*a_x;
			fyn += rhon*a_y;
			fzn += rhon*a_z;
		}
	}
	if(!is_E && z-p.zp_lo<p.zp_n) { // top sponge (zp_n = 0 unless the term is on and this domain owns the top)
		const float sigma = p.sigma[(uint32_t)p.top_z-1u-z];   // layer index (Nzg-2) - (z+Oz), FX/kernel.cpp:1600
		const uint32_t n_ref = x+(y+(uint32_t)p.top_z*p.Ny)*p.Px;
		fxn += rhon*sigma*(u[n_ref]-uxn);
		fyn += rhon*sigma*(u[(size_t)p.Np+n_ref]-uyn);
		fzn += rhon*sigma*(u[2ull*p.Np+n_ref]-uzn);
	}
	if(p.has_F) {
		fxn += F[n];
		fyn += F[(size_t)p.Np+n];
		fzn += F[2ull*p.Np+n];
	}
}

// ---------------------------------------------------------------- collision of one cell, FX/kernel.cpp:1502-1515,1686-1748
// Smagorinsky-Lilly relaxation rate, FX/kernel.cpp:1723-1737; sums run over i=1..18 in order, zero terms dropped (the
// leading 0 + n of each sum too: it can only turn -0 into +0, and every sum is squared)
__device__ __forceinline__ float smagorinsky_Q(const float* n_) {
	float Hxx = n_[ 1], Hyy = n_[ 3], Hzz = n_[ 5], Hxy = n_[ 7], Hxz = n_[ 9], Hyz = n_[11];
	Hxx += n_[ 2]; Hxx += n_[ 7]; Hxx += n_[ 8]; Hxx += n_[ 9]; Hxx += n_[10]; Hxx += n_[13]; Hxx += n_[14]; Hxx += n_[15]; Hxx += n_[16];
	Hyy += n_[ 4]; Hyy += n_[ 7]; Hyy += n_[ 8]; Hyy += n_[11]; Hyy += n_[12]; Hyy += n_[13]; Hyy += n_[14]; Hyy += n_[17]; Hyy += n_[18];
	Hzz += n_[ 6]; Hzz += n_[ 9]; Hzz += n_[10]; Hzz += n_[11]; Hzz += n_[12]; Hzz += n_[15]; Hzz += n_[16]; Hzz += n_[17]; Hzz += n_[18];
	Hxy += n_[ 8]; Hxy += -n_[13]; Hxy += -n_[14];
	Hxz += n_[10]; Hxz += -n_[15]; Hxz += -n_[16];
	Hyz += n_[12]; Hyz += -n_[17]; Hyz += -n_[18];
	return sq(Hxx)+sq(Hyy)+sq(Hzz)+2.0f*(sq(Hxy)+sq(Hxz)+sq(Hyz));
}
__device__ __forceinline__ float smagorinsky_rate_of_Q(const KParams& p, const float rhon, const float Q) {
	return 2.0f/(p.tau0+sqrtf(p.tau0sq+0.76421222f*sqrtf(Q)/rhon));
}
// the same value for an ordinary density (R = recip_prepare(rhon)): FP16C populations keep Q below ~1e4 and tau in [1, 20], plain operands for
// sqrt_in_range and div_by; a Q under 2^-96, where sqrt_in_range may be an ulp off, and a quotient s/rho so small that div_by's unscaled residuals
// lose bits, both add less than a quarter ulp to tau0sq >= 1/4 and change nothing
__device__ __forceinline__ float smagorinsky_rate_plain(const KParams& p, const float Q, const Recip R) {
	const float s = 0.76421222f*sqrt_in_range(Q);
	const float tau = p.tau0+sqrt_in_range(p.tau0sq+div_by(s, R));
	return div_by(2.0f, recip_prepare(tau));
}
__device__ __forceinline__ float smagorinsky_rate(const KParams& p, const float rhon, const float* n_) {
	return smagorinsky_rate_of_Q(p, rhon, smagorinsky_Q(n_));
}
// PLAIN (FP16C storage, LUW_PLAIN_ARITH): divisions by the density and the square roots as plain-range sequences (recip_prepare); R is the prepared
// reciprocal of rhon, odd_density marks the lanes that redo those results with the library forms
struct DensityRecip { Recip R; bool odd; };
template<bool PLAIN> __device__ __forceinline__ float relaxation_rate(const KParams& p, const float rhon, const float* f, const float* feq,
	const DensityRecip& dr) {
	if(!p.subgrid) return p.w;
	float n_[19];
	#pragma unroll
	for(int i=1; i<19; i++) n_[i] = f[i]-feq[i];
	const float Q = smagorinsky_Q(n_);
	if constexpr(PLAIN) {
		float w = smagorinsky_rate_plain(p, Q, dr.R);
		if(dr.odd) w = smagorinsky_rate_of_Q(p, rhon, Q);
		return w;
	} else return smagorinsky_rate_of_Q(p, rhon, Q);
}
// rho, u of the cell (moments, or the stored values on TYPE_E cells) and the force acting on it
template<bool NOFORCE=false, bool PLAIN=false> __device__ __forceinline__ void collide_head(const KParams& p, const uint32_t n, const uint32_t x,
	const uint32_t y, const uint32_t z, const bool is_E, const float* f,
		const float* __restrict__ rho, const float* __restrict__ u, const float* __restrict__ F, float& rhon, float& uxn, float& uyn, float& uzn, float& fxn,
			float& fyn, float& fzn, DensityRecip& dr) {
	if(is_E) {
		rhon = rho[n];
		uxn = u[n];
		uyn = u[(size_t)p.Np+n];
		uzn = u[2ull*p.Np+n];
		if constexpr(PLAIN) { dr.R = recip_prepare(rhon); dr.odd = !density_is_ordinary(rhon); }
	} else if constexpr(PLAIN) {
		float mx, my, mz;
		moment_sums(f, rhon, mx, my, mz);
		dr.R = recip_prepare(rhon); dr.odd = !density_is_ordinary(rhon);
		uxn = div_by(mx, dr.R); uyn = div_by(my, dr.R); uzn = div_by(mz, dr.R);
		if(dr.odd) { uxn = mx/rhon; uyn = my/rhon; uzn = mz/rhon; }
	} else {
		calculate_rho_u(f, rhon, uxn, uyn, uzn);
	}
	if constexpr(NOFORCE) fxn = fyn = fzn = 0.0f;   // the caller knows that nothing can push the cells of its launch box (luw_core.hip, box_force_mode)
	else assemble_force(p, n, x, y, z, is_E, rhon, uxn, uyn, uzn, u, F, fxn, fyn, fzn);
}
// the general tail: Guo forcing, equilibrium override on TYPE_E cells
template<bool PLAIN=false> __device__ __forceinline__ void collide_tail_general(const KParams& p, const bool is_E, const bool forced, const float fxn,
	const float fyn, const float fzn,
		float* f, const float rhon, float& uxn, float& uyn, float& uzn, const DensityRecip& dr) {
	float feq[19], Fin[19];
	if(forced) {
		float rho2;
		if constexpr(PLAIN) {
			rho2 = div_by(0.5f, dr.R);
			if(dr.odd) { asm volatile(""); rho2 = 0.5f/rhon; } // (the empty asm keeps this a branch, see collide_cell_pk)
		} else rho2 = 0.5f/rhon;
		uxn = clampf(fmaf(fxn, rho2, uxn), -DEF_C, DEF_C);
		uyn = clampf(fmaf(fyn, rho2, uyn), -DEF_C, DEF_C);
		uzn = clampf(fmaf(fzn, rho2, uzn), -DEF_C, DEF_C);
		calculate_forcing_terms(uxn, uyn, uzn, fxn, fyn, fzn, Fin);
	} else {
		uxn = clampf(uxn, -DEF_C, DEF_C);
		uyn = clampf(uyn, -DEF_C, DEF_C);
		uzn = clampf(uzn, -DEF_C, DEF_C);
		#pragma unroll
		for(int i=0; i<19; i++) Fin[i] = 0.0f;
	}
	calculate_f_eq(rhon, uxn, uyn, uzn, feq);
	const float w = relaxation_rate<PLAIN>(p, rhon, f, feq, dr);
	const float c_tau = fmaf(w, -0.5f, 1.0f);
	const float omw = 1.0f-w;
	#pragma unroll
	for(int i=0; i<19; i++) {
		const float Fi = Fin[i]*c_tau;
		f[i] = is_E ? feq[i] : fmaf(omw, f[i], fmaf(w, feq[i], Fi));
	}
}
// in: streamed-in DDFs f[19], flags byte.  out: post-collision DDFs in f[19]; rho/u (after the half-force
// shift and the +-c clamp) in rhon,uxn,uyn,uzn.
// NOFORCE: no force can act on any cell of the launch (box_force_mode): without the force assembly the FP16C kernel needs 69 instead of 89
// VGPRs (76 instead of 93 with the thermal lattice) and no scalar spills -- 7 resp. 6 waves per SIMD instead of 5
// PLAIN: the populations come out of FP16C storage (bounded operands): plain-range divisions and square roots (recip_prepare)
template<bool FAST=true, bool NOFORCE=false, bool PLAIN=false> __device__ __forceinline__ void collide_cell(const KParams& p, const uint32_t n,
	const uint32_t x, const uint32_t y, const uint32_t z, const uint8_t flagsn,
		float* f, const float* __restrict__ rho, const float* __restrict__ u, const float* __restrict__ F, float& rhon, float& uxn, float& uyn, float& uzn,
			float* u_before_force = nullptr) {
	const bool is_E = (flagsn&TYPE_BO)==TYPE_E;
	float fxn, fyn, fzn;
	DensityRecip dr{};
	collide_head<NOFORCE, PLAIN>(p, n, x, y, z, is_E, f, rho, u, F, rhon, uxn, uyn, uzn, fxn, fyn, fzn, dr);
	// what the thermal lattice advects with (FX/kernel.cpp:1669)
	if(u_before_force) { u_before_force[0] = uxn; u_before_force[1] = uyn; u_before_force[2] = uzn; }
	// A cell without any force (the bulk of an urban case: no volume force, outside the nudging / sponge zones, no
	// Coriolis) has Fin_i = +-0 exactly and u += 0/(2 rho); skipping that arithmetic is value-identical
	// (fma(w, feq, +-0) == w*feq) as long as rho != 0.
	const bool forced = fxn!=0.0f||fyn!=0.0f||fzn!=0.0f;
	// Wave-uniform fast path: when no lane of the wave is a TYPE_E cell or feels a force (interior waves of an urban case),
	// the equilibrium override and the Guo terms drop out for the whole wave -- one scalar branch instead of 19 selects, 19
	// products with zero and the zero-filled Fin registers per lane.  Same values (+-0 aside) as the general path below.
	if(FAST&&__ballot(is_E||forced)==0ull) {
		float feq[19];
		uxn = clampf(uxn, -DEF_C, DEF_C);
		uyn = clampf(uyn, -DEF_C, DEF_C);
		uzn = clampf(uzn, -DEF_C, DEF_C);
		calculate_f_eq(rhon, uxn, uyn, uzn, feq);
		const float w = relaxation_rate<PLAIN>(p, rhon, f, feq, dr);
		const float omw = 1.0f-w;
		#pragma unroll
		for(int i=0; i<19; i++) f[i] = fmaf(omw, f[i], w*feq[i]);
		return;
	}
	collide_tail_general<PLAIN>(p, is_E, forced, fxn, fyn, fzn, f, rhon, uxn, uyn, uzn, dr);
}

} // namespace luw
#include "luw_device_pair.hpp"     // the collision on packed pairs (exact) ...
#include "luw_device_native.hpp"   // ... and in native arithmetic
namespace luw {

// position-only test: can buffer nudging or the top sponge act on this cell (the zones of assemble_force)?
__device__ __forceinline__ bool in_force_zone(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z) {
	return x-p.zw_lo<p.zw_n || x-p.ze_lo<p.ze_n || y-p.zs_lo<p.zs_n || y-p.zn_lo<p.zn_n || z-p.zt_lo<p.zt_n || z-p.zp_lo<p.zp_n;
}

// ---------------------------------------------------------------- thermal D3Q7 lattice (TEMPERATURE), FX/kernel.cpp:1306-1335,1639-1684
// D3Q7 equilibrium of the shifted populations: rest T / 4 - 1 / 4, axis pair a: (T / 8)(1 +- 4 u_a) - 1 / 8 as one fma each on T / 2 and (T - 1) / 8
__device__ __forceinline__ void calculate_g_eq(const float T, const float ux, const float uy, const float uz, float* geq) {
	const float half_T = 0.5f*T, shift = 0.125f*(T-1.0f);
	const float ua[3] = { ux, uy, uz };
	geq[0] = fmaf(0.25f, T, -0.25f);
	#pragma unroll
	for(int a=0; a<3; a++) { geq[2*a+1] = fmaf(half_T, ua[a], shift); geq[2*a+2] = fmaf(half_T, -ua[a], shift); }
}
// the cell update on streamed-in populations g[7] (in place): T = sum g + 1 (or the preset on TYPE_T cells), top sponge on T, BGK with w_T
// (TYPE_T: g = g_eq); writes T of a cell that is not preset.  FX/kernel.cpp:1652-1684
// write_T: like rho and u (write_fields), T is a by-product -- every step recomputes it from g; the only T the kernel READS are presets and, under the
// sponge, the top layer's -- so it is stored by the steps whose fields are looked at (the last of a run, sampled steps, every step on request or when
// a top-layer cell under the sponge is not a preset: reference_cells_are_inputs)
__device__ __forceinline__ void thermal_cell(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y, const uint32_t z, const uint8_t flagsn,
		const float ux, const float uy, const float uz, float* __restrict__ Tf, float* g, const bool write_T = true) {
	const bool preset = (flagsn&TYPE_T)!=0u;
	float Tn;
	if(preset) Tn = Tf[n];
	else { Tn = 0.0f; for(int i=0; i<7; i++) Tn += g[i]; Tn += 1.0f; }
	if(!preset && (flagsn&TYPE_BO)!=TYPE_E && z-p.zp_lo<p.zp_n) Tn = fmaf(p.sigma[(uint32_t)p.top_z-1u-z], Tf[x+(y+(uint32_t)p.top_z*p.Ny)*p.Px]-Tn, Tn);
	float geq[7];
	calculate_g_eq(Tn, ux, uy, uz, geq);
	if(preset) { for(int i=0; i<7; i++) g[i] = geq[i]; }
	else {
		if(write_T) Tf[n] = Tn;
		const float omw = 1.0f-p.w_T;
		for(int i=0; i<7; i++) g[i] = fmaf(omw, g[i], p.w_T*geq[i]);
	}
}
// One cell of the temperature lattice: Esoteric-Pull stream-in, T = sum g + 1 (or the preset on TYPE_T cells), top sponge
// on T, BGK with w_T (TYPE_T: g = g_eq), stream-out.  n, jx, jy, jz are ELEMENT indices of the cell and its +x, +y, +z
// neighbours; (ux,uy,uz) is the velocity before the force shift.  The buoyancy term it would add to the force carries the
// factor (fx,fy,fz), which LUW sets to zero (FX/setup.cpp:4935): T is a passive scalar here.
template<typename T,
	int PARITY> __device__ __forceinline__ void thermal_collide(const KParams& p, const uint32_t n, const uint32_t jx, const uint32_t jy, const uint32_t jz,
		const uint32_t x, const uint32_t y, const uint32_t z, const uint8_t flagsn, const float ux, const float uy, const float uz, const T* __restrict__ gi,
			float* __restrict__ Tf, float* g, const bool write_T = true) {
	const size_t Np = p.Np;
	const uint32_t jn[3] = { jx, jy, jz };
	g[0] = ddf_decode<T>(gi[n]);
	#pragma unroll
	for(int k=0; k<3; k++) {
		const int i = 2*k+1;
		g[i  ] = ddf_decode<T>(gi[(size_t)(PARITY ? i : i+1)*Np+n]);
		g[i+1] = ddf_decode<T>(gi[(size_t)(PARITY ? i+1 : i)*Np+jn[k]]);
	}
	thermal_cell(p, n, x, y, z, flagsn, ux, uy, uz, Tf, g, write_T);
}
// stream-out of the 7 encoded populations (Esoteric-Pull slots); code_of(i) yields the storage value of population i
template<typename T, int PARITY, typename F> __device__ __forceinline__ void thermal_store(const KParams& p, const uint32_t n, const uint32_t jx,
	const uint32_t jy, const uint32_t jz, T* __restrict__ gi, F code_of) {
	const size_t Np = p.Np;
	const uint32_t jn[3] = { jx, jy, jz };
	gi[n] = code_of(0);
	#pragma unroll
	for(int k=0; k<3; k++) {
		const int i = 2*k+1;
		gi[(size_t)(PARITY ? i+1 : i)*Np+jn[k]] = code_of(i);
		gi[(size_t)(PARITY ? i : i+1)*Np+n] = code_of(i+1);
	}
}
} // namespace luw
