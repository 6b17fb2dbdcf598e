// luw_device_pair.hpp -- the collision of one cell on PACKED pairs of opposite directions (v_pk_fma / mul / add_f32), bit-identical to collide_cell of
// luw_device.hpp: what the FP16C pair kernel runs with `--arith exact`.  Included by luw_device.hpp.
#pragma once

namespace luw {

// ---------------------------------------------------------------- the same collision on PACKED pairs
// The 18 moving populations as nine pairs (f[2k+1], f[2k+2]) of opposite directions in 64-bit register pairs: the fast
// path's equilibria, non-equilibrium parts and relaxation are v_pk_fma/mul/add_f32 on those pairs (two IEEE operations per
// instruction, same roundings as the scalar code: value-identical).  Sums whose order is fixed (moments, stress tensor)
// read the halves.  Used where the VALU is the limit (FP16C pair kernel).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(const float a) { f32x2 r = { a, a }; return r; }
__device__ __forceinline__ f32x2 pm2(const float a) { f32x2 r = { a, -a }; return r; }
__device__ __forceinline__ void calculate_f_eq_pk(const float rho, const float ux, const float uy, const float uz, float& feq0, f32x2* feqp) {
	const EqCommon e = eq_common(rho, ux, uy, uz, feq0);
	for_each_pair<0>([&](auto kc) { // both populations of pair k in the two halves of packed instructions: (q / 2 + s, q / 2 - s), then the weights
		constexpr int k = decltype(kc)::value, cls = k<3 ? 0 : 1;
		const float s = cdot<2*k+1>(e.s3[0], e.s3[1], e.s3[2]);
		const f32x2 inner = __builtin_elementwise_fma(splat2(0.5f), splat2(fmaf(s, s, e.q0)), pm2(s));
		feqp[k] = __builtin_elementwise_fma(splat2(e.lead[cls]), inner, splat2(e.base[cls]));
	});
}
// Guo terms of the pair (2k+1, 2k+2): c_(2k+2) = -c_(2k+1), so both are w9 fma(+-cF, +-cu + 1/3, uF) (FX/kernel.cpp:1103-1113)
template<int K> __device__ __forceinline__ f32x2 forcing_pair(const float ux, const float uy, const float uz, const float fx, const float fy, const float fz,
	const float uF) {
	constexpr int I = 2*K+1;
	constexpr float w9 = 9.0f*(I<7 ? DEF_WS : DEF_WE);
	// c_(2k+2) . v = -(c_(2k+1) . v) exactly (-a-b and -(a+b) round alike), so the second lane takes the first lane's sums through the packed
	// instructions' negate modifiers instead of two more additions each
	const float cF = cdot<I>(fx, fy, fz), cu = cdot<I>(ux, uy, uz);
	const f32x2 a = { cF, -cF }, b = { cu, -cu };
	return splat2(w9)*__builtin_elementwise_fma(a, b+splat2(0.33333334f), splat2(uF));
}
// All cases in one: wave-uniform switches for "some lane may feel a force" and "some lane is a TYPE_E cell" select the
// extra work; a lane for which the switch is on without need computes with F = 0 (u + 0/(2 rho) = u, Fin = +-0:
// value-identical, +-0 aside, to the scalar code's per-lane shortcut).
// FORCE says what the caller knows about the launch box (luw_core.hip, pair_force_mode):
//   PAIR_FORCE_NONE     no force can act on any of its cells (no Coriolis, volume force or force field, box outside the nudging / sponge
//                       zones): the force assembly and the Guo terms are compiled out, which is what lets the kernel fit 5 waves per SIMD
//                       (86 instead of 109 VGPRs, no scalar spills).  TYPE_E cells then take no selects either: the caller decodes their
//                       populations as f = 0 and this routine relaxes them with w = 1, so that fma(1 - w, f, w f_eq) = fma(0, 0, f_eq) =
//                       f_eq bit for bit (f_eq is never -0: an exact cancellation gives +0, and at rho = 1, u = 0 every term is +0),
//                       whatever the Smagorinsky rate of such a lane came out as -- instead of keeping the nineteen equilibria for a
//                       select behind the relaxation;
//   PAIR_FORCE_UNIFORM  volume force and / or Coriolis act on every cell, nothing position-dependent does: no zone tests, no wave-uniform
//                       switch, no scalar spills; TYPE_E lanes as above, with the Guo term's factor c_tau = 0 on top (feq + Fi 0 = feq):
//                       96 VGPRs, 5 waves per SIMD as well;
//   PAIR_FORCE_ANY      everything, switched per wave.
enum { PAIR_FORCE_NONE = 0, PAIR_FORCE_UNIFORM = 1, PAIR_FORCE_ANY = 2 };

// ORDINARY densities (density_is_ordinary: 2^-60 <= rho < 2^60 -- anything a lattice that has not blown up holds) take the divisions and square
// roots as the library's instruction sequences minus their range handling (recip_prepare), the five divisions by the density sharing one
// reciprocal: 808 instead of 876 VALU instructions per lane in the force-free kernel, 992 instead of 1074 with uniform forces.  Lanes with any
// other density (zero, negative, NaN, absurd) redo exactly those results with the library forms inside rarely taken divergent blocks, so the
// values are the IEEE ones for EVERY input.  (A per-wave vote between two complete collisions was measured first: the duplicated code cost
// more than the arithmetic saved, 1024x1024x256 + Coriolis 4.05 -> 4.25 ms; this form 4.05 -> 3.89, profiles/r03_plain_arith_ab.txt.)
template<int FORCE=PAIR_FORCE_ANY> __device__ __forceinline__ void collide_cell_pk(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y,
	const uint32_t z, const uint8_t flagsn, const bool may_force,
		float& f0, f32x2* fp, const float* __restrict__ rho, const float* __restrict__ u, const float* __restrict__ F, float& rhon, float& uxn, float& uyn,
			float& uzn, float* u_before_force = nullptr,
		// refs: zone references fetched early (fetch_force_refs); own: a TYPE_E cell's rho / u fetched early (wb, tu)
		const ForceRefs* refs = nullptr, const ForceRefs* own = nullptr) {
	constexpr bool PLAIN = LUW_PLAIN_ARITH!=0;
	const bool is_E = (flagsn&TYPE_BO)==TYPE_E;
	const bool wave_has_E = __ballot(is_E)!=0ull;
	float rho_m, mx, my, mz;
	{
		float f[19];
		f[0] = f0;
		#pragma unroll
		for(int k=0; k<9; k++) { f[2*k+1] = fp[k].x; f[2*k+2] = fp[k].y; }
		moment_sums(f, rho_m, mx, my, mz);
	}
	rhon = rho_m;
	[[maybe_unused]] Recip R{};   // of the density the divisions below use: the moment sum, or (TYPE_E lanes) the stored field
	[[maybe_unused]] bool odd_density = false;
	if constexpr(PLAIN) {
		R = recip_prepare(rho_m); uxn = div_by(mx, R); uyn = div_by(my, R); uzn = div_by(mz, R);
		odd_density = !density_is_ordinary(rho_m);
		if(odd_density) { uxn = mx/rho_m; uyn = my/rho_m; uzn = mz/rho_m; }
	} else { uxn = mx/rho_m; uyn = my/rho_m; uzn = mz/rho_m; }
	if(wave_has_E) {
		if(is_E) {
			if(own) { rhon = own->wb; uxn = own->tu[0]; uyn = own->tu[1]; uzn = own->tu[2]; } // fetched ahead of the decode
			else {
				rhon = rho[n];
				uxn = u[n];
				uyn = u[(size_t)p.Np+n];
				uzn = u[2ull*p.Np+n];
			}
		}
		// rhon: the field value on TYPE_E lanes, the moment sum elsewhere
		if constexpr(PLAIN) { R = recip_prepare(rhon); odd_density = !density_is_ordinary(rhon); }
	}
	// what the thermal lattice advects with (FX/kernel.cpp:1669)
	if(u_before_force) { u_before_force[0] = uxn; u_before_force[1] = uyn; u_before_force[2] = uzn; }
	const bool forced = FORCE==PAIR_FORCE_UNIFORM || (FORCE==PAIR_FORCE_ANY && may_force);
	f32x2 Finp[9]; float Fin0 = 0.0f;
	if(forced) {
		float fxn, fyn, fzn;
		assemble_force<(FORCE==PAIR_FORCE_ANY)>(p, n, x, y, z, is_E, rhon, uxn, uyn, uzn, u, F, fxn, fyn, fzn, refs);
		float rho2;
		if constexpr(PLAIN) {
			rho2 = div_by(0.5f, R);
			// (the empty asm keeps this a branch: a lone division would be hoisted in front of a select and run for every lane)
			if(odd_density) { asm volatile(""); rho2 = 0.5f/rhon; }
		} else rho2 = 0.5f/rhon;
		uxn = clampf(fmaf(fxn, rho2, uxn), -DEF_C, DEF_C);
		uyn = clampf(fmaf(fyn, rho2, uyn), -DEF_C, DEF_C);
		uzn = clampf(fmaf(fzn, rho2, uzn), -DEF_C, DEF_C);
		const float uF = -0.33333334f*fmaf(uxn, fxn, fmaf(uyn, fyn, uzn*fzn));
		Fin0 = 9.0f*DEF_W0*uF;
		Finp[0] = forcing_pair<0>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[1] = forcing_pair<1>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[2] = forcing_pair<2>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[3] = forcing_pair<3>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[4] = forcing_pair<4>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[5] = forcing_pair<5>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[6] = forcing_pair<6>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[7] = forcing_pair<7>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[8] = forcing_pair<8>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
	} else {
		uxn = clampf(uxn, -DEF_C, DEF_C);
		uyn = clampf(uyn, -DEF_C, DEF_C);
		uzn = clampf(uzn, -DEF_C, DEF_C);
	}
	float feq0; f32x2 feqp[9];
	calculate_f_eq_pk(rhon, uxn, uyn, uzn, feq0, feqp);
	float w = p.w;
	if(p.subgrid) {
		float n_[19];
		#pragma unroll
		for(int k=0; k<9; k++) { const f32x2 d = fp[k]-feqp[k]; n_[2*k+1] = d.x; n_[2*k+2] = d.y; }
		const float Q = smagorinsky_Q(n_);
		if constexpr(PLAIN) { w = smagorinsky_rate_plain(p, Q, R); if(odd_density) w = smagorinsky_rate_of_Q(p, rhon, Q); }
		else w = smagorinsky_rate_of_Q(p, rhon, Q);
	}
	constexpr bool E_BY_RATE = FORCE!=PAIR_FORCE_ANY;   // TYPE_E lanes through the relaxation rate (f = 0, w = 1, no Guo term) instead of nineteen selects
	if constexpr(E_BY_RATE) { if(wave_has_E) w = is_E ? 1.0f : w; }
	const float omw = 1.0f-w;
	float r0; f32x2 rp[9];
	if(forced) {
		float c_tau = fmaf(w, -0.5f, 1.0f);
		if constexpr(E_BY_RATE) { if(wave_has_E) c_tau = is_E ? 0.0f : c_tau; }
		r0 = fmaf(omw, f0, fmaf(w, feq0, Fin0*c_tau));
		#pragma unroll
		for(int k=0; k<9; k++) rp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], __builtin_elementwise_fma(splat2(w), feqp[k], Finp[k]*splat2(c_tau)));
	} else {
		r0 = fmaf(omw, f0, w*feq0);
		#pragma unroll
		for(int k=0; k<9; k++) rp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], splat2(w)*feqp[k]);
	}
	if constexpr(!E_BY_RATE) {
		if(wave_has_E) {
			r0 = is_E ? feq0 : r0;
			#pragma unroll
			for(int k=0; k<9; k++) { rp[k].x = is_E ? feqp[k].x : rp[k].x; rp[k].y = is_E ? feqp[k].y : rp[k].y; }
		}
	}
	f0 = r0;
	#pragma unroll
	for(int k=0; k<9; k++) fp[k] = rp[k];
}

} // namespace luw
