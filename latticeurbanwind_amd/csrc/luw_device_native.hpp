// luw_device_native.hpp -- the collision of one cell in the hardware's own arithmetic (LUW_OPT_NATIVE_ARITH): the FP16C kernels' default in the deck driver and
// the bench.  Included by luw_device.hpp, behind luw_device_pair.hpp (f32x2, the force modes).
#pragma once

namespace luw {

// ---------------------------------------------------------------- the same collision in NATIVE arithmetic (LUW_OPT_NATIVE_ARITH)
// The contract above (every operation rounded like the CPU restatement's) is this project's, not the reference's: the reference kernel is compiled by the
// OpenCL driver with -cl-mad-enable and native division / square root (FX/opencl.hpp:305, FX/kernel.cpp:1088-1100,1735) and is not bit-defined.  Where the
// VALU is the limit (FP16C pair kernel) the same formulas can run with the hardware's own operations and the sums in any order:
//   * one v_rcp_f32 of the density serves u = m / rho, F / (2 rho) and sqrt(Q) / rho; v_sqrt_f32 and v_rcp_f32 for the Smagorinsky rate
//     (w = 1 / (tau0 / 2 + sqrt(tau0^2 + 0.76421222 sqrt(Q) / rho) / 2));
//   * moments from the nine pair sums s_k = f[2k+1] + f[2k+2] and differences d_k = f[2k+1] - f[2k+2] (c_(2k+2) = -c_(2k+1)): rho = f0 + sum s_k + 1,
//     mx = d0 + d3 + d4 + d6 + d7, ...; 40 additions in short trees instead of 46 in chains;
//   * the stress tensor from the NON-EQUILIBRIUM PAIR SUMS alone: c c is the same for both directions of a pair, so Pi = sum_k (c c)_k (n_(2k+1) + n_(2k+2))
//     and n_(2k+1) + n_(2k+2) = s_k - (rho w_k A_k + 2 (rho - 1) w_k) with A_k = (3 c.u)^2 - 3 u^2 of the equilibrium (its +-3 c.u parts cancel):
//     18 + 15 operations instead of 18 + 36, and the nineteen equilibria are formed only once the rate is known (no f_eq registers across Q);
//   * Guo terms with the constants folded: c_tau w9_i [(c_i.F)(c_i.u + 1/3) - u.F / 3] = fma(c_i.H, 3 c_i.u + 1, uH) with H = c_tau w9_i F / 3, the
//     3 c.u of the equilibrium reused, added inside the relaxation's own fma;
//   * fused multiply-adds wherever a product feeds a sum (written out: see below), v_med3 for the +-c clamp, -2 omega from the host.
// TYPE_E lanes in every FORCE mode: decoded as f = 0 by the caller, relaxed with w = 1 and c_tau = 0 -> f_eq (collide_cell_pk, E_BY_RATE).
// Values differ from the exact kernels' in the last bits of each operation; with FP16C storage (2^-12 relative per stored value) those differences
// surface as different roundings of single populations, exactly like the reference's own arithmetic against the restatement's.
// The exact kernels stay the default and the anchor of every bit-for-bit test; tests/test_gpu_native_arith.py holds the tolerance gates of this one.
#ifndef LUW_NATIVE_RCP_NEWTON
#define LUW_NATIVE_RCP_NEWTON 0   /* 1: one Newton step behind the density's v_rcp_f32 (A/B: the u-RMSE against the oracle does not change) */
#endif
// RAW (pair kernel without the thermal lattice): the populations arrive and leave SCALED by 2^-112 -- the bit pattern the codec's shift-and-mask produces
// and consumes -- so that neither the decode nor the encode multiplies: every place the populations enter is linear in them, and the power of two moves
// into a factor that exists anyway (rho = fma(sum, 2^112, 1); u = m (2^112 / rho); n_k = fma(s_k, 2^112, -eq_k); out = (1 - w) f + 2^-112 (w f_eq + F)).
// A power of two commutes with every rounding as long as nothing underflows: the scaled populations are multiples of 2^-137 (the float denormal
// quantum is 2^-149), their sums round like the unscaled ones; the outputs are rounded to 2^-149 = 2^-37 in lattice units where an FP16C code step is
// 2^-25 at least.  The encode is then the reference's own formula on the float's bits (add 0x800, drop 12 bits: FX/kernel.cpp:870-875) under the default
// rounding mode -- no switch to round-toward-zero, which the exact kernels need for their 2^-112 product alone.
__device__ __forceinline__ f32x2 sum_and_negated_difference(const f32x2 a) { // { x + y, y - x } in one packed addition
	f32x2 r;
	asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
	return r;
}
// Every fused multiply-add is written out (the translation unit is compiled with -ffp-contract=off): the operation sequence is the same in every
// instantiation and kernel that inlines this function, so a cell gets the same bits whether its row runs in the pair or in the one-cell kernel, in a whole
// lattice or in a domain of a decomposed one (tests/test_gpu_native_arith.py::test_native_arithmetic_gives_a_cell_the_same_values_in_either_kernel).
// on_fields(rho, ux, uy, uz): called once the cell's density and (force-shifted, clamped) velocity are final, i.e. BEFORE the relaxation -- the caller stores
// the fields there (last step of a run) instead of keeping four registers alive through the relaxation loop.
struct NoFieldSink { __device__ __forceinline__ void operator()(float, float, float, float) const {} };
template<int FORCE=PAIR_FORCE_ANY, bool RAW=false, typename FieldSink=NoFieldSink> __device__ __forceinline__ void collide_cell_pk_native(const KParams& p,
	const uint32_t n, const uint8_t flagsn, const bool may_force, float& f0, f32x2* fp, const float* __restrict__ rho, const float* __restrict__ u,
	const float* __restrict__ F, float& rhon, float& uxn, float& uyn, float& uzn, float* u_before_force = nullptr, const ForceRefs* refs = nullptr,
	const ForceRefs* own = nullptr, const FieldSink on_fields = FieldSink{}) {
	constexpr float UP = RAW ? 0x1p+112f : 1.0f, DOWN = RAW ? 0x1p-112f : 1.0f;
	const bool is_E = (flagsn&TYPE_BO)==TYPE_E;
	const bool wave_has_E = __ballot(is_E)!=0ull;
	// pair sums s_k = f[2k+1] + f[2k+2] (.x) and NEGATED differences -d_k = f[2k+2] - f[2k+1] (.y), one packed addition per pair; the three pairs of
	// pairs that share an axis (k = 3 / 6: +-x +-y, 4 / 7: +-x +-z, 5 / 8: +-y +-z) summed as pairs again: { s + s', -(d + d') }
	f32x2 sd[9];
	float nmx, nmy, nmz;                                           // -m = -sum c f
	{
		#pragma unroll
		for(int k=0; k<9; k++) sd[k] = sum_and_negated_difference(fp[k]);
		const f32x2 p36 = sd[3]+sd[6], p47 = sd[4]+sd[7], p58 = sd[5]+sd[8];
		const float sum = ((f0+sd[0].x)+(sd[1].x+sd[2].x))+((p36.x+p47.x)+p58.x);
		rhon = fmaf(sum, UP, 1.0f);
		nmx = (sd[0].y+p36.y)+p47.y;
		nmy = (sd[1].y+(sd[3].y-sd[6].y))+p58.y;
		nmz = (sd[2].y+(sd[4].y-sd[7].y))+(sd[5].y-sd[8].y);
	}
	if(wave_has_E) { if(is_E) rhon = own ? own->wb : rho[n]; }
	float r = __builtin_amdgcn_rcpf(rhon);
	if constexpr(LUW_NATIVE_RCP_NEWTON!=0) r = fmaf(fmaf(-rhon, r, 1.0f), r, r);
	{ const float nr = -UP*r; uxn = nmx*nr; uyn = nmy*nr; uzn = nmz*nr; }
	if(wave_has_E) {
		if(is_E) {
			if(own) { uxn = own->tu[0]; uyn = own->tu[1]; uzn = own->tu[2]; }
			else { uxn = u[n]; uyn = u[(size_t)p.Np+n]; uzn = u[2ull*p.Np+n]; }
		}
	}
	if(u_before_force) { u_before_force[0] = uxn; u_before_force[1] = uyn; u_before_force[2] = uzn; }
	const bool forced = FORCE==PAIR_FORCE_UNIFORM || (FORCE==PAIR_FORCE_ANY && may_force);
	float fxn = 0.0f, fyn = 0.0f, fzn = 0.0f;
	if(forced) {
		fxn = p.fx; fyn = p.fy; fzn = p.fz;
		if(p.coriolis) { // -2 rho omega x u
			fxn = fmaf(rhon, fmaf(p.m2omy, uzn, -(p.m2omz*uyn)), fxn);
			fyn = fmaf(rhon, fmaf(p.m2omz, uxn, -(p.m2omx*uzn)), fyn);
			fzn = fmaf(rhon, fmaf(p.m2omx, uyn, -(p.m2omy*uxn)), fzn);
		}
		if constexpr(FORCE==PAIR_FORCE_ANY) {
			if(refs) { // zone references fetched ahead (fetch_force_refs): nudging towards the nearest owned face, top sponge
				if(refs->zn) {
					const float wr = (refs->wb*p.buffer_inv_tau)*rhon;
					fxn = fmaf(wr, refs->tu[0]-uxn, fxn);
					fyn = fmaf(wr, refs->tu[1]-uyn, fyn);
					if(p.nudge_vertical==1u) fzn = fmaf(wr, refs->tu[2]-uzn, fzn);
				}
				if(refs->zs) {
					const float sr = refs->sg*rhon;
					fxn = fmaf(sr, refs->su[0]-uxn, fxn);
					fyn = fmaf(sr, refs->su[1]-uyn, fyn);
					fzn = fmaf(sr, refs->su[2]-uzn, fzn);
				}
			}
			if(p.has_F) { fxn += F[n]; fyn += F[(size_t)p.Np+n]; fzn += F[2ull*p.Np+n]; }
		}
		const float rho2 = 0.5f*r;
		uxn = fmaf(fxn, rho2, uxn); uyn = fmaf(fyn, rho2, uyn); uzn = fmaf(fzn, rho2, uzn);
	}
	uxn = __builtin_amdgcn_fmed3f(uxn, -DEF_C, DEF_C);
	uyn = __builtin_amdgcn_fmed3f(uyn, -DEF_C, DEF_C);
	uzn = __builtin_amdgcn_fmed3f(uzn, -DEF_C, DEF_C);
	on_fields(rhon, uxn, uyn, uzn);
	// equilibrium ingredients (FX/kernel.cpp:1016-1055): f_eq(2k+1 / 2k+2) = rho w_k (A_k / 2 +- v_k) + (rho - 1) w_k, v_k = 3 c_k.u, A_k = v_k^2 - 3 u^2
	const float c3 = -3.0f*fmaf(uzn, uzn, fmaf(uyn, uyn, uxn*uxn));
	const float ux3 = 3.0f*uxn, uy3 = 3.0f*uyn, uz3 = 3.0f*uzn;
	const float v[9] = { ux3, uy3, uz3, ux3+uy3, ux3+uz3, uy3+uz3, ux3-uy3, ux3-uz3, uy3-uz3 };
	const float rhom1 = rhon-1.0f;
	const float rhos = DEF_WS*rhon, rhoe = DEF_WE*rhon, rhom1s = DEF_WS*rhom1, rhom1e = DEF_WE*rhom1;
	float A[9];
	#pragma unroll
	for(int k=0; k<9; k++) A[k] = fmaf(v[k], v[k], c3);
	float w = p.w;
	// Smagorinsky-Lilly, FX/kernel.cpp:1723-1737, from the non-equilibrium pair sums n_(2k+1) + n_(2k+2) = s_k - (rho w_k A_k + 2 (rho - 1) w_k)
	if(p.subgrid) {
		const float rm2s = 2.0f*rhom1s, rm2e = 2.0f*rhom1e;
		float sn[9];
		#pragma unroll
		for(int k=0; k<9; k++) sn[k] = fmaf(sd[k].x, UP, -fmaf(k<3 ? rhos : rhoe, A[k], k<3 ? rm2s : rm2e));
		const float Hxx = (sn[0]+(sn[3]+sn[4]))+(sn[6]+sn[7]), Hyy = (sn[1]+(sn[3]+sn[5]))+(sn[6]+sn[8]), Hzz = (sn[2]+(sn[4]+sn[5]))+(sn[7]+sn[8]);
		const float Hxy = sn[3]-sn[6], Hxz = sn[4]-sn[7], Hyz = sn[5]-sn[8];
		const float Q = fmaf(2.0f, fmaf(Hyz, Hyz, fmaf(Hxz, Hxz, Hxy*Hxy)), fmaf(Hzz, Hzz, fmaf(Hyy, Hyy, Hxx*Hxx)));
		const float sq = 0.76421222f*__builtin_amdgcn_sqrtf(Q);
		w = __builtin_amdgcn_rcpf(fmaf(0.5f, __builtin_amdgcn_sqrtf(fmaf(sq, r, p.tau0sq)), p.half_tau0));
	}
	float c_tau = fmaf(-0.5f, w, 1.0f);
	if(wave_has_E) { w = is_E ? 1.0f : w; c_tau = is_E ? 0.0f : c_tau; }
	const float omw = 1.0f-w;
	// relaxation with the rate folded into the equilibrium's coefficients: w f_eq(+-) = W (A / 2 +- v) + M, W = w rho w_k, M = w (rho - 1) w_k (RAW: times
	// 2^-112)
	const float wd = w*DOWN;
	const float Ws = wd*rhos, We = wd*rhoe, Ms = wd*rhom1s, Me = wd*rhom1e;
	const float weq0 = wd*(DEF_W0*fmaf(rhon, 0.5f*c3, rhom1));      // w f_eq of the rest population
	if(forced) {
		// c_tau Fin_i = fma(+-c.H, +-v + 1, uH): H = c_tau w9 F / 3 (w9 = 1/2 axis, 1/4 diagonal), uH = -c_tau w9 (u.F) / 3; Fin_0 = -c_tau (u.F); the constant
		// part uH joins M
		const float cs = (c_tau*DOWN)*0.16666667f;
		const float hx = cs*fxn, hy = cs*fyn, hz = cs*fzn;
		const float dots = cs*fmaf(uzn, fzn, fmaf(uyn, fyn, uxn*fxn));    // = -uH of the axis pairs
		const float ex = 0.5f*hx, ey = 0.5f*hy, ez = 0.5f*hz;
		const float Mds = Ms-dots, Mde = fmaf(-0.5f, dots, Me);
		f0 = fmaf(omw, f0, fmaf(-6.0f, dots, weq0));
		// uniform-force instantiation (everything in registers, 96 of them for 5 waves): the six diagonal 3 c.u are formed AGAIN here instead of living from
		// the
		// equilibrium ingredients on -- six additions for six registers, without which the kernel spills (the empty asm keeps the compiler from reusing them)
		float vx = ux3, vy = uy3, vz = uz3;
		if constexpr(FORCE==PAIR_FORCE_UNIFORM) asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz));
		#pragma unroll
		for(int k=0; k<9; k++) {
			// c_k.H formed where it is used (six values live instead of nine)
			const float cH = k==0 ? hx : k==1 ? hy : k==2 ? hz : k==3 ? ex+ey : k==4 ? ex+ez : k==5 ? ey+ez : k==6 ? ex-ey : k==7 ? ex-ez : ey-ez;
			const float vk = FORCE!=PAIR_FORCE_UNIFORM ? v[k]
				: k==0 ? vx : k==1 ? vy : k==2 ? vz : k==3 ? vx+vy : k==4 ? vx+vz : k==5 ? vy+vz : k==6 ? vx-vy : k==7 ? vx-vz : vy-vz;
			const f32x2 in = __builtin_elementwise_fma(splat2(0.5f), splat2(A[k]), pm2(vk));
			const f32x2 fin = __builtin_elementwise_fma(pm2(cH), pm2(vk)+splat2(1.0f), splat2(k<3 ? Mds : Mde));
			fp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], __builtin_elementwise_fma(splat2(k<3 ? Ws : We), in, fin));
		}
	} else {
		f0 = fmaf(omw, f0, weq0);
		#pragma unroll
		for(int k=0; k<9; k++) {
			const f32x2 in = __builtin_elementwise_fma(splat2(0.5f), splat2(A[k]), pm2(v[k]));
			fp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], __builtin_elementwise_fma(splat2(k<3 ? Ws : We), in, splat2(k<3 ? Ms : Me)));
		}
	}
}

} // namespace luw
