// luw_device_cross.hpp -- the collision of the TWO cells of a pair-kernel lane AT ONCE: every quantity is a 64-bit register pair whose halves belong to
// cell x and cell x + 1, and every addition, product and fma is ONE packed instruction (v_pk_add / v_pk_mul / v_pk_fma_f32, two IEEE operations each)
// for both cells.  The kernels of luw_device.hpp collide a lane's cells one after the other and pack within a cell (the two directions of a pair), which
// leaves the moments, the stress tensor, the force assembly and the relaxation rate -- two thirds of the arithmetic -- as one instruction per value and
// cell.  Across the cells everything packs; what stays per half are the operations the hardware has no packed form of: v_rcp / v_sqrt, min / max,
// selects, and the integer work of the FP16C codec.
//
// Arithmetic contract: each half performs exactly the operation sequence of collide_cell_pk (luw_device.hpp) on its own cell -- same order, same
// roundings (-ffp-contract=off, explicit fma only) -- so the results are bit-identical to it and to the CPU oracle (tests/test_gpu_parity.py and the rest
// of the bit-for-bit suite run through this path).  What is computed follows FX/kernel.cpp:1016-1055 (f_eq), :1075-1100 (moments), :1103-1113 (Guo),
// :1516-1623 (forces), :1686-1748 (collision).
// Included by luw_core.hip behind luw_kernels_common.hpp (static_for_pairs), inside `using namespace luw`.
#pragma once

__device__ __forceinline__ f32x2 fma2(const f32x2 a, const f32x2 b, const f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 rcp2(const f32x2 a) { return f32x2{ __builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y) }; }
__device__ __forceinline__ f32x2 sel2(const bool c0, const bool c1, const f32x2 a, const f32x2 b) { return f32x2{ c0 ? a.x : b.x, c1 ? a.y : b.y }; }
__device__ __forceinline__ f32x2 clamp2(const f32x2 a) { return f32x2{ clampf(a.x, -DEF_C, DEF_C), clampf(a.y, -DEF_C, DEF_C) }; }

// recip_prepare / div_by / sqrt_in_range of luw_device.hpp on both halves
struct Recip2 { f32x2 d, r; };
__device__ __forceinline__ Recip2 recip_prepare2(const f32x2 d) {
	const f32x2 r0 = rcp2(d);
	const f32x2 e = fma2(-d, r0, splat2(1.0f));
	return Recip2{ d, fma2(e, r0, r0) };
}
__device__ __forceinline__ f32x2 div_by2(const f32x2 n, const Recip2 R) {
	f32x2 q = n*R.r;
	q = fma2(fma2(-R.d, q, n), R.r, q);
	return fma2(fma2(-R.d, q, n), R.r, q);
}
__device__ __forceinline__ f32x2 sqrt_in_range2(const f32x2 x) {
	const f32x2 s = { __builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y) };
	const f32x2 down = { __uint_as_float(__float_as_uint(s.x)-1u), __uint_as_float(__float_as_uint(s.y)-1u) };
	const f32x2 up = { __uint_as_float(__float_as_uint(s.x)+1u), __uint_as_float(__float_as_uint(s.y)+1u) };
	const f32x2 r_down = fma2(-down, s, x), r_up = fma2(-up, s, x);
	const f32x2 t = sel2(r_down.x<=0.0f, r_down.y<=0.0f, down, s);
	return sel2(r_up.x>0.0f, r_up.y>0.0f, up, t);
}

// c_I . (a, b, c), I odd, on both halves (cdot of luw_device.hpp)
template<int I> __device__ __forceinline__ f32x2 cdot2(const f32x2 a, const f32x2 b, const f32x2 c) {
	if constexpr(I==1) return a; else if constexpr(I==3) return b; else if constexpr(I==5) return c;
	else if constexpr(I==7) return a+b; else if constexpr(I==9) return a+c; else if constexpr(I==11) return b+c;
	else if constexpr(I==13) return a-b; else if constexpr(I==15) return a-c; else return b-c;   // 17
}

// zone references and own fields of both cells, as pairs (fetch_force_refs fills one ForceRefs per cell)
struct ForceRefs2 { f32x2 tu[3], wb, su[3], sg; bool zn[2], zs[2]; };
__device__ __forceinline__ ForceRefs2 pair_refs(const ForceRefs& a, const ForceRefs& b) {
	ForceRefs2 r;
	#pragma unroll
	for(int c=0; c<3; c++) { r.tu[c] = f32x2{ a.tu[c], b.tu[c] }; r.su[c] = f32x2{ a.su[c], b.su[c] }; }
	r.wb = f32x2{ a.wb, b.wb }; r.sg = f32x2{ a.sg, b.sg };
	r.zn[0] = a.zn; r.zn[1] = b.zn; r.zs[0] = a.zs; r.zs[1] = b.zs;
	return r;
}

// Both cells of a lane.  f[19]: the streamed-in populations (halves: cell x, cell x + 1), TYPE_E cells decoded as zeros by the caller (relaxed with
// w = 1 and c_tau = 0 they come out as f_eq bit for bit: collide_cell_pk, E_BY_RATE).  live[c]: the cell is collided; a cell that is not passes its
// populations through, pre-swapped for the Esoteric-Pull store (k_stream_collide_p).  is_E[c] is only set for live cells.
// FORCE as in collide_cell_pk.  refs (FORCE ANY): zone references fetched ahead; own: rho / u of TYPE_E cells fetched ahead (wb, tu).
// out: post-collision populations in f, rho / u (after the half-force shift and the clamp) in rhon, ux, uy, uz.
template<int FORCE> __device__ __forceinline__ void collide_cross(const KParams& p, const uint32_t n, const bool live[2], const bool is_E[2], const bool may_force,
		f32x2* f, const float* __restrict__ F, f32x2& rhon, f32x2& ux, f32x2& uy, f32x2& uz, const ForceRefs2* refs, const ForceRefs2& own) {
	const bool wave_has_E = __ballot(is_E[0]||is_E[1])!=0ull;
	const bool wave_has_idle = __ballot(!live[0]||!live[1])!=0ull;
	// ---- moments, FX/kernel.cpp:1075-1100 (moment_sums: the same chains)
	f32x2 rho_m, mx, my, mz;
	{
		f32x2 r = f[0];
		#pragma unroll
		for(int i=1; i<19; i++) r = r+f[i];
		rho_m = r+splat2(1.0f);
		mx = f[ 1]-f[ 2]+f[ 7]-f[ 8]+f[ 9]-f[10]+f[13]-f[14]+f[15]-f[16];
		my = f[ 3]-f[ 4]+f[ 7]-f[ 8]+f[11]-f[12]+f[14]-f[13]+f[17]-f[18];
		mz = f[ 5]-f[ 6]+f[ 9]-f[10]+f[11]-f[12]+f[16]-f[15]+f[18]-f[17];
	}
	rhon = rho_m;
	Recip2 R = recip_prepare2(rho_m);
	ux = div_by2(mx, R); uy = div_by2(my, R); uz = div_by2(mz, R);
	// a density outside the plain range (density_is_ordinary) redoes its quotients with the library's division: rarely taken, per half
	bool odd[2] = { live[0]&&!density_is_ordinary(rho_m.x), live[1]&&!density_is_ordinary(rho_m.y) };
	if(odd[0]) { asm volatile(""); ux.x = mx.x/rho_m.x; uy.x = my.x/rho_m.x; uz.x = mz.x/rho_m.x; }
	if(odd[1]) { asm volatile(""); ux.y = mx.y/rho_m.y; uy.y = my.y/rho_m.y; uz.y = mz.y/rho_m.y; }
	if(wave_has_E) { // TYPE_E cells: rho and u are the stored fields (FX/kernel.cpp:1503-1515), fetched ahead by the caller
		rhon = sel2(is_E[0], is_E[1], own.wb, rhon);
		ux = sel2(is_E[0], is_E[1], own.tu[0], ux); uy = sel2(is_E[0], is_E[1], own.tu[1], uy); uz = sel2(is_E[0], is_E[1], own.tu[2], uz);
		R = recip_prepare2(rhon);
		odd[0] = live[0]&&!density_is_ordinary(rhon.x); odd[1] = live[1]&&!density_is_ordinary(rhon.y);
	}
	const bool forced = FORCE==PAIR_FORCE_UNIFORM || (FORCE==PAIR_FORCE_ANY && may_force);
	f32x2 fx = splat2(0.0f), fy = fx, fz = fx, uF = fx;
	if(forced) {
		// ---- force assembly, FX/kernel.cpp:1516-1623 (assemble_force)
		fx = splat2(p.fx); fy = splat2(p.fy); fz = splat2(p.fz);
		if(p.coriolis) {
			const f32x2 m2r = splat2(-2.0f)*rhon;
			fx = fx+m2r*(splat2(p.omy)*uz-splat2(p.omz)*uy);
			fy = fy+m2r*(splat2(p.omz)*ux-splat2(p.omx)*uz);
			fz = fz+m2r*(splat2(p.omx)*uy-splat2(p.omy)*ux);
		}
		if constexpr(FORCE==PAIR_FORCE_ANY) {
			if(refs) {
				if(__ballot(refs->zn[0]||refs->zn[1])!=0ull) { // buffer nudging towards the nearest owned face
					const f32x2 wt = refs->wb*splat2(p.buffer_inv_tau);
					const f32x2 ax = wt*(refs->tu[0]-ux), ay = wt*(refs->tu[1]-uy);
					fx = sel2(refs->zn[0], refs->zn[1], fx+rhon*ax, fx);
					fy = sel2(refs->zn[0], refs->zn[1], fy+rhon*ay, fy);
					// (nudge_vertical off: a_z = 0 and f_z + rho 0 = f_z, -0 aside: a sum with +0 never is -0 unless f_z is, and then every term below is a zero)
					if(p.nudge_vertical==1u) fz = sel2(refs->zn[0], refs->zn[1], fz+rhon*(wt*(refs->tu[2]-uz)), fz);
					else fz = sel2(refs->zn[0], refs->zn[1], fz+rhon*splat2(0.0f), fz);
				}
				if(__ballot(refs->zs[0]||refs->zs[1])!=0ull) { // top sponge
					const f32x2 rs = rhon*refs->sg;
					fx = sel2(refs->zs[0], refs->zs[1], fx+rs*(refs->su[0]-ux), fx);
					fy = sel2(refs->zs[0], refs->zs[1], fy+rs*(refs->su[1]-uy), fy);
					fz = sel2(refs->zs[0], refs->zs[1], fz+rs*(refs->su[2]-uz), fz);
				}
			}
			if(p.has_F) {
				const size_t Np = p.Np;
				fx = fx+f32x2{ F[n], F[n+1u] }; fy = fy+f32x2{ F[Np+n], F[Np+n+1u] }; fz = fz+f32x2{ F[2u*Np+n], F[2u*Np+n+1u] };
			}
		}
		// ---- half-force shift and clamp, FX/kernel.cpp:1686-1700
		f32x2 rho2 = div_by2(splat2(0.5f), R);
		if(odd[0]) { asm volatile(""); rho2.x = 0.5f/rhon.x; }
		if(odd[1]) { asm volatile(""); rho2.y = 0.5f/rhon.y; }
		ux = clamp2(fma2(fx, rho2, ux)); uy = clamp2(fma2(fy, rho2, uy)); uz = clamp2(fma2(fz, rho2, uz));
		uF = splat2(-0.33333334f)*fma2(ux, fx, fma2(uy, fy, uz*fz));
	} else {
		ux = clamp2(ux); uy = clamp2(uy); uz = clamp2(uz);
	}
	// ---- equilibrium, FX/kernel.cpp:1016-1055: feq[2k+1] = fma(rw, fma(0.5, fma(v, v, c3), v), rm1w), feq[2k+2] the same with -v
	const f32x2 rhom1 = rhon-splat2(1.0f);
	const f32x2 c3 = splat2(-3.0f)*(ux*ux+uy*uy+uz*uz);
	const f32x2 ux3 = ux*splat2(3.0f), uy3 = uy*splat2(3.0f), uz3 = uz*splat2(3.0f);
	f32x2 feq[19];
	feq[0] = splat2(DEF_W0)*fma2(rhon, splat2(0.5f)*c3, rhom1);
	const f32x2 rhos = splat2(DEF_WS)*rhon, rhoe = splat2(DEF_WE)*rhon, rhom1s = splat2(DEF_WS)*rhom1, rhom1e = splat2(DEF_WE)*rhom1;
	// the stress tensor of the non-equilibrium parts gathers while the equilibria are formed: each sum takes its terms in the order of smagorinsky_Q
	f32x2 Hxx, Hyy, Hzz, Hxy, Hxz, Hyz;
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		const f32x2 v = cdot2<i>(ux3, uy3, uz3);
		const f32x2 A = fma2(v, v, c3);
		const f32x2 rw = i<7 ? rhos : rhoe, rm = i<7 ? rhom1s : rhom1e;
		feq[i] = fma2(rw, fma2(splat2(0.5f), A, v), rm);
		feq[i+1] = fma2(rw, fma2(splat2(0.5f), A, -v), rm);
		if(p.subgrid) {
			const f32x2 na = f[i]-feq[i], nb = f[i+1]-feq[i+1];
			if constexpr(i==1) { Hxx = na; Hxx = Hxx+nb; }
			else if constexpr(i==3) { Hyy = na; Hyy = Hyy+nb; }
			else if constexpr(i==5) { Hzz = na; Hzz = Hzz+nb; }
			else if constexpr(i==7) { Hxx = Hxx+na; Hxx = Hxx+nb; Hyy = Hyy+na; Hyy = Hyy+nb; Hxy = na; Hxy = Hxy+nb; }
			else if constexpr(i==9) { Hxx = Hxx+na; Hxx = Hxx+nb; Hzz = Hzz+na; Hzz = Hzz+nb; Hxz = na; Hxz = Hxz+nb; }
			else if constexpr(i==11) { Hyy = Hyy+na; Hyy = Hyy+nb; Hzz = Hzz+na; Hzz = Hzz+nb; Hyz = na; Hyz = Hyz+nb; }
			else if constexpr(i==13) { Hxx = Hxx+na; Hxx = Hxx+nb; Hyy = Hyy+na; Hyy = Hyy+nb; Hxy = Hxy+(-na); Hxy = Hxy+(-nb); }
			else if constexpr(i==15) { Hxx = Hxx+na; Hxx = Hxx+nb; Hzz = Hzz+na; Hzz = Hzz+nb; Hxz = Hxz+(-na); Hxz = Hxz+(-nb); }
			else { Hyy = Hyy+na; Hyy = Hyy+nb; Hzz = Hzz+na; Hzz = Hzz+nb; Hyz = Hyz+(-na); Hyz = Hyz+(-nb); }
		}
	});
	// ---- Smagorinsky-Lilly relaxation rate, FX/kernel.cpp:1723-1737 (smagorinsky_rate_plain; odd densities: smagorinsky_rate_of_Q)
	f32x2 w = splat2(p.w);
	if(p.subgrid) {
		const f32x2 Q = Hxx*Hxx+Hyy*Hyy+Hzz*Hzz+splat2(2.0f)*(Hxy*Hxy+Hxz*Hxz+Hyz*Hyz);
		const f32x2 s = splat2(0.76421222f)*sqrt_in_range2(Q);
		const f32x2 tau = splat2(p.tau0)+sqrt_in_range2(splat2(p.tau0sq)+div_by2(s, R));
		w = div_by2(splat2(2.0f), recip_prepare2(tau));
		if(odd[0]) { asm volatile(""); w.x = smagorinsky_rate_of_Q(p, rhon.x, Q.x); }
		if(odd[1]) { asm volatile(""); w.y = smagorinsky_rate_of_Q(p, rhon.y, Q.y); }
	}
	f32x2 c_tau = fma2(w, splat2(-0.5f), splat2(1.0f));
	if(wave_has_E) { w = sel2(is_E[0], is_E[1], splat2(1.0f), w); c_tau = sel2(is_E[0], is_E[1], splat2(0.0f), c_tau); }
	const f32x2 omw = splat2(1.0f)-w;
	// ---- relaxation (+ Guo terms, FX/kernel.cpp:1103-1113), FX/kernel.cpp:1739-1748
	f32x2 r[19];
	if(forced) {
		r[0] = fma2(omw, f[0], fma2(w, feq[0], (splat2(9.0f*DEF_W0)*uF)*c_tau));
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			constexpr float w9 = 9.0f*(i<7 ? DEF_WS : DEF_WE);
			const f32x2 cF = cdot2<i>(fx, fy, fz), cu = cdot2<i>(ux, uy, uz);
			const f32x2 Fa = splat2(w9)*fma2(cF, cu+splat2(0.33333334f), uF);
			const f32x2 Fb = splat2(w9)*fma2(-cF, (-cu)+splat2(0.33333334f), uF);
			r[i] = fma2(omw, f[i], fma2(w, feq[i], Fa*c_tau));
			r[i+1] = fma2(omw, f[i+1], fma2(w, feq[i+1], Fb*c_tau));
		});
	} else {
		#pragma unroll
		for(int i=0; i<19; i++) r[i] = fma2(omw, f[i], w*feq[i]);
	}
	if(wave_has_idle) { // a cell that is not collided hands its populations back, each in its partner's place
		r[0] = sel2(live[0], live[1], r[0], f[0]);
		#pragma unroll
		for(int i=1; i<19; i+=2) { const f32x2 a = f[i], b = f[i+1]; r[i] = sel2(live[0], live[1], r[i], b); r[i+1] = sel2(live[0], live[1], r[i+1], a); }
	}
	#pragma unroll
	for(int i=0; i<19; i++) f[i] = r[i];
}
