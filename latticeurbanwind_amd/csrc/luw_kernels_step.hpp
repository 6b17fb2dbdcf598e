// luw_kernels_step.hpp -- the product collide-stream kernel with one cell per lane, k_stream_collide_s (FP32; FP16C where the pair kernel does not
// apply), and the addressing both step kernels share; the FP16C kernel with two cells per lane, k_stream_collide_p, is luw_kernels_pair.hpp
// Device code of libluw_core.so; included by luw_core.hip only (after luw_device.hpp, inside `using namespace luw`).
#pragma once

// ---------------------------------------------------------------- scalar kernel: 1 cell per lane
// Addressing: every DDF access is (uniform plane base in SGPRs) + (32-bit byte offset in one VGPR), the
// global_load/store "saddr" form; the 10 byte offsets (own cell + 9 neighbours) stay live from the loads to the
// stores instead of 38 64-bit addresses.  Byte offsets fit 32 bits because Np*sizeof(T) <= 2^32 (checked on the host).
template<bool NT, typename T> __device__ __forceinline__ T ldo(const T* plane, const uint32_t byte_off) {
	return ldg<NT>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(plane)+byte_off));
}
template<bool NT, typename T> __device__ __forceinline__ void sto(T* plane, const uint32_t byte_off, const T v) {
	stg<NT>(reinterpret_cast<T*>(reinterpret_cast<char*>(plane)+byte_off), v);
}
// Addressing.  y and z of a block are uniform (they come from blockIdx), so the start of every row a cell touches -- its own
// and those of its y/z neighbours (periodic wrap, FX/kernel.cpp:920-958) -- is a uniform 64-bit ELEMENT offset that the scalar
// unit adds to the plane base; the only per-lane parts are the byte offsets of x and of x+1 within a row.  Two offset VGPRs
// instead of ten, and no limit on the plane size (a per-plane 32-bit byte offset would stop at 2^30 FP32 cells).
struct RowOff { size_t r00; int64_t p0, _0p, pp, m0, _0m, pm; };   // own row (y,z); steps to the rows (y+1,z), (y,z+1), (y+1,z+1), (y-1,z), (y,z-1), (y+1,z-1)
struct LaneOff { uint32_t x, xp; };                            // byte offsets of x and of x+1 (wrapped) within a row
__device__ __forceinline__ RowOff row_offsets(const KParams& p, const uint32_t y, const uint32_t z) {
	// the neighbour rows as the own row plus a step that is +-(one row / one z plane) or, at the periodic wrap, the way back
	// across the lattice: selects and additions on the scalar unit, a single 64-bit product for the own row
	const uint32_t A = p.Px*p.Ny;                                  // cells of a z plane (< 2^32: it divides Np)
	const int64_t sy = (int64_t)p.Px, wy = (int64_t)(p.Px*(p.Ny-1u)), sz = (int64_t)A, wz = (int64_t)((uint64_t)A*(p.Nz-1u));
	const int64_t dyp = y+1u==p.Ny ? -wy : sy, dym = y==0u ? wy : -sy;
	const int64_t dzp = z+1u==p.Nz ? -wz : sz, dzm = z==0u ? wz : -sz;
	RowOff r;
	r.r00 = (size_t)z*A+(size_t)(y*p.Px);
	r.p0 = dyp; r._0p = dzp; r.pp = dyp+dzp; r.m0 = dym; r._0m = dzm; r.pm = dyp+dzm;
	return r;
}
template<typename T> __device__ __forceinline__ LaneOff lane_offsets(const KParams& p, const uint32_t x) {
	LaneOff o;
	o.x = x*(uint32_t)sizeof(T); o.xp = (x+1u==p.Nx ? 0u : x+1u)*(uint32_t)sizeof(T);
	return o;
}
// the +c_I neighbour (I odd) lives nrow<I>() cells after the own row's start, at lane offset nlane<I>()
template<int I> __device__ __forceinline__ int64_t nrow(const RowOff& r) {
	if constexpr(I==1) return 0; else if constexpr(I==3) return r.p0; else if constexpr(I==5) return r._0p;
	else if constexpr(I==7) return r.p0; else if constexpr(I==9) return r._0p; else if constexpr(I==11) return r.pp;
	else if constexpr(I==13) return r.m0; else if constexpr(I==15) return r._0m; else return r.pm;
}
template<int I> __device__ __forceinline__ uint32_t nlane(const LaneOff& o) {
	if constexpr(I==1||I==7||I==9||I==13||I==15) return o.xp; else return o.x;
}

// The same addresses as ONE 32-bit byte offset per neighbour within a plane (own cell + 9 neighbours in VGPRs, plane bases
// without the row): needs Np*sizeof(T) <= 2^32, and is what the FP32 kernel uses when that holds -- its shorter scalar
// prologue lets a wave issue its loads earlier, worth 1 % at 512^3 on the HBM-bound kernel; the VALU-bound FP16C kernels and
// larger lattices take the row form above.
struct NbrOff { uint32_t n, j1, j3, j5, j7, j9, j11, j13, j15, j17; };
template<typename T> __device__ __forceinline__ NbrOff neighbor_offsets(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z) {
	const uint32_t xp = x+1u==p.Nx ? 0u : x+1u;
	const uint32_t y0 = y*p.Px, yp = (y+1u==p.Ny ? 0u : y+1u)*p.Px, ym = (y==0u ? p.Ny-1u : y-1u)*p.Px;
	const uint32_t A = p.Px*p.Ny;
	const uint32_t z0 = z*A, zp = (z+1u==p.Nz ? 0u : z+1u)*A, zm = (z==0u ? p.Nz-1u : z-1u)*A;
	constexpr uint32_t B = (uint32_t)sizeof(T);
	NbrOff o;
	o.n = (x+y0+z0)*B;
	o.j1 = (xp+y0+z0)*B; o.j3 = (x+yp+z0)*B; o.j5 = (x+y0+zp)*B;
	o.j7 = (xp+yp+z0)*B; o.j9 = (xp+y0+zp)*B; o.j11 = (x+yp+zp)*B;
	o.j13 = (xp+ym+z0)*B; o.j15 = (xp+y0+zm)*B; o.j17 = (x+yp+zm)*B;
	return o;
}
template<int I> __device__ __forceinline__ uint32_t nbr(const NbrOff& o) {
	if constexpr(I==1) return o.j1; else if constexpr(I==3) return o.j3; else if constexpr(I==5) return o.j5;
	else if constexpr(I==7) return o.j7; else if constexpr(I==9) return o.j9; else if constexpr(I==11) return o.j11;
	else if constexpr(I==13) return o.j13; else if constexpr(I==15) return o.j15; else return o.j17;
}
// one interface over both forms: plane pointer adjustment (uniform) + lane byte offset of the own cell / the +c_I neighbour
template<typename T, bool FLAT> struct CellAddr;
template<typename T> struct CellAddr<T, false> {
	RowOff rb; LaneOff o; uint32_t n;
	// returns what the caller adds to its lattice pointer: from then on it points at the own row (uniform)
	__device__ __forceinline__ size_t init(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z, const bool noshift) {
		rb = row_offsets(p, y, z); o = lane_offsets<T>(p, x);
		if(noshift) o.xp = o.x;
		n = x+(uint32_t)rb.r00;
		return rb.r00;
	}
	template<int I> __device__ __forceinline__ int64_t row() const { return nrow<I>(rb); }
	template<int I> __device__ __forceinline__ uint32_t lane() const { return nlane<I>(o); }
	__device__ __forceinline__ uint32_t own() const { return o.x; }
	__device__ __forceinline__ uint32_t jx() const { return o.xp/(uint32_t)sizeof(T)+(uint32_t)rb.r00; }
	__device__ __forceinline__ uint32_t jy() const { return n+(uint32_t)rb.p0; }
	__device__ __forceinline__ uint32_t jz() const { return n+(uint32_t)rb._0p; }
	__device__ __forceinline__ void redefine() { asm volatile("" : "+v"(o.x), "+v"(o.xp)); }
};
template<typename T> struct CellAddr<T, true> {
	NbrOff o; uint32_t n;
	__device__ __forceinline__ size_t init(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z, const bool noshift) {
		o = neighbor_offsets<T>(p, x, y, z);
		if(noshift) {
			o.j1 = o.n;
			o.j7 = o.j3;
			o.j9 = o.j5;
			o.j13 = (x+(y==0u ? p.Ny-1u : y-1u)*p.Px+z*p.Px*p.Ny)*(uint32_t)sizeof(T);
			o.j15 = (x+y*p.Px+(z==0u ? p.Nz-1u : z-1u)*p.Px*p.Ny)*(uint32_t)sizeof(T);
		}
		n = o.n/(uint32_t)sizeof(T);
		return 0u;
	}
	template<int I> __device__ __forceinline__ int64_t row() const { return 0; }
	template<int I> __device__ __forceinline__ uint32_t lane() const { return nbr<I>(o); }
	__device__ __forceinline__ uint32_t own() const { return o.n; }
	__device__ __forceinline__ uint32_t jx() const { return o.j1/(uint32_t)sizeof(T); }
	__device__ __forceinline__ uint32_t jy() const { return o.j3/(uint32_t)sizeof(T); }
	__device__ __forceinline__ uint32_t jz() const { return o.j5/(uint32_t)sizeof(T); }
	__device__ __forceinline__ void redefine() {
		asm volatile("" : "+v"(o.n), "+v"(o.j1), "+v"(o.j3), "+v"(o.j5), "+v"(o.j7), "+v"(o.j9), "+v"(o.j11), "+v"(o.j13), "+v"(o.j15), "+v"(o.j17));
	}
};

// MODE 0 is the product kernel.  MODE 1 ("copy": no collision) and MODE 2 ("noshift": x+1 neighbours replaced by x) are
// measurement-only variants that isolate the memory system's share of the step; they do not compute physics.
// XFACE input (luw_set_x_face_inputs): element of the x-face receive buffers that holds what the halo cell (x halo, y + dy, z + dz) of this row's border cell
// received -- a = y + z Ny as in k_insert_fi<0>, periodic where y / z are not split (a split axis keeps y + dy inside its halo layers by itself)
__device__ __forceinline__ size_t xface_in_elem(const KParams& p, const uint32_t y, const uint32_t z, const int dy, const int dz) {
	uint32_t yy = y+(uint32_t)dy, zz = z+(uint32_t)dz;                   // unsigned wrap-around is undone by the two selects
	if(dy<0&&y==0u) yy = p.Ny-1u; else if(dy>0&&yy==p.Ny) yy = 0u;
	if(dz<0&&z==0u) zz = p.Nz-1u; else if(dz>0&&zz==p.Nz) zz = 0u;
	return (size_t)yy+(size_t)zz*p.Ny;
}
// NT: 0 default cache policy, 1 non-temporal everywhere, 2 non-temporal on the 14 aligned planes and default policy on
// the five x+1 planes, whose wave-edge lines are shared between neighbouring waves (product setting, measured best).
// Waves per SIMD: the FP32 kernel is HBM-bound and measurably better with at most 4 resident waves (3.40 vs 3.43 ms at 512^3,
// 6.68 vs 6.98 ms at 1024x1024x256: fewer concurrent row fronts, better DRAM page locality) even though its ~95 VGPRs would
// allow 5; the FP16C kernel is VALU-bound and takes all the waves its registers allow.
#ifndef LUW_MAXW_F32
#define LUW_MAXW_F32 4
#endif
// STATS: this step is a statistics sample (stats_welford, luw_kernels_common.hpp); product MODE 0 only.
// NATIVE (FP16C, LUW_OPT_NATIVE_ARITH): the pair kernel's native-arithmetic collision (collide_cell_pk_native) on this kernel's one cell per lane -- the rows
// too narrow or unaligned for the pair kernel, so that a native run is native everywhere.
// XFACE (x-split domains, luw_set_x_face_buffers): the cells of the first / last owned x column also put their five outgoing populations into the face
// buffers -- element (b A + a), a = y + z Ny, b as in FX/kernel.cpp:2223-2229: exactly what transfer_extract_fi (k_extract_fi, direction 0) would copy out
// of the lattice behind this kernel, one element per 128-byte line.  A cell of those columns that is not collided (TYPE_S / TYPE_G) forwards what its
// slots hold, like the extract kernel: the populations a fluid cell beyond the cut pushed into it come back that way (bounce-back across a cut).
// With xin_p / xin_m set (luw_set_x_face_inputs) the same cells take the five populations that ARRIVE through their face from the receive buffers of the last
// exchange instead of from the lattice -- what k_insert_fi<0> would have put, one element per 128-byte line, into the slots these very loads read: the first
// column's own slots A(1, 7, 13, 9, 15) (buffer "from -x", element of the halo cell the population left: y - c_y, z - c_z) and the last column's
// neighbour slots B of the same pairs in the halo column (buffer "from +x", element y + c_y, z + c_z).
template<typename T, int PARITY, int MODE=0, int NT=2, bool FLAT=false, bool STATS=false, bool NOFORCE=false, bool NATIVE=false, bool XFACE=false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((sizeof(T)==4 && LUW_MAXW_F32<4) ? LUW_MAXW_F32 : 4, sizeof(T)==4 ? LUW_MAXW_F32 : 8)))
void k_stream_collide_s(const KParams p, const Box b, const int xa, T* __restrict__ fi, float* __restrict__ rho, float* __restrict__ u,
		const uint8_t* __restrict__ flags, const float* __restrict__ F, const int write_fields, T* __restrict__ gi = nullptr, float* __restrict__ Tf = nullptr,
		const StatsArgs S = StatsArgs{}, T* __restrict__ xf_p = nullptr, T* __restrict__ xf_m = nullptr, const T* __restrict__ xin_p = nullptr,
			const T* __restrict__ xin_m = nullptr) {
	// lanes are laid over the row in 64-cell blocks aligned with the memory lines (xa = b.x0 rounded down to such a block
	// start), whatever the box: lanes left of b.x0 idle
	// (workgroups go round-robin to the 8 XCDs; remapping them so that each XCD owns one contiguous eighth of the box was
	// measured 3.5 % slower, 3.50 vs 3.39 ms at 512^3 -- no population is shared between workgroups, so there is nothing
	// for an XCD's L2 to reuse, and eight distant fronts cost DRAM page locality)
	uint32_t bix, biy;
	xcd_row_order(p, bix, biy);
	const int xi = xa+(int)(bix*blockDim.x+threadIdx.x);
	if(xi<(int)b.x0||xi>=(int)b.x1) return;
	const uint32_t x = (uint32_t)xi, y = b.y0+biy, z = b.z0+blockIdx.z;
	if(cell_is_halo(p, x, y, z)) return;
	CellAddr<T, FLAT> a;
	fi += a.init(p, x, y, z, MODE==2);
	const uint32_t n = a.n;
	const uint8_t flagsn = flags[n];
	const size_t Np = p.Np;
	// XFACE: the face buffers' element of this cell, and the five values (b = 0..4: populations 1, 7, 13, 9, 15 towards +x, their partners 2, 8, 14, 10, 16
	// towards -x) as value_of(i) hands them over
	[[maybe_unused]] auto xface_out = [&](auto value_of) {
		const size_t A = (size_t)p.Ny*p.Nz, e = (size_t)y+(size_t)z*p.Ny;
		if(x==p.Nx-2u) { xf_p[e] = value_of(1); xf_p[A+e] = value_of(7); xf_p[2u*A+e] = value_of(13); xf_p[3u*A+e] = value_of(9); xf_p[4u*A+e] = value_of(15); }
		if(x==1u) { xf_m[e] = value_of(2); xf_m[A+e] = value_of(8); xf_m[2u*A+e] = value_of(14); xf_m[3u*A+e] = value_of(10); xf_m[4u*A+e] = value_of(16); }
	};
	// XFACE input: put(k, value) for the five loads of this cell that the receive buffers replace, k = the index the fluid path loads them into (k odd: own
	// slot A(k); k even: slot B(k - 1) of the +c neighbour)
	[[maybe_unused]] auto xface_in = [&](auto put) {
		const size_t A = (size_t)p.Ny*p.Nz;
		if(x==1u&&xin_m) {
			put(1, xin_m[xface_in_elem(p, y, z, 0, 0)]); put(7, xin_m[A+xface_in_elem(p, y, z, -1, 0)]); put(13, xin_m[2u*A+xface_in_elem(p, y, z, 1, 0)]);
			put(9, xin_m[3u*A+xface_in_elem(p, y, z, 0, -1)]); put(15, xin_m[4u*A+xface_in_elem(p, y, z, 0, 1)]);
		}
		if(x==p.Nx-2u&&xin_p) {
			put(2, xin_p[xface_in_elem(p, y, z, 0, 0)]); put(8, xin_p[A+xface_in_elem(p, y, z, 1, 0)]); put(14, xin_p[2u*A+xface_in_elem(p, y, z, -1, 0)]);
			put(10, xin_p[3u*A+xface_in_elem(p, y, z, 0, 1)]); put(16, xin_p[4u*A+xface_in_elem(p, y, z, 0, -1)]);
		}
	};
	if((flagsn&TYPE_BO)==TYPE_S||(flagsn&TYPE_SU)==TYPE_G) {
		if constexpr(STATS) stats_hold_constant_cell(Np, S, n, rho, u);
		if constexpr(XFACE) { // forward what the slots hold: population i sits in slot B(i) of the +c_i neighbour, its partner i + 1 in slot A(i) of the cell
			if(x==p.Nx-2u||x==1u) {
				T held[19];
				static_for_pairs([&](auto ic) {
					constexpr int i = decltype(ic)::value;
					if constexpr(i==1||i==7||i==9||i==13||i==15) {
						held[i] = ldo<false>(fi+(size_t)slotB<PARITY>(i)*Np+a.template row<i>(), a.template lane<i>());
						held[i+1] = ldo<false>(fi+(size_t)slotA<PARITY>(i)*Np, a.own());
					}
				});
				if((x==1u) ? xin_m!=nullptr : xin_p!=nullptr) {
					// slot A(k) is held[k + 1], slot B(k - 1) is held[k - 1]; the lattice takes the values too (a pack kernel may read these slots later)
					xface_in([&](const int k, const T v) { held[(k&1) ? k+1 : k-1] = v; });
					static_for_pairs([&](auto ic) {
						constexpr int i = decltype(ic)::value;
						if constexpr(i==1||i==7||i==9||i==13||i==15) {
							if(x==1u) sto<false>(fi+(size_t)slotA<PARITY>(i)*Np, a.own(), held[i+1]);
							else sto<false>(fi+(size_t)slotB<PARITY>(i)*Np+a.template row<i>(), a.template lane<i>(), held[i]);
						}
					});
				}
				xface_out([&](const int i) { return held[i]; });
			}
		}
		return;
	}
	// RAW (native arithmetic without the thermal lattice): the populations stay scaled by 2^-112 from decode to encode, exactly as in the pair kernel
	// (collide_cell_pk_native<.., RAW>): a cell gets the same values whichever of the two kernels its row takes
	constexpr bool RAW = NATIVE && MODE==0;
	auto decode = [](const T v) {
		if constexpr(RAW) return __uint_as_float(((uint32_t)(int32_t)(int16_t)v<<12)&0x87FFF000u); else return ddf_decode<T>(v);
	};
	// x-face inputs of a border lane's cell: fetched HERE, ahead of the DDF loads (like the pair kernel's), so that the wave waits for memory once.  Behind
	// the loads, where the values are merged in, they were a second round trip for every wave that holds a border column: alone on the device a 128-cell x
	// slab of a 514x514x512 domain lost 12 % to it and a whole-row box 3 %, now 3.6 % and 0.4 % (tools/box_rate_probe.py, PROBE_XFACE=inout;
	// profiles/r05_zchunks.txt); beside the interior, as in the default schedule, the step is the same within 0.5 %
	[[maybe_unused]] T xv[5];
	[[maybe_unused]] bool xin_here = false;
	if constexpr(XFACE) {
		xin_here = (x==1u&&xin_m!=nullptr)||(x==p.Nx-2u&&xin_p!=nullptr);
		if(xin_here) {
			const T* const src = x==1u ? xin_m : xin_p;
			const int sg = x==1u ? -1 : 1;
			const size_t A = (size_t)p.Ny*p.Nz;
			xv[0] = src[xface_in_elem(p, y, z, 0, 0)]; xv[1] = src[A+xface_in_elem(p, y, z, sg, 0)]; xv[2] = src[2u*A+xface_in_elem(p, y, z, -sg, 0)];
			xv[3] = src[3u*A+xface_in_elem(p, y, z, 0, sg)]; xv[4] = src[4u*A+xface_in_elem(p, y, z, 0, -sg)];
		}
	}
	float f[19];
	f[0] = decode(ldo<(NT!=0)>(fi, a.own()));
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
		f[i  ] = decode(ldo<(NT!=0)>(fi+(size_t)slotA<PARITY>(i)*Np, a.own()));
		f[i+1] = decode(ldo<(NT==1||(NT==2&&!shifted))>(fi+(size_t)slotB<PARITY>(i)*Np+a.template row<i>(), a.template lane<i>()));
	});
	if constexpr(XFACE) {   // (first column: the own slots A(1, 7, 13, 9, 15); last column: the slots B in the halo column, loaded into f[2, 8, 14, 10, 16])
		if(xin_here) {
			if(x==1u) { f[1] = decode(xv[0]); f[7] = decode(xv[1]); f[13] = decode(xv[2]); f[9] = decode(xv[3]); f[15] = decode(xv[4]); }
			else { f[2] = decode(xv[0]); f[8] = decode(xv[1]); f[14] = decode(xv[2]); f[10] = decode(xv[3]); f[16] = decode(xv[4]); }
		}
	}
	[[maybe_unused]] float g[7];   // MODE 4: post-collision populations of the thermal lattice
	if constexpr(MODE!=1) {
		float rhon, uxn, uyn, uzn;
		if constexpr(NATIVE) {
			static_assert(sizeof(T)==2&&(MODE==0||MODE==4)&&!(STATS&&MODE==4), "native arithmetic: FP16C steps (sampled ones without the thermal lattice)");
			const bool is_E = (flagsn&TYPE_BO)==TYPE_E;
			ForceRefs refs;
			fetch_force_refs(p, n, x, y, z, true, is_E, rho, u, refs);
			const bool may_force = p.coriolis || p.has_F || p.fx!=0.0f || p.fy!=0.0f || p.fz!=0.0f || __ballot(refs.zn||refs.zs)!=0ull;
			f32x2 fp[9];
			#pragma unroll
			for(int k=0; k<9; k++) { const f32x2 t = { is_E ? 0.0f : f[2*k+1], is_E ? 0.0f : f[2*k+2] }; fp[k] = t; }
			float f0 = is_E ? 0.0f : f[0], u0[3];
			collide_cell_pk_native<PAIR_FORCE_ANY, RAW>(p, n, flagsn, may_force, f0, fp, rho, u, F, rhon, uxn, uyn, uzn, MODE==4 ? u0 : nullptr, &refs, &refs);
			f[0] = f0;
			#pragma unroll
			for(int k=0; k<9; k++) { f[2*k+1] = fp[k].x; f[2*k+2] = fp[k].y; }
			if constexpr(MODE==4) thermal_collide<T, PARITY>(p, n, a.jx(), a.jy(), a.jz(), x, y, z, flagsn, u0[0], u0[1], u0[2], gi, Tf, g, write_fields!=0);
		} else if constexpr(MODE==4) { // MODE 4: with the thermal lattice (LUW_OPT_TEMPERATURE)
			float u0[3];
			collide_cell<true, NOFORCE, (sizeof(T)==2&&LUW_PLAIN_ARITH!=0)>(p, n, x, y, z, flagsn, f, rho, u, F, rhon, uxn, uyn, uzn, u0);
			thermal_collide<T, PARITY>(p, n, a.jx(), a.jy(), a.jz(), x, y, z, flagsn, u0[0], u0[1], u0[2], gi, Tf, g, write_fields!=0);
		} else
		// MODE 3: general path only (A/B)
		collide_cell<(MODE!=3), NOFORCE, (sizeof(T)==2&&LUW_PLAIN_ARITH!=0)>(p, n, x, y, z, flagsn, f, rho, u, F, rhon, uxn, uyn, uzn);
		if(write_fields && (flagsn&TYPE_BO)!=TYPE_E) {
			rho[n] = rhon;
			if constexpr(NATIVE) { u[n] = uxn+0.0f; u[Np+n] = uyn+0.0f; u[2ull*Np+n] = uzn+0.0f; } // (zero velocities as +0, like the pair kernel's native path)
			else { u[n] = uxn; u[Np+n] = uyn; u[2ull*Np+n] = uzn; }
		}
		if constexpr(STATS) {
			if((flagsn&TYPE_BO)==TYPE_E) stats_welford_from_fields(Np, S, n, rho, u); // TYPE_E keeps its input fields (UPDATE_FIELDS skips it)
			else if constexpr(NATIVE) stats_welford(Np, S, n, rhon, uxn+0.0f, uyn+0.0f, uzn+0.0f);   // (zero velocities as +0, like the field write)
			else stats_welford(Np, S, n, rhon, uxn, uyn, uzn);
		}
	}
	// the ten offsets pass through an empty asm so that they are (re)defined as 32-bit values in the block that holds the
	// stores: instruction selection works per basic block, and without seeing the zero-extension there it builds nineteen
	// 64-bit addresses (v_lshl_add_u64 + a VGPR pair each) instead of the saddr form the loads use
	a.redefine();
	if constexpr(sizeof(T)==2&&MODE!=1) { // FP16C, nothing but the stores left: the 3-instruction encode under round-toward-zero
		uint32_t c[19];
		if constexpr(MODE==4) {
			uint32_t cg[7];
			fp16c_encode19_hi_rtz_final(f, c, g, cg);
			thermal_store<T, PARITY>(p, n, a.jx(), a.jy(), a.jz(), gi, [&](const int i) { return (T)(cg[i]>>16); });
		} else if constexpr(RAW) {
			#pragma unroll
			for(int i=0; i<19; i++) c[i] = fp16c_code_hi_of_scaled(f[i]);
		} else fp16c_encode19_hi_rtz_final(f, c);
		sto<(NT!=0)>(fi, a.own(), (T)(c[0]>>16));
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
			sto<(NT==1||(NT==2&&!shifted))>(fi+(size_t)slotB<PARITY>(i)*Np+a.template row<i>(), a.template lane<i>(), (T)(c[i]>>16));
			sto<(NT!=0)>(fi+(size_t)slotA<PARITY>(i)*Np, a.own(), (T)(c[i+1]>>16));
		});
		if constexpr(XFACE) xface_out([&](const int i) { return (T)(c[i]>>16); });
		return;
	}
	if constexpr(MODE==4) {
		thermal_store<T, PARITY>(p, n, a.jx(), a.jy(), a.jz(), gi, [&](const int i) { return ddf_encode<T>(g[i]); });
	}
	sto<(NT!=0)>(fi, a.own(), ddf_encode<T>(f[0]));
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
		sto<(NT==1||(NT==2&&!shifted))>(fi+(size_t)slotB<PARITY>(i)*Np+a.template row<i>(), a.template lane<i>(), ddf_encode<T>(f[i]));
		sto<(NT!=0)>(fi+(size_t)slotA<PARITY>(i)*Np, a.own(), ddf_encode<T>(f[i+1]));
	});
	if constexpr(XFACE) xface_out([&](const int i) { return ddf_encode<T>(f[i]); });
}
