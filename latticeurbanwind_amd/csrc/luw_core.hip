// luw_core.hip -- HIP kernels (gfx950) + C-ABI of the MI355X-native D3Q19 core.  See include/luw_core.h.
//
// Kernels
//   k_initialize          one-off: f_eq(rho,u) -> Esoteric-Pull store with t=1      (FX/kernel.cpp:1370-1452)
//   k_stream_collide_s    1 cell per lane, dword accesses, direct neighbour addressing (reference-style access
//                         pattern; the correctness baseline and the A/B partner of the vector kernel)
//   k_stream_collide_v    V=4 (FP32) cells per lane: every population moves as one aligned 16-byte access per
//                         lane; the five x+1 populations are aligned loads shifted across lanes with wave64
//                         cross-lane ops, so a wave touches whole 1-KiB row segments only.  Solid cells
//                         pass their populations through unchanged (each DDF slot is owned by exactly one cell
//                         per step, so this is value-identical to "do not touch memory", FX/kernel.cpp:1490).
//   k_extract_fi/k_insert_fi  halo pack/unpack of the 5 outgoing DDFs per face cell   (FX/kernel.cpp:2241-2270)
//
// Memory layout in HBM: SoA planes fi[q][z][y][x] with x-pitch Px (multiple of 64) and plane stride Np=Px*Ny*Nz, every
// array shifted by a lead pad so that the first owned cell of a row starts a 256-byte block (lead_alloc);
// rho[Np], u[3][Np], flags[Np], F[3][Np] share the pitch.  Host mirrors keep the reference layout (pitch Nx).
#include "luw_device.hpp"
#include "../../include/luw_core.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace luw;

// =====================================================================================================
// kernels
// =====================================================================================================

// Esoteric-Pull slots (FX/kernel.cpp:1338-1351): for odd i, A(i) is the plane read/written at the cell itself
// (carries f[i] in, f[i+1] out), B(i) the plane read/written at the +c_i neighbour (f[i+1] in, f[i] out).
template<int PARITY> __device__ __forceinline__ constexpr int slotA(const int i) { return PARITY ? i : i+1; }
template<int PARITY> __device__ __forceinline__ constexpr int slotB(const int i) { return PARITY ? i+1 : i; }

__device__ __forceinline__ bool cell_is_halo(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z) {
	return (p.halo_x&&(x==0u||x>=p.Nx-1u))||(p.halo_y&&(y==0u||y>=p.Ny-1u))||(p.halo_z&&(z==0u||z>=p.Nz-1u));
}

struct Box { uint32_t x0, x1, y0, y1, z0, z1; };

// neighbour offsets of one cell (periodic wrap, FX/kernel.cpp:920-958), as 32-bit device offsets
struct Nbr { uint32_t j[19]; };
__device__ __forceinline__ void neighbors(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z, uint32_t* j) {
	const uint32_t xp = x+1u==p.Nx ? 0u : x+1u, xm = x==0u ? p.Nx-1u : x-1u;
	const uint32_t y0 = y*p.Px, yp = (y+1u==p.Ny ? 0u : y+1u)*p.Px, ym = (y==0u ? p.Ny-1u : y-1u)*p.Px;
	const uint32_t A = p.Px*p.Ny;
	const uint32_t z0 = z*A, zp = (z+1u==p.Nz ? 0u : z+1u)*A, zm = (z==0u ? p.Nz-1u : z-1u)*A;
	j[ 0] = x+y0+z0;
	j[ 1] = xp+y0+z0; j[ 2] = xm+y0+z0;
	j[ 3] = x +yp+z0; j[ 4] = x +ym+z0;
	j[ 5] = x +y0+zp; j[ 6] = x +y0+zm;
	j[ 7] = xp+yp+z0; j[ 8] = xm+ym+z0;
	j[ 9] = xp+y0+zp; j[10] = xm+y0+zm;
	j[11] = x +yp+zp; j[12] = x +ym+zm;
	j[13] = xp+ym+z0; j[14] = xm+yp+z0;
	j[15] = xp+y0+zm; j[16] = xm+y0+zp;
	j[17] = x +yp+zm; j[18] = x +ym+zp;
}

template<typename T> __global__ __launch_bounds__(256) void k_initialize(const KParams p, T* __restrict__ fi, const float* __restrict__ rho, float* __restrict__ u, const uint8_t* __restrict__ flags, T* __restrict__ gi, const float* __restrict__ Tf) {
	const uint32_t x = blockIdx.x*blockDim.x+threadIdx.x, y = blockIdx.y, z = blockIdx.z;
	if(x>=p.Nx) return;
	if(cell_is_halo(p, x, y, z)) return;
	uint32_t j[19];
	neighbors(p, x, y, z, j);
	const uint32_t n = j[0];
	if((flags[n]&TYPE_BO)==TYPE_S) { // FX/kernel.cpp:1386-1399: u = 0 on every solid cell
		u[n] = 0.0f; u[(size_t)p.Np+n] = 0.0f; u[2ull*p.Np+n] = 0.0f;
	}
	float feq[19];
	calculate_f_eq(rho[n], u[n], u[(size_t)p.Np+n], u[2ull*p.Np+n], feq);
	// store_f with t = 1 (odd), FX/kernel.cpp:1451
	fi[n] = ddf_encode<T>(feq[0]);
	#pragma unroll
	for(int i=1; i<19; i+=2) {
		fi[(size_t)slotB<1>(i)*p.Np+j[i]] = ddf_encode<T>(feq[i]);
		fi[(size_t)slotA<1>(i)*p.Np+n] = ddf_encode<T>(feq[i+1]);
	}
	if(gi) { // TEMPERATURE: store_g(geq(T, u), t = 1), FX/kernel.cpp:1442-1449
		float geq[7];
		calculate_g_eq(Tf[n], u[n], u[(size_t)p.Np+n], u[2ull*p.Np+n], geq);
		gi[n] = ddf_encode<T>(geq[0]);
		#pragma unroll
		for(int i=1; i<7; i+=2) {
			gi[(size_t)(i+1)*p.Np+j[i]] = ddf_encode<T>(geq[i]);
			gi[(size_t)i*p.Np+n] = ddf_encode<T>(geq[i+1]);
		}
	}
}

// DDF accesses are streaming: every slot is read once and written once per step, so all DDF loads / stores carry
// the non-temporal hint (global_load/store ... nt).  Measured on MI355X (tools/membench.hip, profiles/): the
// 19-plane in-place update moves 5.3 TB/s with the default cache policy and 6.1 TB/s non-temporal.
template<int I> struct IC { static constexpr int value = I; };
template<typename Fn> __device__ __forceinline__ void static_for_pairs(Fn&& fn) { // i = 1,3,...,17 as compile-time constants
	fn(IC<1>{}); fn(IC<3>{}); fn(IC<5>{}); fn(IC<7>{}); fn(IC<9>{}); fn(IC<11>{}); fn(IC<13>{}); fn(IC<15>{}); fn(IC<17>{});
}
template<bool NT, typename T> __device__ __forceinline__ T ldg(const T* p) { if constexpr(NT) return __builtin_nontemporal_load(p); else return *p; }
template<bool NT, typename T> __device__ __forceinline__ void stg(T* p, const T v) { if constexpr(NT) __builtin_nontemporal_store(v, p); else *p = v; }

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
		if(noshift) { o.j1 = o.n; o.j7 = o.j3; o.j9 = o.j5; o.j13 = (x+(y==0u ? p.Ny-1u : y-1u)*p.Px+z*p.Px*p.Ny)*(uint32_t)sizeof(T); o.j15 = (x+y*p.Px+(z==0u ? p.Nz-1u : z-1u)*p.Px*p.Ny)*(uint32_t)sizeof(T); }
		n = o.n/(uint32_t)sizeof(T);
		return 0u;
	}
	template<int I> __device__ __forceinline__ int64_t row() const { return 0; }
	template<int I> __device__ __forceinline__ uint32_t lane() const { return nbr<I>(o); }
	__device__ __forceinline__ uint32_t own() const { return o.n; }
	__device__ __forceinline__ uint32_t jx() const { return o.j1/(uint32_t)sizeof(T); }
	__device__ __forceinline__ uint32_t jy() const { return o.j3/(uint32_t)sizeof(T); }
	__device__ __forceinline__ uint32_t jz() const { return o.j5/(uint32_t)sizeof(T); }
	__device__ __forceinline__ void redefine() { asm volatile("" : "+v"(o.n), "+v"(o.j1), "+v"(o.j3), "+v"(o.j5), "+v"(o.j7), "+v"(o.j9), "+v"(o.j11), "+v"(o.j13), "+v"(o.j15), "+v"(o.j17)); }
};

// MODE 0 is the product kernel.  MODE 1 ("copy": no collision) and MODE 2 ("noshift": x+1 neighbours replaced by x) are
// measurement-only variants that isolate the memory system's share of the step; they do not compute physics.
// NT: 0 default cache policy, 1 non-temporal everywhere, 2 non-temporal on the 14 aligned planes and default policy on
// the five x+1 planes, whose wave-edge lines are shared between neighbouring waves (product setting, measured best).
// Waves per SIMD: the FP32 kernel is HBM-bound and measurably better with at most 4 resident waves (3.40 vs 3.43 ms at 512^3,
// 6.68 vs 6.98 ms at 1024x1024x256: fewer concurrent row fronts, better DRAM page locality) even though its ~95 VGPRs would
// allow 5; the FP16C kernel is VALU-bound and takes all the waves its registers allow.
#ifndef LUW_MAXW_F32
#define LUW_MAXW_F32 4
#endif
template<typename T, int PARITY, int MODE=0, int NT=2, bool FLAT=false> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((sizeof(T)==4 && LUW_MAXW_F32<4) ? LUW_MAXW_F32 : 4, sizeof(T)==4 ? LUW_MAXW_F32 : 8))) void k_stream_collide_s(const KParams p, const Box b, const int xa, T* __restrict__ fi, float* __restrict__ rho, float* __restrict__ u,
		const uint8_t* __restrict__ flags, const float* __restrict__ F, const int write_fields, T* __restrict__ gi = nullptr, float* __restrict__ Tf = nullptr) {
	// lanes are laid over the row in 64-cell blocks aligned with the memory lines (xa = b.x0 rounded down to such a block
	// start), whatever the box: lanes left of b.x0 idle
	// (workgroups go round-robin to the 8 XCDs; remapping them so that each XCD owns one contiguous eighth of the box was
	// measured 3.5 % slower, 3.50 vs 3.39 ms at 512^3 -- no population is shared between workgroups, so there is nothing
	// for an XCD's L2 to reuse, and eight distant fronts cost DRAM page locality)
	const int xi = xa+(int)(blockIdx.x*blockDim.x+threadIdx.x);
	if(xi<(int)b.x0||xi>=(int)b.x1) return;
	const uint32_t x = (uint32_t)xi, y = b.y0+blockIdx.y, z = b.z0+blockIdx.z;
	if(cell_is_halo(p, x, y, z)) return;
	CellAddr<T, FLAT> a;
	fi += a.init(p, x, y, z, MODE==2);
	const uint32_t n = a.n;
	const uint8_t flagsn = flags[n];
	if((flagsn&TYPE_BO)==TYPE_S||(flagsn&TYPE_SU)==TYPE_G) return;
	const size_t Np = p.Np;
	float f[19];
	f[0] = ddf_decode<T>(ldo<(NT!=0)>(fi, a.own()));
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
		f[i  ] = ddf_decode<T>(ldo<(NT!=0)>(fi+(size_t)slotA<PARITY>(i)*Np, a.own()));
		f[i+1] = ddf_decode<T>(ldo<(NT==1||(NT==2&&!shifted))>(fi+(size_t)slotB<PARITY>(i)*Np+a.template row<i>(), a.template lane<i>()));
	});
	[[maybe_unused]] float g[7];   // MODE 4: post-collision populations of the thermal lattice
	if constexpr(MODE!=1) {
		float rhon, uxn, uyn, uzn;
		if constexpr(MODE==4) { // MODE 4: with the thermal lattice (LUW_OPT_TEMPERATURE)
			float u0[3];
			collide_cell<true>(p, n, x, y, z, flagsn, f, rho, u, F, rhon, uxn, uyn, uzn, u0);
			thermal_collide<T, PARITY>(p, n, a.jx(), a.jy(), a.jz(), x, y, z, flagsn, u0[0], u0[1], u0[2], gi, Tf, g);
		} else
		collide_cell<(MODE!=3)>(p, n, x, y, z, flagsn, f, rho, u, F, rhon, uxn, uyn, uzn); // MODE 3: general path only (A/B)
		if(write_fields && (flagsn&TYPE_BO)!=TYPE_E) {
			rho[n] = rhon;
			u[n] = uxn;
			u[Np+n] = uyn;
			u[2ull*Np+n] = uzn;
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
		} else fp16c_encode19_hi_rtz_final(f, c);
		sto<(NT!=0)>(fi, a.own(), (T)(c[0]>>16));
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
			sto<(NT==1||(NT==2&&!shifted))>(fi+(size_t)slotB<PARITY>(i)*Np+a.template row<i>(), a.template lane<i>(), (T)(c[i]>>16));
			sto<(NT!=0)>(fi+(size_t)slotA<PARITY>(i)*Np, a.own(), (T)(c[i+1]>>16));
		});
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
}

// ---------------------------------------------------------------- pair kernel: 2 cells per lane (FP16C DDFs)
// With 2-byte DDFs the scalar kernel moves only 128 B per wave instruction, and that access width tops out near 5.2 TB/s
// (tools/membench half).  Here a lane owns the cells (x, x+1), x even, and moves both FP16C codes of a plane with ONE
// dword access -- the same bytes per instruction as the FP32 scalar kernel.  Straight planes are 4-byte aligned; the x+1
// planes are read/written at a 2-byte offset (a dword access on a 2-byte boundary, which the hardware serves), except for
// the lane at the row end whose second neighbour wraps to x = 0.  The two cells are collided ONE AFTER THE OTHER (an
// asm fence between the passes keeps the compiler from interleaving them), so the register footprint is that of the
// scalar kernel plus the 19 finished values of the first cell: 4 waves/SIMD.  Both cells are encoded at the tail under
// round-toward-zero (fp16c_code_hi_in_rtz_mode) and merged into dwords with one byte permute per plane.
// A cell that must not be processed (solid / halo) passes its populations through; its values are pre-swapped so that
// the Esoteric-Pull store puts them back where they came from (every slot has exactly one writing cell per step, so this
// rewrite races with nobody).  Requires an even b.x0, an even b.x1 (or b.x1 = an odd Nx: the row's last cell then pairs with the row padding and is the only
// one processed by its lane) and rows whose x = 0 sits on a 4-byte boundary (the host
// falls back to the scalar kernel otherwise).
typedef uint32_t u32_a2 __attribute__((aligned(2)));
template<bool NT> __device__ __forceinline__ uint32_t ld_pair(const uint16_t* plane, const uint32_t byte_off) {
	const char* ptr = reinterpret_cast<const char*>(plane)+byte_off;
	if constexpr(NT) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(ptr));
	else return *reinterpret_cast<const u32_a2*>(ptr);
}
template<bool NT> __device__ __forceinline__ void st_pair(uint16_t* plane, const uint32_t byte_off, const uint32_t v) {
	char* ptr = reinterpret_cast<char*>(plane)+byte_off;
	if constexpr(NT) __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(ptr));
	else *reinterpret_cast<u32_a2*>(ptr) = v;
}
// every value of v[0..N) passes through a volatile asm: what produces them is ordered before, what consumes them after
template<int N> __device__ __forceinline__ void asm_fence(float* v) {
	static_assert(N==19, "written for the 19 DDFs of a cell");
	asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
	asm volatile("" : "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]));
}
__device__ __forceinline__ void asm_fence9(float& f0, f32x2* v) {
	asm volatile("" : "+v"(f0), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
}
__device__ __forceinline__ void asm_fence_u(uint32_t* v) {
	asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
	asm volatile("" : "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]));
}
template<int PARITY, int MODE=0> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_stream_collide_p(const KParams p, const Box b, uint16_t* __restrict__ fi, float* __restrict__ rho, float* __restrict__ u,
		const uint8_t* __restrict__ flags, const float* __restrict__ F, const int write_fields) {
	const uint32_t x = b.x0+2u*(blockIdx.x*blockDim.x+threadIdx.x), y = b.y0+blockIdx.y, z = b.z0+blockIdx.z;
	if(x>=b.x1) return;
	const RowOff rb = row_offsets(p, y, z);
	LaneOff o = lane_offsets<uint16_t>(p, x);                      // offsets of cell x; cell x+1 sits 2 bytes further
	const bool wrap = x+2u==p.Nx;                                  // cell x+1 is the last of the row: its x+1 neighbour is x = 0
	const bool tail = x+1u==p.Nx;                                  // odd Nx: cell x is the last of the row, "cell x+1" is the row's padding
	const uint32_t n = x+(uint32_t)rb.r00;
	fi += rb.r00;                                                  // own row (uniform)
	const size_t Np = p.Np;
	const uint32_t fl2 = *reinterpret_cast<const uint16_t*>(flags+n);
	const uint8_t fl[2] = { (uint8_t)(fl2&0xFFu), (uint8_t)(fl2>>8) };
	bool proc[2];
	#pragma unroll
	for(int c=0; c<2; c++) proc[c] = !cell_is_halo(p, x+c, y, z) && (fl[c]&TYPE_BO)!=TYPE_S && (fl[c]&TYPE_SU)!=TYPE_G;
	if(tail) proc[1] = false;                                      // passes through: reads and rewrites padding, except on the x+1 planes (below)
	if(!proc[0]&&!proc[1]) return;
	// All 19 dword loads in one straight run.  For the row-end lane (wrap) the high half of the five x+1 dwords is the element
	// behind the row end (the next row's x = 0, or the slack behind the plane: in bounds, see lead_alloc / the plane skew):
	// it is replaced by the wrapped neighbour in ONE divergent fix-up block behind the loads.
	uint32_t raw[19];                                              // low half: cell x, high half: cell x+1
	raw[0] = ld_pair<true>(fi, o.x);
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
		raw[i] = ld_pair<true>(fi+(size_t)slotA<PARITY>(i)*Np, o.x);
		raw[i+1] = ld_pair<!shifted>(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb), nlane<i>(o));
	});
	if(wrap) {
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			if constexpr(i==1||i==7||i==9||i==13||i==15) { // the dword started at x+1 = Nx-1 of the neighbour row; x+2 wraps to that row's x = 0
				const uint32_t hi = *(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb));
				raw[i+1] = (raw[i+1]&0xFFFFu)|(hi<<16);
			}
		});
	}
	// wave-uniform: can any cell of this wave feel a force (then the Guo terms are computed for the whole wave)?
	const bool may_force = p.coriolis || p.has_F || p.fx!=0.0f || p.fy!=0.0f || p.fz!=0.0f || __ballot(in_force_zone(p, x, y, z)||in_force_zone(p, x+1u, y, z))!=0ull;
	// one cell: decode its half of the 19 dwords into f0 and the nine (f[2k+1], f[2k+2]) pairs, collide on the packed pairs
	// (or pre-swap for the pass-through)
	auto one_cell = [&](const int c, float& f0, f32x2* fp) {
		auto bits = [&](const int q) { // (sign-extended half) << 12 in one SDWA shift, then the mask of half_to_float_custom_sx
			uint32_t t;
			if(c) asm("v_lshlrev_b32_sdwa %0, 12, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(t) : "v"(raw[q]));
			else asm("v_lshlrev_b32_sdwa %0, 12, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(t) : "v"(raw[q]));
			return t&0x87FFF000u;
		};
		f0 = __uint_as_float(bits(0))*0x1p+112f;
		#pragma unroll
		for(int k=0; k<9; k++) { const f32x2 t = { __uint_as_float(bits(2*k+1)), __uint_as_float(bits(2*k+2)) }; fp[k] = t*splat2(0x1p+112f); }
		if(MODE!=1&&proc[c]) { // MODE 1: measurement-only, no collision (every cell passes through)
			float rhon, uxn, uyn, uzn;
			collide_cell_pk(p, n+c, x+c, y, z, fl[c], may_force, f0, fp, rho, u, F, rhon, uxn, uyn, uzn);
			if(write_fields && (fl[c]&TYPE_BO)!=TYPE_E) {
				rho[n+c] = rhon;
				u[n+c] = uxn;
				u[Np+n+c] = uyn;
				u[2ull*Np+n+c] = uzn;
			}
		} else {
			#pragma unroll
			for(int k=0; k<9; k++) { const f32x2 t = { fp[k].y, fp[k].x }; fp[k] = t; }
		}
	};
	float fa0, fb0; f32x2 fa[9], fb[9];
	one_cell(0, fa0, fa);
	asm_fence9(fa0, fa); asm_fence_u(raw);                         // cell x is finished before cell x+1 starts
	one_cell(1, fb0, fb);
	asm_fence9(fa0, fa); asm_fence9(fb0, fb);                      // all floating-point work is done ...
	asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3"); // ... before the wave's FP32 rounding mode becomes RTZ (see luw_device.hpp)
	// codes of both cells in the high halves, merged per plane: q = 0 or 2k+1+h
	uint32_t ca[19], cb[19];
	ca[0] = fp16c_code_hi_in_rtz_mode(fa0); cb[0] = fp16c_code_hi_in_rtz_mode(fb0);
	#pragma unroll
	for(int k=0; k<9; k++) {
		fp16c_code2_hi_in_rtz_mode(fa[k], ca[2*k+1], ca[2*k+2]);
		fp16c_code2_hi_in_rtz_mode(fb[k], cb[2*k+1], cb[2*k+2]);
	}
	auto pack = [&](const int q) { return __builtin_amdgcn_perm(cb[q], ca[q], 0x07060302u); };
	uint32_t cs[5];   // the five x+1 planes, stored last (dword or, on the row-end lane, two halves)
	#define LUW_REDEFINE_OFFSETS asm volatile("" : "+v"(o.x), "+v"(o.xp)) /* saddr stores, see k_stream_collide_s */
	LUW_REDEFINE_OFFSETS;
	st_pair<true>(fi, o.x, pack(0));
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
		constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
		if constexpr(!shifted) st_pair<true>(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb), nlane<i>(o), pack(i));
		else cs[k] = pack(i);
		st_pair<true>(fi+(size_t)slotA<PARITY>(i)*Np, o.x, pack(i+1));
	});
	if(!wrap&&!tail) {
		LUW_REDEFINE_OFFSETS;
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
			if constexpr(i==1||i==7||i==9||i==13||i==15) st_pair<false>(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb), nlane<i>(o), cs[k]);
		});
	} else if(tail) { // the only real cell is x = Nx-1: its x+1 neighbour is the row's x = 0 (the dword load above already started there);
		// x = 1 belongs to another lane's stores, so only the low half goes out
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			if constexpr(i==1||i==7||i==9||i==13||i==15) {
				constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
				*(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb)) = (uint16_t)(cs[k]&0xFFFFu);
			}
		});
	} else {
		LUW_REDEFINE_OFFSETS;
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			if constexpr(i==1||i==7||i==9||i==13||i==15) {
				constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
				uint16_t* B = fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb);   // row of the neighbours: x+1 = Nx-1 is its last cell, x+2 its first
				B[p.Nx-1u] = (uint16_t)(cs[k]&0xFFFFu);
				B[0] = (uint16_t)(cs[k]>>16);
			}
		});
	}
	#undef LUW_REDEFINE_OFFSETS
}

// ---------------------------------------------------------------- vector kernel: V cells per lane
template<typename T, int V> struct Pack { T v[V]; };
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template<int BYTES> struct RawT;
template<> struct RawT<2> { typedef uint16_t type; };
template<> struct RawT<4> { typedef uint32_t type; };
template<> struct RawT<8> { typedef u32x2 type; };
template<> struct RawT<16> { typedef u32x4 type; };
// one aligned V*sizeof(T)-byte access per lane (4, 8 or 16 bytes), non-temporal
template<typename T, int V> __device__ __forceinline__ Pack<T, V> vload(const T* ptr) {
	typedef typename RawT<V*sizeof(T)>::type R;
	union { R r; Pack<T, V> p; } c;
	c.r = __builtin_nontemporal_load(reinterpret_cast<const R*>(ptr));
	return c.p;
}
template<typename T, int V> __device__ __forceinline__ void vstore(T* ptr, const Pack<T, V>& v) {
	typedef typename RawT<V*sizeof(T)>::type R;
	union { R r; Pack<T, V> p; } c;
	c.p = v;
	__builtin_nontemporal_store(c.r, reinterpret_cast<R*>(ptr));
}
template<typename T> __device__ __forceinline__ T lane_down(const T v) { // value held by lane+1
	return (T)__shfl_down((int)v, 1, 64);
}
template<> __device__ __forceinline__ float lane_down<float>(const float v) { return __shfl_down(v, 1, 64); }
template<typename T> __device__ __forceinline__ T lane_up(const T v) { // value held by lane-1
	return (T)__shfl_up((int)v, 1, 64);
}
template<> __device__ __forceinline__ float lane_up<float>(const float v) { return __shfl_up(v, 1, 64); }

// Launch geometry: blockDim = (VX, RY), VX a power of two <= 256, VX*RY = 256.  blockIdx.x = rowblock*nchunk + chunk.
// A lane owns the V cells X..X+V-1 (X = V*k) of row (y,z); rows are enumerated r = (z-z0)*(y1-y0) + (y-y0); k runs
// over the vectors that overlap [b.x0,b.x1).  Lanes whose V cells all lie inside the box ("full") move whole
// vectors; lanes on the box edge (or holding row padding) store element-wise and only what in-box cells own, so a
// launch never writes a DDF slot owned by a cell outside its box (required when halo unpack / shell passes of the
// multi-GPU driver run concurrently on another stream).
template<typename T, int V, int PARITY> __global__ __launch_bounds__(256) void k_stream_collide_v(const KParams p, const Box b, const uint32_t nchunk, T* __restrict__ fi, float* __restrict__ rho, float* __restrict__ u,
		const uint8_t* __restrict__ flags, const float* __restrict__ F, const int write_fields) {
	const uint32_t kfirst = b.x0/V, klast = (b.x1-1u)/V;
	const uint32_t chunk = blockIdx.x%nchunk, rowblock = blockIdx.x/nchunk;
	const uint32_t k = kfirst+chunk*blockDim.x+threadIdx.x;
	const uint32_t ny = b.y1-b.y0;
	const uint32_t r = rowblock*blockDim.y+threadIdx.y;
	const bool row_ok = r<ny*(b.z1-b.z0);
	const uint32_t y = b.y0+(row_ok ? r%ny : 0u), z = b.z0+(row_ok ? r/ny : 0u);
	const bool active = row_ok && k<=klast;
	const uint32_t X = V*(active ? k : kfirst);
	const uint32_t lane = (threadIdx.y*blockDim.x+threadIdx.x)&63u;
	auto is_full = [&](const uint32_t kk) { return V*kk>=b.x0 && V*kk+V<=b.x1; };
	const bool full = active && is_full(k);
	// lane+1 / lane-1 hold the neighbouring vectors k+1 / k-1 of the same row?
	const bool nb_next = lane<63u && threadIdx.x+1u<blockDim.x && k+1u<=klast;
	const bool nb_prev = lane>0u && threadIdx.x>0u;
	const bool next_full = nb_next && is_full(k+1u);
	const bool prev_full = nb_prev && is_full(k-1u);

	const uint32_t Arow = p.Px*p.Ny;
	const uint32_t yp = (y+1u==p.Ny ? 0u : y+1u), ym = (y==0u ? p.Ny-1u : y-1u);
	const uint32_t zp = (z+1u==p.Nz ? 0u : z+1u), zm = (z==0u ? p.Nz-1u : z-1u);
	const uint32_t r00 = y*p.Px+z*Arow;     // own row
	const uint32_t rp0 = yp*p.Px+z*Arow, rm0 = ym*p.Px+z*Arow;
	const uint32_t r0p = y*p.Px+zp*Arow, r0m = y*p.Px+zm*Arow;
	const uint32_t rpp = yp*p.Px+zp*Arow, rpm = yp*p.Px+zm*Arow;
	const uint32_t n0 = r00+X;

	// the lane that holds cell x = Nx-1 wraps to x = 0 of the same row for its x+1 neighbour
	const uint32_t kw = (p.Nx-1u)/V, cw = (p.Nx-1u)%V;
	const bool is_wrap = active && k==kw;

	float f[19][V];
	uint8_t fl[V];
	if(active) {
		if constexpr(V==4) { const uchar4 t = *reinterpret_cast<const uchar4*>(flags+n0); fl[0] = t.x; fl[1] = t.y; fl[2] = t.z; fl[3] = t.w; }
		else if constexpr(V==2) { const uchar2 t = *reinterpret_cast<const uchar2*>(flags+n0); fl[0] = t.x; fl[1] = t.y; }
		else fl[0] = flags[n0];
	} else {
		#pragma unroll
		for(int c=0; c<V; c++) fl[c] = TYPE_S;
	}
	bool proc[V]; // cell is processed by this launch (in box, not halo, not solid/gas)
	#pragma unroll
	for(int c=0; c<V; c++) {
		const uint32_t x = X+c;
		proc[c] = active && x>=b.x0 && x<b.x1 && !cell_is_halo(p, x, y, z) && (fl[c]&TYPE_BO)!=TYPE_S && (fl[c]&TYPE_SU)!=TYPE_G;
	}

	// ---- load: straight (aligned) populations
	auto load_straight = [&](const int q, const int plane, const uint32_t row) {
		Pack<T, V> t;
		if(active) t = vload<T, V>(fi+(size_t)plane*p.Np+row+X);
		else { for(int c=0; c<V; c++) t.v[c] = (T)0; }
		#pragma unroll
		for(int c=0; c<V; c++) f[q][c] = ddf_decode<T>(t.v[c]);
	};
	// ---- load: populations living at x+1 (aligned vector + first element of the next lane, wrap at the row end)
	auto load_shifted = [&](const int q, const int plane, const uint32_t row) {
		const T* S = fi+(size_t)plane*p.Np+row;
		Pack<T, V> t;
		if(active) t = vload<T, V>(S+X);
		else { for(int c=0; c<V; c++) t.v[c] = (T)0; }
		T e = lane_down<T>(t.v[0]);
		if(active && !nb_next && X+V<p.Px) e = ldg<true>(S+X+V);
		T in[V];
		#pragma unroll
		for(int c=0; c<V-1; c++) in[c] = t.v[c+1];
		in[V-1] = e;
		if(is_wrap) {
			const T wv = ldg<true>(S);
			#pragma unroll
			for(int c=0; c<V; c++) if((uint32_t)c==cw) in[c] = wv;
		}
		#pragma unroll
		for(int c=0; c<V; c++) f[q][c] = ddf_decode<T>(in[c]);
	};
	load_straight(0, 0, r00);
	load_straight( 1, slotA<PARITY>( 1), r00); load_shifted ( 2, slotB<PARITY>( 1), r00); // +00
	load_straight( 3, slotA<PARITY>( 3), r00); load_straight( 4, slotB<PARITY>( 3), rp0); // 0+0
	load_straight( 5, slotA<PARITY>( 5), r00); load_straight( 6, slotB<PARITY>( 5), r0p); // 00+
	load_straight( 7, slotA<PARITY>( 7), r00); load_shifted ( 8, slotB<PARITY>( 7), rp0); // ++0
	load_straight( 9, slotA<PARITY>( 9), r00); load_shifted (10, slotB<PARITY>( 9), r0p); // +0+
	load_straight(11, slotA<PARITY>(11), r00); load_straight(12, slotB<PARITY>(11), rpp); // 0++
	load_straight(13, slotA<PARITY>(13), r00); load_shifted (14, slotB<PARITY>(13), rm0); // +-0
	load_straight(15, slotA<PARITY>(15), r00); load_shifted (16, slotB<PARITY>(15), r0m); // +0-
	load_straight(17, slotA<PARITY>(17), r00); load_straight(18, slotB<PARITY>(17), rpm); // 0+-

	// ---- collide the V cells; everything else passes through
	#pragma unroll
	for(int c=0; c<V; c++) {
		if(proc[c]) {
			const uint8_t flagsn = fl[c];
			float fc[19];
			#pragma unroll
			for(int q=0; q<19; q++) fc[q] = f[q][c];
			float rhon, uxn, uyn, uzn;
			collide_cell(p, n0+c, X+c, y, z, flagsn, fc, rho, u, F, rhon, uxn, uyn, uzn);
			if(write_fields && (flagsn&TYPE_BO)!=TYPE_E) {
				rho[n0+c] = rhon;
				u[n0+c] = uxn;
				u[(size_t)p.Np+n0+c] = uyn;
				u[2ull*p.Np+n0+c] = uzn;
			}
			#pragma unroll
			for(int q=0; q<19; q++) f[q][c] = fc[q];
		} else {
			// pass-through: the store phase swaps the slots of each pair (f[i] leaves through B(i), f[i+1] through
			// A(i)); pre-swap so that every value returns to the slot it was loaded from
			#pragma unroll
			for(int i=1; i<19; i+=2) { const float t = f[i][c]; f[i][c] = f[i+1][c]; f[i+1][c] = t; }
		}
	}

	// ---- store (Esoteric-Pull swap: what came in as f[i] leaves through B(i), f[i+1] through A(i))
	auto store_straight = [&](const int q, const int plane, const uint32_t row) {
		if(!active) return;
		T* S = fi+(size_t)plane*p.Np+row;
		if(full) {
			Pack<T, V> t;
			#pragma unroll
			for(int c=0; c<V; c++) t.v[c] = ddf_encode<T>(f[q][c]);
			vstore<T, V>(S+X, t);
		} else {
			#pragma unroll
			for(int c=0; c<V; c++) if(proc[c]) stg<true>(S+X+c, ddf_encode<T>(f[q][c]));
		}
	};
	auto store_shifted = [&](const int q, const int plane, const uint32_t row) {
		T* S = fi+(size_t)plane*p.Np+row;
		T o[V];
		#pragma unroll
		for(int c=0; c<V; c++) o[c] = ddf_encode<T>(f[q][c]);
		const T pv = lane_up<T>(o[V-1]); // out value of cell X-1 (meaningful when prev_full)
		if(!active) return;
		if(full) {
			if(prev_full) {
				Pack<T, V> t;
				t.v[0] = pv;
				#pragma unroll
				for(int c=1; c<V; c++) t.v[c] = o[c-1];
				vstore<T, V>(S+X, t);
			} else {
				// S[X] is owned by cell X-1, which another wave / an edge lane / nobody in this launch handles
				#pragma unroll
				for(int c=1; c<V; c++) stg<true>(S+X+c, o[c-1]);
			}
			if(!next_full && X+V<p.Nx) stg<true>(S+X+V, o[V-1]); // the element the next vector will not write for us
			if(is_wrap) {
				#pragma unroll
				for(int c=0; c<V; c++) if((uint32_t)c==cw) stg<true>(S, o[c]);
			}
		} else {
			#pragma unroll
			for(int c=0; c<V; c++) if(proc[c]) stg<true>(S+(X+c+1u==p.Nx ? 0u : X+c+1u), o[c]);
		}
	};
	store_straight(0, 0, r00);
	store_shifted ( 1, slotB<PARITY>( 1), r00); store_straight( 2, slotA<PARITY>( 1), r00);
	store_straight( 3, slotB<PARITY>( 3), rp0); store_straight( 4, slotA<PARITY>( 3), r00);
	store_straight( 5, slotB<PARITY>( 5), r0p); store_straight( 6, slotA<PARITY>( 5), r00);
	store_shifted ( 7, slotB<PARITY>( 7), rp0); store_straight( 8, slotA<PARITY>( 7), r00);
	store_shifted ( 9, slotB<PARITY>( 9), r0p); store_straight(10, slotA<PARITY>( 9), r00);
	store_straight(11, slotB<PARITY>(11), rpp); store_straight(12, slotA<PARITY>(11), r00);
	store_shifted (13, slotB<PARITY>(13), rm0); store_straight(14, slotA<PARITY>(13), r00);
	store_shifted (15, slotB<PARITY>(15), r0m); store_straight(16, slotA<PARITY>(15), r00);
	store_straight(17, slotB<PARITY>(17), rpm); store_straight(18, slotA<PARITY>(17), r00);
}

// ---------------------------------------------------------------- halo pack / unpack, FX/kernel.cpp:2188-2270
// Face cell of thread t and its index a in the transfer buffers.  The buffers keep the reference's order (direction 0:
// a = y + z Ny; 1: a = z + x Nz; 2: a = x + y Nx, FX/kernel.cpp:2188-2221), but the THREADS walk along x wherever x lies in the
// face, so that the lattice side of the copy is coalesced (for direction 1 the small buffer side is strided instead).
template<int DIR> __device__ __forceinline__ void face_cell(const KParams& p, const uint32_t t, const uint32_t fixed, uint32_t& x, uint32_t& y, uint32_t& z, uint32_t& a) {
	if constexpr(DIR==0) { x = fixed; y = t%p.Ny; z = t/p.Ny; a = t; }
	else if constexpr(DIR==1) { x = t%p.Nx; y = fixed; z = t/p.Nx; a = x*p.Nz+z; }
	else { x = t%p.Nx; y = t/p.Nx; z = fixed; a = t; }
}
// device index of the neighbour of (x,y,z) in direction c_I (periodic wrap), I compile-time: three selects, no table
template<int I> __device__ __forceinline__ uint32_t neighbor_index(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z) {
	constexpr int cx = (I==1||I==7||I==9||I==13||I==15) ? 1 : (I==2||I==8||I==10||I==14||I==16) ? -1 : 0;
	constexpr int cy = (I==3||I==7||I==11||I==14||I==17) ? 1 : (I==4||I==8||I==12||I==13||I==18) ? -1 : 0;
	constexpr int cz = (I==5||I==9||I==11||I==16||I==18) ? 1 : (I==6||I==10||I==12||I==15||I==17) ? -1 : 0;
	const uint32_t xs = cx>0 ? (x+1u==p.Nx ? 0u : x+1u) : cx<0 ? (x==0u ? p.Nx-1u : x-1u) : x;
	const uint32_t ys = cy>0 ? (y+1u==p.Ny ? 0u : y+1u) : cy<0 ? (y==0u ? p.Ny-1u : y-1u) : y;
	const uint32_t zs = cz>0 ? (z+1u==p.Nz ? 0u : z+1u) : cz<0 ? (z==0u ? p.Nz-1u : z-1u) : z;
	return xs+(ys+zs*p.Ny)*p.Px;
}
// the 5 D3Q19 populations that leave through face (DIR, side), FX/kernel.cpp:2223-2229 
template<int DIR, int PM, int BB> struct TransferIndex {
	static constexpr int table[30] = { 1, 7, 13, 9, 15,  2, 8, 14, 10, 16,  3, 7, 14, 11, 17,  4, 8, 13, 12, 18,  5, 9, 16, 11, 18,  6, 10, 15, 12, 17 };
	static constexpr int value = table[(2*DIR+PM)*5+BB];
};
// G = false: the 5 D3Q19 populations of a face (fi); G = true: the single D3Q7 population of the thermal lattice (gi, i = side+1,
// FX/kernel.cpp:2338-2351) -- same slot algebra, the D3Q7 neighbours are the first six of the D3Q19 list.
// Everything about a population is compile-time (direction is a template parameter), so no per-thread index table exists.
template<typename T, bool G, int DIR, int PM, int BB> __device__ __forceinline__ void extract_one(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z, const uint32_t a, const uint32_t A, const uint32_t t_odd, T* __restrict__ buf, const T* __restrict__ fi) {
	constexpr int i = G ? 2*DIR+PM+1 : TransferIndex<DIR, PM, BB>::value;
	const uint32_t plane = t_odd ? ((i&1) ? i+1 : i-1) : i;
	const uint32_t n = (i&1) ? neighbor_index<i>(p, x, y, z) : x+(y+z*p.Ny)*p.Px;
	buf[(size_t)BB*A+a] = fi[(size_t)plane*p.Np+n];
}
template<typename T, bool G, int DIR, int PM, int BB> __device__ __forceinline__ void insert_one(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z, const uint32_t a, const uint32_t A, const uint32_t t_odd, const T* __restrict__ buf, T* __restrict__ fi) {
	constexpr int i = G ? 2*DIR+PM+1 : TransferIndex<DIR, PM, BB>::value;
	const uint32_t plane = t_odd ? i : ((i&1) ? i+1 : i-1);
	const uint32_t n = (i&1) ? x+(y+z*p.Ny)*p.Px : neighbor_index<i-1>(p, x, y, z);
	fi[(size_t)plane*p.Np+n] = buf[(size_t)BB*A+a];
}
template<typename T, bool G, int DIR> __global__ __launch_bounds__(256) void k_extract_fi(const KParams p, const uint32_t A, const uint32_t t_odd, T* __restrict__ buf_p, T* __restrict__ buf_m, const T* __restrict__ fi) {
	const uint32_t t = blockIdx.x*blockDim.x+threadIdx.x;
	if(t>=A) return;
	const uint32_t Nd = DIR==0 ? p.Nx : DIR==1 ? p.Ny : p.Nz;
	uint32_t x, y, z, a;
	face_cell<DIR>(p, t, Nd-2u, x, y, z, a);
	extract_one<T, G, DIR, 0, 0>(p, x, y, z, a, A, t_odd, buf_p, fi);
	if constexpr(!G) { extract_one<T, G, DIR, 0, 1>(p, x, y, z, a, A, t_odd, buf_p, fi); extract_one<T, G, DIR, 0, 2>(p, x, y, z, a, A, t_odd, buf_p, fi); extract_one<T, G, DIR, 0, 3>(p, x, y, z, a, A, t_odd, buf_p, fi); extract_one<T, G, DIR, 0, 4>(p, x, y, z, a, A, t_odd, buf_p, fi); }
	face_cell<DIR>(p, t, 1u, x, y, z, a);
	extract_one<T, G, DIR, 1, 0>(p, x, y, z, a, A, t_odd, buf_m, fi);
	if constexpr(!G) { extract_one<T, G, DIR, 1, 1>(p, x, y, z, a, A, t_odd, buf_m, fi); extract_one<T, G, DIR, 1, 2>(p, x, y, z, a, A, t_odd, buf_m, fi); extract_one<T, G, DIR, 1, 3>(p, x, y, z, a, A, t_odd, buf_m, fi); extract_one<T, G, DIR, 1, 4>(p, x, y, z, a, A, t_odd, buf_m, fi); }
}
template<typename T, bool G, int DIR> __global__ __launch_bounds__(256) void k_insert_fi(const KParams p, const uint32_t A, const uint32_t t_odd, const T* __restrict__ buf_p, const T* __restrict__ buf_m, T* __restrict__ fi) {
	const uint32_t t = blockIdx.x*blockDim.x+threadIdx.x;
	if(t>=A) return;
	const uint32_t Nd = DIR==0 ? p.Nx : DIR==1 ? p.Ny : p.Nz;
	uint32_t x, y, z, a;
	face_cell<DIR>(p, t, Nd-1u, x, y, z, a);
	insert_one<T, G, DIR, 0, 0>(p, x, y, z, a, A, t_odd, buf_p, fi);
	if constexpr(!G) { insert_one<T, G, DIR, 0, 1>(p, x, y, z, a, A, t_odd, buf_p, fi); insert_one<T, G, DIR, 0, 2>(p, x, y, z, a, A, t_odd, buf_p, fi); insert_one<T, G, DIR, 0, 3>(p, x, y, z, a, A, t_odd, buf_p, fi); insert_one<T, G, DIR, 0, 4>(p, x, y, z, a, A, t_odd, buf_p, fi); }
	face_cell<DIR>(p, t, 0u, x, y, z, a);
	insert_one<T, G, DIR, 1, 0>(p, x, y, z, a, A, t_odd, buf_m, fi);
	if constexpr(!G) { insert_one<T, G, DIR, 1, 1>(p, x, y, z, a, A, t_odd, buf_m, fi); insert_one<T, G, DIR, 1, 2>(p, x, y, z, a, A, t_odd, buf_m, fi); insert_one<T, G, DIR, 1, 3>(p, x, y, z, a, A, t_odd, buf_m, fi); insert_one<T, G, DIR, 1, 4>(p, x, y, z, a, A, t_odd, buf_m, fi); }
}
// ---------------------------------------------------------------- mesh voxeliser
// voxelize_mesh with direction 2 (z rays; LUW always voxelises TYPE_S along z, FX/lbm.cpp:1427-1430) for a static mesh:
// one lane per (x,y) column casts a ray from the bottom of the padded bounding box through ALL triangles
// (Moeller-Trumbore), sorts up to 64 hit distances and fills the cells between odd/even crossings
// (FX/kernel.cpp:2381-2471).  Arithmetic mirrors what the reference's OpenCL build executes on this hardware: 1/g is the
// hardware reciprocal v_rcp_f32 (OpenCL's 2.5-ulp 1.0f/g) and dot / cross are the fma chains of the OpenCL device library
// (dot = mad(z,z', mad(y,y', x*x')), cross.x = mad(a.y, b.z, -(a.z*b.y)) ...).  Faces of LUW geometry sit on exact lattice
// planes by construction (ground slab pmin -> 1), where the (ushort)d truncation depends on exactly these roundings.
__device__ __forceinline__ float vdot(const float ax, const float ay, const float az, const float bx, const float by, const float bz) {
	return fmaf(az, bz, fmaf(ay, by, ax*bx)); // dot(float3) of the OpenCL device library: mad(z, z', mad(y, y', x*x'))
}
struct VoxGrid { uint32_t Nx, Ny, Nz, Px; int Ox, Oy, Oz; uint64_t Np; }; // lattice of the pass: a solver domain, or a bare global lattice (luw_voxelize_lattice)
// One block = one 16x16 tile of columns; it visits only the triangles binned to the tile (tile_start / tile_tri: CSR, triangle
// ids ascending, so hits are met in the reference's order and the 64-entry cut-off falls on the same hits).  The bins hold
// every triangle whose xy bounding box, grown by one cell, touches the tile -- the margin the reference itself uses when it
// hands a domain its triangle subset (FX/lbm.cpp:1455-1487).
constexpr uint32_t VOX_TILE = 16u;
__global__ __launch_bounds__(256) void k_voxelize_z(const VoxGrid p, uint8_t* __restrict__ flags, const float* __restrict__ u, const uint8_t flag, const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_tri,
		const float* __restrict__ p0, const float* __restrict__ p1, const float* __restrict__ p2, const float x0, const float y0, const float z0, const float x1, const float y1, const float z1) {
	const uint32_t x = blockIdx.x*VOX_TILE+threadIdx.x%VOX_TILE, y = blockIdx.y*VOX_TILE+threadIdx.x/VOX_TILE;
	if(x>=p.Nx||y>=p.Ny) return;
	const uint32_t tile = blockIdx.x+blockIdx.y*gridDim.x, k0 = tile_start[tile], k1 = tile_start[tile+1u];
	const int zs = min(max((int)z0-p.Oz, 0), (int)p.Nz-1);
	const float rx = (float)((int)x+p.Ox), ry = (float)((int)y+p.Oy), rz = (float)(zs+p.Oz); // position(xyz)+offset = global index coordinates
	if(rx<x0||ry<y0||rx>=x1||ry>=y1) return;
	uint32_t intersections = 0u, intersections_check = 0u;
	uint16_t distances[64];
	for(uint32_t kk=k0; kk<k1; kk++) {
		const uint32_t i = tile_tri[kk];
		const float ax = p0[3u*i], ay = p0[3u*i+1u], az = p0[3u*i+2u];
		const float ux = p1[3u*i]-ax, uy = p1[3u*i+1u]-ay, uz = p1[3u*i+2u]-az;
		const float vx = p2[3u*i]-ax, vy = p2[3u*i+1u]-ay, vz = p2[3u*i+2u]-az;
		const float wx = rx-ax, wy = ry-ay, wz = rz-az;
		// h = cross(r_direction, v) with r_direction = (0,0,1); q = cross(w, u)
		const float hx = fmaf(0.0f, vz, -(1.0f*vy)), hy = fmaf(1.0f, vx, -(0.0f*vz)), hz = fmaf(0.0f, vy, -(0.0f*vx));
		const float qx = fmaf(wy, uz, -(wz*uy)), qy = fmaf(wz, ux, -(wx*uz)), qz = fmaf(wx, uy, -(wy*ux));
		const float g = vdot(ux, uy, uz, hx, hy, hz);
		const float f = __builtin_amdgcn_rcpf(g);
		const float sv = f*vdot(wx, wy, wz, hx, hy, hz), tv = f*vdot(0.0f, 0.0f, 1.0f, qx, qy, qz), d = f*vdot(vx, vy, vz, qx, qy, qz);
		if(g!=0.0f&&sv>=0.0f&&sv<1.0f&&tv>=0.0f&&sv+tv<1.0f) {
			if(d>0.0f) { if(intersections<64u&&d<65536.0f) distances[intersections] = (uint16_t)d; intersections++; }
			else intersections_check++;
		}
	}
	const uint32_t ns = min(intersections, 64u);
	for(uint32_t i=1u; i<ns; i++) { // insertion sort
		const uint16_t t = distances[i];
		int j = (int)i-1;
		while(j>=0&&distances[j]>t) { distances[j+1] = distances[j]; j--; }
		distances[j+1] = t;
	}
	bool inside = (intersections%2u)&&(intersections_check%2u);
	uint32_t k = (intersections%2u)!=(intersections_check%2u);
	const uint32_t h0 = (uint32_t)zs;
	const uint32_t hmax = (uint32_t)min(max((int)z1-p.Oz, 0), (int)p.Nz);
	const uint32_t hmesh = h0+(ns>0u ? (uint32_t)distances[min(intersections-1u, 63u)] : 0u);
	for(uint32_t h=h0; h<hmax; h++) {
		while(k<intersections&&h>h0+(uint32_t)distances[min(k, 63u)]) { inside = !inside; k++; }
		inside = inside&&(k<intersections&&h<hmesh);
		const uint64_t n = (uint64_t)x+((uint64_t)y+(uint64_t)h*p.Ny)*p.Px;
		uint8_t fl = flags[n];
		if(inside) fl = (uint8_t)((fl&~TYPE_BO)|flag);
		else if((fl&TYPE_BO)==TYPE_S&&(!u||(u[n]==0.0f&&u[p.Np+n]==0.0f&&u[2ull*p.Np+n]==0.0f))) fl = (uint8_t)(fl&~flag); // was solid with the mesh's velocity (static: 0), FX/kernel.cpp:2451-2462
		flags[n] = fl;
	}
}

// ---------------------------------------------------------------- probe gather: u at a short list of cells -> packed [i][3]
__global__ void k_gather_u(const uint32_t count, const uint32_t* __restrict__ cell, const float* __restrict__ u, const size_t Np, float* __restrict__ out) {
	const uint32_t i = blockIdx.x*blockDim.x+threadIdx.x;
	if(i>=count) return;
	const uint32_t n = cell[i];
	out[3u*i] = u[n]; out[3u*i+1u] = u[Np+n]; out[3u*i+2u] = u[2ull*Np+n];
}

// ---------------------------------------------------------------- von-Karman synthetic-turbulence inlet
// vk_inlet_apply, FX/kernel.cpp:2495-2571: u[cell] = u_base + sigma * sum_m A_m cos(k_m.p + omega_m t + phi_m) on the inlet
// cells (TYPE_E: the collide step then relaxes them to f_eq(rho, u)).  One lane per inlet point; cosf is the same device
// library function (ocml) the reference's OpenCL build calls.
__global__ __launch_bounds__(256) void k_vk_inlet_apply(const uint32_t use_interp, const float t0, const float t1, const float alpha, const uint32_t P, const uint32_t M, const uint32_t V,
		const uint32_t* __restrict__ point_cell, const uint8_t* __restrict__ point_face, const float* __restrict__ point_data, const float* __restrict__ mode_data, float* __restrict__ u, const size_t Np) {
	const uint32_t i = blockIdx.x*blockDim.x+threadIdx.x;
	if(i>=P) return;
	const uint32_t n = point_cell[i];
	const uint32_t fid = (uint32_t)(point_face[i]&0x07u);
	const float px = point_data[i], py = point_data[(size_t)P+i], pz = point_data[2ull*P+i];
	const float ubx = point_data[3ull*P+i], uby = point_data[4ull*P+i], ubz = point_data[5ull*P+i];
	const float sigma = point_data[6ull*P+i];
	if(fid>=5u||!(sigma>0.0f)) { u[n] = ubx; u[Np+n] = uby; u[2ull*Np+n] = ubz; return; }
	const uint32_t fbase = fid*M;
	float qx = 0.0f, qy = 0.0f, qz = 0.0f;
	for(uint32_t m=0u; m<M; ++m) {
		const uint32_t idx = fbase+m;
		const float kx = mode_data[idx], ky = mode_data[(size_t)V+idx], kz = mode_data[2ull*V+idx], omega = mode_data[3ull*V+idx];
		const float Ax = mode_data[4ull*V+idx], Ay = mode_data[5ull*V+idx], Az = mode_data[6ull*V+idx];
		const float phix = mode_data[7ull*V+idx], phiy = mode_data[8ull*V+idx], phiz = mode_data[9ull*V+idx];
		const float phase0 = fmaf(kx, px, fmaf(ky, py, fmaf(kz, pz, omega*t0)));
		float vx = Ax*cosf(phase0+phix), vy = Ay*cosf(phase0+phiy), vz = Az*cosf(phase0+phiz);
		if(use_interp!=0u) {
			const float phase1 = fmaf(kx, px, fmaf(ky, py, fmaf(kz, pz, omega*t1)));
			const float vx1 = Ax*cosf(phase1+phix), vy1 = Ay*cosf(phase1+phiy), vz1 = Az*cosf(phase1+phiz);
			vx = fmaf(alpha, vx1-vx, vx); vy = fmaf(alpha, vy1-vy, vy); vz = fmaf(alpha, vz1-vz, vz);
		}
		qx += vx; qy += vy; qz += vz;
	}
	u[n] = fmaf(sigma, qx, ubx);
	u[Np+n] = fmaf(sigma, qy, uby);
	u[2ull*Np+n] = fmaf(sigma, qz, ubz);
}

// ---------------------------------------------------------------- on-device time averaging
// The reference downloads u,rho at every sampled step and runs Welford's update on the host
// (accumulate_from_buffers, FX/setup.cpp:4441-4488).  Same arithmetic, same operation order, on the device: mean and M2 of
// the three velocity components, mean of rho.  One lane per cell, x fastest.
__global__ __launch_bounds__(256) void k_stats_accumulate(const KParams p, const float inv_n, const float* __restrict__ rho, const float* __restrict__ u,
		float* __restrict__ avg_u, float* __restrict__ avg_rho, float* __restrict__ m2, const float* __restrict__ Tf, float* __restrict__ avg_T) {
	const uint32_t x = blockIdx.x*blockDim.x+threadIdx.x, y = blockIdx.y, z = blockIdx.z;
	if(x>=p.Nx) return;
	const uint32_t n = x+(y+z*p.Ny)*p.Px;
	// every array is streamed exactly once per sample: non-temporal accesses keep them out of the way of the step kernel's lines
	if(avg_T) { const float ta = ldg<true>(avg_T+n); stg<true>(avg_T+n, ta+(ldg<true>(Tf+n)-ta)*inv_n); } // FX/setup.cpp:4481-4484
	const size_t Np = p.Np;
	#pragma unroll
	for(int c=0; c<3; c++) {
		const float v = ldg<true>(u+c*Np+n);
		float mean = ldg<true>(avg_u+c*Np+n);
		const float delta = v-mean;
		mean += delta*inv_n;
		const float delta2 = v-mean;
		stg<true>(m2+c*Np+n, ldg<true>(m2+c*Np+n)+delta*delta2);
		stg<true>(avg_u+c*Np+n, mean);
	}
	const float r = ldg<true>(rho+n);
	const float ra = ldg<true>(avg_rho+n);
	stg<true>(avg_rho+n, ra+(r-ra)*inv_n);
}

// ---------------------------------------------------------------- self-check of the fast FP16C codec
// counts inputs for which the fast codec differs from the literal restatement of FX/kernel.cpp:864-875:
// all 2^16 codes (decode, compared as bit patterns) and every float bit pattern with |x| < 2^103 (encode; exponent field
// < 230 -- the codec's stated domain, luw_device.hpp)
__global__ __launch_bounds__(256) void k_codec_check(unsigned long long* __restrict__ mismatches) {
	const uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x, nth = gridDim.x*blockDim.x;
	unsigned long long bad = 0ull;
	for(uint32_t c=tid; c<65536u; c+=nth) bad += __float_as_uint(half_to_float_custom(c))!=__float_as_uint(half_to_float_custom_ref(c));
	for(unsigned long long v=tid; v<(1ull<<32); v+=nth) { if(((uint32_t)v&0x7F800000u)>=(230u<<23)) continue; const float x = __uint_as_float((uint32_t)v); bad += float_to_half_custom(x)!=float_to_half_custom_ref(x); }
	// the product kernel's 3-instruction encode (fp16c_encode19_hi_rtz_final): every bit pattern except NaNs, under RTZ; the
	// reference formula it is compared with is integer-only, so the mode does not touch it
	asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3" : "+v"(bad));
	for(unsigned long long v=tid; v<(1ull<<32); v+=nth) {
		const uint32_t b = (uint32_t)v;
		if((b&0x7F800000u)==0x7F800000u&&(b&0x007FFFFFu)!=0u) continue;
		bad += (fp16c_code_hi_in_rtz_mode(__uint_as_float(b))>>16)!=float_to_half_custom_ref(__uint_as_float(b));
	}
	asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(bad));
	if(bad) atomicAdd(mismatches, bad);
}

// =====================================================================================================
// host side
// =====================================================================================================
static thread_local std::string g_last_error;
static int fail(const int code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr) do { const hipError_t e_ = (expr); if(e_!=hipSuccess) return fail(LUW_ERR_DEVICE, std::string(#expr)+": "+hipGetErrorString(e_)); } while(0)

// FX/utilities.hpp:2603-2634,2741-2750 + strtof: the float the reference kernel sees after device_defines()
// printed it as 9-digit decimal text (FX/lbm.cpp:664,774,780).
static float literal_roundtrip(float x) {
	bool neg = false;
	if(x<0.0f) { neg = true; x = -x; }
	if(std::isnan(x)||std::isinf(x)) return neg ? -x : x;
	int exponent = 0;
	if(x>=10.0f) {
		if(x>=1E32f) { x *= 1E-32f; exponent += 32; }
		if(x>=1E16f) { x *= 1E-16f; exponent += 16; }
		if(x>= 1E8f) { x *=  1E-8f; exponent +=  8; }
		if(x>= 1E4f) { x *=  1E-4f; exponent +=  4; }
		if(x>= 1E2f) { x *=  1E-2f; exponent +=  2; }
		if(x>= 1E1f) { x *=  1E-1f; exponent +=  1; }
	}
	if(x>0.0f&&x<=1.0f) {
		if(x<1E-31f) { x *=  1E32f; exponent -= 32; }
		if(x<1E-15f) { x *=  1E16f; exponent -= 16; }
		if(x< 1E-7f) { x *=   1E8f; exponent -=  8; }
		if(x< 1E-3f) { x *=   1E4f; exponent -=  4; }
		if(x< 1E-1f) { x *=   1E2f; exponent -=  2; }
		if(x<  1E0f) { x *=   1E1f; exponent -=  1; }
	}
	uint32_t integral = (uint32_t)x;
	const float remainder = (x-(float)integral)*1E8f;
	uint32_t decimal = (uint32_t)remainder;
	if(remainder-(float)decimal>=0.5f) {
		decimal++;
		if(decimal>=100000000u) { decimal = 0u; integral++; if(integral>=10u) { integral = 1u; exponent++; } }
	}
	char text[64];
	if(exponent!=0) snprintf(text, sizeof(text), "%s%u.%08uE%d", neg ? "-" : "", integral, decimal, exponent);
	else snprintf(text, sizeof(text), "%s%u.%08u", neg ? "-" : "", integral, decimal);
	return strtof(text, nullptr);
}

struct luw_solver {
	luw_config cfg;
	KParams kp;
	uint64_t N = 0;          // Nx*Ny*Nz
	uint64_t t = 0;
	bool initialized = false;
	bool fields_current = true; // device rho,u reflect the state after the last executed step
	size_t ddf_bytes = 4;
	void* d_fi = nullptr;
	float* d_rho = nullptr; float* d_u = nullptr; uint8_t* d_flags = nullptr; float* d_F = nullptr;
	float* d_wbuf = nullptr; float* d_sigma = nullptr;
	float* d_avg_u = nullptr; float* d_avg_rho = nullptr; float* d_m2 = nullptr; uint64_t avg_count = 0ull;
	uint32_t vk_P = 0u, vk_M = 0u; int vk_stride = 1; bool vk_interp = false, vk_active = false; uint64_t vk_last_t = ~0ull;
	uint32_t* d_vk_cell = nullptr; uint8_t* d_vk_face = nullptr; float* d_vk_point = nullptr; float* d_vk_mode = nullptr;
	float* h_rho = nullptr; float* h_u = nullptr; uint8_t* h_flags = nullptr; float* h_F = nullptr;
	void* d_gi = nullptr; float* d_T = nullptr; float* h_T = nullptr; float* d_avg_T = nullptr; // TEMPERATURE
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	uint32_t kernel = LUW_KERNEL_AUTO;
	std::vector<void*> raw; // hipMalloc'ed blocks behind the lattice-sized arrays (lead_alloc)
	uint32_t gather_count = 0u; uint32_t* d_gather_cell = nullptr; float* d_gather_out = nullptr; // probe columns
};

// Lattice-sized device arrays start LEAD elements into their allocation, LEAD = 64 - halo_x: with the x pitch a multiple of
// 64 elements, the first OWNED cell of every row (x = halo_x) then begins a 256-byte (FP32) / 128-byte (FP16C) block, so
// that a wave's 64 consecutive cells are exactly the lines it touches -- in a halo'ed domain (Nx = 512 + 2) just as in a
// single one.  Measured on MI355X before this: interior kernel of a 514x514x512 domain 7.3 ms vs 3.5 ms for 512^3.
static hipError_t lead_alloc(luw_solver* s, void** base, const size_t elems, const size_t elem_bytes) {
	void* r = nullptr;
	const size_t lead = (size_t)(64u-s->kp.halo_x)*elem_bytes, total = elems*elem_bytes+64u*elem_bytes;
	hipError_t e = hipMalloc(&r, total);
	if(e!=hipSuccess) return e;
	s->raw.push_back(r);
	e = hipMemsetAsync(r, 0, total, s->stream); // padding / not-yet-uploaded memory must hold defined values
	*base = (char*)r+lead;
	return e;
}

static int set_device(const luw_solver* s) { HIP_TRY(hipSetDevice(s->cfg.device)); return LUW_OK; }

static int copy_pitched(void* dst, const void* src, const size_t elem, const luw_solver* s, const uint32_t planes, const bool to_device, hipStream_t st) {
	const size_t rows = (size_t)s->cfg.Ny*s->cfg.Nz;
	for(uint32_t c=0u; c<planes; c++) {
		if(to_device) HIP_TRY(hipMemcpy2DAsync((char*)dst+(size_t)c*s->kp.Np*elem, (size_t)s->kp.Px*elem, (const char*)src+(size_t)c*s->N*elem, (size_t)s->cfg.Nx*elem, (size_t)s->cfg.Nx*elem, rows, hipMemcpyHostToDevice, st));
		else HIP_TRY(hipMemcpy2DAsync((char*)dst+(size_t)c*s->N*elem, (size_t)s->cfg.Nx*elem, (const char*)src+(size_t)c*s->kp.Np*elem, (size_t)s->kp.Px*elem, (size_t)s->cfg.Nx*elem, rows, hipMemcpyDeviceToHost, st));
	}
	return LUW_OK;
}

template<typename T, int V> static void launch_vec(luw_solver* s, const Box& b, const int write_fields) {
	T* fi = (T*)s->d_fi;
	const bool odd = (s->t&1ull)!=0ull;
	const uint32_t nvec = (b.x1-1u)/V-b.x0/V+1u;          // vectors overlapping [x0,x1)
	uint32_t vx = 1u; while(vx<nvec&&vx<256u) vx <<= 1;   // power of two
	const uint32_t ry = 256u/vx;
	const uint32_t nchunk = (nvec+vx-1u)/vx;
	const uint32_t rows = (b.y1-b.y0)*(b.z1-b.z0);
	const dim3 grid(((rows+ry-1u)/ry)*nchunk), block(vx, ry);
	if(odd) hipLaunchKernelGGL((k_stream_collide_v<T, V, 1>), grid, block, 0, s->stream, s->kp, b, nchunk, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
	else hipLaunchKernelGGL((k_stream_collide_v<T, V, 0>), grid, block, 0, s->stream, s->kp, b, nchunk, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
}
// threads per block for a row of nx lanes: whole waves, at most 256, chosen so that the blocks of a row carry the fewest idle
// lanes (a 375-lane row of the pair kernel: 3 x 128 instead of 2 x 256; ties go to the larger block)
static uint32_t row_block(const uint32_t nx) {
	if(nx<=256u) return ((nx+63u)/64u)*64u;
	uint32_t best = 256u, best_lanes = ((nx+255u)/256u)*256u;
	for(uint32_t bx : {192u, 128u, 64u}) { const uint32_t lanes = ((nx+bx-1u)/bx)*bx; if(lanes<best_lanes) { best = bx; best_lanes = lanes; } }
	return best;
}
template<typename T> static void launch_scalar(luw_solver* s, const Box& b, const int write_fields) {
	T* fi = (T*)s->d_fi;
	const bool odd = (s->t&1ull)!=0ull;
	const int xa = (int)b.x0-(int)((b.x0+64u-s->kp.halo_x)&63u); // block start of the line that holds b.x0 (see lead_alloc)
	const uint32_t nx = (uint32_t)((int)b.x1-xa);
	const uint32_t bx = row_block(nx);
	const dim3 grid((nx+bx-1u)/bx, b.y1-b.y0, b.z1-b.z0), block(bx);
	const int mode = s->kernel==LUW_KERNEL_EXP_COPY ? 1 : s->kernel==LUW_KERNEL_EXP_NOSHIFT ? 2 : s->kernel==LUW_KERNEL_SCALAR_CACHED ? 3 : s->kernel==LUW_KERNEL_SCALAR_NT_ALL ? 4 : s->kernel==LUW_KERNEL_SCALAR_GENERAL ? 5 : 0;
	// FLAT addressing (one 32-bit byte offset per neighbour within a plane) for FP32 lattices whose planes fit it; the row form otherwise
	static const bool force_row = getenv("LUW_ADDR_ROW")!=nullptr;   // test aid: the row form also where the flat form would be valid
	const bool flat = sizeof(T)==4u && (uint64_t)s->kp.Np*sizeof(T)<=(1ull<<32) && !force_row;
	#define LUW_LAUNCH_SF(PAR, MODE, NT, FL) hipLaunchKernelGGL((k_stream_collide_s<T, PAR, MODE, NT, FL>), grid, block, 0, s->stream, s->kp, b, xa, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields)
	#define LUW_LAUNCH_S(PAR, MODE, NT) do { if constexpr(sizeof(T)==4u) { if(flat) LUW_LAUNCH_SF(PAR, MODE, NT, true); else LUW_LAUNCH_SF(PAR, MODE, NT, false); } else LUW_LAUNCH_SF(PAR, MODE, NT, false); } while(0)
	if(s->d_gi) { // thermal lattice on: the product kernel plus the D3Q7 cell update
		#define LUW_LAUNCH_T(PAR, FL) hipLaunchKernelGGL((k_stream_collide_s<T, PAR, 4, 2, FL>), grid, block, 0, s->stream, s->kp, b, xa, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields, (T*)s->d_gi, s->d_T)
		if constexpr(sizeof(T)==4u) { if(flat) { if(odd) LUW_LAUNCH_T(1, true); else LUW_LAUNCH_T(0, true); } else { if(odd) LUW_LAUNCH_T(1, false); else LUW_LAUNCH_T(0, false); } }
		else { if(odd) LUW_LAUNCH_T(1, false); else LUW_LAUNCH_T(0, false); }
		#undef LUW_LAUNCH_T
	}
	else if(mode==0) { if(odd) LUW_LAUNCH_S(1, 0, 2); else LUW_LAUNCH_S(0, 0, 2); }
	else if(mode==1) { if(odd) LUW_LAUNCH_S(1, 1, 1); else LUW_LAUNCH_S(0, 1, 1); }
	else if(mode==2) { if(odd) LUW_LAUNCH_S(1, 2, 1); else LUW_LAUNCH_S(0, 2, 1); }
	else if(mode==3) { if(odd) LUW_LAUNCH_S(1, 0, 0); else LUW_LAUNCH_S(0, 0, 0); }
	else if(mode==5) { if(odd) LUW_LAUNCH_S(1, 3, 2); else LUW_LAUNCH_S(0, 3, 2); }
	else { if(odd) LUW_LAUNCH_S(1, 0, 1); else LUW_LAUNCH_S(0, 0, 1); }
	#undef LUW_LAUNCH_SF
	#undef LUW_LAUNCH_S
}

static void launch_pair(luw_solver* s, const Box& b, const int write_fields) {
	uint16_t* fi = (uint16_t*)s->d_fi;
	const bool odd = (s->t&1ull)!=0ull;
	const uint32_t nx = (b.x1-b.x0+1u)/2u;                         // an odd count only when the box ends at an odd Nx: the last lane owns one cell
	const uint32_t bx = row_block(nx);
	const dim3 grid((nx+bx-1u)/bx, b.y1-b.y0, b.z1-b.z0), block(bx);
	static const bool copy_only = getenv("LUW_PAIR_COPY")!=nullptr;   // measurement aid: the kernel's memory path alone (no physics)
	if(copy_only) { if(odd) hipLaunchKernelGGL((k_stream_collide_p<1, 1>), grid, block, 0, s->stream, s->kp, b, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
		else hipLaunchKernelGGL((k_stream_collide_p<0, 1>), grid, block, 0, s->stream, s->kp, b, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields); return; }
	if(odd) hipLaunchKernelGGL((k_stream_collide_p<1>), grid, block, 0, s->stream, s->kp, b, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
	else hipLaunchKernelGGL((k_stream_collide_p<0>), grid, block, 0, s->stream, s->kp, b, fi, s->d_rho, s->d_u, s->d_flags, s->d_F, write_fields);
}

// Kernel choice.  LUW_KERNEL_AUTO = the scalar kernel (FP32: 39.5k MLUPS at 512^3; vector kernels 20-29k) and, for FP16C rows
// wide enough, the pair kernel (profiles/r01_kernel_ab.md).  The other kernels stay selectable for A/B runs.
static int launch_stream_collide(luw_solver* s, const Box& b, const int write_fields) {
	if(b.x0>=b.x1||b.y0>=b.y1||b.z0>=b.z1) return LUW_OK; // empty box
	if(b.x1>s->cfg.Nx||b.y1>s->cfg.Ny||b.z1>s->cfg.Nz) return fail(LUW_ERR_INVALID, "stream_collide: box exceeds the local lattice");
	if(b.y1-b.y0>65535u||b.z1-b.z0>65535u) return fail(LUW_ERR_INVALID, "stream_collide: box too large for the launch geometry");
	const bool fp16 = s->ddf_bytes==2u;
	uint32_t k = s->kernel;
	// AUTO: the scalar kernel, except FP16C rows of at least two waves of pairs, which take the pair kernel (dword accesses, packed
	// FP32 collision: 69.0k vs 67.2k MLUPS at 512^3, 63.1k vs 61.1k with Coriolis)
	if(k==LUW_KERNEL_AUTO) k = (fp16 && b.x1-b.x0>=256u) ? LUW_KERNEL_PAIR : LUW_KERNEL_SCALAR;
	if(s->d_gi) k = LUW_KERNEL_SCALAR; // the thermal cell update lives in the scalar kernel only
	if(s->kp.halo_x&&(k==LUW_KERNEL_VEC4||k==LUW_KERNEL_VEC2||k==LUW_KERNEL_VEC1)) k = LUW_KERNEL_SCALAR; // the vector kernels assume rows that start on a 16-byte boundary at x = 0
	// pair kernel: FP16C; pairs start on a 4-byte boundary -- at even x, or at odd x when x is split (the row's lead pad then puts
	// x = 1 on a line start, lead_alloc); the range holds whole pairs, except that it may end at an odd Nx of an unsplit row (the
	// last cell then pairs with the row padding)
	if(k==LUW_KERNEL_PAIR) {
		const bool starts_aligned = ((b.x0+s->kp.halo_x)&1u)==0u;
		const bool whole_pairs = ((b.x1-b.x0)&1u)==0u || (!s->kp.halo_x && b.x1==s->cfg.Nx);
		if(!fp16||!starts_aligned||!whole_pairs) k = LUW_KERNEL_SCALAR;
	}
	if(k==LUW_KERNEL_PAIR) launch_pair(s, b, write_fields);
	else if(k==LUW_KERNEL_VEC4) { if(fp16) launch_vec<uint16_t, 4>(s, b, write_fields); else launch_vec<float, 4>(s, b, write_fields); }
	else if(k==LUW_KERNEL_VEC2) { if(fp16) launch_vec<uint16_t, 2>(s, b, write_fields); else launch_vec<float, 2>(s, b, write_fields); }
	else if(k==LUW_KERNEL_VEC1) { if(fp16) launch_vec<uint16_t, 1>(s, b, write_fields); else launch_vec<float, 1>(s, b, write_fields); }
	else { if(fp16) launch_scalar<uint16_t>(s, b, write_fields); else launch_scalar<float>(s, b, write_fields); }
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}

// launch helper: picks the template instance for (storage type, lattice, direction)
template<bool G, bool INSERT> static void launch_transfer(luw_solver* s, const uint32_t direction, void* buf_p, void* buf_m) {
	const uint32_t A = (uint32_t)luw_get_area(s, direction);
	const dim3 grid((A+255u)/256u), block(256);
	const uint32_t odd = (uint32_t)(s->t&1ull);
	void* lat = G ? s->d_gi : s->d_fi;
	#define LUW_TR(TT, DD) do { if constexpr(INSERT) hipLaunchKernelGGL((k_insert_fi<TT, G, DD>), grid, block, 0, s->stream, s->kp, A, odd, (const TT*)buf_p, (const TT*)buf_m, (TT*)lat); \
		else hipLaunchKernelGGL((k_extract_fi<TT, G, DD>), grid, block, 0, s->stream, s->kp, A, odd, (TT*)buf_p, (TT*)buf_m, (const TT*)lat); } while(0)
	if(s->ddf_bytes==2u) { if(direction==0u) LUW_TR(uint16_t, 0); else if(direction==1u) LUW_TR(uint16_t, 1); else LUW_TR(uint16_t, 2); }
	else { if(direction==0u) LUW_TR(float, 0); else if(direction==1u) LUW_TR(float, 1); else LUW_TR(float, 2); }
	#undef LUW_TR
}

// Where the driver places the DDF array physically changes the step time of this 19-stream kernel by up to 10 % on MI355X:
// allocations of the same size at the same virtual address come out in two classes (512^3 FP32: 3.40-3.50 ms vs 3.70-3.85 ms
// per step, persistent for the life of the allocation; tools/placement_probe.py).  Nothing in the HIP API selects the class,
// so large solvers allocate a few candidates, time the real kernel on each (zero DDFs = rest state, flags 0 = all fluid: a
// valid, full-cost step) and keep the fastest.  LUW_TUNE_PLACEMENT=0 disables it; skipped when memory is short.
static int tune_ddf_placement(luw_solver* s) {
	const size_t bytes = 19ull*s->kp.Np*s->ddf_bytes;
	const char* env = getenv("LUW_TUNE_PLACEMENT");
	int candidates = env ? atoi(env) : 6;
	if(bytes<(1ull<<30)||candidates<2) return LUW_OK;
	const Box whole = { 0u, s->cfg.Nx, 0u, s->cfg.Ny, 0u, s->cfg.Nz };
	hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
	auto step_ms = [&](float& ms) -> int { // two steps (both parities) after one untimed
		s->initialized = true; s->t = 0ull;
		if(int e = launch_stream_collide(s, whole, 0)) return e;
		s->t = 1ull;
		HIP_TRY(hipEventRecord(e0, s->stream));
		if(int e = launch_stream_collide(s, whole, 0)) return e;
		s->t = 2ull;
		if(int e = launch_stream_collide(s, whole, 0)) return e;
		HIP_TRY(hipEventRecord(e1, s->stream));
		HIP_TRY(hipEventSynchronize(e1));
		HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
		s->initialized = false; s->t = 0ull;
		return LUW_OK;
	};
	float best_ms = 0.0f;
	if(int e = step_ms(best_ms)) return e;
	void* best_raw = s->raw.front(); void* best_fi = s->d_fi; // fi is the first lead_alloc of luw_create
	std::vector<void*> losers;
	// FP32 (HBM-bound): a placement of the fast class moves >= 6.0 TB/s of algorithmic bytes in this probe; while none has been
	// found the search goes on past the nominal count, up to twice as many candidates (memory permitting)
	const double probe_bytes = 2.0*153.0*(double)s->cfg.Nx*(double)s->cfg.Ny*(double)s->cfg.Nz;
	auto fast_class = [&](const float ms) { return s->ddf_bytes!=4u || probe_bytes/((double)ms*1e-3)>=6.0e12; };
	for(int k=1; k<2*candidates; k++) {
		if(k>=candidates&&fast_class(best_ms)) break;
		size_t free_b = 0u, total_b = 0u;
		if(hipMemGetInfo(&free_b, &total_b)!=hipSuccess||free_b<bytes+(8ull<<30)) break; // keep headroom for the rest of the run
		void* fi = nullptr;
		if(lead_alloc(s, &fi, 19ull*s->kp.Np, s->ddf_bytes)!=hipSuccess) { (void)hipGetLastError(); break; }
		void* raw = s->raw.back(); s->raw.pop_back();
		s->d_fi = fi;
		float ms = 0.0f;
		if(int e = step_ms(ms)) { (void)hipFree(raw); s->d_fi = best_fi; return e; }
		if(getenv("LUW_TUNE_VERBOSE")) fprintf(stderr, "luw: placement candidate %d: %.3f ms per 2 steps (best so far %.3f)\n", k, ms, best_ms);
		if(ms<best_ms) { losers.push_back(best_raw); best_ms = ms; best_raw = raw; best_fi = fi; }
		else losers.push_back(raw);
	}
	for(void* r : losers) (void)hipFree(r); // released only now, so that no candidate reuses the pages of another
	s->raw.front() = best_raw; s->d_fi = best_fi;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	HIP_TRY(hipMemsetAsync(best_raw, 0, bytes+64u*s->ddf_bytes, s->stream)); // the probe steps left zeros, but be explicit
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

extern "C" {

int luw_abi_version(void) { return LUW_ABI_VERSION; }
const char* luw_last_error(void) { return g_last_error.c_str(); }
int luw_device_count(int* count) {
	if(!count) return fail(LUW_ERR_INVALID, "luw_device_count: null argument");
	HIP_TRY(hipGetDeviceCount(count));
	return LUW_OK;
}

void luw_destroy(luw_solver* s) {
	if(!s) return;
	(void)hipSetDevice(s->cfg.device);
	if(s->own_stream) (void)hipStreamSynchronize(s->own_stream);
	for(void* r : s->raw) (void)hipFree(r); // fi, rho, u, flags, F, statistics
	(void)hipFree(s->d_wbuf); (void)hipFree(s->d_sigma);
	(void)hipFree(s->d_gather_cell); (void)hipFree(s->d_gather_out);
	(void)hipFree(s->d_vk_cell); (void)hipFree(s->d_vk_face); (void)hipFree(s->d_vk_point); (void)hipFree(s->d_vk_mode);
	(void)hipHostFree(s->h_rho); (void)hipHostFree(s->h_u); (void)hipHostFree(s->h_flags); (void)hipHostFree(s->h_F); (void)hipHostFree(s->h_T);
	if(s->own_stream) (void)hipStreamDestroy(s->own_stream);
	delete s;
}

int luw_create(const luw_config* cfg, luw_solver** out) {
	if(!cfg||!out) return fail(LUW_ERR_INVALID, "luw_create: null argument");
	*out = nullptr;
	if(cfg->struct_size!=sizeof(luw_config)) return fail(LUW_ERR_INVALID, "luw_create: luw_config size mismatch (ABI)");
	if((uint64_t)cfg->Nx*cfg->Ny*cfg->Nz==0ull) return fail(LUW_ERR_INVALID, "Grid point number is 0."); // FX/lbm.cpp:1123
	if(cfg->Dx*cfg->Dy*cfg->Dz==0u) return fail(LUW_ERR_INVALID, "You specified 0 LBM grid domains."); // FX/lbm.cpp:1124
	if(cfg->nu==0.0f) return fail(LUW_ERR_INVALID, "Viscosity cannot be 0."); // FX/lbm.cpp:1141
	if(cfg->nu<0.0f) return fail(LUW_ERR_INVALID, "Viscosity cannot be negative."); // FX/lbm.cpp:1142
	if(cfg->ddf_format!=LUW_DDF_FP32&&cfg->ddf_format!=LUW_DDF_FP16C) return fail(LUW_ERR_INVALID, "luw_create: unknown ddf_format");
	if((cfg->Dx>1u&&cfg->Nx<3u)||(cfg->Dy>1u&&cfg->Ny<3u)||(cfg->Dz>1u&&cfg->Nz<3u)) return fail(LUW_ERR_INVALID, "luw_create: split axes need at least one interior cell between the halo layers");
	if((cfg->options&LUW_OPT_TEMPERATURE)&&!(cfg->alpha>=0.0f)) return fail(LUW_ERR_INVALID, "luw_create: thermal diffusivity must not be negative");
	if(cfg->buffer_nudging_active&&cfg->buffer_n_cells==0u) return fail(LUW_ERR_INVALID, "luw_create: buffer_n_cells must be > 0");
	if(cfg->top_sponge_active&&cfg->sponge_n_cells==0u) return fail(LUW_ERR_INVALID, "luw_create: sponge_n_cells must be > 0");
	const uint32_t Px = (cfg->Nx+63u)&~63u; // rows are whole 256-byte blocks (see lead_alloc)
	// plane stride: the lattice plus a skew of 33 line blocks (8448 B in FP32).  With a bare power-of-two stride the 19 planes of
	// a cell sit at the same offset of 19 equally aligned regions; the skew measured 2-4 % faster on every lattice shape tried
	// (512^3 3.50 -> 3.40 ms, 768x768x256 3.87 -> 3.70 ms, 1024x1024x256 6.80 -> 6.69 ms; odd small multiples behave alike).
	const uint64_t Np = (uint64_t)Px*cfg->Ny*cfg->Nz+64ull*33ull;
	if(Np>=(1ull<<32)) return fail(LUW_ERR_INVALID, "luw_create: more than 2^32 (padded) cells per domain are not supported (32-bit cell indices)");
	int ndev = 0;
	HIP_TRY(hipGetDeviceCount(&ndev));
	if(cfg->device<0||cfg->device>=ndev) return fail(LUW_ERR_INVALID, "luw_create: no such HIP device"); // FX/lbm.cpp:961-979
	HIP_TRY(hipSetDevice(cfg->device));

	luw_solver* s = new luw_solver();
	s->cfg = *cfg;
	s->N = (uint64_t)cfg->Nx*cfg->Ny*cfg->Nz;
	s->ddf_bytes = cfg->ddf_format==LUW_DDF_FP16C ? 2u : 4u;
	s->kernel = cfg->kernel;
	if(const char* ke = getenv("LUW_KERNEL")) s->kernel = (uint32_t)atoi(ke); // test / A-B aid: overrides the kernel choice of callers that expose none (the deck driver)
	KParams& k = s->kp;
	memset(&k, 0, sizeof(k));
	k.Nx = cfg->Nx; k.Ny = cfg->Ny; k.Nz = cfg->Nz; k.Px = Px; k.Np = (uint32_t)Np;
	k.halo_x = cfg->Dx>1u; k.halo_y = cfg->Dy>1u; k.halo_z = cfg->Dz>1u;
	k.Ox = cfg->Ox; k.Oy = cfg->Oy; k.Oz = cfg->Oz;
	k.w = literal_roundtrip(1.0f/(3.0f*cfg->nu+0.5f)); // FX/lbm.hpp:140, FX/lbm.cpp:664
	k.fx = cfg->fx; k.fy = cfg->fy; k.fz = cfg->fz;
	k.tau0 = 1.0f/k.w; k.tau0sq = k.tau0*k.tau0;
	k.omx = cfg->omega_x; k.omy = cfg->omega_y; k.omz = cfg->omega_z; k.coriolis = k.omx!=0.0f||k.omy!=0.0f||k.omz!=0.0f;
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
	k.has_F = (cfg->options&LUW_OPT_FORCE_FIELD) ? 1u : 0u;
	k.w_T = (cfg->options&LUW_OPT_TEMPERATURE) ? literal_roundtrip(1.0f/(2.0f*cfg->alpha+0.5f)) : 0.0f; // FX/lbm.cpp:750

	auto oom = [&](const char* what) { luw_destroy(s); return fail(LUW_ERR_NOMEM, std::string("luw_create: allocation failed: ")+what); };
	if(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking)!=hipSuccess) return oom("stream");
	s->stream = s->own_stream;
	// (memset inside lead_alloc runs on the solver's own non-blocking stream: the legacy NULL stream does not order against it)
	if(lead_alloc(s, &s->d_fi, 19ull*Np, s->ddf_bytes)!=hipSuccess) return oom("fi");
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
		if(hipMalloc((void**)&s->d_wbuf, wb.size()*4u)!=hipSuccess||hipMemcpy(s->d_wbuf, wb.data(), wb.size()*4u, hipMemcpyHostToDevice)!=hipSuccess) return oom("wbuf");
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
		if(hipMalloc((void**)&s->d_sigma, sg.size()*4u)!=hipSuccess||hipMemcpy(s->d_sigma, sg.data(), sg.size()*4u, hipMemcpyHostToDevice)!=hipSuccess) return oom("sigma");
		k.sigma = s->d_sigma;
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	if(int e = tune_ddf_placement(s)) { luw_destroy(s); return e; } // last: the probe steps run the complete kernel (nudging / sponge tables included)
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
	if(int e = copy_pitched(s->d_fi, host_src, s->ddf_bytes, s, 19u, true, s->stream)) return e;
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

int luw_run(luw_solver* s, uint64_t steps);
int luw_upload(luw_solver* s, uint32_t mask);
int luw_download(luw_solver* s, uint32_t mask);
static int vk_apply(luw_solver* s);
static int voxelize_launch(const VoxGrid& vg, uint8_t* d_flags, const float* d_u, const uint8_t flag, const uint32_t ntri, const float* p0, const float* p1, const float* p2, const float* const d[3], const float* pmin, const float* pmax, hipStream_t st);
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
	for(int k=0; k<3; k++) { if(hipMalloc((void**)&d[k], 12ull*triangle_number)!=hipSuccess) { for(int q=0; q<3; q++) (void)hipFree(d[q]); return fail(LUW_ERR_NOMEM, "luw_voxelize_mesh: allocation failed"); } HIP_TRY(hipMemcpy(d[k], h[k], 12ull*triangle_number, hipMemcpyHostToDevice)); }
	if(int e = luw_upload(s, LUW_MASK_FLAGS|LUW_MASK_U)) return e; // the host mirror is authoritative before the first run
	const VoxGrid vg = { s->kp.Nx, s->kp.Ny, s->kp.Nz, s->kp.Px, s->kp.Ox, s->kp.Oy, s->kp.Oz, (uint64_t)s->kp.Np };
	const int rc = voxelize_launch(vg, s->d_flags, s->d_u, flag, triangle_number, p0, p1, p2, d, pmin, pmax, s->stream);
	for(int k=0; k<3; k++) (void)hipFree(d[k]);
	if(rc!=LUW_OK) return rc;
	return luw_download(s, LUW_MASK_FLAGS); // LBM::voxelize_mesh_on_device leaves the result in lbm.flags
}

// bins + launch shared by luw_voxelize_mesh (a solver's domain) and luw_voxelize_lattice (bare lattice)
static int voxelize_launch(const VoxGrid& vg, uint8_t* d_flags, const float* d_u, const uint8_t flag, const uint32_t ntri, const float* p0, const float* p1, const float* p2, const float* const d[3], const float* pmin, const float* pmax, hipStream_t st) {
	const uint32_t tx = (vg.Nx+VOX_TILE-1u)/VOX_TILE, ty = (vg.Ny+VOX_TILE-1u)/VOX_TILE;
	const bool brute = getenv("LUW_VOXELIZE_ALL_TRIANGLES")!=nullptr; // test aid: every tile sees every triangle
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
	for(uint32_t i=0u; i<ntri; i++) { int a0, a1, b0, b1; range(i, a0, a1, b0, b1); for(int b=b0; b<=b1; b++) for(int a=a0; a<=a1; a++) start[(size_t)a+(size_t)b*tx+1u]++; }
	for(size_t t=0u; t<(size_t)tx*ty; t++) { if((uint64_t)start[t]+start[t+1u]>0xFFFFFFFFull) return fail(LUW_ERR_INVALID, "voxelize: triangle bins exceed 2^32 entries"); start[t+1u] += start[t]; }
	tri.resize(std::max<size_t>(start.back(), 1u));
	{ std::vector<uint32_t> fill(start.begin(), start.end()-1);
	  for(uint32_t i=0u; i<ntri; i++) { int a0, a1, b0, b1; range(i, a0, a1, b0, b1); for(int b=b0; b<=b1; b++) for(int a=a0; a<=a1; a++) tri[fill[(size_t)a+(size_t)b*tx]++] = i; } }
	uint32_t* d_start = nullptr; uint32_t* d_tri = nullptr;
	if(hipMalloc((void**)&d_start, 4ull*start.size())!=hipSuccess||hipMalloc((void**)&d_tri, 4ull*tri.size())!=hipSuccess) { (void)hipFree(d_start); (void)hipFree(d_tri); return fail(LUW_ERR_NOMEM, "voxelize: allocation failed"); }
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
	if(hipMalloc((void**)&s->d_gather_cell, 4ull*count)!=hipSuccess||hipMalloc((void**)&s->d_gather_out, 12ull*count)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_gather_attach: allocation failed");
	HIP_TRY(hipMemcpy(s->d_gather_cell, c.data(), 4ull*count, hipMemcpyHostToDevice));
	s->gather_count = count;
	return LUW_OK;
}
int luw_gather_u(luw_solver* s, float* out) {
	if(!s||!out) return fail(LUW_ERR_INVALID, "luw_gather_u: bad argument");
	if(s->gather_count==0u) return LUW_OK;
	if(!s->fields_current) return fail(LUW_ERR_STATE, "luw_gather_u: rho,u on the device are stale (the last step did not write fields)");
	if(int e = set_device(s)) return e;
	hipLaunchKernelGGL(k_gather_u, dim3((s->gather_count+255u)/256u), dim3(256), 0, s->stream, s->gather_count, s->d_gather_cell, s->d_u, (size_t)s->kp.Np, s->d_gather_out);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(out, s->d_gather_out, 12ull*s->gather_count, hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return LUW_OK;
}

int luw_voxelize_lattice(int device, uint32_t Nx, uint32_t Ny, uint32_t Nz, uint32_t triangle_number, const float* p0, const float* p1, const float* p2, const float* bounds, uint8_t flag, uint8_t* flags) {
	if(!p0||!p1||!p2||!bounds||!flags||triangle_number==0u||(uint64_t)Nx*Ny*Nz==0ull||(uint64_t)Nx*Ny>0xFFFFFF00ull) return fail(LUW_ERR_INVALID, "luw_voxelize_lattice: bad argument");
	HIP_TRY(hipSetDevice(device));
	const uint64_t N = (uint64_t)Nx*Ny*Nz;
	uint8_t* d_flags = nullptr; float* d[3] = { nullptr, nullptr, nullptr };
	auto cleanup = [&]() { (void)hipFree(d_flags); for(int k=0; k<3; k++) (void)hipFree(d[k]); };
	if(hipMalloc((void**)&d_flags, N)!=hipSuccess) return fail(LUW_ERR_NOMEM, "luw_voxelize_lattice: allocation failed");
	const float* h[3] = { p0, p1, p2 };
	for(int k=0; k<3; k++) if(hipMalloc((void**)&d[k], 12ull*triangle_number)!=hipSuccess||hipMemcpy(d[k], h[k], 12ull*triangle_number, hipMemcpyHostToDevice)!=hipSuccess) { cleanup(); return fail(LUW_ERR_NOMEM, "luw_voxelize_lattice: allocation failed"); }
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
	(void)hipFree(s->d_vk_cell); (void)hipFree(s->d_vk_face); (void)hipFree(s->d_vk_point); (void)hipFree(s->d_vk_mode);
	s->d_vk_cell = nullptr; s->d_vk_face = nullptr; s->d_vk_point = nullptr; s->d_vk_mode = nullptr;
	s->vk_active = false; s->vk_P = s->vk_M = 0u;
	return LUW_OK;
}
int luw_vk_inlet_attach(luw_solver* s, uint64_t point_count, uint64_t mode_count, const uint64_t* point_cell, const uint8_t* point_face, const float* point_data, const float* mode_data, int update_stride, int stride_interpolation) {
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
	if(hipMalloc((void**)&s->d_vk_cell, P*4u)!=hipSuccess||hipMalloc((void**)&s->d_vk_face, P)!=hipSuccess||hipMalloc((void**)&s->d_vk_point, 7ull*P*4u)!=hipSuccess||hipMalloc((void**)&s->d_vk_mode, 10ull*V*4u)!=hipSuccess)
		return fail(LUW_ERR_NOMEM, "luw_vk_inlet_attach: allocation failed");
	HIP_TRY(hipMemcpy(s->d_vk_cell, cell.data(), P*4u, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(s->d_vk_face, point_face, P, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(s->d_vk_point, point_data, 7ull*P*4u, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(s->d_vk_mode, mode_data, 10ull*V*4u, hipMemcpyHostToDevice));
	s->vk_P = (uint32_t)P; s->vk_M = (uint32_t)mode_count; s->vk_stride = update_stride>1 ? update_stride : 1; s->vk_interp = stride_interpolation!=0;
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
		if(lead_alloc(s, (void**)&s->d_avg_u, 3ull*Np, 4u)!=hipSuccess||lead_alloc(s, (void**)&s->d_avg_rho, Np, 4u)!=hipSuccess||lead_alloc(s, (void**)&s->d_m2, 3ull*Np, 4u)!=hipSuccess)
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
		std::vector<float> tmp(3ull*N);
		if((e = copy_pitched(tmp.data(), s->d_avg_u, 4u, s, 3u, false, s->stream))) return e;
		HIP_TRY(hipStreamSynchronize(s->stream));
		for(uint64_t n=0ull; n<N; n++) { avg_u[3ull*n] = tmp[n]; avg_u[3ull*n+1ull] = tmp[N+n]; avg_u[3ull*n+2ull] = tmp[2ull*N+n]; }
	}
	if(avg_rho) if((e = copy_pitched(avg_rho, s->d_avg_rho, 4u, s, 1u, false, s->stream))) return e;
	float* m2h[3] = { m2_u, m2_v, m2_w };
	for(int c=0; c<3; c++) if(m2h[c]) {
		const size_t rows = (size_t)s->cfg.Ny*s->cfg.Nz;
		HIP_TRY(hipMemcpy2DAsync(m2h[c], (size_t)s->cfg.Nx*4u, s->d_m2+(size_t)c*s->kp.Np, (size_t)s->kp.Px*4u, (size_t)s->cfg.Nx*4u, rows, hipMemcpyDeviceToHost, s->stream));
	}
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

int luw_set_f(luw_solver* s, float fx, float fy, float fz) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_set_f: null solver");
	s->cfg.fx = s->kp.fx = fx; s->cfg.fy = s->kp.fy = fy; s->cfg.fz = s->kp.fz = fz;
	return LUW_OK;
}
int luw_set_coriolis(luw_solver* s, float ox, float oy, float oz) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_set_coriolis: null solver");
	s->cfg.omega_x = s->kp.omx = ox; s->cfg.omega_y = s->kp.omy = oy; s->cfg.omega_z = s->kp.omz = oz;
	s->kp.coriolis = ox!=0.0f||oy!=0.0f||oz!=0.0f;
	return LUW_OK;
}

int luw_initialize(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_initialize: null solver");
	if(int e = luw_upload(s, LUW_MASK_RHO|LUW_MASK_U|LUW_MASK_FLAGS|LUW_MASK_F|LUW_MASK_T)) return e;
	const uint32_t bx = s->cfg.Nx>=256u ? 256u : ((s->cfg.Nx+63u)/64u)*64u;
	const dim3 grid((s->cfg.Nx+bx-1u)/bx, s->cfg.Ny, s->cfg.Nz), block(bx);
	if(s->ddf_bytes==2u) hipLaunchKernelGGL((k_initialize<uint16_t>), grid, block, 0, s->stream, s->kp, (uint16_t*)s->d_fi, s->d_rho, s->d_u, s->d_flags, (uint16_t*)s->d_gi, s->d_T);
	else hipLaunchKernelGGL((k_initialize<float>), grid, block, 0, s->stream, s->kp, (float*)s->d_fi, s->d_rho, s->d_u, s->d_flags, (float*)s->d_gi, s->d_T);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s->stream));
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
	s->fields_current = write_fields!=0; // callers cover the lattice with boxes of one step using the same flag
	return launch_stream_collide(s, b, write_fields);
}
int luw_set_kernel(luw_solver* s, uint32_t kernel) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_set_kernel: null solver");
	s->kernel = kernel;
	return LUW_OK;
}
int luw_increment_time_step(luw_solver* s, uint64_t steps) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_increment_time_step: null solver");
	s->t += steps;
	return LUW_OK;
}

int luw_reset_time_step(luw_solver* s) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_reset_time_step: null solver");
	s->t = 0ull;
	return LUW_OK;
}

// VonKarmanInletUpdater::update + compute_time_params_ (FX/setup.cpp:538-558,1118-1140): at most once per time step
static int vk_apply(luw_solver* s) {
	if(!s->vk_active||s->vk_last_t==s->t) return LUW_OK;
	s->vk_last_t = s->t;
	const uint64_t t = s->t, stride = s->vk_stride>1 ? (uint64_t)s->vk_stride : 1ull;
	uint32_t use_interp = 0u; float t0 = (float)t, t1 = (float)t, alpha = 0.0f;
	if(stride>1ull) {
		const uint64_t anchor = (t/stride)*stride;
		if(s->vk_interp) { use_interp = 1u; t0 = (float)anchor; t1 = (float)(anchor+stride); alpha = (float)(t-anchor)/(float)stride; }
		else { t0 = (float)anchor; t1 = t0; }
	}
	hipLaunchKernelGGL(k_vk_inlet_apply, dim3((s->vk_P+255u)/256u), dim3(256), 0, s->stream, use_interp, t0, t1, alpha, s->vk_P, s->vk_M, 5u*s->vk_M, s->d_vk_cell, s->d_vk_face, s->d_vk_point, s->d_vk_mode, s->d_u, (size_t)s->kp.Np);
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}

static int run_steps(luw_solver* s, uint64_t steps, double* mean_kernel_ms) {
	if(!s) return fail(LUW_ERR_INVALID, "luw_run: null solver");
	if(int e = set_device(s)) return e;
	if(!s->initialized) { if(int e = luw_initialize(s)) return e; } // LBM::run initialises on first use, FX/lbm.cpp:1294-1296
	const Box whole = { 0u, s->cfg.Nx, 0u, s->cfg.Ny, 0u, s->cfg.Nz };
	const bool every = (s->cfg.options&LUW_OPT_UPDATE_FIELDS_EVERY_STEP)!=0u;
	std::vector<hipEvent_t> ev;
	if(mean_kernel_ms) {
		ev.resize(2u*steps);
		for(auto& e : ev) HIP_TRY(hipEventCreate(&e));
	}
	for(uint64_t i=0ull; i<steps; i++) {
		const int wf = (every||i+1ull==steps) ? 1 : 0;
		if(int e = vk_apply(s)) return e; // pre_step_update of the reference's run loop, FX/setup.cpp:4872
		if(mean_kernel_ms) HIP_TRY(hipEventRecord(ev[2u*i], s->stream));
		if(int e = launch_stream_collide(s, whole, wf)) return e;
		if(mean_kernel_ms) HIP_TRY(hipEventRecord(ev[2u*i+1u], s->stream));
		s->t++;
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	if(steps>0ull) s->fields_current = true;
	if(mean_kernel_ms) {
		double sum = 0.0;
		for(uint64_t i=0ull; i<steps; i++) { float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, ev[2u*i], ev[2u*i+1u])); sum += (double)ms; }
		for(auto& e : ev) (void)hipEventDestroy(e);
		*mean_kernel_ms = steps ? sum/(double)steps : 0.0;
	}
	return LUW_OK;
}
int luw_run(luw_solver* s, uint64_t steps) { return run_steps(s, steps, nullptr); }
int luw_run_timed(luw_solver* s, uint64_t steps, double* mean_kernel_ms) {
	if(!mean_kernel_ms) return fail(LUW_ERR_INVALID, "luw_run_timed: null output");
	return run_steps(s, steps, mean_kernel_ms);
}

int luw_enqueue_extract_fi(luw_solver* s, uint32_t direction, void* buf_p, void* buf_m) {
	if(!s||direction>2u||!buf_p||!buf_m) return fail(LUW_ERR_INVALID, "luw_enqueue_extract_fi: bad argument");
	if(int e = set_device(s)) return e;
	launch_transfer<false, false>(s, direction, buf_p, buf_m);
	HIP_TRY(hipGetLastError());
	return LUW_OK;
}
int luw_enqueue_insert_fi(luw_solver* s, uint32_t direction, const void* buf_p, const void* buf_m) {
	if(!s||direction>2u||!buf_p||!buf_m) return fail(LUW_ERR_INVALID, "luw_enqueue_insert_fi: bad argument");
	if(int e = set_device(s)) return e;
	launch_transfer<false, true>(s, direction, const_cast<void*>(buf_p), const_cast<void*>(buf_m));
	HIP_TRY(hipGetLastError());
	return LUW_OK;
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

} // extern "C"
