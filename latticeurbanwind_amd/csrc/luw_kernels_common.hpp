// luw_kernels_common.hpp -- Esoteric-Pull slot algebra, boxes, cache-policy load/store helpers, k_initialize
// Device code of libluw_core.so; included by luw_core.hip only (after luw_device.hpp, inside `using namespace luw`).
#pragma once

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

template<typename T> __global__ __launch_bounds__(256) void k_initialize(const KParams p, T* __restrict__ fi, const float* __restrict__ rho,
	float* __restrict__ u, const uint8_t* __restrict__ flags, T* __restrict__ gi, const float* __restrict__ Tf) {
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

// ---------------------------------------------------------------- statistics fused into the step ("fused epilogue")
// A sampled step updates Welford's mean / M2 (accumulate_from_buffers, FX/setup.cpp:4441-4488) with the rho,u this step shows,
// straight from the registers of the cell update: the 16 B/cell round trip through rho,u (written by the step, read back by
// k_stats_accumulate) disappears.  Same operations in the same order as k_stats_accumulate, hence the same bits.  The seven
// new values pass through an empty volatile asm before they are stored: volatile asms keep their order, so the arithmetic
// cannot sink behind the FP16C kernels' switch to round-toward-zero at the tail.
struct StatsArgs { float* avg_u; float* avg_rho; float* m2; float inv_n; };
#ifndef LUW_STATS_NT
#define LUW_STATS_NT true   /* cache policy of the scalar kernels' statistics accesses (A/B: -DLUW_STATS_NT=false) */
#endif
__device__ __forceinline__ void stats_welford(const size_t Np, const StatsArgs& S, const uint32_t n, const float r, const float ux, const float uy,
	const float uz) {
	const float v[3] = { ux, uy, uz };
	float mean[3], m2n[3];
	#pragma unroll
	for(int c=0; c<3; c++) {
		mean[c] = ldg<LUW_STATS_NT>(S.avg_u+c*Np+n);
		const float delta = v[c]-mean[c];
		mean[c] += delta*S.inv_n;
		const float delta2 = v[c]-mean[c];
		m2n[c] = ldg<LUW_STATS_NT>(S.m2+c*Np+n)+delta*delta2;
	}
	const float ra = ldg<LUW_STATS_NT>(S.avg_rho+n);
	float rn = ra+(r-ra)*S.inv_n;
	asm volatile("" : "+v"(mean[0]), "+v"(mean[1]), "+v"(mean[2]), "+v"(m2n[0]), "+v"(m2n[1]), "+v"(m2n[2]), "+v"(rn));
	#pragma unroll
	for(int c=0; c<3; c++) { stg<LUW_STATS_NT>(S.m2+c*Np+n, m2n[c]); stg<LUW_STATS_NT>(S.avg_u+c*Np+n, mean[c]); }
	stg<LUW_STATS_NT>(S.avg_rho+n, rn);
}
// pair kernel: the cells (n, n+1) of one lane together, 8-byte accesses (n is even there).  A cell without a sample (halo, row
// padding) gets its old values written back.  Separate 4-byte passes per cell would fetch every line twice (measured: slower
// than the separate statistics kernel).
struct PairSample { float r[2], ux[2], uy[2], uz[2]; bool has[2]; };
__device__ __forceinline__ void stats_welford_pair(const size_t Np, const StatsArgs& S, const uint32_t n, const PairSample& q) {
	f32x2 mean[3], m2n[3];
	#pragma unroll
	for(int c=0; c<3; c++) {
		const float* v = c==0 ? q.ux : c==1 ? q.uy : q.uz;
		const f32x2 m0 = ldg<true>(reinterpret_cast<const f32x2*>(S.avg_u+c*Np+n)), s0 = ldg<true>(reinterpret_cast<const f32x2*>(S.m2+c*Np+n));
		const float mo[2] = { m0.x, m0.y }, so[2] = { s0.x, s0.y };
		float mn[2], sn[2];
		#pragma unroll
		for(int h=0; h<2; h++) {
			const float delta = v[h]-mo[h];
			const float mm = mo[h]+delta*S.inv_n;
			const float delta2 = v[h]-mm;
			mn[h] = q.has[h] ? mm : mo[h];
			sn[h] = q.has[h] ? so[h]+delta*delta2 : so[h];
		}
		mean[c] = f32x2{ mn[0], mn[1] }; m2n[c] = f32x2{ sn[0], sn[1] };
	}
	const f32x2 ra = ldg<true>(reinterpret_cast<const f32x2*>(S.avg_rho+n));
	f32x2 rn = { q.has[0] ? ra.x+(q.r[0]-ra.x)*S.inv_n : ra.x, q.has[1] ? ra.y+(q.r[1]-ra.y)*S.inv_n : ra.y };
	asm volatile("" : "+v"(mean[0]), "+v"(mean[1]), "+v"(mean[2]), "+v"(rn), "+v"(m2n[0]), "+v"(m2n[1]), "+v"(m2n[2]));
	#pragma unroll
	for(int c=0; c<3; c++) { stg<true>(reinterpret_cast<f32x2*>(S.m2+c*Np+n), m2n[c]); stg<true>(reinterpret_cast<f32x2*>(S.avg_u+c*Np+n), mean[c]); }
	stg<true>(reinterpret_cast<f32x2*>(S.avg_rho+n), rn);
}
// A cell the step never updates (solid): its rho,u are constants, and Welford's update of a constant c from the reset state is
// exactly mean = c, M2 = +0 after any number of samples (0 + (c-0)*1 = c; then c + (c-c)*inv_n = c; M2 = 0 + c*(c-c) = +0).
// Stored as such, without arithmetic: a wave runs this branch AFTER its fluid lanes have gone through the kernel's tail, i.e.
// with the FP16C kernels' round-toward-zero mode already set (the mode is per wave, not per lane) -- no floating-point
// instruction may sit here (tests/test_isa_contract.py follows the control flow behind the switch).
__device__ __forceinline__ void stats_hold_constant_cell(const size_t Np, const StatsArgs& S, const uint32_t n, const float* __restrict__ rho,
	const float* __restrict__ u) {
	#pragma unroll
	for(int c=0; c<3; c++) { stg<true>(S.avg_u+c*Np+n, u[c*Np+n]); stg<true>(S.m2+c*Np+n, 0.0f); }
	stg<true>(S.avg_rho+n, rho[n]);
}
// a cell whose fields are inputs (TYPE_E): the sample is what rho,u hold
__device__ __forceinline__ void stats_welford_from_fields(const size_t Np, const StatsArgs& S, const uint32_t n, const float* __restrict__ rho,
	const float* __restrict__ u) {
	stats_welford(Np, S, n, rho[n], u[n], u[Np+n], u[2ull*Np+n]);
}

