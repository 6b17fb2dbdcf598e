// luw_kernels_vec.hpp -- A/B only: k_stream_collide_v, V cells per lane with aligned accesses and wave64 lane shifts (parity-tested, slower than the product
// kernels)
// Device code of libluw_core.so; included by luw_core.hip only (after luw_device.hpp, inside `using namespace luw`).
#pragma once

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
template<typename T, int V, int PARITY> __global__ __launch_bounds__(256) void k_stream_collide_v(const KParams p, const Box b, const uint32_t nchunk,
	T* __restrict__ fi, float* __restrict__ rho, float* __restrict__ u,
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

