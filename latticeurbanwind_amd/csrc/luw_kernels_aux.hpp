// luw_kernels_aux.hpp -- halo pack/unpack, mesh voxeliser, probe gather, von-Karman inlet, on-device statistics, FP16C codec self-check
// Device code of libluw_core.so; included by luw_core.hip only (after luw_device.hpp, inside `using namespace luw`).
#pragma once

// ---------------------------------------------------------------- halo pack / unpack, FX/kernel.cpp:2188-2270
// Face cell of thread t and its index a in the transfer buffers: x runs fastest wherever x lies in the face (direction 0: a = y + z Ny; 1: a = x + z Nx;
// 2: a = x + y Nx), so that threads, lattice rows AND buffer are walked together.  The reference orders its y faces a = z + x Nz (FX/kernel.cpp:2188-2221);
// the order inside a buffer is private to the two kernels that fill and drain it -- the host only moves the bytes (FX/lbm.cpp:1908-1934) -- and with the
// reference's order every element of a y face costs its own 128-byte line on the buffer side (pack 0.046 -> 0.017 ms per step on a 514x514x512 FP32 rank).
template<int DIR> __device__ __forceinline__ void face_cell(const KParams& p, const uint32_t t, const uint32_t fixed, uint32_t& x, uint32_t& y, uint32_t& z,
	uint32_t& a) {
	if constexpr(DIR==0) { x = fixed; y = t%p.Ny; z = t/p.Ny; a = t; }
	else if constexpr(DIR==1) { x = t%p.Nx; y = fixed; z = t/p.Nx; a = t; }
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
template<typename T, bool G, int DIR, int PM, int BB> __device__ __forceinline__ void extract_one(const KParams& p, const uint32_t x, const uint32_t y,
	const uint32_t z, const uint32_t a, const uint32_t A, const uint32_t t_odd, T* __restrict__ buf, const T* __restrict__ fi) {
	constexpr int i = G ? 2*DIR+PM+1 : TransferIndex<DIR, PM, BB>::value;
	const uint32_t plane = t_odd ? ((i&1) ? i+1 : i-1) : i;
	const uint32_t n = (i&1) ? neighbor_index<i>(p, x, y, z) : x+(y+z*p.Ny)*p.Px;
	buf[(size_t)BB*A+a] = fi[(size_t)plane*p.Np+n];
}
template<typename T, bool G, int DIR, int PM, int BB> __device__ __forceinline__ void insert_one(const KParams& p, const uint32_t x, const uint32_t y,
	const uint32_t z, const uint32_t a, const uint32_t A, const uint32_t t_odd, const T* __restrict__ buf, T* __restrict__ fi) {
	constexpr int i = G ? 2*DIR+PM+1 : TransferIndex<DIR, PM, BB>::value;
	const uint32_t plane = t_odd ? i : ((i&1) ? i+1 : i-1);
	const uint32_t n = (i&1) ? x+(y+z*p.Ny)*p.Px : neighbor_index<i-1>(p, x, y, z);
	fi[(size_t)plane*p.Np+n] = buf[(size_t)BB*A+a];
}
template<typename T, bool G, int DIR> __global__ __launch_bounds__(256) void k_extract_fi(const KParams p, const uint32_t A, const uint32_t t_odd,
	T* __restrict__ buf_p, T* __restrict__ buf_m, const T* __restrict__ fi) {
	const uint32_t t = blockIdx.x*blockDim.x+threadIdx.x;
	if(t>=A) return;
	const uint32_t Nd = DIR==0 ? p.Nx : DIR==1 ? p.Ny : p.Nz;
	uint32_t x, y, z, a;
	face_cell<DIR>(p, t, Nd-2u, x, y, z, a);
	extract_one<T, G, DIR, 0, 0>(p, x, y, z, a, A, t_odd, buf_p, fi);
	if constexpr(!G) {
		extract_one<T, G, DIR, 0, 1>(p, x, y, z, a, A, t_odd, buf_p, fi);
		extract_one<T, G, DIR, 0, 2>(p, x, y, z, a, A, t_odd, buf_p, fi);
		extract_one<T, G, DIR, 0, 3>(p, x, y, z, a, A, t_odd, buf_p, fi);
		extract_one<T, G, DIR, 0, 4>(p, x, y, z, a, A, t_odd, buf_p, fi);
	}
	face_cell<DIR>(p, t, 1u, x, y, z, a);
	extract_one<T, G, DIR, 1, 0>(p, x, y, z, a, A, t_odd, buf_m, fi);
	if constexpr(!G) {
		extract_one<T, G, DIR, 1, 1>(p, x, y, z, a, A, t_odd, buf_m, fi);
		extract_one<T, G, DIR, 1, 2>(p, x, y, z, a, A, t_odd, buf_m, fi);
		extract_one<T, G, DIR, 1, 3>(p, x, y, z, a, A, t_odd, buf_m, fi);
		extract_one<T, G, DIR, 1, 4>(p, x, y, z, a, A, t_odd, buf_m, fi);
	}
}
template<typename T, bool G, int DIR> __global__ __launch_bounds__(256) void k_insert_fi(const KParams p, const uint32_t A, const uint32_t t_odd,
	const T* __restrict__ buf_p, const T* __restrict__ buf_m, T* __restrict__ fi) {
	const uint32_t t = blockIdx.x*blockDim.x+threadIdx.x;
	if(t>=A) return;
	const uint32_t Nd = DIR==0 ? p.Nx : DIR==1 ? p.Ny : p.Nz;
	uint32_t x, y, z, a;
	if(buf_p) { // (a null buffer: that side is left alone -- xin_settle, luw_launch.hpp)
		face_cell<DIR>(p, t, Nd-1u, x, y, z, a);
		insert_one<T, G, DIR, 0, 0>(p, x, y, z, a, A, t_odd, buf_p, fi);
		if constexpr(!G) {
			insert_one<T, G, DIR, 0, 1>(p, x, y, z, a, A, t_odd, buf_p, fi);
			insert_one<T, G, DIR, 0, 2>(p, x, y, z, a, A, t_odd, buf_p, fi);
			insert_one<T, G, DIR, 0, 3>(p, x, y, z, a, A, t_odd, buf_p, fi);
			insert_one<T, G, DIR, 0, 4>(p, x, y, z, a, A, t_odd, buf_p, fi);
		}
	}
	if(buf_m) {
		face_cell<DIR>(p, t, 0u, x, y, z, a);
		insert_one<T, G, DIR, 1, 0>(p, x, y, z, a, A, t_odd, buf_m, fi);
		if constexpr(!G) {
			insert_one<T, G, DIR, 1, 1>(p, x, y, z, a, A, t_odd, buf_m, fi);
			insert_one<T, G, DIR, 1, 2>(p, x, y, z, a, A, t_odd, buf_m, fi);
			insert_one<T, G, DIR, 1, 3>(p, x, y, z, a, A, t_odd, buf_m, fi);
			insert_one<T, G, DIR, 1, 4>(p, x, y, z, a, A, t_odd, buf_m, fi);
		}
	}
}
// ---------------------------------------------------------------- edge populations: the halo exchange in ONE phase
// The reference swaps faces axis by axis, x then y then z, each face including the halo rims of the axes before it, so that a population that crosses two
// cuts at once reaches the diagonal neighbour in two hops (FX/lbm.cpp:1908-1934).  In D3Q19 exactly ONE population crosses a given pair of cuts in a given
// diagonal direction (7, 8, 13, 14 for x-y; 9, 10, 15, 16 for x-z; 11, 12, 17, 18 for y-z; none crosses three), along the line of cells where the two faces
// meet.  Sent straight to the diagonal neighbour -- edge e = i - 7 carries population i to the domain in direction c_i -- the faces of all axes can travel in
// the same batch: what the rims of the faces carried (the first hop of exactly these populations; everything else in a rim is never read) is overwritten by the
// edge insert, which therefore comes last.  Slots (Esoteric-Pull, FX/kernel.cpp:1338-1351; io = the odd partner of i):
//   odd i   leaves in slot B(i) of the halo-halo line beyond the sender's corner cells, arrives in slot B(i) of the receiver's OWNED corner line opposite;
//   even i  leaves in slot A(io) of the sender's owned corner line (where the cell stored it), arrives in slot A(io) of the receiver's halo-halo line opposite.
// The same cells and values the two-hop route delivers (tests/test_gpu_halo.py, tests/test_distributed_gloo.py hold both routes to the oracle).
struct EdgeBufs { void* p[12]; };
// coordinate along a cut axis: SENDER side / RECEIVER side of population i (sign s = c_i on that axis)
__device__ __forceinline__ uint32_t edge_coord(const bool odd_pop, const bool sender, const int s, const uint32_t N) {
	if(odd_pop) return sender ? (s>0 ? N-1u : 0u) : (s>0 ? 1u : N-2u);        // halo-halo line beyond the corner -> owned corner line opposite
	return sender ? (s>0 ? N-2u : 1u) : (s>0 ? 0u : N-1u);                    // owned corner line -> halo-halo line opposite
}
// INSERT with xin_p / xin_m set (the x faces wait in their receive buffers, luw_set_x_face_inputs): an edge across the x cut lands where the readers of those
// buffers look for it -- the rim element of the halo cell that owns the slot (an odd population: the cell one step back along c from the owned corner cell, in
// the buffer that came from -x; an even one: the halo-halo cell itself, in the buffer that came from +x) -- instead of in the lattice.
template<typename T, bool INSERT> __global__ __launch_bounds__(256) void k_edges(const KParams p, const uint32_t t_odd, const EdgeBufs bufs,
	T* __restrict__ fi, T* __restrict__ xin_p = nullptr, T* __restrict__ xin_m = nullptr) {
	const uint32_t e = blockIdx.y, l = blockIdx.x*blockDim.x+threadIdx.x;
	T* const buf = (T*)bufs.p[e];
	if(!buf) return;
	const int i = 7+(int)e, io = (i&1) ? i : i-1;
	// c_i (FX/kernel.cpp:890-893): 7..12 = (+,+) and (-,-) for the axis pairs (x, y), (x, z), (y, z); 13..18 = (+,-) and (-,+) for the same pairs
	const int pair = ((int)e%6)/2, sa = (i&1) ? 1 : -1, sb = e<6u ? sa : -sa;
	const int ax_a = pair==2 ? 1 : 0, ax_b = pair==0 ? 1 : 2, ax_c = 3-ax_a-ax_b;
	const uint32_t N[3] = { p.Nx, p.Ny, p.Nz };
	if(l>=N[ax_c]) return;
	uint32_t c[3];
	c[ax_a] = edge_coord((i&1)!=0, !INSERT, sa, N[ax_a]); c[ax_b] = edge_coord((i&1)!=0, !INSERT, sb, N[ax_b]); c[ax_c] = l;
	// slot A(io, t) = t odd ? io : io + 1 holds population io + 1 at the cell; slot B(io, t) = t odd ? io + 1 : io holds population io at the +c_io neighbour
	const uint32_t plane = (i&1) ? (t_odd ? (uint32_t)io+1u : (uint32_t)io) : (t_odd ? (uint32_t)io : (uint32_t)io+1u);
	const size_t n = (size_t)plane*p.Np+c[0]+((size_t)c[1]+(size_t)c[2]*p.Ny)*p.Px;
	if constexpr(INSERT) {
		T* const face = pair==2 ? nullptr : (i&1) ? xin_m : xin_p;
		if(face) {
			const int k = io==7 ? 1 : io==13 ? 2 : io==9 ? 3 : 4;                                  // place of the pair in the face buffers (1, 7, 13, 9, 15)
			if(i&1) c[ax_b] = (uint32_t)((int)c[ax_b]-sb);                                         // the halo cell that owns the slot
			face[(size_t)k*p.Ny*p.Nz+c[1]+(size_t)c[2]*p.Ny] = buf[l];
		} else fi[n] = buf[l];
	} else buf[l] = fi[n];
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
// lattice of the pass: a solver domain, or a bare global lattice (luw_voxelize_lattice)
struct VoxGrid { uint32_t Nx, Ny, Nz, Px; int Ox, Oy, Oz; uint64_t Np; };
// One block = one 16x16 tile of columns; it visits only the triangles binned to the tile (tile_start / tile_tri: CSR, triangle
// ids ascending, so hits are met in the reference's order and the 64-entry cut-off falls on the same hits).  The bins hold
// every triangle whose xy bounding box, grown by one cell, touches the tile -- the margin the reference itself uses when it
// hands a domain its triangle subset (FX/lbm.cpp:1455-1487).
constexpr uint32_t VOX_TILE = 16u;
__global__ __launch_bounds__(256) void k_voxelize_z(const VoxGrid p, uint8_t* __restrict__ flags, const float* __restrict__ u, const uint8_t flag,
	const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_tri,
		const float* __restrict__ p0, const float* __restrict__ p1, const float* __restrict__ p2, const float x0, const float y0, const float z0,
			const float x1, const float y1, const float z1) {
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
		// was solid with the mesh's velocity (static: 0), FX/kernel.cpp:2451-2462
		else if((fl&TYPE_BO)==TYPE_S&&(!u||(u[n]==0.0f&&u[p.Np+n]==0.0f&&u[2ull*p.Np+n]==0.0f))) fl = (uint8_t)(fl&~flag);
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
// point_cell = nullptr: the values go to a PACKED buffer (u[c*Np + i], Np = P) instead of the lattice -- the evaluation of the NEXT
// step's inlet values then runs beside the current step on another stream (VALU-bound here, HBM-bound there) and k_vk_scatter puts
// them into place in a few microseconds.
__global__ __launch_bounds__(256) void k_vk_inlet_apply(const uint32_t use_interp, const float t0, const float t1, const float alpha, const uint32_t P,
	const uint32_t M, const uint32_t V,
		const uint32_t* __restrict__ point_cell, const uint8_t* __restrict__ point_face, const float* __restrict__ point_data,
			const float* __restrict__ mode_data, float* __restrict__ u, const size_t Np) {
	const uint32_t i = blockIdx.x*blockDim.x+threadIdx.x;
	if(i>=P) return;
	const uint32_t n = point_cell ? point_cell[i] : i;
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

__global__ __launch_bounds__(256) void k_vk_scatter(const uint32_t P, const uint32_t* __restrict__ point_cell, const float* __restrict__ val,
	float* __restrict__ u, const size_t Np) {
	const uint32_t i = blockIdx.x*blockDim.x+threadIdx.x;
	if(i>=P) return;
	const uint32_t n = point_cell[i];
	u[n] = val[i]; u[Np+n] = val[(size_t)P+i]; u[2ull*Np+n] = val[2ull*P+i];
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
	for(unsigned long long v=tid; v<(1ull<<32); v+=nth) {
		if(((uint32_t)v&0x7F800000u)>=(230u<<23)) continue;
		const float x = __uint_as_float((uint32_t)v);
		bad += float_to_half_custom(x)!=float_to_half_custom_ref(x);
	}
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

// the plain-range division and square root of luw_device.hpp (recip_prepare / div_by / sqrt_in_range) against the library forms they replace:
// mismatches[0]: square roots -- every float x = 0 or 2^-96 <= x < Inf; mismatches[1]: quotients n/d -- every float d of [1/4, 4] (2^25 + 1
// of them) and 2^26 hashed d of the whole ordinary range [2^-60, 2^60), 64 numerators each: 0, 0.5, 2, and hashed floats of either sign from
// [2^-25, 64) (bit patterns compared; a zero quotient's sign is not, see recip_prepare); mismatches[2]: the same denominators with numerators
// on the momentum grid of the FP16C kernels, n = k 2^-25, |k| < 2^24
__device__ __forceinline__ void arith_check_denominator(const uint32_t db, unsigned long long& bad_div, unsigned long long& bad_grid) {
	const float d = __uint_as_float(db);
	const Recip R = recip_prepare(d);
	uint32_t h = db*2654435761u+12345u;
	for(int k=0; k<64; k++) {
		h = h*1664525u+1013904223u;
		uint32_t nb = (h&0x807FFFFFu)|((102u+(h>>23)%31u)<<23);   // exponent field 102 .. 132: 2^-25 <= |n| < 64
		if(k==0) nb = 0u; else if(k==1) nb = 0x3F000000u; else if(k==2) nb = 0x40000000u;
		const float n = __uint_as_float(nb);
		const uint32_t want = __float_as_uint(n/d), got = __float_as_uint(div_by(n, R));
		bad_div += (want!=got)&&(((want|got)&0x7FFFFFFFu)!=0u);
		// a multiple of 2^-25 below 1/2 in magnitude: what the moment sums of FP16C populations are
		const float ng = (float)((int32_t)(h>>7)-(1<<24))*0x1p-25f;
		const uint32_t wg = __float_as_uint(ng/d), gg = __float_as_uint(div_by(ng, R));
		bad_grid += (wg!=gg)&&(((wg|gg)&0x7FFFFFFFu)!=0u);
	}
}
__global__ __launch_bounds__(256) void k_arith_check(unsigned long long* __restrict__ mismatches) {
	const uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x, nth = gridDim.x*blockDim.x;
	unsigned long long bad_sqrt = 0ull, bad_div = 0ull, bad_grid = 0ull;
	for(unsigned long long v=tid; v<0x7F800000ull; v+=nth) {
		const uint32_t b = (uint32_t)v;
		if(b!=0u&&b<(31u<<23)) continue;                           // below 2^-96: the library pre-scales, the callers do not care (smagorinsky_rate_plain)
		const float x = __uint_as_float(b);
		bad_sqrt += __float_as_uint(sqrt_in_range(x))!=__float_as_uint(sqrtf(x));
	}
	for(uint32_t db=0x3E800000u+tid; db<=0x40800000u; db+=nth) arith_check_denominator(db, bad_div, bad_grid);
	for(uint32_t i=tid; i<(1u<<26); i+=nth) {                      // the whole ordinary range: bit patterns 0x21800000 .. 0x5D7FFFFF
		const uint32_t h = i*2246822519u+374761393u;
		const uint32_t db = 0x21800000u+(uint32_t)(((unsigned long long)h*0x3C000000ull)>>32);
		if(density_is_ordinary(__uint_as_float(db))) arith_check_denominator(db, bad_div, bad_grid); else bad_div++;
	}
	if(bad_sqrt) atomicAdd(mismatches, bad_sqrt);
	if(bad_div) atomicAdd(mismatches+1, bad_div);
	if(bad_grid) atomicAdd(mismatches+2, bad_grid);
}

// ---- test hook (luw_dev_schedule_jitter): one lane that keeps its stream busy for about `us` microseconds -- whatever is enqueued behind it runs late.
// Bounded twice: by the 100 MHz wall clock and by the number of sleeps, so that it ends whatever the clock does.
__global__ void k_delay(const uint32_t us) {
	const uint64_t t0 = wall_clock64();
	for(uint32_t i=0u; i<4u*us+4u && wall_clock64()-t0<100ull*us; i++) __builtin_amdgcn_s_sleep(32);
}
