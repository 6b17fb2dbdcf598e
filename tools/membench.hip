// membench.hip -- HBM microbenchmarks that bound what the collide-stream access pattern can reach on MI355X.
// Not part of the product; run through gpurun:  hipcc --offload-arch=gfx950 -O3 -o membench tools/membench.hip && ./membench
// Every test moves the bytes of one D3Q19 step on N = 512^3 cells (19 x 4 B read + 19 x 4 B written per cell) and
// reports GB/s = 152 B x N / time.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e!=hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

// S planes, W floats per lane (1,2,4); LAYOUT 0: plane-major out[s*P + n]; LAYOUT 1: row-interleaved out[(row*S + s)*NX + x]
template<int S, int W, int LAYOUT, bool INPLACE> __global__ __launch_bounds__(256) void k_update(const float* __restrict__ in, float* __restrict__ out, const size_t P, const unsigned NX) {
	const size_t e = ((size_t)blockIdx.x*blockDim.x+threadIdx.x)*W; // element index within a plane
	if(e>=P) return;
	float v[S][W];
	#pragma unroll
	for(int s=0; s<S; s++) {
		size_t idx;
		if(LAYOUT==0) idx = (size_t)s*P+e; else { const size_t row = e/NX, x = e%NX; idx = (row*S+s)*NX+x; }
		if constexpr(W==4) { const float4 t = *reinterpret_cast<const float4*>(in+idx); v[s][0]=t.x; v[s][1]=t.y; v[s][2]=t.z; v[s][3]=t.w; }
		else if constexpr(W==2) { const float2 t = *reinterpret_cast<const float2*>(in+idx); v[s][0]=t.x; v[s][1]=t.y; }
		else v[s][0] = in[idx];
	}
	float* o = INPLACE ? const_cast<float*>(in) : out;
	#pragma unroll
	for(int s=0; s<S; s++) {
		size_t idx;
		if(LAYOUT==0) idx = (size_t)s*P+e; else { const size_t row = e/NX, x = e%NX; idx = (row*S+s)*NX+x; }
		if constexpr(W==4) { *reinterpret_cast<float4*>(o+idx) = make_float4(v[s][0]+1.0f, v[s][1]+1.0f, v[s][2]+1.0f, v[s][3]+1.0f); }
		else if constexpr(W==2) { *reinterpret_cast<float2*>(o+idx) = make_float2(v[s][0]+1.0f, v[s][1]+1.0f); }
		else o[idx] = v[s][0]+1.0f;
	}
}
// 19-plane in-place dword/float4 update with (a) occupancy limited through a dynamic LDS request and (b) optional
// non-temporal loads / stores
template<int S, int W, bool NT> __global__ __launch_bounds__(256) void k_update_occ(float* __restrict__ io, const size_t P) {
	extern __shared__ float lds[];
	const size_t e = ((size_t)blockIdx.x*blockDim.x+threadIdx.x)*W;
	if(e>=P) return;
	float v[S][W];
	#pragma unroll
	for(int s=0; s<S; s++) {
		const float* p = io+(size_t)s*P+e;
		#pragma unroll
		for(int c=0; c<W; c++) v[s][c] = NT ? __builtin_nontemporal_load(p+c) : p[c];
	}
	if(v[0][0]==-123.0f) lds[threadIdx.x] = v[0][0];
	#pragma unroll
	for(int s=0; s<S; s++) {
		float* p = io+(size_t)s*P+e;
		#pragma unroll
		for(int c=0; c<W; c++) { if(NT) __builtin_nontemporal_store(v[s][c]+1.0f, p+c); else p[c] = v[s][c]+1.0f; }
	}
}
// read-only / write-only of S planes (dword)
template<int S, int W> __global__ __launch_bounds__(256) void k_read(const float* __restrict__ in, float* __restrict__ sink, const size_t P) {
	const size_t e = ((size_t)blockIdx.x*blockDim.x+threadIdx.x)*W;
	if(e>=P) return;
	float acc = 0.0f;
	#pragma unroll
	for(int s=0; s<S; s++) {
		if constexpr(W==4) { const float4 t = *reinterpret_cast<const float4*>(in+(size_t)s*P+e); acc += t.x+t.y+t.z+t.w; }
		else acc += in[(size_t)s*P+e];
	}
	if(acc==123.456f) sink[0] = acc;
}
template<int S, int W> __global__ __launch_bounds__(256) void k_write(float* __restrict__ out, const size_t P) {
	const size_t e = ((size_t)blockIdx.x*blockDim.x+threadIdx.x)*W;
	if(e>=P) return;
	#pragma unroll
	for(int s=0; s<S; s++) {
		if constexpr(W==4) *reinterpret_cast<float4*>(out+(size_t)s*P+e) = make_float4(1.0f, 2.0f, 3.0f, (float)s);
		else out[(size_t)s*P+e] = (float)s;
	}
}

// FP16C-shaped traffic: 19 planes of 2-byte elements updated in place, W elements per lane (W = 1: ushort accesses, 128 B per
// wave instruction; W = 2: one dword per lane, 256 B), 5 of the 19 planes accessed one element further along the row like the
// x+1 populations (for W = 2 that is a dword on a 2-byte boundary), occupancy limited through the LDS request
template<int W, bool NT> __global__ __launch_bounds__(256) void k_update_half(uint16_t* __restrict__ io, const size_t P) {
	extern __shared__ float lds[];
	const size_t e = ((size_t)blockIdx.x*blockDim.x+threadIdx.x)*W;
	if(e+W+1>P) return;
	uint32_t v[19];
	#pragma unroll
	for(int s=0; s<19; s++) {
		const bool shifted = s==2||s==8||s==10||s==14||s==16;
		const uint16_t* p = io+(size_t)s*P+e+(shifted ? 1 : 0);
		if constexpr(W==1) v[s] = (NT&&!shifted) ? __builtin_nontemporal_load(p) : *p;
		else { if(NT&&!shifted) v[s] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p)); else __builtin_memcpy(&v[s], p, 4); }
	}
	if(v[0]==0xDEADBEEFu) lds[threadIdx.x] = 1.0f;
	#pragma unroll
	for(int s=0; s<19; s++) {
		const bool shifted = s==2||s==8||s==10||s==14||s==16;
		uint16_t* p = io+(size_t)s*P+e+(shifted ? 1 : 0);
		const uint32_t r = v[s]+0x00010001u;
		if constexpr(W==1) { if(NT&&!shifted) __builtin_nontemporal_store((uint16_t)r, p); else *p = (uint16_t)r; }
		else { if(NT&&!shifted) __builtin_nontemporal_store(r, reinterpret_cast<uint32_t*>(p)); else __builtin_memcpy(p, &r, 4); }
	}
}
// layouts for the 19 planes under the product access pattern (dword per lane, non-temporal, occupancy through LDS):
// LAYOUT 0 plane-major [s][row][x]; 1 row-interleaved [row][s][x]; 2 slab-interleaved [z][s][y][x] (NY rows per slab)
template<int LAYOUT, bool NT> __global__ __launch_bounds__(256) void k_update_layout(float* __restrict__ io, const size_t P, const unsigned NX, const unsigned NY) {
	extern __shared__ float lds[];
	const size_t e = (size_t)blockIdx.x*blockDim.x+threadIdx.x;
	if(e>=P) return;
	const size_t row = e/NX, x = e%NX;
	float v[19];
	#pragma unroll
	for(int s=0; s<19; s++) {
		size_t idx;
		if(LAYOUT==0) idx = (size_t)s*P+e; else if(LAYOUT==1) idx = (row*19+s)*NX+x; else { const size_t z = row/NY, y = row%NY; idx = ((z*19+s)*NY+y)*NX+x; }
		v[s] = NT ? __builtin_nontemporal_load(io+idx) : io[idx];
	}
	if(v[0]==-123.0f) lds[threadIdx.x] = v[0];
	#pragma unroll
	for(int s=0; s<19; s++) {
		size_t idx;
		if(LAYOUT==0) idx = (size_t)s*P+e; else if(LAYOUT==1) idx = (row*19+s)*NX+x; else { const size_t z = row/NY, y = row%NY; idx = ((z*19+s)*NY+y)*NX+x; }
		if(NT) __builtin_nontemporal_store(v[s]+1.0f, io+idx); else io[idx] = v[s]+1.0f;
	}
}

template<typename F> static double time_ms(F launch, int reps=20) {
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	for(int i=0; i<3; i++) launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for(int i=0; i<reps; i++) launch();
	CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
	float ms; CHECK(hipEventElapsedTime(&ms, a, b));
	return ms/reps;
}

int main(int argc, char** argv) {
	const size_t N = 512ull*512ull*512ull; const unsigned NX = 512;
	float *A, *B; CHECK(hipMalloc(&A, 19*N*4)); CHECK(hipMalloc(&B, 19*N*4));
	CHECK(hipMemset(A, 0, 19*N*4)); CHECK(hipMemset(B, 0, 19*N*4));
	const double bytes = 152.0*N;
	auto rep = [&](const char* name, double ms, double b) { printf("%-58s %8.3f ms  %8.1f GB/s\n", name, ms, b/ms/1e6); fflush(stdout); };
	if(argc>1&&!strcmp(argv[1], "layout")) { // memory layouts of the 19 planes
		#define LAY(L, NTV, name) { CHECK(hipFuncSetAttribute((const void*)k_update_layout<L, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048)); \
			for(int blocks_per_cu : {8, 4, 3, 2}) { const size_t lds_bytes = blocks_per_cu>=8 ? 0 : (size_t)(160*1024/blocks_per_cu-2048); char nm[128]; snprintf(nm, sizeof(nm), "%s %d blocks/CU", name, blocks_per_cu); \
			rep(nm, time_ms([&]{ hipLaunchKernelGGL((k_update_layout<L, NTV>), dim3((unsigned)((N+255)/256)), dim3(256), lds_bytes, 0, A, N, NX, 512u); }), bytes); } }
		LAY(0, true, "plane-major      nt") LAY(1, true, "row-interleaved  nt") LAY(2, true, "slab-interleaved nt")
		LAY(0, false, "plane-major        ") LAY(1, false, "row-interleaved    ") LAY(2, false, "slab-interleaved   ")
		return 0;
	}
	if(argc>1&&!strcmp(argv[1], "half")) { // FP16C-shaped traffic only
		uint16_t* H = reinterpret_cast<uint16_t*>(A);
		CHECK(hipFuncSetAttribute((const void*)k_update_half<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		CHECK(hipFuncSetAttribute((const void*)k_update_half<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		CHECK(hipFuncSetAttribute((const void*)k_update_half<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		CHECK(hipFuncSetAttribute((const void*)k_update_half<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		for(int blocks_per_cu : {8, 6, 5, 4, 3, 2}) {
			const size_t lds_bytes = blocks_per_cu>=8 ? 0 : (size_t)(160*1024/blocks_per_cu-2048);
			char name[128];
			snprintf(name, sizeof(name), "half in-place 19 planes ushort/lane %d blocks/CU nt", blocks_per_cu);
			rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_half<1, true>), dim3((unsigned)((N+255)/256)), dim3(256), lds_bytes, 0, H, N); }), 76.0*N);
			snprintf(name, sizeof(name), "half in-place 19 planes dword/lane  %d blocks/CU nt", blocks_per_cu);
			rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_half<2, true>), dim3((unsigned)((N/2+255)/256)), dim3(256), lds_bytes, 0, H, N); }), 76.0*N);
			snprintf(name, sizeof(name), "half in-place 19 planes ushort/lane %d blocks/CU", blocks_per_cu);
			rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_half<1, false>), dim3((unsigned)((N+255)/256)), dim3(256), lds_bytes, 0, H, N); }), 76.0*N);
			snprintf(name, sizeof(name), "half in-place 19 planes dword/lane  %d blocks/CU", blocks_per_cu);
			rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_half<2, false>), dim3((unsigned)((N/2+255)/256)), dim3(256), lds_bytes, 0, H, N); }), 76.0*N);
		}
		return 0;
	}
	#define RUN(name, S, W, L, IP, total) { const size_t P = (total)/(S); rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update<S, W, L, IP>), dim3((unsigned)((P/W+255)/256)), dim3(256), 0, 0, A, B, P, NX); }), bytes); }
	RUN("copy A->B  1 plane-pair  float4", 1, 4, 0, false, 19*N)
	RUN("copy A->B  1 plane-pair  dword", 1, 1, 0, false, 19*N)
	RUN("in-place   1 plane       float4", 1, 4, 0, true, 19*N)
	RUN("in-place   1 plane       dword", 1, 1, 0, true, 19*N)
	RUN("copy A->B  19 planes     float4", 19, 4, 0, false, 19*N)
	RUN("copy A->B  19 planes     dword", 19, 1, 0, false, 19*N)
	RUN("in-place   19 planes     float4", 19, 4, 0, true, 19*N)
	RUN("in-place   19 planes     float2", 19, 2, 0, true, 19*N)
	RUN("in-place   19 planes     dword", 19, 1, 0, true, 19*N)
	RUN("in-place   19 row-interleaved planes float4", 19, 4, 1, true, 19*N)
	RUN("in-place   19 row-interleaved planes dword", 19, 1, 1, true, 19*N)
	RUN("in-place   4 planes      dword", 4, 1, 0, true, 19*N)
	RUN("in-place   8 planes      dword", 8, 1, 0, true, 19*N)
	for(int blocks_per_cu : {8, 4, 3, 2, 1}) {
		const size_t lds_bytes = blocks_per_cu>=8 ? 0 : (size_t)(160*1024/blocks_per_cu-2048);
		char name[128];
		const size_t P = N;
		CHECK(hipFuncSetAttribute((const void*)k_update_occ<19, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		CHECK(hipFuncSetAttribute((const void*)k_update_occ<19, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		CHECK(hipFuncSetAttribute((const void*)k_update_occ<19, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		CHECK(hipFuncSetAttribute((const void*)k_update_occ<19, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-2048));
		snprintf(name, sizeof(name), "in-place 19 planes dword   %d blocks/CU", blocks_per_cu);
		rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_occ<19, 1, false>), dim3((unsigned)((P+255)/256)), dim3(256), lds_bytes, 0, A, P); }), bytes);
		snprintf(name, sizeof(name), "in-place 19 planes dword   %d blocks/CU nontemporal", blocks_per_cu);
		rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_occ<19, 1, true>), dim3((unsigned)((P+255)/256)), dim3(256), lds_bytes, 0, A, P); }), bytes);
		snprintf(name, sizeof(name), "in-place 19 planes float4  %d blocks/CU", blocks_per_cu);
		rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_occ<19, 4, false>), dim3((unsigned)((P/4+255)/256)), dim3(256), lds_bytes, 0, A, P); }), bytes);
		snprintf(name, sizeof(name), "in-place 19 planes float4  %d blocks/CU nontemporal", blocks_per_cu);
		rep(name, time_ms([&]{ hipLaunchKernelGGL((k_update_occ<19, 4, true>), dim3((unsigned)((P/4+255)/256)), dim3(256), lds_bytes, 0, A, P); }), bytes);
	}
	{ const size_t P = N; rep("read only  19 planes dword", time_ms([&]{ hipLaunchKernelGGL((k_read<19, 1>), dim3((unsigned)((P+255)/256)), dim3(256), 0, 0, A, B, P); }), 76.0*N); }
	{ const size_t P = N; rep("read only  19 planes float4", time_ms([&]{ hipLaunchKernelGGL((k_read<19, 4>), dim3((unsigned)((P/4+255)/256)), dim3(256), 0, 0, A, B, P); }), 76.0*N); }
	{ const size_t P = N; rep("write only 19 planes dword", time_ms([&]{ hipLaunchKernelGGL((k_write<19, 1>), dim3((unsigned)((P+255)/256)), dim3(256), 0, 0, B, P); }), 76.0*N); }
	{ const size_t P = N; rep("write only 19 planes float4", time_ms([&]{ hipLaunchKernelGGL((k_write<19, 4>), dim3((unsigned)((P/4+255)/256)), dim3(256), 0, 0, B, P); }), 76.0*N); }
	return 0;
}
