// lds_dma_probe.hip -- what global_load_lds_* does on gfx950 for the cases the FP16C pair kernel needs: a dword on a 2-byte boundary,
// a single active lane, a 16-bit element.  hipcc --offload-arch=gfx950 -O2 -o /tmp/ldp tools/lds_dma_probe.hip && /tmp/ldp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(64) void probe(const uint16_t* __restrict__ src, uint32_t* __restrict__ out) {
	__shared__ uint32_t buf[4][80];
	const int l = threadIdx.x;
	for(int k=0; k<4; k++) for(int i=l; i<80; i+=64) buf[k][i] = 0xDEAD0000u+i;
	__syncthreads();
	const char* base = reinterpret_cast<const char*>(src);
	// 0: aligned dword per lane
	__builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(base+4u*l), &buf[0][0], 4, 0, 0);
	// 1: dword on a 2-byte boundary per lane
	__builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(base+2u+4u*l), &buf[1][0], 4, 0, 0);
	// 2: one active lane (63), dword, from element 1000
	if(l==63) __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(base+2000u), &buf[2][0], 4, 0, 0);
	// 3: 16-bit elements, every lane: element 500+l
	__builtin_amdgcn_global_load_lds(reinterpret_cast<const uint16_t*>(base+1000u+2u*l), &buf[3][0], 2, 0, 0);
	__builtin_amdgcn_s_waitcnt(0x0f70);
	__syncthreads();
	for(int k=0; k<4; k++) for(int i=l; i<80; i+=64) out[k*80+i] = buf[k][i];
}
int main() {
	std::vector<uint16_t> h(4096); for(int i=0; i<4096; i++) h[i] = (uint16_t)i;
	uint16_t* d; uint32_t* o; hipMalloc(&d, 8192); hipMalloc(&o, 4*80*4); hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o); hipDeviceSynchronize();
	std::vector<uint32_t> r(320); hipMemcpy(r.data(), o, 1280, hipMemcpyDeviceToHost);
	const char* name[4] = { "aligned dword, lane l <- elements (2l, 2l+1)", "dword at +2 bytes, lane l <- elements (2l+1, 2l+2)", "only lane 63 active, dword of elements (1000, 1001)", "16-bit, lane l <- element 500+l" };
	for(int k=0; k<4; k++) { printf("%s\n", name[k]); for(int i=0; i<68; i++) printf("%08x%s", r[k*80+i], i%8==7 ? "\n" : " "); printf("\n"); }
	return 0;
}
