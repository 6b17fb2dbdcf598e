// xgmi_store_probe.hip -- first-contact probe (bare HIP): kernels of device SRC store into a hipMalloc'ed buffer of device DST in the three shapes the halo
// exchange of this repo uses, timed with events on SRC and checked on DST:
//   rows256  a pack kernel's stores: consecutive lanes write consecutive 4-byte elements of the five planes of a face buffer (256 B per wave and store)
//   one4     the x faces written by the FP32 step kernel (luw_kernels_step.hpp, xface_out): ONE lane of every 256-thread block stores one 4-byte element
//            per plane -- element = the block's row, so neighbouring rows fill a 128-byte line from different workgroups at different times
//   one2     the same from the FP16C kernels: 2-byte elements
// A face of a 512^3 domain is 5 planes x 262144 elements; `one*` launches one block per element like the step kernel does (its 255 other lanes idle here).
// What first contact needs from it (tools/first_contact.sh stage 2b): do 1.3 M scattered 2-4-byte remote stores per face and step cost more than a coalesced
// pack kernel + the same bytes?  On ONE device (SRC = DST) the numbers mean nothing for a wire; the run is a rehearsal of the program.
// usage: xgmi_store_probe <src device> <dst device> [elements per plane = 262144] [repetitions = 20]
// build: hipcc --offload-arch=gfx950 -O2 -o xgmi_store_probe xgmi_store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { const hipError_t e_ = (x); if(e_!=hipSuccess) { printf("FAILED %s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while(0)

template<typename T> __global__ __launch_bounds__(256) void k_rows(T* dst, const uint32_t A, const T tag) { // coalesced: every lane one element per plane
	const uint32_t e = blockIdx.x*blockDim.x+threadIdx.x;
	if(e>=A) return;
	#pragma unroll
	for(uint32_t b=0u; b<5u; b++) dst[(size_t)b*A+e] = (T)(tag+(T)(e%251u)+(T)b);
}
template<typename T> __global__ __launch_bounds__(256) void k_one(T* dst, const uint32_t A, const T tag) {  // one lane per block: element = the block
	const uint32_t e = blockIdx.x;
	if(threadIdx.x!=0u||e>=A) return;
	#pragma unroll
	for(uint32_t b=0u; b<5u; b++) dst[(size_t)b*A+e] = (T)(tag+(T)(e%251u)+(T)b);
}
template<typename T> static int check(const T* dev, const int dst_dev, const uint32_t A, const T tag, const char* what) {
	std::vector<T> h(5u*(size_t)A);
	CHECK(hipSetDevice(dst_dev));
	CHECK(hipMemcpy(h.data(), dev, h.size()*sizeof(T), hipMemcpyDeviceToHost));
	size_t bad = 0u;
	for(uint32_t b=0u; b<5u; b++) for(uint32_t e=0u; e<A; e++) bad += h[(size_t)b*A+e]!=(T)(tag+(T)(e%251u)+(T)b);
	if(bad) { printf("FAILED %s: %zu of %zu elements did not arrive\n", what, bad, h.size()); return 3; }
	return 0;
}
template<typename T, bool ROWS> static int shape(const char* name, const int src, const int dst, void* buf, const uint32_t A, const int reps, hipStream_t st) {
	CHECK(hipSetDevice(dst));
	CHECK(hipMemset(buf, 0, 5u*(size_t)A*sizeof(T)));
	CHECK(hipDeviceSynchronize());
	CHECK(hipSetDevice(src));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const dim3 grid(ROWS ? (A+255u)/256u : A), block(256);
	auto launch = [&](const T tag) {
		if(ROWS) hipLaunchKernelGGL((k_rows<T>), grid, block, 0, st, (T*)buf, A, tag); else hipLaunchKernelGGL((k_one<T>), grid, block, 0, st, (T*)buf, A, tag);
	};
	for(int k=0; k<3; k++) launch((T)1);
	CHECK(hipEventRecord(e0, st));
	for(int k=0; k<reps; k++) launch((T)(2+k%7));
	CHECK(hipEventRecord(e1, st));
	CHECK(hipEventSynchronize(e1));
	CHECK(hipGetLastError());
	float ms = 0.0f; CHECK(hipEventElapsedTime(&ms, e0, e1));
	const double us = (double)ms*1e3/reps, bytes = 5.0*A*sizeof(T);
	printf("%-8s %2zu-byte elements: %8.1f us per face  %8.2f GB/s  %8.1f M stores/s (%u elements x 5 planes, %d launches)\n", name, sizeof(T), us,
		bytes/us*1e-3, 5.0*A/us, A, reps);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	return check<T>((const T*)buf, dst, A, (T)(2+(reps-1)%7), name);
}
int main(int argc, char** argv) {
	const int src = argc>1 ? atoi(argv[1]) : 0, dst = argc>2 ? atoi(argv[2]) : 0;
	const uint32_t A = argc>3 ? (uint32_t)strtoul(argv[3], nullptr, 10) : 262144u;
	const int reps = argc>4 ? atoi(argv[4]) : 20;
	int n = 0; CHECK(hipGetDeviceCount(&n));
	if(src<0||dst<0||src>=n||dst>=n||A==0u||reps<1) { printf("FAILED: devices %d -> %d of %d, %u elements, %d repetitions\n", src, dst, n, A, reps); return 2; }
	if(src!=dst) {
		int can = 0; CHECK(hipDeviceCanAccessPeer(&can, src, dst));
		if(!can) { printf("FAILED: device %d cannot access device %d (no peer access: this pair takes the staged / RCCL transports)\n", src, dst); return 4; }
		CHECK(hipSetDevice(src));
		const hipError_t e = hipDeviceEnablePeerAccess(dst, 0u);
		if(e!=hipSuccess&&e!=hipErrorPeerAccessAlreadyEnabled) { printf("FAILED hipDeviceEnablePeerAccess: %s\n", hipGetErrorString(e)); return 2; }
		(void)hipGetLastError();
		uint32_t link = 0u, hops = 0u; (void)hipExtGetLinkTypeAndHopCount(src, dst, &link, &hops);
		printf("xgmi_store_probe: device %d -> device %d, link type %u, %u hop(s)\n", src, dst, link, hops);
	} else printf("xgmi_store_probe: device %d -> itself: a REHEARSAL of the program, the rates say nothing about a wire\n", src);
	CHECK(hipSetDevice(dst));
	void* buf = nullptr; CHECK(hipMalloc(&buf, 5u*(size_t)A*4u));
	CHECK(hipSetDevice(src));
	hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	if(int e = shape<float, true>("rows256", src, dst, buf, A, reps, st)) return e;
	if(int e = shape<float, false>("one4", src, dst, buf, A, reps, st)) return e;
	if(int e = shape<uint16_t, true>("rows128", src, dst, buf, A, reps, st)) return e;
	if(int e = shape<uint16_t, false>("one2", src, dst, buf, A, reps, st)) return e;
	printf("xgmi_store_probe: all values arrived\n");
	return 0;
}
