// vmm_unmap_repro.hip -- stand-alone check of how this ROCm's virtual-memory API wants a multi-chunk mapping torn down (round-3 luw_core.hip unmapped the
// WHOLE reserved range with one hipMemUnmap call and ignored the return code; releasing such a block in the middle of luw_create "crashed a later
// dev_free", and a second placement search "aborted inside the runtime": luw_core.hip:598-599, 615-618 of round 3).
//   build: hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_unmap_repro tools/vmm_unmap_repro.hip      run: /tmp/vmm_unmap_repro whole|chunk [rounds]
// Per round: block A (3 chunks of 256 MiB) and block B (same) are mapped and written by a kernel, B is torn down ("whole": one hipMemUnmap over the
// range; "chunk": one hipMemUnmap per mapped chunk), a block C is mapped and written, A and C are torn down the same way.  Prints every non-success
// return code and the free device memory before / after: a leak or an error names the faulty sequence.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
struct Block { void* base = nullptr; size_t total = 0, chunk = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
#define CHK(x) do { hipError_t e_ = (x); if(e_!=hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); errors++; } } while(0)
static int errors = 0;
__global__ void fill(unsigned* p, size_t n, unsigned v) { size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; for(; i<n; i += (size_t)gridDim.x*blockDim.x) p[i] = v; }
// bytes = 0: nchunk whole chunks; else whole chunks of `chunk` and a last piece cut to the remainder (2 MiB granularity), like luw_core.hip's dev_alloc
static bool map_block(Block& b, int dev, size_t chunk, int nchunk, size_t bytes = 0) {
	hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
	const size_t gran = 2ull<<20;
	b.chunk = chunk; b.total = bytes ? (bytes/chunk)*chunk+((bytes%chunk+gran-1)/gran)*gran : chunk*nchunk;
	CHK(hipMemAddressReserve(&b.base, b.total, chunk, nullptr, 0ull));
	for(size_t off=0; off<b.total; off+=chunk) {
		const size_t len = b.total-off<chunk ? b.total-off : chunk;
		hipMemGenericAllocationHandle_t h;
		CHK(hipMemCreate(&h, len, &prop, 0ull));
		CHK(hipMemMap((char*)b.base+off, len, 0u, h, 0ull));
		b.h.push_back(h);
	}
	hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
	CHK(hipMemSetAccess(b.base, b.total, &acc, 1u));
	fill<<<1024, 256>>>((unsigned*)b.base, b.total/4u, 7u);
	CHK(hipDeviceSynchronize());
	return true;
}
static void unmap_block(Block& b, bool whole) {
	if(whole) CHK(hipMemUnmap(b.base, b.total));
	else for(size_t k=0; k<b.h.size(); k++) CHK(hipMemUnmap((char*)b.base+k*b.chunk, b.total-k*b.chunk<b.chunk ? b.total-k*b.chunk : b.chunk));
	for(auto& h : b.h) CHK(hipMemRelease(h));
	CHK(hipMemAddressFree(b.base, b.total));
	b = Block{};
}
int main(int argc, char** argv) {
	const bool whole = argc>1&&strcmp(argv[1], "whole")==0;
	const int rounds = argc>2 ? atoi(argv[2]) : 4;
	CHK(hipSetDevice(0));
	size_t free0 = 0, tot = 0; CHK(hipMemGetInfo(&free0, &tot));
	if(argc>1&&strncmp(argv[1], "search", 6)==0) { // the allocation sequence of luw_create's placement search for a 2.6 GB DDF array, twice per "solver"
		const bool by_chunk = strcmp(argv[1], "search-chunk")==0;
		const size_t bytes = 19ull*(512ull*258*258+513*64)*4ull+256u;
		for(int r=0; r<rounds; r++) {
			Block A, B, D; void* C = nullptr;
			map_block(A, 0, 1024ull<<20, 0, bytes);
			map_block(B, 0, 2048ull<<20, 0, bytes); unmap_block(B, !by_chunk);
			CHK(hipMalloc(&C, bytes)); fill<<<1024, 256>>>((unsigned*)C, bytes/4u, 3u); CHK(hipDeviceSynchronize()); CHK(hipFree(C));
			map_block(D, 0, 512ull<<20, 0, bytes); unmap_block(D, !by_chunk);
			fill<<<1024, 256>>>((unsigned*)A.base, A.total/4u, 9u); CHK(hipDeviceSynchronize());
			unmap_block(A, !by_chunk);
			size_t f = 0; CHK(hipMemGetInfo(&f, &tot));
			printf("solver %d (%s): free %.1f MiB (start %.1f), errors so far %d\n", r, by_chunk ? "per chunk" : "whole range", f/1048576.0, free0/1048576.0, errors); fflush(stdout);
		}
		printf("done: %d errors\n", errors);
		return errors ? 1 : 0;
	}
	for(int r=0; r<rounds; r++) {
		Block A, B, C;
		map_block(A, 0, 256ull<<20, 3); map_block(B, 0, 256ull<<20, 3);
		unmap_block(B, whole);
		map_block(C, 0, 256ull<<20, 3);
		fill<<<1024, 256>>>((unsigned*)A.base, A.total/4u, 9u); CHK(hipDeviceSynchronize());
		unmap_block(A, whole); unmap_block(C, whole);
		size_t f = 0; CHK(hipMemGetInfo(&f, &tot));
		printf("round %d (%s): free %.1f MiB (start %.1f), errors so far %d\n", r, whole ? "one hipMemUnmap over the range" : "hipMemUnmap per chunk", f/1048576.0, free0/1048576.0, errors);
		fflush(stdout);
	}
	void* p = nullptr; CHK(hipMalloc(&p, 1ull<<30)); CHK(hipFree(p));
	printf("done: %d errors\n", errors);
	return errors ? 1 : 0;
}
