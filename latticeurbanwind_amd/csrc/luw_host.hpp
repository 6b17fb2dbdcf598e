// luw_host.hpp -- host side basics of libluw_core.so: error reporting, the tuning table (every environment knob, read once), the reference's
// 9-significant-digit float text.  Included by luw_core.hip only.
#pragma once

// =====================================================================================================
// host side
// =====================================================================================================
static thread_local std::string g_last_error;
static int fail(const int code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr) do { \
	const hipError_t e_ = (expr); \
	if(e_!=hipSuccess) return fail(LUW_ERR_DEVICE, std::string(#expr)+": "+hipGetErrorString(e_)); \
} while(0)

// ---------------------------------------------------------------- tuning table
// Every knob the library takes from the environment, read ONCE into this table at first use (luw_dev_reload_tuning() reads it again: tests and A/B tools
// that change the environment between two solvers of one process).  Nothing else in the library calls getenv, and no run / step path reads the
// environment.  INTEGRATION.md section 5 lists each knob with its default and purpose; tests/test_capi_library.py holds the two lists together.
struct Tuning {
	size_t alloc_chunk = 1024ull<<20; // LUW_ALLOC = vmm:<MiB> (physical chunk size of lattice-sized arrays) | vmm:one (~0: one piece) | malloc (0: hipMalloc)
	// LUW_TEST_AIDS=<comma list>: other product paths for the same values, taken on request by the tests that hold them to the default ones
	bool addr_row = false;            //   addr_row: FP32 kernel in the row addressing form also where the flat form would do
	bool fuse_stats = true;           //   separate_stats: sampled steps use the separate statistics kernel instead of the fused epilogue
	bool vk_ahead = true;             //   vk_inline: von-Karman inlet evaluated in line instead of one step ahead on a side stream
	bool voxelize_all = false;        //   voxelize_all: every voxeliser tile tests every triangle (no bins)
	int placement_candidates = -1;    // LUW_TUNE_PLACEMENT=<n>: allocations of the DDF array luw_create may try (0 / 1: no search; default 6)
	double placement_bar = 0.0;       // LUW_TUNE_FAST=<TB/s>: probe rate from which a placement is kept without further candidates (99: try all; test aid)
	uint32_t x_shell = 0u;            // LUW_X_SHELL=<cells>: thickness of the x boundary slabs of a decomposed step (0: 128; A/B aid)
	int group_transport = LUW_TRANSPORT_PEER; bool group_transport_bad = false; // LUW_GROUP_TRANSPORT = peer | staged | rccl (luw_group_create)
	bool group_sequential = false;    // LUW_GROUP_EXCHANGE=sequential: luw_group_* exchanges in the reference's three phases also where one round would do
	bool group_x_packed = false;      // LUW_GROUP_EXCHANGE=one_packed: one round, but the x faces through the pack / unpack kernels instead of the step kernels
	bool group_overlap = true;        // LUW_GROUP_OVERLAP=0: whole box as ONE launch, then the exchange (no boundary shell, no second stream in the step)
	uint64_t jitter_seed = 0ull; uint32_t jitter_us = 0u; // LUW_SCHEDULE_JITTER=<seed>:<max us>: schedule fuzzing from the first kernel on (schedule_jitter below)
	int xcd_rows = -1;                // LUW_XCD_ROWS=0 / 1: workgroup order of the step kernels (KParams::xcd_rows) for every lattice; unset: luw_create's rule
	bool group_threads = false;       // LUW_GROUP_THREADS=1: one host thread per domain in luw_group_run
#ifdef LUW_AB_KERNELS                 // tools build only
	int ab_kernel = -1;               // LUW_KERNEL=<id>: overrides the kernel choice of callers that expose none
	bool ab_pair_copy = false;        // LUW_PAIR_COPY: the pair kernel's memory path alone (no physics)
#endif
};
static std::atomic<uint32_t> g_injected_faults{0u};     // luw_dev_inject_fault (include/luw_core_dev.h): test hooks
// luw_dev_schedule_jitter: every second launch of a step / pack / unpack kernel is held back by a random delay on its stream.  A result that depends on one
// kernel being faster than another -- a missing event between two streams -- then shows as a difference from the oracle.
static std::atomic<uint64_t> g_jitter_state{0ull};
static std::atomic<uint32_t> g_jitter_max_us{0u};
static void schedule_jitter(hipStream_t st) {
	const uint32_t max_us = g_jitter_max_us.load(std::memory_order_relaxed);
	if(!max_us) return;
	uint64_t z = g_jitter_state.fetch_add(0x9E3779B97F4A7C15ull)+0x9E3779B97F4A7C15ull;   // splitmix64
	z = (z^(z>>30))*0xBF58476D1CE4E5B9ull; z = (z^(z>>27))*0x94D049BB133111EBull; z ^= z>>31;
	if(z&1ull) return;
	hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, st, 1u+(uint32_t)((z>>8)%max_us));
}
// Read ONCE per process, on first use (std::call_once: the first use may come from several threads at the same time -- the per-domain threads of
// luw_group, callers' thread pools).  luw_dev_reload_tuning (tests and A/B tools that change the environment between two solvers) reads it again; it must
// not run while another thread is inside the library.
static Tuning g_tuning;
static std::once_flag g_tuning_once;
static void tuning_load() {
	Tuning t;
	[[maybe_unused]] auto on = [](const char* n) { return getenv(n)!=nullptr; };
	if(const char* e = getenv("LUW_ALLOC")) {
		if(strncmp(e, "malloc", 6)==0) t.alloc_chunk = 0u;
		else if(strncmp(e, "vmm:one", 7)==0) t.alloc_chunk = ~(size_t)0u;
		else if(strncmp(e, "vmm:", 4)==0) { const size_t v = (size_t)strtoull(e+4, nullptr, 10); if(v) t.alloc_chunk = v<<20; }
	}
	if(const char* e = getenv("LUW_TEST_AIDS")) {
		const std::string list = std::string(",")+e+",";
		auto has = [&](const char* name) { return list.find(std::string(",")+name+",")!=std::string::npos; };
		t.addr_row = has("addr_row"); t.fuse_stats = !has("separate_stats"); t.vk_ahead = !has("vk_inline"); t.voxelize_all = has("voxelize_all");
	}
	if(const char* e = getenv("LUW_TUNE_PLACEMENT")) t.placement_candidates = atoi(e);
	if(const char* e = getenv("LUW_TUNE_FAST")) t.placement_bar = atof(e);
	if(const char* e = getenv("LUW_X_SHELL")) t.x_shell = (uint32_t)strtoul(e, nullptr, 10);
	if(const char* e = getenv("LUW_GROUP_TRANSPORT")) {
		if(strcmp(e, "rccl")==0) t.group_transport = LUW_TRANSPORT_RCCL;
		else if(strcmp(e, "staged")==0) t.group_transport = LUW_TRANSPORT_STAGED;
		else if(strcmp(e, "peer")!=0&&e[0]) t.group_transport_bad = true;
	}
	{ const char* e = getenv("LUW_GROUP_THREADS"); t.group_threads = e&&e[0]=='1'; }
	{ const char* e = getenv("LUW_GROUP_EXCHANGE"); t.group_sequential = e&&strcmp(e, "sequential")==0; t.group_x_packed = e&&strcmp(e, "one_packed")==0; }
	{ const char* e = getenv("LUW_GROUP_OVERLAP"); t.group_overlap = !(e&&e[0]=='0'); }
	if(const char* e = getenv("LUW_XCD_ROWS")) t.xcd_rows = (e[0]>='0'&&e[0]<='9') ? std::min(atoi(e), 64) : -1;
	if(const char* e = getenv("LUW_SCHEDULE_JITTER")) {
		char* end = nullptr;
		t.jitter_seed = strtoull(e, &end, 10);
		t.jitter_us = (end&&*end==':') ? (uint32_t)std::min<unsigned long>(strtoul(end+1, nullptr, 10), 5000ul) : 0u;
		g_jitter_state.store(t.jitter_seed); g_jitter_max_us.store(t.jitter_us);
	}
#ifdef LUW_AB_KERNELS
	if(const char* e = getenv("LUW_KERNEL")) t.ab_kernel = atoi(e);
	t.ab_pair_copy = on("LUW_PAIR_COPY");
#endif
	g_tuning = t;
}
static const Tuning& tuning() { std::call_once(g_tuning_once, tuning_load); return g_tuning; }

// ---- floats as 9-significant-digit text.  The reference bakes its kernel constants into OpenCL source as decimal text and writes
// VTK headers the same way (to_string(float), FX/utilities.hpp:2603-2634,2741-2750; used at FX/lbm.cpp:664,774,780): what the
// kernel computes with is the float READ BACK from that text, so the digits have to be the reference's, one float operation at a
// time.  Decimal exponent: a binary ladder of powers of ten, each rung applied at most once, from 10^32 down to 10^1 -- scaling
// down while the value is >= 10, scaling up while it is below 1 (the thresholds of the upward ladder sit one decade lower, so
// that the mantissa ends in [1, 10)).  Digits: the integer part, then eight decimals from one truncation of (x - int) * 10^8
// and one round-half-up whose carry may run into the integer part and the exponent.
struct Decimal9 { bool negative, special; uint32_t integral, decimals; int exponent; };
static Decimal9 split_decimal9(float x) {
	Decimal9 d = { x<0.0f, false, 0u, 0u, 0 };
	if(d.negative) x = -x;
	if(std::isnan(x)||std::isinf(x)) { d.special = true; return d; }
	static const struct {
		float at_least, times;
		int decades;
	} down[6] = { { 1E32f, 1E-32f, 32 }, { 1E16f, 1E-16f, 16 }, { 1E8f, 1E-8f, 8 }, { 1E4f, 1E-4f, 4 }, { 1E2f, 1E-2f, 2 }, { 1E1f, 1E-1f, 1 } };
	static const struct {
		float below, times;
		int decades;
	} up[6] = { { 1E-31f, 1E32f, 32 }, { 1E-15f, 1E16f, 16 }, { 1E-7f, 1E8f, 8 }, { 1E-3f, 1E4f, 4 }, { 1E-1f, 1E2f, 2 }, { 1E0f, 1E1f, 1 } };
	if(x>=10.0f) for(const auto& r : down) if(x>=r.at_least) { x *= r.times; d.exponent += r.decades; }
	if(x>0.0f&&x<=1.0f) for(const auto& r : up) if(x<r.below) { x *= r.times; d.exponent -= r.decades; }
	d.integral = (uint32_t)x;
	const float scaled = (x-(float)d.integral)*1E8f;
	d.decimals = (uint32_t)scaled;
	if(scaled-(float)d.decimals>=0.5f&&++d.decimals>=100000000u) { // half up; 0.99999999x carries
		d.decimals = 0u;
		if(++d.integral>=10u) { d.integral = 1u; d.exponent++; }
	}
	return d;
}
static void format_decimal9(const float x, char* text, const size_t size) {
	const Decimal9 d = split_decimal9(x);
	const char* sign = d.negative ? "-" : "";
	if(d.special) snprintf(text, size, "%s%s", sign, std::isnan(x) ? "NaN" : "Inf");
	else if(d.exponent!=0) snprintf(text, size, "%s%u.%08uE%d", sign, d.integral, d.decimals, d.exponent);
	else snprintf(text, size, "%s%u.%08u", sign, d.integral, d.decimals);
}
// the float the reference kernel sees after device_defines() printed it and the OpenCL compiler parsed it
static float literal_roundtrip(const float x) {
	if(std::isnan(x)||std::isinf(x)) return x;
	char text[48];
	format_decimal9(x, text, sizeof(text));
	return strtof(text, nullptr);
}
