/*
 * luw_core_dev.h -- measurement, test and A/B entry points of libluw_core.so.  NOT part of the drop-in boundary: nothing here replaces a member of the
 * reference's LBM class; a maintainer integrating the solver needs include/luw_core.h only.  bench.py, tests/ and tools/ use these.
 */
#ifndef LUW_CORE_DEV_H
#define LUW_CORE_DEV_H

#include "luw_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernel ids of the A/B and measurement-only variants.  They exist in the tools build only (make -C latticeurbanwind_amd/csrc ab -> tools/libluw_core_ab.so,
 * -DLUW_AB_KERNELS); the product library rejects them in luw_create / luw_set_kernel. */
#define LUW_KERNEL_VEC4 2               /* 4 cells / lane, one aligned access per lane and plane, wave64 lane shifts for x+1 populations */
#define LUW_KERNEL_VEC2 3               /* 2 cells / lane (FP16C: one dword per lane and plane) */
#define LUW_KERNEL_SCALAR_CACHED 4      /* scalar kernel with the default cache policy instead of non-temporal DDF accesses */
#define LUW_KERNEL_SCALAR_NT_ALL 5      /* scalar kernel with non-temporal accesses on all 19 planes (the product uses nt on the 14 aligned planes) */
#define LUW_KERNEL_VEC1 6               /* 1 cell / lane with aligned accesses + wave64 lane shifts for the x+1 populations */
#define LUW_KERNEL_SCALAR_GENERAL 8     /* scalar kernel without the wave-uniform "no TYPE_E, no force in this wave" fast path */
#define LUW_KERNEL_EXP_COPY 100         /* measurement only: scalar kernel's loads/stores without the collision (no physics) */
#define LUW_KERNEL_EXP_NOSHIFT 101      /* measurement only: scalar kernel with the x+1 neighbours replaced by x (no physics) */

/* ---- measurement (bench.py) */
/* runs `steps` steps like luw_run and returns the mean duration of the stream_collide kernel in milliseconds, taken with HIP events on the launch stream */
int luw_run_timed(luw_solver* s, uint64_t steps, double* mean_kernel_ms);
int luw_group_run_timed(luw_group* g, uint64_t steps, double* mean_kernel_ms);           /* mean duration of domain 0's interior (or whole-box) kernel */
/* means over the timed launches (luw_domain_step_launch with timed = 1) since the last call; waits for both streams */
int luw_domain_step_timing(luw_domain_step* d, double* kernel_ms, double* shell_ms);
/* what luw_create's placement search did for this solver: candidates probed (0: no search), algorithmic TB/s of the kept candidate's probe, seconds spent
 * in luw_create, and the kind of allocation kept ("1 GiB chunks", "2 GiB chunks", "hipMalloc", ...; "... (no search)" when none ran) */
int luw_dev_placement_info(const luw_solver* s, int* candidates_tried, double* probe_TBps, double* create_seconds, char* kept, uint64_t kept_size);
/* workgroup order of this solver's step kernels: lattice rows per XCD and turn (0: as dispatched; luw_create's rule or LUW_XCD_ROWS), -1 for a null solver */
int luw_dev_workgroup_order(const luw_solver* s);

/* ---- the tuning table: the library reads its environment knobs once, at first use */
int luw_dev_reload_tuning(void);                         /* read the environment again (tests and A/B tools that change it between two solvers) */
int luw_dev_tuning_text(char* text, uint64_t size);      /* the table in effect as "NAME=value NAME=value ..." (every product knob, in the documented order) */

/* ---- test access to the DDFs: copies the 19 planes to / from host memory in the reference's layout fi[i*N + n] (FX/kernel.cpp:877-879), raw storage
 * type (float or uint16_t FP16C codes) */
int luw_download_fi(luw_solver* s, void* host_dst);
int luw_download_gi(luw_solver* s, void* host_dst);   /* thermal DDFs as stored, gi[i*N+n], i = 0..6 */
int luw_upload_fi(luw_solver* s, const void* host_src);

/* ---- fault injection for the multi-domain host (first-contact insurance: the paths a node with real peer links takes when something is missing, exercised
 * on one GPU).  mask bit 0: luw_group_create treats every pair of domains (i, j) with i + j odd as devices WITHOUT peer access -- those pairs fall back
 * to staged copies while the others keep their peer stores; bit 1: ncclCommInitAll "fails" (LUW_GROUP_TRANSPORT=rccl then ends in a clean error, nothing
 * allocated, nothing hanging).  0 clears.  Takes effect for groups created afterwards. */
#define LUW_FAULT_NO_PEER_ODD_PAIRS 1u
#define LUW_FAULT_RCCL_INIT 2u
#define LUW_FAULT_SLOW_FIRST_PLACEMENT 4u   /* luw_create's placement search sees its first candidate 30 % slower than it is: another draw must replace it */
#define LUW_FAULT_UNPACK_WITHOUT_WAIT 8u    /* luw_group_*: unpack kernels do not wait for the neighbours' pack kernels (schedule fuzz: negative control) */
int luw_dev_inject_fault(uint32_t mask);
/* schedule fuzzing: from now on every second step / pack / unpack / edge kernel the library enqueues is held back on its stream by a delay kernel of
 * 1 .. max_us microseconds (drawn from `seed`; at most 5000; 0 switches it off).  Results must not change: one that depends on a kernel being faster than
 * another -- a missing event between two streams or two domains -- differs from the oracle under some seed (tests/test_gpu_schedule_jitter.py). */
int luw_dev_schedule_jitter(uint64_t seed, uint32_t max_us);

/* ---- device self-checks */
/* number of inputs (all 2^16 FP16C codes + all 2^32 floats) for which the kernels' fast FP16C codec differs from the literal formulas of
 * FX/kernel.cpp:864-875; must be 0 */
int luw_selfcheck_fp16c_codec(int device, uint64_t* mismatches);
/* the FP16C kernels' division and square root (the library's correctly rounded instruction sequences without their range handling, csrc/luw_device.hpp)
 * against `a/b` and sqrtf(): mismatches[0] square roots over every float of the range, [1] quotients for every denominator in [1/4, 4] x 64 numerators,
 * [2] the same with numerators on the 2^-25 grid of FP16C moment sums.  All three must be 0. */
int luw_selfcheck_arith(int device, uint64_t* mismatches);

#ifdef __cplusplus
}
#endif
#endif /* LUW_CORE_DEV_H */
