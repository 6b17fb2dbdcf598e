// luw_core.hip -- HIP kernels (gfx950) + C-ABI of the MI355X-native D3Q19 core.  See include/luw_core.h.
//
// Device code (included below):
//   luw_device.hpp          per-cell arithmetic: f_eq, moments, forces, the collision of one cell, thermal cell; with luw_codec.hpp (FP16C codec),
//                           luw_device_pair.hpp (the collision on packed pairs, exact) and luw_device_native.hpp (native arithmetic)
//   luw_kernels_common.hpp  slot algebra, k_initialize (f_eq(rho,u) -> Esoteric-Pull store with t=1, FX/kernel.cpp:1370-1452)
//   luw_kernels_step.hpp    k_stream_collide_s (1 cell per lane: FP32 product kernel, FP16C fallback, thermal lattice) and the shared addressing
//   luw_kernels_pair.hpp    k_stream_collide_p (FP16C product kernel: 2 cells per lane, packed FP32 collision)
//   luw_kernels_aux.hpp     k_extract_fi / k_insert_fi (halo pack/unpack, FX/kernel.cpp:2241-2270), voxeliser, probe gather,
//                           von-Karman inlet, statistics, codec self-check
// Host code (included behind it, in this order):
//   luw_host.hpp            error reporting, the tuning table (every environment knob, read once), the reference's 9-digit float text
//   luw_memory.hpp          device blocks (hipMalloc / chunk-mapped ranges), the solver object, lattice arrays, pitched host <-> device copies
//   luw_launch.hpp          kernel choice per box launch: force modes, x-face output, the tables of all stream_collide instantiations
//   luw_placement.hpp       luw_create's placement search for the DDF array
//   luw_api.hpp / luw_api_aux.hpp / luw_api_run.hpp   the C-ABI of one domain (include/luw_core.h, luw_core_dev.h)
//   luw_step.hpp            one domain's share of a decomposed step: boxes, launches on two streams, pipelining events (luw_domain_step_*; both hosts)
//   luw_group.hpp           the multi-domain host: exchange routes and step loop (+ luw_group_rccl.hpp), luw_group_api.hpp its C-ABI (luw_group_*),
//                           luw_export.hpp the VTK payloads produced on the devices
// This file is the translation unit: nothing but the includes.
//
// Memory layout in HBM: SoA planes fi[q][z][y][x] with x-pitch Px (multiple of 64) and plane stride Np=Px*Ny*Nz, every
// array shifted by a lead pad so that the first owned cell of a row starts a 256-byte block (lead_alloc);
// rho[Np], u[3][Np], flags[Np], F[3][Np] share the pitch.  Host mirrors keep the reference layout (pitch Nx).
#include "luw_device.hpp"
#include "../../include/luw_core_dev.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <functional>
#include <atomic>
#include <mutex>
#include <deque>
#include <algorithm>
#include <memory>
#include <dlfcn.h>
#include <chrono>

using namespace luw;

#include "luw_kernels_common.hpp"
#include "luw_kernels_step.hpp"
#include "luw_kernels_pair.hpp"
#ifdef LUW_AB_KERNELS   // tools build only (make ab; the header lives under tools/ab_kernels/): A/B and measurement-only kernel variants
#include LUW_AB_KERNELS
#endif
#include "luw_kernels_aux.hpp"

#include "luw_host.hpp"
#include "luw_memory.hpp"
#include "luw_launch.hpp"
#include "luw_placement.hpp"
#include "luw_api.hpp"
#include "luw_api_aux.hpp"
#include "luw_api_run.hpp"

#include "luw_step.hpp"
#include "luw_group.hpp"
#include "luw_group_api.hpp"
#include "luw_export.hpp"
