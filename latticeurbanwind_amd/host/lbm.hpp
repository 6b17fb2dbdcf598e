// lbm.hpp -- header-only C++ mirror of the reference's `LBM` class (FX/lbm.hpp:223-633) over the C-ABI of
// include/luw_core.h.  Same member names and argument meaning, so driver code written against the reference
// (FX/setup.cpp:6018-6078: construct, fill flags/u/rho through the global index, run(0), run(steps),
// u.read_from_device()) compiles against this header with `#include "lbm.hpp"` swapped in -- see INTEGRATION.md.
// Single domain per object (one process per GPU).  No HIP headers needed by the includer.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/luw_core.h"

namespace luw_host {

typedef unsigned int uint;
typedef unsigned char uchar;
typedef uint64_t ulong;

#define TYPE_S LUW_TYPE_S
#define TYPE_E LUW_TYPE_E
#define TYPE_T LUW_TYPE_T

inline void luw_check(const int rc) { // reference: print_error + exit(1), FX/utilities.hpp:4370-4382
	if(rc!=LUW_OK) { std::fprintf(stderr, "| Error: %s\n", luw_last_error()); std::exit(1); }
}

// process-global solver configuration the reference keeps in globals of setup.cpp / lbm.cpp (FX/setup.cpp:205-220)
struct SolverGlobals {
	bool buffer_nudging_active = false; int buffer_n_cells = 1; int buffer_downstream_face_id = 0; float buffer_inv_tau_lbmu = 0.0f; int buffer_nudge_vertical = 0;
	bool top_sponge_active = false; int sponge_n_cells = 1; float sponge_inv_tau_lbmu = 0.0f;
	bool fp16c = false; int device = 0;
};
inline SolverGlobals& solver_globals() { static SolverGlobals g; return g; }

class LBM {
	luw_solver* s = nullptr;
	uint Nx = 1u, Ny = 1u, Nz = 1u;
	bool initialized = false;
public:
	struct ScalarField { // Memory_Container<float/uchar>, FX/lbm.hpp:248-424
		LBM* lbm = nullptr; void* host = nullptr; uint32_t mask = 0u;
		template<typename T> T* data() { return static_cast<T*>(host); }
		void read_from_device() { luw_check(luw_download(lbm->s, mask)); }
		void write_to_device() { luw_check(luw_upload(lbm->s, mask)); }
	};
	struct FloatField : ScalarField { float& operator[](const ulong n) { return static_cast<float*>(host)[n]; } };
	struct FlagField : ScalarField { uchar& operator[](const ulong n) { return static_cast<uchar*>(host)[n]; } };
	struct VectorField : ScalarField { // lbm.u.x[n] etc., FX/lbm.hpp:359-372
		struct Component { float* p = nullptr; float& operator[](const ulong n) { return p[n]; } } x, y, z;
	};
	FloatField rho; VectorField u; FlagField flags; VectorField F; FloatField T;

	// LBM(Nx, Ny, Nz, Dx, Dy, Dz, nu, fx, fy, fz, sigma, alpha, beta), FX/lbm.hpp:444: only Dx=Dy=Dz=1 per object here
	// (multi-GPU = one object per process, latticeurbanwind_amd/distributed.py); sigma/alpha/beta belong to
	// extensions outside this path and must be 0 / are ignored (thermal lattice: DESIGN.md section 1)
	// alpha >= 0 switches the thermal D3Q7 lattice on (the reference's TEMPERATURE extension with LBM(..., alpha, beta); beta acts
	// through (fx,fy,fz), which LUW keeps at zero)
	LBM(const uint Nx_, const uint Ny_, const uint Nz_, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f, const bool force_field = false, const float alpha = -1.0f) {
		Nx = Nx_; Ny = Ny_; Nz = Nz_;
		const SolverGlobals& g = solver_globals();
		luw_config c = {};
		c.struct_size = sizeof(luw_config);
		c.Nx = Nx; c.Ny = Ny; c.Nz = Nz; c.Dx = c.Dy = c.Dz = 1u;
		c.nu = nu; c.fx = fx; c.fy = fy; c.fz = fz;
		c.ddf_format = g.fp16c ? LUW_DDF_FP16C : LUW_DDF_FP32;
		c.options = (force_field ? LUW_OPT_FORCE_FIELD : 0u)|(alpha>=0.0f ? LUW_OPT_TEMPERATURE : 0u);
		c.alpha = alpha>=0.0f ? alpha : 0.0f;
		c.buffer_nudging_active = g.buffer_nudging_active; c.buffer_n_cells = (uint32_t)g.buffer_n_cells; c.buffer_inv_tau_lbmu = g.buffer_inv_tau_lbmu;
		c.buffer_nudge_vertical = g.buffer_nudge_vertical; c.buffer_downstream_face_id = g.buffer_downstream_face_id;
		c.top_sponge_active = g.top_sponge_active; c.sponge_n_cells = (uint32_t)g.sponge_n_cells; c.sponge_inv_tau_lbmu = g.sponge_inv_tau_lbmu;
		c.device = g.device; c.kernel = LUW_KERNEL_AUTO;
		luw_check(luw_create(&c, &s));
		const ulong N = get_N();
		rho.lbm = this; rho.host = luw_host_ptr(s, LUW_FIELD_RHO); rho.mask = LUW_MASK_RHO;
		flags.lbm = this; flags.host = luw_host_ptr(s, LUW_FIELD_FLAGS); flags.mask = LUW_MASK_FLAGS;
		u.lbm = this; u.host = luw_host_ptr(s, LUW_FIELD_U); u.mask = LUW_MASK_U;
		float* up = static_cast<float*>(u.host); u.x.p = up; u.y.p = up+N; u.z.p = up+2ull*N;
		T.lbm = this; T.host = luw_host_ptr(s, LUW_FIELD_T); T.mask = LUW_MASK_T;
		F.lbm = this; F.host = luw_host_ptr(s, LUW_FIELD_F); F.mask = LUW_MASK_F;
		if(F.host) { float* fp = static_cast<float*>(F.host); F.x.p = fp; F.y.p = fp+N; F.z.p = fp+2ull*N; }
	}
	~LBM() { luw_destroy(s); }
	LBM(const LBM&) = delete; LBM& operator=(const LBM&) = delete;

	void run(const ulong steps = 0ull, const ulong total_steps = 0ull) { // FX/lbm.cpp:1292-1312; run(0) = upload + initialize
		(void)total_steps;
		if(!initialized) { luw_check(luw_initialize(s)); initialized = true; }
		if(steps>0ull) luw_check(luw_run(s, steps));
	}
	uint get_Nx() const { return Nx; } uint get_Ny() const { return Ny; } uint get_Nz() const { return Nz; }
	ulong get_N() const { return (ulong)Nx*(ulong)Ny*(ulong)Nz; }
	ulong get_t() const { return luw_get_t(s); }
	void set_f(const float fx, const float fy, const float fz) { luw_check(luw_set_f(s, fx, fy, fz)); }
	void set_coriolis(const float ox, const float oy, const float oz) { luw_check(luw_set_coriolis(s, ox, oy, oz)); }
	void coordinates(const ulong n, uint& x, uint& y, uint& z) const { const ulong t = n%((ulong)Nx*(ulong)Ny); x = (uint)(t%(ulong)Nx); y = (uint)(t/(ulong)Nx); z = (uint)(n/((ulong)Nx*(ulong)Ny)); }
	ulong index(const uint x, const uint y, const uint z) const { return (ulong)x+((ulong)y+(ulong)z*(ulong)Ny)*(ulong)Nx; }
	void position(const uint x, const uint y, const uint z, float& px, float& py, float& pz) const { // FX/lbm.hpp:523-525
		px = (float)x-0.5f*(float)Nx+0.5f; py = (float)y-0.5f*(float)Ny+0.5f; pz = (float)z-0.5f*(float)Nz+0.5f;
	}
	// lbm.voxelize_mesh_on_device(mesh, TYPE_S) for a static mesh, FX/lbm.hpp:560 / FX/lbm.cpp:1411: corners are float3
	// arrays (xyz triples) in lattice index coordinates, pmin/pmax the Mesh's bounds; result lands in flags[] (host mirror)
	void voxelize_mesh_on_device(const uint triangle_number, const float* p0, const float* p1, const float* p2, const float* pmin, const float* pmax, const uchar flag = 0x01) {
		const float bounds[6] = { pmin[0], pmin[1], pmin[2], pmax[0], pmax[1], pmax[2] };
		luw_check(luw_voxelize_mesh(s, triangle_number, p0, p1, p2, bounds, flag));
	}
	luw_solver* handle() { return s; }
};

} // namespace luw_host
