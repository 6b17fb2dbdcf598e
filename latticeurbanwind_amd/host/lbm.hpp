// lbm.hpp -- header-only C++ mirror of the reference's `LBM` class (FX/lbm.hpp:223-633) over the C-ABI of
// include/luw_core.h.  Same constructors, member names and argument meaning, so driver code written against the reference
// (FX/setup.cpp:6018-6078: `LBM lbm(lbm_N, Dx, Dy, Dz, lbm_nu, 0.0f, 0.0f, 0.0f, 0.0f, lbm_alpha, lbm_beta);`, fill
// flags/u/rho through the global index, run(0), run(steps), u.read_from_device()) compiles against this header with
// `#include "lbm.hpp"` swapped in -- see INTEGRATION.md.
// Like the reference's object it owns ALL Dx*Dy*Dz domains in one process (luw_group_*: one HIP device and stream pair per
// domain, halos between them inside the library); the host arrays are indexed globally, n = x + (y + z*Ny)*Nx.
// No HIP headers needed by the includer.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/luw_core.h"

namespace luw_host {

typedef unsigned int uint;
typedef unsigned char uchar;
typedef uint64_t ulong;
// FX/utilities.hpp uint3, as far as LBM's constructors need it
struct uint3 { uint x, y, z; uint3(const uint x_ = 0u, const uint y_ = 0u, const uint z_ = 0u) : x(x_), y(y_), z(z_) {} };

#define TYPE_S LUW_TYPE_S
#define TYPE_E LUW_TYPE_E
#define TYPE_T LUW_TYPE_T

inline void luw_check(const int rc) { // reference: print_error + exit(1), FX/utilities.hpp:4370-4382
	if(rc!=LUW_OK) { std::fprintf(stderr, "| Error: %s\n", luw_last_error()); std::exit(1); }
}

// process-global solver configuration: what the reference keeps in globals of setup.cpp / lbm.cpp (FX/setup.cpp:205-220) and in
// compile-time defines of its build (FX/defines.hpp: FP16C, TEMPERATURE), as run-time switches
struct SolverGlobals {
	bool buffer_nudging_active = false;
	int buffer_n_cells = 1;
	int buffer_downstream_face_id = 0;
	float buffer_inv_tau_lbmu = 0.0f;
	int buffer_nudge_vertical = 0;
	bool top_sponge_active = false; int sponge_n_cells = 1; float sponge_inv_tau_lbmu = 0.0f;
	bool fp16c = false;          // #define FP16C
	bool native_arith = false;   // LUW_OPT_NATIVE_ARITH (FP16C collision in the hardware's own arithmetic, like the reference's -cl-mad-enable build)
	bool temperature = false;    // #define TEMPERATURE: the alpha passed to the constructor takes effect (thermal D3Q7 lattice, lbm.T)
	bool force_field = false;    // #define FORCE_FIELD: allocate lbm.F (LUW never writes it)
	int device = 0;              // first HIP device; domain d runs on device + d unless `devices` says otherwise
	std::vector<int> devices;    // explicit device per domain (several domains on one GPU: test set-ups)
	uint32_t kernel = LUW_KERNEL_AUTO;
};
inline SolverGlobals& solver_globals() { static SolverGlobals g; return g; }

class LBM {
	luw_group* g = nullptr;
	luw_solver* s0 = nullptr;     // the only domain when D = 1: its mirrors ARE the host arrays
	uint Nx = 1u, Ny = 1u, Nz = 1u, Dx = 1u, Dy = 1u, Dz = 1u;
	bool initialized = false;
	std::unique_ptr<char[]> global_store[6]; // D > 1: global host arrays by LUW_FIELD_* id
	void construct(const uint Nx_, const uint Ny_, const uint Nz_, const uint Dx_, const uint Dy_, const uint Dz_, const float nu, const float fx,
		const float fy, const float fz, const float sigma, const float alpha, const float beta) {
		const uint NDx = (Nx_/Dx_)*Dx_, NDy = (Ny_/Dy_)*Dy_, NDz = (Nz_/Dz_)*Dz_; // make resolution equally divisible by domains, FX/lbm.cpp:1058-1060
		if(NDx!=Nx_||NDy!=Ny_||NDz!=Nz_)
			std::printf("| Warning: LBM grid (%ux%ux%u) is not equally divisible in domains (%ux%ux%u). Changing resolution to (%ux%ux%u).\n", Nx_, Ny_, Nz_,
			Dx_, Dy_, Dz_, NDx, NDy, NDz);
		Nx = NDx; Ny = NDy; Nz = NDz; Dx = Dx_; Dy = Dy_; Dz = Dz_;
		const SolverGlobals& G = solver_globals();
		if(sigma!=0.0f) { std::fprintf(stderr, "| Error: surface tension (SURFACE extension) is not part of this solver.\n"); std::exit(1); }
		if(beta!=0.0f&&(fx!=0.0f||fy!=0.0f||fz!=0.0f)&&G.temperature) {
			std::fprintf(stderr, "| Error: buoyancy (beta with a non-zero volume force) is not part of this solver; LUW runs with (fx,fy,fz) = 0.\n");
			std::exit(1);
		}
		luw_config c = {};
		c.struct_size = sizeof(luw_config);
		c.Nx = Nx; c.Ny = Ny; c.Nz = Nz; c.Dx = Dx; c.Dy = Dy; c.Dz = Dz;
		c.nu = nu; c.fx = fx; c.fy = fy; c.fz = fz;
		c.ddf_format = G.fp16c ? LUW_DDF_FP16C : LUW_DDF_FP32;
		c.options = (G.force_field ? LUW_OPT_FORCE_FIELD : 0u)|(G.temperature ? LUW_OPT_TEMPERATURE : 0u)|(G.native_arith ? LUW_OPT_NATIVE_ARITH : 0u);
		c.alpha = G.temperature ? alpha : 0.0f;
		c.buffer_nudging_active = G.buffer_nudging_active; c.buffer_n_cells = (uint32_t)G.buffer_n_cells; c.buffer_inv_tau_lbmu = G.buffer_inv_tau_lbmu;
		c.buffer_nudge_vertical = G.buffer_nudge_vertical; c.buffer_downstream_face_id = G.buffer_downstream_face_id;
		c.top_sponge_active = G.top_sponge_active; c.sponge_n_cells = (uint32_t)G.sponge_n_cells; c.sponge_inv_tau_lbmu = G.sponge_inv_tau_lbmu;
		c.device = G.device; c.kernel = G.kernel;
		const uint D = Dx*Dy*Dz;
		if(!G.devices.empty()&&G.devices.size()!=D) {
			std::fprintf(stderr, "| Error: the device list names %zu devices for %u domains.\n", G.devices.size(), D);
			std::exit(1);
		}
		luw_check(luw_group_create(&c, G.devices.empty() ? nullptr : G.devices.data(), &g));
		const ulong N = get_N();
		auto bind = [&](ScalarField& f, const int field, const uint32_t mask, const uint comps, const size_t elem, const float fill) {
			f.lbm = this; f.field = field; f.mask = mask;
			if(D==1u) { f.host = luw_host_ptr(s0, field); return; }
			if(!luw_host_ptr(luw_group_domain(g, 0u), field)) { f.host = nullptr; return; }
			global_store[field].reset(new char[(size_t)N*comps*elem]);
			f.host = global_store[field].get();
			if(elem==4u) {
				float* p = static_cast<float*>(f.host);
				for(ulong n=0ull; n<N*comps; n++) p[n] = fill;
			} else std::memset(f.host, 0, (size_t)N*comps);
		};
		if(D==1u) s0 = luw_group_domain(g, 0u);
		bind(rho, LUW_FIELD_RHO, LUW_MASK_RHO, 1u, 4u, 1.0f); // Memory<float>(device, N, 1u, true, true, 1.0f), FX/lbm.cpp:286
		bind(u, LUW_FIELD_U, LUW_MASK_U, 3u, 4u, 0.0f);
		bind(flags, LUW_FIELD_FLAGS, LUW_MASK_FLAGS, 1u, 1u, 0.0f);
		bind(F, LUW_FIELD_F, LUW_MASK_F, 3u, 4u, 0.0f);
		bind(T, LUW_FIELD_T, LUW_MASK_T, 1u, 4u, 1.0f);
		float* up = static_cast<float*>(u.host); u.x.p = up; u.y.p = up+N; u.z.p = up+2ull*N;
		if(F.host) { float* fp = static_cast<float*>(F.host); F.x.p = fp; F.y.p = fp+N; F.z.p = fp+2ull*N; }
	}
public:
	struct ScalarField { // Memory_Container<float/uchar>, FX/lbm.hpp:248-424: one global index space over the domains' host buffers
		LBM* lbm = nullptr; void* host = nullptr; uint32_t mask = 0u; int field = 0;
		template<typename T> T* data() { return static_cast<T*>(host); }
		void read_from_device() { // FX/lbm.hpp:406-412
			if(lbm->s0) { luw_check(luw_download(lbm->s0, mask)); return; }
			luw_check(luw_group_download(lbm->g, mask)); luw_check(luw_group_gather(lbm->g, field, host));
		}
		void write_to_device() { // FX/lbm.hpp:413-416
			if(lbm->s0) { luw_check(luw_upload(lbm->s0, mask)); return; }
			luw_check(luw_group_scatter(lbm->g, field, host)); luw_check(luw_group_upload(lbm->g, mask));
		}
	};
	struct FloatField : ScalarField { float& operator[](const ulong n) { return static_cast<float*>(host)[n]; } };
	struct FlagField : ScalarField { uchar& operator[](const ulong n) { return static_cast<uchar*>(host)[n]; } };
	struct VectorField : ScalarField { // lbm.u.x[n] etc., FX/lbm.hpp:359-372
		struct Component { float* p = nullptr; float& operator[](const ulong n) { return p[n]; } } x, y, z;
	};
	FloatField rho; VectorField u; FlagField flags; VectorField F; FloatField T;

	// the reference's constructors, FX/lbm.hpp:444-450 (the particle overloads belong to an extension outside this path).
	// sigma must be 0 (SURFACE is not compiled in the shipped build either); alpha takes effect when the thermal lattice is on
	// (SolverGlobals::temperature, the build's TEMPERATURE define); beta acts through (fx,fy,fz), which LUW keeps at zero.
	LBM(const uint Nx_, const uint Ny_, const uint Nz_, const uint Dx_, const uint Dy_, const uint Dz_, const float nu, const float fx = 0.0f,
		const float fy = 0.0f, const float fz = 0.0f, const float sigma = 0.0f, const float alpha = 0.0f, const float beta = 0.0f) {
		construct(Nx_, Ny_, Nz_, Dx_, Dy_, Dz_, nu, fx, fy, fz, sigma, alpha, beta);
	}
	LBM(const uint Nx_, const uint Ny_, const uint Nz_, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f,
		const float sigma = 0.0f, const float alpha = 0.0f, const float beta = 0.0f) {
		construct(Nx_, Ny_, Nz_, 1u, 1u, 1u, nu, fx, fy, fz, sigma, alpha, beta);
	}
	LBM(const uint3 N, const uint Dx_, const uint Dy_, const uint Dz_, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f,
		const float sigma = 0.0f, const float alpha = 0.0f, const float beta = 0.0f) {
		construct(N.x, N.y, N.z, Dx_, Dy_, Dz_, nu, fx, fy, fz, sigma, alpha, beta);
	}
	LBM(const uint3 N, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f, const float sigma = 0.0f, const float alpha = 0.0f,
		const float beta = 0.0f) {
		construct(N.x, N.y, N.z, 1u, 1u, 1u, nu, fx, fy, fz, sigma, alpha, beta);
	}
	~LBM() { luw_group_destroy(g); }
	LBM(const LBM&) = delete; LBM& operator=(const LBM&) = delete;

	void run(const ulong steps = 0ull, const ulong total_steps = 0ull) { // FX/lbm.cpp:1292-1312; run(0) = upload + initialize
		(void)total_steps;
		if(!initialized) {
			if(!s0) for(ScalarField* f : { (ScalarField*)&rho, (ScalarField*)&u, (ScalarField*)&flags, (ScalarField*)&F, (ScalarField*)&T }) if(f->host)
				luw_check(luw_group_scatter(g, f->field, f->host));
			luw_check(luw_group_initialize(g)); initialized = true;
		}
		if(steps>0ull) luw_check(luw_group_run(g, steps));
	}
	uint get_Nx() const { return Nx; } uint get_Ny() const { return Ny; } uint get_Nz() const { return Nz; }
	uint get_Dx() const { return Dx; } uint get_Dy() const { return Dy; } uint get_Dz() const { return Dz; } uint get_D() const { return Dx*Dy*Dz; }
	ulong get_N() const { return (ulong)Nx*(ulong)Ny*(ulong)Nz; }
	ulong get_t() const { return luw_group_get_t(g); }
	void set_f(const float fx, const float fy, const float fz) { luw_check(luw_group_set_f(g, fx, fy, fz)); }
	void set_coriolis(const float ox, const float oy, const float oz) { luw_check(luw_group_set_coriolis(g, ox, oy, oz)); }
	void coordinates(const ulong n, uint& x, uint& y, uint& z) const {
		const ulong t = n%((ulong)Nx*(ulong)Ny);
		x = (uint)(t%(ulong)Nx);
		y = (uint)(t/(ulong)Nx);
		z = (uint)(n/((ulong)Nx*(ulong)Ny));
	}
	ulong index(const uint x, const uint y, const uint z) const { return (ulong)x+((ulong)y+(ulong)z*(ulong)Ny)*(ulong)Nx; }
	void position(const uint x, const uint y, const uint z, float& px, float& py, float& pz) const { // FX/lbm.hpp:523-525
		px = (float)x-0.5f*(float)Nx+0.5f; py = (float)y-0.5f*(float)Ny+0.5f; pz = (float)z-0.5f*(float)Nz+0.5f;
	}
	// lbm.voxelize_mesh_on_device(mesh, TYPE_S) for a static mesh, FX/lbm.hpp:560 / FX/lbm.cpp:1411: corners are float3
	// arrays (xyz triples) in lattice index coordinates, pmin/pmax the Mesh's bounds; every domain voxelises its own box
	// (FX/lbm.cpp:1455-1587); the result lands in flags[]
	void voxelize_mesh_on_device(const uint triangle_number, const float* p0, const float* p1, const float* p2, const float* pmin, const float* pmax,
		const uchar flag = 0x01) {
		const float bounds[6] = { pmin[0], pmin[1], pmin[2], pmax[0], pmax[1], pmax[2] };
		if(s0) { luw_check(luw_voxelize_mesh(s0, triangle_number, p0, p1, p2, bounds, flag)); return; }
		luw_check(luw_group_scatter(g, LUW_FIELD_FLAGS, flags.host)); luw_check(luw_group_scatter(g, LUW_FIELD_U, u.host));
		luw_check(luw_group_voxelize_mesh(g, triangle_number, p0, p1, p2, bounds, flag));
		luw_check(luw_group_gather(g, LUW_FIELD_FLAGS, flags.host));
	}
	// ---- what LUW's run loop does around the solver (FX/setup.cpp:4117-4911), on the device(s) here
	// von-Karman inlet tables with GLOBAL cell indices: every domain takes the points it owns (FX/setup.cpp:1012-1057)
	void vk_inlet_attach(const uint64_t point_count, const uint64_t mode_count, const uint64_t* point_cell, const uint8_t* point_face, const float* point_data,
		const float* mode_data, const int update_stride, const int stride_interpolation) {
		if(s0) luw_check(luw_vk_inlet_attach(s0, point_count, mode_count, point_cell, point_face, point_data, mode_data, update_stride, stride_interpolation));
		else luw_check(luw_group_vk_inlet_attach(g, point_count, mode_count, point_cell, point_face, point_data, mode_data, update_stride,
			stride_interpolation));
	}
	void stats_reset() { luw_check(luw_group_stats_reset(g)); }
	void run_sampled(const ulong steps, const ulong first_sample, const ulong stride) { luw_check(luw_group_run_sampled(g, steps, first_sample, stride)); }
	// layout of write_avg_vtk (FX/setup.cpp:2513-2683): avg_u AoS [3n+c], the others [n]; avg_T may be null
	void stats_download(float* avg_u, float* avg_rho, float* m2_u, float* m2_v, float* m2_w, float* avg_T, uint64_t* count) {
		if(s0) { luw_check(luw_stats_download(s0, avg_u, avg_rho, m2_u, m2_v, m2_w, count)); if(avg_T) luw_check(luw_stats_download_T(s0, avg_T)); }
		else luw_check(luw_group_stats_download(g, avg_u, avg_rho, m2_u, m2_v, m2_w, avg_T, count));
	}
	uint64_t stats_count() const { return luw_group_stats_count(g); }
	// Memory_Container::write_vtk's payload (FX/lbm.hpp:330-356) and the sections of write_avg_vtk without the full-field download: every domain
	// converts its own cells on its device (SoA -> AoS, SI units, big-endian), a writer thread puts the slabs into the open file (luw_group_export_vtk)
	void export_vtk(const int source, const luw_export_params& prm, const uint Nz_write, const int fd, const uint64_t file_offset) {
		luw_check(luw_group_export_vtk(g, source, &prm, Nz_write, fd, file_offset));
	}
	void gather_attach(const uint32_t count, const uint64_t* cells) { luw_check(luw_group_gather_attach(g, count, cells)); }
	void gather_u(float* out) { luw_check(luw_group_gather_u(g, out)); }
	luw_group* group() { return g; }
	luw_solver* domain(const uint d) { return luw_group_domain(g, d); } // lbm.lbm_domain[d]
};

} // namespace luw_host
