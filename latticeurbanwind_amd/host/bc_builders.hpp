// Boundary builders of the NWP (*.luw) mode and the flux correction shared with the profile mode.
// Host-side set-up, written against raw host mirrors (flags u8[N], u SoA f32[3N], n = x+(y+z*Ny)*Nx) so that the same code
// serves the driver and tests.  Arithmetic follows the reference statement by statement where values depend on it
// (FP32 on the host without contraction, doubles where the reference uses doubles):
//   SurfData CSV reader            FX/setup.cpp:2293-2462
//   patch-driven 2-D face fields   FX/setup.cpp:1796-2094 (PatchSurfaceField2D), fill :5120-5267
//   nearest-sample inlet           FX/interpolation.cpp:52-69, fill :71-209
//   KNN-HD inlet (high_order)      FX/interpolation_hd.cpp:57-411 (K = 64 on the nearest outer plane, Gaussian-weighted
//                                  6-term quadratic least squares in double, partial pivoting), fill :437-745
//   flux correction                FX/fluxcorrection.cpp:28-194
// Temperature boundaries (buoyancy = true and a T column in the CSV): apply_*_temperature below fill lbm.T of TYPE_T cells.
// The inputs (CSV reader, face fields, ground plane, DEM) are bc_inputs.hpp.
#pragma once
#include "bc_inputs.hpp"
namespace luw_host {

// view of the solver's host mirrors
struct HostLattice {
	uint32_t Nx = 1u, Ny = 1u, Nz = 1u; uint8_t* flags = nullptr; float* u = nullptr; // u SoA: x[N], y[N], z[N]
	uint64_t N() const { return (uint64_t)Nx*Ny*Nz; }
	void coords(const uint64_t n, uint32_t& x, uint32_t& y, uint32_t& z) const {
		const uint64_t t = n%((uint64_t)Nx*Ny);
		x = (uint32_t)(t%Nx);
		y = (uint32_t)(t/Nx);
		z = (uint32_t)(n/((uint64_t)Nx*Ny));
	}
	uint64_t index(const uint32_t x, const uint32_t y, const uint32_t z) const { return (uint64_t)x+((uint64_t)y+(uint64_t)z*Ny)*Nx; }
	V3 position(const uint32_t x, const uint32_t y, const uint32_t z) const {
		V3 p;
		p.x = (float)x-0.5f*(float)Nx+0.5f;
		p.y = (float)y-0.5f*(float)Ny+0.5f;
		p.z = (float)z-0.5f*(float)Nz+0.5f;
		return p;
	}
	void set_u(const uint64_t n, const V3& v) { const uint64_t M = N(); u[n] = v.x; u[M+n] = v.y; u[2ull*M+n] = v.z; }
};

inline unsigned bc_worker_threads() {
	unsigned hw = std::thread::hardware_concurrency(); if(hw==0u) hw = 4u;
	if(const char* e = std::getenv("LBM_NUM_THREADS")) { const long v = std::strtol(e, nullptr, 10); if(v>0) hw = (unsigned)std::min<long>(v, (long)hw); }
	return hw;
}
template<typename F> inline void bc_parallel_for(const uint64_t n, F body) { // static contiguous split like FX/utilities.hpp:64-97
	const unsigned T = (unsigned)std::max<uint64_t>(1ull, std::min<uint64_t>(bc_worker_threads(), n));
	std::vector<std::thread> th; th.reserve(T);
	for(unsigned t=0u; t<T; t++) th.emplace_back([=]() { for(uint64_t i=n*t/T; i<n*(t+1u)/T; i++) body(i); });
	for(auto& x : th) x.join();
}

struct PatchBcCounts { uint64_t mapped = 0ull, missing = 0ull, outlet = 0ull, grounded = 0ull, below_support = 0ull, terrain_clipped = 0ull; };

// Patch-driven 2-D boundary mapping (samples carry a patch id), FX/setup.cpp:5120-5267.  `fields[1..5]` are the face velocity
// fields; `ground` is built from patch 0 with value z (lattice units).  side_ref_z_cap: side cells above the core top sample
// the profile at the cap height (top-sponge grid extension).
inline PatchBcCounts apply_patch_boundaries(HostLattice& L, const std::vector<PatchField2D>& fields, const PatchField2D& ground,
	const std::string& downstream_bc, const bool downstream_open_face, const int side_ref_z_cap) {
	PatchBcCounts out;
	const uint32_t Nx = L.Nx, Ny = L.Ny, Nz = L.Nz; const uint64_t N = L.N();
	const int downstream_patch = downstream_to_patch(downstream_bc);
	const V3 zero{};
	if(ground.has_samples()) { // cells under the terrain surface become solid whatever the STL says
		std::atomic<uint64_t> clipped{0ull};
		bc_parallel_for(N, [&](const uint64_t n) {
			if(L.flags[n]&0x01u) return;
			uint32_t x, y, z; L.coords(n, x, y, z);
			const V3 p = L.position(x, y, z);
			if(p.z<ground.eval(p.x, p.y).x) { L.flags[n] = 0x01u; L.set_u(n, zero); clipped.fetch_add(1ull, std::memory_order_relaxed); }
		});
		out.terrain_clipped = clipped.load();
	}
	const std::vector<uint8_t> vox(L.flags, L.flags+N); // classification reads the pre-fill flags
	auto solid = [&](const uint32_t x, const uint32_t y, const uint32_t z) { return (vox[L.index(x, y, z)]&0x01u)!=0u; };
	std::atomic<uint64_t> mapped{0ull}, missing{0ull}, outlet{0ull}, grounded{0ull}, below{0ull};
	bc_parallel_for(N, [&](const uint64_t n) {
		uint32_t x, y, z; L.coords(n, x, y, z);
		if(z==0u) { L.flags[n] = 0x01u; L.set_u(n, zero); return; }
		const int patch = boundary_cell_to_patch(x, y, z, Nx, Ny, Nz);
		if(patch<0) return;
		bool under = false; // the cell next to it towards the interior is solid
		if(patch==PATCH_WEST&&Nx>1u) under = solid(1u, y, z);
		else if(patch==PATCH_EAST&&Nx>1u) under = solid(Nx-2u, y, z);
		else if(patch==PATCH_SOUTH&&Ny>1u) under = solid(x, 1u, z);
		else if(patch==PATCH_NORTH&&Ny>1u) under = solid(x, Ny-2u, z);
		if((vox[n]&0x01u)||under) { L.flags[n] = 0x01u; L.set_u(n, zero); grounded.fetch_add(1ull, std::memory_order_relaxed); return; }
		const PatchField2D& f = fields[(size_t)patch];
		if(!f.has_samples()) { missing.fetch_add(1ull, std::memory_order_relaxed); return; }
		V3 p = L.position(x, y, z);
		const bool side = patch==PATCH_WEST||patch==PATCH_EAST||patch==PATCH_SOUTH||patch==PATCH_NORTH;
		if(side&&side_ref_z_cap>=0&&(int)z>side_ref_z_cap) p.z = L.position(x, y, (uint32_t)side_ref_z_cap).z;
		float a, b;
		if(!patch_plane_coords(patch, p, a, b)) { missing.fetch_add(1ull, std::memory_order_relaxed); return; }
		if(side&&f.below_sample_support(a, b)) { L.flags[n] = 0x01u; L.set_u(n, zero); below.fetch_add(1ull, std::memory_order_relaxed); return; }
		L.flags[n] = 0x02u;
		if(downstream_open_face&&patch==downstream_patch) { outlet.fetch_add(1ull, std::memory_order_relaxed); return; }
		L.set_u(n, f.eval(a, b));
		mapped.fetch_add(1ull, std::memory_order_relaxed);
	});
	out.mapped = mapped.load(); out.missing = missing.load(); out.outlet = outlet.load(); out.grounded = grounded.load(); out.below_support = below.load();
	return out;
}

// nearest sample in 3-D, first index wins ties (NearestNeighborInterpolator, FX/interpolation.cpp:52-61)
struct SampleCloud { std::vector<V3> P, U; };
inline V3 nearest_sample_velocity(const SampleCloud& c, const V3& pos) {
	float best = FLT_MAX; V3 u{};
	for(size_t i=0u; i<c.P.size(); i++) {
		const float dx = pos.x-c.P[i].x, dy = pos.y-c.P[i].y, dz = pos.z-c.P[i].z;
		const float d2 = dx*dx+dy*dy+dz*dz;
		if(d2<best) { best = d2; u = c.U[i]; }
	}
	return u;
}

// KNN-HD: the 64 samples of the nearest outer plane closest to `pos` in that plane, Gaussian weights (sigma^2 = R^2/4 with R
// the largest kept distance), local quadratic least squares; falls back to the weighted mean (FX/interpolation_hd.cpp:184-411)
class KnnSurfaceInterpolator {
	const SampleCloud& c_;
	float xmin_ = 0, xmax_ = 0, ymin_ = 0, ymax_ = 0, zmin_ = 0, zmax_ = 0, plane_tol_ = 0;
	std::vector<int> on_plane_[5];
	// Gaussian elimination, partial pivoting, 3 right-hand sides
	static bool solve6(double A[6][6], double bx[6], double by[6], double bz[6], double ax[6], double ay[6], double az[6]) {
		for(int k=0; k<6; k++) {
			int piv = k; double big = std::fabs(A[k][k]);
			for(int i=k+1; i<6; i++) { const double v = std::fabs(A[i][k]); if(v>big) { big = v; piv = i; } }
			if(big<1e-18) return false;
			if(piv!=k) {
				for(int j=0; j<6; j++) std::swap(A[k][j], A[piv][j]);
				std::swap(bx[k], bx[piv]);
				std::swap(by[k], by[piv]);
				std::swap(bz[k], bz[piv]);
			}
			const double inv = 1.0/A[k][k];
			for(int i=k+1; i<6; i++) {
				const double f = A[i][k]*inv;
				if(f==0.0) continue;
				for(int j=k; j<6; j++) A[i][j] -= f*A[k][j];
				bx[i] -= f*bx[k]; by[i] -= f*by[k]; bz[i] -= f*bz[k];
			}
		}
		for(int i=5; i>=0; i--) {
			double sx = bx[i], sy = by[i], sz = bz[i];
			for(int j=i+1; j<6; j++) { sx -= A[i][j]*ax[j]; sy -= A[i][j]*ay[j]; sz -= A[i][j]*az[j]; }
			if(std::fabs(A[i][i])<1e-18) return false;
			const double inv = 1.0/A[i][i];
			ax[i] = sx*inv; ay[i] = sy*inv; az[i] = sz*inv;
		}
		return true;
	}
	void local(const int plane, const V3& p, const V3& pos, float& s1, float& s2) const {
		if(plane<=1) { s1 = p.y-pos.y; s2 = p.z-pos.z; } else if(plane<=3) { s1 = p.x-pos.x; s2 = p.z-pos.z; } else { s1 = p.x-pos.x; s2 = p.y-pos.y; }
	}
public:
	explicit KnnSurfaceInterpolator(const SampleCloud& c) : c_(c) { // the bounds are the same for every query (the reference recomputes them per call)
		if(c.P.empty()) return;
		xmin_ = xmax_ = c.P[0].x; ymin_ = ymax_ = c.P[0].y; zmin_ = zmax_ = c.P[0].z;
		for(const V3& p : c.P) {
			if(p.x<xmin_) xmin_ = p.x;
			if(p.x>xmax_) xmax_ = p.x;
			if(p.y<ymin_) ymin_ = p.y;
			if(p.y>ymax_) ymax_ = p.y;
			if(p.z<zmin_) zmin_ = p.z;
			if(p.z>zmax_) zmax_ = p.z;
		}
		float ext = xmax_-xmin_; if(ymax_-ymin_>ext) ext = ymax_-ymin_; if(zmax_-zmin_>ext) ext = zmax_-zmin_;
		plane_tol_ = 1e-5f*ext+1e-6f;
		// samples of each outer plane, in file order (what the per-query plane filter of the reference keeps; built once)
		for(int i=0; i<(int)c.P.size(); i++) {
			const V3& p = c.P[i];
			const float off[5] = {p.x-xmin_, p.x-xmax_, p.y-ymin_, p.y-ymax_, p.z-zmax_};
			for(int k=0; k<5; k++) if(std::fabs(off[k])<=plane_tol_) on_plane_[k].push_back(i);
		}
	}
	V3 eval(const V3& pos) const {
		constexpr int K = 64; constexpr float eps2 = 1e-16f;
		V3 zero{};
		const int Pn = (int)c_.P.size();
		if(Pn==0) return zero;
		const float d[5] = {std::fabs(pos.x-xmin_), std::fabs(pos.x-xmax_), std::fabs(pos.y-ymin_), std::fabs(pos.y-ymax_), std::fabs(pos.z-zmax_)};
		int plane = 0; float dmin = d[0];
		for(int k=1; k<5; k++) if(d[k]<dmin) { dmin = d[k]; plane = k; }
		float best_r2[K]; int best_i[K]; int filled = 0; float kept_max = 0.0f;
		for(const int i : on_plane_[plane]) {
			const V3& p = c_.P[i];
			float s1, s2; local(plane, p, pos, s1, s2);
			const float r2 = s1*s1+s2*s2;
			if(r2<=eps2) return c_.U[i]; // sample at the query point
			if(filled<K) { best_r2[filled] = r2; best_i[filled] = i; if(r2>kept_max) kept_max = r2; filled++; }
			else {
				int wk = 0; float wr = best_r2[0];
				for(int k=1; k<K; k++) if(best_r2[k]>wr) { wr = best_r2[k]; wk = k; }
				if(r2<wr) { best_r2[wk] = r2; best_i[wk] = i; kept_max = best_r2[0]; for(int k=1; k<K; k++) if(best_r2[k]>kept_max) kept_max = best_r2[k]; }
			}
		}
		if(filled==0) return zero;
		const double sigma2 = 0.25*(double)std::max(kept_max, 1e-12f);
		auto weight = [&](const int idx, double& q1, double& q2) {
			float s1, s2;
			local(plane, c_.P[idx], pos, s1, s2);
			q1 = (double)s1;
			q2 = (double)s2;
			return std::exp(-(q1*q1+q2*q2)/(2.0*sigma2));
		};
		if(filled>=6) {
			double A[6][6] = {}, bx[6] = {}, by[6] = {}, bz[6] = {};
			for(int k=0; k<filled; k++) {
				double q1, q2; const double w = weight(best_i[k], q1, q2);
				const double phi[6] = {1.0, q1, q2, q1*q1, q1*q2, q2*q2};
				for(int i=0; i<6; i++) { const double wi = w*phi[i]; for(int j=0; j<6; j++) A[i][j] += wi*phi[j]; }
				const V3& u = c_.U[best_i[k]];
				for(int i=0; i<6; i++) { const double wp = w*phi[i]; bx[i] += wp*(double)u.x; by[i] += wp*(double)u.y; bz[i] += wp*(double)u.z; }
			}
			double ax[6] = {}, ay[6] = {}, az[6] = {};
			if(solve6(A, bx, by, bz, ax, ay, az)) { V3 r; r.x = (float)ax[0]; r.y = (float)ay[0]; r.z = (float)az[0]; return r; }
		}
		double wx = 0.0, wy = 0.0, wz = 0.0, ws = 0.0;
		for(int k=0; k<filled; k++) {
			double q1, q2;
			const double w = weight(best_i[k], q1, q2);
			const V3& u = c_.U[best_i[k]];
			wx += w*(double)u.x;
			wy += w*(double)u.y;
			wz += w*(double)u.z;
			ws += w;
		}
		if(ws<=0.0) return zero;
		const double inv = 1.0/ws;
		V3 r; r.x = (float)(wx*inv); r.y = (float)(wy*inv); r.z = (float)(wz*inv); return r;
	}
};

// Fill for the sample-cloud interpolators (FX/interpolation.cpp:71-209 and FX/interpolation_hd.cpp:437-745): z = 0 plane solid,
// outer faces TYPE_E with u = inlet(position), downstream face left without velocity when it is open.  The two reference
// variants differ in one detail that is kept: the nearest-sample version leaves non-inlet cells of an open downstream face
// TYPE_E too (same result).  `inlet` must be thread-safe.
inline uint64_t apply_cloud_boundaries(HostLattice& L, const std::string& downstream_bc, const bool downstream_open_face, const int side_ref_z_cap,
	const std::function<V3(const V3&)>& inlet) {
	const uint32_t Nx = L.Nx, Ny = L.Ny, Nz = L.Nz; const uint64_t N = L.N();
	std::vector<uint64_t> cells;
	for(uint64_t n=0ull; n<N; n++) {
		uint32_t x, y, z; L.coords(n, x, y, z);
		if(z==0u) { L.flags[n] = 0x01u; continue; }
		if(!(x==0u||x==Nx-1u||y==0u||y==Ny-1u||z==Nz-1u)) continue;
		L.flags[n] = 0x02u;
		if(!(downstream_open_face&&is_downstream_cell(x, y, Nx, Ny, downstream_bc))) cells.push_back(n);
	}
	std::atomic<uint64_t> next{0ull};
	const unsigned T = bc_worker_threads();
	std::vector<std::thread> th;
	for(unsigned t=0u; t<T; t++) th.emplace_back([&]() { // dynamic chunks: the cost per cell is uneven (plane filter)
		for(;;) {
			const uint64_t s = next.fetch_add(256ull, std::memory_order_relaxed);
			if(s>=cells.size()) break;
			for(uint64_t i=s; i<std::min<uint64_t>(s+256ull, cells.size()); i++) {
				const uint64_t n = cells[i];
				uint32_t x, y, z; L.coords(n, x, y, z);
				V3 p = L.position(x, y, z);
				const bool side = x==0u||x==Nx-1u||y==0u||y==Ny-1u;
				if(side_ref_z_cap>=0&&side&&z!=Nz-1u&&(int)z>side_ref_z_cap) p.z = L.position(x, y, (uint32_t)side_ref_z_cap).z;
				L.set_u(n, inlet(p));
			}
		}
	});
	for(auto& x : th) x.join();
	return (uint64_t)cells.size();
}

// ---- temperature boundaries (use_temperature_bc: buoyancy on and a T column in the CSV), lattice units, T = 1 at the reference
struct TemperatureCounts { uint64_t mapped = 0ull, missing = 0ull, ground_cells = 0ull, ground_columns = 0ull; };
// patch-driven: every non-solid outer cell (z > 0) of a patch with samples gets T = field(a, b), clamped to the CSV's range,
// and the TYPE_T bit; an open downstream face is left alone (FX/setup.cpp:5268-5311)
inline void apply_patch_temperature(HostLattice& L, float* T, const std::vector<PatchField2D>& tfields, const std::string& downstream_bc,
	const bool downstream_open_face, const float Tmin, const float Tmax, TemperatureCounts& cnt) {
	const int dp = downstream_to_patch(downstream_bc);
	std::atomic<uint64_t> mapped{0ull}, missing{0ull};
	bc_parallel_for(L.N(), [&](const uint64_t n) {
		uint32_t x, y, z; L.coords(n, x, y, z);
		if(z==0u) return;
		const int patch = boundary_cell_to_patch(x, y, z, L.Nx, L.Ny, L.Nz);
		if(patch<0||(L.flags[n]&0x01u)||(downstream_open_face&&patch==dp)) return;
		const PatchField2D& f = tfields[(size_t)patch];
		float a, b;
		if(!f.has_samples()||!patch_plane_coords(patch, L.position(x, y, z), a, b)) { missing.fetch_add(1ull, std::memory_order_relaxed); return; }
		T[n] = fminf(fmaxf(f.eval(a, b).x, Tmin), Tmax);
		L.flags[n] = (uint8_t)(L.flags[n]|0x04u);
		mapped.fetch_add(1ull, std::memory_order_relaxed);
	});
	cnt.mapped = mapped.load(); cnt.missing = missing.load();
}
// sample-cloud variants (FX/setup.cpp:5360-5520 high order, :5561-5590 low order): T = 1 below the base height, else the
// interpolator's value; high order marks every outer cell (solid ones too), low order the inlet faces
inline void apply_cloud_temperature(HostLattice& L, float* T, const std::string& downstream_bc, const bool downstream_open_face, const bool high_order,
	const float z_threshold, const float Tmin, const float Tmax,
		const std::function<float(const V3&)>& interp, TemperatureCounts& cnt) {
	const int dp = downstream_to_patch(downstream_bc);
	std::vector<uint64_t> cells;
	for(uint64_t n=0ull; n<L.N(); n++) {
		uint32_t x, y, z; L.coords(n, x, y, z);
		if(z==0u||!(x==0u||x==L.Nx-1u||y==0u||y==L.Ny-1u||z==L.Nz-1u)) continue;
		if(downstream_open_face) {
			// face priority x, y, z
			if(high_order) {
				const int face_patch = x==0u ? PATCH_WEST : x==L.Nx-1u ? PATCH_EAST : y==0u ? PATCH_SOUTH : y==L.Ny-1u ? PATCH_NORTH : PATCH_TOP;
				if(face_patch==dp) continue;
			}
			else if(is_downstream_cell(x, y, L.Nx, L.Ny, downstream_bc)) continue;
		}
		cells.push_back(n);
	}
	bc_parallel_for((uint64_t)cells.size(), [&](const uint64_t i) {
		const uint64_t n = cells[i];
		uint32_t x, y, z; L.coords(n, x, y, z);
		const V3 p = L.position(x, y, z);
		T[n] = fminf(fmaxf(p.z<z_threshold ? 1.0f : interp(p), Tmin), Tmax);
		L.flags[n] = (uint8_t)(L.flags[n]|0x04u);
	});
	cnt.mapped = (uint64_t)cells.size();
}
// solid columns take the temperature of the bottom patch (GroundTemperaturePlane2D on patch 0, FX/setup.cpp:5034-5072)
inline void apply_ground_temperature(HostLattice& L, float* T, const GroundPlane2D& plane, const float Tmin, const float Tmax, TemperatureCounts& cnt) {
	std::vector<float> col((size_t)L.Nx*L.Ny);
	bc_parallel_for((uint64_t)col.size(), [&](const uint64_t id) {
		const V3 p = L.position((uint32_t)(id%L.Nx), (uint32_t)(id/L.Nx), 0u);
		col[id] = fminf(fmaxf(plane.eval(p.x, p.y), Tmin), Tmax);
	});
	std::vector<uint8_t> used(col.size(), 0u);
	uint64_t cells = 0ull;
	for(uint64_t n=0ull; n<L.N(); n++) {
		if(!(L.flags[n]&0x01u)) continue;
		const uint64_t id = n%((uint64_t)L.Nx*L.Ny);
		T[n] = col[id]; L.flags[n] = (uint8_t)(L.flags[n]|0x01u|0x04u); used[id] = 1u; cells++;
	}
	cnt.ground_cells = cells; cnt.ground_columns = 0ull; for(const uint8_t v : used) cnt.ground_columns += v;
}
struct TemperatureSummary {
	uint64_t total = 0ull, solid = 0ull, fluid = 0ull, invalid = 0ull;
	float smin = +FLT_MAX, smax = -FLT_MAX, fmin = +FLT_MAX, fmax = -FLT_MAX;
};
inline TemperatureSummary summarize_temperature(const HostLattice& L, const float* T) { // FX/setup.cpp:5075-5117
	TemperatureSummary r;
	for(uint64_t n=0ull; n<L.N(); n++) {
		if(!(L.flags[n]&0x04u)) continue;
		r.total++;
		if(!std::isfinite(T[n])) { r.invalid++; continue; }
		if(L.flags[n]&0x01u) { r.solid++; r.smin = fminf(r.smin, T[n]); r.smax = fmaxf(r.smax, T[n]); }
		else { r.fluid++; r.fmin = fminf(r.fmin, T[n]); r.fmax = fmaxf(r.fmax, T[n]); }
	}
	return r;
}

struct FluxReport {
	double S_in = 0.0, S_out = 0.0, net_before = 0.0, net_after = 0.0, delta = 0.0, avg_delta = 0.0;
	uint64_t corrected = 0ull;
	double face_avg[5] = {0, 0, 0, 0, 0};
	/* Xn, Xp, Yn, Yp, Zp */
};

// Uniform shift of the outward-normal velocity on all non-solid outer-face cells so that the net boundary flux vanishes;
// cells of the downstream face first receive `downstream_fill(x, y, z)` when given (FX/fluxcorrection.cpp:28-194).
inline FluxReport apply_flux_correction(HostLattice& L, const std::string& downstream_bc,
	const std::function<V3(uint32_t, uint32_t, uint32_t)>& downstream_fill) {
	FluxReport r;
	const uint32_t Nx = L.Nx, Ny = L.Ny, Nz = L.Nz; const uint64_t N = L.N();
	enum Face : int { XN = 0, XP = 1, YN = 2, YP = 3, ZP = 4 };
	struct Cell { uint64_t n; int face; };
	std::vector<Cell> cells;
	for(uint64_t n=0ull; n<N; n++) { // ascending n, like the reference's thread-ordered concatenation
		uint32_t x, y, z; L.coords(n, x, y, z);
		if(z==0u) continue;
		const int face = z==Nz-1u ? ZP : x==0u ? XN : x==Nx-1u ? XP : y==0u ? YN : y==Ny-1u ? YP : -1;
		if(face<0) continue;
		if(L.flags[n]&0x01u) continue;
		L.flags[n] = (uint8_t)(L.flags[n]|0x02u);
		cells.push_back(Cell{n, face});
		if(downstream_fill&&is_downstream_cell(x, y, Nx, Ny, downstream_bc)) L.set_u(n, downstream_fill(x, y, z));
	}
	float* ux = L.u; float* uy = L.u+N; float* uz = L.u+2ull*N;
	auto normal = [&](const Cell& c) -> float {
		switch(c.face) { case ZP: return uz[c.n]; case XN: return -ux[c.n]; case XP: return ux[c.n]; case YN: return -uy[c.n]; default: return uy[c.n]; }
	};
	double net = 0.0;
	for(const Cell& c : cells) { const float vn = normal(c); net += (double)vn; if(vn<0.0f) r.S_in += (double)(-vn); else r.S_out += (double)vn; }
	r.net_before = net; r.corrected = (uint64_t)cells.size();
	r.delta = cells.empty() ? 0.0 : -net/(double)cells.size();
	double sum_abs = 0.0, fsum[5] = {0, 0, 0, 0, 0}; uint64_t fcnt[5] = {0, 0, 0, 0, 0};
	for(const Cell& c : cells) {
		float* comp = c.face==ZP ? &uz[c.n] : (c.face==XN||c.face==XP) ? &ux[c.n] : &uy[c.n];
		const float sgn = (c.face==XN||c.face==YN) ? -1.0f : 1.0f;
		const float before = *comp;
		*comp = before+sgn*(float)r.delta;
		const float d = *comp-before;
		const double mag = std::sqrt((double)d*d);
		sum_abs += mag; fsum[c.face] += mag; fcnt[c.face]++;
	}
	for(const Cell& c : cells) r.net_after += (double)normal(c);
	r.avg_delta = cells.empty() ? 0.0 : sum_abs/(double)cells.size();
	for(int f=0; f<5; f++) r.face_avg[f] = fcnt[f] ? fsum[f]/(double)fcnt[f] : 0.0;
	return r;
}

} // namespace luw_host
