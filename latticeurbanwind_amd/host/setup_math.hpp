// setup_math.hpp -- Units (FX/units.hpp), the deck's Config with the reference's defaults, GPU-memory sizing of the grid, profile.dat reader and interpolation
// Part of the deck driver (luw_driver.cpp); included by it only, after lbm.hpp (namespace luw_host, std::string as string).
#pragma once

// ------------------------------------------------------------------------------------------------ units (FX/units.hpp)
struct Units {
	float unit_m = 1.0f, unit_kg = 1.0f, unit_s = 1.0f, unit_K = 1.0f, unit_K_offset = 0.0f; // T_SI = T*unit_K + unit_K_offset
	void set_m_kg_s_K(const float x, const float u, const float rho, const float T, const float si_x, const float si_u, const float si_rho, const float si_T) {
		unit_m = si_x/x; unit_kg = si_rho/rho*(unit_m*unit_m*unit_m); unit_s = u/si_u*unit_m; unit_K = si_T/T; unit_K_offset = 0.0f;
	}
	void set_temperature_reference(const float T_ref, const float si_T_ref) { unit_K_offset = si_T_ref-T_ref*unit_K; } // FX/units.hpp:37-39
	float T(const float si_T) const { return (si_T-unit_K_offset)/unit_K; }
	float si_T(const float T) const { return T*unit_K+unit_K_offset; }
	float si_dT(const float dT) const { return dT*unit_K; }
	float alpha(const float si_alpha) const { return si_alpha*unit_s/(unit_m*unit_m); }
	float beta(const float si_beta) const { return si_beta*unit_K; }
	float x(const float si_x) const { return si_x/unit_m; }
	float si_x(const float x) const { return x*unit_m; }
	float nu(const float si_nu) const { return si_nu*unit_s/(unit_m*unit_m); }
	float si_u(const float u) const { return u*unit_m/unit_s; }
	float si_rho(const float rho) const { return rho*unit_kg/(unit_m*unit_m*unit_m); }
	ulong t(const float si_t) const { return (ulong)std::fmax(si_t/unit_s+0.5f, 0.5f); }
};

// ------------------------------------------------------------------------------------------------ configuration (defaults FX/setup.cpp:183-220)
struct Config {
	string caseName = "example", datetime = "20990101120000", parent, deck_path;
	bool profile_mode = false, dataset_mode = false;
	float z_si_offset = 50.0f;
	bool downstream_open_face = false;
	uint memory = 20000u; float cell_m = 20.0f;
	float si_x = 0.0f, si_y = 0.0f, si_z = 0.0f;
	uint Dx = 1u, Dy = 1u, Dz = 1u;
	uint research_output_steps = 0u, unsteady_output_interval = 0u, purge_avg_steps = 0u, purge_avg_stride = 1u;
	ulong run_nstep_override = 0ull;
	bool out_tke = true, out_ti = true, out_tls = true;
	bool enable_coriolis = false; float cut_lon[2] = {0, 0}, cut_lat[2] = {0, 0}; bool has_cut_lon = false, has_cut_lat = false;
	bool enable_buffer_nudging = true; float buffer_thickness_m = 160.0f, buffer_tau_s = 300.0f; int buffer_nudge_vertical = 0;
	bool enable_top_sponge = true; float sponge_thickness_m = 200.0f, sponge_tau_s = 120.0f; int sponge_ref_mode = 0;
	bool vk_enable = true; int vk_nmodes = 256; float vk_ti = 0.05f, vk_sigma_si = 0.0f, vk_L_si = 100.0f; uint64_t vk_seed = 100ull; int vk_stride = 1;
	VkUcMode vk_uc = VkUcMode::NORM_MEAN;
	bool vk_same = true, vk_interp = false, vk_inflow_only = false;
	VkFaceMode vk_face_mode = VkFaceMode::AUTO_SIDES;
	float vk_aniso[3] = {1.0f, 1.0f, 1.0f};
	std::vector<float> inflow_list, angle_list;
	// command line
	string probes_raw, utm_crs; bool probes_output_defined = false; uint probes_output_steps = 0u; bool has_rotate_deg = false; double rotate_deg = 0.0;
	bool buoyancy = true, buoyancy_explicit = false; // default-on unless explicitly false (FX/setup.cpp:2743)
	// *.luw
	bool nwp_mode = false; string downstream_bc = "+y", downstream_bc_yaw, validation = "unknown"; bool use_high_order = false, flux_correction = false;
	bool fp16c = true, native_arith = true; int device = 0; bool dry_run = false, sizing_only = false; string dump_setup, dump_vk;
	std::vector<int> devices; uint32_t kernel = LUW_KERNEL_AUTO;
};

// memory model of the SHIPPED reference build (D3Q19 FP16C + FORCE_FIELD + TEMPERATURE + GRAPHICS), FX/lbm.cpp:188-228:
// gpu_memory decks were sized against it, so the same deck must give the same grid here
static uint vram_required_mb_per_device(const uint Nx, const uint Ny, const uint Nz, const uint Dx, const uint Dy, const uint Dz) {
	const uint Hx = Dx>1u, Hy = Dy>1u, Hz = Dz>1u;
	const ulong lx = (ulong)(Nx/Dx+2u*Hx), ly = (ulong)(Ny/Dy+2u*Hy), lz = (ulong)(Nz/Dz+2u*Hz), N = lx*ly*lz;
	auto mb = [](const ulong bytes) { return (uint)(bytes/1048576ull); };
	uint m = 0u;
	m += mb(N*19ull*2ull); m += mb(N*4ull); m += mb(N*12ull); m += mb(N);           // fi, rho, u, flags
	m += mb(N*12ull); m += mb(16ull);                                              // F, object_sum
	m += mb(N*7ull*2ull); m += mb(N*4ull);                                          // gi, T
	const ulong pixels = 1920ull*1080ull; m += mb(pixels*4ull); m += mb(pixels*4ull); m += mb(60ull); // bitmap, zbuffer, camera
	if(Dx*Dy*Dz>1u) {
		ulong Amax = 0ull;
		if(Dx>1u) Amax = std::max(Amax, ly*lz); if(Dy>1u) Amax = std::max(Amax, lz*lx); if(Dz>1u) Amax = std::max(Amax, lx*ly);
		m += 2u*mb(Amax*(ulong)std::max(5u*2u, 17u));
	}
	return m;
}
static uint vk_extra_mb(const Config& c, const uint Nx, const uint Ny, const uint Nz) { // FX/setup.cpp:312-333
	if(!c.vk_enable||Nx<2u||Ny<2u||Nz<2u) return 0u;
	const ulong nz_side = Nz>2u ? (ulong)(Nz-2u) : 0ull, nx_inner = Nx>2u ? (ulong)(Nx-2u) : 0ull;
	const ulong pts = 2ull*(ulong)Ny*nz_side+2ull*nx_inner*nz_side+(ulong)Nx*(ulong)Ny;
	const ulong mode_stride = 5ull*(ulong)std::max(1, c.vk_nmodes);
	auto mb = [](const ulong bytes) { return (uint)(bytes/1048576ull); };
	return mb(pts*8ull)+mb(pts)+mb(pts*28ull)+mb(mode_stride*40ull);
}
struct GridEstimate { uint Nx, Ny, Nz, core_mb, extra_mb, total_mb; };
static GridEstimate estimate_from_cell_size(const Config& c, const float cell) { // FX/setup.cpp:345-369
	const float safe = std::fmax(cell, 1.0e-6f);
	GridEstimate e{};
	e.Nx = (uint)std::max(1, (int)(c.si_x/safe+0.5f)); e.Ny = (uint)std::max(1, (int)(c.si_y/safe+0.5f));
	const uint core = (uint)std::max(1, (int)(c.si_z/safe+0.5f));
	const bool ext = c.enable_top_sponge&&c.sponge_tau_s>0.0f&&c.sponge_ref_mode==0&&core>2u;
	e.Nz = core+(ext ? (uint)std::max(1, (int)std::lround(c.sponge_thickness_m/safe)) : 0u);
	e.core_mb = vram_required_mb_per_device(e.Nx, e.Ny, e.Nz, c.Dx, c.Dy, c.Dz);
	e.extra_mb = vk_extra_mb(c, e.Nx, e.Ny, e.Nz);
	e.total_mb = e.core_mb+e.extra_mb;
	return e;
}
// mesh_control = "gpu_memory": the finest cell size whose grid still fits the requested MB per device (FX/setup.cpp:371-407).
// Memory falls monotonically with the cell size, so this is a bracket-and-bisect search in FP32: the coarse end starts at one
// cell per domain edge and doubles until it fits, the fine end halves until it no longer does, then 48 halvings of the bracket.
// The grid that comes out (751x742x174 for the reference's example deck) depends on every float of this sequence.
static float fit_cell_size_to_gpu_memory_request(const Config& c, const uint target_mb) {
	if(target_mb==0u) return 20.0f;
	auto fits = [&](const float cell) { return estimate_from_cell_size(c, cell).total_mb<=target_mb; };
	float coarse = std::fmax(std::fmax(std::fmax(c.si_x, c.si_y), c.si_z+std::fmax(c.sponge_thickness_m, 0.0f)), 1.0f); // feasible end of the bracket
	for(int tries=32; tries>0&&!fits(coarse); tries--) coarse *= 2.0f;
	float fine = coarse*0.5f;                                                                                          // infeasible end
	for(int tries=64; tries>0&&fine>1.0e-6f&&fits(fine); tries--) { coarse = fine; fine *= 0.5f; }
	for(int halvings=48; halvings>0; halvings--) {
		const float mid = 0.5f*(fine+coarse);
		(fits(mid) ? coarse : fine) = mid;
	}
	return coarse;
}

// ------------------------------------------------------------------------------------------------ profile (FX/setup.cpp:2122-2150,2243-2280)
static std::vector<std::pair<float, float>> read_profile_dat(const string& path) {
	std::vector<std::pair<float, float>> out;
	std::ifstream fin(path);
	if(!fin.is_open()) { println("ERROR: could not open profile file "+path); return out; }
	string line;
	while(std::getline(fin, line)) {
		size_t c = line.find("//"); if(c!=string::npos) line.erase(c);
		c = line.find('#'); if(c!=string::npos) line.erase(c);
		line = Deck::strip(line);
		if(line.empty()) continue;
		for(char& ch : line) if(ch==','||ch==';') ch = ' ';
		std::stringstream ss(line);
		float z = 0.0f, u = 0.0f;
		if(!(ss >> z >> u)) continue;
		if(!std::isfinite(z)||!std::isfinite(u)) continue;
		out.push_back({z, u});
	}
	return out;
}
// U(z) between the samples of profile.dat: a cubic Hermite segment through the two neighbouring samples with secant slopes
// (one-sided at the ends of the table, centred inside), constant outside the table (FX/setup.cpp:2243-2280 with the basis of
// FX/utilities.hpp:2374-2377; FP32, the order of the operations below is the reference's).
static float profile_secant(const std::vector<float>& z, const std::vector<float>& u, const size_t i) {
	const size_t n = z.size(), lo = i==0u ? 0u : (i+1u>=n ? n-2u : i-1u), hi = i==0u ? 1u : (i+1u>=n ? n-1u : i+1u);
	const float dz = z[hi]-z[lo];
	return dz!=0.0f ? (u[hi]-u[lo])/dz : 0.0f;
}
static float cubic_hermite(const float y0, const float y1, const float d0, const float d1, const float t) {
	const float t2 = t*t, t3 = t*t*t;
	return (2.0f*t3-3.0f*t2+1.0f)*y0+(-2.0f*t3+3.0f*t2)*y1+(t3-2.0f*t2+t)*d0+(t3-t2)*d1;
}
static float interpolate_profile_cubic(const std::vector<float>& z, const std::vector<float>& u, const float zq) {
	if(z.empty()) return 0.0f;
	if(z.size()==1u||zq<=z.front()) return u.front();
	if(zq>=z.back()) return u.back();
	size_t seg = 0u; // last sample at or below zq (z ascending)
	for(size_t lo = 0u, hi = z.size()-1u; lo<hi; ) { const size_t mid = (lo+hi+1u)/2u; if(z[mid]<=zq) { lo = mid; seg = mid; } else hi = mid-1u; }
	const size_t nxt = std::min(seg+1u, z.size()-1u);
	const float h = z[nxt]-z[seg];
	if(h<=0.0f) return u[seg];
	return cubic_hermite(u[seg], u[nxt], profile_secant(z, u, seg)*h, profile_secant(z, u, nxt)*h, (zq-z[seg])/h);
}

