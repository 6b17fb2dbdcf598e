// vk_inlet.hpp -- host side of the von-Karman synthetic-turbulence inlet: what the reference's
// VonKarmanInletUpdater does on the host (FX/setup.cpp:413-1149) -- select inlet cells on the west/east/south/north/top
// faces, build the random Fourier modes of a von-Karman spectrum, pack the SoA tables -- for ONE domain.  The per-step
// evaluation is the device kernel behind luw_vk_inlet_attach (FX/kernel.cpp:2495-2571).
// Random numbers: std::mt19937_64 + std::uniform_real_distribution<float>, exactly the calls of FX/setup.cpp:794-830
// (their sequence is defined by the C++ standard library in use; the Linux reference build and this code share libstdc++).
#pragma once
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <random>
#include <string>
#include <vector>

namespace luw_host {

enum class VkUcMode { NORM_MEAN = 0, NORMAL_COMPONENT = 1 };
enum class VkFaceMode { AUTO_SIDES = 0, TARGET_INFLOW = 1, EXCLUDE_DOWNSTREAM = 2, EXCLUDE_DOWNSTREAM_SIDES = 3, ALL_SIDES = 4, ALL_SELECTED = 5 };

struct VkRuntimeConfig { // VkInletRuntimeConfig, FX/setup.cpp:281-296
	bool enable = true;
	float ti = 0.05f, sigma_lbm = 0.0f, L_lbm = 100.0f;
	int nmodes = 256;
	uint64_t seed = 100ull;
	int update_stride = 1;
	VkUcMode uc_mode = VkUcMode::NORM_MEAN;
	bool same_realization_all_faces = true, stride_interpolation = false, inflow_only = false;
	VkFaceMode face_mode = VkFaceMode::AUTO_SIDES;
	float aniso[3] = {1.0f, 1.0f, 1.0f};
	int downstream_face_id = -1; // 0 west, 1 east, 2 south, 3 north (FX/setup.cpp:3742-3748)
};
inline VkFaceMode vk_resolve_face_mode(const VkFaceMode m, const bool inflow_only) { // FX/setup.cpp:257-260
	if(m!=VkFaceMode::AUTO_SIDES) return m;
	return inflow_only ? VkFaceMode::EXCLUDE_DOWNSTREAM_SIDES : VkFaceMode::ALL_SIDES;
}

struct VkTables { // what luw_vk_inlet_attach consumes
	uint64_t point_count = 0ull, mode_count = 0ull;
	std::vector<uint64_t> point_cell;   // cell index n = x+(y+z*Ny)*Nx
	std::vector<uint8_t> point_face;    // 0 west 1 east 2 south 3 north 4 top
	std::vector<float> point_data;      // SoA [7][P]: px, py, pz, base_u.x, base_u.y, base_u.z, sigma
	std::vector<float> mode_data;       // SoA [10][5*M]: kx, ky, kz, omega, Ax, Ay, Az, phix, phiy, phiz
	std::array<uint64_t, 5> face_points{}; std::array<float, 5> face_uc{};
	float sigma_min = FLT_MAX, sigma_max = 0.0f; double sigma_sum = 0.0;
};

struct VkMode { float kx = 0, ky = 0, kz = 0, omega = 0, Ax = 0, Ay = 0, Az = 0, phix = 0, phiy = 0, phiz = 0; };

// ---- random Fourier modes of a von-Karman spectrum (the reference's build_modes_for_seed_, FX/setup.cpp:777-850).
// Mode m of M sits in the m-th of M equal slices of log k between k_min = 2 pi / (10 L) and k_max = pi (grid cut-off), at a random
// position inside its slice; its direction is uniform on the sphere (cos(polar) uniform in [-1,1], azimuth uniform); its amplitude
// follows E(k) ~ k^4 / (1 + (kL)^2)^(17/6); three independent phases, one per velocity component; the frequency is that of a
// pattern frozen into a flow of speed u_ref along conv_dir.  Amplitudes are normalised to unit variance of the summed signal
// (0.5 sum a^2 = 1) and weighted per component by the anisotropy factors.
// The tables must equal the reference's bit for bit (tests/test_vk_inlet.py), which fixes two things: every random number comes
// from ONE std::mt19937_64 through std::uniform_real_distribution<float>(0, 1) in the order slice position, cos(polar), azimuth,
// phase x, phase y, phase z -- and the FP32 expressions below keep their operand order.
struct VkSpectrumBand { float log_k_lo = 0.0f, log_k_width = 0.0f; bool valid = false; };
inline VkSpectrumBand vk_spectrum_band(const float L) {
	VkSpectrumBand band;
	const float pi = 3.1415927f, k_hi = pi/1.0f; // cut-off: wavelength of two cells
	float k_lo = 2.0f*pi/(10.0f*L);
	if(!(k_lo>0.0f)||!std::isfinite(k_lo)) k_lo = 1.0e-4f;
	if(k_lo>=0.99f*k_hi) k_lo = 0.1f*k_hi;
	band.log_k_lo = logf(k_lo);
	band.log_k_width = fmaxf(logf(k_hi)-band.log_k_lo, 1.0e-6f);
	band.valid = true;
	return band;
}
inline float vk_spectrum_amplitude(const float k, const float L) { // sqrt of k^4 / (1 + (kL)^2)^(17/6)
	const float kL = k*L;
	const float damping = powf(1.0f+kL*kL, 17.0f/6.0f);
	const float energy = damping>0.0f ? powf(k, 4.0f)/damping : 0.0f;
	return sqrtf(fmaxf(energy, 0.0f));
}
inline bool vk_build_modes_for_seed(const VkRuntimeConfig& cfg, const float u_ref, const float conv_dir[3], const uint64_t seed, std::vector<VkMode>& out) {
	out.clear();
	if(!(cfg.L_lbm>0.0f)||cfg.nmodes<=0) return false;
	const float two_pi = 2.0f*3.1415927f;
	const VkSpectrumBand band = vk_spectrum_band(cfg.L_lbm);
	std::mt19937_64 engine((unsigned long long)seed);
	std::uniform_real_distribution<float> unit(0.0f, 1.0f);
	const size_t M = (size_t)cfg.nmodes;
	out.assign(M, VkMode());
	double energy_sum = 0.0; // sum of squared raw amplitudes (held in Ax until the normalisation pass)
	for(size_t m=0u; m<M; m++) {
		VkMode& md = out[m];
		const float slice_pos = ((float)(int)m+unit(engine))/(float)cfg.nmodes;
		const float k = expf(band.log_k_lo+slice_pos*band.log_k_width);
		const float cos_polar = 2.0f*unit(engine)-1.0f;
		const float azimuth = two_pi*unit(engine);
		const float sin_polar = sqrtf(fmaxf(0.0f, 1.0f-cos_polar*cos_polar));
		md.kx = k*(sin_polar*cosf(azimuth)); md.ky = k*(sin_polar*sinf(azimuth)); md.kz = k*cos_polar;
		md.Ax = vk_spectrum_amplitude(k, cfg.L_lbm);
		energy_sum += (double)md.Ax*(double)md.Ax;
		md.omega = u_ref*(md.kx*conv_dir[0]+md.ky*conv_dir[1]+md.kz*conv_dir[2]);
		md.phix = two_pi*unit(engine);
		md.phiy = two_pi*unit(engine);
		md.phiz = two_pi*unit(engine);
	}
	const double signal_variance = 0.5*energy_sum;
	if(!(signal_variance>0.0)) { out.clear(); return false; }
	const float to_unit_variance = 1.0f/(float)sqrt(signal_variance);
	for(VkMode& md : out) { const float a = md.Ax*to_unit_variance; md.Ax = a*cfg.aniso[0]; md.Ay = a*cfg.aniso[1]; md.Az = a*cfg.aniso[2]; }
	return true;
}
// one stream of modes per face: the 64-bit finaliser of MurmurHash3 (fmix64) over seed xor a per-face multiple of the golden-ratio
// constant (FX/setup.cpp:767-775)
inline uint64_t murmur3_fmix64(uint64_t h) { h ^= h>>33; h *= 0xff51afd7ed558ccdull; h ^= h>>33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h>>33; return h; }
inline uint64_t vk_mix_seed(const uint64_t seed, const uint32_t face_id) { return murmur3_fmix64(seed^(0x9E3779B97F4A7C15ull*(uint64_t)(face_id+1u))); }

// initialize(): collect_face_points_ + build_face_modes_ + the table packing of build_gpu_runtime_ for a single domain.
// flags/u are the host fields in the reference layout AFTER the boundary fill (u = base inflow on TYPE_E cells).
// log receives the reference's console lines.  Returns false when the inlet stays inactive.
template<typename LogFn> inline bool vk_build_tables(const VkRuntimeConfig& cfg_in, const uint32_t Nx, const uint32_t Ny, const uint32_t Nz,
	const uint8_t* flags, const float* u, VkTables& T, LogFn log) {
	VkRuntimeConfig cfg = cfg_in;
	T = VkTables();
	if(!cfg.enable) return false;
	if(!(cfg.L_lbm>0.0f)||cfg.nmodes<=0) return false;
	if(Nx<2u||Ny<2u||Nz<2u) { log("| VK inlet        | disabled: grid too small for turbulent inlet faces         |"); return false; }
	const uint64_t N = (uint64_t)Nx*Ny*Nz;
	enum { WEST = 0, EAST = 1, SOUTH = 2, NORTH = 3, TOP = 4 };
	static const float face_n[5][3] = {{+1, 0, 0}, {-1, 0, 0}, {0, +1, 0}, {0, -1, 0}, {0, 0, -1}};
	auto opposite = [](const int id) { return id==WEST ? EAST : id==EAST ? WEST : id==SOUTH ? NORTH : id==NORTH ? SOUTH : -1; };
	const int target_face = opposite(cfg.downstream_face_id);
	struct P { uint64_t n; uint32_t x, y, z; float bu[3]; };
	std::array<std::vector<P>, 5> pts;
	auto valid = [&](const uint64_t n, const int fid, const uint32_t z) { // cell_is_valid_inlet_, FX/setup.cpp:667-689
		if(z==0u) return false;
		if((flags[n]&0x01u)!=0u) return false;
		if((flags[n]&0x02u)==0u) return false;
		if(cfg.face_mode==VkFaceMode::TARGET_INFLOW) {
			if(target_face>=0&&fid!=target_face) return false;
			if(target_face<0&&fid==TOP&&cfg.inflow_only) return false;
		}
		else if(cfg.face_mode==VkFaceMode::EXCLUDE_DOWNSTREAM) { if(cfg.downstream_face_id>=0&&fid==cfg.downstream_face_id) return false; }
		else if(cfg.face_mode==VkFaceMode::EXCLUDE_DOWNSTREAM_SIDES) {
			if(fid==TOP) return false;
			if(cfg.downstream_face_id>=0&&fid==cfg.downstream_face_id) return false;
		}
		else if(cfg.face_mode==VkFaceMode::ALL_SIDES) { if(fid==TOP) return false; }
		else if(fid==TOP&&cfg.inflow_only) return false;
		return true;
	};
	auto add = [&](const int fid, const uint32_t x, const uint32_t y, const uint32_t z) {
		const uint64_t n = (uint64_t)x+((uint64_t)y+(uint64_t)z*Ny)*Nx;
		if(valid(n, fid, z)) pts[(size_t)fid].push_back(P{n, x, y, z, {u[n], u[N+n], u[2ull*N+n]}});
	};
	for(uint32_t z=1u; z+1u<Nz; ++z) { // FX/setup.cpp:713-737
		for(uint32_t y=0u; y<Ny; ++y) { add(WEST, 0u, y, z); add(EAST, Nx-1u, y, z); }
		if(Nx>2u) for(uint32_t x=1u; x+1u<Nx; ++x) { add(SOUTH, x, 0u, z); add(NORTH, x, Ny-1u, z); }
	}
	for(uint32_t y=0u; y<Ny; ++y) for(uint32_t x=0u; x<Nx; ++x) add(TOP, x, y, Nz-1u);
	std::array<bool, 5> enabled{}; std::array<float, 5> uc{};
	static const char* names[5] = {"west", "east", "south", "north", "top"};
	for(int f=0; f<5; f++) { // FX/setup.cpp:750-764
		if(pts[(size_t)f].empty()) continue;
		float mu[3] = {0, 0, 0};
		for(const P& p : pts[(size_t)f]) { mu[0] += p.bu[0]; mu[1] += p.bu[1]; mu[2] += p.bu[2]; }
		const float cnt = (float)pts[(size_t)f].size();
		mu[0] /= cnt; mu[1] /= cnt; mu[2] /= cnt;
		uc[(size_t)f] = cfg.uc_mode==VkUcMode::NORM_MEAN ? sqrtf(mu[0]*mu[0]+mu[1]*mu[1]+mu[2]*mu[2])
			: fabsf(mu[0]*face_n[f][0]+mu[1]*face_n[f][1]+mu[2]*face_n[f][2]);
		if(!(uc[(size_t)f]>1.0e-7f)) { log(std::string("| VK inlet face   | ")+names[f]+": disabled (Uc is too small)               |"); continue; }
		enabled[(size_t)f] = true;
	}
	uint32_t face_enabled = 0u; float mean_all[3] = {0, 0, 0}; double sum_mag = 0.0; uint64_t count_u = 0ull;
	for(int f=0; f<5; f++) { // FX/setup.cpp:461-477
		if(!enabled[(size_t)f]||pts[(size_t)f].empty()) continue;
		face_enabled++;
		for(const P& p : pts[(size_t)f]) {
			mean_all[0] += p.bu[0];
			mean_all[1] += p.bu[1];
			mean_all[2] += p.bu[2];
			sum_mag += (double)sqrtf(p.bu[0]*p.bu[0]+p.bu[1]*p.bu[1]+p.bu[2]*p.bu[2]);
			count_u++;
		}
		log(std::string("| VK inlet face   | ")+names[f]+": points="+std::to_string(pts[(size_t)f].size())+", Uc="+std::to_string(uc[(size_t)f])+" |");
	}
	if(face_enabled==0u) { log("| VK inlet        | enabled in config, but no valid inflow faces found         |"); return false; }
	if(count_u==0ull) return false;
	const float u_ref = (float)(sum_mag/(double)count_u);
	float conv[3] = {mean_all[0]/(float)count_u, mean_all[1]/(float)count_u, mean_all[2]/(float)count_u};
	const float conv_len = sqrtf(conv[0]*conv[0]+conv[1]*conv[1]+conv[2]*conv[2]);
	if(conv_len>1.0e-7f) { conv[0] /= conv_len; conv[1] /= conv_len; conv[2] /= conv_len; } else { conv[0] = 1.0f; conv[1] = 0.0f; conv[2] = 0.0f; }
	std::array<std::vector<VkMode>, 5> face_modes; // build_face_modes_, FX/setup.cpp:852-884
	if(cfg.same_realization_all_faces) {
		std::vector<VkMode> shared;
		if(!vk_build_modes_for_seed(cfg, u_ref, conv, cfg.seed, shared)) {
			log("| VK inlet        | failed to build VK spectrum modes                           |");
			return false;
		}
		for(int f=0; f<5; f++) if(enabled[(size_t)f]&&!pts[(size_t)f].empty()) face_modes[(size_t)f] = shared;
	} else {
		for(int f=0; f<5; f++) {
			if(!enabled[(size_t)f]||pts[(size_t)f].empty()) continue;
			if(!vk_build_modes_for_seed(cfg, u_ref, conv, vk_mix_seed(cfg.seed, (uint32_t)f), face_modes[(size_t)f])) return false;
		}
	}
	const uint64_t M = (uint64_t)cfg.nmodes, V = 5ull*M; // build_gpu_runtime_, FX/setup.cpp:886-1057
	T.mode_count = M;
	T.mode_data.assign((size_t)(10ull*V), 0.0f);
	for(int f=0; f<5; f++) for(uint64_t m=0ull; m<(uint64_t)face_modes[(size_t)f].size(); ++m) {
		const VkMode& md = face_modes[(size_t)f][(size_t)m]; const uint64_t idx = (uint64_t)f*M+m;
		const float vals[10] = {md.kx, md.ky, md.kz, md.omega, md.Ax, md.Ay, md.Az, md.phix, md.phiy, md.phiz};
		for(int q=0; q<10; q++) T.mode_data[(size_t)((uint64_t)q*V+idx)] = vals[q];
	}
	struct DP { uint64_t n; uint8_t f; float px, py, pz, bu[3], sigma; };
	std::vector<DP> dps;
	for(int f=0; f<5; f++) {
		if(!enabled[(size_t)f]||pts[(size_t)f].empty()) continue;
		for(const P& p : pts[(size_t)f]) {
			const float u_char = cfg.uc_mode==VkUcMode::NORM_MEAN ? sqrtf(p.bu[0]*p.bu[0]+p.bu[1]*p.bu[1]+p.bu[2]*p.bu[2])
				: fabsf(p.bu[0]*face_n[f][0]+p.bu[1]*face_n[f][1]+p.bu[2]*face_n[f][2]);
			const float sigma_local = cfg.ti>0.0f ? cfg.ti*u_char : cfg.sigma_lbm;
			if(!(sigma_local>0.0f)) continue;
			dps.push_back(DP{p.n, (uint8_t)f, (float)p.x, (float)p.y, (float)p.z, {p.bu[0], p.bu[1], p.bu[2]}, sigma_local});
			T.sigma_sum += (double)sigma_local; T.sigma_min = fminf(T.sigma_min, sigma_local); T.sigma_max = fmaxf(T.sigma_max, sigma_local);
			T.face_points[(size_t)f]++;
		}
		T.face_uc[(size_t)f] = uc[(size_t)f];
	}
	if(dps.empty()) { log("| VK inlet        | no valid GPU runtime mapping for inflow faces              |"); return false; }
	const uint64_t Pn = dps.size();
	T.point_count = Pn;
	T.point_cell.resize(Pn); T.point_face.resize(Pn); T.point_data.assign((size_t)(7ull*Pn), 0.0f);
	for(uint64_t i=0ull; i<Pn; ++i) {
		const DP& d = dps[(size_t)i];
		T.point_cell[i] = d.n; T.point_face[i] = d.f;
		const float vals[7] = {d.px, d.py, d.pz, d.bu[0], d.bu[1], d.bu[2], d.sigma};
		for(int q=0; q<7; q++) T.point_data[(size_t)((uint64_t)q*Pn+i)] = vals[q];
	}
	log("| VK inlet        | active: points="+std::to_string(Pn)+", per-face modes="+std::to_string(M)+", L_lbm="+std::to_string(cfg.L_lbm)+", TI="
		+std::to_string(cfg.ti)+", stride="+std::to_string(cfg.update_stride)+" |");
	return true;
}

// compute_time_params_, FX/setup.cpp:1118-1140
inline void vk_time_params(const uint64_t t, const int update_stride, const bool stride_interpolation, uint32_t& use_interp, float& t0, float& t1,
	float& alpha) {
	const uint64_t stride = update_stride>1 ? (uint64_t)update_stride : 1ull;
	if(stride<=1ull) { use_interp = 0u; t0 = (float)t; t1 = t0; alpha = 0.0f; return; }
	if(stride_interpolation) {
		const uint64_t anchor = (t/stride)*stride;
		use_interp = 1u;
		t0 = (float)anchor;
		t1 = (float)(anchor+stride);
		alpha = (float)(t-anchor)/(float)stride;
	}
	else { const uint64_t hold_t = (t/stride)*stride; use_interp = 0u; t0 = (float)hold_t; t1 = t0; alpha = 0.0f; }
}

} // namespace luw_host
