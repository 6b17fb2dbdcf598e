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
// Temperature columns are parsed and reported but not used: the thermal lattice is outside this path.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace luw_host {

enum Patch : int { PATCH_BOTTOM = 0, PATCH_TOP = 1, PATCH_SOUTH = 2, PATCH_NORTH = 3, PATCH_WEST = 4, PATCH_EAST = 5 }; // FX/setup.cpp:225-230
inline const char* patch_name(const int p) { static const char* n[6] = {"bottom", "top", "south", "north", "west", "east"}; return p>=0&&p<6 ? n[p] : "unknown"; }
inline int downstream_to_patch(const std::string& bc) { return bc=="+y" ? PATCH_NORTH : bc=="-y" ? PATCH_SOUTH : bc=="+x" ? PATCH_EAST : bc=="-x" ? PATCH_WEST : -1; }
inline int boundary_cell_to_patch(const uint32_t x, const uint32_t y, const uint32_t z, const uint32_t Nx, const uint32_t Ny, const uint32_t Nz) { // top wins over sides, FX/setup.cpp:1816-1823
	if(z==Nz-1u) return PATCH_TOP; if(x==0u) return PATCH_WEST; if(x==Nx-1u) return PATCH_EAST; if(y==0u) return PATCH_SOUTH; if(y==Ny-1u) return PATCH_NORTH; return -1;
}
inline bool is_downstream_cell(const uint32_t x, const uint32_t y, const uint32_t Nx, const uint32_t Ny, const std::string& bc) {
	return bc=="+y" ? y==Ny-1u : bc=="-y" ? y==0u : bc=="+x" ? x==Nx-1u : bc=="-x" ? x==0u : false;
}

struct V3 { float x = 0.0f, y = 0.0f, z = 0.0f; };
struct SurfSample { V3 p, u; float T = 293.15f; int patch = -1; };
struct SurfData {
	std::vector<SurfSample> rows;
	bool has_T = false, has_patch = false; uint64_t rows_T = 0ull, rows_patch = 0ull; float tmin = 293.15f, tmax = 293.15f;
	std::vector<std::string> warnings;
};

inline std::string bc_trim(const std::string& s) { const char* ws = " \t\r\n"; const size_t b = s.find_first_not_of(ws), e = s.find_last_not_of(ws); return b==std::string::npos ? std::string() : s.substr(b, e-b+1u); }

// SurfData_<datetime>.csv: header X,Y,Z,u,v,w[,T][,patch] (any order, case-insensitive) or legacy positional 6..8 columns
inline bool read_surfdata_csv(const std::string& path, SurfData& out) {
	out = SurfData();
	std::ifstream fin(path);
	if(!fin.is_open()) return false;
	auto split = [](const std::string& s) { std::vector<std::string> c; std::stringstream ss(s); std::string t; while(std::getline(ss, t, ',')) c.push_back(bc_trim(t)); return c; };
	auto lower = [](std::string s) { for(char& ch : s) ch = (char)std::tolower((unsigned char)ch); return s; };
	std::string header;
	if(!std::getline(fin, header)) return true; // empty file: no rows
	const std::vector<std::string> hc = split(header);
	auto col = [&](const char* key) { for(size_t i=0u; i<hc.size(); i++) if(lower(hc[i])==key) return (int)i; return -1; };
	const int ix = col("x"), iy = col("y"), iz = col("z"), iu = col("u"), iv = col("v"), iw = col("w"), it = col("t"), ip = col("patch");
	const bool named = ix>=0&&iy>=0&&iz>=0&&iu>=0&&iv>=0&&iw>=0;
	out.has_patch = ip>=0;
	float tmin = +FLT_MAX, tmax = -FLT_MAX;
	std::string line; uint64_t line_no = 1ull;
	while(std::getline(fin, line)) {
		line_no++;
		const std::vector<std::string> c = split(line);
		if(c.empty()) continue;
		SurfSample s;
		if(named) {
			const int need = std::max(std::max(std::max(ix, iy), std::max(iz, iu)), std::max(iv, iw));
			if((int)c.size()<=need) { out.warnings.push_back("WARNING: malformed line "+std::to_string(line_no)+" in CSV (missing required columns)"); continue; }
			s.p.x = (float)atof(c[ix].c_str()); s.p.y = (float)atof(c[iy].c_str()); s.p.z = (float)atof(c[iz].c_str());
			s.u.x = (float)atof(c[iu].c_str()); s.u.y = (float)atof(c[iv].c_str()); s.u.z = (float)atof(c[iw].c_str());
			if(it>=0&&(int)c.size()>it) { s.T = (float)atof(c[it].c_str()); out.has_T = true; out.rows_T++; tmin = std::fmin(tmin, s.T); tmax = std::fmax(tmax, s.T); }
			if(ip>=0&&(int)c.size()>ip) { s.patch = (int)std::lround((double)atof(c[ip].c_str())); out.has_patch = true; out.rows_patch++; }
			out.rows.push_back(s);
			continue;
		}
		float v[8] = {0.0f}; int nc = 0; // legacy positional rows
		{ std::stringstream ss(line); std::string tok; while(std::getline(ss, tok, ',')) { if(nc<8) v[nc] = (float)atof(bc_trim(tok).c_str()); nc++; } }
		if(nc<6||nc>8) { out.warnings.push_back("WARNING: malformed line "+std::to_string(line_no)+" in CSV (expect 6~8 columns)"); continue; }
		s.p.x = v[0]; s.p.y = v[1]; s.p.z = v[2]; s.u.x = v[3]; s.u.y = v[4]; s.u.z = v[5];
		bool row_T = false;
		if(nc>=8) { s.T = v[6]; row_T = true; s.patch = (int)std::lround((double)v[7]); out.has_patch = true; out.rows_patch++; }
		else if(nc==7) { // 7th column is T or patch: integers 0..5 read as patch
			const float q = v[6];
			if(q>=-0.5f&&q<=5.5f&&fabsf(q-roundf(q))<=1e-4f) { s.patch = (int)std::lround((double)q); out.has_patch = true; out.rows_patch++; }
			else { s.T = q; row_T = true; }
		}
		if(row_T) { out.has_T = true; out.rows_T++; tmin = std::fmin(tmin, s.T); tmax = std::fmax(tmax, s.T); }
		out.rows.push_back(s);
	}
	if(out.has_T) { out.tmin = tmin; out.tmax = tmax; }
	return true;
}

inline bool patch_plane_coords(const int patch, const V3& p, float& a, float& b) { // FX/setup.cpp:1837-1860
	switch(patch) {
		case PATCH_BOTTOM: case PATCH_TOP: a = p.x; b = p.y; return true;
		case PATCH_SOUTH: case PATCH_NORTH: a = p.x; b = p.z; return true;
		case PATCH_WEST: case PATCH_EAST: a = p.y; b = p.z; return true;
		default: a = b = 0.0f; return false;
	}
}

// Piecewise-bilinear field over the samples of one boundary patch: samples are grouped into columns of (nearly) equal `a`,
// each column holds its samples sorted by `b` with near-duplicates merged; evaluation interpolates linearly inside the two
// bracketing columns and then between them, clamping outside (PatchSurfaceField2D, FX/setup.cpp:1862-2094).
class PatchField2D {
	size_t raw_count_ = 0u;
	V3 default_{};
	std::vector<float> a_; // column coordinate
	std::vector<uint32_t> start_; // CSR: column c owns [start_[c], start_[c+1]) of b_/v_
	std::vector<float> b_;
	std::vector<V3> v_;
	struct Raw { float a, b; V3 v; };
	static V3 lerp(const V3& p, const V3& q, const float t) { V3 r; r.x = p.x+t*(q.x-p.x); r.y = p.y+t*(q.y-p.y); r.z = p.z+t*(q.z-p.z); return r; }
	void bracket(const float a, size_t& i0, size_t& i1) const {
		if(a<=a_.front()) i0 = i1 = 0u;
		else if(a>=a_.back()) i0 = i1 = a_.size()-1u;
		else { i1 = (size_t)(std::upper_bound(a_.begin(), a_.end(), a)-a_.begin()); i0 = i1-1u; }
	}
	V3 eval_column(const size_t c, const float b) const {
		const uint32_t s = start_[c], e = start_[c+1u];
		if(e==s) return default_;
		if(e-s==1u) return v_[s];
		if(b<=b_[s]) return v_[s];
		if(b>=b_[e-1u]) return v_[e-1u];
		size_t i1 = (size_t)(std::upper_bound(b_.begin()+s, b_.begin()+e, b)-b_.begin());
		const size_t i0 = i1-1u;
		if(i1>=e) i1 = e-1u;
		const float b0 = b_[i0], b1 = b_[i1];
		const float t = fabsf(b1-b0)>1e-12f ? (b-b0)/(b1-b0) : 0.0f;
		return lerp(v_[i0], v_[i1], t);
	}
public:
	template<typename ValueFn> void build(const std::vector<SurfSample>& samples, const int patch, ValueFn value, const V3& default_value) {
		raw_count_ = 0u; default_ = default_value; a_.clear(); start_.clear(); b_.clear(); v_.clear();
		std::vector<Raw> raw;
		for(const SurfSample& s : samples) { if(s.patch!=patch) continue; float a, b; if(!patch_plane_coords(patch, s.p, a, b)) continue; raw.push_back(Raw{a, b, value(s)}); }
		if(raw.empty()) return;
		raw_count_ = raw.size();
		double sx = 0.0, sy = 0.0, sz = 0.0;
		float amin = raw[0].a, amax = raw[0].a, bmin = raw[0].b, bmax = raw[0].b;
		for(const Raw& r : raw) { sx += (double)r.v.x; sy += (double)r.v.y; sz += (double)r.v.z; amin = fminf(amin, r.a); amax = fmaxf(amax, r.a); bmin = fminf(bmin, r.b); bmax = fmaxf(bmax, r.b); }
		const double inv_n = 1.0/(double)raw.size();
		default_.x = (float)(sx*inv_n); default_.y = (float)(sy*inv_n); default_.z = (float)(sz*inv_n); // mean of the patch
		const float tol_a = fmaxf(1e-6f, 1e-6f*fmaxf(1.0f, amax-amin)), tol_b = fmaxf(1e-6f, 1e-6f*fmaxf(1.0f, bmax-bmin));
		std::sort(raw.begin(), raw.end(), [](const Raw& l, const Raw& r) { if(l.a<r.a) return true; if(l.a>r.a) return false; return l.b<r.b; });
		// columns: a sample joins the current column while it is within tol_a of the column's running mean
		std::vector<size_t> col_begin; std::vector<double> col_sum; std::vector<uint32_t> col_cnt;
		for(size_t i=0u; i<raw.size(); i++) {
			if(!col_begin.empty()) {
				const size_t c = col_begin.size()-1u;
				const float rep = (float)(col_sum[c]/(double)col_cnt[c]);
				if(fabsf(raw[i].a-rep)<=tol_a) { col_sum[c] += (double)raw[i].a; col_cnt[c]++; continue; }
			}
			col_begin.push_back(i); col_sum.push_back((double)raw[i].a); col_cnt.push_back(1u);
		}
		col_begin.push_back(raw.size());
		const size_t nc = col_begin.size()-1u;
		a_.resize(nc); start_.assign(1u, 0u);
		for(size_t c=0u; c<nc; c++) {
			a_[c] = (float)(col_sum[c]/(double)col_cnt[c]);
			std::sort(raw.begin()+(std::ptrdiff_t)col_begin[c], raw.begin()+(std::ptrdiff_t)col_begin[c+1u], [](const Raw& l, const Raw& r) { return l.b<r.b; });
			const size_t first = b_.size();
			double mx = 0.0, my = 0.0, mz = 0.0; uint32_t mc = 0u; // running sums of the entry being merged
			for(size_t i=col_begin[c]; i<col_begin[c+1u]; i++) {
				const Raw& r = raw[i];
				if(b_.size()==first||fabsf(r.b-b_.back())>tol_b) { b_.push_back(r.b); v_.push_back(r.v); mx = (double)r.v.x; my = (double)r.v.y; mz = (double)r.v.z; mc = 1u; }
				else { // near-duplicate b: midpoint of the coordinates, mean of the values
					b_.back() = 0.5f*(b_.back()+r.b);
					mx += (double)r.v.x; my += (double)r.v.y; mz += (double)r.v.z; mc++;
					const double inv = 1.0/(double)mc;
					v_.back().x = (float)(mx*inv); v_.back().y = (float)(my*inv); v_.back().z = (float)(mz*inv);
				}
			}
			start_.push_back((uint32_t)b_.size());
		}
	}
	bool has_samples() const { return raw_count_>0u; }
	size_t raw_count() const { return raw_count_; }
	size_t column_count() const { return a_.size(); }
	V3 eval(const float a, const float b) const {
		if(a_.empty()) return default_;
		if(a_.size()==1u) return eval_column(0u, b);
		size_t i0, i1; bracket(a, i0, i1);
		const V3 v0 = eval_column(i0, b);
		if(i0==i1) return v0;
		const V3 v1 = eval_column(i1, b);
		const float a0 = a_[i0], a1 = a_[i1];
		const float t = fabsf(a1-a0)>1e-12f ? (a-a0)/(a1-a0) : 0.0f;
		return lerp(v0, v1, t);
	}
	bool below_sample_support(const float a, const float b, const float eps = 1e-4f) const { // is (a,b) under the lowest sample of its column(s)?
		if(a_.empty()) return false;
		auto lowest = [&](const size_t c, float& out) { if(start_[c+1u]==start_[c]) return false; out = b_[start_[c]]; return true; };
		if(a_.size()==1u) { float m; return lowest(0u, m) ? b<(m-eps) : false; }
		size_t i0, i1; bracket(a, i0, i1);
		float m0, m1;
		if(!lowest(i0, m0)) return false;
		float m = m0;
		if(i1!=i0) {
			if(!lowest(i1, m1)) return false;
			const float a0 = a_[i0], a1 = a_[i1];
			const float t = fabsf(a1-a0)>1e-12f ? (a-a0)/(a1-a0) : 0.0f;
			m = m0+t*(m1-m0);
		}
		return b<(m-eps);
	}
};

// Terrain height over (x, y) from scattered or gridded points: coordinates are clustered into sorted unique x / y lines
// (tolerance 1e-6 of the extent), each point is binned to its nearest grid node (mean of what lands there, empty nodes
// take the overall mean), evaluation is bilinear with clamping; with fewer than 2 lines in a direction it falls back to the
// nearest raw point (GroundTemperaturePlane2D used as the profile-mode DEM ground plane, FX/setup.cpp:1617-1795,5805-5830).
class GroundPlane2D {
	std::vector<float> xr_, yr_, vr_, xs_, ys_, grid_;
	bool structured_ = false; float default_ = 0.0f;
	static std::vector<float> cluster(std::vector<float> v, const float tol) {
		std::vector<float> out;
		if(v.empty()) return out;
		std::sort(v.begin(), v.end());
		float last = v[0]; out.push_back(last);
		for(size_t i=1u; i<v.size(); i++) {
			if(fabsf(v[i]-last)>tol) { out.push_back(v[i]); last = v[i]; }
			else { out.back() = 0.5f*(out.back()+v[i]); last = out.back(); } // representative stays centred in its cluster
		}
		return out;
	}
	static size_t nearest(const std::vector<float>& a, const float v) {
		auto it = std::lower_bound(a.begin(), a.end(), v);
		if(it==a.begin()) return 0u;
		if(it==a.end()) return a.size()-1u;
		const size_t i1 = (size_t)(it-a.begin()), i0 = i1-1u;
		return fabsf(v-a[i1])<fabsf(v-a[i0]) ? i1 : i0;
	}
	static size_t upper(const std::vector<float>& a, const float v) { auto it = std::upper_bound(a.begin(), a.end(), v); return it==a.end() ? a.size()-1u : (size_t)(it-a.begin()); }
	float nearest_raw(const float xq, const float yq) const {
		float best = FLT_MAX, val = default_;
		for(size_t i=0u; i<vr_.size(); i++) { const float dx = xq-xr_[i], dy = yq-yr_[i], d2 = dx*dx+dy*dy; if(d2<best) { best = d2; val = vr_[i]; } }
		return val;
	}
public:
	void build(const std::vector<float>& x, const std::vector<float>& y, const std::vector<float>& v, const float default_value) {
		xr_ = x; yr_ = y; vr_ = v; xs_.clear(); ys_.clear(); grid_.clear(); structured_ = false; default_ = default_value;
		if(vr_.empty()) return;
		double sum = 0.0; for(const float t : vr_) sum += (double)t;
		default_ = (float)(sum/(double)vr_.size());
		float xmin = xr_[0], xmax = xr_[0], ymin = yr_[0], ymax = yr_[0];
		for(size_t i=1u; i<xr_.size(); i++) { xmin = fminf(xmin, xr_[i]); xmax = fmaxf(xmax, xr_[i]); ymin = fminf(ymin, yr_[i]); ymax = fmaxf(ymax, yr_[i]); }
		xs_ = cluster(xr_, std::max(1e-6f, 1e-6f*fmaxf(1.0f, xmax-xmin)));
		ys_ = cluster(yr_, std::max(1e-6f, 1e-6f*fmaxf(1.0f, ymax-ymin)));
		const size_t nx = xs_.size(), ny = ys_.size();
		if(nx==0u||ny==0u) return;
		std::vector<double> acc(nx*ny, 0.0); std::vector<uint32_t> cnt(nx*ny, 0u);
		for(size_t i=0u; i<vr_.size(); i++) { const size_t id = nearest(ys_, yr_[i])*nx+nearest(xs_, xr_[i]); acc[id] += (double)vr_[i]; cnt[id]++; }
		grid_.assign(nx*ny, default_);
		for(size_t id=0u; id<grid_.size(); id++) if(cnt[id]>0u) grid_[id] = (float)(acc[id]/(double)cnt[id]);
		structured_ = nx>=2u&&ny>=2u;
	}
	bool has_samples() const { return !vr_.empty(); }
	bool structured() const { return structured_; }
	size_t nx() const { return xs_.size(); } size_t ny() const { return ys_.size(); }
	float eval(const float xq, const float yq) const {
		if(vr_.empty()) return default_;
		const size_t nx = xs_.size(), ny = ys_.size();
		if(!structured_||nx<2u||ny<2u) return nearest_raw(xq, yq);
		const float x = fminf(fmaxf(xq, xs_.front()), xs_.back()), y = fminf(fmaxf(yq, ys_.front()), ys_.back());
		const size_t ix1 = upper(xs_, x), iy1 = upper(ys_, y);
		const size_t ix0 = ix1==0u ? 0u : ix1-1u, iy0 = iy1==0u ? 0u : iy1-1u;
		const size_t ia = ix0>=nx-1u ? nx-2u : ix0, ja = iy0>=ny-1u ? ny-2u : iy0, ib = ia+1u, jb = ja+1u;
		const float xa = xs_[ia], xb = xs_[ib], ya = ys_[ja], yb = ys_[jb];
		const float tx = fabsf(xb-xa)>1e-12f ? (x-xa)/(xb-xa) : 0.0f, ty = fabsf(yb-ya)>1e-12f ? (y-ya)/(yb-ya) : 0.0f;
		const float t00 = grid_[ja*nx+ia], t10 = grid_[ja*nx+ib], t01 = grid_[jb*nx+ia], t11 = grid_[jb*nx+ib];
		const float t0 = t00+tx*(t10-t00), t1 = t01+tx*(t11-t01);
		return t0+ty*(t1-t0);
	}
};

struct DemPoints { std::vector<float> x, y, e; float xmin = 0, xmax = 0, ymin = 0, ymax = 0, emin = 0, emax = 0; };
// proj_temp/interpolated_dem.csv: header x,y,elevation (or z), or three positional columns; `;` and tabs count as commas
inline DemPoints read_dem_csv(const std::string& path) { // FX/setup.cpp:2153-2241
	DemPoints d;
	std::ifstream fin(path);
	if(!fin.is_open()) return d;
	auto split = [](const std::string& s) { std::vector<std::string> c; std::stringstream ss(s); std::string t; while(std::getline(ss, t, ',')) c.push_back(bc_trim(t)); return c; };
	auto lower = [](std::string s) { for(char& ch : s) ch = (char)std::tolower((unsigned char)ch); return s; };
	std::string header;
	if(!std::getline(fin, header)) return d;
	const std::vector<std::string> hc = split(header);
	auto col = [&](const char* key) { for(size_t i=0u; i<hc.size(); i++) if(lower(hc[i])==key) return (int)i; return -1; };
	const int ix = col("x"), iy = col("y"); int ie = col("elevation"); if(ie<0) ie = col("z");
	const bool named = ix>=0&&iy>=0&&ie>=0;
	float xmin = +FLT_MAX, xmax = -FLT_MAX, ymin = +FLT_MAX, ymax = -FLT_MAX, emin = +FLT_MAX, emax = -FLT_MAX;
	std::string line;
	while(std::getline(fin, line)) {
		if(line.empty()) continue;
		for(char& ch : line) if(ch==';'||ch=='\t') ch = ',';
		const std::vector<std::string> c = split(line);
		float x, y, e;
		if(named) { if((int)c.size()<=std::max(ix, std::max(iy, ie))) continue; x = (float)atof(c[ix].c_str()); y = (float)atof(c[iy].c_str()); e = (float)atof(c[ie].c_str()); }
		else { if(c.size()<3u) continue; x = (float)atof(c[0].c_str()); y = (float)atof(c[1].c_str()); e = (float)atof(c[2].c_str()); }
		if(!std::isfinite(x)||!std::isfinite(y)||!std::isfinite(e)) continue;
		d.x.push_back(x); d.y.push_back(y); d.e.push_back(e);
		xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); ymin = fminf(ymin, y); ymax = fmaxf(ymax, y); emin = fminf(emin, e); emax = fmaxf(emax, e);
	}
	if(!d.x.empty()) { d.xmin = xmin; d.xmax = xmax; d.ymin = ymin; d.ymax = ymax; d.emin = emin; d.emax = emax; }
	return d;
}

// view of the solver's host mirrors
struct HostLattice {
	uint32_t Nx = 1u, Ny = 1u, Nz = 1u; uint8_t* flags = nullptr; float* u = nullptr; // u SoA: x[N], y[N], z[N]
	uint64_t N() const { return (uint64_t)Nx*Ny*Nz; }
	void coords(const uint64_t n, uint32_t& x, uint32_t& y, uint32_t& z) const { const uint64_t t = n%((uint64_t)Nx*Ny); x = (uint32_t)(t%Nx); y = (uint32_t)(t/Nx); z = (uint32_t)(n/((uint64_t)Nx*Ny)); }
	uint64_t index(const uint32_t x, const uint32_t y, const uint32_t z) const { return (uint64_t)x+((uint64_t)y+(uint64_t)z*Ny)*Nx; }
	V3 position(const uint32_t x, const uint32_t y, const uint32_t z) const { V3 p; p.x = (float)x-0.5f*(float)Nx+0.5f; p.y = (float)y-0.5f*(float)Ny+0.5f; p.z = (float)z-0.5f*(float)Nz+0.5f; return p; }
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
inline PatchBcCounts apply_patch_boundaries(HostLattice& L, const std::vector<PatchField2D>& fields, const PatchField2D& ground, const std::string& downstream_bc, const bool downstream_open_face, const int side_ref_z_cap) {
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
	static bool solve6(double A[6][6], double bx[6], double by[6], double bz[6], double ax[6], double ay[6], double az[6]) { // Gaussian elimination, partial pivoting, 3 right-hand sides
		for(int k=0; k<6; k++) {
			int piv = k; double big = std::fabs(A[k][k]);
			for(int i=k+1; i<6; i++) { const double v = std::fabs(A[i][k]); if(v>big) { big = v; piv = i; } }
			if(big<1e-18) return false;
			if(piv!=k) { for(int j=0; j<6; j++) std::swap(A[k][j], A[piv][j]); std::swap(bx[k], bx[piv]); std::swap(by[k], by[piv]); std::swap(bz[k], bz[piv]); }
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
		for(const V3& p : c.P) { if(p.x<xmin_) xmin_ = p.x; if(p.x>xmax_) xmax_ = p.x; if(p.y<ymin_) ymin_ = p.y; if(p.y>ymax_) ymax_ = p.y; if(p.z<zmin_) zmin_ = p.z; if(p.z>zmax_) zmax_ = p.z; }
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
		auto weight = [&](const int idx, double& q1, double& q2) { float s1, s2; local(plane, c_.P[idx], pos, s1, s2); q1 = (double)s1; q2 = (double)s2; return std::exp(-(q1*q1+q2*q2)/(2.0*sigma2)); };
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
		for(int k=0; k<filled; k++) { double q1, q2; const double w = weight(best_i[k], q1, q2); const V3& u = c_.U[best_i[k]]; wx += w*(double)u.x; wy += w*(double)u.y; wz += w*(double)u.z; ws += w; }
		if(ws<=0.0) return zero;
		const double inv = 1.0/ws;
		V3 r; r.x = (float)(wx*inv); r.y = (float)(wy*inv); r.z = (float)(wz*inv); return r;
	}
};

// Fill for the sample-cloud interpolators (FX/interpolation.cpp:71-209 and FX/interpolation_hd.cpp:437-745): z = 0 plane solid,
// outer faces TYPE_E with u = inlet(position), downstream face left without velocity when it is open.  The two reference
// variants differ in one detail that is kept: the nearest-sample version leaves non-inlet cells of an open downstream face
// TYPE_E too (same result).  `inlet` must be thread-safe.
inline uint64_t apply_cloud_boundaries(HostLattice& L, const std::string& downstream_bc, const bool downstream_open_face, const int side_ref_z_cap, const std::function<V3(const V3&)>& inlet) {
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
inline void apply_patch_temperature(HostLattice& L, float* T, const std::vector<PatchField2D>& tfields, const std::string& downstream_bc, const bool downstream_open_face, const float Tmin, const float Tmax, TemperatureCounts& cnt) {
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
inline void apply_cloud_temperature(HostLattice& L, float* T, const std::string& downstream_bc, const bool downstream_open_face, const bool high_order, const float z_threshold, const float Tmin, const float Tmax,
		const std::function<float(const V3&)>& interp, TemperatureCounts& cnt) {
	const int dp = downstream_to_patch(downstream_bc);
	std::vector<uint64_t> cells;
	for(uint64_t n=0ull; n<L.N(); n++) {
		uint32_t x, y, z; L.coords(n, x, y, z);
		if(z==0u||!(x==0u||x==L.Nx-1u||y==0u||y==L.Ny-1u||z==L.Nz-1u)) continue;
		if(downstream_open_face) {
			if(high_order) { const int face_patch = x==0u ? PATCH_WEST : x==L.Nx-1u ? PATCH_EAST : y==0u ? PATCH_SOUTH : y==L.Ny-1u ? PATCH_NORTH : PATCH_TOP; if(face_patch==dp) continue; } // face priority x, y, z
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
	bc_parallel_for((uint64_t)col.size(), [&](const uint64_t id) { const V3 p = L.position((uint32_t)(id%L.Nx), (uint32_t)(id/L.Nx), 0u); col[id] = fminf(fmaxf(plane.eval(p.x, p.y), Tmin), Tmax); });
	std::vector<uint8_t> used(col.size(), 0u);
	uint64_t cells = 0ull;
	for(uint64_t n=0ull; n<L.N(); n++) {
		if(!(L.flags[n]&0x01u)) continue;
		const uint64_t id = n%((uint64_t)L.Nx*L.Ny);
		T[n] = col[id]; L.flags[n] = (uint8_t)(L.flags[n]|0x01u|0x04u); used[id] = 1u; cells++;
	}
	cnt.ground_cells = cells; cnt.ground_columns = 0ull; for(const uint8_t v : used) cnt.ground_columns += v;
}
struct TemperatureSummary { uint64_t total = 0ull, solid = 0ull, fluid = 0ull, invalid = 0ull; float smin = +FLT_MAX, smax = -FLT_MAX, fmin = +FLT_MAX, fmax = -FLT_MAX; };
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

struct FluxReport { double S_in = 0.0, S_out = 0.0, net_before = 0.0, net_after = 0.0, delta = 0.0, avg_delta = 0.0; uint64_t corrected = 0ull; double face_avg[5] = {0, 0, 0, 0, 0}; /* Xn, Xp, Yn, Yp, Zp */ };

// Uniform shift of the outward-normal velocity on all non-solid outer-face cells so that the net boundary flux vanishes;
// cells of the downstream face first receive `downstream_fill(x, y, z)` when given (FX/fluxcorrection.cpp:28-194).
inline FluxReport apply_flux_correction(HostLattice& L, const std::string& downstream_bc, const std::function<V3(uint32_t, uint32_t, uint32_t)>& downstream_fill) {
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
	auto normal = [&](const Cell& c) -> float { switch(c.face) { case ZP: return uz[c.n]; case XN: return -ux[c.n]; case XP: return ux[c.n]; case YN: return -uy[c.n]; default: return uy[c.n]; } };
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
