// bc_inputs.hpp -- what the boundary builders (bc_builders.hpp) read and interpolate from: the SurfData CSV of NWP decks, the patch-driven 2-D face
// fields, the ground plane and the DEM points of profile decks.  Host-side set-up of the deck driver; arithmetic follows the reference
// statement by statement where values depend on it (FP32 on the host without contraction, doubles where the reference uses doubles):
//   SurfData CSV reader            FX/setup.cpp:2293-2462
//   patch-driven 2-D face fields   FX/setup.cpp:1796-2094 (PatchSurfaceField2D)
//   ground plane, DEM CSV          FX/setup.cpp:2096-2241
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
inline const char* patch_name(const int p) {
	static const char* n[6] = {"bottom", "top", "south", "north", "west", "east"};
	return p>=0&&p<6 ? n[p] : "unknown";
}
inline int downstream_to_patch(const std::string& bc) {
	return bc=="+y" ? PATCH_NORTH : bc=="-y" ? PATCH_SOUTH : bc=="+x" ? PATCH_EAST : bc=="-x" ? PATCH_WEST : -1;
}
// top wins over sides, FX/setup.cpp:1816-1823
inline int boundary_cell_to_patch(const uint32_t x, const uint32_t y, const uint32_t z, const uint32_t Nx, const uint32_t Ny, const uint32_t Nz) {
	if(z==Nz-1u) return PATCH_TOP;
	if(x==0u) return PATCH_WEST;
	if(x==Nx-1u) return PATCH_EAST;
	if(y==0u) return PATCH_SOUTH;
	if(y==Ny-1u) return PATCH_NORTH;
	return -1;
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

inline std::string bc_trim(const std::string& s) {
	const char* ws = " \t\r\n";
	const size_t b = s.find_first_not_of(ws), e = s.find_last_not_of(ws);
	return b==std::string::npos ? std::string() : s.substr(b, e-b+1u);
}

// SurfData_<datetime>.csv: header X,Y,Z,u,v,w[,T][,patch] (any order, case-insensitive) or legacy positional 6..8 columns
inline bool read_surfdata_csv(const std::string& path, SurfData& out) {
	out = SurfData();
	std::ifstream fin(path);
	if(!fin.is_open()) return false;
	auto split = [](const std::string& s) {
		std::vector<std::string> c;
		std::stringstream ss(s);
		std::string t;
		while(std::getline(ss, t, ',')) c.push_back(bc_trim(t));
		return c;
	};
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
			if((int)c.size()<=need) {
				out.warnings.push_back("WARNING: malformed line "+std::to_string(line_no)+" in CSV (missing required columns)");
				continue;
			}
			s.p.x = (float)atof(c[ix].c_str()); s.p.y = (float)atof(c[iy].c_str()); s.p.z = (float)atof(c[iz].c_str());
			s.u.x = (float)atof(c[iu].c_str()); s.u.y = (float)atof(c[iv].c_str()); s.u.z = (float)atof(c[iw].c_str());
			if(it>=0&&(int)c.size()>it) {
				s.T = (float)atof(c[it].c_str());
				out.has_T = true;
				out.rows_T++;
				tmin = std::fmin(tmin, s.T);
				tmax = std::fmax(tmax, s.T);
			}
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
		for(const SurfSample& s : samples) {
			if(s.patch!=patch) continue;
			float a, b;
			if(!patch_plane_coords(patch, s.p, a, b)) continue;
			raw.push_back(Raw{a, b, value(s)});
		}
		if(raw.empty()) return;
		raw_count_ = raw.size();
		double sx = 0.0, sy = 0.0, sz = 0.0;
		float amin = raw[0].a, amax = raw[0].a, bmin = raw[0].b, bmax = raw[0].b;
		for(const Raw& r : raw) {
			sx += (double)r.v.x;
			sy += (double)r.v.y;
			sz += (double)r.v.z;
			amin = fminf(amin, r.a);
			amax = fmaxf(amax, r.a);
			bmin = fminf(bmin, r.b);
			bmax = fmaxf(bmax, r.b);
		}
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
			std::sort(raw.begin()+(std::ptrdiff_t)col_begin[c], raw.begin()+(std::ptrdiff_t)col_begin[c+1u], [](const Raw& l, const Raw& r) {
				return l.b<r.b;
			});
			const size_t first = b_.size();
			double mx = 0.0, my = 0.0, mz = 0.0; uint32_t mc = 0u; // running sums of the entry being merged
			for(size_t i=col_begin[c]; i<col_begin[c+1u]; i++) {
				const Raw& r = raw[i];
				if(b_.size()==first||fabsf(r.b-b_.back())>tol_b) {
					b_.push_back(r.b);
					v_.push_back(r.v);
					mx = (double)r.v.x;
					my = (double)r.v.y;
					mz = (double)r.v.z;
					mc = 1u;
				}
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
	static size_t upper(const std::vector<float>& a, const float v) {
		auto it = std::upper_bound(a.begin(), a.end(), v);
		return it==a.end() ? a.size()-1u : (size_t)(it-a.begin());
	}
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
	auto split = [](const std::string& s) {
		std::vector<std::string> c;
		std::stringstream ss(s);
		std::string t;
		while(std::getline(ss, t, ',')) c.push_back(bc_trim(t));
		return c;
	};
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
		if(named) {
			if((int)c.size()<=std::max(ix, std::max(iy, ie))) continue;
			x = (float)atof(c[ix].c_str());
			y = (float)atof(c[iy].c_str());
			e = (float)atof(c[ie].c_str());
		}
		else { if(c.size()<3u) continue; x = (float)atof(c[0].c_str()); y = (float)atof(c[1].c_str()); e = (float)atof(c[2].c_str()); }
		if(!std::isfinite(x)||!std::isfinite(y)||!std::isfinite(e)) continue;
		d.x.push_back(x); d.y.push_back(y); d.e.push_back(e);
		xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); ymin = fminf(ymin, y); ymax = fmaxf(ymax, y); emin = fminf(emin, e); emax = fmaxf(emax, e);
	}
	if(!d.x.empty()) { d.xmin = xmin; d.xmax = xmax; d.ymin = ymin; d.ymax = ymax; d.emin = emin; d.emax = emax; }
	return d;
}

} // namespace luw_host
