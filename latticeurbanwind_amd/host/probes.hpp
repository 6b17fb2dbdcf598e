// Probe columns of the deck key `probes` (FX/setup.cpp:1160-1616 parsing and geographic mapping, :4268-4395 resolution,
// :4495-4506 sampling, :4718-4760 CSV): a probe is a lon:lat point (or the domain centre) plus an optional offset in grid
// cells ("NNE") or metres ("N10E5.5"); it resolves to one (x, y) column of the lattice whose non-solid cells are sampled
// every step of the probe window and written as RESULTS/<stem>.csv (one row per level, one u:v:w column per time).
// Geographic mapping: WGS84 lon/lat -> UTM (transverse Mercator series, k0 = 0.9996), rotated about the centroid of the
// cut rectangle's corners so that its south edge runs along +x, origin at the rotated rectangle's minimum corner.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

namespace luw_host {

struct ProbeOffset {
	enum Mode { NONE, CELLS, METERS }
	mode = NONE;
	int north_cells = 0, east_cells = 0;
	double north_m = 0.0, east_m = 0.0;
	std::string label;
};
struct ProbeRequest { std::string raw; double lon = 0.0, lat = 0.0; bool centre = false; ProbeOffset off; };
struct GeoFrame {
	bool valid = false;
	int zone = 0;
	bool north = true;
	double rot_deg = 0.0, px = 0.0, py = 0.0, xmin = 0.0, ymin = 0.0, clon = 0.0, clat = 0.0, ex = 1.0, ey = 0.0, nx = 0.0, ny = 1.0;
};
struct ProbeColumn {
	ProbeRequest req;
	std::string stem;
	uint32_t x = 0u, y = 0u;
	std::vector<uint32_t> z;
	std::vector<float> height_si;
	std::vector<double> time_si;
	std::vector<float> uvw_si;
	/* [time][level][3] */
};

inline std::string pr_trim(const std::string& s) {
	const char* ws = " \t\r\n";
	const size_t b = s.find_first_not_of(ws), e = s.find_last_not_of(ws);
	return b==std::string::npos ? std::string() : s.substr(b, e-b+1u);
}
inline std::string fixed_trimmed(const double v, const int prec = 6) { // "%.6f" without trailing zeros
	std::ostringstream o; o << std::fixed << std::setprecision(prec) << v;
	std::string s = o.str();
	if(s.find('.')==std::string::npos) return s;
	while(!s.empty()&&s.back()=='0') s.pop_back();
	if(!s.empty()&&s.back()=='.') s.pop_back();
	return s.empty() ? std::string("0") : s;
}
inline std::string file_safe(std::string s) {
	for(char& c : s) if(!((c>='0'&&c<='9')||(c>='a'&&c<='z')||(c>='A'&&c<='Z')||c=='_'||c=='-'||c=='.')) c = '_';
	while(!s.empty()&&(s.back()=='.'||s.back()==' ')) s.pop_back();
	return s.empty() ? std::string("probe") : s;
}

inline std::vector<std::string> split_probe_list(const std::string& raw) { // [a, "b" NE, c] -> tokens; commas inside quotes do not split
	std::string s = pr_trim(raw);
	const size_t lb = s.find('['), rb = s.rfind(']');
	if(lb!=std::string::npos&&rb!=std::string::npos&&rb>lb) s = s.substr(lb+1u, rb-lb-1u);
	std::vector<std::string> out; std::string tok; char q = 0;
	auto flush = [&]() { const std::string t = pr_trim(tok); if(!t.empty()) out.push_back(t); tok.clear(); };
	for(const char c : s) {
		if(q) { tok.push_back(c); if(c==q) q = 0; }
		else if(c=='"'||c=='\'') { q = c; tok.push_back(c); }
		else if(c==',') flush();
		else tok.push_back(c);
	}
	flush();
	return out;
}
inline bool parse_offset(const std::string& raw, ProbeOffset& o, std::string& err) {
	o = ProbeOffset();
	std::string s; for(const char c : raw) if(!std::isspace((unsigned char)c)) s.push_back((char)std::toupper((unsigned char)c));
	if(s.empty()) return true;
	o.label = s;
	if(std::none_of(s.begin(), s.end(), [](const char c) { return c>='0'&&c<='9'; })) { // letters only: one cell per letter
		o.mode = ProbeOffset::CELLS;
		for(const char c : s) {
			if(c=='N') o.north_cells++;
			else if(c=='S') o.north_cells--;
			else if(c=='E') o.east_cells++;
			else if(c=='W') o.east_cells--;
			else { err = "grid offset can only contain N/S/E/W"; return false; }
		}
		return true;
	}
	o.mode = ProbeOffset::METERS;
	for(size_t i=0u; i<s.size();) {
		const char d = s[i];
		if(d!='N'&&d!='S'&&d!='E'&&d!='W') { err = "meter offset must use N/S/E/W followed by a number"; return false; }
		i++;
		char* end = nullptr; const double v = std::strtod(s.c_str()+i, &end);
		if(end==s.c_str()+i||!std::isfinite(v)) { err = "meter offset is missing a numeric value after direction"; return false; }
		if(v<0.0) { err = "meter offset cannot be negative"; return false; }
		if(d=='N') o.north_m += v; else if(d=='S') o.north_m -= v; else if(d=='E') o.east_m += v; else o.east_m -= v;
		i = (size_t)(end-s.c_str());
	}
	return true;
}
inline bool parse_probe(const std::string& token_in, ProbeRequest& r, std::string& err) {
	r = ProbeRequest(); r.raw = pr_trim(token_in);
	const std::string& t = r.raw;
	if(t.empty()) { err = "empty probe token"; return false; }
	auto lower = [](std::string s) { for(char& c : s) c = (char)std::tolower((unsigned char)c); return s; };
	auto centre_word = [&](const std::string& word, const std::string& rest) {
		const std::string k = lower(pr_trim(word));
		if(k!="center"&&k!="centre") return false;
		r.centre = true;
		return parse_offset(rest, r.off, err);
	};
	if(t.front()=='"'||t.front()=='\'') {
		const size_t close = t.find(t.front(), 1u);
		if(close==std::string::npos) { err = "quoted probe token is missing the closing quote"; return false; }
		if(!centre_word(t.substr(1u, close-1u), pr_trim(t.substr(close+1u)))) {
			if(err.empty()) err = "quoted probe token only supports center/centre";
			return false;
		}
		return true;
	}
	const std::string tl = lower(t);
	if(tl.compare(0u, 6u, "center")==0) return centre_word("center", pr_trim(t.substr(6u)));
	if(tl.compare(0u, 6u, "centre")==0) return centre_word("centre", pr_trim(t.substr(6u)));
	const size_t colon = t.find(':');
	if(colon==std::string::npos) { err = "probe must be lon:lat, center, or centre"; return false; }
	const std::string lon_s = pr_trim(t.substr(0u, colon)), rest = pr_trim(t.substr(colon+1u));
	if(lon_s.empty()||rest.empty()) { err = "probe lon:lat is incomplete"; return false; }
	char* e1 = nullptr; const double lon = std::strtod(lon_s.c_str(), &e1);
	if(e1==lon_s.c_str()||*e1!='\0'||!std::isfinite(lon)) { err = "invalid probe longitude"; return false; }
	char* e2 = nullptr; const double lat = std::strtod(rest.c_str(), &e2);
	if(e2==rest.c_str()||!std::isfinite(lat)) { err = "invalid probe latitude"; return false; }
	r.lon = lon; r.lat = lat;
	return parse_offset(pr_trim(rest.substr((size_t)(e2-rest.c_str()))), r.off, err);
}

// WGS84 -> UTM: the forward transverse-Mercator series of Snyder, "Map Projections -- A Working Manual" (USGS PP 1395, 1987), eqs. 8-9, 8-10,
// 3-21 with the WGS84 ellipsoid and k0 = 0.9996 -- the textbook form the reference uses too (FX/setup.cpp:1288-1337); the result only
// selects a lattice column, so agreement to the last bit is not required (and is not claimed)
inline bool utm_forward(const double lon_deg, const double lat_deg, const int zone, const bool north, double& E, double& Nn) {
	if(zone<1||zone>60||!std::isfinite(lon_deg)||!std::isfinite(lat_deg)||lat_deg<=-90.0||lat_deg>=90.0) return false;
	constexpr double pi = 3.1415926535897932384626433832795, a = 6378137.0, f = 1.0/298.257223563, k0 = 0.9996;
	const double e2 = f*(2.0-f), ep2 = e2/(1.0-e2);
	const double phi = lat_deg*(pi/180.0), lam = lon_deg*(pi/180.0), lam0 = ((double)zone*6.0-183.0)*(pi/180.0);
	const double sp = std::sin(phi), cp = std::cos(phi), tp = std::tan(phi);
	const double Nr = a/std::sqrt(1.0-e2*sp*sp), T = tp*tp, Cc = ep2*cp*cp, A = cp*(lam-lam0);
	const double M = a*((1.0-e2/4.0-3.0*e2*e2/64.0-5.0*e2*e2*e2/256.0)*phi-(3.0*e2/8.0+3.0*e2*e2/32.0+45.0*e2*e2*e2/1024.0)*std::sin(2.0*phi)
		+(15.0*e2*e2/256.0+45.0*e2*e2*e2/1024.0)*std::sin(4.0*phi)-(35.0*e2*e2*e2/3072.0)*std::sin(6.0*phi));
	E = 500000.0+k0*Nr*(A+(1.0-T+Cc)*A*A*A/6.0+(5.0-18.0*T+T*T+72.0*Cc-58.0*ep2)*A*A*A*A*A/120.0);
	Nn = k0*(M+Nr*tp*(A*A/2.0+(5.0-T+9.0*Cc+4.0*Cc*Cc)*A*A*A*A/24.0+(61.0-58.0*T+T*T+600.0*Cc-330.0*ep2)*A*A*A*A*A*A/720.0));
	if(!north) Nn += 10000000.0;
	return std::isfinite(E)&&std::isfinite(Nn);
}
inline void rotate_about(const double x, const double y, const double deg, const double cx, const double cy, double& xr, double& yr) {
	const double th = deg*(3.1415926535897932384626433832795/180.0), c = std::cos(th), s = std::sin(th), dx = x-cx, dy = y-cy;
	xr = c*dx-s*dy+cx; yr = s*dx+c*dy+cy;
}
inline GeoFrame make_geo_frame(const float lon0, const float lon1, const float lat0, const float lat1, const std::string& utm_crs, const bool has_rot,
	const double rot_override) {
	GeoFrame g;
	if(!std::isfinite(lon0)||!std::isfinite(lon1)||!std::isfinite(lat0)||!std::isfinite(lat1)) return g;
	const double lo = std::min((double)lon0, (double)lon1), hi = std::max((double)lon0, (double)lon1), la = std::min((double)lat0, (double)lat1),
		lb = std::max((double)lat0, (double)lat1);
	if(!(hi>lo)||!(lb>la)) return g;
	int zone = 0; bool north = true; bool from_crs = false;
	{ std::string d; for(const char c : pr_trim(utm_crs)) if(c>='0'&&c<='9') d.push_back(c); // EPSG 326zz / 327zz
	  if(!d.empty()) {
		const int code = atoi(d.c_str());
		if(code>=32601&&code<=32660) {
			zone = code-32600;
			north = true;
			from_crs = true;
		} else if(code>=32701&&code<=32760) { zone = code-32700; north = false; from_crs = true; }
	} }
	if(!from_crs) { zone = (int)std::floor((0.5*(lo+hi)+180.0)/6.0)+1; zone = std::min(60, std::max(1, zone)); north = 0.5*(la+lb)>=0.0; }
	double x[4], y[4]; // corners: SW, SE, NE, NW
	if(!utm_forward(lo, la, zone, north, x[0], y[0])||!utm_forward(hi, la, zone, north, x[1], y[1])||!utm_forward(hi, lb, zone, north, x[2], y[2])
		||!utm_forward(lo, lb, zone, north, x[3], y[3])) return g;
	const double cx = 0.25*(x[0]+x[1]+x[2]+x[3]), cy = 0.25*(y[0]+y[1]+y[2]+y[3]);
	const double rot = has_rot ? rot_override : (-std::atan2(y[1]-y[0], x[1]-x[0])*180.0/3.1415926535897932384626433832795);
	double xr[4], yr[4]; for(int k=0; k<4; k++) rotate_about(x[k], y[k], rot, cx, cy, xr[k], yr[k]);
	const double th = rot*(3.1415926535897932384626433832795/180.0);
	g.valid = true; g.zone = zone; g.north = north; g.rot_deg = rot; g.px = cx; g.py = cy;
	g.xmin = std::min(std::min(xr[0], xr[1]), std::min(xr[2], xr[3])); g.ymin = std::min(std::min(yr[0], yr[1]), std::min(yr[2], yr[3]));
	g.clon = 0.5*(lo+hi); g.clat = 0.5*(la+lb);
	g.ex = std::cos(th); g.ey = std::sin(th); g.nx = -std::sin(th); g.ny = std::cos(th);
	return g;
}
inline bool geo_to_local(const GeoFrame& g, const double lon, const double lat, double& xs, double& ys) {
	if(!g.valid) return false;
	double E, Nn; if(!utm_forward(lon, lat, g.zone, g.north, E, Nn)) return false;
	double xr, yr; rotate_about(E, Nn, g.rot_deg, g.px, g.py, xr, yr);
	xs = xr-g.xmin; ys = yr-g.ymin;
	return std::isfinite(xs)&&std::isfinite(ys);
}
inline uint32_t snap_index(const double coord_si, const uint32_t n, const float cell_m) {
	if(n==0u||!(cell_m>0.0f)) return 0u;
	const long i = (long)std::llround(coord_si/(double)cell_m);
	return i<=0l ? 0u : ((uint64_t)i>=(uint64_t)n ? n-1u : (uint32_t)i);
}
// lon/lat (+offset) -> lattice column; false with a reason when the point leaves the domain
inline bool resolve_probe_xy(const ProbeRequest& r, const GeoFrame& g, const uint32_t Nx, const uint32_t Ny, const float cell_m, const float six,
	const float siy, uint32_t& x, uint32_t& y, std::string& why) {
	double bx, by;
	if(!geo_to_local(g, r.centre ? g.clon : r.lon, r.centre ? g.clat : r.lat, bx, by)) { why = "projection failed"; return false; }
	auto inside = [&](const double a, const double b) { return std::isfinite(a)&&std::isfinite(b)&&a>=0.0&&a<=(double)six&&b>=0.0&&b<=(double)siy; };
	if(!inside(bx, by)) { why = "base point is outside CFD domain"; return false; }
	double fx = bx, fy = by;
	if(r.off.mode==ProbeOffset::CELLS) {
		fx = (double)snap_index(bx, Nx, cell_m)*(double)cell_m+(double)r.off.east_cells*(double)cell_m*g.ex+(double)r.off.north_cells*(double)cell_m*g.nx;
		fy = (double)snap_index(by, Ny, cell_m)*(double)cell_m+(double)r.off.east_cells*(double)cell_m*g.ey+(double)r.off.north_cells*(double)cell_m*g.ny;
	} else if(r.off.mode==ProbeOffset::METERS) {
		fx = bx+r.off.east_m*g.ex+r.off.north_m*g.nx;
		fy = by+r.off.east_m*g.ey+r.off.north_m*g.ny;
	}
	if(!inside(fx, fy)) { why = "offset point is outside CFD domain"; return false; }
	x = snap_index(fx, Nx, cell_m); y = snap_index(fy, Ny, cell_m);
	return true;
}
inline std::string probe_stem(const ProbeRequest& r, const GeoFrame& g, const std::string& prefix) {
	std::string s = fixed_trimmed(r.centre ? g.clon : r.lon)+"_"+fixed_trimmed(r.centre ? g.clat : r.lat);
	if(!r.off.label.empty()) s += "_"+file_safe(r.off.label);
	if(!prefix.empty()) s = file_safe(prefix)+s;
	return file_safe(s);
}
inline bool write_probe_csv(const std::string& path, const ProbeColumn& p) {
	std::ofstream f(path, std::ios::out|std::ios::trunc);
	if(!f.is_open()) return false;
	f << "height (m)";
	for(const double t : p.time_si) f << "," << fixed_trimmed(t, 6);
	f << "\n";
	const size_t L = p.z.size(), T = p.time_si.size();
	for(size_t l=0u; l<L; l++) {
		f << fixed_trimmed((double)p.height_si[l], 6);
		for(size_t t=0u; t<T; t++) {
			const size_t b = (t*L+l)*3u;
			f << "," << fixed_trimmed((double)p.uvw_si[b], 6) << ":" << fixed_trimmed((double)p.uvw_si[b+1u], 6) << ":" << fixed_trimmed((double)p.uvw_si[b+2u],
				6);
		}
		f << "\n";
	}
	return true;
}

} // namespace luw_host
