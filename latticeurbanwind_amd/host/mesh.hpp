// mesh.hpp -- binary STL, the LUW mesh transform and the host voxeliser of --dry-run
// Part of the deck driver (luw_driver.cpp); included by it only, after lbm.hpp (namespace luw_host, std::string as string).
#pragma once

// ------------------------------------------------------------------------------------------------ mesh + host voxeliser
struct Mesh { std::vector<float> p0, p1, p2; uint n = 0u; float pmin[3], pmax[3]; };
static void mesh_find_bounds(Mesh& m) { // FX/utilities.hpp:4774-4785: seeded with p0[0] only
	for(int c=0; c<3; c++) m.pmin[c] = m.pmax[c] = m.p0[c];
	for(uint i=1u; i<m.n; i++) for(int c=0; c<3; c++) {
		m.pmin[c] = std::fmin(std::fmin(std::fmin(m.p0[3u*i+c], m.p1[3u*i+c]), m.p2[3u*i+c]), m.pmin[c]);
		m.pmax[c] = std::fmax(std::fmax(std::fmax(m.p0[3u*i+c], m.p1[3u*i+c]), m.p2[3u*i+c]), m.pmax[c]);
	}
}
static bool read_stl(const string& path, Mesh& m) { // binary STL only, FX/utilities.hpp:4835-4866
	std::ifstream f(path, std::ios::in|std::ios::binary);
	if(f.fail()) return false;
	std::vector<char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	if(data.size()<84u) return false;
	uint tn; std::memcpy(&tn, data.data()+80, 4);
	if(tn==0u||data.size()!=84u+50ull*tn) fatal("| Error: File \""+path+"\" is corrupt or unsupported! Only binary .stl files are supported.", 1);
	m.n = tn; m.p0.resize(3u*tn); m.p1.resize(3u*tn); m.p2.resize(3u*tn);
	for(uint i=0u; i<tn; i++) {
		const char* t = data.data()+84u+50ull*i;
		std::memcpy(&m.p0[3u*i], t+12, 12);
		std::memcpy(&m.p1[3u*i], t+24, 12);
		std::memcpy(&m.p2[3u*i], t+36, 12);
	}
	mesh_find_bounds(m);
	return true;
}
static void mesh_scale_translate(Mesh& m, const float scale) { // scale about center 0, then pmin -> (1,1,1): FX/setup.cpp:4086-4087
	for(auto* v : {&m.p0, &m.p1, &m.p2}) for(float& f : *v) f = scale*f;
	for(int c=0; c<3; c++) { m.pmin[c] = scale*m.pmin[c]; m.pmax[c] = scale*m.pmax[c]; }
	float tr[3]; for(int c=0; c<3; c++) tr[c] = 1.0f-m.pmin[c];
	for(auto* v : {&m.p0, &m.p1, &m.p2}) for(size_t i=0u; i<v->size(); i++) (*v)[i] += tr[i%3u];
	for(int c=0; c<3; c++) { m.pmin[c] += tr[c]; m.pmax[c] += tr[c]; }
}
// voxelize_mesh with direction 2 and flag TYPE_S on one domain (FX/kernel.cpp:2381-2471, FX/lbm.cpp:498), on the host.
// IEEE 1/g here vs the device reciprocal of the OpenCL build: faces exactly on lattice planes may land one cell off
// (DESIGN.md section 3).
static ulong voxelize_z(const Mesh& m, const uint Nx, const uint Ny, const uint Nz, std::vector<uchar>& flags) {
	const float x0 = m.pmin[0]-2.0f, y0 = m.pmin[1]-2.0f, z0 = m.pmin[2]-2.0f, x1 = m.pmax[0]+2.0f, y1 = m.pmax[1]+2.0f, z1 = m.pmax[2]+2.0f;
	auto clampi = [](const int v, const int lo, const int hi) { return std::max(lo, std::min(hi, v)); };
	const uint zstart = (uint)clampi((int)z0, 0, (int)Nz-1), hmax = (uint)clampi((int)z1, 0, (int)Nz);
	std::atomic<ulong> solid{0ull};
	parallel_for((ulong)Nx*(ulong)Ny, [&](const ulong a) {
		const uint x = (uint)(a%Nx), y = (uint)(a/Nx);
		const float rx = (float)x, ry = (float)y, rz = (float)zstart;
		if(rx<x0||ry<y0||rx>=x1||ry>=y1) return;
		uint intersections = 0u, check = 0u;
		unsigned short dist[64];
		for(uint i=0u; i<m.n; i++) {
			const float* a0 = &m.p0[3u*i]; const float* a1 = &m.p1[3u*i]; const float* a2 = &m.p2[3u*i];
			const float u[3] = {a1[0]-a0[0], a1[1]-a0[1], a1[2]-a0[2]}, v[3] = {a2[0]-a0[0], a2[1]-a0[1], a2[2]-a0[2]}, w[3] = {rx-a0[0], ry-a0[1], rz-a0[2]};
			const float h[3] = {0.0f*v[2]-1.0f*v[1], 1.0f*v[0]-0.0f*v[2], 0.0f*v[1]-0.0f*v[0]};             // cross(r_direction, v)
			const float q[3] = {w[1]*u[2]-w[2]*u[1], w[2]*u[0]-w[0]*u[2], w[0]*u[1]-w[1]*u[0]};               // cross(w, u)
			const float g = u[0]*h[0]+u[1]*h[1]+u[2]*h[2], f = 1.0f/g, s = f*(w[0]*h[0]+w[1]*h[1]+w[2]*h[2]), t = f*(0.0f*q[0]+0.0f*q[1]+1.0f*q[2]),
				d = f*(v[0]*q[0]+v[1]*q[1]+v[2]*q[2]);
			if(g!=0.0f&&s>=0.0f&&s<1.0f&&t>=0.0f&&s+t<1.0f) {
				if(d>0.0f) { if(intersections<64u&&d<65536.0f) dist[intersections] = (unsigned short)d; intersections++; } else check++;
			}
		}
		const uint ns = std::min(intersections, 64u);
		std::sort(dist, dist+ns);
		bool inside = (intersections%2u)&&(check%2u);
		uint k = (intersections%2u)!=(check%2u);
		const uint h0 = zstart;
		const uint hmesh = h0+(ns>0u ? (uint)dist[std::min(intersections-1u, 63u)] : 0u);
		ulong cnt = 0ull;
		for(uint h=h0; h<hmax; h++) {
			while(k<intersections&&h>h0+(uint)dist[std::min(k, 63u)]) { inside = !inside; k++; }
			inside = inside&&(k<intersections&&h<hmesh);
			if(inside) { const ulong n = (ulong)x+((ulong)y+(ulong)h*Ny)*Nx; flags[n] = (uchar)((flags[n]&~0x03)|TYPE_S); cnt++; }
		}
		solid += cnt;
	});
	return solid.load();
}

