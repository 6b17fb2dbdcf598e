// luw_export.hpp -- VTK payloads straight from the devices (Memory_Container::write_vtk, FX/lbm.hpp:307-356; write_avg_vtk, FX/setup.cpp:2513-2683).
// Device + host code of libluw_core.so; included by luw_core.hip only, after luw_group.hpp.
//
// The reference downloads every field in full, stitches the domains' buffers into one global index space on the host, converts SoA -> AoS,
// scales to SI units, swaps to big-endian and writes.  Here each domain's device does all of that for its own cells -- one kernel per z slab writes
// big-endian SI values in the file's own order (x fastest, then y, then z; components interleaved) -- the slab goes to pinned host memory with
// one strided copy per plane and domain, and a writer thread puts it into the file at its offset (pwrite) while the next slab is produced.
// The derived fields of the averaged output (fluid mask, tke, TI, TLS) are computed in that kernel from the statistics the devices hold; nothing
// lattice-sized is gathered on the host.  Same bytes as the host path (the arithmetic is the host's, one rounding per operation).
#pragma once
#include <unistd.h>
#include <condition_variable>
#include <mutex>

struct ExportSrc {          // device arrays of one domain (pitch Px, plane stride Np); unused ones null
	const float* a; const float* m2; const uint8_t* flags;
	float factor, offset; int affine;
	float u_factor, grid_dx, tls_cap, inv_n; int has_m2, want_tke, want_ti, want_tls;
	uint32_t gNx, gNy, gNz_out;    // global lattice as written (clamps of the TLS stencil)
};
__device__ __forceinline__ uint32_t to_big_endian(const float v) { return __builtin_bswap32(__float_as_uint(v)); }
// lane = one owned cell of plane z0 + blockIdx.z; dst[((pz*oy + y')*ox + x')*comps + c]
template<int SRC> __global__ __launch_bounds__(256) void k_export_slab(const KParams p, const ExportSrc e, const uint32_t z0, uint32_t* __restrict__ dst) {
	const uint32_t H0 = p.halo_x, H1 = p.halo_y;
	const uint32_t ox = p.Nx-2u*H0, oy = p.Ny-2u*H1;
	const uint32_t xo = blockIdx.x*blockDim.x+threadIdx.x;
	if(xo>=ox) return;
	const uint32_t x = xo+H0, y = blockIdx.y+H1, z = z0+blockIdx.z;
	const uint32_t n = x+(y+z*p.Ny)*p.Px;
	const size_t Np = p.Np;
	const size_t o = ((size_t)blockIdx.z*oy+blockIdx.y)*ox+xo;
	if constexpr(SRC==LUW_EXPORT_U||SRC==LUW_EXPORT_AVG_U) {
		#pragma unroll
		for(int c=0; c<3; c++) { const float v = e.a[c*Np+n]; dst[3u*o+c] = to_big_endian(SRC==LUW_EXPORT_U ? e.factor*v : v*e.factor+0.0f); }
	} else if constexpr(SRC==LUW_EXPORT_RHO) dst[o] = to_big_endian(e.factor*e.a[n]);
	else if constexpr(SRC==LUW_EXPORT_T) dst[o] = to_big_endian(e.affine ? e.a[n]*e.factor+e.offset : e.factor*e.a[n]);
	else if constexpr(SRC==LUW_EXPORT_AVG_RHO||SRC==LUW_EXPORT_AVG_T) dst[o] = to_big_endian(e.a[n]*e.factor+e.offset);
	else { // derived fields of write_avg_vtk: 0 on solid cells, without second moments (one sample) and when the deck did not ask for them
		const bool solid = (e.flags[n]&TYPE_S)!=0u;
		float v = 0.0f;
		if constexpr(SRC==LUW_EXPORT_FLUID) v = solid ? 0.0f : 1.0f;
		else if(e.has_m2&&!solid&&(e.want_tke||e.want_ti||e.want_tls)) {
			const float var_u = fmaxf(e.m2[n]*e.inv_n, 0.0f), var_v = fmaxf(e.m2[Np+n]*e.inv_n, 0.0f), var_w = fmaxf(e.m2[2u*Np+n]*e.inv_n, 0.0f),
				var_sum = var_u+var_v+var_w;
			if constexpr(SRC==LUW_EXPORT_TKE) { if(e.want_tke) v = 0.5f*var_sum; }
			else if constexpr(SRC==LUW_EXPORT_TI) {
				if(e.want_ti) {
					const float ax = e.a[n], ay = e.a[Np+n], az = e.a[2u*Np+n];
					const float umag = sqrtf(ax*ax+ay*ay+az*az);
					if(umag>1.0e-9f&&var_sum>0.0f) v = sqrtf(var_sum*(1.0f/3.0f))/umag;
				}
			} else if(e.want_tls) { // TLS: central differences of the SI mean velocity, one-sided at the edges of the WRITTEN lattice
				const uint32_t gx = (uint32_t)((int32_t)x+p.Ox), gy = (uint32_t)((int32_t)y+p.Oy), gz = (uint32_t)((int32_t)z+p.Oz);
				const uint32_t xm = gx>0u ? x-1u : x, xp = gx+1u<e.gNx ? x+1u : x, ym = gy>0u ? y-1u : y, yp = gy+1u<e.gNy ? y+1u : y, zm = gz>0u ? z-1u : z,
					zp = gz+1u<e.gNz_out ? z+1u : z;
				const uint32_t ixm = xm+(y+z*p.Ny)*p.Px, ixp = xp+(y+z*p.Ny)*p.Px, iym = x+(ym+z*p.Ny)*p.Px, iyp = x+(yp+z*p.Ny)*p.Px, izm = x+(y+zm*p.Ny)*p.Px,
					izp = x+(y+zp*p.Ny)*p.Px;
				const float idx_ = xp>xm ? 1.0f/((float)(xp-xm)*e.grid_dx) : 0.0f, idy = yp>ym ? 1.0f/((float)(yp-ym)*e.grid_dx) : 0.0f, idz = zp>zm
					? 1.0f/((float)(zp-zm)*e.grid_dx) : 0.0f;
				auto su = [&](const uint32_t i, const uint32_t c) { return e.a[c*Np+i]*e.u_factor; };
				const float duxdx = (su(ixp, 0u)-su(ixm, 0u))*idx_, duydx = (su(ixp, 1u)-su(ixm, 1u))*idx_, duzdx = (su(ixp, 2u)-su(ixm, 2u))*idx_;
				const float duxdy = (su(iyp, 0u)-su(iym, 0u))*idy, duydy = (su(iyp, 1u)-su(iym, 1u))*idy, duzdy = (su(iyp, 2u)-su(iym, 2u))*idy;
				const float duxdz = (su(izp, 0u)-su(izm, 0u))*idz, duydz = (su(izp, 1u)-su(izm, 1u))*idz, duzdz = (su(izp, 2u)-su(izm, 2u))*idz;
				const float Sxy = 0.5f*(duxdy+duydx), Sxz = 0.5f*(duxdz+duzdx), Syz = 0.5f*(duydz+duzdy);
				const float S_mag = sqrtf(fmaxf(0.0f, 2.0f*(duxdx*duxdx+duydy*duydy+duzdz*duzdz+2.0f*(Sxy*Sxy+Sxz*Sxz+Syz*Syz))));
				const float k_local = 0.5f*var_sum*(e.u_factor*e.u_factor);
				const float tls_local = (S_mag>1.0e-10f&&k_local>0.0f) ? sqrtf(k_local)/S_mag : 0.0f;
				v = fminf(fmaxf(tls_local, 0.0f), e.tls_cap);
			}
		}
		dst[o] = to_big_endian(v*e.factor+0.0f);
	}
}
// one cell layer of a float field (comps planes of stride Np) <-> a packed buffer [comps][A]; layer = coordinate `c` along `axis`
template<bool SCATTER> __global__ __launch_bounds__(256) void k_field_layer(const KParams p, const uint32_t axis, const uint32_t c, const uint32_t comps,
	float* __restrict__ field, float* __restrict__ buf) {
	const uint32_t N[3] = { p.Nx, p.Ny, p.Nz };
	const uint32_t a1 = (axis+1u)%3u, a2 = (axis+2u)%3u, A = N[a1]*N[a2];
	const uint32_t i = blockIdx.x*blockDim.x+threadIdx.x;
	if(i>=A) return;
	uint32_t xyz[3]; xyz[axis] = c; xyz[a1] = i%N[a1]; xyz[a2] = i/N[a1];
	const size_t n = xyz[0]+((size_t)xyz[1]+(size_t)xyz[2]*p.Ny)*p.Px;
	for(uint32_t k=0u; k<comps; k++) { if(SCATTER) field[(size_t)k*p.Np+n] = buf[(size_t)k*A+i]; else buf[(size_t)k*A+i] = field[(size_t)k*p.Np+n]; }
}

// halo layers of a float field of every domain <- the neighbours' outermost owned layers (through the host: a one-off before the TLS stencil runs)
static int group_fill_field_halos(luw_group* g, const std::function<float*(luw_solver*)>& field_of, const uint32_t comps) {
	for(int a=0; a<3; a++) {
		if(!g->H[a]) continue;
		const size_t n = g->dom.size();
		std::vector<std::vector<float>> face_p(n), face_m(n);   // outermost owned layer at the + side / - side of every domain
		for(size_t i=0; i<n; i++) {
			GroupDomain& d = g->dom[i];
			GROUP_TRY(group_set_device(d));
			const uint32_t A = (uint32_t)luw_get_area(d.s, (uint32_t)a);
			float* buf = nullptr;
			if(hipMalloc((void**)&buf, (size_t)comps*A*4u)!=hipSuccess) return fail(LUW_ERR_NOMEM, "export: halo staging");
			struct Free { float* p; ~Free() { (void)hipFree(p); } } fr{ buf };
			for(int side=0; side<2; side++) {
				const uint32_t c = side==0 ? d.lN[a]-2u : 1u;
				hipLaunchKernelGGL(k_field_layer<false>, dim3((A+255u)/256u), dim3(256), 0, d.compute, d.s->kp, (uint32_t)a, c, comps, field_of(d.s), buf);
				std::vector<float>& out = side==0 ? face_p[i] : face_m[i];
				out.resize((size_t)comps*A);
				HIP_TRY(hipMemcpyAsync(out.data(), buf, (size_t)comps*A*4u, hipMemcpyDeviceToHost, d.compute));
				HIP_TRY(hipStreamSynchronize(d.compute));
			}
		}
		for(size_t i=0; i<n; i++) {
			GroupDomain& d = g->dom[i];
			GROUP_TRY(group_set_device(d));
			const uint32_t A = (uint32_t)luw_get_area(d.s, (uint32_t)a);
			float* buf = nullptr;
			if(hipMalloc((void**)&buf, (size_t)comps*A*4u)!=hipSuccess) return fail(LUW_ERR_NOMEM, "export: halo staging");
			struct Free { float* p; ~Free() { (void)hipFree(p); } } fr{ buf };
			for(int side=0; side<2; side++) { // my + halo layer <- the + neighbour's lowest owned layer; my - halo layer <- the - neighbour's highest
				const std::vector<float>& in = side==0 ? face_m[d.nbr[a][0]] : face_p[d.nbr[a][1]];
				const uint32_t c = side==0 ? d.lN[a]-1u : 0u;
				HIP_TRY(hipMemcpyAsync(buf, in.data(), (size_t)comps*A*4u, hipMemcpyHostToDevice, d.compute));
				hipLaunchKernelGGL(k_field_layer<true>, dim3((A+255u)/256u), dim3(256), 0, d.compute, d.s->kp, (uint32_t)a, c, comps, field_of(d.s), buf);
				HIP_TRY(hipStreamSynchronize(d.compute));
			}
		}
	}
	return LUW_OK;
}

// writer thread: takes filled slabs in order and pwrite()s them
struct SlabWriter {
	int fd; std::thread th; std::mutex m; std::condition_variable cv;
	struct Job { const char* data; size_t bytes; uint64_t offset; int slot; };
	std::vector<Job> queue; bool done = false; int error = 0; int busy[2] = { 0, 0 };
	explicit SlabWriter(const int fd_) : fd(fd_) {
		th = std::thread([this]() {
			for(;;) {
				Job j;
				{
					std::unique_lock<std::mutex> l(m);
					cv.wait(l, [&] { return done||!queue.empty(); });
					if(queue.empty()) return;
					j = queue.front();
					queue.erase(queue.begin());
				}
				size_t put = 0u;
				while(put<j.bytes) {
					const ssize_t w = pwrite(fd, j.data+put, j.bytes-put, (off_t)(j.offset+put));
					if(w<=0) { std::lock_guard<std::mutex> l(m); error = 1; break; }
					put += (size_t)w;
				}
				{ std::lock_guard<std::mutex> l(m); busy[j.slot] = 0; }
				cv.notify_all();
			}
		});
	}
	void wait_free(const int slot) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return busy[slot]==0; }); }
	void submit(const Job& j) { { std::lock_guard<std::mutex> l(m); busy[j.slot] = 1; queue.push_back(j); } cv.notify_all(); }
	int finish() { { std::lock_guard<std::mutex> l(m); done = true; } cv.notify_all(); if(th.joinable()) th.join(); return error; }
	~SlabWriter() { (void)finish(); }
};

extern "C" int luw_group_export_vtk(luw_group* g, int source, const luw_export_params* prm, uint32_t Nz_write, int fd, uint64_t file_offset) {
	if(!g||!prm||fd<0||source<0||source>LUW_EXPORT_TLS) return fail(LUW_ERR_INVALID, "luw_group_export_vtk: bad argument");
	if(prm->struct_size!=sizeof(luw_export_params)) return fail(LUW_ERR_INVALID, "luw_group_export_vtk: luw_export_params size mismatch (ABI)");
	const uint32_t comps = (source==LUW_EXPORT_U||source==LUW_EXPORT_AVG_U) ? 3u : 1u;
	const uint32_t gNx = g->gN[0], gNy = g->gN[1], nzw = std::min(Nz_write ? Nz_write : g->gN[2], g->gN[2]);
	const bool from_stats = source>=LUW_EXPORT_AVG_U&&source!=LUW_EXPORT_FLUID;
	for(GroupDomain& d : g->dom) {
		if(from_stats&&!d.s->d_avg_u) return fail(LUW_ERR_STATE, "luw_group_export_vtk: no statistics have been accumulated");
		if((source==LUW_EXPORT_T&&!d.s->d_T)||(source==LUW_EXPORT_AVG_T&&!d.s->d_avg_T))
			return fail(LUW_ERR_STATE, "luw_group_export_vtk: the solver has no temperature field");
		// (T is stored with rho and u: by the steps whose fields are looked at, thermal_cell)
		if((source==LUW_EXPORT_U||source==LUW_EXPORT_RHO||source==LUW_EXPORT_T)&&!d.s->fields_current)
			return fail(LUW_ERR_STATE, "luw_group_export_vtk: rho, u, T on the device are stale (the last step did not write fields)");
	}
	if(source==LUW_EXPORT_TLS&&prm->want_tls&&g->dom.size()>1u) // the stencil reads the mean velocity of cells next door
		GROUP_TRY(group_fill_field_halos(g, [](luw_solver* s) { return s->d_avg_u; }, 3u));
	// z slabs of about 192 MiB, two pinned buffers; per domain a device staging buffer of one slab of its own cells
	const size_t plane_bytes = (size_t)gNx*gNy*comps*4u;
	const uint32_t S = (uint32_t)std::max<size_t>(1u, std::min<size_t>(nzw, (192ull<<20)/plane_bytes));
	char* slab[2] = { nullptr, nullptr };
	struct Pinned { char** p; ~Pinned() { (void)hipHostFree(p[0]); (void)hipHostFree(p[1]); } } pinned{ slab };
	for(int k=0; k<2; k++) if(hipHostMalloc((void**)&slab[k], (size_t)S*plane_bytes)!=hipSuccess) {
		slab[k] = nullptr;
		return fail(LUW_ERR_NOMEM, "luw_group_export_vtk: pinned slab");
	}
	std::vector<uint32_t*> stage(g->dom.size(), nullptr);
	struct Stage {
		std::vector<uint32_t*>& v;
		luw_group* g;
		~Stage() { for(size_t i=0; i<v.size(); i++) { (void)hipSetDevice(g->dom[i].device); (void)hipFree(v[i]); } }
	} stage_free{ stage, g };
	for(size_t i=0; i<g->dom.size(); i++) {
		GroupDomain& d = g->dom[i];
		GROUP_TRY(group_set_device(d));
		const size_t ox = d.lN[0]-2u*g->H[0], oy = d.lN[1]-2u*g->H[1];
		if(hipMalloc((void**)&stage[i], (size_t)S*oy*ox*comps*4u)!=hipSuccess) {
			stage[i] = nullptr;
			return fail(LUW_ERR_NOMEM, "luw_group_export_vtk: device staging");
		}
	}
	SlabWriter writer(fd);
	int slot = 0;
	for(uint32_t zs=0u; zs<nzw; zs+=S, slot ^= 1) {
		const uint32_t ze = std::min(zs+S, nzw);
		writer.wait_free(slot);
		for(size_t i=0; i<g->dom.size(); i++) {
			GroupDomain& d = g->dom[i];
			const uint32_t H0 = g->H[0], H1 = g->H[1], H2 = g->H[2];
			const uint32_t ox = d.lN[0]-2u*H0, oy = d.lN[1]-2u*H1, oz = d.lN[2]-2u*H2;
			const uint32_t gz0 = (uint32_t)(d.O[2]+(int32_t)H2), gy0 = (uint32_t)(d.O[1]+(int32_t)H1), gx0 = (uint32_t)(d.O[0]+(int32_t)H0);
			const uint32_t a = std::max(zs, gz0), b = std::min(ze, gz0+oz); // planes of this slab the domain owns
			if(a>=b) continue;
			GROUP_TRY(group_set_device(d));
			luw_solver* s = d.s;
			ExportSrc e{};
			e.factor = prm->factor; e.offset = prm->offset; e.affine = prm->affine;
			e.u_factor = prm->u_factor;
			e.grid_dx = prm->grid_dx;
			e.tls_cap = prm->tls_cap;
			e.want_tke = prm->want_tke;
			e.want_ti = prm->want_ti;
			e.want_tls = prm->want_tls;
			e.has_m2 = s->avg_count>1ull ? 1 : 0; e.inv_n = e.has_m2 ? 1.0f/(float)s->avg_count : 0.0f;
			e.gNx = gNx; e.gNy = gNy; e.gNz_out = nzw; e.flags = s->d_flags; e.m2 = s->d_m2;
			switch(source) {
				case LUW_EXPORT_U: e.a = s->d_u; break; case LUW_EXPORT_RHO: e.a = s->d_rho; break; case LUW_EXPORT_T: e.a = s->d_T; break;
				case LUW_EXPORT_AVG_RHO: e.a = s->d_avg_rho; break; case LUW_EXPORT_AVG_T: e.a = s->d_avg_T; break; default: e.a = s->d_avg_u; break;
			}
			const uint32_t bx = ox>=256u ? 256u : ((ox+63u)/64u)*64u;
			const dim3 grid((ox+bx-1u)/bx, oy, b-a), block(bx);
			const uint32_t z_local = a-gz0+H2;
			#define LUW_EXPORT_CASE(SRC) case SRC: hipLaunchKernelGGL((k_export_slab<SRC>), grid, block, 0, d.compute, s->kp, e, z_local, stage[i]); break
			switch(source) {
				LUW_EXPORT_CASE(LUW_EXPORT_U);
				LUW_EXPORT_CASE(LUW_EXPORT_RHO);
				LUW_EXPORT_CASE(LUW_EXPORT_T);
				LUW_EXPORT_CASE(LUW_EXPORT_AVG_U);
				LUW_EXPORT_CASE(LUW_EXPORT_AVG_RHO);
				LUW_EXPORT_CASE(LUW_EXPORT_AVG_T);
				LUW_EXPORT_CASE(LUW_EXPORT_FLUID);
				LUW_EXPORT_CASE(LUW_EXPORT_TKE);
				LUW_EXPORT_CASE(LUW_EXPORT_TI);
				LUW_EXPORT_CASE(LUW_EXPORT_TLS);
			}
			#undef LUW_EXPORT_CASE
			HIP_TRY(hipGetLastError());
			const size_t row = (size_t)ox*comps*4u, pitch = (size_t)gNx*comps*4u;
			for(uint32_t z=a; z<b; z++) { // one strided copy per plane: rows of this domain into their place in the global plane
				char* dstp = slab[slot]+((size_t)(z-zs)*gNy+gy0)*pitch+(size_t)gx0*comps*4u;
				const char* srcp = (const char*)stage[i]+(size_t)(z-a)*oy*row;
				HIP_TRY(hipMemcpy2DAsync(dstp, pitch, srcp, row, row, oy, hipMemcpyDeviceToHost, d.compute));
			}
		}
		for(GroupDomain& d : g->dom) { GROUP_TRY(group_set_device(d)); HIP_TRY(hipStreamSynchronize(d.compute)); }
		writer.submit(SlabWriter::Job{ slab[slot], (size_t)(ze-zs)*plane_bytes, file_offset+(uint64_t)zs*plane_bytes, slot });
	}
	if(writer.finish()) return fail(LUW_ERR_DEVICE, "luw_group_export_vtk: writing the file failed");
	return LUW_OK;
}
