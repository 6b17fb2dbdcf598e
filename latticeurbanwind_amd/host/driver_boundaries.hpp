// driver_boundaries.hpp -- the boundary fill of one case on the host arrays of the LBM object: NWP decks (SurfData samples: patch-driven 2-D
// mapping or sample cloud, temperatures, flux correction; FX/setup.cpp:4931-5632), profile decks (:5914-5995, :6043-6119) and dataset decks
// (:5655-5688).  The builders themselves are bc_builders.hpp.  Part of the deck driver (luw_driver.cpp); included by it only, after driver_state.hpp.
#pragma once

inline void Driver::begin_boundaries() {
	print_section_title("BUILD BOUNDARY CONDITIONS");
	mapped = 0ull; terrain_solid = 0ull; outlet = 0ull;
	ground_xy.clear();
	lattice = HostLattice{}; lattice.Nx = Nx; lattice.Ny = Ny; lattice.Nz = Nz; lattice.flags = flags; lattice.u = u;
	g_progress.emit("interface_interpolation", "Interface interpolation",
		c.nwp_mode ? (surf.has_patch ? "Patch-driven 2D boundary mapping" : c.use_high_order ? "High-order boundary interpolation"
			: "Nearest-sample boundary interpolation")
		: c.profile_mode ? "Applying profile boundary conditions" : "Applying uniform inflow boundary conditions", 0ll, 1ll, true);
}

inline void Driver::report_flux(const FluxReport& fr) const { // FX/fluxcorrection.cpp:180-192
		g_progress.emit("flux_correction", "Flux correction", "avg dU = "+to_string_dd(fr.delta, 3u)+" m/s, net after = "+to_string_dd(fr.net_after, 3u), 1ll,
			1ll, false);
		println("| Flux correction | S_in="+to_string_dd(fr.S_in, 3u)+", S_out="+to_string_dd(fr.S_out, 3u)+", net_before="+to_string_dd(fr.net_before, 3u)
			+" |");
		println("| Flux correction | avg_dU="+to_string_dd(fr.delta, 3u)+" m/s, corrected="+to_string_u(fr.corrected)+", net_after="
			+to_string_dd(fr.net_after, 3u)+" |");
		println("| Flux correction | per-face dU: Xn="+to_string_dd(fr.face_avg[0], 3u)+", Xp="+to_string_dd(fr.face_avg[1], 3u)+", Yn="
			+to_string_dd(fr.face_avg[2], 3u)+", Yp="+to_string_dd(fr.face_avg[3], 3u)+", Zp="+to_string_dd(fr.face_avg[4], 3u)+" m/s |");
}

inline void Driver::fill_nwp_boundaries() { // FX/setup.cpp:4931-5632
	HostLattice HL = lattice;
	const V3 org = HL.position(0u, 0u, 0u);
	std::vector<SurfSample> smp; smp.reserve(surf.rows.size()); // SI -> lattice units (:3963-3979), then shifted to cell-centre coordinates (:4940-4946)
	for(const SurfSample& r : surf.rows) {
		SurfSample q; q.patch = r.patch; q.T = use_temperature_bc ? units.T(r.T) : 1.0f;
		q.p.x = units.x(r.p.x); q.p.y = units.x(r.p.y); q.p.z = units.x(r.p.z);
		q.u.x = r.u.x*u_scale; q.u.y = r.u.y*u_scale; q.u.z = r.u.z*u_scale;
		q.p.x += org.x; q.p.y += org.y; q.p.z += org.z;
		smp.push_back(q);
	}
	const float z0_lbmu = org.z;
	println("| CDF data loaded | "+alignl(57u, to_string_u(surf.rows.size()))+" |");
	std::function<V3(uint, uint, uint)> downstream_fill;
	SampleCloud cloud; std::unique_ptr<KnnSurfaceInterpolator> knn;
	std::vector<PatchField2D> face_fields(6);
	if(surf.has_patch) { // patch-driven 2-D mapping, :5120-5267
		ulong counts[6] = {0, 0, 0, 0, 0, 0};
		for(const SurfSample& q : smp) if(q.patch>=0&&q.patch<=5) counts[q.patch]++;
		println("| Patch samples   | "+alignr(8u, string(patch_name(PATCH_BOTTOM)))+" = "+alignl(47u, to_string_u(counts[PATCH_BOTTOM]))+" |");
		for(int pt=PATCH_TOP; pt<=PATCH_EAST; ++pt) {
			face_fields[(size_t)pt].build(smp, pt, [](const SurfSample& q) { return q.u; }, V3{});
			println("|                 | "+alignr(8u, string(patch_name(pt)))+" = "+alignl(47u, to_string_u(counts[pt]))+" |");
		}
		PatchField2D ground; V3 gdef; gdef.x = z0_lbmu;
		ground.build(smp, PATCH_BOTTOM, [](const SurfSample& q) { V3 v; v.x = q.p.z; return v; }, gdef);
		const PatchBcCounts pc = apply_patch_boundaries(HL, face_fields, ground, case_bc, c.downstream_open_face, side_ref_z_cap);
		if(pc.terrain_clipped>0ull)
			println("| Terrain clip    | below-terrain cells forced to solid: "+to_string_u(pc.terrain_clipped)+"                    |");
		println("| Velocity BC     | patch-driven 2D mapping: "+to_string_u(pc.mapped)+" cells                 |");
		if(pc.grounded>0ull) println("|                 | underground no-slip cells: "+to_string_u(pc.grounded)+"                     |");
		if(pc.below_support>0ull) println("|                 | side cells below terrain support -> solid: "+to_string_u(pc.below_support)+"     |");
		if(pc.outlet>0ull) println("|                 | downstream outlet cells: "+to_string_u(pc.outlet)+" (no fixed velocity)        |");
		if(pc.missing>0ull) println("|                 | WARNING: missing patch samples for "+to_string_u(pc.missing)+" cells         |");
		mapped = pc.mapped; outlet = pc.outlet; terrain_solid = pc.grounded+pc.below_support+pc.terrain_clipped;
		const int dp = downstream_to_patch(case_bc);
		downstream_fill = [&face_fields, HL, dp](const uint x, const uint y, const uint z) -> V3 {
			if(dp<PATCH_TOP||dp>PATCH_EAST||!face_fields[(size_t)dp].has_samples()) return V3{};
			float a, b; if(!patch_plane_coords(dp, HL.position(x, y, z), a, b)) return V3{};
			return face_fields[(size_t)dp].eval(a, b);
		};
	} else {
		cloud.P.reserve(smp.size()); cloud.U.reserve(smp.size());
		for(const SurfSample& q : smp) { cloud.P.push_back(q.p); cloud.U.push_back(q.u); }
		std::function<V3(const V3&)> inlet;
		if(c.use_high_order) { // :5354-5359, FX/interpolation_hd.cpp
			knn.reset(new KnnSurfaceInterpolator(cloud));
			const float z_base = units.x(c.z_si_offset)+z0_lbmu;
			const KnnSurfaceInterpolator* k = knn.get();
			inlet = [k, z_base](const V3& p) -> V3 { return p.z<z_base ? V3{} : k->eval(p); };
			println("| using high order surface based inlet interpolator (HD)");
		} else { // :5555-5558, FX/interpolation.cpp
			const float z_off = units.x(c.z_si_offset);
			const SampleCloud* cl = &cloud;
			inlet = [cl, z0_lbmu, z_off](const V3& p) -> V3 { return p.z<z0_lbmu+z_off ? V3{} : nearest_sample_velocity(*cl, p); };
		}
		println("| Threads used for BC connection: "+to_string_u(bc_worker_threads())+"                                 |");
		mapped = apply_cloud_boundaries(HL, case_bc, c.downstream_open_face, side_ref_z_cap, inlet);
		downstream_fill = [inlet, HL](const uint x, const uint y, const uint z) -> V3 { return inlet(HL.position(x, y, z)); };
	}
	auto temperature_summary = [&](const string& tag) { // FX/setup.cpp:5075-5117
		const TemperatureSummary ts = summarize_temperature(HL, Tcell);
		println("| Temperature BC  | summary ["+tag+"]: TYPE_T total="+to_string_u(ts.total)+", solid="+to_string_u(ts.solid)+", fluid="+to_string_u(ts.fluid)
			+"            |");
		if(ts.solid>0ull)
			println("| Temperature BC  | solid TYPE_T range SI: "+fmtf(units.si_T(ts.smin))+" .. "+fmtf(units.si_T(ts.smax))+" K                      |");
		if(ts.fluid>0ull)
			println("| Temperature BC  | fluid TYPE_T range SI: "+fmtf(units.si_T(ts.fmin))+" .. "+fmtf(units.si_T(ts.fmax))+" K                      |");
		if(ts.invalid>0ull) println("| Temperature BC  | WARNING: non-finite TYPE_T cells = "+to_string_u(ts.invalid)+"                         |");
	};
	string t_tag;
	if(use_temperature_bc) {
		TemperatureCounts tc;
		SampleCloud tcloud; std::unique_ptr<KnnSurfaceInterpolator> tknn;
		if(surf.has_patch) { // :4986-5012, :5268-5311
			t_tag = "patch-2d";
			std::vector<PatchField2D> tfields(6);
			V3 tdef; tdef.x = 1.0f;
			for(int pt=PATCH_TOP; pt<=PATCH_EAST; ++pt) {
				tfields[(size_t)pt].build(smp, pt, [](const SurfSample& q) { V3 v; v.x = q.T; return v; }, tdef);
				ulong cntp = 0ull; float mn = +FLT_MAX, mx = -FLT_MAX;
				for(const SurfSample& q : smp) if(q.patch==pt) { cntp++; mn = fminf(mn, q.T); mx = fmaxf(mx, q.T); }
				if(cntp>0ull) println("| T patch         | "+string(patch_name(pt))+": n="+to_string_u(cntp)+", SI "+fmtf(units.si_T(mn))+" .. "
					+fmtf(units.si_T(mx))+" K                    |");
				else println("| T patch         | "+string(patch_name(pt))+": n=0                                           |");
			}
			apply_patch_temperature(HL, Tcell, tfields, case_bc, c.downstream_open_face, T_bc_min, T_bc_max, tc);
			println("| Temperature BC  | patch-driven 2D mapping: "+to_string_u(tc.mapped)+" cells              |");
			if(tc.missing>0ull) println("|                 | WARNING: missing patch samples for "+to_string_u(tc.missing)+" cells      |");
		} else {
			tcloud.P.reserve(smp.size()); tcloud.U.reserve(smp.size());
			for(const SurfSample& q : smp) { tcloud.P.push_back(q.p); V3 v; v.x = q.T; tcloud.U.push_back(v); }
			if(c.use_high_order) {
				t_tag = "high-order";
				tknn.reset(new KnnSurfaceInterpolator(tcloud));
				const KnnSurfaceInterpolator* k = tknn.get();
				apply_cloud_temperature(HL, Tcell, case_bc, c.downstream_open_face, true, units.x(c.z_si_offset)+z0_lbmu, T_bc_min, T_bc_max, [k](const V3& p) {
					return k->eval(p).x;
				}, tc);
				println("| Temperature BC  | per-face interpolation done on 5 boundary surfaces        |");
				println("| Temperature BC  | mapped "+to_string_u(tc.mapped)+" cells (high-order)      |");
			} else {
				t_tag = "low-order";
				const SampleCloud* cl = &tcloud;
				apply_cloud_temperature(HL, Tcell, case_bc, c.downstream_open_face, false, z0_lbmu+units.x(c.z_si_offset), T_bc_min, T_bc_max,
					[cl](const V3& p) {
					return nearest_sample_velocity(*cl, p).x;
				}, tc);
				println("| Temperature BC  | mapped "+to_string_u(tc.mapped)+" cells (low-order)                     |");
			}
		}
		if(surf.has_patch) { // ground temperature plane from patch 0, :5019-5072 (in every boundary mode when the CSV carries patches)
			std::vector<float> gx, gy, gt;
			for(const SurfSample& q : smp) if(q.patch==PATCH_BOTTOM) { gx.push_back(q.p.x); gy.push_back(q.p.y); gt.push_back(q.T); }
			GroundPlane2D tplane; tplane.build(gx, gy, gt, 1.0f);
			if(tplane.has_samples()) {
				println("| Ground T plane  | enabled from patch=0 ("+to_string_u(gt.size())+" samples, grid "+to_string_u(tplane.nx())+"x"
					+to_string_u(tplane.ny())+", mode="+(tplane.structured() ? string("2D bilinear") : string("2D nearest"))+") |");
				apply_ground_temperature(HL, Tcell, tplane, T_bc_min, T_bc_max, tc);
				println("| Ground T plane  | mapped "+to_string_u(tc.ground_cells)+" solid cells, unique (x,y)="+to_string_u(tc.ground_columns)+" ["+t_tag
					+"]                                |");
				if(tc.ground_cells==0ull) println("| Ground T plane  | WARNING: no solid cells were found                          |");
			} else println("| Ground T plane  | patch column detected, but no patch=0 samples found        |");
		}
		temperature_summary(t_tag);
	}
	g_progress.emit("interface_interpolation", "Interface interpolation", "Boundary conditions completed", 1ll, 1ll, false);
	print_kv_row("Boundary init", "complete. Time: ["+now_str()+"]");
	if(c.flux_correction) {
		print_kv_row("Flux correction", "starting. Time: ["+now_str()+"]");
		g_progress.emit("flux_correction", "Flux correction", "Balancing boundary mass flux", 0ll, 1ll, true);
		report_flux(apply_flux_correction(HL, case_bc, downstream_fill));
		if(use_temperature_bc) temperature_summary(t_tag+"/post-flux");
	} else print_kv_row("Flux correction", "skipped. Set flux_correction=true to enable");
}

inline void Driver::fill_profile_boundaries() { // FX/setup.cpp:5914-5995,6043-6078
	if(use_dem_ground) { // per-column terrain height, cells under it become solid
		const float zmin = pos_z_of(0u), zmax = pos_z_of(Nz-1u);
		ground_xy.assign((size_t)Nx*Ny, flat_ground);
		parallel_for((ulong)Nx*Ny, [&](const ulong id) {
			const uint x = (uint)(id%Nx), y = (uint)(id/Nx);
			float zg = ground_plane.eval((float)x-0.5f*(float)Nx+0.5f, (float)y-0.5f*(float)Ny+0.5f);
			if(!std::isfinite(zg)) zg = flat_ground;
			ground_xy[id] = fminf(fmaxf(zg, zmin), zmax);
		});
		float gmin = +FLT_MAX, gmax = -FLT_MAX;
		for(const float zg : ground_xy) { gmin = fminf(gmin, zg); gmax = fmaxf(gmax, zg); }
		println("| Terrain ground  | mapped z(SI) range "+to_string_fd(units.si_x(gmin-origin_z), 3u)+" .. "+to_string_fd(units.si_x(gmax-origin_z), 3u)
			+" m                     |");
		std::atomic<ulong> clipped{0ull};
		parallel_for(N, [&](const ulong n) {
			if((flags[n]&TYPE_S)!=0u) return;
			const ulong t = n%((ulong)Nx*Ny); const uint z = (uint)(n/((ulong)Nx*Ny));
			if(pos_z_of(z)<ground_xy[t]) { flags[n] = TYPE_S; u[n] = u[N+n] = u[2ull*N+n] = 0.0f; clipped++; }
		});
		if(clipped.load()>0ull) println("| Terrain clip    | below-terrain cells forced to solid: "+to_string_u(clipped.load())+"                    |");
	}
	parallel_for(N, [&](const ulong n) {
		const uint z = (uint)(n/((ulong)Nx*Ny));
		if((flags[n]&TYPE_S)!=0u) { u[n] = u[N+n] = u[2ull*N+n] = 0.0f; return; }
		const float um = profile_speed(pos_z_of(z), ground_at(n%((ulong)Nx*Ny)));
		u[n] = dir_x*um; u[N+n] = dir_y*um; u[2ull*N+n] = 0.0f;
	});
	parallel_for(N, [&](const ulong n) {
		const ulong t = n%((ulong)Nx*Ny); const uint x = (uint)(t%Nx), y = (uint)(t/Nx), z = (uint)(n/((ulong)Nx*Ny));
		if(z==0u) { flags[n] = TYPE_S; u[n] = u[N+n] = u[2ull*N+n] = 0.0f; return; }
		if(!(x==0u||x==Nx-1u||y==0u||y==Ny-1u||z==Nz-1u)) return;
		if((flags[n]&TYPE_S)!=0u) return;
		const float pz = pos_z_of(z);
		const float ground_z = ground_at(t);
		if(pz<=ground_z) { flags[n] = TYPE_S; u[n] = u[N+n] = u[2ull*N+n] = 0.0f; terrain_solid++; return; }
		flags[n] = (uchar)(flags[n]|TYPE_E);
		if(c.downstream_open_face&&is_downstream(x, y)) { outlet++; return; }
		float pze = pz;
		const bool side = x==0u||x==Nx-1u||y==0u||y==Ny-1u;
		if(side&&side_ref_z_cap>=0&&(int)z>side_ref_z_cap) pze = pos_z_of((uint)side_ref_z_cap);
		const float um = profile_speed(pze, ground_z);
		u[n] = dir_x*um; u[N+n] = dir_y*um; u[2ull*N+n] = 0.0f;
		mapped++;
	});
	println("| Velocity BC     | profile boundaries mapped: "+to_string_u(mapped.load())+" cells                |");
	if(outlet.load()>0ull) println("|                 | downstream outlet cells: "+to_string_u(outlet.load())+" (no fixed velocity)        |");
	if(terrain_solid.load()>0ull)
		println("|                 | boundary cells below local terrain -> solid: "+to_string_u(terrain_solid.load())+"                     |");
}

inline void Driver::profile_flux_correction() { // FX/setup.cpp:6087-6119
	HostLattice HL = lattice;
	if(c.flux_correction) {
		print_kv_row("Flux correction", "starting. Time: ["+now_str()+"]");
		g_progress.emit("flux_correction", "Flux correction", "Balancing boundary mass flux", 0ll, 1ll, true);
		report_flux(apply_flux_correction(HL, case_bc, [&](const uint x, const uint y, const uint z) -> V3 {
			float pze = pos_z_of(z);
			if((x==0u||x==Nx-1u||y==0u||y==Ny-1u)&&side_ref_z_cap>=0&&(int)z>side_ref_z_cap) pze = pos_z_of((uint)side_ref_z_cap);
			const float um = profile_speed(pze, ground_at((ulong)y*Nx+x));
			V3 v; v.x = dir_x*um; v.y = dir_y*um; v.z = 0.0f; return v;
		}));
	} else print_kv_row("Flux correction", "skipped. Set flux_correction=true to enable");
}

inline void Driver::fill_dataset_boundaries() { // FX/setup.cpp:5655-5688
	for(ulong n=0ull; n<N; n++) { u[n] = uin[0]; u[N+n] = uin[1]; u[2ull*N+n] = uin[2]; }
	const bool has_ground = Nz>1u;
	for(ulong n=0ull; n<N; n++) {
		const ulong t = n%((ulong)Nx*Ny); const uint x = (uint)(t%Nx), y = (uint)(t/Nx), z = (uint)(n/((ulong)Nx*Ny));
		if(has_ground&&z==0u) { flags[n] = TYPE_S; continue; }
		if(x==0u||x==Nx-1u||y==0u||y==Ny-1u||(has_ground&&z==Nz-1u)) {
			flags[n] = TYPE_E;
			if(c.downstream_open_face&&is_downstream(x, y)) continue;
			u[n] = uin[0]; u[N+n] = uin[1]; u[2ull*N+n] = uin[2];
		}
	}
}
