// driver_output.hpp -- what a case leaves behind: the final raw VTKs (write_final_transient, FX/setup.cpp:4762-4776), transform.info (:4778-4798),
// the _avg VTK with its derived fields (finalize_avg + write_avg_vtk, :4693-4717, :2513-2683) produced by the devices or, as the cross-check
// path of the tests (LUW_HOST_VTK=1), through the host, and the probe CSVs (:4718-4760).  Part of the deck driver (luw_driver.cpp); included
// by it only, after driver_state.hpp.
#pragma once

inline void Driver::write_final_fields() { // write_final_transient, FX/setup.cpp:4762-4776
	LBM& lbm = *lbm_p;
	const ulong t = lbm.get_t();
	std::vector<string> saved;
	const string fn = default_filename(vtk_dir, "u", t), fr = default_filename(vtk_dir, "rho", t), ft = default_filename(vtk_dir, "T", t);
	if(host_vtk_path()) {
		if(last_u_vtk_t!=t) { lbm.u.read_from_device(); write_field_vtk(fn, geom, lbm.u.data<float>(), 3u, units.si_u(1.0f)); }
		lbm.rho.read_from_device(); write_field_vtk(fr, geom, lbm.rho.data<float>(), 1u, units.si_rho(1.0f));
		if(use_temperature_bc) { lbm.T.read_from_device(); write_field_vtk(ft, geom, lbm.T.data<float>(), 1u, units.unit_K, units.unit_K_offset, true); }
	} else {
		if(last_u_vtk_t!=t) write_device_field_vtk(lbm, fn, geom, LUW_EXPORT_U, 3u, units.si_u(1.0f));
		write_device_field_vtk(lbm, fr, geom, LUW_EXPORT_RHO, 1u, units.si_rho(1.0f));
		if(use_temperature_bc) write_device_field_vtk(lbm, ft, geom, LUW_EXPORT_T, 1u, units.unit_K, units.unit_K_offset, true);
	}
	if(last_u_vtk_t!=t) saved.push_back(fn);
	saved.push_back(fr);
	if(use_temperature_bc) saved.push_back(ft);
	bool first = true; for(const string& f : saved) { print_kv_row((first&&last_u_vtk_t!=t) ? "VTK file" : "", f+" saved"); first = false; }
	g_progress.emit("save", "Saving results", saved.size()==1u ? saved.back() : to_string_u(saved.size())+" files saved; last: "+saved.back(),
		(long long)saved.size(), (long long)saved.size(), false);
}

inline void Driver::write_transform_info() const { // maybe_write_transform_info, FX/setup.cpp:4778-4798
	println("| Writing transform.info...                                                  |");
	const string info_path = c.parent+"/proj_temp/transform.info";
	std::ofstream info(info_path);
	if(info.is_open()) {
		const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u);
		info << "dt = " << std::fixed << std::setprecision(10) << dt_si << "s\n";
		info.close();
		println("| Successfully wrote "+info_path+" |");
	}
	else println("ERROR: Could not open "+info_path+" for writing.");
}

// finalize_avg + write_avg_vtk (FX/setup.cpp:4693-4717,2513-2683) with the devices producing every section
inline void Driver::write_avg_vtk_from_devices() {
	LBM& lbm = *lbm_p;
	const uint64_t avg_count = lbm.stats_count();
	if(avg_count>0ull) {
		const string fn = default_filename(results_vtk_dir, vtk_prefix+c.datetime+"_avg", lbm.get_t());
		VtkFile f(fn);
		f.text(vtk_header(fn, geom));
		const float u_factor = units.si_u(1.0f), rho_factor = units.si_rho(1.0f);
		auto section = [&](const string& name, const int source, const uint comps, luw_export_params prm) {
			f.text("SCALARS "+name+" float "+to_string_u(comps)+"\nLOOKUP_TABLE default\n");
			prm.u_factor = u_factor; prm.grid_dx = fmaxf(geom.spacing, 1.0e-12f); prm.tls_cap = (float)std::max(std::max(Nx, Ny), Nz_out)*prm.grid_dx;
			prm.want_tke = c.out_tke ? 1 : 0; prm.want_ti = c.out_ti ? 1 : 0; prm.want_tls = c.out_tls ? 1 : 0;
			f.payload(lbm, source, prm, geom, comps);
		};
		section("u_avg", LUW_EXPORT_AVG_U, 3u, export_params(u_factor));
		section("rho_avg", LUW_EXPORT_AVG_RHO, 1u, export_params(rho_factor));
		// Kelvin: FX/setup.cpp:2526-2528,2580-2582
		if(use_temperature_bc) section("T_avg", LUW_EXPORT_AVG_T, 1u, export_params(units.si_dT(1.0f), units.si_T(0.0f)));
		phase_mark("  u_avg, rho_avg written");
		section("fluid", LUW_EXPORT_FLUID, 1u, export_params(1.0f));
		if(c.out_tke) section("tke", LUW_EXPORT_TKE, 1u, export_params(u_factor*u_factor));
		if(c.out_ti) section("TI", LUW_EXPORT_TI, 1u, export_params(1.0f));
		if(c.out_tls) section("TLS", LUW_EXPORT_TLS, 1u, export_params(1.0f));
		print_kv_row("VTK file", fn+" saved");
		g_progress.emit("save", "Saving results", fn, 1ll, 1ll, false);
		print_kv_row("Avg samples", to_string_u(avg_count));
	}
}

inline void Driver::write_avg_vtk_through_host() { // the same file through the host (cross-check path)
	LBM& lbm = *lbm_p;
	// 7 floats per cell, every one of them overwritten by the download: no value-initialisation (a 1.4 GB memset at 50 M cells)
	std::unique_ptr<float[]> stats_mem(new float[7ull*N]); uint64_t avg_count = 0ull;
	float* const avg_u = stats_mem.get(); float* const avg_rho = avg_u+3ull*N; float* const m2u = avg_rho+N; float* const m2v = m2u+N; float* const m2w = m2v+N;
	std::vector<float> avg_T; if(use_temperature_bc) avg_T.resize(N);
	lbm.stats_download(avg_u, avg_rho, m2u, m2v, m2w, use_temperature_bc ? avg_T.data() : nullptr, &avg_count);
	phase_mark("  statistics download");
	if(avg_count>0ull) {
		const string fn = default_filename(results_vtk_dir, vtk_prefix+c.datetime+"_avg", lbm.get_t());
		std::filesystem::create_directories(std::filesystem::path(fn).parent_path());
		std::ofstream file(fn, std::ios::out|std::ios::binary);
		const string header = vtk_header(fn, geom); file.write(header.c_str(), (std::streamsize)header.length());
		const ulong points = (ulong)Nx*Ny*Nz_out;
		const float u_factor = units.si_u(1.0f), rho_factor = units.si_rho(1.0f), spacing = geom.spacing;
		// one conversion buffer for all fields, fully written before each use
		std::unique_ptr<float[]> conv(new float[3ull*points]); float* const buf = conv.get();
		auto write_field = [&](const string& name, const float* data, const uint comps, const float factor) {
			const string fh = "SCALARS "+name+" float "+to_string_u(comps)+"\nLOOKUP_TABLE default\n"; file.write(fh.c_str(), (std::streamsize)fh.length());
			parallel_for(points, [&](const ulong i) { for(uint d=0u; d<comps; d++) buf[i*comps+d] = reverse_bytes(data[i*comps+d]*factor+0.0f); });
			file.write((const char*)buf, (std::streamsize)(points*comps*4ull));
		};
		write_field("u_avg", avg_u, 3u, u_factor);
		write_field("rho_avg", avg_rho, 1u, rho_factor);
		if(use_temperature_bc) { // T_avg in Kelvin: factor si_dT(1), offset si_T(0), FX/setup.cpp:2526-2528,2580-2582
								const string fh = "SCALARS T_avg float 1\nLOOKUP_TABLE default\n"; file.write(fh.c_str(), (std::streamsize)fh.length());
			const float tf = units.si_dT(1.0f), to = units.si_T(0.0f);
			parallel_for(points, [&](const ulong i) { buf[i] = reverse_bytes(avg_T[i]*tf+to); });
			file.write((const char*)buf, (std::streamsize)(points*4ull));
		}
		// derived fields: every element is set by the loop below (defaults first), so the arrays start uninitialised
		std::unique_ptr<float[]> derived(new float[4ull*points]);
		float* const fluid = derived.get(); float* const tke = fluid+points; float* const ti = tke+points; float* const tls = ti+points;
		const bool has_m2 = avg_count>1ull; const float inv_n = has_m2 ? 1.0f/(float)avg_count : 0.0f;
		const float grid_dx = fmaxf(spacing, 1.0e-12f); const ulong plane = (ulong)Nx*Ny;
		const float tls_cap = (float)std::max(std::max(Nx, Ny), Nz_out)*grid_dx;
		const uchar* fl = lbm.flags.data<uchar>();
		auto su = [&](const ulong idx, const uint comp) { return avg_u[3ull*idx+comp]*u_factor; };
		parallel_for(points, [&](const ulong n) {
			const bool solid = (fl[n]&TYPE_S)!=0u;
			fluid[n] = solid ? 0.0f : 1.0f;
			tke[n] = 0.0f; ti[n] = 0.0f; tls[n] = 0.0f;
			if(!has_m2||solid) return;
			if(!(c.out_tke||c.out_ti||c.out_tls)) return;
			const float var_u = fmaxf(m2u[n]*inv_n, 0.0f), var_v = fmaxf(m2v[n]*inv_n, 0.0f), var_w = fmaxf(m2w[n]*inv_n, 0.0f), var_sum = var_u+var_v+var_w;
			if(c.out_tke) tke[n] = 0.5f*var_sum;
			if(c.out_ti) {
				const ulong i3 = 3ull*n;
				const float umag = sqrtf(avg_u[i3]*avg_u[i3]+avg_u[i3+1ull]*avg_u[i3+1ull]+avg_u[i3+2ull]*avg_u[i3+2ull]);
				if(umag>1.0e-9f&&var_sum>0.0f) ti[n] = sqrtf(var_sum*(1.0f/3.0f))/umag;
			}
			if(!c.out_tls) return;
			const ulong z = n/plane, rem = n-z*plane, y = rem/Nx, x = rem-y*Nx;
			const ulong xm = x>0ull ? x-1ull : x, xp = x+1ull<Nx ? x+1ull : x, ym = y>0ull ? y-1ull : y, yp = y+1ull<Ny ? y+1ull : y, zm = z>0ull ? z-1ull : z,
				zp = z+1ull<Nz_out ? z+1ull : z;
			const ulong ixm = xm+(y+z*Ny)*Nx, ixp = xp+(y+z*Ny)*Nx, iym = x+(ym+z*Ny)*Nx, iyp = x+(yp+z*Ny)*Nx, izm = x+(y+zm*Ny)*Nx, izp = x+(y+zp*Ny)*Nx;
			const float idx_ = xp>xm ? 1.0f/((float)(xp-xm)*grid_dx) : 0.0f, idy = yp>ym ? 1.0f/((float)(yp-ym)*grid_dx) : 0.0f, idz = zp>zm
				? 1.0f/((float)(zp-zm)*grid_dx) : 0.0f;
			const float duxdx = (su(ixp, 0u)-su(ixm, 0u))*idx_, duydx = (su(ixp, 1u)-su(ixm, 1u))*idx_, duzdx = (su(ixp, 2u)-su(ixm, 2u))*idx_;
			const float duxdy = (su(iyp, 0u)-su(iym, 0u))*idy, duydy = (su(iyp, 1u)-su(iym, 1u))*idy, duzdy = (su(iyp, 2u)-su(iym, 2u))*idy;
			const float duxdz = (su(izp, 0u)-su(izm, 0u))*idz, duydz = (su(izp, 1u)-su(izm, 1u))*idz, duzdz = (su(izp, 2u)-su(izm, 2u))*idz;
			const float Sxy = 0.5f*(duxdy+duydx), Sxz = 0.5f*(duxdz+duzdx), Syz = 0.5f*(duydz+duzdy);
			const float S_mag = sqrtf(fmaxf(0.0f, 2.0f*(duxdx*duxdx+duydy*duydy+duzdz*duzdz+2.0f*(Sxy*Sxy+Sxz*Sxz+Syz*Syz))));
			const float k_local = 0.5f*var_sum*(u_factor*u_factor);
			const float tls_local = (S_mag>1.0e-10f&&k_local>0.0f) ? sqrtf(k_local)/S_mag : 0.0f;
			tls[n] = fminf(fmaxf(tls_local, 0.0f), tls_cap);
		});
		phase_mark("  u_avg, rho_avg written; tke/TI/TLS computed");
		write_field("fluid", fluid, 1u, 1.0f);
		if(c.out_tke) write_field("tke", tke, 1u, u_factor*u_factor);
		if(c.out_ti) write_field("TI", ti, 1u, 1.0f);
		if(c.out_tls) write_field("TLS", tls, 1u, 1.0f);
		print_kv_row("VTK file", fn+" saved");
		g_progress.emit("save", "Saving results", fn, 1ll, 1ll, false);
		print_kv_row("Avg samples", to_string_u(avg_count));
	}
}

inline void Driver::write_probe_files() const { // FX/setup.cpp:4718-4760
	std::filesystem::create_directories(c.parent+"/RESULTS");
	ulong written = 0ull;
	for(const ProbeColumn& pc : probes) {
		const string path = c.parent+"/RESULTS/"+pc.stem+".csv";
		if(write_probe_csv(path, pc)) written++;
		else print_kv_row("Probe output", "failed to open "+path);
	}
	print_kv_row("Probe files", to_string_u(written)+" CSV saved to RESULTS");
	// FX/setup.cpp:4754-4759
	g_progress.emit("save", "Saving results", to_string_u(written)+" probe CSV file(s) saved to RESULTS", (long long)written, (long long)written, false);
}
