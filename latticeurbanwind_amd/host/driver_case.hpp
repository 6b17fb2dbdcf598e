// driver_case.hpp -- one (inflow, angle) case of main_setup up to the time loop: case header and per-case units (FX/setup.cpp:5690-5740), the LBM
// object and the voxeliser (:4089, :4935, :5720, :6018), the von-Karman inlet tables (:3762-3799, :417-534), the test dump, the output plan
// and the probe columns (:4269-4395).  Part of the deck driver (luw_driver.cpp); included by it only, after driver_state.hpp.
#pragma once

inline void Driver::run_case(const Case& which) {
	begin_case(which);
	phase_mark("deck, sizing, mesh, profile");
	create_solver_and_voxelize();
	phase_mark("solver create + voxelise");
	begin_boundaries();
	if(c.nwp_mode) fill_nwp_boundaries(); // FX/setup.cpp:4931-5632
	else if(c.profile_mode) fill_profile_boundaries(); // FX/setup.cpp:5914-5995,6043-6078
	else fill_dataset_boundaries(); // FX/setup.cpp:5655-5688
	if(!c.nwp_mode) {
		g_progress.emit("interface_interpolation", "Interface interpolation", c.profile_mode ? "Profile boundary conditions completed"
			: "Boundary conditions completed", 1ll, 1ll, false);
		print_kv_row("Boundary init", "complete. Time: ["+now_str()+"]");
	}
	if(c.profile_mode) profile_flux_correction(); // FX/setup.cpp:6087-6119
	build_vk_inlet();
	if(!c.dump_setup.empty()&&case_index==1u) dump_setup_file();
	plan_outputs();
	resolve_probes();
	if(c.dry_run) return;
	phase_mark("boundary conditions");
	run_solver(); // run_lbm, FX/setup.cpp:4117-4911
	phase_mark("solver loop");
	write_final_fields();
	phase_mark("final raw VTKs");
	if(c.research_output_steps>0u) write_transform_info();
	if(avg_window>0ull) { if(host_vtk_path()) write_avg_vtk_through_host(); else write_avg_vtk_from_devices(); }
	phase_mark("statistics download + avg VTK");
	if(!probes.empty()) write_probe_files();
	print_kv_row("Task finished", "["+now_str()+"]");
}

inline void Driver::begin_case(const Case& which) {
	cs = which; ++case_index;
	lbm_p.reset(); // the previous case's solver goes before the next one is built
	Nx = (lbmN[0]/c.Dx)*c.Dx; Ny = (lbmN[1]/c.Dy)*c.Dy; Nz = (lbmN[2]/c.Dz)*c.Dz;
	N = (ulong)Nx*Ny*Nz;
	const float deg2rad = 3.14159265358979323846f/180.0f, angle_rad = cs.angle_deg*deg2rad;
	dir_x = -sinf(angle_rad); dir_y = -cosf(angle_rad);
	uin[0] = uin[1] = uin[2] = 0.0f;
	if(c.nwp_mode) vtk_prefix = "";
	else if(c.dataset_mode) { // FX/setup.cpp:5690-5740
		si_ref_u = cs.inflow_si; u_scale = lbm_ref_u/si_ref_u;
		units.set_m_kg_s_K((float)Ny, lbm_ref_u, 1.0f, 1.0f, c.si_y, si_ref_u, si_rho, 293.15f);
		lbm_nu = units.nu(si_nu);
		update_coriolis();
		const float speed = cs.inflow_si*u_scale;
		uin[0] = -sinf(angle_rad)*speed; uin[1] = -cosf(angle_rad)*speed;
		dir_x = uin[0]; dir_y = uin[1];
		vtk_prefix = "DG_"+format_tag(cs.inflow_si)+"_"+format_tag(cs.angle_deg)+"_";
		println("|-----------------------------------------------------------------------------|");
		println("| Dataset case    | "+alignr(57u, to_string_u(case_index)+"/"+to_string_u(cases.size()))+" |");
		println("| Inflow / Angle  | "+alignr(57u, format_tag(cs.inflow_si)+" m/s, "+format_tag(cs.angle_deg)+" deg")+" |");
	} else {
		vtk_prefix = cases.size()==1u ? string("") : "ANG_"+format_tag(cs.angle_deg)+"_";
		println("|-----------------------------------------------------------------------------|");
		println("| Profile case    | "
			+alignr(57u, to_string_u(case_index)+"/"+to_string_u(cases.size())+" (remaining "+to_string_u(cases.size()-case_index)+")")+" |");
		println("| Angle           | "+alignr(57u, format_tag(cs.angle_deg)+" deg")+" |");
	}
	if(!c.nwp_mode) println("| SI Reference U  | "+alignr(57u, format_tag(si_ref_u)+" m/s")+" |");
	case_bc = c.nwp_mode ? c.downstream_bc : bc_from_dir(dir_x, dir_y);
	print_section_title("DEVICE INFORMATION");
	if(!c.nwp_mode) print_kv_row("Downstream BC", case_bc+(c.dataset_mode ? " (auto from batch angle)" : " (auto from profile angle)"));
	update_buffer_nudging(case_bc); update_top_sponge();
}

inline void Driver::create_solver_and_voxelize() {
	G.temperature = use_temperature_bc; // the thermal lattice runs exactly when the reference writes T outputs
	flags_store.clear(); u_store.clear(); T_store.clear();
	nvox = 0ull;
	if(c.dry_run) { flags_store.assign(N, 0u); u_store.assign(3ull*N, 0.0f); if(use_temperature_bc) T_store.assign(N, 1.0f); }
	else {
		const uint Dn = c.Dx*c.Dy*c.Dz; // LBM_Domain's constructor reports per device (FX/lbm.cpp:265-280); here all domains are built in one call
		g_progress.emit("gpu_memory", "Configuring GPU memory", "Allocating CFD buffers on "+to_string_u(Dn)+" device(s)", 0ll, (long long)Dn, false);
		// FX/setup.cpp:4935,5720,6018
		lbm_p.reset(new LBM(uint3(lbmN[0], lbmN[1], lbmN[2]), c.Dx, c.Dy, c.Dz, lbm_nu, 0.0f, 0.0f, 0.0f, 0.0f, lbm_alpha, 0.0f));
		g_progress.emit("gpu_memory", "Configuring GPU memory", to_string_u(Dn)+" device(s): buffers ready", (long long)Dn, (long long)Dn, false);
	}
	if(lbm_p&&lbm_p->get_D()>1u) {
		string devs; for(uint d=0u; d<lbm_p->get_D(); d++) {
			int dv = 0;
			luw_check(luw_group_domain_info(lbm_p->group(), d, nullptr, nullptr, &dv));
			devs += (d ? "," : "")+to_string_u((ulong)dv);
		}
		print_kv_row("Domains", to_string_u(lbm_p->get_D())+" domains ("+to_string_u(c.Dx)+"x"+to_string_u(c.Dy)+"x"+to_string_u(c.Dz)+") of "
			+to_string_u(Nx/c.Dx)+"x"+to_string_u(Ny/c.Dy)+"x"+to_string_u(Nz/c.Dz)+" cells on HIP devices "+devs);
		print_kv_row("", string("halo faces: ")+(luw_group_direct_peer_stores(lbm_p->group()) ? "peer stores of the pack kernels (xGMI)" : "hipMemcpyPeerAsync")
			+(luw_group_overlaps(lbm_p->group()) ? ", overlapped with the interior" : ", after the whole-box kernel"));
		// like the reference, nudging / sponge act only inside domains that own the face (FX/kernel.cpp:1537-1541,1598): say so when a zone is cut
		const uint bz = G.buffer_nudging_active ? (uint)G.buffer_n_cells : 0u, sz = G.top_sponge_active ? (uint)G.sponge_n_cells : 0u;
		if((c.Dz>1u&&std::max(bz, sz)+1u>Nz/c.Dz)||(c.Dy>1u&&bz+1u>Ny/c.Dy)||(c.Dx>1u&&bz+1u>Nx/c.Dx))
			println("| WARNING: a nudging / sponge zone is thicker than a domain: cells of the zone in domains that do not own the face get no forcing (as in "
				"the reference). |");
	}
	flags = c.dry_run ? flags_store.data() : lbm_p->flags.data<uchar>();
	u = c.dry_run ? u_store.data() : lbm_p->u.data<float>();
	Tcell = !use_temperature_bc ? nullptr : c.dry_run ? T_store.data() : lbm_p->T.data<float>(); // lbm.T, pre-filled with 1 (FX/lbm.cpp:304)
	if(c.dry_run) nvox = voxelize_z(mesh, Nx, Ny, Nz, flags_store); // no GPU: host restatement of the kernel
	else { // lbm.voxelize_mesh_on_device(mesh), FX/setup.cpp:4089: every domain voxelises its own box
		const long long Dn = (long long)lbm_p->get_D(); // FX/lbm.cpp:1413-1418,1593-1598
		g_progress.emit("voxelization", "Voxelizing geometry", to_string_u(mesh.n)+" triangles across "+to_string_u((ulong)Dn)+" domain(s)", 0ll, Dn, false);
		lbm_p->voxelize_mesh_on_device(mesh.n, mesh.p0.data(), mesh.p1.data(), mesh.p2.data(), mesh.pmin, mesh.pmax, TYPE_S);
		g_progress.emit("voxelization", "Voxelizing geometry", "Finished domain "+to_string_u((ulong)Dn)+"/"+to_string_u((ulong)Dn), Dn, Dn, false);
		for(ulong n=0ull; n<N; n++) nvox += (flags[n]&TYPE_S)!=0u;
	}
	println("| Info: Voxelized cells (whole domain global, no halos): solid = "+to_string_u(nvox)+", fluid = "+to_string_u(N-nvox)+", total = "+to_string_u(N)
		+".");
	println("| Voxelization done.                                                          |");
}

inline void Driver::build_vk_inlet() { // make_vk_runtime_config + VonKarmanInletUpdater::initialize, FX/setup.cpp:3762-3799,417-534
	vk = VkTables{}; vk_on = false;
	if(!c.vk_enable) return;
	VkRuntimeConfig vc;
	vc.ti = c.vk_ti; vc.sigma_lbm = c.vk_sigma_si*units.unit_s/units.unit_m; vc.L_lbm = units.x(c.vk_L_si);
	vc.nmodes = c.vk_nmodes; vc.seed = c.vk_seed; vc.update_stride = c.vk_stride; vc.uc_mode = c.vk_uc;
	vc.same_realization_all_faces = c.vk_same; vc.stride_interpolation = c.vk_interp; vc.inflow_only = c.vk_inflow_only;
	vc.face_mode = vk_resolve_face_mode(c.vk_face_mode, c.vk_inflow_only);
	for(int k=0; k<3; k++) vc.aniso[k] = c.vk_aniso[k];
	vc.downstream_face_id = case_bc=="-x" ? 0 : case_bc=="+x" ? 1 : case_bc=="-y" ? 2 : case_bc=="+y" ? 3 : -1;
	if(!(vc.L_lbm>0.0f)) println("| WARNING: vk_inlet_l converts to non-positive LBM value. Disabled.            |");
	else vk_on = vk_build_tables(vc, Nx, Ny, Nz, flags, u, vk, [](const string& l) { println(l); });
	if(!vk_on) println(c.profile_mode ? "| VK inlet        | profile case: no valid inflow faces.                       |"
		: "| VK inlet        | dataset case: no valid inflow faces.                       |");
	if(vk_on&&!c.dump_vk.empty()&&case_index==1u) {
		std::ofstream vf(c.dump_vk, std::ios::binary); const uint64_t hdr[2] = {vk.point_count, vk.mode_count};
		vf.write((const char*)hdr, 16);
		vf.write((const char*)vk.point_cell.data(), (std::streamsize)(8ull*vk.point_count));
		vf.write((const char*)vk.point_face.data(), (std::streamsize)vk.point_count);
		vf.write((const char*)vk.point_data.data(), (std::streamsize)(28ull*vk.point_count));
		vf.write((const char*)vk.mode_data.data(), (std::streamsize)(200ull*vk.mode_count));
	}
}

// raw initial state for tests: header (Nx,Ny,Nz,Nz_core as u32; nu, si_u_factor, si_rho_factor as f32) + flags + u + rho(=1)
inline void Driver::dump_setup_file() const {
	std::ofstream df(c.dump_setup, std::ios::binary);
	const uint hdr[4] = {Nx, Ny, Nz, Nz_core};
	const float fh[8] = {lbm_nu, units.si_u(1.0f), units.si_rho(1.0f), G.buffer_inv_tau_lbmu, G.sponge_inv_tau_lbmu, scale_geom, omega[1], omega[2]};
	const int ih[8] = {G.buffer_nudging_active, G.buffer_n_cells, G.buffer_downstream_face_id, G.buffer_nudge_vertical, G.top_sponge_active, G.sponge_n_cells,
		(int)nvox, (int)mapped.load()};
	df.write((const char*)hdr, 16); df.write((const char*)fh, 32); df.write((const char*)ih, 32);
	df.write((const char*)flags, (std::streamsize)N); df.write((const char*)u, (std::streamsize)(12ull*N));
	// optional trailer: T in lattice units
	if(use_temperature_bc) {
		const float th[2] = {units.unit_K, units.unit_K_offset};
		df.write("TEMP", 4);
		df.write((const char*)th, 8);
		df.write((const char*)Tcell, (std::streamsize)(4ull*N));
	}
}

inline void Driver::plan_outputs() {
	total_steps = (c.run_nstep_override>0ull ? c.run_nstep_override : 20001ull)+(ulong)c.research_output_steps;
	unsteady = (ulong)c.unsteady_output_interval;
	results_vtk_dir = c.parent+"/RESULTS/vtk/";
	vtk_dir = results_vtk_dir+vtk_prefix+c.datetime+"_raw_";
	Nz_out = (top_sponge_grid_extend&&Nz_core<Nz) ? Nz_core : Nz;
	geom = VtkGeom{Nx, Ny, Nz, Nz_out, units.si_x(1.0f), {0, 0, 0}};
	{ const uint NN[3] = {Nx, Ny, Nz}; for(int k=0; k<3; k++) geom.origin[k] = geom.spacing*(0.5f-0.5f*(float)NN[k])+vtk_origin_shift[k]; }
	avg_window = c.purge_avg_steps>0u ? std::min((ulong)c.purge_avg_steps, total_steps) : 0ull;
	avg_stride = std::max((ulong)1u, (ulong)c.purge_avg_stride);
	avg_start_t = avg_window>0ull ? total_steps-avg_window+1ull : ~0ull;
	dt_si_d = (double)c.cell_m*((double)lbm_ref_u/(double)si_ref_u);
	probe_window = probe_requests.empty() ? 0ull : (c.probes_output_defined&&c.probes_output_steps>0u) ? std::min((ulong)c.probes_output_steps, total_steps)
		: (c.purge_avg_steps>0u||c.research_output_steps>0u) ? std::min((ulong)std::max(c.purge_avg_steps, c.research_output_steps), total_steps) : total_steps;
	probe_start_t = probe_window>0ull ? total_steps-probe_window+1ull : ~0ull;
}

inline void Driver::resolve_probes() { // FX/setup.cpp:4269-4395
	probes.clear(); probe_cells.clear();
	if(!probe_requests.empty()) {
		if(!probe_geo.valid) print_kv_row("Probes", "disabled: geographic mapping is unavailable");
		else {
			std::vector<string> used;
			for(const ProbeRequest& rq : probe_requests) {
				ProbeColumn pc; pc.req = rq; string why;
				bool ok = resolve_probe_xy(rq, probe_geo, Nx, Ny, c.cell_m, c.si_x, c.si_y, pc.x, pc.y, why);
				if(ok) {
					for(uint z=0u; z<Nz; ++z) if((flags[(ulong)pc.x+((ulong)pc.y+(ulong)z*Ny)*Nx]&TYPE_S)==0u) pc.z.push_back(z);
					if(pc.z.empty()) { ok = false; why = "resolved column has no fluid cell"; }
				}
				if(!ok) { println("| WARNING: probe '"+rq.raw+"' ignored: "+why+"                |"); continue; }
				for(const uint z : pc.z) pc.height_si.push_back((float)(((double)z-(double)pc.z.front()+0.5)*(double)c.cell_m));
				string stem = probe_stem(rq, probe_geo, vtk_prefix);
				if(std::find(used.begin(), used.end(), stem)!=used.end()) {
					uint k = 2u;
					string u2 = stem;
					while(std::find(used.begin(), used.end(), u2)!=used.end()) u2 = stem+"_"+to_string_u(k++);
					stem = u2;
				}
				used.push_back(stem); pc.stem = stem;
				probes.push_back(std::move(pc));
			}
			if(probes.empty()) print_kv_row("Probes", "0 valid probe column after geometry/domain checks");
			else {
				print_kv_row("Probes", to_string_u(probes.size())+" active, "
					+(probe_window>=total_steps ? string("entire run") : "last "+to_string_u(probe_window)+" step(s)"));
				bool first = true;
				for(const ProbeColumn& pc : probes) {
					print_kv_row(first ? "Probe cell" : "", pc.stem+" -> ("+to_string_u(pc.x)+","+to_string_u(pc.y)+"), levels="+to_string_u(pc.z.size()));
					first = false;
					for(const uint z : pc.z) probe_cells.push_back((uint64_t)pc.x+((uint64_t)pc.y+(uint64_t)z*Ny)*Nx);
				}
			}
		}
	}
}
