// driver_setup.hpp -- the sections of main_setup that run once per deck, before any case: probe list, parameter table, grid sizing
// (FX/setup.cpp:3552-3568), inflow inputs (:3600-3729), units and forcing constants (:3731-3903, :3944-3987), geometry (:4001-4113) and the
// profile table (:5777-5912).  Part of the deck driver (luw_driver.cpp); included by it only, after driver_state.hpp.
#pragma once

inline void Driver::read_probe_requests() { // FX/setup.cpp:3396-3426
	if(!c.probes_raw.empty()) {
		for(const string& tok : split_probe_list(c.probes_raw)) {
			ProbeRequest rq; string err;
			if(!parse_probe(tok, rq, err)) { println("| WARNING: ignore probe token '"+tok+"': "+err+"                        |"); continue; }
			probe_requests.push_back(rq);
		}
		if(probe_requests.empty()) println("| WARNING: probes is defined but no valid probe token was parsed.                |");
		else if(!(c.has_cut_lon&&c.has_cut_lat)) println("| WARNING: probes requires cut_lon_manual/cut_lat_manual for lon-lat mapping.    |");
		else {
			probe_geo = make_geo_frame(c.cut_lon[0], c.cut_lon[1], c.cut_lat[0], c.cut_lat[1], c.utm_crs, c.has_rotate_deg, c.rotate_deg);
			if(!probe_geo.valid) println("| WARNING: failed to build probes geographic mapping. Probes are disabled.       |");
		}
	}
	if(c.nwp_mode) { // FX/setup.cpp:3446-3475: the reference asks on stdin; with no terminal attached an empty answer means "continue"
		string v = c.validation; std::transform(v.begin(), v.end(), v.begin(), ::tolower);
		if(v!="pass"&&v!="true"&&v!="1") {
			println("|-----------------------------------------------------------------------------|");
			println("| WARNING: Validation status is '"+c.validation+"'. Pre-processing may be incomplete or invalid. |");
			println("| Proceeding (non-interactive).                                               |");
		}
	}
}

inline void Driver::print_parameters() const {
	println("|"+string(CONSOLE_WIDTH-2u, ' ')+"|");
	print_section_title("PARAMETER INFORMATION");
	println("| Configure deck  | "+alignr(57u, c.deck_path)+" |");
	println("| Casename / Time | "+alignr(40u, c.caseName)+alignr(17u, c.datetime)+" |");
	println("| Basement Height | "+alignr(55u, fmtf(c.z_si_offset))+" m |");
	println("| SI Size (m)     | "+alignr(12u, " X:")+alignl(11u, fmtf(c.si_x))+"   Y: "+alignl(11u, fmtf(c.si_y))+"   Z: "+alignl(11u, fmtf(c.si_z))+" | ");
	if(c.nwp_mode) {
		println("| Downstream BC   | "+alignr(57u, c.downstream_bc)+" |");
		println("| Normal Yaw      | "+alignr(53u, c.downstream_bc_yaw)+" deg |");
	}
	else {
		println("| Downstream BC   | "+alignr(57u, "auto by angle (dominant axis)")+" |");
		println("| Normal Yaw      | "+alignr(57u, "auto by angle list")+" |");
	}
	println("| Downstream Open | "+alignr(57u, c.downstream_open_face ? string("true") : string("false"))+" |");
	println("| GPU Decompose   | "+alignr(49u, to_string_u(c.Dx))+", "+alignr(2u, to_string_u(c.Dy))+", "+alignr(2u, to_string_u(c.Dz))+" |");
	println("| Run Steps       | "+alignr(57u, c.run_nstep_override>0ull ? to_string_u(c.run_nstep_override)+" (run_nstep)" : string("20001 (default)"))+" |");
	{ // FX/setup.cpp:3507-3527
		string d = "off"; if(!probe_requests.empty()) {
			d = to_string_u(probe_requests.size())+" request(s)";
			if(!probe_geo.valid) d += " (mapping unavailable)";
		}
		println("| Probes         | "+alignr(57u, d)+" |");
		string w = "n/a";
		if(!probe_requests.empty()) w = (c.probes_output_defined&&c.probes_output_steps>0u)
			? "last "+to_string_u(c.probes_output_steps)+" step(s) via probes_output" : (c.purge_avg_steps>0u||c.research_output_steps>0u)
			? "fallback last "+to_string_u(std::max(c.purge_avg_steps, c.research_output_steps))+" step(s)" : string("entire simulation");
		println("| Probes Window  | "+alignr(57u, w)+" |");
	}
	println("| DDF storage     | "+alignr(57u, c.fp16c ? string("FP16C (as the shipped reference build)") : string("FP32"))+" |");
	if(c.fp16c) println("| Arithmetic      | "+alignr(57u, c.native_arith ? string("native (v_rcp / v_sqrt, fused multiply-adds; --arith exact)")
		: string("exact (bit-equal to the CPU restatement; --arith native)"))+" |");
}

inline void Driver::size_lattice() { // FX/setup.cpp:3552-3568
	lbmN[0] = (uint)std::max(1, (int)(c.si_x/c.cell_m+0.5f)); lbmN[1] = (uint)std::max(1, (int)(c.si_y/c.cell_m+0.5f));
	sponge_cells_cfg = std::max(1, (int)std::lround(c.sponge_thickness_m/c.cell_m));
	Nz_core = (uint)std::max(1, (int)(c.si_z/c.cell_m+0.5f));
	top_sponge_grid_extend = c.enable_top_sponge&&c.sponge_tau_s>0.0f&&c.sponge_ref_mode==0&&Nz_core>2u;
	lbmN[2] = Nz_core+(top_sponge_grid_extend ? (uint)sponge_cells_cfg : 0u);
	side_ref_z_cap = top_sponge_grid_extend ? (int)Nz_core-1 : -1;
	const uint Nx = lbmN[0], Ny = lbmN[1], Nz = lbmN[2];
	print_section_title("DOMAIN AND TRANSFORMATION");
	println("| Grid Resolution | "+alignr(45u, to_string_u(Nx))+","+alignr(5u, to_string_u(Ny))+","+alignr(5u, to_string_u(Nz))+" (nCell = "
		+to_string_u((ulong)Nx*Ny*Nz)+") |");
	if(top_sponge_grid_extend) println("| Top sponge grid | "
		+alignr(57u, "core Nz="+to_string_u(Nz_core)+", ext="+to_string_u((ulong)sponge_cells_cfg)+", total Nz="+to_string_u(Nz))+" |");
	{
		const uint core = vram_required_mb_per_device(Nx, Ny, Nz, c.Dx, c.Dy, c.Dz), extra = vk_extra_mb(c, Nx, Ny, Nz);
		if(extra>0u) println("| GPU Estimate    | "
			+alignr(57u, to_string_u(c.Dx*c.Dy*c.Dz)+"x "+to_string_u(core+extra)+" MB (core "+to_string_u(core)+" + extra "+to_string_u(extra)+")")+" |");
		else println("| GPU Estimate    | "+alignr(57u, to_string_u(c.Dx*c.Dy*c.Dz)+"x "+to_string_u(core)+" MB")+" |");
	}
}

inline void Driver::read_inflow_inputs() { // what sets si_ref_u: the SurfData CSV, the inflow list or profile.dat
	if(c.nwp_mode) { // FX/setup.cpp:3600-3650
		const string csv = c.parent+"/proj_temp/SurfData_"+c.datetime+".csv";
		if(!read_surfdata_csv(csv, surf)) println("ERROR: could not open CSV "+csv);
		for(const string& w : surf.warnings) println(w);
		if(surf.rows.empty()) fatal("| ERROR: no inlet samples when computing si_ref_u. Aborting...                |");
		float max_u = 0.0f;
		for(const SurfSample& sm : surf.rows) { const float speed = std::sqrt(sm.u.x*sm.u.x+sm.u.y*sm.u.y+sm.u.z*sm.u.z); if(speed>max_u) max_u = speed; }
		si_ref_u = max_u;
		if(surf.has_T&&surf.rows_T>0ull) { // adaptive affine temperature map, FX/setup.cpp:3627-3648
			float tmin = surf.tmin, tmax = surf.tmax; if(tmin>tmax) std::swap(tmin, tmax);
			if(std::isfinite(tmin)&&std::isfinite(tmax)&&tmax>0.0f) {
				const float tref = 0.5f*(tmin+tmax);
				if(std::isfinite(tref)&&tref>0.0f) { temperature_ref_kelvin = tref; temperature_ref_adaptive = true; }
				const float thalf = 0.5f*(tmax-tmin);
				temperature_scale_kelvin = (std::isfinite(thalf)&&thalf>1.0e-6f) ? thalf : 1.0f; temperature_scale_adaptive = true;
			}
		}
	} else if(c.dataset_mode) {
		if(c.inflow_list.empty()) fatal("| ERROR: dataset generation requires inflow list (inflow=[...]).              |");
		if(c.angle_list.empty()) fatal("| ERROR: dataset generation requires angle list (angle=[...]).                |");
		si_ref_u = *std::max_element(c.inflow_list.begin(), c.inflow_list.end());
	} else { // FX/setup.cpp:3660-3729
		if(c.angle_list.empty()) fatal("| ERROR: profile forcing requires angle list (angle=[...]).                   |");
		const float agl = c.si_z-c.z_si_offset;
		if(agl<=0.0f) fatal("| ERROR: invalid profile domain height. Check si_z_cfd/base_height.           |");
		auto smp = read_profile_dat(c.parent+"/wind_bc/profile.dat");
		if(smp.empty()) fatal("| ERROR: no profile samples found. Aborting...                                |");
		std::sort(smp.begin(), smp.end(), [](const auto& a, const auto& b) { return a.first<b.first; });
		for(const auto& s : smp) {
			if(!prof_z.empty()&&std::fabs(s.first-prof_z.back())<1e-6f) { prof_u.back() = s.second; continue; }
			prof_z.push_back(s.first);
			prof_u.push_back(s.second);
		}
		if(prof_z.size()<2u) fatal("| ERROR: profile.dat needs at least two valid samples. Aborting...            |");
		if(agl>1.0f&&prof_z.back()<=1.5f) {
			for(float& z : prof_z) z *= agl;
			println("| Profile z unit  | normalized -> scaled by domain AGL height                 |");
		}
		float max_u = 0.0f; for(const float v : prof_u) if(v>max_u) max_u = v;
		if(max_u<=0.0f) fatal("| ERROR: profile.dat has non-positive max U. Aborting...                      |");
		si_ref_u = max_u;
		println("| Profile samples | "+alignr(57u, to_string_u(prof_z.size()))+" |");
		println("| Profile z range | "+alignr(24u, fmtf(prof_z.front()))+" to "+alignl(16u, fmtf(prof_z.back()))+" m |");
		println("| Profile domain  | "+alignr(57u, fmtf(agl))+" m AGL |");
	}
}

inline void Driver::update_coriolis() { // FX/setup.cpp:3800-3823
	if(!c.enable_coriolis) return;
	const float lat = 0.5f*(c.cut_lat[0]+c.cut_lat[1]);
	const float Om = 7.292115e-5f, deg2rad = 3.14159265358979323846f/180.0f, lat_rad = lat*deg2rad;
	const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u);
	omega[0] = 0.0f*dt_si; omega[1] = Om*cosf(lat_rad)*dt_si; omega[2] = Om*sinf(lat_rad)*dt_si;
}
inline void Driver::update_buffer_nudging(const string& bc) { // FX/setup.cpp:3844-3856
	const uint Nx = lbmN[0], Ny = lbmN[1], Nz = lbmN[2];
	G.buffer_downstream_face_id = buffer_face_id_from_bc(bc);
	const uint min_dim = std::min(Nx, std::min(Ny, Nz)), max_nbuf = std::max(1u, min_dim/4u);
	int nbuf = (int)std::lround(c.buffer_thickness_m/c.cell_m);
	if(nbuf<1) nbuf = 1; if((uint)nbuf>max_nbuf) nbuf = (int)max_nbuf;
	G.buffer_n_cells = nbuf;
	const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u);
	G.buffer_inv_tau_lbmu = c.buffer_tau_s>0.0f ? dt_si/c.buffer_tau_s : 0.0f;
	G.buffer_nudging_active = c.enable_buffer_nudging&&c.buffer_tau_s>0.0f;
	G.buffer_nudge_vertical = c.buffer_nudge_vertical;
}
inline void Driver::update_top_sponge() { // FX/setup.cpp:3867-3881
	const uint Nz = lbmN[2];
	int ns = std::max(sponge_cells_cfg, 1);
	if(Nz>2u) ns = std::min(ns, (int)Nz-2);
	G.sponge_n_cells = ns;
	const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u);
	G.sponge_inv_tau_lbmu = c.sponge_tau_s>0.0f ? dt_si/c.sponge_tau_s : 0.0f;
	G.top_sponge_active = top_sponge_grid_extend&&c.sponge_tau_s>0.0f&&c.sponge_ref_mode==0&&Nz_core>2u;
}

inline void Driver::set_units_and_forcing() {
	const uint Ny = lbmN[1];
	units.set_m_kg_s_K((float)Ny, lbm_ref_u, 1.0f, 1.0f, c.si_y, si_ref_u, si_rho, temperature_scale_kelvin);
	units.set_temperature_reference(1.0f, temperature_ref_kelvin); // T_lbm = 1.0 maps to the (adaptive) reference temperature, FX/setup.cpp:3731-3732
	lbm_alpha = units.alpha(2.10E-5f);                // thermal diffusivity of air, FX/setup.cpp:3738-3741
	use_temperature_bc = c.nwp_mode&&c.buoyancy&&surf.has_T;
	println("| Info: Unit Conversion: 1 cell = "+to_string_fd(1000.0f*units.si_x(1.0f), 3u)+" mm, 1 s = "+to_string_u(units.t(1.0f))+" time steps");
	u_scale = lbm_ref_u/si_ref_u;
	lbm_nu = units.nu(si_nu);
	G.fp16c = c.fp16c; G.native_arith = c.native_arith; G.device = c.device; G.devices = c.devices; G.kernel = c.kernel;
	if(c.nwp_mode) {
		println("| SI Reference U  | "+alignl(7u, fmtf(si_ref_u))+alignl(50u, "m/s")+" |");
		println("| LBM Reference U | "+alignl(7u, fmtf(lbm_ref_u))+alignl(50u, "(Nondimensionalized)")+" |");
		println("| Temp Reference  | "
			+alignr(57u, fmtf(temperature_ref_kelvin)+(temperature_ref_adaptive ? " K (auto center of input Tmin/Tmax)" : " K (default)"))+" |");
		println("| Temp Scale      | "+alignr(57u, fmtf(temperature_scale_kelvin)
			+(temperature_scale_adaptive ? " K per 1.0 T_lbm (auto from input range)" : " K per 1.0 T_lbm (default)"))+" |");
		println("| Thermal alpha   | "+alignr(57u, to_string_fd(lbm_alpha, 8u))+" |");
		println("| Thermal tau_T   | "+alignr(57u, to_string_fd(2.0f*lbm_alpha+0.5f, 8u))+" |");
		println("| Thermal beta    | "
			+alignr(57u, c.buoyancy ? to_string_fd(units.beta(1.0f/temperature_ref_kelvin), 8u) : string("0 (disabled by buoyancy=false)"))+" |");
		if(c.buoyancy) println("| Thermal note    | temperature is advected as a passive scalar: the solver's volume force is zero, as in the reference |");
	}
	update_coriolis(); update_buffer_nudging(c.nwp_mode ? c.downstream_bc : string("+y")); update_top_sponge();
	if(c.nwp_mode&&c.enable_coriolis) {
		print_kv_row("Coriolis", "enabled. center(lon,lat)=("+to_string_fd(0.5f*(c.cut_lon[0]+c.cut_lon[1]), 6u)+", "
			+to_string_fd(0.5f*(c.cut_lat[0]+c.cut_lat[1]), 6u)+") deg");
		print_kv_row("", "Omega(lbmu)=("+to_string_fd(omega[0], 8u)+", "+to_string_fd(omega[1], 8u)+", "+to_string_fd(omega[2], 8u)+") per step");
	}
	else if(c.nwp_mode) print_kv_row("Coriolis", "disabled by 'coriolis_term' setting in .luw");
	print_kv_row("Buffer nudging", G.buffer_nudging_active ? (c.nwp_mode ? "enabled" : "enabled (downstream face auto by angle)") : "disabled");
	print_kv_row("", "Nbuf="+to_string_u((ulong)G.buffer_n_cells)+" cells, tau_s="+to_string_fd(c.buffer_tau_s, 6u)+" s");
	print_kv_row("", "inv_tau_lbmu="+to_string_fd(G.buffer_inv_tau_lbmu, 8u)+", downstream_face_id="
		+(c.nwp_mode ? to_string_u((ulong)G.buffer_downstream_face_id) : string("auto"))+", nudge_vertical="+to_string_u((ulong)G.buffer_nudge_vertical));
	print_kv_row("Top sponge", G.top_sponge_active ? "enabled" : "disabled");
	print_kv_row("", "Nsponge="+to_string_u((ulong)G.sponge_n_cells)+" cells, tau_s="+to_string_fd(c.sponge_tau_s, 6u)+" s");
	print_kv_row("", "inv_tau_lbmu="+to_string_fd(G.sponge_inv_tau_lbmu, 8u)+", ref_mode="+std::to_string(c.sponge_ref_mode));
	if(G.top_sponge_active) print_kv_row("", "core_top_z="+to_string_u(Nz_core-1u)+", side_ref_cap_z="+std::to_string(side_ref_z_cap));

	if(c.nwp_mode) { // FX/setup.cpp:3944-3987
		if(surf.has_T) {
			println("| T column        | detected ("+to_string_u(surf.rows_T)+" rows)                               |");
			println("| CSV T range SI  | "+alignr(24u, fmtf(surf.tmin))+" to "+alignl(16u, fmtf(surf.tmax))+" K |");
			println(c.buoyancy ? "| Temperature BC  | enabled from CSV T (Kelvin -> nondimensionalized)               |"
				: "| Temperature BC  | buoyancy=false, ignore T column                                 |");
		} else println("| T column        | not found, keep legacy velocity-only boundary behavior           |");
		if(use_temperature_bc) {
			ulong out_of_range = 0ull; for(const SurfSample& r : surf.rows) if(r.T<223.15f||r.T>343.15f) out_of_range++;
			if(out_of_range>0ull) println("| WARNING: "+to_string_u(out_of_range)+" temperature samples are outside [-50C, 70C].                |");
			T_bc_min = units.T(surf.tmin); T_bc_max = units.T(surf.tmax); if(T_bc_min>T_bc_max) std::swap(T_bc_min, T_bc_max);
		}
	}
}

inline void Driver::load_geometry() { // FX/setup.cpp:4001-4113
	const uint Nx = lbmN[0], Ny = lbmN[1], Nz = lbmN[2];
	print_section_title("LOADING GEOMETRY AND VOXELIZE");
	string stl_path;
	{
		const std::filesystem::path dir = std::filesystem::path(c.parent)/"proj_temp";
		if(!std::filesystem::exists(dir)) fatal("ERROR: directory not found: "+dir.string());
		std::vector<string> names;
		for(const auto& e : std::filesystem::directory_iterator(dir)) if(e.is_regular_file()) names.push_back(e.path().filename().string());
		std::sort(names.begin(), names.end());
		auto ends = [](const string& s, const string& suf) { return s.size()>=suf.size()&&s.substr(s.size()-suf.size())==suf; };
		const string a = c.caseName+"_DEM_PF.stl", b = c.caseName+"_DG.stl";
		if(c.profile_mode&&std::filesystem::is_regular_file(dir/a)) stl_path = (dir/a).string();
		else if(std::filesystem::is_regular_file(dir/b)) stl_path = (dir/b).string();
		else {
			std::vector<string> order; if(c.profile_mode) order.push_back("_DEM_PF.stl"); order.push_back("_DG.stl"); order.push_back(".stl");
			for(const string& suf : order) {
				for(const string& n : names) if(ends(n, suf)) { stl_path = (dir/n).string(); break; }
				if(!stl_path.empty()) break;
			}
		}
		if(stl_path.empty()) fatal("ERROR: no STL file under "+dir.string());
	}
	if(!read_stl(stl_path, mesh)) fatal("ERROR: failed to load STL");
	println("| Info: Loading \""+stl_path+"\" with "+to_string_u(mesh.n)+" triangles.");
	g_progress.emit("load_stl", "Loading STL", stl_path+" ("+to_string_u(mesh.n)+" triangles)", 0ll, 1ll, false); // FX/utilities.hpp:4850-4887
	g_progress.emit("load_stl", "Loading STL", stl_path+" loaded", 1ll, 1ll, false);
	for(int k=0; k<3; k++) { stl_min[k] = mesh.pmin[k]; stl_max[k] = mesh.pmax[k]; }
	{ const uint NN[3] = {Nx, Ny, Nz}; for(int k=0; k<3; k++) vtk_origin_shift[k] = stl_min[k]-units.si_x(0.5f-0.5f*(float)NN[k]); }
	scale_geom = units.x(c.si_x)/(stl_max[0]-stl_min[0]);
	mesh_scale_translate(mesh, scale_geom);
	print_kv_row("Geometry STL", stl_path);
	print_kv_row("STL bounds SI", "x=["+to_string_fd(stl_min[0], 3u)+", "+to_string_fd(stl_max[0], 3u)+"], y=["+to_string_fd(stl_min[1], 3u)+", "
		+to_string_fd(stl_max[1], 3u)+"], z=["+to_string_fd(stl_min[2], 3u)+", "+to_string_fd(stl_max[2], 3u)+"]");
	print_kv_row("Geometry", "scaled by "+to_string_fd(scale_geom, 4u)+", ready for voxelization");
	if(c.profile_mode) { // FX/setup.cpp:4095-4113
		dem = read_dem_csv(c.parent+"/proj_temp/interpolated_dem.csv");
		if(!dem.x.empty()) {
			print_kv_row("Terrain DEM", "Loaded "+to_string_u(dem.x.size())+" points from interpolated_dem.csv");
			print_kv_row("DEM bounds SI", "x=["+to_string_fd(dem.xmin, 3u)+", "+to_string_fd(dem.xmax, 3u)+"], y=["+to_string_fd(dem.ymin, 3u)+", "
				+to_string_fd(dem.ymax, 3u)+"], elev=["+to_string_fd(dem.emin, 3u)+", "+to_string_fd(dem.emax, 3u)+"]");
		} else print_kv_row("Terrain DEM", "interpolated_dem.csv not found or empty, fallback to flat ground");
	}
}

inline void Driver::build_profile_table() { // FX/setup.cpp:5777-5912
	const uint Nx = lbmN[0], Ny = lbmN[1], Nz = lbmN[2];
	origin_z = 0.5f-0.5f*(float)Nz;
	flat_ground = origin_z+units.x(c.z_si_offset);
	ground_z_min = ground_z_max = flat_ground;
	if(c.profile_mode&&!dem.x.empty()) { // DEM points -> STL frame -> lattice units, FX/setup.cpp:5790-5847
		const float origin_x = 0.5f-0.5f*(float)Nx, origin_y = 0.5f-0.5f*(float)Ny;
		const float dem_rx = dem.xmax-dem.xmin, dem_ry = dem.ymax-dem.ymin, stl_rx = stl_max[0]-stl_min[0], stl_ry = stl_max[1]-stl_min[1];
		if(dem_rx>1.0e-6f&&dem_ry>1.0e-6f&&stl_rx>1.0e-6f&&stl_ry>1.0e-6f) {
			const float sx = stl_rx/dem_rx, sy = stl_ry/dem_ry;
			if(fmaxf(fabsf(sx-1.0f), fabsf(sy-1.0f))>0.02f||fabsf(dem.xmin-stl_min[0])/stl_rx>0.02f||fabsf(dem.ymin-stl_min[1])/stl_ry>0.02f) {
				println("| Terrain DEM     | WARNING: DEM/STL XY bounds mismatch. Apply affine bounds alignment. |");
				println("|                 | DEM->STL scale x="+to_string_fd(sx, 6u)+", y="+to_string_fd(sy, 6u)+"                             |");
			}
			std::vector<float> gx, gy, gz;
			ground_z_min = +FLT_MAX; ground_z_max = -FLT_MAX;
			for(size_t i=0u; i<dem.x.size(); i++) {
				const float xs = stl_min[0]+(dem.x[i]-dem.xmin)*sx, ys = stl_min[1]+(dem.y[i]-dem.ymin)*sy, zs = c.z_si_offset+dem.e[i];
				const float xl = origin_x+(xs-stl_min[0])*scale_geom, yl = origin_y+(ys-stl_min[1])*scale_geom, zl = origin_z+(zs-stl_min[2])*scale_geom;
				if(!std::isfinite(xl)||!std::isfinite(yl)||!std::isfinite(zl)) continue;
				gx.push_back(xl); gy.push_back(yl); gz.push_back(zl);
				ground_z_min = fminf(ground_z_min, zl); ground_z_max = fmaxf(ground_z_max, zl);
			}
			if(!gz.empty()) { ground_plane.build(gx, gy, gz, flat_ground); use_dem_ground = ground_plane.has_samples(); }
			if(use_dem_ground) println("| Terrain DEM     | profile ground enabled. z(SI) range "+to_string_fd(units.si_x(ground_z_min-origin_z), 3u)+" .. "
				+to_string_fd(units.si_x(ground_z_max-origin_z), 3u)+" m |");
			else { println("| Terrain DEM     | no valid points after mapping, fallback to flat ground     |"); ground_z_min = ground_z_max = flat_ground; }
		} else println("| Terrain DEM     | invalid DEM or STL XY range, fallback to flat ground       |");
	}
	if(c.profile_mode) {
		const float solver_top_si = units.si_x((float)(Nz-1u));
		const float core_top_si = side_ref_z_cap>=0 ? units.si_x((float)side_ref_z_cap) : solver_top_si;
		float ground_min_si = units.si_x(ground_z_min-origin_z), ground_max_si = units.si_x(ground_z_max-origin_z);
		if(!std::isfinite(ground_min_si)) ground_min_si = c.z_si_offset;
		if(!std::isfinite(ground_max_si)) ground_max_si = ground_min_si;
		float table_top = solver_top_si-ground_min_si;
		if(!std::isfinite(table_top)||table_top<=0.0f) table_top = std::max(profile_dz, c.si_z-ground_min_si);
		table_top = std::max(table_top, profile_dz);
		const uint steps = (uint)std::ceil(table_top/profile_dz);
		float umin = 0.0f, umax = 0.0f;
		prof_lbmu.assign(steps+1u, 0.0f);
		for(uint i=0u; i<=steps; ++i) {
			const float zq = std::min(table_top, (float)i*profile_dz);
			float v = interpolate_profile_cubic(prof_z, prof_u, zq);
			if(v<0.0f) v = 0.0f;
			if(i==0u) umin = umax = v; umin = std::min(umin, v); umax = std::max(umax, v);
			prof_lbmu[i] = v*u_scale;
		}
		println("| Profile table   | local-terrain AGL top="+to_string_fd(table_top, 3u)+" m, core_top="+to_string_fd(core_top_si, 3u)+" m, solver_top="
			+to_string_fd(solver_top_si, 3u)+" m |");
		println("| Profile ground  | z(SI) min/max="+to_string_fd(ground_min_si, 3u)+" / "+to_string_fd(ground_max_si, 3u)+" m |");
		println("| Profile U range | "+alignr(24u, fmtf(umin))+" to "+alignl(16u, fmtf(umax))+" m/s |");
	}
}
inline float Driver::profile_speed(const float pos_z, const float ground_z) const { // FX/setup.cpp:5901-5912
	if(pos_z<=ground_z) return 0.0f;
	const float inv_dz = 1.0f/profile_dz;
	const uint last = (uint)(prof_lbmu.size()-1u);
	float z_agl = units.si_x(pos_z-ground_z);
	if(z_agl<0.0f) z_agl = 0.0f;
	long idx = std::lround(z_agl*inv_dz);
	if(idx<0l) idx = 0l;
	return prof_lbmu[std::min((uint)idx, last)];
}

inline void Driver::list_cases() {
	if(c.nwp_mode) cases.push_back({0.0f, 0.0f});
	else if(c.dataset_mode) { for(const float in : c.inflow_list) for(const float an : c.angle_list) cases.push_back({in, an}); }
	else for(const float an : c.angle_list) cases.push_back({0.0f, an});
}
