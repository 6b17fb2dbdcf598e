// luw_driver.cpp -- process-level drop-in for the reference's solver executable `FluidX3D <deck>`
// (argv contract FX/setup.cpp:2768-2773; project dir = the deck's parent, FX/setup.cpp:3391), host side in C++ over
// the C-ABI (lbm.hpp mirror).  Re-states just enough of main_setup (FX/setup.cpp:2726-6154) to consume the same
// deck / proj_temp / wind_bc files and to write the same RESULTS/vtk files:
//   deck grammar FX/setup.cpp:40-178, key handlers :2918-3305, mesh_control :3364-3390 (gpu_memory bisection :335-407
//   with the shipped build's memory model FX/lbm.cpp:188-228), grid sizing :3552-3568, profile samples :3660-3729,
//   Units FX/units.hpp:21-67, Coriolis / buffer / sponge constants FX/setup.cpp:3800-3903, STL FX/utilities.hpp:4835-4888
//   + transform FX/setup.cpp:4070-4087, profile table :5777-5912, flags/u fill :5914-5995 (profile mode) and :5655-5688
//   (dataset mode), run loop :4117-4911, VTK writers FX/lbm.hpp:307-356 and FX/setup.cpp:2513-2683.
// Modes: *.luw (NWP: SurfData CSV boundaries), *.luwpf (profile, with optional DEM ground plane) and *.luwdg (dataset).
// Temperature (buoyancy = true and a T column in the CSV): boundary temperatures, thermal lattice, T / T_avg outputs.
// Not in this build (announced on the console, never silently): PNG frames.
// Differences by design: time averaging runs on the device (luw_stats_*); --dry-run voxelises on the host; the von-Karman
// inlet tables are built here (vk_inlet.hpp) and evaluated on the device before every step.
// Decks with n_gpu = [Dx, Dy, Dz] run all Dx*Dy*Dz domains in THIS process, one HIP device each (like the reference's LBM object;
// halos between the devices inside the library, luw_group_*).
// Options after the deck path (the reference ignores extra args): --ddf fp32|fp16c (default fp16c = shipped build),
//   --device N (first device; domain d runs on N + d), --devices a,b,.. (explicit device per domain), --kernel auto|scalar|pair,
//   --dry-run (host stage only, no GPU), --sizing-only (stop after grid / unit / buffer / sponge numbers),
//   --dump-setup FILE (raw initial state of the first case).
#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>
#include <cmath>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fcntl.h>
#include <unistd.h>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <thread>
#include <unordered_map>
#include <vector>

#include "lbm.hpp"
#include "vk_inlet.hpp"
#include "bc_builders.hpp"
#include "probes.hpp"
#include "progress.hpp"
#include "deck.hpp"

using namespace luw_host;
using std::string;
#include "console.hpp"
#include "setup_math.hpp"
#include "mesh.hpp"
#include "vtk_writer.hpp"
#include "deck_config.hpp"

// ------------------------------------------------------------------------------------------------ main
int main(int argc, char** argv) {
	Config c;
	if(!parse_command_line(argc, argv, c)) return -1;
	println(hr_plain());
	println("|"+alignc(CONSOLE_WIDTH-2u, "LatticeUrbanWind LUW core for AMD Instinct MI355X (HIP, D3Q19 SRT + Smagorinsky)")+"|");
	println(hr_plain());
	read_deck(c);
	std::vector<ProbeRequest> probe_requests; GeoFrame probe_geo; // FX/setup.cpp:3396-3426
	if(!c.probes_raw.empty()) {
		for(const string& tok : split_probe_list(c.probes_raw)) {
			ProbeRequest rq; string err;
			if(!parse_probe(tok, rq, err)) { println("| WARNING: ignore probe token '"+tok+"': "+err+"                        |"); continue; }
			probe_requests.push_back(rq);
		}
		if(probe_requests.empty()) println("| WARNING: probes is defined but no valid probe token was parsed.                |");
		else if(!(c.has_cut_lon&&c.has_cut_lat)) println("| WARNING: probes requires cut_lon_manual/cut_lat_manual for lon-lat mapping.    |");
		else { probe_geo = make_geo_frame(c.cut_lon[0], c.cut_lon[1], c.cut_lat[0], c.cut_lat[1], c.utm_crs, c.has_rotate_deg, c.rotate_deg); if(!probe_geo.valid) println("| WARNING: failed to build probes geographic mapping. Probes are disabled.       |"); }
	}
	if(c.nwp_mode) { // FX/setup.cpp:3446-3475: the reference asks on stdin; with no terminal attached an empty answer means "continue"
		string v = c.validation; std::transform(v.begin(), v.end(), v.begin(), ::tolower);
		if(v!="pass"&&v!="true"&&v!="1") { println("|-----------------------------------------------------------------------------|"); println("| WARNING: Validation status is '"+c.validation+"'. Pre-processing may be incomplete or invalid. |"); println("| Proceeding (non-interactive).                                               |"); }
	}

	println("|"+string(CONSOLE_WIDTH-2u, ' ')+"|");
	print_section_title("PARAMETER INFORMATION");
	println("| Configure deck  | "+alignr(57u, c.deck_path)+" |");
	println("| Casename / Time | "+alignr(40u, c.caseName)+alignr(17u, c.datetime)+" |");
	println("| Basement Height | "+alignr(55u, fmtf(c.z_si_offset))+" m |");
	println("| SI Size (m)     | "+alignr(12u, " X:")+alignl(11u, fmtf(c.si_x))+"   Y: "+alignl(11u, fmtf(c.si_y))+"   Z: "+alignl(11u, fmtf(c.si_z))+" | ");
	if(c.nwp_mode) { println("| Downstream BC   | "+alignr(57u, c.downstream_bc)+" |"); println("| Normal Yaw      | "+alignr(53u, c.downstream_bc_yaw)+" deg |"); }
	else { println("| Downstream BC   | "+alignr(57u, "auto by angle (dominant axis)")+" |"); println("| Normal Yaw      | "+alignr(57u, "auto by angle list")+" |"); }
	println("| Downstream Open | "+alignr(57u, c.downstream_open_face ? string("true") : string("false"))+" |");
	println("| GPU Decompose   | "+alignr(49u, to_string_u(c.Dx))+", "+alignr(2u, to_string_u(c.Dy))+", "+alignr(2u, to_string_u(c.Dz))+" |");
	println("| Run Steps       | "+alignr(57u, c.run_nstep_override>0ull ? to_string_u(c.run_nstep_override)+" (run_nstep)" : string("20001 (default)"))+" |");
	{ // FX/setup.cpp:3507-3527
		string d = "off"; if(!probe_requests.empty()) { d = to_string_u(probe_requests.size())+" request(s)"; if(!probe_geo.valid) d += " (mapping unavailable)"; }
		println("| Probes         | "+alignr(57u, d)+" |");
		string w = "n/a";
		if(!probe_requests.empty()) w = (c.probes_output_defined&&c.probes_output_steps>0u) ? "last "+to_string_u(c.probes_output_steps)+" step(s) via probes_output" : (c.purge_avg_steps>0u||c.research_output_steps>0u) ? "fallback last "+to_string_u(std::max(c.purge_avg_steps, c.research_output_steps))+" step(s)" : string("entire simulation");
		println("| Probes Window  | "+alignr(57u, w)+" |");
	}
	println("| DDF storage     | "+alignr(57u, c.fp16c ? string("FP16C (as the shipped reference build)") : string("FP32"))+" |");

	const float lbm_ref_u = 0.10f; float si_ref_u = 10.0f; const float si_nu = 1.48E-5f, si_rho = 1.225f;
	const uint Nx = (uint)std::max(1, (int)(c.si_x/c.cell_m+0.5f)), Ny = (uint)std::max(1, (int)(c.si_y/c.cell_m+0.5f));
	const int sponge_cells_cfg = std::max(1, (int)std::lround(c.sponge_thickness_m/c.cell_m));
	const uint Nz_core = (uint)std::max(1, (int)(c.si_z/c.cell_m+0.5f));
	const bool top_sponge_grid_extend = c.enable_top_sponge&&c.sponge_tau_s>0.0f&&c.sponge_ref_mode==0&&Nz_core>2u;
	const uint Nz = Nz_core+(top_sponge_grid_extend ? (uint)sponge_cells_cfg : 0u);
	const int side_ref_z_cap = top_sponge_grid_extend ? (int)Nz_core-1 : -1;
	print_section_title("DOMAIN AND TRANSFORMATION");
	println("| Grid Resolution | "+alignr(45u, to_string_u(Nx))+","+alignr(5u, to_string_u(Ny))+","+alignr(5u, to_string_u(Nz))+" (nCell = "+to_string_u((ulong)Nx*Ny*Nz)+") |");
	if(top_sponge_grid_extend) println("| Top sponge grid | "+alignr(57u, "core Nz="+to_string_u(Nz_core)+", ext="+to_string_u((ulong)sponge_cells_cfg)+", total Nz="+to_string_u(Nz))+" |");
	{
		const uint core = vram_required_mb_per_device(Nx, Ny, Nz, c.Dx, c.Dy, c.Dz), extra = vk_extra_mb(c, Nx, Ny, Nz);
		if(extra>0u) println("| GPU Estimate    | "+alignr(57u, to_string_u(c.Dx*c.Dy*c.Dz)+"x "+to_string_u(core+extra)+" MB (core "+to_string_u(core)+" + extra "+to_string_u(extra)+")")+" |");
		else println("| GPU Estimate    | "+alignr(57u, to_string_u(c.Dx*c.Dy*c.Dz)+"x "+to_string_u(core)+" MB")+" |");
	}
	std::vector<float> prof_z, prof_u;
	float temperature_ref_kelvin = 293.15f, temperature_scale_kelvin = 293.15f; bool temperature_ref_adaptive = false, temperature_scale_adaptive = false;
	SurfData surf;
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
		for(const auto& s : smp) { if(!prof_z.empty()&&std::fabs(s.first-prof_z.back())<1e-6f) { prof_u.back() = s.second; continue; } prof_z.push_back(s.first); prof_u.push_back(s.second); }
		if(prof_z.size()<2u) fatal("| ERROR: profile.dat needs at least two valid samples. Aborting...            |");
		if(agl>1.0f&&prof_z.back()<=1.5f) { for(float& z : prof_z) z *= agl; println("| Profile z unit  | normalized -> scaled by domain AGL height                 |"); }
		float max_u = 0.0f; for(const float v : prof_u) if(v>max_u) max_u = v;
		if(max_u<=0.0f) fatal("| ERROR: profile.dat has non-positive max U. Aborting...                      |");
		si_ref_u = max_u;
		println("| Profile samples | "+alignr(57u, to_string_u(prof_z.size()))+" |");
		println("| Profile z range | "+alignr(24u, fmtf(prof_z.front()))+" to "+alignl(16u, fmtf(prof_z.back()))+" m |");
		println("| Profile domain  | "+alignr(57u, fmtf(agl))+" m AGL |");
	}
	Units units;
	units.set_m_kg_s_K((float)Ny, lbm_ref_u, 1.0f, 1.0f, c.si_y, si_ref_u, si_rho, temperature_scale_kelvin);
	units.set_temperature_reference(1.0f, temperature_ref_kelvin); // T_lbm = 1.0 maps to the (adaptive) reference temperature, FX/setup.cpp:3731-3732
	const float lbm_alpha = units.alpha(2.10E-5f);                // thermal diffusivity of air, FX/setup.cpp:3738-3741
	const bool use_temperature_bc = c.nwp_mode&&c.buoyancy&&surf.has_T;
	println("| Info: Unit Conversion: 1 cell = "+to_string_fd(1000.0f*units.si_x(1.0f), 3u)+" mm, 1 s = "+to_string_u(units.t(1.0f))+" time steps");
	float u_scale = lbm_ref_u/si_ref_u;
	float lbm_nu = units.nu(si_nu);
	float omega[3] = {0.0f, 0.0f, 0.0f};
	auto update_coriolis = [&]() { // FX/setup.cpp:3800-3823
		if(!c.enable_coriolis) return;
		const float lat = 0.5f*(c.cut_lat[0]+c.cut_lat[1]);
		const float Om = 7.292115e-5f, deg2rad = 3.14159265358979323846f/180.0f, lat_rad = lat*deg2rad;
		const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u);
		omega[0] = 0.0f*dt_si; omega[1] = Om*cosf(lat_rad)*dt_si; omega[2] = Om*sinf(lat_rad)*dt_si;
	};
	SolverGlobals& G = solver_globals();
	G.fp16c = c.fp16c; G.device = c.device; G.devices = c.devices; G.kernel = c.kernel;
	auto buffer_face_id_from_bc = [](const string& bc) { return bc=="-x" ? 1 : bc=="+x" ? 2 : bc=="-y" ? 3 : bc=="+y" ? 4 : 0; };
	auto bc_from_dir = [](const float dx, const float dy) -> string { if(fabsf(dx)>=fabsf(dy)) return dx>=0.0f ? "+x" : "-x"; return dy>=0.0f ? "+y" : "-y"; };
	auto update_buffer_nudging = [&](const string& bc) { // FX/setup.cpp:3844-3856
		G.buffer_downstream_face_id = buffer_face_id_from_bc(bc);
		const uint min_dim = std::min(Nx, std::min(Ny, Nz)), max_nbuf = std::max(1u, min_dim/4u);
		int nbuf = (int)std::lround(c.buffer_thickness_m/c.cell_m);
		if(nbuf<1) nbuf = 1; if((uint)nbuf>max_nbuf) nbuf = (int)max_nbuf;
		G.buffer_n_cells = nbuf;
		const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u);
		G.buffer_inv_tau_lbmu = c.buffer_tau_s>0.0f ? dt_si/c.buffer_tau_s : 0.0f;
		G.buffer_nudging_active = c.enable_buffer_nudging&&c.buffer_tau_s>0.0f;
		G.buffer_nudge_vertical = c.buffer_nudge_vertical;
	};
	auto update_top_sponge = [&]() { // FX/setup.cpp:3867-3881
		int ns = std::max(sponge_cells_cfg, 1);
		if(Nz>2u) ns = std::min(ns, (int)Nz-2);
		G.sponge_n_cells = ns;
		const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u);
		G.sponge_inv_tau_lbmu = c.sponge_tau_s>0.0f ? dt_si/c.sponge_tau_s : 0.0f;
		G.top_sponge_active = top_sponge_grid_extend&&c.sponge_tau_s>0.0f&&c.sponge_ref_mode==0&&Nz_core>2u;
	};
	if(c.nwp_mode) {
		println("| SI Reference U  | "+alignl(7u, fmtf(si_ref_u))+alignl(50u, "m/s")+" |");
		println("| LBM Reference U | "+alignl(7u, fmtf(lbm_ref_u))+alignl(50u, "(Nondimensionalized)")+" |");
		println("| Temp Reference  | "+alignr(57u, fmtf(temperature_ref_kelvin)+(temperature_ref_adaptive ? " K (auto center of input Tmin/Tmax)" : " K (default)"))+" |");
		println("| Temp Scale      | "+alignr(57u, fmtf(temperature_scale_kelvin)+(temperature_scale_adaptive ? " K per 1.0 T_lbm (auto from input range)" : " K per 1.0 T_lbm (default)"))+" |");
		println("| Thermal alpha   | "+alignr(57u, to_string_fd(lbm_alpha, 8u))+" |");
		println("| Thermal tau_T   | "+alignr(57u, to_string_fd(2.0f*lbm_alpha+0.5f, 8u))+" |");
		println("| Thermal beta    | "+alignr(57u, c.buoyancy ? to_string_fd(units.beta(1.0f/temperature_ref_kelvin), 8u) : string("0 (disabled by buoyancy=false)"))+" |");
		if(c.buoyancy) println("| Thermal note    | temperature is advected as a passive scalar: the solver's volume force is zero, as in the reference |");
	}
	update_coriolis(); update_buffer_nudging(c.nwp_mode ? c.downstream_bc : string("+y")); update_top_sponge();
	if(c.nwp_mode&&c.enable_coriolis) { print_kv_row("Coriolis", "enabled. center(lon,lat)=("+to_string_fd(0.5f*(c.cut_lon[0]+c.cut_lon[1]), 6u)+", "+to_string_fd(0.5f*(c.cut_lat[0]+c.cut_lat[1]), 6u)+") deg"); print_kv_row("", "Omega(lbmu)=("+to_string_fd(omega[0], 8u)+", "+to_string_fd(omega[1], 8u)+", "+to_string_fd(omega[2], 8u)+") per step"); }
	else if(c.nwp_mode) print_kv_row("Coriolis", "disabled by 'coriolis_term' setting in .luw");
	print_kv_row("Buffer nudging", G.buffer_nudging_active ? (c.nwp_mode ? "enabled" : "enabled (downstream face auto by angle)") : "disabled");
	print_kv_row("", "Nbuf="+to_string_u((ulong)G.buffer_n_cells)+" cells, tau_s="+to_string_fd(c.buffer_tau_s, 6u)+" s");
	print_kv_row("", "inv_tau_lbmu="+to_string_fd(G.buffer_inv_tau_lbmu, 8u)+", downstream_face_id="+(c.nwp_mode ? to_string_u((ulong)G.buffer_downstream_face_id) : string("auto"))+", nudge_vertical="+to_string_u((ulong)G.buffer_nudge_vertical));
	print_kv_row("Top sponge", G.top_sponge_active ? "enabled" : "disabled");
	print_kv_row("", "Nsponge="+to_string_u((ulong)G.sponge_n_cells)+" cells, tau_s="+to_string_fd(c.sponge_tau_s, 6u)+" s");
	print_kv_row("", "inv_tau_lbmu="+to_string_fd(G.sponge_inv_tau_lbmu, 8u)+", ref_mode="+std::to_string(c.sponge_ref_mode));
	if(G.top_sponge_active) print_kv_row("", "core_top_z="+to_string_u(Nz_core-1u)+", side_ref_cap_z="+std::to_string(side_ref_z_cap));

	float T_bc_min = 1.0f, T_bc_max = 1.0f;
	if(c.nwp_mode) { // FX/setup.cpp:3944-3987
		if(surf.has_T) {
			println("| T column        | detected ("+to_string_u(surf.rows_T)+" rows)                               |");
			println("| CSV T range SI  | "+alignr(24u, fmtf(surf.tmin))+" to "+alignl(16u, fmtf(surf.tmax))+" K |");
			println(c.buoyancy ? "| Temperature BC  | enabled from CSV T (Kelvin -> nondimensionalized)               |" : "| Temperature BC  | buoyancy=false, ignore T column                                 |");
		} else println("| T column        | not found, keep legacy velocity-only boundary behavior           |");
		if(use_temperature_bc) {
			ulong out_of_range = 0ull; for(const SurfSample& r : surf.rows) if(r.T<223.15f||r.T>343.15f) out_of_range++;
			if(out_of_range>0ull) println("| WARNING: "+to_string_u(out_of_range)+" temperature samples are outside [-50C, 70C].                |");
			T_bc_min = units.T(surf.tmin); T_bc_max = units.T(surf.tmax); if(T_bc_min>T_bc_max) std::swap(T_bc_min, T_bc_max);
		}
	}
	if(c.sizing_only) { println(hr_plain()); return 0; }
	// ---- geometry, FX/setup.cpp:4001-4093
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
			for(const string& suf : order) { for(const string& n : names) if(ends(n, suf)) { stl_path = (dir/n).string(); break; } if(!stl_path.empty()) break; }
		}
		if(stl_path.empty()) fatal("ERROR: no STL file under "+dir.string());
	}
	Mesh mesh;
	if(!read_stl(stl_path, mesh)) fatal("ERROR: failed to load STL");
	println("| Info: Loading \""+stl_path+"\" with "+to_string_u(mesh.n)+" triangles.");
	g_progress.emit("load_stl", "Loading STL", stl_path+" ("+to_string_u(mesh.n)+" triangles)", 0ll, 1ll, false); // FX/utilities.hpp:4850-4887
	g_progress.emit("load_stl", "Loading STL", stl_path+" loaded", 1ll, 1ll, false);
	const float stl_min[3] = {mesh.pmin[0], mesh.pmin[1], mesh.pmin[2]}, stl_max[3] = {mesh.pmax[0], mesh.pmax[1], mesh.pmax[2]};
	float vtk_origin_shift[3];
	{ const uint NN[3] = {Nx, Ny, Nz}; for(int k=0; k<3; k++) vtk_origin_shift[k] = stl_min[k]-units.si_x(0.5f-0.5f*(float)NN[k]); }
	const float scale_geom = units.x(c.si_x)/(stl_max[0]-stl_min[0]);
	mesh_scale_translate(mesh, scale_geom);
	print_kv_row("Geometry STL", stl_path);
	print_kv_row("STL bounds SI", "x=["+to_string_fd(stl_min[0], 3u)+", "+to_string_fd(stl_max[0], 3u)+"], y=["+to_string_fd(stl_min[1], 3u)+", "+to_string_fd(stl_max[1], 3u)+"], z=["+to_string_fd(stl_min[2], 3u)+", "+to_string_fd(stl_max[2], 3u)+"]");
	print_kv_row("Geometry", "scaled by "+to_string_fd(scale_geom, 4u)+", ready for voxelization");
	DemPoints dem;
	if(c.profile_mode) { // FX/setup.cpp:4095-4113
		dem = read_dem_csv(c.parent+"/proj_temp/interpolated_dem.csv");
		if(!dem.x.empty()) {
			print_kv_row("Terrain DEM", "Loaded "+to_string_u(dem.x.size())+" points from interpolated_dem.csv");
			print_kv_row("DEM bounds SI", "x=["+to_string_fd(dem.xmin, 3u)+", "+to_string_fd(dem.xmax, 3u)+"], y=["+to_string_fd(dem.ymin, 3u)+", "+to_string_fd(dem.ymax, 3u)+"], elev=["+to_string_fd(dem.emin, 3u)+", "+to_string_fd(dem.emax, 3u)+"]");
		} else print_kv_row("Terrain DEM", "interpolated_dem.csv not found or empty, fallback to flat ground");
	}

	// ---- profile table, FX/setup.cpp:5777-5912
	const float origin_z = 0.5f-0.5f*(float)Nz;
	const float flat_ground = origin_z+units.x(c.z_si_offset);
	std::vector<float> prof_lbmu;
	const float profile_dz = 0.1f;
	GroundPlane2D ground_plane; bool use_dem_ground = false; float ground_z_min = flat_ground, ground_z_max = flat_ground;
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
			if(use_dem_ground) println("| Terrain DEM     | profile ground enabled. z(SI) range "+to_string_fd(units.si_x(ground_z_min-origin_z), 3u)+" .. "+to_string_fd(units.si_x(ground_z_max-origin_z), 3u)+" m |");
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
		println("| Profile table   | local-terrain AGL top="+to_string_fd(table_top, 3u)+" m, core_top="+to_string_fd(core_top_si, 3u)+" m, solver_top="+to_string_fd(solver_top_si, 3u)+" m |");
		println("| Profile ground  | z(SI) min/max="+to_string_fd(ground_min_si, 3u)+" / "+to_string_fd(ground_max_si, 3u)+" m |");
		println("| Profile U range | "+alignr(24u, fmtf(umin))+" to "+alignl(16u, fmtf(umax))+" m/s |");
	}
	auto profile_speed = [&](const float pos_z, const float ground_z) -> float { // FX/setup.cpp:5901-5912
		if(pos_z<=ground_z) return 0.0f;
		const float inv_dz = 1.0f/profile_dz;
		const uint last = (uint)(prof_lbmu.size()-1u);
		float z_agl = units.si_x(pos_z-ground_z);
		if(z_agl<0.0f) z_agl = 0.0f;
		long idx = std::lround(z_agl*inv_dz);
		if(idx<0l) idx = 0l;
		return prof_lbmu[std::min((uint)idx, last)];
	};

	// ---- cases
	struct Case { float inflow_si, angle_deg; };
	std::vector<Case> cases;
	if(c.nwp_mode) cases.push_back({0.0f, 0.0f});
	else if(c.dataset_mode) { for(const float in : c.inflow_list) for(const float an : c.angle_list) cases.push_back({in, an}); }
	else for(const float an : c.angle_list) cases.push_back({0.0f, an});
	// LBM::LBM makes the resolution equally divisible by the domains (FX/lbm.cpp:1058-1060): everything up to here used lbm_N
	// (units, profile table, mesh transform, buffer / sponge sizes), everything inside a case uses the lattice of the LBM object
	const uint lbmN[3] = {Nx, Ny, Nz};
	uint case_index = 0u;
	for(const Case& cs : cases) {
		++case_index;
		const uint Nx = (lbmN[0]/c.Dx)*c.Dx, Ny = (lbmN[1]/c.Dy)*c.Dy, Nz = (lbmN[2]/c.Dz)*c.Dz;
		const ulong N = (ulong)Nx*Ny*Nz;
		auto pos_z_of = [&](const uint z) { return (float)z-0.5f*(float)Nz+0.5f; }; // lbm.position(x, y, z).z
		const float deg2rad = 3.14159265358979323846f/180.0f, angle_rad = cs.angle_deg*deg2rad;
		float dir_x = -sinf(angle_rad), dir_y = -cosf(angle_rad);
		float uin[3] = {0.0f, 0.0f, 0.0f};
		string vtk_prefix;
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
			println("| Profile case    | "+alignr(57u, to_string_u(case_index)+"/"+to_string_u(cases.size())+" (remaining "+to_string_u(cases.size()-case_index)+")")+" |");
			println("| Angle           | "+alignr(57u, format_tag(cs.angle_deg)+" deg")+" |");
		}
		if(!c.nwp_mode) println("| SI Reference U  | "+alignr(57u, format_tag(si_ref_u)+" m/s")+" |");
		const string case_bc = c.nwp_mode ? c.downstream_bc : bc_from_dir(dir_x, dir_y);
		print_section_title("DEVICE INFORMATION");
		if(!c.nwp_mode) print_kv_row("Downstream BC", case_bc+(c.dataset_mode ? " (auto from batch angle)" : " (auto from profile angle)"));
		update_buffer_nudging(case_bc); update_top_sponge();

		phase_mark("deck, sizing, mesh, profile");
		// host state of this case: the LBM object's global host arrays (lbm.flags[n], lbm.u.x[n], lbm.T[n]) or, without a GPU, plain vectors
		G.temperature = use_temperature_bc; // the thermal lattice runs exactly when the reference writes T outputs
		std::vector<uchar> flags_store; std::vector<float> u_store, T_store;
		std::unique_ptr<LBM> lbm_p;
		ulong nvox = 0ull;
		if(c.dry_run) { flags_store.assign(N, 0u); u_store.assign(3ull*N, 0.0f); if(use_temperature_bc) T_store.assign(N, 1.0f); }
		else {
			const uint Dn = c.Dx*c.Dy*c.Dz; // LBM_Domain's constructor reports per device (FX/lbm.cpp:265-280); here all domains are built in one call
			g_progress.emit("gpu_memory", "Configuring GPU memory", "Allocating CFD buffers on "+to_string_u(Dn)+" device(s)", 0ll, (long long)Dn, false);
			lbm_p.reset(new LBM(uint3(lbmN[0], lbmN[1], lbmN[2]), c.Dx, c.Dy, c.Dz, lbm_nu, 0.0f, 0.0f, 0.0f, 0.0f, lbm_alpha, 0.0f)); // FX/setup.cpp:4935,5720,6018
			g_progress.emit("gpu_memory", "Configuring GPU memory", to_string_u(Dn)+" device(s): buffers ready", (long long)Dn, (long long)Dn, false);
		}
		if(lbm_p&&lbm_p->get_D()>1u) {
			string devs; for(uint d=0u; d<lbm_p->get_D(); d++) { int dv = 0; luw_check(luw_group_domain_info(lbm_p->group(), d, nullptr, nullptr, &dv)); devs += (d ? "," : "")+to_string_u((ulong)dv); }
			print_kv_row("Domains", to_string_u(lbm_p->get_D())+" domains ("+to_string_u(c.Dx)+"x"+to_string_u(c.Dy)+"x"+to_string_u(c.Dz)+") of "+to_string_u(Nx/c.Dx)+"x"+to_string_u(Ny/c.Dy)+"x"+to_string_u(Nz/c.Dz)+" cells on HIP devices "+devs);
			print_kv_row("", string("halo faces: ")+(luw_group_direct_peer_stores(lbm_p->group()) ? "peer stores of the pack kernels (xGMI)" : "hipMemcpyPeerAsync")+(luw_group_overlaps(lbm_p->group()) ? ", overlapped with the interior" : ", after the whole-box kernel"));
			// like the reference, nudging / sponge act only inside domains that own the face (FX/kernel.cpp:1537-1541,1598): say so when a zone is cut
			const uint bz = G.buffer_nudging_active ? (uint)G.buffer_n_cells : 0u, sz = G.top_sponge_active ? (uint)G.sponge_n_cells : 0u;
			if((c.Dz>1u&&std::max(bz, sz)+1u>Nz/c.Dz)||(c.Dy>1u&&bz+1u>Ny/c.Dy)||(c.Dx>1u&&bz+1u>Nx/c.Dx))
				println("| WARNING: a nudging / sponge zone is thicker than a domain: cells of the zone in domains that do not own the face get no forcing (as in the reference). |");
		}
		uchar* const flags = c.dry_run ? flags_store.data() : lbm_p->flags.data<uchar>();
		float* const u = c.dry_run ? u_store.data() : lbm_p->u.data<float>();
		float* const Tcell = !use_temperature_bc ? nullptr : c.dry_run ? T_store.data() : lbm_p->T.data<float>(); // lbm.T, pre-filled with 1 (FX/lbm.cpp:304)
		if(c.dry_run) nvox = voxelize_z(mesh, Nx, Ny, Nz, flags_store); // no GPU: host restatement of the kernel
		else { // lbm.voxelize_mesh_on_device(mesh), FX/setup.cpp:4089: every domain voxelises its own box
			const long long Dn = (long long)lbm_p->get_D(); // FX/lbm.cpp:1413-1418,1593-1598
			g_progress.emit("voxelization", "Voxelizing geometry", to_string_u(mesh.n)+" triangles across "+to_string_u((ulong)Dn)+" domain(s)", 0ll, Dn, false);
			lbm_p->voxelize_mesh_on_device(mesh.n, mesh.p0.data(), mesh.p1.data(), mesh.p2.data(), mesh.pmin, mesh.pmax, TYPE_S);
			g_progress.emit("voxelization", "Voxelizing geometry", "Finished domain "+to_string_u((ulong)Dn)+"/"+to_string_u((ulong)Dn), Dn, Dn, false);
			for(ulong n=0ull; n<N; n++) nvox += (flags[n]&TYPE_S)!=0u;
		}
		phase_mark("solver create + voxelise");
		println("| Info: Voxelized cells (whole domain global, no halos): solid = "+to_string_u(nvox)+", fluid = "+to_string_u(N-nvox)+", total = "+to_string_u(N)+".");
		println("| Voxelization done.                                                          |");
		print_section_title("BUILD BOUNDARY CONDITIONS");
		auto is_downstream = [&](const uint x, const uint y) { return case_bc=="+y" ? y==Ny-1u : case_bc=="-y" ? y==0u : case_bc=="+x" ? x==Nx-1u : case_bc=="-x" ? x==0u : false; };
		std::atomic<ulong> mapped{0ull}, terrain_solid{0ull}, outlet{0ull};
		std::vector<float> ground_xy; // terrain height per column (profile mode with a DEM), else flat
		auto ground_at = [&](const ulong id) { return ground_xy.empty() ? flat_ground : ground_xy[id]; };
		HostLattice HL; HL.Nx = Nx; HL.Ny = Ny; HL.Nz = Nz; HL.flags = flags; HL.u = u;
		auto report_flux = [&](const FluxReport& fr) { // FX/fluxcorrection.cpp:180-192
			g_progress.emit("flux_correction", "Flux correction", "avg dU = "+to_string_dd(fr.delta, 3u)+" m/s, net after = "+to_string_dd(fr.net_after, 3u), 1ll, 1ll, false);
			println("| Flux correction | S_in="+to_string_dd(fr.S_in, 3u)+", S_out="+to_string_dd(fr.S_out, 3u)+", net_before="+to_string_dd(fr.net_before, 3u)+" |");
			println("| Flux correction | avg_dU="+to_string_dd(fr.delta, 3u)+" m/s, corrected="+to_string_u(fr.corrected)+", net_after="+to_string_dd(fr.net_after, 3u)+" |");
			println("| Flux correction | per-face dU: Xn="+to_string_dd(fr.face_avg[0], 3u)+", Xp="+to_string_dd(fr.face_avg[1], 3u)+", Yn="+to_string_dd(fr.face_avg[2], 3u)+", Yp="+to_string_dd(fr.face_avg[3], 3u)+", Zp="+to_string_dd(fr.face_avg[4], 3u)+" m/s |");
		};
		g_progress.emit("interface_interpolation", "Interface interpolation", c.nwp_mode ? (surf.has_patch ? "Patch-driven 2D boundary mapping" : c.use_high_order ? "High-order boundary interpolation" : "Nearest-sample boundary interpolation") : c.profile_mode ? "Applying profile boundary conditions" : "Applying uniform inflow boundary conditions", 0ll, 1ll, true);
		if(c.nwp_mode) { // FX/setup.cpp:4931-5632
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
				if(pc.terrain_clipped>0ull) println("| Terrain clip    | below-terrain cells forced to solid: "+to_string_u(pc.terrain_clipped)+"                    |");
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
				println("| Temperature BC  | summary ["+tag+"]: TYPE_T total="+to_string_u(ts.total)+", solid="+to_string_u(ts.solid)+", fluid="+to_string_u(ts.fluid)+"            |");
				if(ts.solid>0ull) println("| Temperature BC  | solid TYPE_T range SI: "+fmtf(units.si_T(ts.smin))+" .. "+fmtf(units.si_T(ts.smax))+" K                      |");
				if(ts.fluid>0ull) println("| Temperature BC  | fluid TYPE_T range SI: "+fmtf(units.si_T(ts.fmin))+" .. "+fmtf(units.si_T(ts.fmax))+" K                      |");
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
						if(cntp>0ull) println("| T patch         | "+string(patch_name(pt))+": n="+to_string_u(cntp)+", SI "+fmtf(units.si_T(mn))+" .. "+fmtf(units.si_T(mx))+" K                    |");
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
						apply_cloud_temperature(HL, Tcell, case_bc, c.downstream_open_face, true, units.x(c.z_si_offset)+z0_lbmu, T_bc_min, T_bc_max, [k](const V3& p) { return k->eval(p).x; }, tc);
						println("| Temperature BC  | per-face interpolation done on 5 boundary surfaces        |");
						println("| Temperature BC  | mapped "+to_string_u(tc.mapped)+" cells (high-order)      |");
					} else {
						t_tag = "low-order";
						const SampleCloud* cl = &tcloud;
						apply_cloud_temperature(HL, Tcell, case_bc, c.downstream_open_face, false, z0_lbmu+units.x(c.z_si_offset), T_bc_min, T_bc_max, [cl](const V3& p) { return nearest_sample_velocity(*cl, p).x; }, tc);
						println("| Temperature BC  | mapped "+to_string_u(tc.mapped)+" cells (low-order)                     |");
					}
				}
				if(surf.has_patch) { // ground temperature plane from patch 0, :5019-5072 (in every boundary mode when the CSV carries patches)
					std::vector<float> gx, gy, gt;
					for(const SurfSample& q : smp) if(q.patch==PATCH_BOTTOM) { gx.push_back(q.p.x); gy.push_back(q.p.y); gt.push_back(q.T); }
					GroundPlane2D tplane; tplane.build(gx, gy, gt, 1.0f);
					if(tplane.has_samples()) {
						println("| Ground T plane  | enabled from patch=0 ("+to_string_u(gt.size())+" samples, grid "+to_string_u(tplane.nx())+"x"+to_string_u(tplane.ny())+", mode="+(tplane.structured() ? string("2D bilinear") : string("2D nearest"))+") |");
						apply_ground_temperature(HL, Tcell, tplane, T_bc_min, T_bc_max, tc);
						println("| Ground T plane  | mapped "+to_string_u(tc.ground_cells)+" solid cells, unique (x,y)="+to_string_u(tc.ground_columns)+" ["+t_tag+"]                                |");
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
		} else if(c.profile_mode) { // FX/setup.cpp:5914-5995,6043-6078
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
				println("| Terrain ground  | mapped z(SI) range "+to_string_fd(units.si_x(gmin-origin_z), 3u)+" .. "+to_string_fd(units.si_x(gmax-origin_z), 3u)+" m                     |");
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
			if(terrain_solid.load()>0ull) println("|                 | boundary cells below local terrain -> solid: "+to_string_u(terrain_solid.load())+"                     |");
		} else { // FX/setup.cpp:5655-5688
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
		if(!c.nwp_mode) { g_progress.emit("interface_interpolation", "Interface interpolation", c.profile_mode ? "Profile boundary conditions completed" : "Boundary conditions completed", 1ll, 1ll, false); print_kv_row("Boundary init", "complete. Time: ["+now_str()+"]"); }
		if(c.profile_mode) { // FX/setup.cpp:6087-6119
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
		VkTables vk; bool vk_on = false;
		if(c.vk_enable) { // make_vk_runtime_config + VonKarmanInletUpdater::initialize, FX/setup.cpp:3762-3799,417-534
			VkRuntimeConfig vc;
			vc.ti = c.vk_ti; vc.sigma_lbm = c.vk_sigma_si*units.unit_s/units.unit_m; vc.L_lbm = units.x(c.vk_L_si);
			vc.nmodes = c.vk_nmodes; vc.seed = c.vk_seed; vc.update_stride = c.vk_stride; vc.uc_mode = c.vk_uc;
			vc.same_realization_all_faces = c.vk_same; vc.stride_interpolation = c.vk_interp; vc.inflow_only = c.vk_inflow_only;
			vc.face_mode = vk_resolve_face_mode(c.vk_face_mode, c.vk_inflow_only);
			for(int k=0; k<3; k++) vc.aniso[k] = c.vk_aniso[k];
			vc.downstream_face_id = case_bc=="-x" ? 0 : case_bc=="+x" ? 1 : case_bc=="-y" ? 2 : case_bc=="+y" ? 3 : -1;
			if(!(vc.L_lbm>0.0f)) println("| WARNING: vk_inlet_l converts to non-positive LBM value. Disabled.            |");
			else vk_on = vk_build_tables(vc, Nx, Ny, Nz, flags, u, vk, [](const string& l) { println(l); });
			if(!vk_on) println(c.profile_mode ? "| VK inlet        | profile case: no valid inflow faces.                       |" : "| VK inlet        | dataset case: no valid inflow faces.                       |");
			if(vk_on&&!c.dump_vk.empty()&&case_index==1u) {
				std::ofstream vf(c.dump_vk, std::ios::binary); const uint64_t hdr[2] = {vk.point_count, vk.mode_count};
				vf.write((const char*)hdr, 16); vf.write((const char*)vk.point_cell.data(), (std::streamsize)(8ull*vk.point_count)); vf.write((const char*)vk.point_face.data(), (std::streamsize)vk.point_count);
				vf.write((const char*)vk.point_data.data(), (std::streamsize)(28ull*vk.point_count)); vf.write((const char*)vk.mode_data.data(), (std::streamsize)(200ull*vk.mode_count));
			}
		}
		if(!c.dump_setup.empty()&&case_index==1u) { // raw initial state for tests: header (Nx,Ny,Nz,Nz_core as u32; nu, si_u_factor, si_rho_factor as f32) + flags + u + rho(=1)
			std::ofstream df(c.dump_setup, std::ios::binary);
			const uint hdr[4] = {Nx, Ny, Nz, Nz_core}; const float fh[8] = {lbm_nu, units.si_u(1.0f), units.si_rho(1.0f), G.buffer_inv_tau_lbmu, G.sponge_inv_tau_lbmu, scale_geom, omega[1], omega[2]};
			const int ih[8] = {G.buffer_nudging_active, G.buffer_n_cells, G.buffer_downstream_face_id, G.buffer_nudge_vertical, G.top_sponge_active, G.sponge_n_cells, (int)nvox, (int)mapped.load()};
			df.write((const char*)hdr, 16); df.write((const char*)fh, 32); df.write((const char*)ih, 32);
			df.write((const char*)flags, (std::streamsize)N); df.write((const char*)u, (std::streamsize)(12ull*N));
			if(use_temperature_bc) { const float th[2] = {units.unit_K, units.unit_K_offset}; df.write("TEMP", 4); df.write((const char*)th, 8); df.write((const char*)Tcell, (std::streamsize)(4ull*N)); } // optional trailer: T in lattice units
		}
		const ulong total_steps = (c.run_nstep_override>0ull ? c.run_nstep_override : 20001ull)+(ulong)c.research_output_steps;
		const ulong unsteady = (ulong)c.unsteady_output_interval;
		const string results_vtk_dir = c.parent+"/RESULTS/vtk/";
		const string vtk_dir = results_vtk_dir+vtk_prefix+c.datetime+"_raw_";
		const uint Nz_out = (top_sponge_grid_extend&&Nz_core<Nz) ? Nz_core : Nz;
		VtkGeom geom{Nx, Ny, Nz, Nz_out, units.si_x(1.0f), {0, 0, 0}};
		{ const uint NN[3] = {Nx, Ny, Nz}; for(int k=0; k<3; k++) geom.origin[k] = geom.spacing*(0.5f-0.5f*(float)NN[k])+vtk_origin_shift[k]; }
		const ulong avg_window = c.purge_avg_steps>0u ? std::min((ulong)c.purge_avg_steps, total_steps) : 0ull;
		const ulong avg_stride = std::max((ulong)1u, (ulong)c.purge_avg_stride);
		const ulong avg_start_t = avg_window>0ull ? total_steps-avg_window+1ull : ~0ull;
		// probes, FX/setup.cpp:4269-4395
		const double dt_si_d = (double)c.cell_m*((double)lbm_ref_u/(double)si_ref_u);
		const ulong probe_window = probe_requests.empty() ? 0ull : (c.probes_output_defined&&c.probes_output_steps>0u) ? std::min((ulong)c.probes_output_steps, total_steps)
			: (c.purge_avg_steps>0u||c.research_output_steps>0u) ? std::min((ulong)std::max(c.purge_avg_steps, c.research_output_steps), total_steps) : total_steps;
		const ulong probe_start_t = probe_window>0ull ? total_steps-probe_window+1ull : ~0ull;
		std::vector<ProbeColumn> probes; std::vector<uint64_t> probe_cells;
		if(!probe_requests.empty()) {
			if(!probe_geo.valid) print_kv_row("Probes", "disabled: geographic mapping is unavailable");
			else {
				std::vector<string> used;
				for(const ProbeRequest& rq : probe_requests) {
					ProbeColumn pc; pc.req = rq; string why;
					bool ok = resolve_probe_xy(rq, probe_geo, Nx, Ny, c.cell_m, c.si_x, c.si_y, pc.x, pc.y, why);
					if(ok) { for(uint z=0u; z<Nz; ++z) if((flags[(ulong)pc.x+((ulong)pc.y+(ulong)z*Ny)*Nx]&TYPE_S)==0u) pc.z.push_back(z); if(pc.z.empty()) { ok = false; why = "resolved column has no fluid cell"; } }
					if(!ok) { println("| WARNING: probe '"+rq.raw+"' ignored: "+why+"                |"); continue; }
					for(const uint z : pc.z) pc.height_si.push_back((float)(((double)z-(double)pc.z.front()+0.5)*(double)c.cell_m));
					string stem = probe_stem(rq, probe_geo, vtk_prefix);
					if(std::find(used.begin(), used.end(), stem)!=used.end()) { uint k = 2u; string u2 = stem; while(std::find(used.begin(), used.end(), u2)!=used.end()) u2 = stem+"_"+to_string_u(k++); stem = u2; }
					used.push_back(stem); pc.stem = stem;
					probes.push_back(std::move(pc));
				}
				if(probes.empty()) print_kv_row("Probes", "0 valid probe column after geometry/domain checks");
				else {
					print_kv_row("Probes", to_string_u(probes.size())+" active, "+(probe_window>=total_steps ? string("entire run") : "last "+to_string_u(probe_window)+" step(s)"));
					bool first = true;
					for(const ProbeColumn& pc : probes) { print_kv_row(first ? "Probe cell" : "", pc.stem+" -> ("+to_string_u(pc.x)+","+to_string_u(pc.y)+"), levels="+to_string_u(pc.z.size())); first = false; for(const uint z : pc.z) probe_cells.push_back((uint64_t)pc.x+((uint64_t)pc.y+(uint64_t)z*Ny)*Nx); }
				}
			}
		}
		if(c.dry_run) continue;

		phase_mark("boundary conditions");
		// ---- run_lbm, FX/setup.cpp:4117-4911
		LBM& lbm = *lbm_p;
		lbm.set_coriolis(omega[0], omega[1], omega[2]);
		if(vk_on) lbm.vk_inlet_attach(vk.point_count, vk.mode_count, vk.point_cell.data(), vk.point_face.data(), vk.point_data.data(), vk.mode_data.data(), c.vk_stride, c.vk_interp ? 1 : 0);
		print_section_title("LBM SOLVER INFORMATION");
		if(use_temperature_bc) print_kv_row("Export mode", "include temperature T field in Kelvin");
		if(Nz_out<Nz) print_kv_row("VTK z output", "core Nz="+to_string_u(Nz_out)+" of solver Nz="+to_string_u(Nz)+" (top sponge omitted)");
		print_kv_row("Run steps", to_string_u(total_steps)+(c.run_nstep_override>0ull ? " (run_nstep override)" : " (default)"));
		if(avg_window>0ull) { print_kv_row("Avg stride", "sample every "+to_string_u(avg_stride)+" step(s) in purge_avg window (on-device accumulation)"); lbm.stats_reset(); }
		if(!probe_cells.empty()) lbm.gather_attach((uint32_t)probe_cells.size(), probe_cells.data());
		std::vector<float> probe_buf(3u*probe_cells.size());
		lbm.run(0u, total_steps);
		phase_mark("upload + initialise");
		print_section_title("SOLVER START");
		// ---- the time loop.  The device runs batches of steps without any host round trip inside; between batches the host looks at the
		// clock, refreshes the running row / the GUI's progress line and handles whatever must be observed at that step (unsteady output,
		// probe samples).  A batch ends at the next such step, and otherwise after about a quarter of a second of work.
		const double bytes_per_cell = (c.fp16c ? 77.0 : 153.0)+(use_temperature_bc ? (c.fp16c ? 32.0 : 60.0) : 0.0); // DDFs + flags (+ thermal lattice), DESIGN.md section 5
		StepRateMeter meter; meter.configure(total_steps, avg_window>0ull ? avg_start_t : ~0ull);
		const bool console_row = !g_progress.gui(); // FX/info.cpp:225: the GUI gets protocol lines instead of the table
		if(console_row) { println(ProgressTable::top()); println(ProgressTable::header()); }
		auto last_gui = std::chrono::steady_clock::time_point{};
		auto show_progress = [&](const bool force) {
			const ulong t = lbm.get_t();
			if(console_row) reprint_row(ProgressTable::row(N, bytes_per_cell, meter, t, total_steps));
			const auto now = std::chrono::steady_clock::now();
			if(!g_progress.gui()||(!force&&t<total_steps&&last_gui.time_since_epoch().count()!=0&&now-last_gui<std::chrono::milliseconds(120))) return; // FX/setup.cpp:4144-4164
			last_gui = now;
			g_progress.emit("solve", "Solving CFD", to_string_u(t)+"/"+to_string_u(total_steps)+" steps | "+to_string_fd((float)meter.steps_per_second(t), 3u)+" Steps/s | ETA "+clock_text(meter.remaining_seconds(t)), (long long)t, (long long)total_steps, false);
		};
		auto note_saved = [&](const std::vector<string>& files) { // flush_vtk_saved_files, FX/setup.cpp:4192-4218
			if(files.empty()) return;
			if(console_row) std::cout << "\r" << string(CONSOLE_WIDTH, ' ') << "\r";
			bool first = true; for(const string& f : files) { print_kv_row(first ? "VTK file" : "", f+" saved"); first = false; }
			g_progress.emit("save", "Saving results", files.size()==1u ? files.back() : to_string_u(files.size())+" files saved; last: "+files.back(), (long long)files.size(), (long long)files.size(), false);
		};
		const auto t_start = std::chrono::steady_clock::now();
		ulong last_u_vtk_t = ~0ull;
		ulong batch_cap = 16ull; // first batch: the reference's 16-step "Normal benchmark" (FX/setup.cpp:4799-4841) doubles as the speed sample
		g_progress.emit("speed_estimate", "Estimating solve speed", "Benchmarking normal LBM solver", 0ll, (long long)std::min<ulong>(batch_cap, total_steps), false);
		bool speed_reported = false;
		while(lbm.get_t()<total_steps) {
			// the next step at which something must be observed (unsteady output / probe sample / end); fields are written by the last step of each batch
			ulong next = std::min(total_steps, lbm.get_t()+batch_cap);
			if(unsteady>0ull) next = std::min(next, (ulong)((lbm.get_t()/unsteady+1ull)*unsteady));
			if(!probes.empty()) next = std::min(next, std::max((ulong)(lbm.get_t()+1ull), probe_start_t)); // every step of the probe window is observed
			if(avg_window>0ull&&lbm.get_t()+1ull<avg_start_t) next = std::min<ulong>(next, avg_start_t-(ulong)1u); // a batch belongs to ONE stage of the time estimate
			// statistics samples that fall into (t, next] ride along (run_sampled): first sample s, then every avg_stride-th step
			ulong first_sample = 0ull;
			if(avg_window>0ull) { const ulong t1 = lbm.get_t()+1ull; ulong sm = std::max(t1, avg_start_t); const ulong off = (sm-avg_start_t)%avg_stride; if(off!=0ull) sm += avg_stride-off; if(sm<=next) first_sample = sm; }
			const ulong nsteps = next-lbm.get_t();
			const auto b0 = std::chrono::steady_clock::now();
			if(first_sample>0ull) lbm.run_sampled(nsteps, first_sample-lbm.get_t(), avg_stride);
			else lbm.run(nsteps, total_steps);
			const double bsec = std::chrono::duration<double>(std::chrono::steady_clock::now()-b0).count();
			const ulong t = lbm.get_t();
			meter.add_batch(t, nsteps, bsec);
			if(!speed_reported) { speed_reported = true; g_progress.emit("speed_estimate", "Estimating solve speed", "Benchmarking normal LBM solver step "+to_string_u(nsteps)+"/"+to_string_u(nsteps), (long long)nsteps, (long long)nsteps, false); }
			batch_cap = std::max<ulong>((ulong)16u, std::min<ulong>((ulong)1u<<20, (ulong)(0.25*meter.steps_per_second(t))));
			show_progress(false); // about 0.25 s of work per batch
			if(unsteady>0ull&&t%unsteady==0ull) {
				const string fn = default_filename(vtk_dir, "u", t);
				if(host_vtk_path()) { lbm.u.read_from_device(); write_field_vtk(fn, geom, lbm.u.data<float>(), 3u, units.si_u(1.0f)); }
				else write_device_field_vtk(lbm, fn, geom, LUW_EXPORT_U, 3u, units.si_u(1.0f));
				note_saved({fn}); last_u_vtk_t = t;
			}
			if(!probes.empty()&&t>=probe_start_t) { // FX/setup.cpp:4498-4509
				lbm.gather_u(probe_buf.data());
				size_t k = 0u;
				for(ProbeColumn& pc : probes) { pc.time_si.push_back((double)t*dt_si_d); for(size_t l=0u; l<pc.z.size(); l++, k++) for(int d=0; d<3; d++) pc.uvw_si.push_back(units.si_u(probe_buf[3u*k+(size_t)d])); }
			}
		}
		show_progress(true);
		if(console_row) { std::cout << "\r"; println(ProgressTable::row(N, bytes_per_cell, meter, lbm.get_t(), total_steps)); println(ProgressTable::bottom()); } // the final row also goes into the log
		const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now()-t_start).count();
		print_kv_row("Solver", to_string_u(total_steps)+" steps in "+to_string_fd((float)secs, 3u)+" s = "+to_string_fd((float)((double)N*(double)total_steps/secs*1e-6), 1u)+" MLUPs");
		phase_mark("solver loop");
		{ // write_final_transient, FX/setup.cpp:4762-4776
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
			g_progress.emit("save", "Saving results", saved.size()==1u ? saved.back() : to_string_u(saved.size())+" files saved; last: "+saved.back(), (long long)saved.size(), (long long)saved.size(), false);
		}
		phase_mark("final raw VTKs");
		if(c.research_output_steps>0u) { // maybe_write_transform_info, FX/setup.cpp:4778-4798
			println("| Writing transform.info...                                                  |");
			const string info_path = c.parent+"/proj_temp/transform.info";
			std::ofstream info(info_path);
			if(info.is_open()) { const float dt_si = c.cell_m*(lbm_ref_u/si_ref_u); info << "dt = " << std::fixed << std::setprecision(10) << dt_si << "s\n"; info.close(); println("| Successfully wrote "+info_path+" |"); }
			else println("ERROR: Could not open "+info_path+" for writing.");
		}
		if(avg_window>0ull&&!host_vtk_path()) { // finalize_avg + write_avg_vtk (FX/setup.cpp:4693-4717,2513-2683) with the devices producing every section
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
				if(use_temperature_bc) section("T_avg", LUW_EXPORT_AVG_T, 1u, export_params(units.si_dT(1.0f), units.si_T(0.0f))); // Kelvin: FX/setup.cpp:2526-2528,2580-2582
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
		if(avg_window>0ull&&host_vtk_path()) { // the same file through the host (cross-check path)
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
				std::unique_ptr<float[]> conv(new float[3ull*points]); float* const buf = conv.get(); // one conversion buffer for all fields, fully written before each use
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
					if(c.out_ti) { const ulong i3 = 3ull*n; const float umag = sqrtf(avg_u[i3]*avg_u[i3]+avg_u[i3+1ull]*avg_u[i3+1ull]+avg_u[i3+2ull]*avg_u[i3+2ull]); if(umag>1.0e-9f&&var_sum>0.0f) ti[n] = sqrtf(var_sum*(1.0f/3.0f))/umag; }
					if(!c.out_tls) return;
					const ulong z = n/plane, rem = n-z*plane, y = rem/Nx, x = rem-y*Nx;
					const ulong xm = x>0ull ? x-1ull : x, xp = x+1ull<Nx ? x+1ull : x, ym = y>0ull ? y-1ull : y, yp = y+1ull<Ny ? y+1ull : y, zm = z>0ull ? z-1ull : z, zp = z+1ull<Nz_out ? z+1ull : z;
					const ulong ixm = xm+(y+z*Ny)*Nx, ixp = xp+(y+z*Ny)*Nx, iym = x+(ym+z*Ny)*Nx, iyp = x+(yp+z*Ny)*Nx, izm = x+(y+zm*Ny)*Nx, izp = x+(y+zp*Ny)*Nx;
					const float idx_ = xp>xm ? 1.0f/((float)(xp-xm)*grid_dx) : 0.0f, idy = yp>ym ? 1.0f/((float)(yp-ym)*grid_dx) : 0.0f, idz = zp>zm ? 1.0f/((float)(zp-zm)*grid_dx) : 0.0f;
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
		phase_mark("statistics download + avg VTK");
		if(!probes.empty()) { // FX/setup.cpp:4718-4760
			std::filesystem::create_directories(c.parent+"/RESULTS");
			ulong written = 0ull;
			for(const ProbeColumn& pc : probes) { const string path = c.parent+"/RESULTS/"+pc.stem+".csv"; if(write_probe_csv(path, pc)) written++; else print_kv_row("Probe output", "failed to open "+path); }
			print_kv_row("Probe files", to_string_u(written)+" CSV saved to RESULTS");
			g_progress.emit("save", "Saving results", to_string_u(written)+" probe CSV file(s) saved to RESULTS", (long long)written, (long long)written, false); // FX/setup.cpp:4754-4759
		}
		print_kv_row("Task finished", "["+now_str()+"]");
	}
	println(hr_plain());
	return 0;
}
