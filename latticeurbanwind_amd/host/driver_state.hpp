// driver_state.hpp -- what the reference's main_setup (FX/setup.cpp:2726-6154) keeps in the locals of one 3400-line function, as ONE object whose
// member functions are the sections of that function in their order: deck -> sizes -> inflow inputs -> units and forcing -> geometry -> profile
// table, then per (inflow, angle) case: solver + voxeliser -> boundary fill -> inlet tables -> output plan -> time loop -> writers.
// Part of the deck driver (luw_driver.cpp); included by it only.  Definitions: driver_setup.hpp (before the cases), driver_case.hpp (set-up of a
// case), driver_boundaries.hpp (boundary fill), driver_run.hpp (time loop), driver_output.hpp (files).
#pragma once

struct Driver {
	// ---- the deck and what follows from it before any case (driver_setup.hpp)
	Config c;
	std::vector<ProbeRequest> probe_requests; GeoFrame probe_geo; // FX/setup.cpp:3396-3426
	const float lbm_ref_u = 0.10f; float si_ref_u = 10.0f; const float si_nu = 1.48E-5f, si_rho = 1.225f;
	uint lbmN[3] = {1u, 1u, 1u}; // lbm_N: the resolution before LBM::LBM makes it divisible by the domains
	int sponge_cells_cfg = 1; uint Nz_core = 1u; bool top_sponge_grid_extend = false; int side_ref_z_cap = -1;
	std::vector<float> prof_z, prof_u; // profile.dat samples (SI)
	float temperature_ref_kelvin = 293.15f, temperature_scale_kelvin = 293.15f; bool temperature_ref_adaptive = false, temperature_scale_adaptive = false;
	SurfData surf;
	Units units; float lbm_alpha = 0.0f; bool use_temperature_bc = false; float u_scale = 1.0f, lbm_nu = 0.0f; float omega[3] = {0.0f, 0.0f, 0.0f};
	float T_bc_min = 1.0f, T_bc_max = 1.0f;
	SolverGlobals& G = solver_globals();
	Mesh mesh; float stl_min[3] = {0, 0, 0}, stl_max[3] = {0, 0, 0}, vtk_origin_shift[3] = {0, 0, 0}, scale_geom = 1.0f; DemPoints dem;
	float origin_z = 0.0f, flat_ground = 0.0f; std::vector<float> prof_lbmu; const float profile_dz = 0.1f;
	GroundPlane2D ground_plane; bool use_dem_ground = false; float ground_z_min = 0.0f, ground_z_max = 0.0f;
	struct Case { float inflow_si, angle_deg; };
	std::vector<Case> cases;

	void read_probe_requests();
	void print_parameters() const;
	void size_lattice();
	void read_inflow_inputs();
	void set_units_and_forcing();
	void load_geometry();
	void build_profile_table();
	void list_cases();
	void update_coriolis();
	void update_buffer_nudging(const string& bc);
	void update_top_sponge();
	float profile_speed(const float pos_z, const float ground_z) const;
	static int buffer_face_id_from_bc(const string& bc) { return bc=="-x" ? 1 : bc=="+x" ? 2 : bc=="-y" ? 3 : bc=="+y" ? 4 : 0; }
	static string bc_from_dir(const float dx, const float dy) { if(fabsf(dx)>=fabsf(dy)) return dx>=0.0f ? "+x" : "-x"; return dy>=0.0f ? "+y" : "-y"; }

	// ---- one case: everything inside uses the lattice of the LBM object (divisible by the domains), FX/lbm.cpp:1058-1060
	Case cs{0.0f, 0.0f}; uint case_index = 0u;
	uint Nx = 1u, Ny = 1u, Nz = 1u; ulong N = 1ull;
	float dir_x = 0.0f, dir_y = 0.0f, uin[3] = {0.0f, 0.0f, 0.0f};
	string vtk_prefix, case_bc;
	// host state: the LBM object's global host arrays (lbm.flags[n], lbm.u.x[n], lbm.T[n]) or, without a GPU, plain vectors
	std::vector<uchar> flags_store; std::vector<float> u_store, T_store;
	std::unique_ptr<LBM> lbm_p;
	ulong nvox = 0ull;
	uchar* flags = nullptr; float* u = nullptr; float* Tcell = nullptr;
	std::atomic<ulong> mapped{0ull}, terrain_solid{0ull}, outlet{0ull};
	std::vector<float> ground_xy; // terrain height per column (profile mode with a DEM), else flat
	HostLattice lattice;
	VkTables vk; bool vk_on = false;
	// output plan
	ulong total_steps = 0ull, unsteady = 0ull; string results_vtk_dir, vtk_dir; uint Nz_out = 1u; VtkGeom geom{};
	ulong avg_window = 0ull, avg_stride = 1ull, avg_start_t = ~0ull;
	double dt_si_d = 0.0; ulong probe_window = 0ull, probe_start_t = ~0ull;
	std::vector<ProbeColumn> probes; std::vector<uint64_t> probe_cells;
	ulong last_u_vtk_t = ~0ull;

	float pos_z_of(const uint z) const { return (float)z-0.5f*(float)Nz+0.5f; } // lbm.position(x, y, z).z
	bool is_downstream(const uint x, const uint y) const {
		return case_bc=="+y" ? y==Ny-1u : case_bc=="-y" ? y==0u : case_bc=="+x" ? x==Nx-1u : case_bc=="-x" ? x==0u : false;
	}
	float ground_at(const ulong id) const { return ground_xy.empty() ? flat_ground : ground_xy[id]; }

	void run_case(const Case& which); // the whole case, in the order of the functions below
	void begin_case(const Case& which);
	void create_solver_and_voxelize();
	void begin_boundaries();
	void fill_nwp_boundaries();
	void fill_profile_boundaries();
	void fill_dataset_boundaries();
	void profile_flux_correction();
	void report_flux(const FluxReport& fr) const;
	void build_vk_inlet();
	void dump_setup_file() const;
	void plan_outputs();
	void resolve_probes();
	void run_solver();
	void write_final_fields();
	void write_transform_info() const;
	void write_avg_vtk_from_devices();
	void write_avg_vtk_through_host();
	void write_probe_files() const;
};
