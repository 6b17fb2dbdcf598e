// deck_config.hpp -- command line and deck -> Config: the solver-consumed keys of FX/setup.cpp:2911-3308 for the three deck modes, the clamps of
// :3315-3362 and mesh_control (:3364-3390, gpu_memory bisection included).  Prints what the reference prints while it reads.
// Part of the deck driver (luw_driver.cpp); included by it only, after console.hpp, setup_math.hpp and deck.hpp.
#pragma once

// options after the deck path (the reference ignores extra arguments with a warning, FX/setup.cpp:2768-2773); false: usage error
static bool parse_command_line(const int argc, char** argv, Config& c) {
	if(argc<2) {
		std::fprintf(stderr, "usage: %s <deck.luwpf|.luwdg> [--ddf fp32|fp16c] [--arith exact|native] [--device N] [--dry-run] [--dump-setup FILE]\n", argv[0]);
		return false;
	}
	c.deck_path = argv[1];
	for(int i=2; i<argc; i++) {
		const string a = argv[i];
		if(a=="--ddf"&&i+1<argc) { const string v = argv[++i]; c.fp16c = v!="fp32"; }
		// FP16C collision arithmetic: native (default: the hardware's own division / square root and fused multiply-adds, LUW_OPT_NATIVE_ARITH -- what the
		// reference's OpenCL build does, and as close to its fields as the bit-exact kernels are: DESIGN.md section 3) or exact (bit-equal to the CPU oracle)
		else if(a=="--arith"&&i+1<argc) { const string v = argv[++i]; c.native_arith = v!="exact"; }
		else if(a=="--device"&&i+1<argc) c.device = std::atoi(argv[++i]);
		else if(a=="--dry-run") c.dry_run = true;
		else if(a=="--sizing-only") { c.dry_run = true; c.sizing_only = true; } // stop after the derived numbers (no lattice-sized host arrays)
		else if(a=="--dump-setup"&&i+1<argc) c.dump_setup = argv[++i];
		else if(a=="--dump-vk"&&i+1<argc) c.dump_vk = argv[++i];
		// HIP device per domain (several domains may share one)
		else if(a=="--devices"&&i+1<argc) {
			std::stringstream ss(argv[++i]);
			string tok;
			while(std::getline(ss, tok, ',')) if(!tok.empty()) c.devices.push_back(std::atoi(tok.c_str()));
		}
		else if(a=="--kernel"&&i+1<argc) {
			const string v = argv[++i];
			c.kernel = v=="scalar" ? LUW_KERNEL_SCALAR : v=="pair" ? LUW_KERNEL_PAIR : LUW_KERNEL_AUTO;
		}
		else println("| WARNING: extra CLI arg ignored: "+a);
	}
	return true;
}

// mode from the suffix of the deck, every key the solver consumes, the clamps, the cell size of the grid; project directory = the parent of the deck
// (FX/setup.cpp:3391)
static void read_deck(Config& c) {
	{ // mode from the suffix, FX/setup.cpp:2792-2810
		string ext = std::filesystem::path(c.deck_path).extension().string();
		std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
		if(ext==".luwdg") { c.dataset_mode = true; println("| Dataset generation mode enabled (*.luwdg).                                  |"); }
		else if(ext==".luwpf") { c.profile_mode = true; println("| Profile forcing mode enabled (*.luwpf).                                     |"); }
		else c.nwp_mode = true; // *.luw: boundaries from proj_temp/SurfData_<datetime>.csv
	}
	std::ifstream fin(c.deck_path);
	if(!fin.is_open()) fatal("ERROR: config not found. Please provide a valid *.luw, *.luwdg, or *.luwpf and rerun.");
	Deck deck_file; deck_file.load(fin);
	const auto& deck = deck_file.entries();
	string mesh_control, gpu_memory_val, cell_size_val;
	auto second_val = [](const string& r) {
		const size_t cpos = r.find(','), rpos = r.find(']', cpos);
		return (float)atof(r.substr(cpos+1u, rpos-cpos-1u).c_str());
	};
	auto parse_float_list = [](const string& r, std::vector<float>& out) {
		out.clear(); string s = Deck::strip(r); const size_t lb = s.find('['), rb = s.find(']', lb);
		const string inside = (lb!=string::npos&&rb!=string::npos&&rb>lb) ? s.substr(lb+1u, rb-lb-1u) : s;
		std::stringstream ss(inside); string tok;
		while(std::getline(ss, tok, ',')) { const string t = Deck::strip(tok); if(!t.empty()) out.push_back((float)atof(t.c_str())); }
	};
	auto parse_pair = [](const string& r, float& a, float& b) {
		const size_t lb = r.find('['), rb = r.find(']', lb);
		if(lb==string::npos||rb==string::npos) return;
		std::stringstream ss(r.substr(lb+1u, rb-lb-1u));
		string tok;
		int i = 0;
		while(std::getline(ss, tok, ',')) { const float v = (float)atof(Deck::strip(tok).c_str()); if(i==0) a = v; else if(i==1) b = v; i++; }
	};
	for(const auto& e : deck) { // FX/setup.cpp:2911-3308 (solver-consumed keys of the supported modes)
		const string& key = e.first; const string& val = e.second; const string uq = Deck::text(val); bool pb = false;
		if(key=="casename") c.caseName = uq;
		else if(key=="datetime") c.datetime = uq;
		else if(key=="buoyancy") {
			string v = uq;
			if(!v.empty()) {
				std::transform(v.begin(), v.end(), v.begin(), ::tolower);
				c.buoyancy_explicit = true;
				bool parsed = true;
				c.buoyancy = Deck::flag(v, parsed) ? parsed : true;
			}
		}
		else if(key=="downstream_bc") c.downstream_bc = uq;
		else if(key=="downstream_bc_yaw") c.downstream_bc_yaw = uq;
		else if(key=="high_order") { if(!uq.empty()&&Deck::flag(uq, pb)) c.use_high_order = pb; }
		else if(key=="flux_correction") { if(!uq.empty()&&Deck::flag(uq, pb)) c.flux_correction = pb; }
		else if(key=="validation") c.validation = uq;
		else if(key=="downstream_open_face") { if(!uq.empty()&&Deck::flag(uq, pb)) c.downstream_open_face = pb; }
		else if(key=="base_height") { if(!uq.empty()) c.z_si_offset = (float)atof(val.c_str()); }
		else if(key=="memory_lbm") { if(!uq.empty()) c.memory = (uint)atoi(val.c_str()); }
		else if(key=="si_x_cfd") { if(!uq.empty()) c.si_x = second_val(val); }
		else if(key=="si_y_cfd") { if(!uq.empty()) c.si_y = second_val(val); }
		else if(key=="si_z_cfd") { if(!uq.empty()) c.si_z = second_val(val); }
		else if(key=="enable_buffer_nudging") { if(!uq.empty()&&Deck::flag(uq, pb)) c.enable_buffer_nudging = pb; }
		else if(key=="buffer_thickness_m") { if(!uq.empty()) c.buffer_thickness_m = (float)atof(uq.c_str()); }
		else if(key=="buffer_tau_s") { if(!uq.empty()) c.buffer_tau_s = (float)atof(uq.c_str()); }
		else if(key=="buffer_nudge_vertical") { if(!uq.empty()&&Deck::flag(uq, pb)) c.buffer_nudge_vertical = pb ? 1 : 0; }
		else if(key=="enable_top_sponge") { if(!uq.empty()&&Deck::flag(uq, pb)) c.enable_top_sponge = pb; }
		else if(key=="sponge_thickness_m") { if(!uq.empty()) c.sponge_thickness_m = (float)atof(uq.c_str()); }
		else if(key=="sponge_tau_s") { if(!uq.empty()) c.sponge_tau_s = (float)atof(uq.c_str()); }
		else if(key=="sponge_ref_mode") {
			string v = uq;
			std::transform(v.begin(), v.end(), v.begin(), ::tolower);
			c.sponge_ref_mode = (v=="0"||v=="mode0"||v=="mode_0") ? 0 : (v=="1"||v=="mode1"||v=="mode_1"||v=="geostrophic") ? 1 : atoi(v.c_str());
		}
		else if(key=="mesh_control") mesh_control = uq;
		else if(key=="gpu_memory") gpu_memory_val = uq;
		else if(key=="cell_size") cell_size_val = uq;
		else if(key=="n_gpu") {
			if(!uq.empty()) {
				const size_t lb = val.find('['), rb = val.find(']', lb);
				if(lb!=string::npos&&rb!=string::npos) {
					std::stringstream ss(val.substr(lb+1u, rb-lb-1u));
					string tok;
					uint v[3] = {c.Dx, c.Dy, c.Dz};
					int i = 0;
					while(std::getline(ss, tok, ',')&&i<3) v[i++] = (uint)atoi(Deck::strip(tok).c_str());
					if(i==3) { c.Dx = v[0]; c.Dy = v[1]; c.Dz = v[2]; }
				}
			}
		}
		else if(key=="research_output") { if(!uq.empty()) c.research_output_steps = (uint)atoi(val.c_str()); }
		else if(key=="unsteady_output") { if(!uq.empty()) { const int v = atoi(uq.c_str()); c.unsteady_output_interval = v>0 ? (uint)v : 0u; } }
		else if(key=="run_nstep") { if(!uq.empty()) { const long long v = atoll(uq.c_str()); c.run_nstep_override = v>0ll ? (ulong)v : 0ull; } }
		else if(key=="purge_avg") { if(!uq.empty()) { const int v = atoi(val.c_str()); c.purge_avg_steps = v>0 ? (uint)v : 0u; } }
		else if(key=="purge_avg_stride") { if(!uq.empty()) { const int v = atoi(uq.c_str()); c.purge_avg_stride = v>0 ? (uint)v : 1u; } }
		else if(key=="output_tke_ti_tls") {
			const string lt = Deck::strip(uq);
			const size_t lb = lt.find('['), rb = lt.find(']', lb);
			if(!lt.empty()&&lb!=string::npos&&rb!=string::npos&&rb>lb) {
				c.out_tke = c.out_ti = c.out_tls = false;
				std::stringstream ss(lt.substr(lb+1u, rb-lb-1u));
				string tok;
				while(std::getline(ss, tok, ',')) {
					string it = Deck::strip(tok);
					std::transform(it.begin(), it.end(), it.begin(), ::tolower);
					if(it=="tke") c.out_tke = true;
					else if(it=="ti") c.out_ti = true;
					else if(it=="tls") c.out_tls = true;
				}
			}
		}
		else if(key=="coriolis_term") {
			string v = uq;
			std::transform(v.begin(), v.end(), v.begin(), ::tolower);
			if(!v.empty()&&Deck::flag(v, pb)) c.enable_coriolis = pb;
		}
		else if(key=="turb_inflow_enable") { if(!uq.empty()&&Deck::flag(uq, pb)) c.vk_enable = pb; }
		else if(key=="vk_inlet_nmodes") { if(!uq.empty()) c.vk_nmodes = atoi(uq.c_str()); }
		else if(key=="vk_inlet_ti") { if(!uq.empty()) c.vk_ti = (float)atof(uq.c_str()); }
		else if(key=="vk_inlet_sigma") { if(!uq.empty()) c.vk_sigma_si = (float)atof(uq.c_str()); }
		else if(key=="vk_inlet_l") { if(!uq.empty()) c.vk_L_si = (float)atof(uq.c_str()); }
		else if(key=="vk_inlet_seed") {
			if(!uq.empty()) {
				char* end = nullptr;
				const unsigned long long v = std::strtoull(uq.c_str(), &end, 10);
				if(end!=uq.c_str()) c.vk_seed = (uint64_t)v;
			}
		}
		else if(key=="vk_inlet_update_stride") { if(!uq.empty()) c.vk_stride = atoi(uq.c_str()); }
		else if(key=="vk_inlet_uc_mode") {
			string v = uq;
			std::transform(v.begin(), v.end(), v.begin(), ::toupper);
			if(v=="NORM_MEAN") c.vk_uc = VkUcMode::NORM_MEAN;
			else if(v=="NORMAL_COMPONENT") c.vk_uc = VkUcMode::NORMAL_COMPONENT;
		}
		else if(key=="vk_inlet_same_realization_all_faces") { if(!uq.empty()&&Deck::flag(uq, pb)) c.vk_same = pb; }
		else if(key=="vk_inlet_stride_interpolation") { if(!uq.empty()&&Deck::flag(uq, pb)) c.vk_interp = pb; }
		else if(key=="vk_inlet_inflow_only") { if(!uq.empty()&&Deck::flag(uq, pb)) c.vk_inflow_only = pb; }
		else if(key=="vk_inlet_face_mode")
			{ string v = uq; std::transform(v.begin(), v.end(), v.begin(), [](unsigned char ch) { return ch=='-' ? '_' : (char)std::toupper(ch); });
			if(v=="AUTO"||v=="AUTO_SIDES"||v=="BY_INFLOW_ONLY"||v=="BUSINESS_DEFAULT"||v=="DEFAULT") c.vk_face_mode = VkFaceMode::AUTO_SIDES;
			else if(v=="TARGET_INFLOW"||v=="INFLOW"||v=="TARGET"||v=="UPSTREAM_ONLY") c.vk_face_mode = VkFaceMode::TARGET_INFLOW;
			else if(v=="EXCLUDE_DOWNSTREAM"||v=="EXCEPT_DOWNSTREAM"||v=="ALL_EXCEPT_DOWNSTREAM"||v=="NON_DOWNSTREAM")
				c.vk_face_mode = VkFaceMode::EXCLUDE_DOWNSTREAM;
			else if(v=="EXCLUDE_DOWNSTREAM_SIDES"||v=="EXCEPT_DOWNSTREAM_SIDES"||v=="SIDE_EXCEPT_DOWNSTREAM"||v=="SIDES_EXCEPT_DOWNSTREAM"
				||v=="NON_DOWNSTREAM_SIDES"||v=="SIDE_FACES_EXCEPT_DOWNSTREAM") c.vk_face_mode = VkFaceMode::EXCLUDE_DOWNSTREAM_SIDES;
			else if(v=="ALL_SIDES"||v=="SIDE_FACES"||v=="ALL_SIDE_FACES"||v=="SIDES_ONLY"||v=="ALL_SIDES_NO_TOP") c.vk_face_mode = VkFaceMode::ALL_SIDES;
			else if(v=="ALL"||v=="ALL_SELECTED"||v=="ALL_FACES") c.vk_face_mode = VkFaceMode::ALL_SELECTED; }
		else if(key=="vk_inlet_anisotropy") {
			if(!uq.empty()) {
				const size_t lb = uq.find('['), rb = uq.find(']', lb);
				const string in = (lb!=string::npos&&rb!=string::npos&&rb>lb) ? uq.substr(lb+1u, rb-lb-1u) : uq;
				std::stringstream ss(in);
				string tok;
				float v[3];
				int i = 0;
				bool ok = true;
				while(std::getline(ss, tok, ',')&&i<3) {
					const string t = Deck::strip(tok);
					char* end = nullptr;
					const float f = std::strtof(t.c_str(), &end);
					if(t.empty()||end==t.c_str()) { ok = false; break; }
					v[i++] = f;
				}
				if(ok&&i==3) for(int k=0; k<3; k++) c.vk_aniso[k] = (std::isfinite(v[k])&&v[k]>=0.0f) ? v[k] : 1.0f;
			}
		}
		else if(key=="cut_lon_manual") { if(!uq.empty()) { parse_pair(val, c.cut_lon[0], c.cut_lon[1]); c.has_cut_lon = true; } }
		else if(key=="cut_lat_manual") { if(!uq.empty()) { parse_pair(val, c.cut_lat[0], c.cut_lat[1]); c.has_cut_lat = true; } }
		else if(key=="probes") c.probes_raw = Deck::strip(val);
		else if(key=="probes_output") {
			if(!uq.empty()) {
				const int v = atoi(uq.c_str());
				c.probes_output_defined = true;
				if(v>0) c.probes_output_steps = (uint)v;
				else { c.probes_output_steps = 0u; println("| WARNING: probes_output must be > 0 to take effect. Fallback to legacy window. |"); }
			}
		}
		else if(key=="utm_crs") { if(!uq.empty()) c.utm_crs = uq; }
		else if(key=="rotate_deg") {
			if(!uq.empty()) {
				char* end = nullptr;
				const double v = std::strtod(uq.c_str(), &end);
				if(end!=uq.c_str()&&std::isfinite(v)) { c.rotate_deg = v; c.has_rotate_deg = true; }
			}
		}
		else if(key=="inflow") { if(!uq.empty()) parse_float_list(val, c.inflow_list); }
		else if(key=="angle") { if(!uq.empty()) parse_float_list(val, c.angle_list); }
	}
	if(!c.memory) c.memory = 6000u;
	if(c.Dx==0u) c.Dx = 1u; if(c.Dy==0u) c.Dy = 1u; if(c.Dz==0u) c.Dz = 1u;
	if(c.vk_ti<0.0f) c.vk_ti = 0.0f; if(c.vk_ti>1.0f&&c.vk_ti<=100.0f) c.vk_ti *= 0.01f; // FX/setup.cpp:3315-3362
	if(c.vk_sigma_si<0.0f) c.vk_sigma_si = 0.0f; if(c.vk_L_si<0.0f) c.vk_L_si = 0.0f;
	if(c.vk_nmodes<=0) c.vk_nmodes = 256; if(c.vk_nmodes>512) c.vk_nmodes = 512;
	if(c.vk_stride<=0) c.vk_stride = 1;
	if(c.vk_enable&&!(c.vk_L_si>0.0f)) { println("| WARNING: turb_inflow_enable=true but L is invalid. VK inlet disabled.         |"); c.vk_enable = false; }
	if(c.vk_enable&&!(c.vk_ti>0.0f||c.vk_sigma_si>0.0f)) {
		println("| WARNING: turb_inflow_enable=true but TI/sigma is invalid. VK inlet disabled.  |");
		c.vk_enable = false;
	}
	{ // mesh_control, FX/setup.cpp:3364-3390
		bool applied = false;
		if(mesh_control=="gpu_memory") {
			if(!Deck::strip(gpu_memory_val).empty()) {
				const uint mm = (uint)atoi(Deck::strip(gpu_memory_val).c_str());
				if(mm>0u) { c.memory = mm; c.cell_m = fit_cell_size_to_gpu_memory_request(c, c.memory); applied = true; }
			}
		}
		else if(mesh_control=="cell_size") {
			if(!Deck::strip(cell_size_val).empty()) {
				const float cs = (float)atof(Deck::strip(cell_size_val).c_str());
				if(cs>0.0f&&std::isfinite(cs)) { c.cell_m = cs; applied = true; }
			}
		}
		if(!applied) c.cell_m = 20.0f;
	}
	c.parent = std::filesystem::path(c.deck_path).parent_path().string();
	if(c.parent.empty()) c.parent = ".";
	if(c.profile_mode&&c.enable_coriolis&&!(c.has_cut_lon&&c.has_cut_lat)) {
		println("| WARNING: coriolis_term=true but cut_lon_manual/cut_lat_manual is missing in *.luwpf. |");
		println("| WARNING: Coriolis is auto-disabled for Profile mode.                         |");
		c.enable_coriolis = false;
	}
	{ // console log tee, FX/setup.cpp:2502-2511
		std::error_code ec; std::filesystem::create_directories(std::filesystem::path(c.parent)/"proj_temp", ec);
		const string lp = (std::filesystem::path(c.parent)/"proj_temp"/(now_str("%Y%m%d%H%M%S")+"_lbm.log")).string();
		if(!ec&&!c.dry_run) { g_log.open(lp); if(g_log.is_open()) println("| Console log     | "+lp+" |"); }
	}
}
