// driver_run.hpp -- run_lbm of main_setup (FX/setup.cpp:4117-4911): upload + initialise, then the time loop in batches between the steps at which
// the host must look (unsteady output, probe samples, the first sample of the averaging window).  Part of the deck driver (luw_driver.cpp);
// included by it only, after driver_state.hpp.
#pragma once

inline void Driver::run_solver() {
	LBM& lbm = *lbm_p;
	lbm.set_coriolis(omega[0], omega[1], omega[2]);
	if(vk_on) lbm.vk_inlet_attach(vk.point_count, vk.mode_count, vk.point_cell.data(), vk.point_face.data(), vk.point_data.data(), vk.mode_data.data(),
		c.vk_stride, c.vk_interp ? 1 : 0);
	print_section_title("LBM SOLVER INFORMATION");
	if(use_temperature_bc) print_kv_row("Export mode", "include temperature T field in Kelvin");
	if(Nz_out<Nz) print_kv_row("VTK z output", "core Nz="+to_string_u(Nz_out)+" of solver Nz="+to_string_u(Nz)+" (top sponge omitted)");
	print_kv_row("Run steps", to_string_u(total_steps)+(c.run_nstep_override>0ull ? " (run_nstep override)" : " (default)"));
	if(avg_window>0ull) {
		print_kv_row("Avg stride", "sample every "+to_string_u(avg_stride)+" step(s) in purge_avg window (on-device accumulation)");
		lbm.stats_reset();
	}
	if(!probe_cells.empty()) lbm.gather_attach((uint32_t)probe_cells.size(), probe_cells.data());
	std::vector<float> probe_buf(3u*probe_cells.size());
	lbm.run(0u, total_steps);
	phase_mark("upload + initialise");
	print_section_title("SOLVER START");
	// ---- the time loop.  The device runs batches of steps without any host round trip inside; between batches the host looks at the
	// clock, refreshes the running row / the GUI's progress line and handles whatever must be observed at that step (unsteady output,
	// probe samples).  A batch ends at the next such step, and otherwise after about a quarter of a second of work.
	// DDFs + flags (+ the 7 planes of the thermal lattice; rho, u and T are stored by the last step of a batch only), DESIGN.md section 5
	const double bytes_per_cell = (c.fp16c ? 77.0 : 153.0)+(use_temperature_bc ? (c.fp16c ? 28.0 : 56.0) : 0.0);
	StepRateMeter meter; meter.configure(total_steps, avg_window>0ull ? avg_start_t : ~0ull);
	const bool console_row = !g_progress.gui(); // FX/info.cpp:225: the GUI gets protocol lines instead of the table
	if(console_row) { println(ProgressTable::top()); println(ProgressTable::header()); }
	auto last_gui = std::chrono::steady_clock::time_point{};
	auto show_progress = [&](const bool force) {
		const ulong t = lbm.get_t();
		if(console_row) reprint_row(ProgressTable::row(N, bytes_per_cell, meter, t, total_steps));
		const auto now = std::chrono::steady_clock::now();
		// FX/setup.cpp:4144-4164
		if(!g_progress.gui()||(!force&&t<total_steps&&last_gui.time_since_epoch().count()!=0&&now-last_gui<std::chrono::milliseconds(120))) return;
		last_gui = now;
		g_progress.emit("solve", "Solving CFD", to_string_u(t)+"/"+to_string_u(total_steps)+" steps | "+to_string_fd((float)meter.steps_per_second(t), 3u)
			+" Steps/s | ETA "+clock_text(meter.remaining_seconds(t)), (long long)t, (long long)total_steps, false);
	};
	auto note_saved = [&](const std::vector<string>& files) { // flush_vtk_saved_files, FX/setup.cpp:4192-4218
		if(files.empty()) return;
		if(console_row) std::cout << "\r" << string(CONSOLE_WIDTH, ' ') << "\r";
		bool first = true; for(const string& f : files) { print_kv_row(first ? "VTK file" : "", f+" saved"); first = false; }
		g_progress.emit("save", "Saving results", files.size()==1u ? files.back() : to_string_u(files.size())+" files saved; last: "+files.back(),
			(long long)files.size(), (long long)files.size(), false);
	};
	const auto t_start = std::chrono::steady_clock::now();
	last_u_vtk_t = ~0ull;
	ulong batch_cap = 16ull; // first batch: the reference's 16-step "Normal benchmark" (FX/setup.cpp:4799-4841) doubles as the speed sample
	g_progress.emit("speed_estimate", "Estimating solve speed", "Benchmarking normal LBM solver", 0ll, (long long)std::min<ulong>(batch_cap, total_steps),
		false);
	bool speed_reported = false;
	while(lbm.get_t()<total_steps) {
		// the next step at which something must be observed (unsteady output / probe sample / end); fields are written by the last step of each batch
		ulong next = std::min(total_steps, lbm.get_t()+batch_cap);
		if(unsteady>0ull) next = std::min(next, (ulong)((lbm.get_t()/unsteady+1ull)*unsteady));
		if(!probes.empty()) next = std::min(next, std::max((ulong)(lbm.get_t()+1ull), probe_start_t)); // every step of the probe window is observed
		// a batch belongs to ONE stage of the time estimate
		if(avg_window>0ull&&lbm.get_t()+1ull<avg_start_t) next = std::min<ulong>(next, avg_start_t-(ulong)1u);
		// statistics samples that fall into (t, next] ride along (run_sampled): first sample s, then every avg_stride-th step
		ulong first_sample = 0ull;
		if(avg_window>0ull) {
			const ulong t1 = lbm.get_t()+1ull;
			ulong sm = std::max(t1, avg_start_t);
			const ulong off = (sm-avg_start_t)%avg_stride;
			if(off!=0ull) sm += avg_stride-off;
			if(sm<=next) first_sample = sm;
		}
		const ulong nsteps = next-lbm.get_t();
		const auto b0 = std::chrono::steady_clock::now();
		if(first_sample>0ull) lbm.run_sampled(nsteps, first_sample-lbm.get_t(), avg_stride);
		else lbm.run(nsteps, total_steps);
		const double bsec = std::chrono::duration<double>(std::chrono::steady_clock::now()-b0).count();
		const ulong t = lbm.get_t();
		meter.add_batch(t, nsteps, bsec);
		if(!speed_reported) {
			speed_reported = true;
			g_progress.emit("speed_estimate", "Estimating solve speed", "Benchmarking normal LBM solver step "+to_string_u(nsteps)+"/"+to_string_u(nsteps),
				(long long)nsteps, (long long)nsteps, false);
		}
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
			for(ProbeColumn& pc : probes) {
				pc.time_si.push_back((double)t*dt_si_d);
				for(size_t l=0u; l<pc.z.size(); l++, k++) for(int d=0; d<3; d++) pc.uvw_si.push_back(units.si_u(probe_buf[3u*k+(size_t)d]));
			}
		}
	}
	show_progress(true);
	// the final row also goes into the log
	if(console_row) { std::cout << "\r"; println(ProgressTable::row(N, bytes_per_cell, meter, lbm.get_t(), total_steps)); println(ProgressTable::bottom()); }
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now()-t_start).count();
	print_kv_row("Solver", to_string_u(total_steps)+" steps in "+to_string_fd((float)secs, 3u)+" s = "
		+to_string_fd((float)((double)N*(double)total_steps/secs*1e-6), 1u)+" MLUPs");
}
