// progress.hpp -- what the solver process tells its parent while it works.
//  (1) the GUI protocol: one line `[[LUW_PROGRESS]]{json}` on stdout per event when LUW_PROGRESS_MODE is gui / 1 / true
//      (emitter FX/utilities.hpp:3126-3178, consumer core/luw_progress.py:79-92 and gui/src/CommandRunner.cpp:261); keys
//      stage, label, detail, current, total, indeterminate; stages gpu_memory, load_stl, voxelization,
//      interface_interpolation, flux_correction, speed_estimate, solve, save;
//  (2) the console's running row `MLUPs | Bandwidth | Steps/s | Current Step | Time Remaining` (FX/info.cpp:44-69): MLUPs =
//      N 1e-6 / dt_smooth, GB/s = N * bytes-per-cell / dt_smooth, with the smoothed step time of FX/info.cpp:15-24 and the
//      two-stage time estimate (solver stage + mean-field stage) of FX/info.cpp:107-126,150-176.
// Host code of the deck driver; no device dependencies.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>

namespace luw_host {

class ProgressChannel { // (1)
	bool on = false;
	std::function<void(const std::string&)> sink; // where a protocol line goes (stdout, unbuffered)
	static std::string json_text(const std::string& s) {
		std::string o; o.reserve(s.size()+8u);
		for(const char ch : s) {
			if(ch=='\\') o += "\\\\";
			else if(ch=='"') o += "\\\"";
			else if(ch=='\n') o += "\\n";
			else if(ch=='\r') o += "\\r";
			else if(ch=='\t') o += "\\t";
			else o.push_back(ch);
		}
		return o;
	}
public:
	explicit ProgressChannel(std::function<void(const std::string&)> out) : sink(std::move(out)) {
		const char* e = std::getenv("LUW_PROGRESS_MODE");
		const std::string m = e ? e : "";
		on = m=="gui"||m=="1"||m=="true";
	}
	bool gui() const { return on; }
	void emit(const std::string& stage, const std::string& label, const std::string& detail = "", const long long current = -1ll, const long long total = -1ll,
		const bool indeterminate = true) const {
		if(!on) return;
		sink("[[LUW_PROGRESS]]{\"stage\":\""+json_text(stage)+"\",\"label\":\""+json_text(label)+"\",\"detail\":\""+json_text(detail)+"\",\"current\":"
			+std::to_string(current)+",\"total\":"+std::to_string(total)+",\"indeterminate\":"+(indeterminate ? "true" : "false")+"}");
	}
};

inline std::string clock_text(const double seconds) { // "1d 02h 03m 04s"-style durations like print_time (FX/utilities.hpp)
	const uint64_t s = (uint64_t)(seconds<0.0 ? 0.0 : seconds+0.5);
	const uint64_t d = s/86400ull, h = (s%86400ull)/3600ull, m = (s%3600ull)/60ull, r = s%60ull;
	char b[64];
	if(d) std::snprintf(b, sizeof(b), "%llud %02lluh %02llum %02llus", (unsigned long long)d, (unsigned long long)h, (unsigned long long)m,
		(unsigned long long)r);
	else if(h) std::snprintf(b, sizeof(b), "%lluh %02llum %02llus", (unsigned long long)h, (unsigned long long)m, (unsigned long long)r);
	else if(m) std::snprintf(b, sizeof(b), "%llum %02llus", (unsigned long long)m, (unsigned long long)r);
	else std::snprintf(b, sizeof(b), "%llus", (unsigned long long)r);
	return b;
}

// Step-time bookkeeping of a run with an optional statistics window at its end.  A sample is the wall time of a batch of steps
// divided by their number; the estimate is a running mean over the first eight samples and an exponential average (weight 1/5)
// afterwards, kept separately for the solver stage and for the mean-field (statistics) stage.
class StepRateMeter {
	struct Estimate {
		double seconds = 0.0; uint64_t samples = 0ull;
		void add(const double v) {
			if(!(v>0.0)) return;
			seconds = samples==0ull ? v : seconds+(samples<8ull ? 1.0/(double)(samples+1ull) : 0.2)*(v-seconds);
			samples++;
		}
	};
	Estimate normal, window;
	uint64_t total = 0ull, window_start = ~0ull; // window_start: first step (counted from 1) inside the statistics window
public:
	void configure(const uint64_t total_steps, const uint64_t window_first_step) {
		total = total_steps;
		window_start = window_first_step;
		normal = Estimate{};
		window = Estimate{};
	}
	void add_batch(const uint64_t t_after, const uint64_t steps, const double seconds) {
		if(steps==0ull) return;
		const bool in_window = window_start!=~0ull&&t_after>=window_start;
		(in_window ? window : normal).add(seconds/(double)steps);
	}
	double step_seconds(const uint64_t t) const { // of the stage step t+1 belongs to
		const bool in_window = window_start!=~0ull&&t+1ull>=window_start;
		const double v = in_window&&window.samples ? window.seconds : normal.samples ? normal.seconds : window.seconds;
		return v>1.0e-9 ? v : 1.0e-9;
	}
	double steps_per_second(const uint64_t t) const { return 1.0/step_seconds(t); }
	double remaining_seconds(const uint64_t t) const {
		if(t>=total) return 0.0;
		const double ns = normal.samples ? normal.seconds : window.seconds, ws = window.samples ? window.seconds : ns;
		if(window_start==~0ull||window_start>total) return (double)(total-t)*ns;
		const uint64_t first_w = window_start>0ull ? window_start-1ull : 0ull; // steps done before the window
		const uint64_t rem_normal = t<first_w ? first_w-t : 0ull, rem_window = total-(t<first_w ? first_w : t);
		return (double)rem_normal*ns+(double)rem_window*ws;
	}
};

// (2) the table under "LBM SOLVER INFORMATION": column widths of FX/info.cpp:5-9
class ProgressTable {
	static std::string centre(const unsigned n, const std::string& x) {
		if(x.size()>=n) return x.substr(0u, n);
		const unsigned l = (n-(unsigned)x.size())/2u;
		return std::string(l, ' ')+x+std::string(n-(unsigned)x.size()-l, ' ');
	}
	static constexpr unsigned W[5] = { 9u, 13u, 11u, 19u, 36u };
public:
	static std::string top() { return "|---------.-------'-----.-----------.-------------------.------------------------------------|"; }
	static std::string bottom() { return "|---------'-------------'-----------'-------------------'------------------------------------|"; }
	static std::string header() {
		return "|"+centre(W[0], "MLUPs")+"|"+centre(W[1], "Bandwidth")+"|"+centre(W[2], "Steps/s")+"|"+centre(W[3], "Current Step")+"|"
			+centre(W[4], "Time Remaining")+"|";
	}
	static std::string row(const uint64_t cells, const double bytes_per_cell, const StepRateMeter& m, const uint64_t t, const uint64_t total) {
		const double dt = m.step_seconds(t>0ull ? t-1ull : 0ull);
		const unsigned pct = total ? (unsigned)((double)(t>total ? total : t)*100.0/(double)total) : 100u;
		char cur[48]; std::snprintf(cur, sizeof(cur), "%llu %3u%%", (unsigned long long)t, pct);
		return "|"+centre(W[0], std::to_string((uint64_t)((double)cells*1.0e-6/dt)))+"|"
			+centre(W[1], std::to_string((uint64_t)((double)cells*bytes_per_cell*1.0e-9/dt))+" GB/s")+"|"
			+centre(W[2], std::to_string((uint64_t)(1.0/dt)))+"|"+centre(W[3], cur)+"|"+centre(W[4], clock_text(m.remaining_seconds(t)))+"|";
	}
};

} // namespace luw_host
