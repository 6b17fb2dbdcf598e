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
// Options after the deck path (the reference ignores extra args): --ddf fp32|fp16c (default fp16c = shipped build), --arith native|exact (FP16C; default
// native),
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

#include "driver_state.hpp"
#include "driver_setup.hpp"
#include "driver_case.hpp"
#include "driver_boundaries.hpp"
#include "driver_run.hpp"
#include "driver_output.hpp"

// ------------------------------------------------------------------------------------------------ main
int main(int argc, char** argv) {
	Driver d;
	if(!parse_command_line(argc, argv, d.c)) return -1;
	println(hr_plain());
	println("|"+alignc(CONSOLE_WIDTH-2u, "LatticeUrbanWind LUW core for AMD Instinct MI355X (HIP, D3Q19 SRT + Smagorinsky)")+"|");
	println(hr_plain());
	read_deck(d.c);
	d.read_probe_requests();
	d.print_parameters();
	d.size_lattice();
	d.read_inflow_inputs();
	d.set_units_and_forcing();
	if(d.c.sizing_only) { println(hr_plain()); return 0; }
	d.load_geometry();
	d.build_profile_table();
	d.list_cases();
	for(const Driver::Case& cs : d.cases) d.run_case(cs);
	println(hr_plain());
	return 0;
}
