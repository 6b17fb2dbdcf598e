// vtk_writer.hpp -- VTK file names, headers and the two payload paths (devices: LBM::export_vtk; host conversion: LUW_HOST_VTK=1)
// Part of the deck driver (luw_driver.cpp); included by it only, after lbm.hpp (namespace luw_host, std::string as string).
#pragma once

// ------------------------------------------------------------------------------------------------ VTK writers
static string default_filename(const string& path, const string& name, const ulong t) { // FX/lbm.cpp:235-239
	string time = "00000000"+to_string_u(t);
	time = time.substr(time.length()-9u, 9u);
	return path+name+"-"+time+".vtk";
}
struct VtkGeom { uint Nx, Ny, Nz, Nz_out; float spacing; float origin[3]; };
static string vtk_header(const string& filename, const VtkGeom& g) {
	const string base = filename.substr(filename.find_last_of("/\\")+1u);
	const ulong points = (ulong)g.Nx*(ulong)g.Ny*(ulong)g.Nz_out;
	return "# vtk DataFile Version 3.0\nFluidX3D "+base+"\nBINARY\nDATASET STRUCTURED_POINTS\n"
		"DIMENSIONS "+to_string_u(g.Nx)+" "+to_string_u(g.Ny)+" "+to_string_u(g.Nz_out)+"\n"
		"ORIGIN "+to_string_f(g.origin[0])+" "+to_string_f(g.origin[1])+" "+to_string_f(g.origin[2])+"\n"
		"SPACING "+to_string_f(g.spacing)+" "+to_string_f(g.spacing)+" "+to_string_f(g.spacing)+"\n"
		"POINT_DATA "+to_string_u(points)+"\n";
}
// Memory_Container::write_vtk (FX/lbm.hpp:307-356): SoA host field -> AoS big-endian floats in SI units
static void write_field_vtk(const string& filename, const VtkGeom& g, const float* data, const uint comps, const float factor, const float offset = 0.0f,
	const bool affine = false) {
	std::filesystem::create_directories(std::filesystem::path(filename).parent_path());
	std::ofstream file(filename, std::ios::out|std::ios::binary);
	const string header = vtk_header(filename, g)+"SCALARS data float "+to_string_u(comps)+"\nLOOKUP_TABLE default\n";
	file.write(header.c_str(), (std::streamsize)header.length());
	const ulong N = (ulong)g.Nx*g.Ny*g.Nz, points = (ulong)g.Nx*g.Ny*g.Nz_out;
	std::unique_ptr<float[]> conv(new float[points*comps]); float* const buf = conv.get(); // every element is written below: no value-initialisation
	// the field named T goes through units.si_T (value*unit_K + offset), every other one through its unit factor (FX/lbm.hpp:343)
	parallel_for(points, [&](const ulong i) {
		for(uint d=0u; d<comps; d++) buf[i*comps+d] = reverse_bytes(affine ? data[(ulong)d*N+i]*factor+offset : factor*data[(ulong)d*N+i]);
	});
	file.write((const char*)buf, (std::streamsize)(points*comps*4ull));
}

// The same files without the host in the data path (default; LUW_HOST_VTK=1 keeps the host conversion above, the cross-check of the tests): header by
// this process, payload by the devices through LBM::export_vtk -- no field download, no global conversion buffer.
static bool host_vtk_path() { static const bool on = std::getenv("LUW_HOST_VTK")&&std::getenv("LUW_HOST_VTK")[0]=='1'; return on; }
struct VtkFile { // an open output file and the offset of its next byte
	int fd = -1; uint64_t pos = 0ull;
	explicit VtkFile(const string& filename) {
		std::filesystem::create_directories(std::filesystem::path(filename).parent_path());
		fd = ::open(filename.c_str(), O_WRONLY|O_CREAT|O_TRUNC, 0644);
		if(fd<0) fatal("ERROR: cannot open "+filename+" for writing.");
	}
	~VtkFile() { if(fd>=0) ::close(fd); }
	void text(const string& t) {
		size_t put = 0u;
		while(put<t.size()) {
			const ssize_t w = ::pwrite(fd, t.data()+put, t.size()-put, (off_t)(pos+put));
			if(w<=0) fatal("ERROR: writing a VTK header failed.");
			put += (size_t)w;
		}
		pos += t.size();
	}
	void payload(LBM& lbm, const int source, const luw_export_params& prm, const VtkGeom& g, const uint comps) {
		lbm.export_vtk(source, prm, g.Nz_out, fd, pos);
		pos += (uint64_t)g.Nx*g.Ny*g.Nz_out*comps*4ull;
	}
};
static luw_export_params export_params(const float factor, const float offset = 0.0f, const bool affine = false) {
	luw_export_params p{}; p.struct_size = sizeof(p); p.factor = factor; p.offset = offset; p.affine = affine ? 1 : 0; return p;
}
static void write_device_field_vtk(LBM& lbm, const string& filename, const VtkGeom& g, const int source, const uint comps, const float factor,
	const float offset = 0.0f, const bool affine = false) {
	VtkFile f(filename);
	f.text(vtk_header(filename, g)+"SCALARS data float "+to_string_u(comps)+"\nLOOKUP_TABLE default\n");
	f.payload(lbm, source, export_params(factor, offset, affine), g, comps);
}

