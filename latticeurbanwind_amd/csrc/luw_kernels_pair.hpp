// luw_kernels_pair.hpp -- the FP16C product collide-stream kernel, k_stream_collide_p: two cells per lane, one dword per lane and plane.
// Device code of libluw_core.so; included by luw_core.hip only, behind luw_kernels_step.hpp (row addressing, ldo / sto).
#pragma once

// ---------------------------------------------------------------- pair kernel: 2 cells per lane (FP16C DDFs)
// With 2-byte DDFs the scalar kernel moves only 128 B per wave instruction, and that access width tops out near 5.2 TB/s
// (tools/membench half).  Here a lane owns the cells (x, x+1), x even, and moves both FP16C codes of a plane with ONE
// dword access -- the same bytes per instruction as the FP32 scalar kernel.  Straight planes are 4-byte aligned; the x+1
// planes are read/written at a 2-byte offset (a dword access on a 2-byte boundary, which the hardware serves), except for
// the lane at the row end whose second neighbour wraps to x = 0.  The two cells are collided ONE AFTER THE OTHER (an
// asm fence between the passes keeps the compiler from interleaving them), so the register footprint is that of the
// scalar kernel plus the 19 finished values of the first cell: 4 waves/SIMD.  Both cells are encoded at the tail under
// round-toward-zero (fp16c_code_hi_in_rtz_mode) and merged into dwords with one byte permute per plane.
// A cell that must not be processed (solid / halo) passes its populations through; its values are pre-swapped so that
// the Esoteric-Pull store puts them back where they came from (every slot has exactly one writing cell per step, so this
// rewrite races with nobody).  Requires an even b.x0, an even b.x1 (or b.x1 = an odd Nx: the row's last cell then pairs with the row padding and is the only
// one processed by its lane) and rows whose x = 0 sits on a 4-byte boundary (the host
// falls back to the scalar kernel otherwise).
typedef uint32_t u32_a2 __attribute__((aligned(2)));
template<bool NT> __device__ __forceinline__ uint32_t ld_pair(const uint16_t* plane, const uint32_t byte_off) {
	const char* ptr = reinterpret_cast<const char*>(plane)+byte_off;
	if constexpr(NT) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(ptr));
	else return *reinterpret_cast<const u32_a2*>(ptr);
}
template<bool NT> __device__ __forceinline__ void st_pair(uint16_t* plane, const uint32_t byte_off, const uint32_t v) {
	char* ptr = reinterpret_cast<char*>(plane)+byte_off;
	if constexpr(NT) __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(ptr));
	else *reinterpret_cast<u32_a2*>(ptr) = v;
}
// every value of v[0..N) passes through a volatile asm: what produces them is ordered before, what consumes them after
template<int N> __device__ __forceinline__ void asm_fence(float* v) {
	static_assert(N==19, "written for the 19 DDFs of a cell");
	asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
	asm volatile("" : "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]));
}
__device__ __forceinline__ void asm_fence9(float& f0, f32x2* v) {
	asm volatile("" : "+v"(f0), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
}
__device__ __forceinline__ void asm_fence_u(uint32_t* v) {
	asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
	asm volatile("" : "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]));
}
// FORCE (plain steps only): what can push the cells of the box -- see collide_cell_pk; the two specialised modes run 5 waves per SIMD instead of 4.
// PARK: LDS as an extension of the register file.  Each lane owns 19 dwords of LDS (slot q of a wave: 64 consecutive dwords, so every
// access is conflict-free).  The 19 raw dwords wait there while cell x is collided (only its decoded half is in registers), then trade
// places with the 19 finished values of cell x, which wait there while cell x+1 is collided and are encoded straight out of LDS at
// the tail.  76 DS operations per lane, none of them VALU work; the kernel's register footprint becomes that of the one-cell kernel.
#ifndef LUW_PARK_WAVES_NONE
#define LUW_PARK_WAVES_NONE 7
#endif
#ifndef LUW_PARK_WAVES_UNIFORM
#define LUW_PARK_WAVES_UNIFORM 6
#endif
#ifndef LUW_PARK_WAVES_ANY
#define LUW_PARK_WAVES_ANY 5
#endif
#ifndef LUW_THERMAL_WAVES_NONE
#define LUW_THERMAL_WAVES_NONE 5
#endif
#ifndef LUW_THERMAL_WAVES_UNIFORM
#define LUW_THERMAL_WAVES_UNIFORM 5
#endif
#ifndef LUW_THERMAL_WAVES_ANY
#define LUW_THERMAL_WAVES_ANY 4
#endif
#ifndef LUW_NATIVE_WAVES_NONE
#define LUW_NATIVE_WAVES_NONE 5
#endif
#ifndef LUW_NATIVE_WAVES_UNIFORM
#define LUW_NATIVE_WAVES_UNIFORM 5
#endif
#ifndef LUW_NATIVE_WAVES_ANY
#define LUW_NATIVE_WAVES_ANY 5
#endif
#ifndef LUW_NATIVE_THERMAL_WAVES_NONE
#define LUW_NATIVE_THERMAL_WAVES_NONE 5
#endif
#ifndef LUW_NATIVE_THERMAL_WAVES_UNIFORM
#define LUW_NATIVE_THERMAL_WAVES_UNIFORM 5
#endif
#ifndef LUW_NATIVE_THERMAL_WAVES_ANY
#define LUW_NATIVE_THERMAL_WAVES_ANY 4
#endif
constexpr int pair_waves(const int force, const bool park, const bool thermal = false, const bool native = false) {
	if(native) {
		if(thermal) return force==PAIR_FORCE_NONE ? LUW_NATIVE_THERMAL_WAVES_NONE : force==PAIR_FORCE_UNIFORM ? LUW_NATIVE_THERMAL_WAVES_UNIFORM
			: LUW_NATIVE_THERMAL_WAVES_ANY;
		if(force==PAIR_FORCE_ANY&&!park) return 4;   // (the sampled step's instantiation keeps both cells in registers, like the exact one)
		return force==PAIR_FORCE_NONE ? LUW_NATIVE_WAVES_NONE : force==PAIR_FORCE_UNIFORM ? LUW_NATIVE_WAVES_UNIFORM : LUW_NATIVE_WAVES_ANY;
	}
	// (a wave more each spills to scratch)
	if(thermal) return force==PAIR_FORCE_NONE ? LUW_THERMAL_WAVES_NONE : force==PAIR_FORCE_UNIFORM ? LUW_THERMAL_WAVES_UNIFORM : LUW_THERMAL_WAVES_ANY;
	return park ? (force==PAIR_FORCE_NONE ? LUW_PARK_WAVES_NONE : force==PAIR_FORCE_UNIFORM ? LUW_PARK_WAVES_UNIFORM : LUW_PARK_WAVES_ANY)
		: (force==PAIR_FORCE_ANY ? 4 : 5);
}
#ifndef LUW_PAIR_PREFETCH
#define LUW_PAIR_PREFETCH 1 /* general parked instantiation: nudging / sponge references fetched with the DDF loads (fetch_force_refs) */
#endif
#ifndef LUW_PAIR_OWN_EARLY
#define LUW_PAIR_OWN_EARLY 1
#endif
constexpr bool pair_prefetch(const int force, const bool park) { return LUW_PAIR_PREFETCH!=0 && park && force==PAIR_FORCE_ANY; }
constexpr uint32_t pair_park_bytes_per_wave(const bool thermal, const int force = PAIR_FORCE_NONE) {
	return ((thermal ? 26u : 19u)+(pair_prefetch(force, true) ? 8u : 0u))*64u*4u;
}
// THERMAL (LUW_OPT_TEMPERATURE): the D3Q7 lattice of both cells the same way -- seven more dwords per lane (plane 0 and the three (A, B) pairs of
// +x, +y, +z: the +x plane on a 2-byte boundary like the five x+1 planes of the D3Q19 lattice), the cell update of luw_device.hpp (thermal_cell)
// behind each collision with the velocity before the force shift, the seven codes of both cells merged per plane at the tail.
// NATIVE (LUW_OPT_NATIVE_ARITH): the collision in the hardware's own arithmetic (collide_cell_pk_native, luw_device.hpp); same memory path, same codec.
// XFACE: see k_stream_collide_s.  Here the first owned column (x = 1) is cell x of its lane, the last (x = Nx - 2) cell x + 1 of its lane.
// x-face INPUT (luw_set_x_face_inputs): every x-face instantiation but the uniform-force ones, which fill their 96 VGPRs (5 waves) without it -- five more
// live registers behind the loads spill a pair into scratch, and parked in LDS the kernel loses what the saved unpack kernel gains (2.355 against 2.30 ms on
// the FP16C + Coriolis rank of [4,2,1]).  Their launches have the library run the unpack kernel for their side.
// (With the thermal lattice the second cell's values are parked in LDS anyway and every force mode has the registers.)
constexpr bool pair_reads_x_face_inputs(const int force, const bool thermal = false) { return thermal || force!=PAIR_FORCE_UNIFORM; }
template<int PARITY, int MODE=0, bool STATS=false, int FORCE=PAIR_FORCE_ANY, bool PARK=false, bool THERMAL=false, bool NATIVE=false, bool XFACE=false>
__global__ __launch_bounds__(256)
	__attribute__((amdgpu_waves_per_eu(pair_waves(FORCE, PARK, THERMAL, NATIVE), pair_waves(FORCE, PARK, THERMAL, NATIVE))))
void k_stream_collide_p(const KParams p, const Box b, uint16_t* __restrict__ fi, float* __restrict__ rho, float* __restrict__ u,
		const uint8_t* __restrict__ flags, const float* __restrict__ F, const int write_fields, const StatsArgs S = StatsArgs{},
		uint16_t* __restrict__ gi = nullptr, float* __restrict__ Tf = nullptr, uint16_t* __restrict__ xf_p = nullptr, uint16_t* __restrict__ xf_m = nullptr,
			const uint16_t* __restrict__ xin_p = nullptr, const uint16_t* __restrict__ xin_m = nullptr) {
	static_assert(!(THERMAL&&STATS), "the thermal lattice keeps the separate statistics kernel");
	// (with the thermal lattice: the D3Q19 faces; the D3Q7 faces keep their kernels)
	static_assert(!XFACE||(!STATS&&MODE==0), "x-face output: plain steps");
	constexpr int NSLOT = THERMAL ? 26 : 19;                         // PARK: dwords per lane in LDS (PRE: eight more behind them)
	// (the native collision takes the zone terms from the prefetched references ONLY: its general instantiations always fetch, sampled steps included)
	constexpr bool PRE = ((pair_prefetch(FORCE, PARK) && !STATS) || (NATIVE && FORCE==PAIR_FORCE_ANY)) && MODE==0;
	constexpr bool OWN = LUW_PAIR_OWN_EARLY!=0 && !PRE;
	// RAW: the populations stay scaled by 2^-112 from the decode's shift-and-mask to the encode's (collide_cell_pk_native<FORCE, RAW>, luw_device.hpp): no
	// products in the codec, no switch of the rounding mode (the thermal lattice's exact cell update keeps the scaled-up form)
	constexpr bool RAW = NATIVE && !THERMAL;
	uint32_t bix, biy;
	xcd_row_order(p, bix, biy);
	const uint32_t x = b.x0+2u*(bix*blockDim.x+threadIdx.x), y = b.y0+biy, z = b.z0+blockIdx.z;
	if(x>=b.x1) return;
	const RowOff rb = row_offsets(p, y, z);
	LaneOff o = lane_offsets<uint16_t>(p, x);                      // offsets of cell x; cell x+1 sits 2 bytes further
	const bool wrap = x+2u==p.Nx;                                  // cell x+1 is the last of the row: its x+1 neighbour is x = 0
	const bool tail = x+1u==p.Nx;                                  // odd Nx: cell x is the last of the row, "cell x+1" is the row's padding
	const uint32_t n = x+(uint32_t)rb.r00;
	fi += rb.r00;                                                  // own row (uniform)
	const size_t Np = p.Np;
	const uint32_t fl2 = *reinterpret_cast<const uint16_t*>(flags+n);
	const uint8_t fl[2] = { (uint8_t)(fl2&0xFFu), (uint8_t)(fl2>>8) };
	bool proc[2];
	#pragma unroll
	for(int c=0; c<2; c++) proc[c] = !cell_is_halo(p, x+c, y, z) && (fl[c]&TYPE_BO)!=TYPE_S && (fl[c]&TYPE_SU)!=TYPE_G;
	if(tail) proc[1] = false;                                      // passes through: reads and rewrites padding, except on the x+1 planes (below)
	// STATS: a cell that is not updated but belongs to the lattice (solid) samples the fields it holds
	[[maybe_unused]] PairSample smp;
	[[maybe_unused]] auto sample_from_fields = [&](const int c) {
		smp.r[c] = rho[n+c];
		smp.ux[c] = u[n+c];
		smp.uy[c] = u[Np+n+c];
		smp.uz[c] = u[2ull*Np+n+c];
		smp.has[c] = true;
	};
	[[maybe_unused]] auto sample_idle_cell = [&](const int c) {
		smp.r[c] = smp.ux[c] = smp.uy[c] = smp.uz[c] = 0.0f;
		smp.has[c] = false;
		if(!(c==1&&tail) && !cell_is_halo(p, x+c, y, z)) sample_from_fields(c);
	};
	[[maybe_unused]] const size_t xfA = (size_t)p.Ny*p.Nz, xfe = (size_t)y+(size_t)z*p.Ny;   // XFACE: face area and this row's element of the face buffers
	// x-face inputs: the five values of a border lane's cell, fetched from the receive buffers ahead of the DDF loads and merged into the loaded dwords behind
	// them (first column: own slots, low halves; last column: the slots in the halo column, high halves)
	[[maybe_unused]] uint32_t xv[5];
	[[maybe_unused]] bool xin_first = false, xin_last = false;
	if constexpr(XFACE&&pair_reads_x_face_inputs(FORCE, THERMAL)) {
		if(xin_p||xin_m) {   // (each side on its own: a host may have handed over one of them only)
			xin_first = x==1u&&xin_m; xin_last = x+3u==p.Nx&&xin_p;
			if(xin_first||xin_last) {
				const uint16_t* const src = xin_first ? xin_m : xin_p;
				const int sg = xin_first ? -1 : 1;
				xv[0] = src[xface_in_elem(p, y, z, 0, 0)]; xv[1] = src[xfA+xface_in_elem(p, y, z, sg, 0)]; xv[2] = src[2u*xfA+xface_in_elem(p, y, z, -sg, 0)];
				xv[3] = src[3u*xfA+xface_in_elem(p, y, z, 0, sg)]; xv[4] = src[4u*xfA+xface_in_elem(p, y, z, 0, -sg)];
			}
		}
	}
	if(!proc[0]&&!proc[1]) {
		if constexpr(STATS) { // two idle cells (solid / halo / padding): constants, stored without arithmetic (stats_hold_constant_cell)
			if(!cell_is_halo(p, x, y, z)) stats_hold_constant_cell(Np, S, n, rho, u);
			if(!tail&&!cell_is_halo(p, x+1u, y, z)) stats_hold_constant_cell(Np, S, n+1u, rho, u);
		}
		if constexpr(XFACE) { // two cells that are not collided forward what their slots hold (a lane with one live cell does so by its pass-through)
			static_for_pairs([&](auto ic) {
				constexpr int i = decltype(ic)::value;
				if constexpr(i==1||i==7||i==9||i==13||i==15) {
					constexpr int k = i==1 ? 0 : i==7 ? 1 : i==13 ? 2 : i==9 ? 3 : 4;
					// first column (cell x): its own slot A(i) holds population i + 1; last column (cell x + 1): slot B(i) of its +c_i neighbour, at x + 2,
					// holds i
					// (x-face inputs: what those slots would hold sits in xv; the lattice takes it too -- a pack kernel may read these slots later)
					if(xin_first) fi[(size_t)slotA<PARITY>(i)*Np+x] = (uint16_t)xv[k];
					if(xin_last) *(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb)+(x+2u)) = (uint16_t)xv[k];
					if(x==1u) xf_m[(size_t)k*xfA+xfe] = xin_first ? (uint16_t)xv[k] : fi[(size_t)slotA<PARITY>(i)*Np+x];
					if(x+3u==p.Nx) xf_p[(size_t)k*xfA+xfe] = xin_last ? (uint16_t)xv[k] : *(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb)+(x+2u));
				}
			});
		}
		return;
	}
	// All 19 dword loads in one straight run.  For the row-end lane (wrap) the high half of the five x+1 dwords is the element
	// behind the row end (the next row's x = 0, or the slack behind the plane: in bounds, see lead_alloc / the plane skew):
	// it is replaced by the wrapped neighbour in ONE divergent fix-up block behind the loads.
	uint32_t raw[19];                                              // low half: cell x, high half: cell x+1
	raw[0] = ld_pair<true>(fi, o.x);
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
		raw[i] = ld_pair<true>(fi+(size_t)slotA<PARITY>(i)*Np, o.x);
		raw[i+1] = ld_pair<!shifted>(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb), nlane<i>(o));
	});
	if(wrap) {
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			if constexpr(i==1||i==7||i==9||i==13||i==15) { // the dword started at x+1 = Nx-1 of the neighbour row; x+2 wraps to that row's x = 0
				const uint32_t hi = *(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb));
				raw[i+1] = (raw[i+1]&0xFFFFu)|(hi<<16);
			}
		});
	}
	if constexpr(XFACE&&pair_reads_x_face_inputs(FORCE, THERMAL)) {
		if(xin_first||xin_last) {
			static_for_pairs([&](auto ic) {
				constexpr int i = decltype(ic)::value;
				if constexpr(i==1||i==7||i==9||i==13||i==15) {
					constexpr int k = i==1 ? 0 : i==7 ? 1 : i==13 ? 2 : i==9 ? 3 : 4;
					if(xin_first) raw[i] = (raw[i]&0xFFFF0000u)|xv[k];
					if(xin_last) raw[i+1] = (raw[i+1]&0xFFFFu)|(xv[k]<<16);
				}
			});
		}
	}
	[[maybe_unused]] uint32_t rawg[7];                               // THERMAL: the same for the seven planes of the temperature lattice
	if constexpr(THERMAL) {
		gi += rb.r00;
		rawg[0] = ld_pair<true>(gi, o.x);
		rawg[1] = ld_pair<true>(gi+(size_t)slotA<PARITY>(1)*Np, o.x); rawg[2] = ld_pair<false>(gi+(size_t)slotB<PARITY>(1)*Np+nrow<1>(rb), nlane<1>(o));
		rawg[3] = ld_pair<true>(gi+(size_t)slotA<PARITY>(3)*Np, o.x); rawg[4] = ld_pair<true>(gi+(size_t)slotB<PARITY>(3)*Np+nrow<3>(rb), nlane<3>(o));
		rawg[5] = ld_pair<true>(gi+(size_t)slotA<PARITY>(5)*Np, o.x); rawg[6] = ld_pair<true>(gi+(size_t)slotB<PARITY>(5)*Np+nrow<5>(rb), nlane<5>(o));
		if(wrap) { const uint32_t hi = *(gi+(size_t)slotB<PARITY>(1)*Np+nrow<1>(rb)); rawg[2] = (rawg[2]&0xFFFFu)|(hi<<16); }
	}
	// PRE: the nudging / sponge references of both cells go out behind the DDF loads; those of cell x+1 wait in LDS like its raw dwords.  (In front of
	// the row-end lane's fix-up, which waits for the DDF loads, they were no faster: profiles/r03_stall_counters.md.)
	[[maybe_unused]] ForceRefs refs[2];
	if constexpr(PRE) {
		#pragma unroll
		for(int c=0; c<2; c++) fetch_force_refs(p, n+c, x+c, y, z, proc[c], (fl[c]&TYPE_BO)==TYPE_E, rho, u, refs[c]);
	}
	// wave-uniform: can any cell of this wave feel a force (then the Guo terms are computed for the whole wave)?
	// (PRE: the fetch above has already decided, cell by cell, whether a zone term acts)
	bool zone_lane;
	if constexpr(PRE) zone_lane = refs[0].zn||refs[0].zs||refs[1].zn||refs[1].zs; else zone_lane = in_force_zone(p, x, y, z)||in_force_zone(p, x+1u, y, z);
	const bool may_force = FORCE==PAIR_FORCE_ANY && (p.coriolis || p.has_F || p.fx!=0.0f || p.fy!=0.0f || p.fz!=0.0f || __ballot(zone_lane)!=0ull);
	// specialised modes: TYPE_E cells decode to f = 0 (collide_cell_pk relaxes them with w = 1)
	constexpr bool E_BY_RATE = NATIVE || FORCE!=PAIR_FORCE_ANY;
	// (only cells that are collided: a halo or padding cell passes what it decodes through unchanged, whatever its flag)
	[[maybe_unused]] const uint32_t dmask[2] = { (E_BY_RATE&&proc[0]&&(fl[0]&TYPE_BO)==TYPE_E) ? 0u : 0x87FFF000u,
		(E_BY_RATE&&proc[1]&&(fl[1]&TYPE_BO)==TYPE_E) ? 0u : 0x87FFF000u };
	// one cell: decode its half of the 19 dwords into f0 and the nine (f[2k+1], f[2k+2]) pairs, collide on the packed pairs
	// (or pre-swap for the pass-through)
	auto one_cell = [&](const int c, float& f0, f32x2* fp, [[maybe_unused]] float* g) {
		auto bits = [&](const int q) { // (sign-extended half) << 12 in one SDWA shift, then the mask of half_to_float_custom_sx
			uint32_t t;
			if(c) asm("v_lshlrev_b32_sdwa %0, 12, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(t) : "v"(raw[q]));
			else asm("v_lshlrev_b32_sdwa %0, 12, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(t) : "v"(raw[q]));
			if constexpr(E_BY_RATE) return t&dmask[c]; else return t&0x87FFF000u;
		};
		// OWN (the instantiations without the fetch above): a TYPE_E cell's own rho and u go out HERE, in front of its decode, instead of behind its moments
		[[maybe_unused]] ForceRefs own;
		if constexpr(OWN) {
			asm volatile("" : "=v"(own.tu[0]), "=v"(own.tu[1]), "=v"(own.tu[2]), "=v"(own.wb));
			if(MODE!=1&&proc[c]&&(fl[c]&TYPE_BO)==TYPE_E) { own.wb = rho[n+c]; own.tu[0] = u[n+c]; own.tu[1] = u[Np+n+c]; own.tu[2] = u[2ull*Np+n+c]; }
		}
		f0 = RAW ? __uint_as_float(bits(0)) : __uint_as_float(bits(0))*0x1p+112f;
		#pragma unroll
		for(int k=0; k<9; k++) {
			const f32x2 t = { __uint_as_float(bits(2*k+1)), __uint_as_float(bits(2*k+2)) };
			if constexpr(RAW) fp[k] = t; else fp[k] = t*splat2(0x1p+112f);
		}
		if constexpr(THERMAL) {
			#pragma unroll
			for(int q=0; q<7; q++) {
				uint32_t t;
				if(c) asm("v_lshlrev_b32_sdwa %0, 12, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(t) : "v"(rawg[q]));
				else asm("v_lshlrev_b32_sdwa %0, 12, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(t) : "v"(rawg[q]));
				g[q] = __uint_as_float(t&0x87FFF000u)*0x1p+112f;
			}
		}
		if(MODE!=1&&proc[c]) { // MODE 1: measurement-only, no collision (every cell passes through)
			float rhon, uxn, uyn, uzn;
			[[maybe_unused]] float u0[3];
			// (native: rho / u of the last step of a run are stored from inside the collision, as soon as they are final)
			auto store_fields = [&](const float r_, const float ux_, const float uy_, const float uz_) {
				if(write_fields && (fl[c]&TYPE_BO)!=TYPE_E) {
					uint32_t nw = n+(uint32_t)c;
					asm volatile("" : "+v"(nw));
					// (+ 0: a zero velocity is stored as +0 whatever the signs of the zero populations it came from -- those depend on the kernel that ran)
					rho[nw] = r_; u[nw] = ux_+0.0f; u[Np+nw] = uy_+0.0f; u[2ull*Np+nw] = uz_+0.0f;
				}
			};
			if constexpr(NATIVE) collide_cell_pk_native<FORCE, RAW>(p, n+c, fl[c], may_force, f0, fp, rho, u, F, rhon, uxn, uyn, uzn, THERMAL ? u0 : nullptr,
				PRE ? &refs[c] : nullptr, PRE ? &refs[c] : OWN ? &own : nullptr, store_fields);
			else collide_cell_pk<FORCE>(p, n+c, x+c, y, z, fl[c], may_force, f0, fp, rho, u, F, rhon, uxn, uyn, uzn, THERMAL ? u0 : nullptr, PRE ? &refs[c]
				: nullptr, PRE ? &refs[c] : OWN ? &own : nullptr);
			if constexpr(THERMAL) thermal_cell(p, n+c, x+c, y, z, fl[c], u0[0], u0[1], u0[2], Tf, g, write_fields!=0);
			if(!NATIVE && write_fields && (fl[c]&TYPE_BO)!=TYPE_E) {
				// (the index passes through an empty asm: its 64-bit address arithmetic is then done HERE, in the block of the last step of a run, instead of
				// being
				// hoisted in front of both collisions, where the pair of registers it occupied made the uniform-force kernel spill to scratch at 96 VGPRs)
				uint32_t nw = n+(uint32_t)c;
				asm volatile("" : "+v"(nw));
				rho[nw] = rhon;
				u[nw] = uxn;
				u[Np+nw] = uyn;
				u[2ull*Np+nw] = uzn;
			}
			if constexpr(STATS) {
				if((fl[c]&TYPE_BO)==TYPE_E) sample_from_fields(c);
				// (native: a zero velocity is sampled as +0, the value the field write stores)
				else if constexpr(NATIVE) { smp.r[c] = rhon; smp.ux[c] = uxn+0.0f; smp.uy[c] = uyn+0.0f; smp.uz[c] = uzn+0.0f; smp.has[c] = true; }
				else { smp.r[c] = rhon; smp.ux[c] = uxn; smp.uy[c] = uyn; smp.uz[c] = uzn; smp.has[c] = true; }
			}
		} else {
			#pragma unroll
			for(int k=0; k<9; k++) { const f32x2 t = { fp[k].y, fp[k].x }; fp[k] = t; }
			if constexpr(THERMAL) { for(int k=0; k<3; k++) { const float t = g[2*k+1]; g[2*k+1] = g[2*k+2]; g[2*k+2] = t; } }
			if constexpr(STATS) sample_idle_cell(c);
		}
	};
	float fa0, fb0; f32x2 fa[9], fb[9];
	[[maybe_unused]] float ga[7], gb[7];
	[[maybe_unused]] uint32_t* slot = nullptr;                     // PARK: this lane's column in its wave's LDS region, slot[64*q]
	if constexpr(PARK) {
		extern __shared__ uint32_t pair_park[];
		slot = pair_park+(threadIdx.x>>6)*((uint32_t)(NSLOT+(PRE ? 8 : 0))*64u)+(threadIdx.x&63u);
		#pragma unroll
		for(int q=0; q<19; q++) slot[64*q] = raw[q];
		if constexpr(THERMAL) { for(int q=0; q<7; q++) slot[64*(19+q)] = rawg[q]; }
		if constexpr(PRE) {
			float* const fs = reinterpret_cast<float*>(slot)+64*NSLOT;
			fs[0] = refs[1].tu[0]; fs[64] = refs[1].tu[1]; fs[128] = refs[1].tu[2]; fs[192] = refs[1].wb;
			fs[256] = refs[1].su[0]; fs[320] = refs[1].su[1]; fs[384] = refs[1].su[2]; fs[448] = refs[1].sg;
		}
		asm volatile("" ::: "memory");
	}
	one_cell(0, fa0, fa, ga);
	if constexpr(PARK) {
		asm_fence9(fa0, fa);                                       // cell x is finished ...
		asm volatile("" ::: "memory");
		// ... and trades places with the raw dwords: one value out, one in, so that the two sets never sit in registers together
		{ const uint32_t t = slot[0]; slot[0] = __float_as_uint(fa0); raw[0] = t; }
		#pragma unroll
		for(int k=0; k<9; k++) {
			const uint32_t t0 = slot[64*(2*k+1)], t1 = slot[64*(2*k+2)];
			slot[64*(2*k+1)] = __float_as_uint(fa[k].x); slot[64*(2*k+2)] = __float_as_uint(fa[k].y);
			raw[2*k+1] = t0; raw[2*k+2] = t1;
		}
		if constexpr(THERMAL) { for(int q=0; q<7; q++) { const uint32_t t = slot[64*(19+q)]; slot[64*(19+q)] = __float_as_uint(ga[q]); rawg[q] = t; } }
		if constexpr(PRE) {
			const float* const fs = reinterpret_cast<const float*>(slot)+64*NSLOT;
			refs[1].tu[0] = fs[0]; refs[1].tu[1] = fs[64]; refs[1].tu[2] = fs[128]; refs[1].wb = fs[192];
			refs[1].su[0] = fs[256]; refs[1].su[1] = fs[320]; refs[1].su[2] = fs[384]; refs[1].sg = fs[448];
		}
		asm volatile("" ::: "memory");
		asm_fence_u(raw);
	} else {
		asm_fence9(fa0, fa); asm_fence_u(raw);                     // cell x is finished before cell x+1 starts
	}
	one_cell(1, fb0, fb, gb);
	if constexpr(STATS) stats_welford_pair(Np, S, n, smp);        // both cells' samples, one 8-byte access per array
	if constexpr(!PARK) asm_fence9(fa0, fa);
	asm_fence9(fb0, fb);                                           // all floating-point work is done ...
	if constexpr(THERMAL) {
		asm volatile("" : "+v"(gb[0]), "+v"(gb[1]), "+v"(gb[2]), "+v"(gb[3]), "+v"(gb[4]), "+v"(gb[5]), "+v"(gb[6]));
		if constexpr(!PARK) asm volatile("" : "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]), "+v"(ga[4]), "+v"(ga[5]), "+v"(ga[6]));
	}
	// ... before the wave's FP32 rounding mode becomes RTZ (see luw_device.hpp; RAW: the encode has no floating-point operation, the mode stays)
	if constexpr(RAW) { if constexpr(PARK) asm volatile("" ::: "memory"); }
	else if constexpr(PARK) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3" ::: "memory");
	else asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
	// codes of both cells in the high halves, merged per plane: q = 0 or 2k+1+h
	uint32_t ca[19], cb[19];
	auto code1 = [&](const float v) { if constexpr(RAW) return fp16c_code_hi_of_scaled(v); else return fp16c_code_hi_in_rtz_mode(v); };
	auto code2 = [&](const f32x2 v, uint32_t& c0, uint32_t& c1) {
		if constexpr(RAW) { c0 = fp16c_code_hi_of_scaled(v.x); c1 = fp16c_code_hi_of_scaled(v.y); } else fp16c_code2_hi_in_rtz_mode(v, c0, c1);
	};
	cb[0] = code1(fb0);
	#pragma unroll
	for(int k=0; k<9; k++) code2(fb[k], cb[2*k+1], cb[2*k+2]);
	if constexpr(PARK) {
		ca[0] = code1(__uint_as_float(slot[0]));
		#pragma unroll
		for(int k=0; k<9; k++) {
			const f32x2 v = { __uint_as_float(slot[64*(2*k+1)]), __uint_as_float(slot[64*(2*k+2)]) };
			code2(v, ca[2*k+1], ca[2*k+2]);
		}
	} else {
		ca[0] = code1(fa0);
		#pragma unroll
		for(int k=0; k<9; k++) code2(fa[k], ca[2*k+1], ca[2*k+2]);
	}
	[[maybe_unused]] uint32_t cg[7];                               // THERMAL: the merged dwords of the temperature lattice
	if constexpr(THERMAL) {
		#pragma unroll
		for(int q=0; q<7; q++) {
			float a;
			if constexpr(PARK) a = __uint_as_float(slot[64*(19+q)]); else a = ga[q];
			cg[q] = __builtin_amdgcn_perm(fp16c_code_hi_in_rtz_mode(gb[q]), fp16c_code_hi_in_rtz_mode(a), 0x07060302u);
		}
	}
	if constexpr(XFACE) { // codes sit in the high halves: ca = cell x, cb = cell x + 1
		if(x==1u) {
			xf_m[xfe] = (uint16_t)(ca[2]>>16); xf_m[xfA+xfe] = (uint16_t)(ca[8]>>16); xf_m[2u*xfA+xfe] = (uint16_t)(ca[14]>>16);
			xf_m[3u*xfA+xfe] = (uint16_t)(ca[10]>>16); xf_m[4u*xfA+xfe] = (uint16_t)(ca[16]>>16);
		}
		if(x+3u==p.Nx) {
			xf_p[xfe] = (uint16_t)(cb[1]>>16); xf_p[xfA+xfe] = (uint16_t)(cb[7]>>16); xf_p[2u*xfA+xfe] = (uint16_t)(cb[13]>>16);
			xf_p[3u*xfA+xfe] = (uint16_t)(cb[9]>>16); xf_p[4u*xfA+xfe] = (uint16_t)(cb[15]>>16);
		}
	}
	auto pack = [&](const int q) { return __builtin_amdgcn_perm(cb[q], ca[q], 0x07060302u); };
	uint32_t cs[5];   // the five x+1 planes, stored last (dword or, on the row-end lane, two halves)
	// General kernel: the plane bases of the stores are formed again from a plane stride the compiler cannot connect with the one the loads used, so
	// that the 19 64-bit bases of the loads die with the loads instead of living through both collisions -- with the zone parameters on top they
	// do not fit the scalar registers and went into lanes of a spill VGPR (40 v_writelane + 88 v_readlane per lane, static; now 12 + 24, and 92
	// VALU instructions fewer).  The other instantiations have room (no spills) and keep the bases (the rebuilt ones cost 150 scalar instructions).
	size_t Np_tail = p.Np;
	if constexpr(FORCE==PAIR_FORCE_ANY) asm volatile("" : "+s"(Np_tail));
	#define Np Np_tail
	#define LUW_REDEFINE_OFFSETS asm volatile("" : "+v"(o.x), "+v"(o.xp)) /* saddr stores, see k_stream_collide_s */
	LUW_REDEFINE_OFFSETS;
	st_pair<true>(fi, o.x, pack(0));
	static_for_pairs([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		constexpr bool shifted = i==1||i==7||i==9||i==13||i==15;
		constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
		if constexpr(!shifted) st_pair<true>(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb), nlane<i>(o), pack(i));
		else cs[k] = pack(i);
		st_pair<true>(fi+(size_t)slotA<PARITY>(i)*Np, o.x, pack(i+1));
	});
	if constexpr(THERMAL) { // the aligned planes of the temperature lattice; its +x plane goes with the five x+1 planes below
		st_pair<true>(gi, o.x, cg[0]);
		st_pair<true>(gi+(size_t)slotA<PARITY>(1)*Np, o.x, cg[2]);
		st_pair<true>(gi+(size_t)slotB<PARITY>(3)*Np+nrow<3>(rb), nlane<3>(o), cg[3]); st_pair<true>(gi+(size_t)slotA<PARITY>(3)*Np, o.x, cg[4]);
		st_pair<true>(gi+(size_t)slotB<PARITY>(5)*Np+nrow<5>(rb), nlane<5>(o), cg[5]); st_pair<true>(gi+(size_t)slotA<PARITY>(5)*Np, o.x, cg[6]);
	}
	if(!wrap&&!tail) {
		LUW_REDEFINE_OFFSETS;
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
			if constexpr(i==1||i==7||i==9||i==13||i==15) st_pair<false>(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb), nlane<i>(o), cs[k]);
		});
		if constexpr(THERMAL) st_pair<false>(gi+(size_t)slotB<PARITY>(1)*Np+nrow<1>(rb), nlane<1>(o), cg[1]);
	} else if(tail) { // the only real cell is x = Nx-1: its x+1 neighbour is the row's x = 0 (the dword load above already started there);
		// x = 1 belongs to another lane's stores, so only the low half goes out
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			if constexpr(i==1||i==7||i==9||i==13||i==15) {
				constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
				*(fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb)) = (uint16_t)(cs[k]&0xFFFFu);
			}
		});
		if constexpr(THERMAL) *(gi+(size_t)slotB<PARITY>(1)*Np+nrow<1>(rb)) = (uint16_t)(cg[1]&0xFFFFu);
	} else {
		LUW_REDEFINE_OFFSETS;
		static_for_pairs([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			if constexpr(i==1||i==7||i==9||i==13||i==15) {
				constexpr int k = i==1 ? 0 : i==7 ? 1 : i==9 ? 2 : i==13 ? 3 : 4;
				uint16_t* B = fi+(size_t)slotB<PARITY>(i)*Np+nrow<i>(rb);   // row of the neighbours: x+1 = Nx-1 is its last cell, x+2 its first
				B[p.Nx-1u] = (uint16_t)(cs[k]&0xFFFFu);
				B[0] = (uint16_t)(cs[k]>>16);
			}
		});
		if constexpr(THERMAL) {
			uint16_t* B = gi+(size_t)slotB<PARITY>(1)*Np+nrow<1>(rb);
			B[p.Nx-1u] = (uint16_t)(cg[1]&0xFFFFu);
			B[0] = (uint16_t)(cg[1]>>16);
		}
	}
	#undef LUW_REDEFINE_OFFSETS
	#undef Np
}
