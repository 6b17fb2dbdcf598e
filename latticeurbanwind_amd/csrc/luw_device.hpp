// luw_device.hpp -- device-side building blocks of the D3Q19 collide-stream step for gfx950 (wave64).
//
// What is computed follows the reference kernel `stream_collide` (FX/kernel.cpp:1475-1780, with the
// device functions it calls: f_eq :1016-1055, moments :1075-1100, Guo forcing :1103-1113, FP16C codec
// :864-875).  HOW it is computed is ours: coordinates come from the launch geometry (no div/mod per cell),
// the sin^2 ramps of the nudging/sponge terms come from host-built tables, DDFs are addressed as per-plane
// base pointer + 32-bit cell offset, and the vector kernels move 4 cells per lane.
//
// Arithmetic contract (shared with the CPU oracle used by the tests): FP32, fmaf() exactly where the
// reference writes fma(), every other operation individually rounded (this translation unit is compiled with
// -ffp-contract=off), IEEE-correct division and square root (hipcc default).  Zero-valued terms of the
// c_i-weighted sums are dropped: adding (+-0) never changes a value, so results are value-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace luw {

constexpr float DEF_W0 = 1.0f/3.0f;   // FX/lbm.cpp:674-676
constexpr float DEF_WS = 1.0f/18.0f;
constexpr float DEF_WE = 1.0f/36.0f;
constexpr float DEF_C = 0.57735027f;  // FX/lbm.cpp:663

constexpr uint8_t TYPE_S = 0x01, TYPE_E = 0x02, TYPE_BO = 0x03, TYPE_T = 0x04, TYPE_SU = 0x38, TYPE_G = 0x20;

// Everything the kernels need besides the big arrays; passed by value (lands in SGPRs).
struct KParams {
	uint32_t Nx, Ny, Nz;      // local lattice
	uint32_t Px;              // x pitch of every device array (multiple of 64, >= Nx)
	uint32_t Np;              // plane stride = Px*Ny*Nz (< 2^32, checked on the host)
	uint32_t halo_x, halo_y, halo_z; // 1 if that axis is split over domains (cells 0 and N-1 are halo: FX/kernel.cpp:856-859)
	int32_t Ox, Oy, Oz;
	float w;                  // def_w
	float tau0, tau0sq;       // 1/w and its square as the kernel would compute them (IEEE division / product, done once on the host)
	float fx, fy, fz;
	float omx, omy, omz;
	uint32_t coriolis;        // any component of omega nonzero
	uint32_t subgrid;
	// BUFFER_NUDGING / TOP_SPONGE, FX/lbm.cpp:613-625,770-782
	uint32_t buffer_active, buffer_N, nudge_vertical, downstream_face;
	float buffer_inv_tau;
	uint32_t sponge_active, sponge_N;
	uint32_t Nxg, Nyg, Nzg;   // global extents
	int32_t west_x, east_x, south_y, north_y, top_z;   // local coordinates of the global outer faces
	uint32_t has_w, has_e, has_s, has_n, has_t;
	const float* wbuf;        // wbuf[d] = sin^2(pi/2 (1 - d/Nbuf)),           d = 0..Nbuf   (FX/kernel.cpp:1581-1583)
	const float* sigma;       // sigma[d] = inv_tau sin^2(pi/2 (1 - d/(Ns-1))), d = 0..Ns-1  (FX/kernel.cpp:1604-1606)
	// The same zones as cell ranges of THIS domain, filled by the host (luw_core.hip, set_zone_ranges): a coordinate c lies in a zone when
	// (c - lo) < n as unsigned numbers; n = 0 where the zone does not exist here (term off, face not owned, downstream face).  One subtraction and
	// one compare per face instead of the four conditions of FX/kernel.cpp:1537-1541 each; same cells.
	uint32_t zw_lo, zw_n, ze_lo, ze_n;   // buffer nudging next to the west / east face (x)
	uint32_t zs_lo, zs_n, zn_lo, zn_n;   // south / north face (y)
	uint32_t zt_lo, zt_n;                // top (z)
	uint32_t zp_lo, zp_n;                // top sponge layers (z)
	uint32_t has_F;
	float w_T;                // TEMPERATURE: def_w_T = 1/(2 alpha + 1/2), FX/lbm.cpp:750 (0 when the thermal lattice is off)
	// native-arithmetic kernels (LUW_OPT_NATIVE_ARITH, collide_cell_pk_native): host-folded constants
	float half_tau0;          // tau0 / 2
	float m2omx, m2omy, m2omz; // -2 omega
};

// ---------------------------------------------------------------- FP16C codec, FX/kernel.cpp:864-875
// The kernels' codec is bit-identical to the reference's formulas for every input: checked exhaustively on the GPU
// (tests/test_gpu_parity.py::test_fp16c_codec_exhaustive: all 2^16 codes, all 2^32 floats) against the literal
// restatements *_ref below.
__device__ __forceinline__ float half_to_float_custom_ref(const uint32_t x) { // literal restatement, FX/kernel.cpp:864-869
	const uint32_t e = (x&0x7800u)>>11;
	const uint32_t m = (x&0x07FFu)<<12;
	const uint32_t v = __float_as_uint((float)m)>>23;
	return __uint_as_float((x&0x8000u)<<16 | (uint32_t)(e!=0u)*((e+112u)<<23|m) | (uint32_t)((e==0u)&(m!=0u))*((v-37u)<<23|((m<<((150u-v)&31u))&0x007FF000u)));
}
__device__ __forceinline__ uint32_t float_to_half_custom_ref(const float x) { // literal restatement, FX/kernel.cpp:870-875
	const uint32_t b = __float_as_uint(x)+0x00000800u;
	const uint32_t e = (b&0x7F800000u)>>23;
	const uint32_t m = b&0x007FFFFFu;
	return (b&0x80000000u)>>16 | (uint32_t)(e>112u)*((((e-112u)<<11)&0x7800u)|m>>12) | (uint32_t)((e<113u)&(e>100u))*((((0x007FF800u+m)>>((124u-e)&31u))
		+1u)>>1);
}
// Decode: the 15 exponent+mantissa bits placed at float bits 12..26 form a tiny float 2^(e-127)(1+m/2048) (or, for
// e = 0, the float DENORMAL m 2^-137); one exact multiplication by 2^112 turns both into the FP16C value
// 2^(e-15)(1+m/2048) resp. m 2^-25 -- the same numbers the reference builds with its integer formula
// (FX/kernel.cpp:864-869).  Needs FP32 denormals enabled (hipcc default).  No branches: DDF values are small deviations
// from equilibrium, so FP16C denormals (|f| < 6.1e-5) are common and branching on them costs more than it saves
// (measured).  The code arrives SIGN-EXTENDED (global_load_sshort does that for free), so bit 31 already holds the sign
// after the shift and one mask clears the copies of it that landed in the high exponent bits: shift, and, multiply.
__device__ __forceinline__ float half_to_float_custom_sx(const int32_t xs) {
	return __uint_as_float(((uint32_t)xs<<12)&0x87FFF000u)*0x1p+112f;
}
__device__ __forceinline__ float half_to_float_custom(const uint32_t x) { return half_to_float_custom_sx((int32_t)(int16_t)(uint16_t)x); }
// Encode: the reference formula (FX/kernel.cpp:870-875) rounds |x| half away from zero onto the FP16C grid.  With
// v = |x| 2^25 (an exact exponent shift):
//   rn = (bits(v) + 0x800 - (137<<23)) >> 12 (arithmetic) is the reference's normal-range code (add 0x800, drop 12 mantissa
//        bits, rebias by 112; the carry runs into the exponent field by itself); it is negative below 2^-15;
//   rd = floor(v + 1/2) is its denormal-range code, the integer m = round_half_up(|x| 2^25) (V_CVT_RPI_I32_F32 rounds
//        exactly that way, without an intermediate float sum).
// rd grows linearly and rn logarithmically with |x|, and they coincide on the first normal binade [2^-14, 2^-13), where
// the FP16C grid spacing equals the denormal spacing.  So rn <= rd everywhere, both are >= 2048 from 2^-14 up and rd <= 2048
// below: the median of (rn, rd, 2048) is rn for normal and rd for denormal magnitudes -- no compare/select.
// The result is left in the HIGH half of the register (sign already in place at bit 31, low half unspecified) for
// global_store_short_d16_hi; float_to_half_custom() shifts it down for callers that want the code as a number.
// Equal to the literal formula for every float with |x| < 2^103 including denormals, the carry cases next to 2^-14 and
// the 4-bit exponent wrap from |x| >= 2 (checked exhaustively on the device, luw_selfcheck_fp16c_codec); beyond that (v
// overflows; NaN) the codes differ -- a lattice holding such values has long since blown up.
__device__ __forceinline__ uint32_t float_to_half_custom_hi(const float x) {
	const float v = fabsf(x)*0x1p+25f;
	int32_t rd, mag;
	asm("v_cvt_rpi_i32_f32_e32 %0, %1" : "=v"(rd) : "v"(v));
	const int32_t rn = (int32_t)(__float_as_uint(v)+(0x00000800u-(137u<<23)))>>12;
	asm("v_med3_i32 %0, %1, %2, %3" : "=v"(mag) : "v"(rn), "v"(rd), "s"(2048));
	uint32_t code;   // bits 0..30 from mag<<16, bit 31 from x (spelled out because the compiler expands the or-of-ands to three instructions)
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(code) : "s"(0x7FFFFFFF), "v"((uint32_t)mag<<16), "v"(x));
	return code;
}
__device__ __forceinline__ uint32_t float_to_half_custom(const float x) { return float_to_half_custom_hi(x)>>16; }
// Encode of the 19 post-collision DDFs at the very end of a kernel, 3 instructions each.  Under round-TOWARD-ZERO the
// product v = |x| 2^-112 carries the whole reference formula in its bit pattern: for |x| >= 2^-14 it is exact and its
// exponent field is the rebiased FP16C exponent; below, v is a float DENORMAL whose mantissa field is floor(|x| 2^37), and
// the hardware's denormalisation is exactly the reference's variable shift.  (bits(v) + 0x800) >> 12 is then
// round_half_up onto the FP16C grid in both ranges (floor(floor(y)/4096 + 1/2) = floor(y/4096 + 1/2)), the carry runs into the
// exponent by itself, and a 4-bit shift to the left instead leaves the code in bits 16..30 for the d16_hi store; one
// bit-field insert adds the sign.  With the default round-to-nearest-even the denormal range would be rounded twice
// (ties at 2^-37 before the half-up at 2^-25), hence the mode switch: the FP32 rounding mode of THIS wave is set to RTZ
// by s_setreg and stays so -- the caller must have nothing but integer work and stores left.  All f[] pass through the
// two asm statements, so every floating-point instruction that produces them is ordered before the switch.
// Same codes as the literal formula for every finite float and +-Inf (device self-check, luw_selfcheck_fp16c_codec).
__device__ __forceinline__ uint32_t fp16c_code_hi_in_rtz_mode(const float x) { // the wave's FP32 rounding mode must be RTZ
	uint32_t v, c, code;
	asm volatile("v_mul_f32_e64 %0, |%1|, %2" : "=v"(v) : "v"(x), "s"(0x1p-112f));
	asm("v_add_lshl_u32 %0, %1, %2, 4" : "=v"(c) : "v"(v), "s"(0x800));
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(code) : "s"(0x7FFFFFFF), "v"(c), "v"(x));
	return code;
}
// two values at once: one packed multiplication (the sign needs no |.|: it is shifted out and re-inserted from x)
typedef float f32x2_codec __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fp16c_code2_hi_in_rtz_mode(const f32x2_codec x, uint32_t& c0, uint32_t& c1) {
	f32x2_codec v; const f32x2_codec k = { 0x1p-112f, 0x1p-112f };
	asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(x), "v"(k));
	uint32_t t0, t1;
	asm("v_add_lshl_u32 %0, %1, %2, 4" : "=v"(t0) : "v"(v.x), "s"(0x800));
	asm("v_add_lshl_u32 %0, %1, %2, 4" : "=v"(t1) : "v"(v.y), "s"(0x800));
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(c0) : "s"(0x7FFFFFFF), "v"(t0), "v"(x.x));
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(c1) : "s"(0x7FFFFFFF), "v"(t1), "v"(x.y));
}
// the same code from a value that is ALREADY scaled by 2^-112 (native-arithmetic pair kernel, collide_cell_pk_native<.., RAW>): the float's own bits carry the
// reference formula -- add 0x800, drop 12 bits; the sign bit leaves with the 4-bit shift and comes back through the bit-field insert.  Any rounding mode.
__device__ __forceinline__ uint32_t fp16c_code_hi_of_scaled(const float x) {
	uint32_t c, code;
	asm("v_add_lshl_u32 %0, %1, %2, 4" : "=v"(c) : "v"(x), "s"(0x800));
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(code) : "s"(0x7FFFFFFF), "v"(c), "v"(x));
	return code;
}
// g/cg: the 7 populations of the thermal lattice, encoded in the same region (nullptr without it)
__device__ __forceinline__ void fp16c_encode19_hi_rtz_final(float* f, uint32_t* code, float* g = nullptr, uint32_t* cg = nullptr) {
	asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]));
	if(g) asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]));
	asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3" : "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]), "+v"(f[16]),
		"+v"(f[17]), "+v"(f[18]));
	#pragma unroll
	for(int i=0; i<19; i++) code[i] = fp16c_code_hi_in_rtz_mode(f[i]);
	if(g) {
		#pragma unroll
		for(int i=0; i<7; i++) cg[i] = fp16c_code_hi_in_rtz_mode(g[i]);
	}
}
template<typename T> __device__ __forceinline__ float ddf_decode(const T v);
template<> __device__ __forceinline__ float ddf_decode<float>(const float v) { return v; }
template<> __device__ __forceinline__ float ddf_decode<uint16_t>(const uint16_t v) { return half_to_float_custom_sx((int32_t)(int16_t)v); }
template<typename T> __device__ __forceinline__ T ddf_encode(const float v);
template<> __device__ __forceinline__ float ddf_encode<float>(const float v) { return v; }
template<> __device__ __forceinline__ uint16_t ddf_encode<uint16_t>(const float v) { return (uint16_t)(float_to_half_custom_hi(v)>>16); }

__device__ __forceinline__ float sq(const float x) { return x*x; }
__device__ __forceinline__ float clampf(const float x, const float a, const float b) { return fminf(fmaxf(x, a), b); }

// ---------------------------------------------------------------- IEEE division and square root without the range handling
// `a/b` and `sqrtf(x)` compile to the correctly rounded sequences of the device library: for the division two v_div_scale (operands
// scaled out of the denormal / overflow ranges), v_rcp_f32, two Newton fmas, the quotient, two residual corrections, v_div_fmas
// (undoes the scaling) and v_div_fixup (zeros, infinities, NaNs): 11 instructions; for the square root a 2^32 pre-scaling of inputs
// below 2^-96, v_sqrt_f32, the two neighbours of its result, their residuals, two selects, the un-scaling and a class test: 15.
// Where the VALU is the limit (FP16C kernels) and the operands sit in the plain range, the SAME instruction sequences run without the
// scaling and the special-case tails -- bit-identical results by construction, since those parts are identities there:
//   division n/d, 2^-60 <= d < 2^60 (an "ordinary" density: density_is_ordinary, tested per lane; other lanes redo the quotient with `/`),
//     n = 0 or 2^-25 <= |n| < 64 (moment sums of FP16C populations: multiples of 2^-25 below 38; the constants 0.5 and 2): v_div_scale leaves
//     both operands alone (ISA: it scales for a denormal d, 1/d or n/d, for |n| < 2^-103, for exponents 96 or more apart) and VCC = 0, so
//     v_div_fmas is a plain fma; v_div_fixup only re-applies the sign, which matters for n = -0 alone (this sequence gives +0: the sign of a
//     zero, never a value).  The reciprocal refinement depends on d only: several numerators share it.
//   square root, x = 0 or x >= 2^-96: no pre-scaling, and the class test (zero / infinity pass through) is covered by v_sqrt_f32 itself:
//     for x = 0 both neighbour tests fail (NaN and +0 residuals) and 0 stays.
// Checked on the device against the library forms: luw_selfcheck_arith (every float of the square root's range; > 2^32 quotients).
#ifndef LUW_PLAIN_ARITH
#define LUW_PLAIN_ARITH 1   /* 0: the library's division / square root everywhere (A/B builds) */
#endif
struct Recip { float d, r; };
__device__ __forceinline__ Recip recip_prepare(const float d) {
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e = fmaf(-d, r0, 1.0f);
	return Recip{ d, fmaf(e, r0, r0) };
}
__device__ __forceinline__ float div_by(const float n, const Recip R) {
	float q = n*R.r;
	q = fmaf(fmaf(-R.d, q, n), R.r, q);
	return fmaf(fmaf(-R.d, q, n), R.r, q);
}
__device__ __forceinline__ float sqrt_in_range(const float x) {
	const float s = __builtin_amdgcn_sqrtf(x);
	const float down = __uint_as_float(__float_as_uint(s)-1u), up = __uint_as_float(__float_as_uint(s)+1u);
	const float r_down = fmaf(-down, s, x), r_up = fmaf(-up, s, x);
	const float t = r_down<=0.0f ? down : s;
	return r_up>0.0f ? up : t;
}
// the densities for which div_by is the library's division: finite, positive, exponent in [-60, 60)
__device__ __forceinline__ bool density_is_ordinary(const float rho) { return __float_as_uint(rho)-0x21800000u<0x3C000000u; }

// ---------------------------------------------------------------- direction algebra, FX/kernel.cpp:890-893
// c_I . (a, b, c) with the zero terms dropped, evaluated left to right like c(i)*a+c(19+i)*b+c(38+i)*c (FX/kernel.cpp:1111).  D3Q19 order: rest, the six
// axis directions (+x -x +y -y +z -z), then the diagonals (++0 --0, +0+ -0-, 0++ 0--, +-0 -+0, +0- -0+, 0+- 0-+): odd I and I + 1 are opposite.
template<int I> __device__ __forceinline__ float cdot(const float a, const float b, const float c) {
	if constexpr(I== 1) return  a; else if constexpr(I== 2) return -a;
	else if constexpr(I== 3) return  b; else if constexpr(I== 4) return -b;
	else if constexpr(I== 5) return  c; else if constexpr(I== 6) return -c;
	else if constexpr(I== 7) return  a+b; else if constexpr(I== 8) return -a-b;
	else if constexpr(I== 9) return  a+c; else if constexpr(I==10) return -a-c;
	else if constexpr(I==11) return  b+c; else if constexpr(I==12) return -b-c;
	else if constexpr(I==13) return  a-b; else if constexpr(I==14) return -a+b;
	else if constexpr(I==15) return  a-c; else if constexpr(I==16) return -a+c;
	else if constexpr(I==17) return  b-c; else return -b+c; // 18
}
template<int K, typename Fn> __device__ __forceinline__ void for_each_pair(Fn&& fn) { // fn(K), fn(K + 1), ..., fn(8) with K a compile-time constant
	fn(std::integral_constant<int, K>{});
	if constexpr(K<8) for_each_pair<K+1>(fn);
}

// ---------------------------------------------------------------- f_eq, FX/kernel.cpp:1016-1055
// Second-order equilibrium of the SHIFTED populations (stored value = f - w_i): per opposite pair k (directions 2k+1, 2k+2) with s = 3 c.u and
// q = s^2 - 3 u^2, f_eq = w_k rho (q / 2 +- s) + w_k (rho - 1).  The nesting of the fused multiply-adds is the reference's (three deep per population:
// fma(s, s, -3 u^2), fma(1/2, q, +-s), fma(w rho, ., w (rho - 1))), which is what makes the values bit-equal to its restatement; the pair loop and the
// two weight classes (k < 3: axis directions, 1/18; else diagonals, 1/36) are this file's.
struct EqCommon { float q0, lead[2], base[2], s3[3]; };            // -3 u^2; w rho and w (rho - 1) per weight class; 3 u
__device__ __forceinline__ EqCommon eq_common(const float rho, const float ux, const float uy, const float uz, float& feq_rest) {
	EqCommon e;
	const float dev = rho-1.0f;
	e.q0 = -3.0f*(sq(ux)+sq(uy)+sq(uz));
	e.s3[0] = ux*3.0f; e.s3[1] = uy*3.0f; e.s3[2] = uz*3.0f;
	feq_rest = DEF_W0*fmaf(rho, 0.5f*e.q0, dev);
	e.lead[0] = DEF_WS*rho; e.lead[1] = DEF_WE*rho; e.base[0] = DEF_WS*dev; e.base[1] = DEF_WE*dev;
	return e;
}
__device__ __forceinline__ void calculate_f_eq(const float rho, const float ux, const float uy, const float uz, float* feq) {
	const EqCommon e = eq_common(rho, ux, uy, uz, feq[0]);
	for_each_pair<0>([&](auto kc) {
		constexpr int k = decltype(kc)::value, cls = k<3 ? 0 : 1;
		const float s = cdot<2*k+1>(e.s3[0], e.s3[1], e.s3[2]);
		const float q = fmaf(s, s, e.q0);
		feq[2*k+1] = fmaf(e.lead[cls], fmaf(0.5f, q, s), e.base[cls]);
		feq[2*k+2] = fmaf(e.lead[cls], fmaf(0.5f, q, -s), e.base[cls]);
	});
}

// ---------------------------------------------------------------- moments, FX/kernel.cpp:1075-1100
// the sums: density and the momentum before the division by it
__device__ __forceinline__ void moment_sums(const float* f, float& rho, float& mx, float& my, float& mz) {
	float r = f[0];
	#pragma unroll
	for(int i=1; i<19; i++) r += f[i];
	rho = r+1.0f;
	mx = f[ 1]-f[ 2]+f[ 7]-f[ 8]+f[ 9]-f[10]+f[13]-f[14]+f[15]-f[16];
	my = f[ 3]-f[ 4]+f[ 7]-f[ 8]+f[11]-f[12]+f[14]-f[13]+f[17]-f[18];
	mz = f[ 5]-f[ 6]+f[ 9]-f[10]+f[11]-f[12]+f[16]-f[15]+f[18]-f[17];
}
__device__ __forceinline__ void calculate_rho_u(const float* f, float& rhon, float& uxn, float& uyn, float& uzn) {
	float rho, ux, uy, uz;
	moment_sums(f, rho, ux, uy, uz);
	rhon = rho;
	uxn = ux/rho;
	uyn = uy/rho;
	uzn = uz/rho;
}

template<int I> __device__ __forceinline__ void forcing_term(const float ux, const float uy, const float uz, const float fx, const float fy, const float fz,
	const float uF, float* Fin) {
	constexpr float w9 = 9.0f*(I<7 ? DEF_WS : DEF_WE);
	Fin[I] = w9*fmaf(cdot<I>(fx, fy, fz), cdot<I>(ux, uy, uz)+0.33333334f, uF);
	if constexpr(I<18) forcing_term<I+1>(ux, uy, uz, fx, fy, fz, uF, Fin);
}
// Guo forcing, FX/kernel.cpp:1103-1113
__device__ __forceinline__ void calculate_forcing_terms(const float ux, const float uy, const float uz, const float fx, const float fy, const float fz,
	float* Fin) {
	const float uF = -0.33333334f*fmaf(ux, fx, fmaf(uy, fy, uz*fz));
	Fin[0] = 9.0f*DEF_W0*uF;
	forcing_term<1>(ux, uy, uz, fx, fy, fz, uF, Fin);
}

// ---------------------------------------------------------------- LUW force assembly, FX/kernel.cpp:1516-1623
// Adds Coriolis, buffer nudging, top sponge and the per-cell force to (fxn,fyn,fzn).  n is the device index of
// the cell, (x,y,z) its local coordinates.  u is the device velocity field (3 planes of stride Np).
// ZONES = false: the caller knows that the cell lies outside the nudging / sponge zones and that there is no force field -- what is
// left is the volume force and Coriolis (the uniform part)
// The nudging / sponge references of one cell, fetched AHEAD of its collision (pair kernel, general instantiation): what assemble_force would read
// through n_ref once the moments are known -- the target velocity of the nearest owned face with its weight, the top layer's velocity with the
// sponge's sigma.  They depend on the position alone, so the loads go out with the DDF loads and their latency (four dependent trips to memory per
// lane pair otherwise, ~3400 of a wave's 21000 cycles on the urban tile: profiles/r03_stall_counters.md) is hidden behind decode and moments.
// A TYPE_E cell takes neither term; its four registers carry what it reads instead, its own rho and u (FX/kernel.cpp:1516-1523), fetched just as early.
struct ForceRefs { float tu[3], wb, su[3], sg; bool zn, zs; };
__device__ __forceinline__ void fetch_force_refs(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y, const uint32_t z, const bool live,
	const bool is_E,
		const float* __restrict__ rho, const float* __restrict__ u, ForceRefs& r) {
	// (values of lanes outside the zones are never read: defined as "whatever the register holds", at no instruction)
	asm volatile("" : "=v"(r.tu[0]), "=v"(r.tu[1]), "=v"(r.tu[2]), "=v"(r.wb), "=v"(r.su[0]), "=v"(r.su[1]), "=v"(r.su[2]), "=v"(r.sg));
	const bool own = live && is_E;                                  // reads its own fields
	const bool act = live && !is_E;                                 // the zone terms can act (a cell that is not collided: nothing)
	bool zn = false;
	uint32_t d_min = 0u, n_ref = n;
	if(p.buffer_active) { // the selection of assemble_force
		const uint32_t d_w = x-(uint32_t)p.west_x, d_e = (uint32_t)p.east_x-x, d_s = y-(uint32_t)p.south_y, d_n = (uint32_t)p.north_y-y,
			d_t = (uint32_t)p.top_z-z;
		const bool in_w = x-p.zw_lo<p.zw_n, in_e = x-p.ze_lo<p.ze_n, in_s = y-p.zs_lo<p.zs_n, in_n = y-p.zn_lo<p.zn_n, in_t = z-p.zt_lo<p.zt_n;
		zn = act && (in_w||in_e||in_s||in_n||in_t);
		if(zn) {
			// assemble_force walks w, e, s, n, t and keeps the first nearest face.  The three that depend on (y, z) alone -- the same for every lane
			// of a row, scalar work in the pair kernel -- are settled among themselves first (same order, same strict <); w and e, which come before
			// them, then only lose to a strictly nearer one: the same winner.
			uint32_t d_row = p.buffer_N+1u, base_row = 0u;
			if(in_s) { if(d_s<d_row) { d_row = d_s; base_row = ((uint32_t)p.south_y+z*p.Ny)*p.Px; } }
			if(in_n) { if(d_n<d_row) { d_row = d_n; base_row = ((uint32_t)p.north_y+z*p.Ny)*p.Px; } }
			if(in_t) { if(d_t<d_row) { d_row = d_t; base_row = (y+(uint32_t)p.top_z*p.Ny)*p.Px; } }
			d_min = p.buffer_N+1u;
			const uint32_t rowyz = (y+z*p.Ny)*p.Px;
			if(in_w) { if(d_w<d_min) { d_min = d_w; n_ref = (uint32_t)p.west_x+rowyz; } }
			if(in_e) { if(d_e<d_min) { d_min = d_e; n_ref = (uint32_t)p.east_x+rowyz; } }
			if(d_row<d_min) { d_min = d_row; n_ref = x+base_row; }
		}
	}
	r.zn = zn;
	// ONE run of four loads for both kinds of lane (two runs in two branches made the second wait for every load in flight before it could reuse
	// the first one's address registers): weight and target velocity of a zone lane, own density and velocity of a TYPE_E lane (n_ref = n there)
	if(zn||own) {
		const float* const first = own ? rho+n : p.wbuf+d_min;
		r.wb = *first;
		r.tu[0] = u[n_ref]; r.tu[1] = u[(size_t)p.Np+n_ref]; r.tu[2] = u[2ull*p.Np+n_ref];
	}
	r.zs = act && z-p.zp_lo<p.zp_n;
	if(r.zs) {
		r.sg = p.sigma[(uint32_t)p.top_z-1u-z];
		const uint32_t n_top = x+(y+(uint32_t)p.top_z*p.Ny)*p.Px;
		r.su[0] = u[n_top]; r.su[1] = u[(size_t)p.Np+n_top]; r.su[2] = u[2ull*p.Np+n_top];
	}
}
// refs (pair kernel, general instantiation only): the references above, already fetched -- same arithmetic on them
template<bool ZONES=true> __device__ __forceinline__ void assemble_force(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y,
	const uint32_t z, const bool is_E,
		const float rhon, const float uxn, const float uyn, const float uzn, const float* __restrict__ u, const float* __restrict__ F, float& fxn, float& fyn,
			float& fzn,
		const ForceRefs* refs = nullptr) {
	fxn = p.fx; fyn = p.fy; fzn = p.fz;
	if(p.coriolis) { // with omega = 0 the three terms are +-0: adding them changes no value
		const float cor_x = -2.0f*rhon*(p.omy*uzn-p.omz*uyn);
		const float cor_y = -2.0f*rhon*(p.omz*uxn-p.omx*uzn);
		const float cor_z = -2.0f*rhon*(p.omx*uyn-p.omy*uxn);
		fxn += cor_x;
		fyn += cor_y;
		fzn += cor_z;
	}
	if constexpr(!ZONES) return;
	if(refs) {
		if(refs->zn) {
			const float a_x = refs->wb*p.buffer_inv_tau*(refs->tu[0]-uxn);
			const float a_y = refs->wb*p.buffer_inv_tau*(refs->tu[1]-uyn);
			const float a_z = p.nudge_vertical==1u ? refs->wb*p.buffer_inv_tau*(refs->tu[2]-uzn) : 0.0f;
			fxn += rhon*a_x;
			fyn += rhon*a_y;
			fzn += rhon*a_z;
		}
		if(refs->zs) {
			fxn += rhon*refs->sg*(refs->su[0]-uxn);
			fyn += rhon*refs->sg*(refs->su[1]-uyn);
			fzn += rhon*refs->sg*(refs->su[2]-uzn);
		}
		if(p.has_F) {
			fxn += F[n];
			fyn += F[(size_t)p.Np+n];
			fzn += F[2ull*p.Np+n];
		}
		return;
	}
	if(p.buffer_active && !is_E) {
		// distance to each face (meaningful inside that face's zone, wrapped-around garbage outside, where it is not used)
		const uint32_t d_w = x-(uint32_t)p.west_x, d_e = (uint32_t)p.east_x-x, d_s = y-(uint32_t)p.south_y, d_n = (uint32_t)p.north_y-y,
			d_t = (uint32_t)p.top_z-z;
		const bool in_w = x-p.zw_lo<p.zw_n, in_e = x-p.ze_lo<p.ze_n, in_s = y-p.zs_lo<p.zs_n, in_n = y-p.zn_lo<p.zn_n, in_t = z-p.zt_lo<p.zt_n;
		if(in_w||in_e||in_s||in_n||in_t) {
			uint32_t d_min = p.buffer_N+1u;
			uint32_t n_ref = n;
			const uint32_t rowyz = (y+z*p.Ny)*p.Px;
			if(in_w) { if(d_w<d_min) { d_min = d_w; n_ref = (uint32_t)p.west_x+rowyz; } }
			if(in_e) { if(d_e<d_min) { d_min = d_e; n_ref = (uint32_t)p.east_x+rowyz; } }
			if(in_s) { if(d_s<d_min) { d_min = d_s; n_ref = x+((uint32_t)p.south_y+z*p.Ny)*p.Px; } }
			if(in_n) { if(d_n<d_min) { d_min = d_n; n_ref = x+((uint32_t)p.north_y+z*p.Ny)*p.Px; } }
			if(in_t) { if(d_t<d_min) { d_min = d_t; n_ref = x+(y+(uint32_t)p.top_z*p.Ny)*p.Px; } }
			const float w_buf = p.wbuf[d_min];
			const float u_target_x = u[n_ref];
			const float u_target_y = u[(size_t)p.Np+n_ref];
			const float u_target_z = u[2ull*p.Np+n_ref];
			const float a_x = w_buf*p.buffer_inv_tau*(u_target_x-uxn);
			const float a_y = w_buf*p.buffer_inv_tau*(u_target_y-uyn);
			const float a_z = p.nudge_vertical==1u ? w_buf*p.buffer_inv_tau*(u_target_z-uzn) : 0.0f;
			fxn += rhon*a_x;
			fyn += rhon*a_y;
			fzn += rhon*a_z;
		}
	}
	if(!is_E && z-p.zp_lo<p.zp_n) { // top sponge (zp_n = 0 unless the term is on and this domain owns the top)
		const float sigma = p.sigma[(uint32_t)p.top_z-1u-z];   // layer index (Nzg-2) - (z+Oz), FX/kernel.cpp:1600
		const uint32_t n_ref = x+(y+(uint32_t)p.top_z*p.Ny)*p.Px;
		fxn += rhon*sigma*(u[n_ref]-uxn);
		fyn += rhon*sigma*(u[(size_t)p.Np+n_ref]-uyn);
		fzn += rhon*sigma*(u[2ull*p.Np+n_ref]-uzn);
	}
	if(p.has_F) {
		fxn += F[n];
		fyn += F[(size_t)p.Np+n];
		fzn += F[2ull*p.Np+n];
	}
}

// ---------------------------------------------------------------- collision of one cell, FX/kernel.cpp:1502-1515,1686-1748
// Smagorinsky-Lilly relaxation rate, FX/kernel.cpp:1723-1737; sums run over i=1..18 in order, zero terms dropped (the
// leading 0 + n of each sum too: it can only turn -0 into +0, and every sum is squared)
__device__ __forceinline__ float smagorinsky_Q(const float* n_) {
	float Hxx = n_[ 1], Hyy = n_[ 3], Hzz = n_[ 5], Hxy = n_[ 7], Hxz = n_[ 9], Hyz = n_[11];
	Hxx += n_[ 2]; Hxx += n_[ 7]; Hxx += n_[ 8]; Hxx += n_[ 9]; Hxx += n_[10]; Hxx += n_[13]; Hxx += n_[14]; Hxx += n_[15]; Hxx += n_[16];
	Hyy += n_[ 4]; Hyy += n_[ 7]; Hyy += n_[ 8]; Hyy += n_[11]; Hyy += n_[12]; Hyy += n_[13]; Hyy += n_[14]; Hyy += n_[17]; Hyy += n_[18];
	Hzz += n_[ 6]; Hzz += n_[ 9]; Hzz += n_[10]; Hzz += n_[11]; Hzz += n_[12]; Hzz += n_[15]; Hzz += n_[16]; Hzz += n_[17]; Hzz += n_[18];
	Hxy += n_[ 8]; Hxy += -n_[13]; Hxy += -n_[14];
	Hxz += n_[10]; Hxz += -n_[15]; Hxz += -n_[16];
	Hyz += n_[12]; Hyz += -n_[17]; Hyz += -n_[18];
	return sq(Hxx)+sq(Hyy)+sq(Hzz)+2.0f*(sq(Hxy)+sq(Hxz)+sq(Hyz));
}
__device__ __forceinline__ float smagorinsky_rate_of_Q(const KParams& p, const float rhon, const float Q) {
	return 2.0f/(p.tau0+sqrtf(p.tau0sq+0.76421222f*sqrtf(Q)/rhon));
}
// the same value for an ordinary density (R = recip_prepare(rhon)): FP16C populations keep Q below ~1e4 and tau in [1, 20], plain operands for
// sqrt_in_range and div_by; a Q under 2^-96, where sqrt_in_range may be an ulp off, and a quotient s/rho so small that div_by's unscaled residuals
// lose bits, both add less than a quarter ulp to tau0sq >= 1/4 and change nothing
__device__ __forceinline__ float smagorinsky_rate_plain(const KParams& p, const float Q, const Recip R) {
	const float s = 0.76421222f*sqrt_in_range(Q);
	const float tau = p.tau0+sqrt_in_range(p.tau0sq+div_by(s, R));
	return div_by(2.0f, recip_prepare(tau));
}
__device__ __forceinline__ float smagorinsky_rate(const KParams& p, const float rhon, const float* n_) {
	return smagorinsky_rate_of_Q(p, rhon, smagorinsky_Q(n_));
}
// PLAIN (FP16C storage, LUW_PLAIN_ARITH): divisions by the density and the square roots as plain-range sequences (recip_prepare); R is the prepared
// reciprocal of rhon, odd_density marks the lanes that redo those results with the library forms
struct DensityRecip { Recip R; bool odd; };
template<bool PLAIN> __device__ __forceinline__ float relaxation_rate(const KParams& p, const float rhon, const float* f, const float* feq,
	const DensityRecip& dr) {
	if(!p.subgrid) return p.w;
	float n_[19];
	#pragma unroll
	for(int i=1; i<19; i++) n_[i] = f[i]-feq[i];
	const float Q = smagorinsky_Q(n_);
	if constexpr(PLAIN) {
		float w = smagorinsky_rate_plain(p, Q, dr.R);
		if(dr.odd) w = smagorinsky_rate_of_Q(p, rhon, Q);
		return w;
	} else return smagorinsky_rate_of_Q(p, rhon, Q);
}
// rho, u of the cell (moments, or the stored values on TYPE_E cells) and the force acting on it
template<bool NOFORCE=false, bool PLAIN=false> __device__ __forceinline__ void collide_head(const KParams& p, const uint32_t n, const uint32_t x,
	const uint32_t y, const uint32_t z, const bool is_E, const float* f,
		const float* __restrict__ rho, const float* __restrict__ u, const float* __restrict__ F, float& rhon, float& uxn, float& uyn, float& uzn, float& fxn,
			float& fyn, float& fzn, DensityRecip& dr) {
	if(is_E) {
		rhon = rho[n];
		uxn = u[n];
		uyn = u[(size_t)p.Np+n];
		uzn = u[2ull*p.Np+n];
		if constexpr(PLAIN) { dr.R = recip_prepare(rhon); dr.odd = !density_is_ordinary(rhon); }
	} else if constexpr(PLAIN) {
		float mx, my, mz;
		moment_sums(f, rhon, mx, my, mz);
		dr.R = recip_prepare(rhon); dr.odd = !density_is_ordinary(rhon);
		uxn = div_by(mx, dr.R); uyn = div_by(my, dr.R); uzn = div_by(mz, dr.R);
		if(dr.odd) { uxn = mx/rhon; uyn = my/rhon; uzn = mz/rhon; }
	} else {
		calculate_rho_u(f, rhon, uxn, uyn, uzn);
	}
	if constexpr(NOFORCE) fxn = fyn = fzn = 0.0f;   // the caller knows that nothing can push the cells of its launch box (luw_core.hip, box_force_mode)
	else assemble_force(p, n, x, y, z, is_E, rhon, uxn, uyn, uzn, u, F, fxn, fyn, fzn);
}
// the general tail: Guo forcing, equilibrium override on TYPE_E cells
template<bool PLAIN=false> __device__ __forceinline__ void collide_tail_general(const KParams& p, const bool is_E, const bool forced, const float fxn,
	const float fyn, const float fzn,
		float* f, const float rhon, float& uxn, float& uyn, float& uzn, const DensityRecip& dr) {
	float feq[19], Fin[19];
	if(forced) {
		float rho2;
		if constexpr(PLAIN) {
			rho2 = div_by(0.5f, dr.R);
			if(dr.odd) { asm volatile(""); rho2 = 0.5f/rhon; } // (the empty asm keeps this a branch, see collide_cell_pk)
		} else rho2 = 0.5f/rhon;
		uxn = clampf(fmaf(fxn, rho2, uxn), -DEF_C, DEF_C);
		uyn = clampf(fmaf(fyn, rho2, uyn), -DEF_C, DEF_C);
		uzn = clampf(fmaf(fzn, rho2, uzn), -DEF_C, DEF_C);
		calculate_forcing_terms(uxn, uyn, uzn, fxn, fyn, fzn, Fin);
	} else {
		uxn = clampf(uxn, -DEF_C, DEF_C);
		uyn = clampf(uyn, -DEF_C, DEF_C);
		uzn = clampf(uzn, -DEF_C, DEF_C);
		#pragma unroll
		for(int i=0; i<19; i++) Fin[i] = 0.0f;
	}
	calculate_f_eq(rhon, uxn, uyn, uzn, feq);
	const float w = relaxation_rate<PLAIN>(p, rhon, f, feq, dr);
	const float c_tau = fmaf(w, -0.5f, 1.0f);
	const float omw = 1.0f-w;
	#pragma unroll
	for(int i=0; i<19; i++) {
		const float Fi = Fin[i]*c_tau;
		f[i] = is_E ? feq[i] : fmaf(omw, f[i], fmaf(w, feq[i], Fi));
	}
}
// in: streamed-in DDFs f[19], flags byte.  out: post-collision DDFs in f[19]; rho/u (after the half-force
// shift and the +-c clamp) in rhon,uxn,uyn,uzn.
// NOFORCE: no force can act on any cell of the launch (box_force_mode): without the force assembly the FP16C kernel needs 69 instead of 89
// VGPRs (76 instead of 93 with the thermal lattice) and no scalar spills -- 7 resp. 6 waves per SIMD instead of 5
// PLAIN: the populations come out of FP16C storage (bounded operands): plain-range divisions and square roots (recip_prepare)
template<bool FAST=true, bool NOFORCE=false, bool PLAIN=false> __device__ __forceinline__ void collide_cell(const KParams& p, const uint32_t n,
	const uint32_t x, const uint32_t y, const uint32_t z, const uint8_t flagsn,
		float* f, const float* __restrict__ rho, const float* __restrict__ u, const float* __restrict__ F, float& rhon, float& uxn, float& uyn, float& uzn,
			float* u_before_force = nullptr) {
	const bool is_E = (flagsn&TYPE_BO)==TYPE_E;
	float fxn, fyn, fzn;
	DensityRecip dr{};
	collide_head<NOFORCE, PLAIN>(p, n, x, y, z, is_E, f, rho, u, F, rhon, uxn, uyn, uzn, fxn, fyn, fzn, dr);
	// what the thermal lattice advects with (FX/kernel.cpp:1669)
	if(u_before_force) { u_before_force[0] = uxn; u_before_force[1] = uyn; u_before_force[2] = uzn; }
	// A cell without any force (the bulk of an urban case: no volume force, outside the nudging / sponge zones, no
	// Coriolis) has Fin_i = +-0 exactly and u += 0/(2 rho); skipping that arithmetic is value-identical
	// (fma(w, feq, +-0) == w*feq) as long as rho != 0.
	const bool forced = fxn!=0.0f||fyn!=0.0f||fzn!=0.0f;
	// Wave-uniform fast path: when no lane of the wave is a TYPE_E cell or feels a force (interior waves of an urban case),
	// the equilibrium override and the Guo terms drop out for the whole wave -- one scalar branch instead of 19 selects, 19
	// products with zero and the zero-filled Fin registers per lane.  Same values (+-0 aside) as the general path below.
	if(FAST&&__ballot(is_E||forced)==0ull) {
		float feq[19];
		uxn = clampf(uxn, -DEF_C, DEF_C);
		uyn = clampf(uyn, -DEF_C, DEF_C);
		uzn = clampf(uzn, -DEF_C, DEF_C);
		calculate_f_eq(rhon, uxn, uyn, uzn, feq);
		const float w = relaxation_rate<PLAIN>(p, rhon, f, feq, dr);
		const float omw = 1.0f-w;
		#pragma unroll
		for(int i=0; i<19; i++) f[i] = fmaf(omw, f[i], w*feq[i]);
		return;
	}
	collide_tail_general<PLAIN>(p, is_E, forced, fxn, fyn, fzn, f, rhon, uxn, uyn, uzn, dr);
}

// ---------------------------------------------------------------- the same collision on PACKED pairs
// The 18 moving populations as nine pairs (f[2k+1], f[2k+2]) of opposite directions in 64-bit register pairs: the fast
// path's equilibria, non-equilibrium parts and relaxation are v_pk_fma/mul/add_f32 on those pairs (two IEEE operations per
// instruction, same roundings as the scalar code: value-identical).  Sums whose order is fixed (moments, stress tensor)
// read the halves.  Used where the VALU is the limit (FP16C pair kernel).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(const float a) { f32x2 r = { a, a }; return r; }
__device__ __forceinline__ f32x2 pm2(const float a) { f32x2 r = { a, -a }; return r; }
__device__ __forceinline__ void calculate_f_eq_pk(const float rho, const float ux, const float uy, const float uz, float& feq0, f32x2* feqp) {
	const EqCommon e = eq_common(rho, ux, uy, uz, feq0);
	for_each_pair<0>([&](auto kc) { // both populations of pair k in the two halves of packed instructions: (q / 2 + s, q / 2 - s), then the weights
		constexpr int k = decltype(kc)::value, cls = k<3 ? 0 : 1;
		const float s = cdot<2*k+1>(e.s3[0], e.s3[1], e.s3[2]);
		const f32x2 inner = __builtin_elementwise_fma(splat2(0.5f), splat2(fmaf(s, s, e.q0)), pm2(s));
		feqp[k] = __builtin_elementwise_fma(splat2(e.lead[cls]), inner, splat2(e.base[cls]));
	});
}
// Guo terms of the pair (2k+1, 2k+2): c_(2k+2) = -c_(2k+1), so both are w9 fma(+-cF, +-cu + 1/3, uF) (FX/kernel.cpp:1103-1113)
template<int K> __device__ __forceinline__ f32x2 forcing_pair(const float ux, const float uy, const float uz, const float fx, const float fy, const float fz,
	const float uF) {
	constexpr int I = 2*K+1;
	constexpr float w9 = 9.0f*(I<7 ? DEF_WS : DEF_WE);
	// c_(2k+2) . v = -(c_(2k+1) . v) exactly (-a-b and -(a+b) round alike), so the second lane takes the first lane's sums through the packed
	// instructions' negate modifiers instead of two more additions each
	const float cF = cdot<I>(fx, fy, fz), cu = cdot<I>(ux, uy, uz);
	const f32x2 a = { cF, -cF }, b = { cu, -cu };
	return splat2(w9)*__builtin_elementwise_fma(a, b+splat2(0.33333334f), splat2(uF));
}
// All cases in one: wave-uniform switches for "some lane may feel a force" and "some lane is a TYPE_E cell" select the
// extra work; a lane for which the switch is on without need computes with F = 0 (u + 0/(2 rho) = u, Fin = +-0:
// value-identical, +-0 aside, to the scalar code's per-lane shortcut).
// FORCE says what the caller knows about the launch box (luw_core.hip, pair_force_mode):
//   PAIR_FORCE_NONE     no force can act on any of its cells (no Coriolis, volume force or force field, box outside the nudging / sponge
//                       zones): the force assembly and the Guo terms are compiled out, which is what lets the kernel fit 5 waves per SIMD
//                       (86 instead of 109 VGPRs, no scalar spills).  TYPE_E cells then take no selects either: the caller decodes their
//                       populations as f = 0 and this routine relaxes them with w = 1, so that fma(1 - w, f, w f_eq) = fma(0, 0, f_eq) =
//                       f_eq bit for bit (f_eq is never -0: an exact cancellation gives +0, and at rho = 1, u = 0 every term is +0),
//                       whatever the Smagorinsky rate of such a lane came out as -- instead of keeping the nineteen equilibria for a
//                       select behind the relaxation;
//   PAIR_FORCE_UNIFORM  volume force and / or Coriolis act on every cell, nothing position-dependent does: no zone tests, no wave-uniform
//                       switch, no scalar spills; TYPE_E lanes as above, with the Guo term's factor c_tau = 0 on top (feq + Fi 0 = feq):
//                       96 VGPRs, 5 waves per SIMD as well;
//   PAIR_FORCE_ANY      everything, switched per wave.
enum { PAIR_FORCE_NONE = 0, PAIR_FORCE_UNIFORM = 1, PAIR_FORCE_ANY = 2 };

// ORDINARY densities (density_is_ordinary: 2^-60 <= rho < 2^60 -- anything a lattice that has not blown up holds) take the divisions and square
// roots as the library's instruction sequences minus their range handling (recip_prepare), the five divisions by the density sharing one
// reciprocal: 808 instead of 876 VALU instructions per lane in the force-free kernel, 992 instead of 1074 with uniform forces.  Lanes with any
// other density (zero, negative, NaN, absurd) redo exactly those results with the library forms inside rarely taken divergent blocks, so the
// values are the IEEE ones for EVERY input.  (A per-wave vote between two complete collisions was measured first: the duplicated code cost
// more than the arithmetic saved, 1024x1024x256 + Coriolis 4.05 -> 4.25 ms; this form 4.05 -> 3.89, profiles/r03_plain_arith_ab.txt.)
template<int FORCE=PAIR_FORCE_ANY> __device__ __forceinline__ void collide_cell_pk(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y,
	const uint32_t z, const uint8_t flagsn, const bool may_force,
		float& f0, f32x2* fp, const float* __restrict__ rho, const float* __restrict__ u, const float* __restrict__ F, float& rhon, float& uxn, float& uyn,
			float& uzn, float* u_before_force = nullptr,
		// refs: zone references fetched early (fetch_force_refs); own: a TYPE_E cell's rho / u fetched early (wb, tu)
		const ForceRefs* refs = nullptr, const ForceRefs* own = nullptr) {
	constexpr bool PLAIN = LUW_PLAIN_ARITH!=0;
	const bool is_E = (flagsn&TYPE_BO)==TYPE_E;
	const bool wave_has_E = __ballot(is_E)!=0ull;
	float rho_m, mx, my, mz;
	{
		float f[19];
		f[0] = f0;
		#pragma unroll
		for(int k=0; k<9; k++) { f[2*k+1] = fp[k].x; f[2*k+2] = fp[k].y; }
		moment_sums(f, rho_m, mx, my, mz);
	}
	rhon = rho_m;
	[[maybe_unused]] Recip R{};   // of the density the divisions below use: the moment sum, or (TYPE_E lanes) the stored field
	[[maybe_unused]] bool odd_density = false;
	if constexpr(PLAIN) {
		R = recip_prepare(rho_m); uxn = div_by(mx, R); uyn = div_by(my, R); uzn = div_by(mz, R);
		odd_density = !density_is_ordinary(rho_m);
		if(odd_density) { uxn = mx/rho_m; uyn = my/rho_m; uzn = mz/rho_m; }
	} else { uxn = mx/rho_m; uyn = my/rho_m; uzn = mz/rho_m; }
	if(wave_has_E) {
		if(is_E) {
			if(own) { rhon = own->wb; uxn = own->tu[0]; uyn = own->tu[1]; uzn = own->tu[2]; } // fetched ahead of the decode
			else {
				rhon = rho[n];
				uxn = u[n];
				uyn = u[(size_t)p.Np+n];
				uzn = u[2ull*p.Np+n];
			}
		}
		// rhon: the field value on TYPE_E lanes, the moment sum elsewhere
		if constexpr(PLAIN) { R = recip_prepare(rhon); odd_density = !density_is_ordinary(rhon); }
	}
	// what the thermal lattice advects with (FX/kernel.cpp:1669)
	if(u_before_force) { u_before_force[0] = uxn; u_before_force[1] = uyn; u_before_force[2] = uzn; }
	const bool forced = FORCE==PAIR_FORCE_UNIFORM || (FORCE==PAIR_FORCE_ANY && may_force);
	f32x2 Finp[9]; float Fin0 = 0.0f;
	if(forced) {
		float fxn, fyn, fzn;
		assemble_force<(FORCE==PAIR_FORCE_ANY)>(p, n, x, y, z, is_E, rhon, uxn, uyn, uzn, u, F, fxn, fyn, fzn, refs);
		float rho2;
		if constexpr(PLAIN) {
			rho2 = div_by(0.5f, R);
			// (the empty asm keeps this a branch: a lone division would be hoisted in front of a select and run for every lane)
			if(odd_density) { asm volatile(""); rho2 = 0.5f/rhon; }
		} else rho2 = 0.5f/rhon;
		uxn = clampf(fmaf(fxn, rho2, uxn), -DEF_C, DEF_C);
		uyn = clampf(fmaf(fyn, rho2, uyn), -DEF_C, DEF_C);
		uzn = clampf(fmaf(fzn, rho2, uzn), -DEF_C, DEF_C);
		const float uF = -0.33333334f*fmaf(uxn, fxn, fmaf(uyn, fyn, uzn*fzn));
		Fin0 = 9.0f*DEF_W0*uF;
		Finp[0] = forcing_pair<0>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[1] = forcing_pair<1>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[2] = forcing_pair<2>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[3] = forcing_pair<3>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[4] = forcing_pair<4>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[5] = forcing_pair<5>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[6] = forcing_pair<6>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[7] = forcing_pair<7>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
		Finp[8] = forcing_pair<8>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
	} else {
		uxn = clampf(uxn, -DEF_C, DEF_C);
		uyn = clampf(uyn, -DEF_C, DEF_C);
		uzn = clampf(uzn, -DEF_C, DEF_C);
	}
	float feq0; f32x2 feqp[9];
	calculate_f_eq_pk(rhon, uxn, uyn, uzn, feq0, feqp);
	float w = p.w;
	if(p.subgrid) {
		float n_[19];
		#pragma unroll
		for(int k=0; k<9; k++) { const f32x2 d = fp[k]-feqp[k]; n_[2*k+1] = d.x; n_[2*k+2] = d.y; }
		const float Q = smagorinsky_Q(n_);
		if constexpr(PLAIN) { w = smagorinsky_rate_plain(p, Q, R); if(odd_density) w = smagorinsky_rate_of_Q(p, rhon, Q); }
		else w = smagorinsky_rate_of_Q(p, rhon, Q);
	}
	constexpr bool E_BY_RATE = FORCE!=PAIR_FORCE_ANY;   // TYPE_E lanes through the relaxation rate (f = 0, w = 1, no Guo term) instead of nineteen selects
	if constexpr(E_BY_RATE) { if(wave_has_E) w = is_E ? 1.0f : w; }
	const float omw = 1.0f-w;
	float r0; f32x2 rp[9];
	if(forced) {
		float c_tau = fmaf(w, -0.5f, 1.0f);
		if constexpr(E_BY_RATE) { if(wave_has_E) c_tau = is_E ? 0.0f : c_tau; }
		r0 = fmaf(omw, f0, fmaf(w, feq0, Fin0*c_tau));
		#pragma unroll
		for(int k=0; k<9; k++) rp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], __builtin_elementwise_fma(splat2(w), feqp[k], Finp[k]*splat2(c_tau)));
	} else {
		r0 = fmaf(omw, f0, w*feq0);
		#pragma unroll
		for(int k=0; k<9; k++) rp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], splat2(w)*feqp[k]);
	}
	if constexpr(!E_BY_RATE) {
		if(wave_has_E) {
			r0 = is_E ? feq0 : r0;
			#pragma unroll
			for(int k=0; k<9; k++) { rp[k].x = is_E ? feqp[k].x : rp[k].x; rp[k].y = is_E ? feqp[k].y : rp[k].y; }
		}
	}
	f0 = r0;
	#pragma unroll
	for(int k=0; k<9; k++) fp[k] = rp[k];
}
// ---------------------------------------------------------------- the same collision in NATIVE arithmetic (LUW_OPT_NATIVE_ARITH)
// The contract above (every operation rounded like the CPU restatement's) is this project's, not the reference's: the reference kernel is compiled by the
// OpenCL driver with -cl-mad-enable and native division / square root (FX/opencl.hpp:305, FX/kernel.cpp:1088-1100,1735) and is not bit-defined.  Where the
// VALU is the limit (FP16C pair kernel) the same formulas can run with the hardware's own operations and the sums in any order:
//   * one v_rcp_f32 of the density serves u = m / rho, F / (2 rho) and sqrt(Q) / rho; v_sqrt_f32 and v_rcp_f32 for the Smagorinsky rate
//     (w = 1 / (tau0 / 2 + sqrt(tau0^2 + 0.76421222 sqrt(Q) / rho) / 2));
//   * moments from the nine pair sums s_k = f[2k+1] + f[2k+2] and differences d_k = f[2k+1] - f[2k+2] (c_(2k+2) = -c_(2k+1)): rho = f0 + sum s_k + 1,
//     mx = d0 + d3 + d4 + d6 + d7, ...; 40 additions in short trees instead of 46 in chains;
//   * the stress tensor from the NON-EQUILIBRIUM PAIR SUMS alone: c c is the same for both directions of a pair, so Pi = sum_k (c c)_k (n_(2k+1) + n_(2k+2))
//     and n_(2k+1) + n_(2k+2) = s_k - (rho w_k A_k + 2 (rho - 1) w_k) with A_k = (3 c.u)^2 - 3 u^2 of the equilibrium (its +-3 c.u parts cancel):
//     18 + 15 operations instead of 18 + 36, and the nineteen equilibria are formed only once the rate is known (no f_eq registers across Q);
//   * Guo terms with the constants folded: c_tau w9_i [(c_i.F)(c_i.u + 1/3) - u.F / 3] = fma(c_i.H, 3 c_i.u + 1, uH) with H = c_tau w9_i F / 3, the
//     3 c.u of the equilibrium reused, added inside the relaxation's own fma;
//   * fused multiply-adds wherever a product feeds a sum (written out: see below), v_med3 for the +-c clamp, -2 omega from the host.
// TYPE_E lanes in every FORCE mode: decoded as f = 0 by the caller, relaxed with w = 1 and c_tau = 0 -> f_eq (collide_cell_pk, E_BY_RATE).
// Values differ from the exact kernels' in the last bits of each operation; with FP16C storage (2^-12 relative per stored value) those differences
// surface as different roundings of single populations, exactly like the reference's own arithmetic against the restatement's.
// The exact kernels stay the default and the anchor of every bit-for-bit test; tests/test_gpu_native_arith.py holds the tolerance gates of this one.
#ifndef LUW_NATIVE_RCP_NEWTON
#define LUW_NATIVE_RCP_NEWTON 0   /* 1: one Newton step behind the density's v_rcp_f32 (A/B: the u-RMSE against the oracle does not change) */
#endif
// RAW (pair kernel without the thermal lattice): the populations arrive and leave SCALED by 2^-112 -- the bit pattern the codec's shift-and-mask produces
// and consumes -- so that neither the decode nor the encode multiplies: every place the populations enter is linear in them, and the power of two moves
// into a factor that exists anyway (rho = fma(sum, 2^112, 1); u = m (2^112 / rho); n_k = fma(s_k, 2^112, -eq_k); out = (1 - w) f + 2^-112 (w f_eq + F)).
// A power of two commutes with every rounding as long as nothing underflows: the scaled populations are multiples of 2^-137 (the float denormal
// quantum is 2^-149), their sums round like the unscaled ones; the outputs are rounded to 2^-149 = 2^-37 in lattice units where an FP16C code step is
// 2^-25 at least.  The encode is then the reference's own formula on the float's bits (add 0x800, drop 12 bits: FX/kernel.cpp:870-875) under the default
// rounding mode -- no switch to round-toward-zero, which the exact kernels need for their 2^-112 product alone.
__device__ __forceinline__ f32x2 sum_and_negated_difference(const f32x2 a) { // { x + y, y - x } in one packed addition
	f32x2 r;
	asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
	return r;
}
// Every fused multiply-add is written out (the translation unit is compiled with -ffp-contract=off): the operation sequence is the same in every
// instantiation and kernel that inlines this function, so a cell gets the same bits whether its row runs in the pair or in the one-cell kernel, in a whole
// lattice or in a domain of a decomposed one (tests/test_gpu_native_arith.py::test_native_arithmetic_gives_a_cell_the_same_values_in_either_kernel).
// on_fields(rho, ux, uy, uz): called once the cell's density and (force-shifted, clamped) velocity are final, i.e. BEFORE the relaxation -- the caller stores
// the fields there (last step of a run) instead of keeping four registers alive through the relaxation loop.
struct NoFieldSink { __device__ __forceinline__ void operator()(float, float, float, float) const {} };
template<int FORCE=PAIR_FORCE_ANY, bool RAW=false, typename FieldSink=NoFieldSink> __device__ __forceinline__ void collide_cell_pk_native(const KParams& p,
	const uint32_t n, const uint8_t flagsn, const bool may_force, float& f0, f32x2* fp, const float* __restrict__ rho, const float* __restrict__ u,
	const float* __restrict__ F, float& rhon, float& uxn, float& uyn, float& uzn, float* u_before_force = nullptr, const ForceRefs* refs = nullptr,
	const ForceRefs* own = nullptr, const FieldSink on_fields = FieldSink{}) {
	constexpr float UP = RAW ? 0x1p+112f : 1.0f, DOWN = RAW ? 0x1p-112f : 1.0f;
	const bool is_E = (flagsn&TYPE_BO)==TYPE_E;
	const bool wave_has_E = __ballot(is_E)!=0ull;
	// pair sums s_k = f[2k+1] + f[2k+2] (.x) and NEGATED differences -d_k = f[2k+2] - f[2k+1] (.y), one packed addition per pair; the three pairs of
	// pairs that share an axis (k = 3 / 6: +-x +-y, 4 / 7: +-x +-z, 5 / 8: +-y +-z) summed as pairs again: { s + s', -(d + d') }
	f32x2 sd[9];
	float nmx, nmy, nmz;                                           // -m = -sum c f
	{
		#pragma unroll
		for(int k=0; k<9; k++) sd[k] = sum_and_negated_difference(fp[k]);
		const f32x2 p36 = sd[3]+sd[6], p47 = sd[4]+sd[7], p58 = sd[5]+sd[8];
		const float sum = ((f0+sd[0].x)+(sd[1].x+sd[2].x))+((p36.x+p47.x)+p58.x);
		rhon = fmaf(sum, UP, 1.0f);
		nmx = (sd[0].y+p36.y)+p47.y;
		nmy = (sd[1].y+(sd[3].y-sd[6].y))+p58.y;
		nmz = (sd[2].y+(sd[4].y-sd[7].y))+(sd[5].y-sd[8].y);
	}
	if(wave_has_E) { if(is_E) rhon = own ? own->wb : rho[n]; }
	float r = __builtin_amdgcn_rcpf(rhon);
	if constexpr(LUW_NATIVE_RCP_NEWTON!=0) r = fmaf(fmaf(-rhon, r, 1.0f), r, r);
	{ const float nr = -UP*r; uxn = nmx*nr; uyn = nmy*nr; uzn = nmz*nr; }
	if(wave_has_E) {
		if(is_E) {
			if(own) { uxn = own->tu[0]; uyn = own->tu[1]; uzn = own->tu[2]; }
			else { uxn = u[n]; uyn = u[(size_t)p.Np+n]; uzn = u[2ull*p.Np+n]; }
		}
	}
	if(u_before_force) { u_before_force[0] = uxn; u_before_force[1] = uyn; u_before_force[2] = uzn; }
	const bool forced = FORCE==PAIR_FORCE_UNIFORM || (FORCE==PAIR_FORCE_ANY && may_force);
	float fxn = 0.0f, fyn = 0.0f, fzn = 0.0f;
	if(forced) {
		fxn = p.fx; fyn = p.fy; fzn = p.fz;
		if(p.coriolis) { // -2 rho omega x u
			fxn = fmaf(rhon, fmaf(p.m2omy, uzn, -(p.m2omz*uyn)), fxn);
			fyn = fmaf(rhon, fmaf(p.m2omz, uxn, -(p.m2omx*uzn)), fyn);
			fzn = fmaf(rhon, fmaf(p.m2omx, uyn, -(p.m2omy*uxn)), fzn);
		}
		if constexpr(FORCE==PAIR_FORCE_ANY) {
			if(refs) { // zone references fetched ahead (fetch_force_refs): nudging towards the nearest owned face, top sponge
				if(refs->zn) {
					const float wr = (refs->wb*p.buffer_inv_tau)*rhon;
					fxn = fmaf(wr, refs->tu[0]-uxn, fxn);
					fyn = fmaf(wr, refs->tu[1]-uyn, fyn);
					if(p.nudge_vertical==1u) fzn = fmaf(wr, refs->tu[2]-uzn, fzn);
				}
				if(refs->zs) {
					const float sr = refs->sg*rhon;
					fxn = fmaf(sr, refs->su[0]-uxn, fxn);
					fyn = fmaf(sr, refs->su[1]-uyn, fyn);
					fzn = fmaf(sr, refs->su[2]-uzn, fzn);
				}
			}
			if(p.has_F) { fxn += F[n]; fyn += F[(size_t)p.Np+n]; fzn += F[2ull*p.Np+n]; }
		}
		const float rho2 = 0.5f*r;
		uxn = fmaf(fxn, rho2, uxn); uyn = fmaf(fyn, rho2, uyn); uzn = fmaf(fzn, rho2, uzn);
	}
	uxn = __builtin_amdgcn_fmed3f(uxn, -DEF_C, DEF_C);
	uyn = __builtin_amdgcn_fmed3f(uyn, -DEF_C, DEF_C);
	uzn = __builtin_amdgcn_fmed3f(uzn, -DEF_C, DEF_C);
	on_fields(rhon, uxn, uyn, uzn);
	// equilibrium ingredients (FX/kernel.cpp:1016-1055): f_eq(2k+1 / 2k+2) = rho w_k (A_k / 2 +- v_k) + (rho - 1) w_k, v_k = 3 c_k.u, A_k = v_k^2 - 3 u^2
	const float c3 = -3.0f*fmaf(uzn, uzn, fmaf(uyn, uyn, uxn*uxn));
	const float ux3 = 3.0f*uxn, uy3 = 3.0f*uyn, uz3 = 3.0f*uzn;
	const float v[9] = { ux3, uy3, uz3, ux3+uy3, ux3+uz3, uy3+uz3, ux3-uy3, ux3-uz3, uy3-uz3 };
	const float rhom1 = rhon-1.0f;
	const float rhos = DEF_WS*rhon, rhoe = DEF_WE*rhon, rhom1s = DEF_WS*rhom1, rhom1e = DEF_WE*rhom1;
	float A[9];
	#pragma unroll
	for(int k=0; k<9; k++) A[k] = fmaf(v[k], v[k], c3);
	float w = p.w;
	// Smagorinsky-Lilly, FX/kernel.cpp:1723-1737, from the non-equilibrium pair sums n_(2k+1) + n_(2k+2) = s_k - (rho w_k A_k + 2 (rho - 1) w_k)
	if(p.subgrid) {
		const float rm2s = 2.0f*rhom1s, rm2e = 2.0f*rhom1e;
		float sn[9];
		#pragma unroll
		for(int k=0; k<9; k++) sn[k] = fmaf(sd[k].x, UP, -fmaf(k<3 ? rhos : rhoe, A[k], k<3 ? rm2s : rm2e));
		const float Hxx = (sn[0]+(sn[3]+sn[4]))+(sn[6]+sn[7]), Hyy = (sn[1]+(sn[3]+sn[5]))+(sn[6]+sn[8]), Hzz = (sn[2]+(sn[4]+sn[5]))+(sn[7]+sn[8]);
		const float Hxy = sn[3]-sn[6], Hxz = sn[4]-sn[7], Hyz = sn[5]-sn[8];
		const float Q = fmaf(2.0f, fmaf(Hyz, Hyz, fmaf(Hxz, Hxz, Hxy*Hxy)), fmaf(Hzz, Hzz, fmaf(Hyy, Hyy, Hxx*Hxx)));
		const float sq = 0.76421222f*__builtin_amdgcn_sqrtf(Q);
		w = __builtin_amdgcn_rcpf(fmaf(0.5f, __builtin_amdgcn_sqrtf(fmaf(sq, r, p.tau0sq)), p.half_tau0));
	}
	float c_tau = fmaf(-0.5f, w, 1.0f);
	if(wave_has_E) { w = is_E ? 1.0f : w; c_tau = is_E ? 0.0f : c_tau; }
	const float omw = 1.0f-w;
	// relaxation with the rate folded into the equilibrium's coefficients: w f_eq(+-) = W (A / 2 +- v) + M, W = w rho w_k, M = w (rho - 1) w_k (RAW: times
	// 2^-112)
	const float wd = w*DOWN;
	const float Ws = wd*rhos, We = wd*rhoe, Ms = wd*rhom1s, Me = wd*rhom1e;
	const float weq0 = wd*(DEF_W0*fmaf(rhon, 0.5f*c3, rhom1));      // w f_eq of the rest population
	if(forced) {
		// c_tau Fin_i = fma(+-c.H, +-v + 1, uH): H = c_tau w9 F / 3 (w9 = 1/2 axis, 1/4 diagonal), uH = -c_tau w9 (u.F) / 3; Fin_0 = -c_tau (u.F); the constant
		// part uH joins M
		const float cs = (c_tau*DOWN)*0.16666667f;
		const float hx = cs*fxn, hy = cs*fyn, hz = cs*fzn;
		const float dots = cs*fmaf(uzn, fzn, fmaf(uyn, fyn, uxn*fxn));    // = -uH of the axis pairs
		const float ex = 0.5f*hx, ey = 0.5f*hy, ez = 0.5f*hz;
		const float Mds = Ms-dots, Mde = fmaf(-0.5f, dots, Me);
		f0 = fmaf(omw, f0, fmaf(-6.0f, dots, weq0));
		// uniform-force instantiation (everything in registers, 96 of them for 5 waves): the six diagonal 3 c.u are formed AGAIN here instead of living from
		// the
		// equilibrium ingredients on -- six additions for six registers, without which the kernel spills (the empty asm keeps the compiler from reusing them)
		float vx = ux3, vy = uy3, vz = uz3;
		if constexpr(FORCE==PAIR_FORCE_UNIFORM) asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz));
		#pragma unroll
		for(int k=0; k<9; k++) {
			// c_k.H formed where it is used (six values live instead of nine)
			const float cH = k==0 ? hx : k==1 ? hy : k==2 ? hz : k==3 ? ex+ey : k==4 ? ex+ez : k==5 ? ey+ez : k==6 ? ex-ey : k==7 ? ex-ez : ey-ez;
			const float vk = FORCE!=PAIR_FORCE_UNIFORM ? v[k]
				: k==0 ? vx : k==1 ? vy : k==2 ? vz : k==3 ? vx+vy : k==4 ? vx+vz : k==5 ? vy+vz : k==6 ? vx-vy : k==7 ? vx-vz : vy-vz;
			const f32x2 in = __builtin_elementwise_fma(splat2(0.5f), splat2(A[k]), pm2(vk));
			const f32x2 fin = __builtin_elementwise_fma(pm2(cH), pm2(vk)+splat2(1.0f), splat2(k<3 ? Mds : Mde));
			fp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], __builtin_elementwise_fma(splat2(k<3 ? Ws : We), in, fin));
		}
	} else {
		f0 = fmaf(omw, f0, weq0);
		#pragma unroll
		for(int k=0; k<9; k++) {
			const f32x2 in = __builtin_elementwise_fma(splat2(0.5f), splat2(A[k]), pm2(v[k]));
			fp[k] = __builtin_elementwise_fma(splat2(omw), fp[k], __builtin_elementwise_fma(splat2(k<3 ? Ws : We), in, splat2(k<3 ? Ms : Me)));
		}
	}
}
// position-only test: can buffer nudging or the top sponge act on this cell (the zones of assemble_force)?
__device__ __forceinline__ bool in_force_zone(const KParams& p, const uint32_t x, const uint32_t y, const uint32_t z) {
	return x-p.zw_lo<p.zw_n || x-p.ze_lo<p.ze_n || y-p.zs_lo<p.zs_n || y-p.zn_lo<p.zn_n || z-p.zt_lo<p.zt_n || z-p.zp_lo<p.zp_n;
}

// ---------------------------------------------------------------- thermal D3Q7 lattice (TEMPERATURE), FX/kernel.cpp:1306-1335,1639-1684
// D3Q7 equilibrium of the shifted populations: rest T / 4 - 1 / 4, axis pair a: (T / 8)(1 +- 4 u_a) - 1 / 8 as one fma each on T / 2 and (T - 1) / 8
__device__ __forceinline__ void calculate_g_eq(const float T, const float ux, const float uy, const float uz, float* geq) {
	const float half_T = 0.5f*T, shift = 0.125f*(T-1.0f);
	const float ua[3] = { ux, uy, uz };
	geq[0] = fmaf(0.25f, T, -0.25f);
	#pragma unroll
	for(int a=0; a<3; a++) { geq[2*a+1] = fmaf(half_T, ua[a], shift); geq[2*a+2] = fmaf(half_T, -ua[a], shift); }
}
// the cell update on streamed-in populations g[7] (in place): T = sum g + 1 (or the preset on TYPE_T cells), top sponge on T, BGK with w_T
// (TYPE_T: g = g_eq); writes T of a cell that is not preset.  FX/kernel.cpp:1652-1684
// write_T: like rho and u (write_fields), T is a by-product -- every step recomputes it from g; the only T the kernel READS are presets and, under the
// sponge, the top layer's -- so it is stored by the steps whose fields are looked at (the last of a run, sampled steps, every step on request or when
// a top-layer cell under the sponge is not a preset: reference_cells_are_inputs)
__device__ __forceinline__ void thermal_cell(const KParams& p, const uint32_t n, const uint32_t x, const uint32_t y, const uint32_t z, const uint8_t flagsn,
		const float ux, const float uy, const float uz, float* __restrict__ Tf, float* g, const bool write_T = true) {
	const bool preset = (flagsn&TYPE_T)!=0u;
	float Tn;
	if(preset) Tn = Tf[n];
	else { Tn = 0.0f; for(int i=0; i<7; i++) Tn += g[i]; Tn += 1.0f; }
	if(!preset && (flagsn&TYPE_BO)!=TYPE_E && z-p.zp_lo<p.zp_n) Tn = fmaf(p.sigma[(uint32_t)p.top_z-1u-z], Tf[x+(y+(uint32_t)p.top_z*p.Ny)*p.Px]-Tn, Tn);
	float geq[7];
	calculate_g_eq(Tn, ux, uy, uz, geq);
	if(preset) { for(int i=0; i<7; i++) g[i] = geq[i]; }
	else {
		if(write_T) Tf[n] = Tn;
		const float omw = 1.0f-p.w_T;
		for(int i=0; i<7; i++) g[i] = fmaf(omw, g[i], p.w_T*geq[i]);
	}
}
// One cell of the temperature lattice: Esoteric-Pull stream-in, T = sum g + 1 (or the preset on TYPE_T cells), top sponge
// on T, BGK with w_T (TYPE_T: g = g_eq), stream-out.  n, jx, jy, jz are ELEMENT indices of the cell and its +x, +y, +z
// neighbours; (ux,uy,uz) is the velocity before the force shift.  The buoyancy term it would add to the force carries the
// factor (fx,fy,fz), which LUW sets to zero (FX/setup.cpp:4935): T is a passive scalar here.
template<typename T,
	int PARITY> __device__ __forceinline__ void thermal_collide(const KParams& p, const uint32_t n, const uint32_t jx, const uint32_t jy, const uint32_t jz,
		const uint32_t x, const uint32_t y, const uint32_t z, const uint8_t flagsn, const float ux, const float uy, const float uz, const T* __restrict__ gi,
			float* __restrict__ Tf, float* g, const bool write_T = true) {
	const size_t Np = p.Np;
	const uint32_t jn[3] = { jx, jy, jz };
	g[0] = ddf_decode<T>(gi[n]);
	#pragma unroll
	for(int k=0; k<3; k++) {
		const int i = 2*k+1;
		g[i  ] = ddf_decode<T>(gi[(size_t)(PARITY ? i : i+1)*Np+n]);
		g[i+1] = ddf_decode<T>(gi[(size_t)(PARITY ? i+1 : i)*Np+jn[k]]);
	}
	thermal_cell(p, n, x, y, z, flagsn, ux, uy, uz, Tf, g, write_T);
}
// stream-out of the 7 encoded populations (Esoteric-Pull slots); code_of(i) yields the storage value of population i
template<typename T, int PARITY, typename F> __device__ __forceinline__ void thermal_store(const KParams& p, const uint32_t n, const uint32_t jx,
	const uint32_t jy, const uint32_t jz, T* __restrict__ gi, F code_of) {
	const size_t Np = p.Np;
	const uint32_t jn[3] = { jx, jy, jz };
	gi[n] = code_of(0);
	#pragma unroll
	for(int k=0; k<3; k++) {
		const int i = 2*k+1;
		gi[(size_t)(PARITY ? i+1 : i)*Np+jn[k]] = code_of(i);
		gi[(size_t)(PARITY ? i : i+1)*Np+n] = code_of(i+1);
	}
}
} // namespace luw
