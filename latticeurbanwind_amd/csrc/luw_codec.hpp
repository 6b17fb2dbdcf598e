// luw_codec.hpp -- the FP16C storage format (1 sign, 4 exponent, 11 mantissa bits: FX/kernel.cpp:864-875) on gfx950: literal restatements of the reference's
// integer formulas, the three-instruction decode, the median-of-three encode, the round-toward-zero tail encode of the step kernels and the product-free
// encode of the native kernels.  Included by luw_device.hpp INSIDE namespace luw.
#pragma once

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
