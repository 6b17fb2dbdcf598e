// valu_microbench.hip -- what single VALU instructions cost on gfx950, in shader clocks per wave64 instruction and SIMD, with 1 / 2 / 4
// resident waves per SIMD.  Each kernel runs a long unrolled stream of INDEPENDENT instances of one instruction (8 destination
// registers in rotation) and reads the shader clock (s_memtime) before and after.  Decides which instructions the VALU-bound FP16C
// kernel can afford.   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/vm tools/valu_microbench.hip && /tmp/vm
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define LOOPS 256

template<int K> __global__ __launch_bounds__(64) void k(unsigned long long* out, float seed) {
	float a0 = seed, a1 = seed+1, a2 = seed+2, a3 = seed+3, a4 = seed+4, a5 = seed+5, a6 = seed+6, a7 = seed+7;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p0 = {seed, seed}, p1 = p0+1.f, p2 = p0+2.f, p3 = p0+3.f, p4 = p0+4.f, p5 = p0+5.f, p6 = p0+6.f, p7 = p0+7.f;
	float b = seed*0.5f, c = seed*0.25f; f2 pb = {b, b}, pc = {c, c};
	unsigned long long q0 = 1, q1 = 2, q2 = 3, q3 = 4, q4 = 5, q5 = 6, q6 = 7, q7 = 8, qb = 77;
	unsigned u0 = __float_as_uint(a0), u1 = u0+1, u2 = u0+2, u3 = u0+3, u4 = u0+4, u5 = u0+5, u6 = u0+6, u7 = u0+7, ub = 12345u;
	const unsigned long long t0 = __builtin_readcyclecounter();
	for(int it=0; it<LOOPS; it++) {
		#define F(i) if constexpr(K==0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c)); \
			else if constexpr(K==1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##i) : "v"(pb), "v"(pc)); \
			else if constexpr(K==2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(pb)); \
			else if constexpr(K==3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##i) : "v"(pb)); \
			else if constexpr(K==4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==6) asm volatile("v_lshlrev_b32_sdwa %0, 12, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(u##i)); \
			else if constexpr(K==7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u##i) : "v"(ub), "v"(u0)); \
			else if constexpr(K==9) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u##i) : "v"(ub), "v"(u0)); \
			else if constexpr(K==10) asm volatile("v_add_lshl_u32 %0, %0, %1, 4" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==12) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i)); \
			else if constexpr(K==13) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i)); \
			else if constexpr(K==14) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a##i) : "v"(b) : "vcc"); \
			else if constexpr(K==15) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c)); \
			else if constexpr(K==16) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c)); \
			else if constexpr(K==17) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c)); \
			else if constexpr(K==18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==19) asm volatile("v_ashrrev_i32 %0, 4, %0" : "+v"(u##i)); \
			else if constexpr(K==20) asm volatile("v_mov_b32 %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==21) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c)); \
			else if constexpr(K==22) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(a##i) : "s"(0x1p-10f)); \
			else if constexpr(K==23) asm volatile("v_cvt_rpi_i32_f32_e32 %0, %0" : "+v"(u##i)); \
			else if constexpr(K==24) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u##i) : "v"(ub), "v"(u0)); \
			else if constexpr(K==25) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u##i) : "v"(ub) : "s20", "s21"); \
			else if constexpr(K==26) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" :: "v"(a##i), "v"(b) : "vcc"); \
			else if constexpr(K==27) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" :: "v"(a##i), "v"(b) : "s20", "s21"); \
			else if constexpr(K==28) asm volatile("v_cmp_class_f32_e32 vcc, %0, %1" :: "v"(a##i), "v"(ub) : "vcc"); \
			else if constexpr(K==29) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==30) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(u##i) : "v"(ub) : "vcc"); \
			else if constexpr(K==31) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(u##i) : "s20"); \
			else if constexpr(K==32) asm volatile("v_writelane_b32 %0, s20, 3" : "+v"(u##i)); \
			else if constexpr(K==33) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(u##i)); \
			else if constexpr(K==34) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==35) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==36) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==37) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c)); \
			else if constexpr(K==38) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==39) asm volatile("v_fma_f32 %0, %0, %1, s20" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==40) asm volatile("v_mul_f32_e32 %0, 0x3f43a369, %0" : "+v"(a##i)); \
			else if constexpr(K==41) asm volatile("v_mul_f32_e32 %0, s20, %0" : "+v"(a##i)); \
			else if constexpr(K==42) asm volatile("v_add_f32_e32 %0, %0, %0" : "+v"(a##i)); \
			else if constexpr(K==43) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %4, %5" : "+v"(a##i), "+v"(p##i) : "v"(b), "v"(c), "v"(pb), "v"(pc)); \
			else if constexpr(K==44) asm volatile("v_lshlrev_b32_e32 %0, 16, %0" : "+v"(u##i)); \
			else if constexpr(K==45) asm volatile("v_and_b32_e32 %0, 0xffff0000, %0" : "+v"(u##i)); \
			else if constexpr(K==46) asm volatile("v_cvt_f32_f16_e32 %0, %0" : "+v"(a##i)); \
			else if constexpr(K==47) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a##i)); \
			else if constexpr(K==48) asm volatile("v_cvt_f16_f32_e32 %0, %0" : "+v"(a##i)); \
			else if constexpr(K==49) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==50) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==51) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q##i) : "v"(qb)); \
			else if constexpr(K==52) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(u##i) : "v"(ub), "v"(u0)); \
			else if constexpr(K==53) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a##i) : "v"(ub)); \
			else if constexpr(K==54) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##i)); \
			else if constexpr(K==55) asm volatile("s_nop 0" ::); \
			else if constexpr(K==56) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u##i) : "v"(ub), "v"(u0)); \
			else if constexpr(K==57) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==58) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(p##i) : "v"(pb), "v"(pc)); \
			else if constexpr(K==59) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u##i)); \
			else if constexpr(K==60) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==61) asm volatile("v_cndmask_b32_e64 %0, %0, 0, s[20:21]" : "+v"(u##i)); \
			else if constexpr(K==62) asm volatile("v_lshlrev_b32_e64 %0, 16, %0" : "+v"(u##i)); \
			else if constexpr(K==63) asm volatile("v_lshrrev_b32_e32 %0, 16, %0" : "+v"(u##i)); \
			else if constexpr(K==64) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==65) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==66) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==67) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==68) asm volatile("v_cvt_f32_i32_e32 %0, %0" : "+v"(a##i)); \
			else if constexpr(K==69) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u##i)); \
			else if constexpr(K==70) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u##i) : "v"(ub), "v"(u0)); \
			else if constexpr(K==71) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u##i) : "v"(ub), "v"(u0)); \
			else if constexpr(K==72) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==73) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f43a369" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==74) asm volatile("v_fmamk_f32 %0, %0, 0x3f43a369, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==75) asm volatile("v_mul_f32_e64 %0, %0, %1 mul:2" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==76) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==77) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==78) asm volatile("v_sub_f32_e32 %0, %0, %1\n\tv_mul_f32_e32 %0, %0, %1" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==79) asm volatile("v_add_u32_e32 %0, s20, %0" : "+v"(u##i)); \
			else if constexpr(K==80) asm volatile("v_add_u32_e32 %0, 0x12345, %0" : "+v"(u##i)); \
			else if constexpr(K==81) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a##i) : "v"(b)); \
			else if constexpr(K==82) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p##i) : "v"(pb)); \
			else if constexpr(K==83) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(u##i)); \
			else if constexpr(K==84) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==85) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c)); \
			else if constexpr(K==86) asm volatile("v_trunc_f32_e32 %0, %0" : "+v"(a##i)); \
			else if constexpr(K==87) asm volatile("v_mov_b64 %0, %1" : "+v"(q##i) : "v"(qb)); \
			else if constexpr(K==88) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(q##i) : "v"(qb)); \
			else if constexpr(K==89) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==90) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(a##i) : "v"(ub), "v"(c)); \
			else if constexpr(K==91) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(u##i) : "v"(ub)); \
			else if constexpr(K==92) asm volatile("v_sub_f32_e32 %0, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(b));
		REP8(F) REP8(F) REP8(F) REP8(F)
		#undef F
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float s = a0+a1+a2+a3+a4+a5+a6+a7+p0.x+p1.y+p2.x+p3.y+p4.x+p5.y+p6.x+p7.y+__uint_as_float(u0^u1^u2^u3^u4^u5^u6^u7)+(float)(q0^q1^q2^q3^q4^q5^q6^q7);
	if(threadIdx.x==0) out[blockIdx.x] = t1-t0;
	if(s==123.456f) out[0] = 0ull; // keep everything alive
}

template<int K> static void run(const char* name) {
	int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	printf("%-34s", name);
	for(int waves : {1, 2, 4, 8}) { // resident waves per SIMD: blocks of one wave, cus*4*waves of them (the dispatcher spreads them evenly)
		const int blocks = cus*4*waves;
		unsigned long long* d; hipMalloc(&d, 8ull*blocks);
		hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(64), 0, 0, d, 1.0f); hipDeviceSynchronize();
		hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(64), 0, 0, d, 1.0f); hipDeviceSynchronize();
		std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), d, 8ull*blocks, hipMemcpyDeviceToHost); hipFree(d);
		std::sort(h.begin(), h.end());
		const double per_wave = (double)h[blocks/2]/(LOOPS*32.0);          // clocks per instruction as ONE wave sees it
		printf("  %dw: %5.2f clk/instr/wave = %5.2f per SIMD", waves, per_wave, per_wave/waves);
	}
	printf("\n");
}
__global__ void k_tick(unsigned long long* out) { // shader-clock ticks (s_memtime) per 100 MHz reference tick (s_memrealtime) while the VALU is busy
	float a = threadIdx.x; const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
	for(int i=0; i<200000; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
	const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), t1 = __builtin_readcyclecounter();
	if(threadIdx.x==0) { out[2*blockIdx.x] = t1-t0; out[2*blockIdx.x+1] = r1-r0; }
	if(a==123.f) out[0] = 0;
}
int main() {
	{ unsigned long long* d; hipMalloc(&d, 16ull*4096); hipLaunchKernelGGL(k_tick, dim3(4096), dim3(64), 0, 0, d); hipDeviceSynchronize();
	  unsigned long long h[2]; hipMemcpy(h, d+2*2048, 16, hipMemcpyDeviceToHost); hipFree(d);
	  printf("s_memtime runs at %.1f MHz (%.4f ticks per 100 MHz reference tick) with all SIMDs busy\n", 100.0*h[0]/h[1], (double)h[0]/h[1]); }
	printf("shader clocks (s_memtime ticks; compare rows, not absolute) per wave64 instruction, independent instructions, N resident waves per SIMD\n");
	run<0>("v_fma_f32"); run<21>("v_fmac_f32"); run<4>("v_add_f32"); run<5>("v_mul_f32"); run<22>("v_mul_f32 |x|, sgpr (e64)");
	run<1>("v_pk_fma_f32"); run<2>("v_pk_mul_f32"); run<3>("v_pk_add_f32");
	run<6>("v_lshlrev_b32_sdwa sext WORD_0"); run<19>("v_ashrrev_i32"); run<7>("v_and_b32"); run<8>("v_perm_b32"); run<9>("v_bfi_b32"); run<10>("v_add_lshl_u32");
	run<11>("v_cndmask_b32"); run<20>("v_mov_b32"); run<17>("v_med3_f32"); run<18>("v_max_f32"); run<23>("v_cvt_rpi_i32_f32");
	run<24>("v_cndmask_b32 d,a,b,vcc (d!=a)"); run<25>("v_cndmask_b32_e64 .., s[20:21]"); run<26>("v_cmp_gt_f32_e32 vcc"); run<27>("v_cmp_gt_f32_e64 s[20:21]");
	run<28>("v_cmp_class_f32_e32 vcc"); run<29>("v_min_u32"); run<30>("v_addc_co_u32 (vcc in+out)"); run<31>("v_readlane_b32"); run<32>("v_writelane_b32");
	run<33>("v_bfe_i32"); run<34>("v_lshl_or_b32"); run<35>("v_add_u32"); run<36>("v_sub_f32"); run<37>("v_fma_f32 with neg"); run<38>("v_fma_f32 inline const 1.0");
	run<39>("v_fma_f32 sgpr operand"); run<40>("v_mul_f32 literal"); run<41>("v_mul_f32 sgpr (e32)"); run<42>("v_add_f32 x,x"); run<43>("v_fma_f32 + v_pk_fma_f32 (pair)");
	run<44>("v_lshlrev_b32 16"); run<45>("v_and_b32 literal"); run<46>("v_cvt_f32_f16"); run<47>("v_cvt_f32_f16_sdwa WORD_1"); run<48>("v_cvt_f16_f32"); run<49>("v_cvt_pkrtz_f16_f32");
	run<50>("v_pk_mul_f16"); run<51>("v_lshl_add_u64"); run<52>("v_xad_u32"); run<53>("v_ldexp_f32"); run<54>("v_rsq_f32"); run<55>("s_nop 0"); run<56>("v_mad_u32_u24"); run<57>("v_mul_lo_u32");
	run<58>("v_pk_fma_f32 op_sel_hi:[0,1,1]"); run<59>("v_mov_b32_dpp quad_perm");
	run<60>("v_cndmask_b32_e64 .., vcc"); run<61>("v_cndmask_b32_e64 d,d,0,s[20:21]"); run<62>("v_lshlrev_b32_e64 16"); run<63>("v_lshrrev_b32 16"); run<64>("v_or_b32"); run<65>("v_xor_b32");
	run<66>("v_sub_u32"); run<67>("v_min_f32"); run<68>("v_cvt_f32_i32"); run<69>("v_bfe_u32"); run<70>("v_and_or_b32"); run<71>("v_add3_u32"); run<72>("v_lshl_add_u32");
	run<73>("v_fmaak_f32 (literal addend)"); run<74>("v_fmamk_f32 (literal factor)"); run<75>("v_mul_f32_e64 mul:2"); run<76>("v_mul_f32_e64 neg"); run<77>("v_add_f32_e64 neg");
	run<78>("v_sub_f32 + v_mul_f32 (pair, dependent)"); run<79>("v_add_u32 sgpr"); run<80>("v_add_u32 literal"); run<81>("v_fma_f32 inline 0.5"); run<82>("v_pk_mul_f32 op_sel_hi:[1,0]");
	run<83>("v_mov_b32_sdwa WORD_1"); run<84>("v_and_b32_sdwa WORD_1"); run<85>("v_max3_f32"); run<86>("v_trunc_f32"); run<87>("v_mov_b64"); run<88>("v_pk_mov_b32"); run<89>("v_mul_u32_u24");
	run<90>("v_fma_mix_f32"); run<91>("v_alignbit_b32"); run<92>("v_sub_f32 + v_cndmask_e64 (pair)");
	run<12>("v_rcp_f32"); run<13>("v_sqrt_f32"); run<14>("v_div_scale_f32"); run<15>("v_div_fmas_f32"); run<16>("v_div_fixup_f32");
	return 0;
}
