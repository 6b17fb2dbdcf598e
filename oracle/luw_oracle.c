/*
 * luw_oracle.c -- CPU restatement of the reference's D3Q19 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load this file; the
 * product (latticeurbanwind_amd/csrc) never links, loads or falls back to it.
 *
 * What it restates (all citations: /root/reference/core/cfd_core/FluidX3D/src/, "FX/"):
 *   FP16C codec            FX/kernel.cpp:864-875
 *   SoA DDF index          FX/kernel.cpp:877-879
 *   c_i / w_i tables       FX/kernel.cpp:880-919, FX/lbm.cpp:674-676
 *   neighbours (periodic)  FX/kernel.cpp:920-974
 *   f_eq                   FX/kernel.cpp:1016-1055
 *   rho,u moments          FX/kernel.cpp:1075-1100
 *   Guo forcing            FX/kernel.cpp:1103-1113
 *   Esoteric-Pull ld/st    FX/kernel.cpp:1338-1351
 *   initialize             FX/kernel.cpp:1370-1452
 *   stream_collide         FX/kernel.cpp:1475-1780 (SRT + SUBGRID + VOLUME_FORCE + FORCE_FIELD +
 *                          EQUILIBRIUM_BOUNDARIES + UPDATE_FIELDS + Coriolis + BUFFER_NUDGING + TOP_SPONGE)
 *   halo extract/insert    FX/kernel.cpp:2188-2270
 *   constants baked into the kernel source as 9-digit decimal text: FX/lbm.cpp:626-782,
 *   FX/utilities.hpp:2603-2634,2741-2750
 *
 * Arithmetic contract: FP32 throughout, fmaf() exactly where the reference writes fma(), every other
 * operation a separately rounded IEEE op (build with -ffp-contract=off), correctly rounded / and sqrtf.
 * The reference itself is compiled by the OpenCL driver with -cl-mad-enable (FX/opencl.hpp:305), i.e. it
 * is not bit-defined across compilers; this file fixes one legal evaluation.
 *
 * PINNING STATUS: the reference's own tests hold no golden vectors for this path.  The
 * oracle is pinned instead by (1) known-answer tests in tests/test_oracle_*.py and (2) fields produced by
 * the REAL reference binary (oracle/_ref/FluidX3D, built by oracle/build_ref.sh) run on an MI355X through
 * the AMD OpenCL runtime, committed under tests/golden/ref_*; see DESIGN.md "Oracle".
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TYPE_S 0x01
#define TYPE_E 0x02
#define TYPE_T 0x04
#define TYPE_BO 0x03
#define TYPE_SU 0x38
#define TYPE_G 0x20

typedef struct LuwOracleCfg {
	uint32_t Nx, Ny, Nz;          /* local lattice (incl. halo layers when D>1), FX/lbm.cpp:627-629 */
	uint32_t Dx, Dy, Dz;          /* number of domains per axis, FX/lbm.cpp:633-635 */
	int32_t Ox, Oy, Oz;           /* global offset of local (0,0,0), FX/lbm.cpp:637-639 */
	float w;                      /* def_w = 1/tau after the decimal-text round trip, FX/lbm.cpp:664 */
	float fx, fy, fz;             /* global volume force, FX/lbm.cpp:345 */
	float omega_x, omega_y, omega_z; /* Coriolis rotation, FX/kernel.cpp:1517-1519 */
	int32_t fp16c;                /* 0: FP32 DDFs, 1: FP16C DDFs (FX/defines.hpp:13-14) */
	int32_t subgrid;              /* SUBGRID (always 1 in the shipped build) */
	int32_t buffer_active;        /* BUFFER_NUDGING, FX/lbm.cpp:771-776 */
	uint32_t buffer_N;
	float buffer_inv_tau;
	int32_t buffer_nudge_vertical;
	int32_t downstream_face;      /* 0 none, 1 west, 2 east, 3 south, 4 north; FX/setup.cpp:3749-3755 */
	int32_t sponge_active;        /* TOP_SPONGE with def_sponge_ref_mode==0, FX/lbm.cpp:777-782 */
	uint32_t sponge_N;
	float sponge_inv_tau;
	float w_T;                    /* TEMPERATURE: def_w_T = 1/(2 alpha + 1/2) after the decimal-text round trip, FX/lbm.cpp:750 */
} LuwOracleCfg;

/* optional thermal D3Q7 lattice (TEMPERATURE, FX/kernel.cpp:1306-1335,1442-1449,1639-1684): gi 7 planes like fi, T[N]; NULL = off */
typedef struct { void* gi; float* T; } LuwOracleThermal;

/* ------------------------------------------------------------------ helpers */
static inline float sq(const float x) { return x*x; }
static inline uint32_t as_uint(const float x) { uint32_t r; memcpy(&r, &x, 4); return r; }
static inline float as_float(const uint32_t x) { float r; memcpy(&r, &x, 4); return r; }
static inline float clampf(const float x, const float a, const float b) { return fminf(fmaxf(x, a), b); } /* OpenCL clamp = min(max(x,a),b) */

/* FX/kernel.cpp:864-869 */
float luwo_half_to_float_custom(const uint16_t x) {
	const uint32_t e = ((uint32_t)x&0x7800u)>>11;
	const uint32_t m = ((uint32_t)x&0x07FFu)<<12;
	const uint32_t v = as_uint((float)m)>>23;
	return as_float(((uint32_t)x&0x8000u)<<16 | (uint32_t)(e!=0u)*((e+112u)<<23|m) | (uint32_t)((e==0u)&(m!=0u))*((v-37u)<<23|((m<<(150u-v))&0x007FF000u)));
}
/* FX/kernel.cpp:870-875 */
uint16_t luwo_float_to_half_custom(const float x) {
	const uint32_t b = as_uint(x)+0x00000800u;
	const uint32_t e = (b&0x7F800000u)>>23;
	const uint32_t m = b&0x007FFFFFu;
	/* note: for e<113 the shift 124-e is evaluated only under the mask (e<113)&(e>100) in effect; keep it in range like the GPU does (shift amount mod 32) */
	const uint32_t sh = (124u-e)&31u;
	return (uint16_t)((b&0x80000000u)>>16 | (uint32_t)(e>112u)*((((e-112u)<<11)&0x7800u)|m>>12) | (uint32_t)((e<113u)&(e>100u))*((((0x007FF800u+m)>>sh)+1u)>>1));
}

/* FX/utilities.hpp:2603-2634 + 2741-2750: to_string(float) as used by device_defines(), then the float
 * literal is parsed back by the OpenCL compiler.  Returns the float the kernel actually sees. */
float luwo_literal_roundtrip(float x) {
	int neg = 0;
	if(x<0.0f) { neg = 1; x = -x; }
	if(isnan(x)||isinf(x)) return neg ? -x : x;
	int exponent = 0;
	if(x>=10.0f) {
		if(x>=1E32f) { x *= 1E-32f; exponent += 32; }
		if(x>=1E16f) { x *= 1E-16f; exponent += 16; }
		if(x>= 1E8f) { x *=  1E-8f; exponent +=  8; }
		if(x>= 1E4f) { x *=  1E-4f; exponent +=  4; }
		if(x>= 1E2f) { x *=  1E-2f; exponent +=  2; }
		if(x>= 1E1f) { x *=  1E-1f; exponent +=  1; }
	}
	if(x>0.0f && x<=1.0f) {
		if(x<1E-31f) { x *=  1E32f; exponent -= 32; }
		if(x<1E-15f) { x *=  1E16f; exponent -= 16; }
		if(x< 1E-7f) { x *=   1E8f; exponent -=  8; }
		if(x< 1E-3f) { x *=   1E4f; exponent -=  4; }
		if(x< 1E-1f) { x *=   1E2f; exponent -=  2; }
		if(x<  1E0f) { x *=   1E1f; exponent -=  1; }
	}
	uint32_t integral = (uint32_t)x;
	const float remainder = (x-(float)integral)*1E8f;
	uint32_t decimal = (uint32_t)remainder;
	if(remainder-(float)decimal>=0.5f) {
		decimal++;
		if(decimal>=100000000u) {
			decimal = 0u;
			integral++;
			if(integral>=10u) { integral = 1u; exponent++; }
		}
	}
	char text[64];
	if(exponent!=0) snprintf(text, sizeof(text), "%s%u.%08uE%d", neg?"-":"", integral, decimal, exponent);
	else snprintf(text, sizeof(text), "%s%u.%08u", neg?"-":"", integral, decimal);
	return strtof(text, NULL);
}

/* D3Q19 tables, FX/kernel.cpp:890-893 */
static const float CX[19] = {0, 1,-1, 0, 0, 0, 0, 1,-1, 1,-1, 0, 0, 1,-1, 1,-1, 0, 0};
static const float CY[19] = {0, 0, 0, 1,-1, 0, 0, 1,-1, 0, 0, 1,-1,-1, 1, 0, 0, 1,-1};
static const float CZ[19] = {0, 0, 0, 0, 0, 1,-1, 0, 0, 1,-1, 1,-1, 0, 0,-1, 1,-1, 1};
#define DEF_W0 (1.0f/3.0f)
#define DEF_WS (1.0f/18.0f)
#define DEF_WE (1.0f/36.0f)
#define DEF_C 0.57735027f
static inline float wi(const uint32_t i) { return i==0u ? DEF_W0 : (i<7u ? DEF_WS : DEF_WE); }

typedef struct { const LuwOracleCfg* c; uint64_t N; } Ctx;

static inline void coordinates(const LuwOracleCfg* c, const uint64_t n, uint32_t* x, uint32_t* y, uint32_t* z) { /* FX/kernel.cpp:833-836 */
	const uint64_t NxNy = (uint64_t)c->Nx*(uint64_t)c->Ny;
	const uint32_t t = (uint32_t)(n%NxNy);
	*x = t%c->Nx; *y = t/c->Nx; *z = (uint32_t)(n/NxNy);
}
static inline uint64_t index3(const LuwOracleCfg* c, const uint32_t x, const uint32_t y, const uint32_t z) { /* FX/kernel.cpp:837-839 */
	return (uint64_t)x+(uint64_t)(y+z*c->Ny)*(uint64_t)c->Nx;
}
static inline int is_halo(const LuwOracleCfg* c, const uint64_t n) { /* FX/kernel.cpp:856-859 */
	uint32_t x, y, z; coordinates(c, n, &x, &y, &z);
	return ((c->Dx>1u)&(x==0u||x>=c->Nx-1u))||((c->Dy>1u)&(y==0u||y>=c->Ny-1u))||((c->Dz>1u)&(z==0u||z>=c->Nz-1u));
}
static void neighbors(const LuwOracleCfg* c, const uint64_t n, uint64_t* j) { /* FX/kernel.cpp:920-958 */
	uint32_t x, y, z; coordinates(c, n, &x, &y, &z);
	const uint64_t Nx = c->Nx, NxNy = (uint64_t)c->Nx*(uint64_t)c->Ny;
	const uint64_t x0 = x, xp = (x+1u)%c->Nx, xm = (x+c->Nx-1u)%c->Nx;
	const uint64_t y0 = (uint64_t)y*Nx, yp = (uint64_t)((y+1u)%c->Ny)*Nx, ym = (uint64_t)((y+c->Ny-1u)%c->Ny)*Nx;
	const uint64_t z0 = (uint64_t)z*NxNy, zp = (uint64_t)((z+1u)%c->Nz)*NxNy, zm = (uint64_t)((z+c->Nz-1u)%c->Nz)*NxNy;
	j[0] = n;
	j[ 1] = xp+y0+z0; j[ 2] = xm+y0+z0;
	j[ 3] = x0+yp+z0; j[ 4] = x0+ym+z0;
	j[ 5] = x0+y0+zp; j[ 6] = x0+y0+zm;
	j[ 7] = xp+yp+z0; j[ 8] = xm+ym+z0;
	j[ 9] = xp+y0+zp; j[10] = xm+y0+zm;
	j[11] = x0+yp+zp; j[12] = x0+ym+zm;
	j[13] = xp+ym+z0; j[14] = xm+yp+z0;
	j[15] = xp+y0+zm; j[16] = xm+y0+zp;
	j[17] = x0+yp+zm; j[18] = x0+ym+zp;
}

static inline float load_fi(const LuwOracleCfg* c, const void* fi, const uint64_t idx) { /* FX/lbm.cpp:706-721 */
	return c->fp16c ? luwo_half_to_float_custom(((const uint16_t*)fi)[idx]) : ((const float*)fi)[idx];
}
static inline void store_fi(const LuwOracleCfg* c, void* fi, const uint64_t idx, const float v) {
	if(c->fp16c) ((uint16_t*)fi)[idx] = luwo_float_to_half_custom(v); else ((float*)fi)[idx] = v;
}
static void load_f(const LuwOracleCfg* c, const uint64_t N, const uint64_t n, float* fhn, const void* fi, const uint64_t* j, const uint64_t t) { /* FX/kernel.cpp:1338-1344 */
	fhn[0] = load_fi(c, fi, n);
	for(uint32_t i=1u; i<19u; i+=2u) {
		fhn[i   ] = load_fi(c, fi, (uint64_t)(t%2ull ? i    : i+1u)*N+n);
		fhn[i+1u] = load_fi(c, fi, (uint64_t)(t%2ull ? i+1u : i   )*N+j[i]);
	}
}
static void store_f(const LuwOracleCfg* c, const uint64_t N, const uint64_t n, const float* fhn, void* fi, const uint64_t* j, const uint64_t t) { /* FX/kernel.cpp:1345-1351 */
	store_fi(c, fi, n, fhn[0]);
	for(uint32_t i=1u; i<19u; i+=2u) {
		store_fi(c, fi, (uint64_t)(t%2ull ? i+1u : i   )*N+j[i], fhn[i   ]);
		store_fi(c, fi, (uint64_t)(t%2ull ? i    : i+1u)*N+n   , fhn[i+1u]);
	}
}

void luwo_calculate_f_eq(const float rho, float ux, float uy, float uz, float* feq) { /* FX/kernel.cpp:1016-1055 */
	const float rhom1 = rho-1.0f;
	const float c3 = -3.0f*(sq(ux)+sq(uy)+sq(uz));
	uz *= 3.0f;
	ux *= 3.0f;
	uy *= 3.0f;
	feq[ 0] = DEF_W0*fmaf(rho, 0.5f*c3, rhom1);
	const float u0=ux+uy, u1=ux+uz, u2=uy+uz, u3=ux-uy, u4=ux-uz, u5=uy-uz;
	const float rhos=DEF_WS*rho, rhoe=DEF_WE*rho, rhom1s=DEF_WS*rhom1, rhom1e=DEF_WE*rhom1;
	feq[ 1] = fmaf(rhos, fmaf(0.5f, fmaf(ux, ux, c3), ux), rhom1s); feq[ 2] = fmaf(rhos, fmaf(0.5f, fmaf(ux, ux, c3), -ux), rhom1s);
	feq[ 3] = fmaf(rhos, fmaf(0.5f, fmaf(uy, uy, c3), uy), rhom1s); feq[ 4] = fmaf(rhos, fmaf(0.5f, fmaf(uy, uy, c3), -uy), rhom1s);
	feq[ 5] = fmaf(rhos, fmaf(0.5f, fmaf(uz, uz, c3), uz), rhom1s); feq[ 6] = fmaf(rhos, fmaf(0.5f, fmaf(uz, uz, c3), -uz), rhom1s);
	feq[ 7] = fmaf(rhoe, fmaf(0.5f, fmaf(u0, u0, c3), u0), rhom1e); feq[ 8] = fmaf(rhoe, fmaf(0.5f, fmaf(u0, u0, c3), -u0), rhom1e);
	feq[ 9] = fmaf(rhoe, fmaf(0.5f, fmaf(u1, u1, c3), u1), rhom1e); feq[10] = fmaf(rhoe, fmaf(0.5f, fmaf(u1, u1, c3), -u1), rhom1e);
	feq[11] = fmaf(rhoe, fmaf(0.5f, fmaf(u2, u2, c3), u2), rhom1e); feq[12] = fmaf(rhoe, fmaf(0.5f, fmaf(u2, u2, c3), -u2), rhom1e);
	feq[13] = fmaf(rhoe, fmaf(0.5f, fmaf(u3, u3, c3), u3), rhom1e); feq[14] = fmaf(rhoe, fmaf(0.5f, fmaf(u3, u3, c3), -u3), rhom1e);
	feq[15] = fmaf(rhoe, fmaf(0.5f, fmaf(u4, u4, c3), u4), rhom1e); feq[16] = fmaf(rhoe, fmaf(0.5f, fmaf(u4, u4, c3), -u4), rhom1e);
	feq[17] = fmaf(rhoe, fmaf(0.5f, fmaf(u5, u5, c3), u5), rhom1e); feq[18] = fmaf(rhoe, fmaf(0.5f, fmaf(u5, u5, c3), -u5), rhom1e);
}

void luwo_calculate_rho_u(const float* f, float* rhon, float* uxn, float* uyn, float* uzn) { /* FX/kernel.cpp:1075-1100 */
	float rho=f[0], ux, uy, uz;
	for(uint32_t i=1u; i<19u; i++) rho += f[i];
	rho += 1.0f;
	ux = f[ 1]-f[ 2]+f[ 7]-f[ 8]+f[ 9]-f[10]+f[13]-f[14]+f[15]-f[16];
	uy = f[ 3]-f[ 4]+f[ 7]-f[ 8]+f[11]-f[12]+f[14]-f[13]+f[17]-f[18];
	uz = f[ 5]-f[ 6]+f[ 9]-f[10]+f[11]-f[12]+f[16]-f[15]+f[18]-f[17];
	*rhon = rho;
	*uxn = ux/rho;
	*uyn = uy/rho;
	*uzn = uz/rho;
}

static void calculate_forcing_terms(const float ux, const float uy, const float uz, const float fx, const float fy, const float fz, float* Fin) { /* FX/kernel.cpp:1103-1113 */
	const float uF = -0.33333334f*fmaf(ux, fx, fmaf(uy, fy, uz*fz));
	Fin[0] = 9.0f*DEF_W0*uF;
	for(uint32_t i=1u; i<19u; i++) {
		Fin[i] = 9.0f*wi(i)*fmaf(CX[i]*fx+CY[i]*fy+CZ[i]*fz, CX[i]*ux+CY[i]*uy+CZ[i]*uz+0.33333334f, uF);
	}
}

/* FX/kernel.cpp:1370-1452 (no SURFACE / MOVING_BOUNDARIES / TEMPERATURE part) */
/* FX/kernel.cpp:1307-1314: +x, -x, +y, -y, +z, -z neighbours (j7[1], j7[3], j7[5] are j[1], j[3], j[5] of the D3Q19 list) */
static void calculate_g_eq(const float T, const float ux, const float uy, const float uz, float* geq) { /* FX/kernel.cpp:1315-1321 */
	const float wsT4 = 0.5f*T, wsTm1 = 0.125f*(T-1.0f);
	geq[0] = fmaf(0.25f, T, -0.25f);
	geq[1] = fmaf(wsT4, ux, wsTm1); geq[2] = fmaf(wsT4, -ux, wsTm1);
	geq[3] = fmaf(wsT4, uy, wsTm1); geq[4] = fmaf(wsT4, -uy, wsTm1);
	geq[5] = fmaf(wsT4, uz, wsTm1); geq[6] = fmaf(wsT4, -uz, wsTm1);
}
static void load_g(const LuwOracleCfg* c, const uint64_t N, const uint64_t n, float* ghn, const void* gi, const uint64_t* j, const uint64_t t) { /* FX/kernel.cpp:1322-1328 */
	ghn[0] = load_fi(c, gi, n);
	for(uint32_t i=1u; i<7u; i+=2u) {
		ghn[i   ] = load_fi(c, gi, (uint64_t)(t%2ull ? i    : i+1u)*N+n);
		ghn[i+1u] = load_fi(c, gi, (uint64_t)(t%2ull ? i+1u : i   )*N+j[i]);
	}
}
static void store_g(const LuwOracleCfg* c, const uint64_t N, const uint64_t n, const float* ghn, void* gi, const uint64_t* j, const uint64_t t) { /* FX/kernel.cpp:1329-1335 */
	store_fi(c, gi, n, ghn[0]);
	for(uint32_t i=1u; i<7u; i+=2u) {
		store_fi(c, gi, (uint64_t)(t%2ull ? i+1u : i   )*N+j[i], ghn[i   ]);
		store_fi(c, gi, (uint64_t)(t%2ull ? i    : i+1u)*N+n,    ghn[i+1u]);
	}
}

void luwo_initialize_thermal(const LuwOracleCfg* c, void* gi, const float* T, const float* u, const uint8_t* flags) { /* FX/kernel.cpp:1442-1449; solids' u is 0 by then */
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	#pragma omp parallel for schedule(static)
	for(int64_t nn=0; nn<(int64_t)N; nn++) {
		const uint64_t n = (uint64_t)nn;
		if(is_halo(c, n)) continue;
		uint64_t j[19];
		neighbors(c, n, j);
		const int solid = (flags[n]&TYPE_BO)==TYPE_S;
		float geq[7];
		calculate_g_eq(T[n], solid ? 0.0f : u[n], solid ? 0.0f : u[N+n], solid ? 0.0f : u[2ull*N+n], geq);
		store_g(c, N, n, geq, gi, j, 1ull);
	}
}

void luwo_initialize(const LuwOracleCfg* c, void* fi, const float* rho, float* u, const uint8_t* flags) {
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	#pragma omp parallel for schedule(static)
	for(int64_t nn=0; nn<(int64_t)N; nn++) {
		const uint64_t n = (uint64_t)nn;
		if(is_halo(c, n)) continue;
		const uint8_t flagsn_bo = flags[n]&TYPE_BO;
		uint64_t j[19];
		neighbors(c, n, j);
		if(flagsn_bo==TYPE_S) { /* both branches of FX/kernel.cpp:1386-1399 end with u=0 on solid cells */
			u[n] = 0.0f; u[N+n] = 0.0f; u[2ull*N+n] = 0.0f;
		}
		float feq[19];
		luwo_calculate_f_eq(rho[n], u[n], u[N+n], u[2ull*N+n], feq);
		store_f(c, N, n, feq, fi, j, 1ull);
	}
}

/* FX/kernel.cpp:1475-1780 for one cell */
static void stream_collide_cell(const LuwOracleCfg* c, const uint64_t N, const uint64_t n, void* fi, float* rho, float* u, const uint8_t* flags, const float* F, const uint64_t t, const LuwOracleThermal* th) {
	if(is_halo(c, n)) return;
	const uint8_t flagsn = flags[n];
	const uint8_t flagsn_bo = flagsn&TYPE_BO, flagsn_su = flagsn&TYPE_SU;
	if(flagsn_bo==TYPE_S||flagsn_su==TYPE_G) return;

	uint64_t j[19];
	neighbors(c, n, j);
	float fhn[19];
	load_f(c, N, n, fhn, fi, j, t);

	float rhon, uxn, uyn, uzn;
	if(flagsn_bo==TYPE_E) {
		rhon = rho[n];
		uxn = u[n];
		uyn = u[N+n];
		uzn = u[2ull*N+n];
	} else {
		luwo_calculate_rho_u(fhn, &rhon, &uxn, &uyn, &uzn);
	}
	float fxn = c->fx, fyn = c->fy, fzn = c->fz;
	const float cor_x = -2.0f*rhon*(c->omega_y*uzn-c->omega_z*uyn);
	const float cor_y = -2.0f*rhon*(c->omega_z*uxn-c->omega_x*uzn);
	const float cor_z = -2.0f*rhon*(c->omega_x*uyn-c->omega_y*uxn);
	fxn += cor_x;
	fyn += cor_y;
	fzn += cor_z;

	/* derived face constants, FX/lbm.cpp:613-625 */
	const uint32_t Nx_global = (c->Nx-2u*(c->Dx>1u))*c->Dx;
	const uint32_t Ny_global = (c->Ny-2u*(c->Dy>1u))*c->Dy;
	const uint32_t Nz_global = (c->Nz-2u*(c->Dz>1u))*c->Dz;
	const int west_local_x = -c->Ox;
	const int east_local_x = (int)Nx_global-1-c->Ox;
	const int south_local_y = -c->Oy;
	const int north_local_y = (int)Ny_global-1-c->Oy;
	const int top_local_z = (int)Nz_global-1-c->Oz;
	const int has_west_face = west_local_x>=0&&west_local_x<(int)c->Nx ? 1 : 0;
	const int has_east_face = east_local_x>=0&&east_local_x<(int)c->Nx ? 1 : 0;
	const int has_south_face = south_local_y>=0&&south_local_y<(int)c->Ny ? 1 : 0;
	const int has_north_face = north_local_y>=0&&north_local_y<(int)c->Ny ? 1 : 0;
	const int has_top_face = top_local_z>=0&&top_local_z<(int)c->Nz ? 1 : 0;

	if(c->buffer_active) { /* FX/kernel.cpp:1523-1595 */
		if(flagsn_bo!=TYPE_E) {
			uint32_t x, y, z; coordinates(c, n, &x, &y, &z);
			const int xg = (int)x+c->Ox;
			const int yg = (int)y+c->Oy;
			const int zg = (int)z+c->Oz;
			const int Nbuf_i = (int)c->buffer_N;
			const int d_w_i = xg;
			const int d_e_i = (int)(Nx_global-1u)-xg;
			const int d_s_i = yg;
			const int d_n_i = (int)(Ny_global-1u)-yg;
			const int d_t_i = (int)(Nz_global-1u)-zg;
			const int in_w = c->downstream_face!=1&&has_west_face==1&&d_w_i>=0&&d_w_i<=Nbuf_i;
			const int in_e = c->downstream_face!=2&&has_east_face==1&&d_e_i>=0&&d_e_i<=Nbuf_i;
			const int in_s = c->downstream_face!=3&&has_south_face==1&&d_s_i>=0&&d_s_i<=Nbuf_i;
			const int in_n = c->downstream_face!=4&&has_north_face==1&&d_n_i>=0&&d_n_i<=Nbuf_i;
			const int in_t = has_top_face==1&&d_t_i>=0&&d_t_i<=Nbuf_i;
			if(in_w||in_e||in_s||in_n||in_t) {
				uint32_t d_min = c->buffer_N+1u;
				uint64_t n_ref = n;
				if(in_w) { const uint32_t d = (uint32_t)d_w_i; if(d<d_min) { d_min = d; n_ref = index3(c, (uint32_t)west_local_x, y, z); } }
				if(in_e) { const uint32_t d = (uint32_t)d_e_i; if(d<d_min) { d_min = d; n_ref = index3(c, (uint32_t)east_local_x, y, z); } }
				if(in_s) { const uint32_t d = (uint32_t)d_s_i; if(d<d_min) { d_min = d; n_ref = index3(c, x, (uint32_t)south_local_y, z); } }
				if(in_n) { const uint32_t d = (uint32_t)d_n_i; if(d<d_min) { d_min = d; n_ref = index3(c, x, (uint32_t)north_local_y, z); } }
				if(in_t) { const uint32_t d = (uint32_t)d_t_i; if(d<d_min) { d_min = d; n_ref = index3(c, x, y, (uint32_t)top_local_z); } }
				const float xi = 1.0f-(float)d_min/(float)c->buffer_N;
				float w_buf = sinf(1.5707963267948966f*xi);
				w_buf *= w_buf;
				const float u_target_x = u[n_ref];
				const float u_target_y = u[N+n_ref];
				const float u_target_z = u[2ull*N+n_ref];
				const float a_x = w_buf*c->buffer_inv_tau*(u_target_x-uxn);
				const float a_y = w_buf*c->buffer_inv_tau*(u_target_y-uyn);
				const float a_z = c->buffer_nudge_vertical==1 ? w_buf*c->buffer_inv_tau*(u_target_z-uzn) : 0.0f;
				fxn += rhon*a_x;
				fyn += rhon*a_y;
				fzn += rhon*a_z;
			}
		}
	}
	if(c->sponge_active) { /* FX/kernel.cpp:1596-1614 */
		if(flagsn_bo!=TYPE_E&&has_top_face==1) {
			uint32_t x, y, z; coordinates(c, n, &x, &y, &z);
			const int d_t_i = (int)(Nz_global-2u)-((int)z+c->Oz);
			const int Nsponge_i = (int)c->sponge_N;
			if(d_t_i>=0&&d_t_i<Nsponge_i) {
				const float xi = Nsponge_i>1 ? 1.0f-(float)d_t_i/(float)(Nsponge_i-1) : 1.0f;
				float sigma = sinf(1.5707963267948966f*xi);
				sigma = c->sponge_inv_tau*sigma*sigma;
				const uint64_t n_ref = index3(c, x, y, (uint32_t)top_local_z);
				fxn += rhon*sigma*(u[n_ref]-uxn);
				fyn += rhon*sigma*(u[N+n_ref]-uyn);
				fzn += rhon*sigma*(u[2ull*N+n_ref]-uzn);
			}
		}
	}

	if(th&&th->gi) { /* TEMPERATURE, FX/kernel.cpp:1639-1684; uses the velocity BEFORE the force shift */
		float ghn[7];
		load_g(c, N, n, ghn, th->gi, j, t);
		float Tn;
		if(flagsn&TYPE_T) Tn = th->T[n];
		else { Tn = 0.0f; for(uint32_t i=0u; i<7u; i++) Tn += ghn[i]; Tn += 1.0f; }
		if(c->sponge_active&&!(flagsn&TYPE_T)&&flagsn_bo!=TYPE_E&&has_top_face==1) { /* FX/kernel.cpp:1653-1667 */
			uint32_t x, y, z; coordinates(c, n, &x, &y, &z);
			const int d_t_i = (int)(Nz_global-2u)-((int)z+c->Oz);
			const int Nsponge_i = (int)c->sponge_N;
			if(d_t_i>=0&&d_t_i<Nsponge_i) {
				const float xi = Nsponge_i>1 ? 1.0f-(float)d_t_i/(float)(Nsponge_i-1) : 1.0f;
				float sigma_T = sinf(1.5707963267948966f*xi);
				sigma_T = c->sponge_inv_tau*sigma_T*sigma_T;
				Tn = fmaf(sigma_T, th->T[index3(c, x, y, (uint32_t)top_local_z)]-Tn, Tn);
			}
		}
		float geq[7];
		calculate_g_eq(Tn, uxn, uyn, uzn, geq);
		if(flagsn&TYPE_T) { for(uint32_t i=0u; i<7u; i++) ghn[i] = geq[i]; }
		else {
			th->T[n] = Tn; /* UPDATE_FIELDS */
			for(uint32_t i=0u; i<7u; i++) ghn[i] = fmaf(1.0f-c->w_T, ghn[i], c->w_T*geq[i]);
		}
		store_g(c, N, n, ghn, th->gi, j, t);
		/* buoyancy: fxn -= fx*beta*(Tn-T_avg) etc. -- LUW constructs the solver with fx=fy=fz=0 (FX/setup.cpp:4935): no-op */
	}
	float Fin[19];
	if(F) { /* FORCE_FIELD, FX/kernel.cpp:1617-1623 */
		fxn += F[n];
		fyn += F[N+n];
		fzn += F[2ull*N+n];
	}
	{ /* VOLUME_FORCE, FX/kernel.cpp:1686-1692 */
		const float rho2 = 0.5f/rhon;
		uxn = clampf(fmaf(fxn, rho2, uxn), -DEF_C, DEF_C);
		uyn = clampf(fmaf(fyn, rho2, uyn), -DEF_C, DEF_C);
		uzn = clampf(fmaf(fzn, rho2, uzn), -DEF_C, DEF_C);
		calculate_forcing_terms(uxn, uyn, uzn, fxn, fyn, fzn, Fin);
	}
	if(flagsn_bo!=TYPE_E) { /* UPDATE_FIELDS, FX/kernel.cpp:1709-1716 */
		rho[n] = rhon;
		u[n] = uxn;
		u[N+n] = uyn;
		u[2ull*N+n] = uzn;
	}
	float feq[19];
	luwo_calculate_f_eq(rhon, uxn, uyn, uzn, feq);
	float w = c->w;
	if(c->subgrid) { /* FX/kernel.cpp:1723-1737 */
		const float tau0 = 1.0f/w;
		float Hxx=0.0f, Hyy=0.0f, Hzz=0.0f, Hxy=0.0f, Hxz=0.0f, Hyz=0.0f;
		for(uint32_t i=1u; i<19u; i++) {
			const float fneqi = fhn[i]-feq[i];
			const float cxi=CX[i], cyi=CY[i], czi=CZ[i];
			Hxx += cxi*cxi*fneqi;
			Hxy += cxi*cyi*fneqi; Hyy += cyi*cyi*fneqi;
			Hxz += cxi*czi*fneqi; Hyz += cyi*czi*fneqi; Hzz += czi*czi*fneqi;
		}
		const float Q = sq(Hxx)+sq(Hyy)+sq(Hzz)+2.0f*(sq(Hxy)+sq(Hxz)+sq(Hyz));
		w = 2.0f/(tau0+sqrtf(sq(tau0)+0.76421222f*sqrtf(Q)/rhon));
	}
	const float c_tau = fmaf(w, -0.5f, 1.0f); /* FX/kernel.cpp:1741-1742 */
	for(uint32_t i=0u; i<19u; i++) Fin[i] *= c_tau;
	for(uint32_t i=0u; i<19u; i++) fhn[i] = flagsn_bo==TYPE_E ? feq[i] : fmaf(1.0f-w, fhn[i], fmaf(w, feq[i], Fin[i])); /* FX/kernel.cpp:1747 */
	store_f(c, N, n, fhn, fi, j, t);
}

/* the literal path: one call of stream_collide_cell per cell, exactly like the reference's one work-item per cell */
void luwo_stream_collide_literal(const LuwOracleCfg* c, void* fi, float* rho, float* u, const uint8_t* flags, const float* F, void* gi, float* T, const uint64_t t) {
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	const LuwOracleThermal th = { gi, T };
	#pragma omp parallel for schedule(static)
	for(int64_t n=0; n<(int64_t)N; n++) stream_collide_cell(c, N, (uint64_t)n, fi, rho, u, flags, F, t, gi ? &th : NULL);
}

/* ================================================================== row-wise path (same operations, organised for the host CPU)
 * stream_collide_cell spends most of its time outside the arithmetic it restates: three coordinates() per cell (64-bit div / mod), six % in neighbors(),
 * the face constants of FX/lbm.cpp:613-625 re-derived per cell, sinf per zone cell, and every fma() a call into libm.  The path below executes THE SAME
 * floating-point statements in THE SAME order for every cell -- each statement of stream_collide_cell appears once, applied to the eight cells of a row
 * chunk at a time (an IEEE single-precision add / mul / fma / div / sqrt yields the same value in a vector lane as in a scalar register; nothing is
 * reassociated or contracted: -ffp-contract=off, explicit fused operations only where the reference writes fma) -- and replaces only the index work:
 * rows are walked with additions, the neighbour rows (periodic wrap, FX/kernel.cpp:920-958) are formed once per row, the face constants once per call,
 * the sin^2 ramps (same sinf calls on the same arguments) once per call into tables.  Cells whose lane is switched off (halo, TYPE_S / TYPE_G: the early
 * returns of FX/kernel.cpp:1486-1496) compute on zeros and store nothing.
 * tests/test_oracle_fast_path.py: both paths give the same bits on every fixture state (FP32, FP16C, forces, zones, thermal lattice, halo'ed domains).
 * luwo_set_fast(0) selects the literal path for every call. */
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define LUWO_HAVE_FAST 1
typedef float v8f __attribute__((vector_size(32)));
typedef int32_t v8i __attribute__((vector_size(32)));
typedef uint32_t v8u __attribute__((vector_size(32)));
static inline v8f vfma(const v8f a, const v8f b, const v8f c) { return (v8f)_mm256_fmadd_ps((__m256)a, (__m256)b, (__m256)c); }
static inline v8f vsqrt(const v8f a) { return (v8f)_mm256_sqrt_ps((__m256)a); }
static inline v8f vsplat(const float a) { return (v8f){ a, a, a, a, a, a, a, a }; }
static inline v8f vsel(const v8i m, const v8f a, const v8f b) { return (v8f)_mm256_blendv_ps((__m256)b, (__m256)a, (__m256)m); } /* m ? a : b */
static inline v8f vclamp(const v8f x, const float a, const float b) { /* fminf(fmaxf(x, a), b): the second operand is returned for a NaN x, like fmaxf */
	return (v8f)_mm256_min_ps(_mm256_max_ps((__m256)x, (__m256)vsplat(a)), (__m256)vsplat(b));
}
static inline v8f vsq(const v8f x) { return x*x; }
static inline v8f v_half_to_float(const v8u x) { /* luwo_half_to_float_custom, lane by lane */
	const v8u e = (x&0x7800u)>>11;
	const v8u m = (x&0x07FFu)<<12;
	const v8u v = ((v8u)(v8f)_mm256_cvtepi32_ps((__m256i)m))>>23;
	const v8u normal = (v8u)(e!=0u)&(((e+112u)<<23)|m);
	const v8u denorm = (v8u)((e==0u)&(m!=0u))&(((v-37u)<<23)|((m<<((150u-v)&31u))&0x007FF000u));
	return (v8f)(((x&0x8000u)<<16)|normal|denorm);
}
static inline v8u v_float_to_half(const v8f x) { /* luwo_float_to_half_custom, lane by lane */
	const v8u b = (v8u)x+0x00000800u;
	const v8u e = (b&0x7F800000u)>>23;
	const v8u m = b&0x007FFFFFu;
	const v8u sh = (124u-e)&31u;
	return ((b&0x80000000u)>>16)|((v8u)(e>112u)&((((e-112u)<<11)&0x7800u)|(m>>12)))|((v8u)((e<113u)&(e>100u))&((((0x007FF800u+m)>>sh)+1u)>>1));
}
typedef struct {
	const LuwOracleCfg* c; uint64_t N, t;
	void* fi; float* rho; float* u; const uint8_t* flags; const float* F; void* gi; float* T;
	uint32_t Nxg, Nyg, Nzg; int west_x, east_x, south_y, north_y, top_z, has_w, has_e, has_s, has_n, has_t;
	float* wbuf; float* sigma;   /* w_buf of FX/kernel.cpp:1581-1583 per distance 0..Nbuf, sigma of :1604-1606 per layer 0..Nsponge-1 */
} FastCtx;
/* eight values of plane `plane` at row base `row` (element index of the row's x = 0), positions xi[l]; contig: xi[l] = xi[0]+l */
static inline v8f fast_load8(const FastCtx* k, const void* lat, const uint64_t plane, const uint64_t row, const uint32_t* xi, const int contig) {
	const uint64_t base = plane*k->N+row;
	if(k->c->fp16c) {
		const uint16_t* p = (const uint16_t*)lat+base;
		v8u raw;
		if(contig) raw = (v8u)_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)(p+xi[0])));
		else raw = (v8u){ p[xi[0]], p[xi[1]], p[xi[2]], p[xi[3]], p[xi[4]], p[xi[5]], p[xi[6]], p[xi[7]] };
		return v_half_to_float(raw);
	} else {
		const float* p = (const float*)lat+base;
		if(contig) return (v8f)_mm256_loadu_ps(p+xi[0]);
		return (v8f){ p[xi[0]], p[xi[1]], p[xi[2]], p[xi[3]], p[xi[4]], p[xi[5]], p[xi[6]], p[xi[7]] };
	}
}
static inline void fast_store8(const FastCtx* k, void* lat, const uint64_t plane, const uint64_t row, const uint32_t* xi, const int contig_all, const uint8_t* on,
		const v8f v) {
	const uint64_t base = plane*k->N+row;
	if(k->c->fp16c) {
		uint16_t* p = (uint16_t*)lat+base;
		const v8u code = v_float_to_half(v);
		if(contig_all) { const __m256i w = (__m256i)code; _mm_storeu_si128((__m128i*)(p+xi[0]), _mm_packus_epi32(_mm256_castsi256_si128(w), _mm256_extracti128_si256(w, 1))); }
		else for(int l=0; l<8; l++) if(on[l]) p[xi[l]] = (uint16_t)code[l];
	} else {
		float* p = (float*)lat+base;
		if(contig_all) _mm256_storeu_ps(p+xi[0], (__m256)v);
		else for(int l=0; l<8; l++) if(on[l]) p[xi[l]] = v[l];
	}
}
static void fast_row(const FastCtx* k, const uint32_t y, const uint32_t z) {
	const LuwOracleCfg* c = k->c;
	const uint64_t N = k->N, Nx = c->Nx, NxNy = Nx*(uint64_t)c->Ny;
	const int odd = (int)(k->t%2ull);
	/* rows of the cell and of its +c_i neighbours, i = 1,3,...,17 (neighbors(): j[i] for odd i) */
	const uint64_t y0 = (uint64_t)y*Nx, yp = (uint64_t)((y+1u)%c->Ny)*Nx, ym = (uint64_t)((y+c->Ny-1u)%c->Ny)*Nx;
	const uint64_t z0 = (uint64_t)z*NxNy, zp = (uint64_t)((z+1u)%c->Nz)*NxNy, zm = (uint64_t)((z+c->Nz-1u)%c->Nz)*NxNy;
	const uint64_t r00 = y0+z0;
	const uint64_t rowj[9] = { y0+z0, yp+z0, y0+zp, yp+z0, y0+zp, yp+zp, ym+z0, y0+zm, yp+zm };
	static const int shifted[9] = { 1, 0, 0, 1, 1, 0, 1, 1, 0 };
	const int yg = (int)y+c->Oy, zg = (int)z+c->Oz;
	/* the parts of the zone conditions (FX/kernel.cpp:1537-1541,1598) that depend on the row alone */
	const int Nbuf_i = (int)c->buffer_N, Nsponge_i = (int)c->sponge_N;
	const int d_s_i = yg, d_n_i = (int)(k->Nyg-1u)-yg, d_t_i = (int)(k->Nzg-1u)-zg;
	const int in_s = c->buffer_active&&c->downstream_face!=3&&k->has_s==1&&d_s_i>=0&&d_s_i<=Nbuf_i;
	const int in_n = c->buffer_active&&c->downstream_face!=4&&k->has_n==1&&d_n_i>=0&&d_n_i<=Nbuf_i;
	const int in_t = c->buffer_active&&k->has_t==1&&d_t_i>=0&&d_t_i<=Nbuf_i;
	const int d_sp_i = (int)(k->Nzg-2u)-zg;
	const int in_sp = c->sponge_active&&k->has_t==1&&d_sp_i>=0&&d_sp_i<Nsponge_i;
	const v8f zero = vsplat(0.0f);
	for(uint32_t x0=0u; x0<c->Nx; x0+=8u) {
		const uint32_t cnt = c->Nx-x0<8u ? c->Nx-x0 : 8u;
		uint32_t xn[8], xp[8]; uint8_t on[8], fl[8];
		int any = 0, all = cnt==8u;
		for(uint32_t l=0u; l<8u; l++) {
			const uint32_t x = l<cnt ? x0+l : x0;               /* lanes behind the row end repeat its chunk's first cell and are switched off */
			xn[l] = x; xp[l] = (x+1u)%c->Nx;
			fl[l] = k->flags[r00+x];
			const int halo = (c->Dx>1u)&&(x==0u||x>=c->Nx-1u);
			on[l] = l<cnt && !halo && (fl[l]&TYPE_BO)!=TYPE_S && (fl[l]&TYPE_SU)!=TYPE_G;
			any |= on[l]; all &= on[l];
		}
		if(!any) continue;
		const int contig_n = cnt==8u, contig_p = cnt==8u&&x0+8u<c->Nx;   /* x+1 positions are consecutive unless the chunk holds the row end */
		v8i m_on, m_E, m_T;
		for(int l=0; l<8; l++) { m_on[l] = on[l] ? -1 : 0; m_E[l] = (fl[l]&TYPE_BO)==TYPE_E ? -1 : 0; m_T[l] = (fl[l]&TYPE_T) ? -1 : 0; }
		/* load_f, FX/kernel.cpp:1338-1344 */
		v8f fhn[19];
		fhn[0] = fast_load8(k, k->fi, 0ull, r00, xn, contig_n);
		for(uint32_t i=1u, q=0u; i<19u; i+=2u, q++) {
			fhn[i   ] = fast_load8(k, k->fi, (uint64_t)(odd ? i    : i+1u), r00, xn, contig_n);
			fhn[i+1u] = fast_load8(k, k->fi, (uint64_t)(odd ? i+1u : i   ), rowj[q], shifted[q] ? xp : xn, shifted[q] ? contig_p : contig_n);
		}
		for(uint32_t i=0u; i<19u; i++) fhn[i] = vsel(m_on, fhn[i], zero);
		/* calculate_rho_u (FX/kernel.cpp:1075-1100) or, on TYPE_E cells, the fields */
		v8f rhon, uxn, uyn, uzn;
		{
			v8f rho_ = fhn[0];
			for(uint32_t i=1u; i<19u; i++) rho_ += fhn[i];
			rho_ += 1.0f;
			const v8f ux = fhn[ 1]-fhn[ 2]+fhn[ 7]-fhn[ 8]+fhn[ 9]-fhn[10]+fhn[13]-fhn[14]+fhn[15]-fhn[16];
			const v8f uy = fhn[ 3]-fhn[ 4]+fhn[ 7]-fhn[ 8]+fhn[11]-fhn[12]+fhn[14]-fhn[13]+fhn[17]-fhn[18];
			const v8f uz = fhn[ 5]-fhn[ 6]+fhn[ 9]-fhn[10]+fhn[11]-fhn[12]+fhn[16]-fhn[15]+fhn[18]-fhn[17];
			rhon = rho_; uxn = ux/rho_; uyn = uy/rho_; uzn = uz/rho_;
		}
		v8f rho_f, ux_f, uy_f, uz_f;   /* rho[n], u[n] of the chunk's cells */
		for(int l=0; l<8; l++) { const uint64_t n = r00+xn[l]; rho_f[l] = k->rho[n]; ux_f[l] = k->u[n]; uy_f[l] = k->u[N+n]; uz_f[l] = k->u[2ull*N+n]; }
		rhon = vsel(m_E, rho_f, rhon); uxn = vsel(m_E, ux_f, uxn); uyn = vsel(m_E, uy_f, uyn); uzn = vsel(m_E, uz_f, uzn);
		v8f fxn = vsplat(c->fx), fyn = vsplat(c->fy), fzn = vsplat(c->fz);
		{
			const v8f cor_x = -2.0f*rhon*(c->omega_y*uzn-c->omega_z*uyn);
			const v8f cor_y = -2.0f*rhon*(c->omega_z*uxn-c->omega_x*uzn);
			const v8f cor_z = -2.0f*rhon*(c->omega_x*uyn-c->omega_y*uxn);
			fxn += cor_x; fyn += cor_y; fzn += cor_z;
		}
		if(c->buffer_active) { /* FX/kernel.cpp:1523-1595: the selection per cell as written there, the arithmetic for the chunk */
			v8i m_z; v8f w_buf = zero, tx = zero, ty = zero, tz = zero;
			int any_z = 0;
			for(int l=0; l<8; l++) {
				m_z[l] = 0;
				if((fl[l]&TYPE_BO)==TYPE_E||!on[l]) continue;
				const uint32_t x = xn[l];
				const int xg = (int)x+c->Ox;
				const int d_w_i = xg, d_e_i = (int)(k->Nxg-1u)-xg;
				const int in_w = c->downstream_face!=1&&k->has_w==1&&d_w_i>=0&&d_w_i<=Nbuf_i;
				const int in_e = c->downstream_face!=2&&k->has_e==1&&d_e_i>=0&&d_e_i<=Nbuf_i;
				if(in_w||in_e||in_s||in_n||in_t) {
					uint32_t d_min = c->buffer_N+1u;
					uint64_t n_ref = r00+x;
					if(in_w) { const uint32_t d = (uint32_t)d_w_i; if(d<d_min) { d_min = d; n_ref = index3(c, (uint32_t)k->west_x, y, z); } }
					if(in_e) { const uint32_t d = (uint32_t)d_e_i; if(d<d_min) { d_min = d; n_ref = index3(c, (uint32_t)k->east_x, y, z); } }
					if(in_s) { const uint32_t d = (uint32_t)d_s_i; if(d<d_min) { d_min = d; n_ref = index3(c, x, (uint32_t)k->south_y, z); } }
					if(in_n) { const uint32_t d = (uint32_t)d_n_i; if(d<d_min) { d_min = d; n_ref = index3(c, x, (uint32_t)k->north_y, z); } }
					if(in_t) { const uint32_t d = (uint32_t)d_t_i; if(d<d_min) { d_min = d; n_ref = index3(c, x, y, (uint32_t)k->top_z); } }
					m_z[l] = -1; any_z = 1;
					w_buf[l] = k->wbuf[d_min];
					tx[l] = k->u[n_ref]; ty[l] = k->u[N+n_ref]; tz[l] = k->u[2ull*N+n_ref];
				}
			}
			if(any_z) {
				const v8f a_x = w_buf*c->buffer_inv_tau*(tx-uxn);
				const v8f a_y = w_buf*c->buffer_inv_tau*(ty-uyn);
				const v8f a_z = c->buffer_nudge_vertical==1 ? w_buf*c->buffer_inv_tau*(tz-uzn) : zero;
				fxn = vsel(m_z, fxn+rhon*a_x, fxn);
				fyn = vsel(m_z, fyn+rhon*a_y, fyn);
				fzn = vsel(m_z, fzn+rhon*a_z, fzn);
			}
		}
		v8i m_sp;   /* the top sponge acts on this lane (FX/kernel.cpp:1596-1614) */
		for(int l=0; l<8; l++) m_sp[l] = (in_sp&&on[l]&&(fl[l]&TYPE_BO)!=TYPE_E) ? -1 : 0;
		const uint64_t row_top = in_sp ? index3(c, 0u, y, (uint32_t)k->top_z) : 0ull;
		if(in_sp) {
			const v8f sigma = vsplat(k->sigma[d_sp_i]);
			v8f sx, sy, sz;
			for(int l=0; l<8; l++) { const uint64_t n_ref = row_top+xn[l]; sx[l] = k->u[n_ref]; sy[l] = k->u[N+n_ref]; sz[l] = k->u[2ull*N+n_ref]; }
			fxn = vsel(m_sp, fxn+rhon*sigma*(sx-uxn), fxn);
			fyn = vsel(m_sp, fyn+rhon*sigma*(sy-uyn), fyn);
			fzn = vsel(m_sp, fzn+rhon*sigma*(sz-uzn), fzn);
		}
		if(k->gi) { /* TEMPERATURE, FX/kernel.cpp:1639-1684 */
			v8f ghn[7];
			ghn[0] = fast_load8(k, k->gi, 0ull, r00, xn, contig_n);
			for(uint32_t i=1u, q=0u; i<7u; i+=2u, q++) {
				ghn[i   ] = fast_load8(k, k->gi, (uint64_t)(odd ? i    : i+1u), r00, xn, contig_n);
				ghn[i+1u] = fast_load8(k, k->gi, (uint64_t)(odd ? i+1u : i   ), rowj[q], shifted[q] ? xp : xn, shifted[q] ? contig_p : contig_n);
			}
			v8f T_f;
			for(int l=0; l<8; l++) T_f[l] = k->T[r00+xn[l]];
			v8f Tn = zero;
			for(uint32_t i=0u; i<7u; i++) Tn += ghn[i];
			Tn += 1.0f;
			Tn = vsel(m_T, T_f, Tn);
			if(in_sp) {
				const v8f sigma_T = vsplat(k->sigma[d_sp_i]);
				v8f T_top;
				for(int l=0; l<8; l++) T_top[l] = k->T[row_top+xn[l]];
				Tn = vsel(m_sp&~m_T, vfma(sigma_T, T_top-Tn, Tn), Tn);
			}
			v8f geq[7];
			{
				const v8f wsT4 = 0.5f*Tn, wsTm1 = 0.125f*(Tn-1.0f);
				geq[0] = vfma(vsplat(0.25f), Tn, vsplat(-0.25f));
				geq[1] = vfma(wsT4, uxn, wsTm1); geq[2] = vfma(wsT4, -uxn, wsTm1);
				geq[3] = vfma(wsT4, uyn, wsTm1); geq[4] = vfma(wsT4, -uyn, wsTm1);
				geq[5] = vfma(wsT4, uzn, wsTm1); geq[6] = vfma(wsT4, -uzn, wsTm1);
			}
			for(int l=0; l<8; l++) if(on[l]&&!(fl[l]&TYPE_T)) k->T[r00+xn[l]] = Tn[l]; /* UPDATE_FIELDS */
			for(uint32_t i=0u; i<7u; i++) ghn[i] = vsel(m_T, geq[i], vfma(vsplat(1.0f-c->w_T), ghn[i], c->w_T*geq[i]));
			fast_store8(k, k->gi, 0ull, r00, xn, all&&contig_n, on, ghn[0]);
			for(uint32_t i=1u, q=0u; i<7u; i+=2u, q++) {
				fast_store8(k, k->gi, (uint64_t)(odd ? i+1u : i   ), rowj[q], shifted[q] ? xp : xn, all&&(shifted[q] ? contig_p : contig_n), on, ghn[i   ]);
				fast_store8(k, k->gi, (uint64_t)(odd ? i    : i+1u), r00, xn, all&&contig_n, on, ghn[i+1u]);
			}
		}
		if(k->F) { /* FORCE_FIELD */
			v8f Fx, Fy, Fz;
			for(int l=0; l<8; l++) { const uint64_t n = r00+xn[l]; Fx[l] = k->F[n]; Fy[l] = k->F[N+n]; Fz[l] = k->F[2ull*N+n]; }
			fxn += Fx; fyn += Fy; fzn += Fz;
		}
		v8f Fin[19];
		{ /* VOLUME_FORCE, FX/kernel.cpp:1686-1692 + calculate_forcing_terms :1103-1113 */
			const v8f rho2 = 0.5f/rhon;
			uxn = vclamp(vfma(fxn, rho2, uxn), -DEF_C, DEF_C);
			uyn = vclamp(vfma(fyn, rho2, uyn), -DEF_C, DEF_C);
			uzn = vclamp(vfma(fzn, rho2, uzn), -DEF_C, DEF_C);
			const v8f uF = -0.33333334f*vfma(uxn, fxn, vfma(uyn, fyn, uzn*fzn));
			Fin[0] = 9.0f*DEF_W0*uF;
			for(uint32_t i=1u; i<19u; i++) Fin[i] = 9.0f*wi(i)*vfma(CX[i]*fxn+CY[i]*fyn+CZ[i]*fzn, CX[i]*uxn+CY[i]*uyn+CZ[i]*uzn+0.33333334f, uF);
		}
		for(int l=0; l<8; l++) if(on[l]&&(fl[l]&TYPE_BO)!=TYPE_E) { /* UPDATE_FIELDS, FX/kernel.cpp:1709-1716 */
			const uint64_t n = r00+xn[l];
			k->rho[n] = rhon[l]; k->u[n] = uxn[l]; k->u[N+n] = uyn[l]; k->u[2ull*N+n] = uzn[l];
		}
		v8f feq[19];
		{ /* luwo_calculate_f_eq */
			const v8f rhom1 = rhon-1.0f;
			const v8f c3 = -3.0f*(vsq(uxn)+vsq(uyn)+vsq(uzn));
			const v8f uz = uzn*3.0f, ux = uxn*3.0f, uy = uyn*3.0f;
			const v8f h = vsplat(0.5f);
			feq[ 0] = DEF_W0*vfma(rhon, 0.5f*c3, rhom1);
			const v8f u0=ux+uy, u1=ux+uz, u2=uy+uz, u3=ux-uy, u4=ux-uz, u5=uy-uz;
			const v8f rhos=DEF_WS*rhon, rhoe=DEF_WE*rhon, rhom1s=DEF_WS*rhom1, rhom1e=DEF_WE*rhom1;
			feq[ 1] = vfma(rhos, vfma(h, vfma(ux, ux, c3), ux), rhom1s); feq[ 2] = vfma(rhos, vfma(h, vfma(ux, ux, c3), -ux), rhom1s);
			feq[ 3] = vfma(rhos, vfma(h, vfma(uy, uy, c3), uy), rhom1s); feq[ 4] = vfma(rhos, vfma(h, vfma(uy, uy, c3), -uy), rhom1s);
			feq[ 5] = vfma(rhos, vfma(h, vfma(uz, uz, c3), uz), rhom1s); feq[ 6] = vfma(rhos, vfma(h, vfma(uz, uz, c3), -uz), rhom1s);
			feq[ 7] = vfma(rhoe, vfma(h, vfma(u0, u0, c3), u0), rhom1e); feq[ 8] = vfma(rhoe, vfma(h, vfma(u0, u0, c3), -u0), rhom1e);
			feq[ 9] = vfma(rhoe, vfma(h, vfma(u1, u1, c3), u1), rhom1e); feq[10] = vfma(rhoe, vfma(h, vfma(u1, u1, c3), -u1), rhom1e);
			feq[11] = vfma(rhoe, vfma(h, vfma(u2, u2, c3), u2), rhom1e); feq[12] = vfma(rhoe, vfma(h, vfma(u2, u2, c3), -u2), rhom1e);
			feq[13] = vfma(rhoe, vfma(h, vfma(u3, u3, c3), u3), rhom1e); feq[14] = vfma(rhoe, vfma(h, vfma(u3, u3, c3), -u3), rhom1e);
			feq[15] = vfma(rhoe, vfma(h, vfma(u4, u4, c3), u4), rhom1e); feq[16] = vfma(rhoe, vfma(h, vfma(u4, u4, c3), -u4), rhom1e);
			feq[17] = vfma(rhoe, vfma(h, vfma(u5, u5, c3), u5), rhom1e); feq[18] = vfma(rhoe, vfma(h, vfma(u5, u5, c3), -u5), rhom1e);
		}
		v8f w = vsplat(c->w);
		if(c->subgrid) { /* FX/kernel.cpp:1723-1737 */
			const float tau0 = 1.0f/c->w;
			v8f Hxx=zero, Hyy=zero, Hzz=zero, Hxy=zero, Hxz=zero, Hyz=zero;
			for(uint32_t i=1u; i<19u; i++) {
				const v8f fneqi = fhn[i]-feq[i];
				const float cxi=CX[i], cyi=CY[i], czi=CZ[i];
				Hxx += cxi*cxi*fneqi;
				Hxy += cxi*cyi*fneqi; Hyy += cyi*cyi*fneqi;
				Hxz += cxi*czi*fneqi; Hyz += cyi*czi*fneqi; Hzz += czi*czi*fneqi;
			}
			const v8f Q = vsq(Hxx)+vsq(Hyy)+vsq(Hzz)+2.0f*(vsq(Hxy)+vsq(Hxz)+vsq(Hyz));
			w = 2.0f/(tau0+vsqrt(sq(tau0)+0.76421222f*vsqrt(Q)/rhon));
		}
		const v8f c_tau = vfma(w, vsplat(-0.5f), vsplat(1.0f)); /* FX/kernel.cpp:1741-1742 */
		for(uint32_t i=0u; i<19u; i++) Fin[i] *= c_tau;
		for(uint32_t i=0u; i<19u; i++) fhn[i] = vsel(m_E, feq[i], vfma(1.0f-w, fhn[i], vfma(w, feq[i], Fin[i]))); /* FX/kernel.cpp:1747 */
		/* store_f, FX/kernel.cpp:1345-1351 */
		fast_store8(k, k->fi, 0ull, r00, xn, all&&contig_n, on, fhn[0]);
		for(uint32_t i=1u, q=0u; i<19u; i+=2u, q++) {
			fast_store8(k, k->fi, (uint64_t)(odd ? i+1u : i   ), rowj[q], shifted[q] ? xp : xn, all&&(shifted[q] ? contig_p : contig_n), on, fhn[i   ]);
			fast_store8(k, k->fi, (uint64_t)(odd ? i    : i+1u), r00, xn, all&&contig_n, on, fhn[i+1u]);
		}
	}
}
static void stream_collide_rows(const LuwOracleCfg* c, void* fi, float* rho, float* u, const uint8_t* flags, const float* F, void* gi, float* T, const uint64_t t) {
	FastCtx k;
	memset(&k, 0, sizeof(k));
	k.c = c; k.N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz; k.t = t;
	k.fi = fi; k.rho = rho; k.u = u; k.flags = flags; k.F = F; k.gi = gi; k.T = T;
	/* derived face constants, FX/lbm.cpp:613-625 (stream_collide_cell derives the same per cell) */
	k.Nxg = (c->Nx-2u*(c->Dx>1u))*c->Dx; k.Nyg = (c->Ny-2u*(c->Dy>1u))*c->Dy; k.Nzg = (c->Nz-2u*(c->Dz>1u))*c->Dz;
	k.west_x = -c->Ox; k.east_x = (int)k.Nxg-1-c->Ox; k.south_y = -c->Oy; k.north_y = (int)k.Nyg-1-c->Oy; k.top_z = (int)k.Nzg-1-c->Oz;
	k.has_w = k.west_x>=0&&k.west_x<(int)c->Nx ? 1 : 0; k.has_e = k.east_x>=0&&k.east_x<(int)c->Nx ? 1 : 0;
	k.has_s = k.south_y>=0&&k.south_y<(int)c->Ny ? 1 : 0; k.has_n = k.north_y>=0&&k.north_y<(int)c->Ny ? 1 : 0;
	k.has_t = k.top_z>=0&&k.top_z<(int)c->Nz ? 1 : 0;
	if(c->buffer_active) { /* w_buf per distance, FX/kernel.cpp:1581-1583 */
		k.wbuf = (float*)malloc(((size_t)c->buffer_N+2u)*sizeof(float));
		for(uint32_t d=0u; d<=c->buffer_N+1u; d++) {
			const float xi = 1.0f-(float)d/(float)c->buffer_N;
			float w_buf = sinf(1.5707963267948966f*xi);
			w_buf *= w_buf;
			k.wbuf[d] = w_buf;
		}
	}
	if(c->sponge_active) { /* sigma per layer, FX/kernel.cpp:1604-1606 */
		const int Nsponge_i = (int)c->sponge_N;
		k.sigma = (float*)malloc(((size_t)c->sponge_N+1u)*sizeof(float));
		for(int d=0; d<Nsponge_i; d++) {
			const float xi = Nsponge_i>1 ? 1.0f-(float)d/(float)(Nsponge_i-1) : 1.0f;
			float sigma = sinf(1.5707963267948966f*xi);
			sigma = c->sponge_inv_tau*sigma*sigma;
			k.sigma[d] = sigma;
		}
	}
	const int64_t rows = (int64_t)c->Ny*(int64_t)c->Nz;
	#pragma omp parallel for schedule(static)
	for(int64_t r=0; r<rows; r++) {
		const uint32_t y = (uint32_t)(r%(int64_t)c->Ny), z = (uint32_t)(r/(int64_t)c->Ny);
		if(((c->Dy>1u)&&(y==0u||y>=c->Ny-1u))||((c->Dz>1u)&&(z==0u||z>=c->Nz-1u))) continue; /* halo rows, FX/kernel.cpp:856-859 */
		fast_row(&k, y, z);
	}
	free(k.wbuf); free(k.sigma);
}
#else
#define LUWO_HAVE_FAST 0
#endif
static int g_fast = 1;
void luwo_set_fast(const int on) { g_fast = on; }
int luwo_get_fast(void) { return g_fast&&LUWO_HAVE_FAST; }

void luwo_stream_collide_thermal(const LuwOracleCfg* c, void* fi, float* rho, float* u, const uint8_t* flags, const float* F, void* gi, float* T, const uint64_t t) {
#if LUWO_HAVE_FAST
	if(g_fast) { stream_collide_rows(c, fi, rho, u, flags, F, gi, T, t); return; }
#endif
	luwo_stream_collide_literal(c, fi, rho, u, flags, F, gi, T, t);
}
void luwo_stream_collide(const LuwOracleCfg* c, void* fi, float* rho, float* u, const uint8_t* flags, const float* F, const uint64_t t) {
	luwo_stream_collide_thermal(c, fi, rho, u, flags, F, NULL, NULL, t);
}

/* LBM::run for a single domain: FX/lbm.cpp:1262-1312 (t increments after each stream_collide) */
void luwo_run(const LuwOracleCfg* c, void* fi, float* rho, float* u, const uint8_t* flags, const float* F, const uint64_t t0, const uint64_t steps) {
	for(uint64_t t=t0; t<t0+steps; t++) luwo_stream_collide(c, fi, rho, u, flags, F, t);
}

/* ------------------------------------------------------------------ halo transfer, FX/kernel.cpp:2188-2270 */
static const uint8_t INDEX_TRANSFER[30] = { /* FX/kernel.cpp:2223-2229 */
	1,  7, 13,  9, 15,
	2,  8, 14, 10, 16,
	3,  7, 14, 11, 17,
	4,  8, 13, 12, 18,
	5,  9, 16, 11, 18,
	6, 10, 15, 12, 17
};
uint64_t luwo_get_area(const LuwOracleCfg* c, const uint32_t direction) {
	const uint64_t A[3] = { (uint64_t)c->Ny*c->Nz, (uint64_t)c->Nz*c->Nx, (uint64_t)c->Nx*c->Ny };
	return A[direction];
}
static uint64_t index_face(const LuwOracleCfg* c, const uint32_t a, const uint32_t direction, const uint32_t fixed) { /* FX/kernel.cpp:2192-2207 */
	if(direction==0u) return index3(c, fixed, a%c->Ny, a/c->Ny);
	if(direction==1u) return index3(c, a/c->Nz, fixed, a%c->Nz);
	return index3(c, a%c->Nx, a/c->Nx, fixed);
}
static inline uint32_t Ndir(const LuwOracleCfg* c, const uint32_t direction) { return direction==0u ? c->Nx : direction==1u ? c->Ny : c->Nz; }

static void copy_ddf(const LuwOracleCfg* c, void* dst, const uint64_t di, const void* src, const uint64_t si) { /* fpxx_copy: raw bits */
	if(c->fp16c) ((uint16_t*)dst)[di] = ((const uint16_t*)src)[si]; else ((float*)dst)[di] = ((const float*)src)[si];
}
void luwo_transfer_extract_fi(const LuwOracleCfg* c, const uint32_t direction, const uint64_t t, void* buf_p, void* buf_m, const void* fi) { /* FX/kernel.cpp:2241-2249,2259-2264 */
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	const uint64_t A = luwo_get_area(c, direction);
	for(uint64_t a=0; a<A; a++) {
		for(uint32_t pm=0u; pm<2u; pm++) {
			const uint64_t n = index_face(c, (uint32_t)a, direction, pm==0u ? Ndir(c, direction)-2u : 1u);
			const uint32_t side = 2u*direction+pm;
			uint64_t j[19]; neighbors(c, n, j);
			for(uint32_t b=0u; b<5u; b++) {
				const uint32_t i = INDEX_TRANSFER[side*5u+b];
				const uint64_t idx = (uint64_t)(t%2ull ? (i%2u ? i+1u : i-1u) : i)*N+(i%2u ? j[i] : n);
				copy_ddf(c, pm==0u ? buf_p : buf_m, b*A+a, fi, idx);
			}
		}
	}
}
void luwo_transfer_insert_fi(const LuwOracleCfg* c, const uint32_t direction, const uint64_t t, const void* buf_p, const void* buf_m, void* fi) { /* FX/kernel.cpp:2250-2258,2265-2270 */
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	const uint64_t A = luwo_get_area(c, direction);
	for(uint64_t a=0; a<A; a++) {
		for(uint32_t pm=0u; pm<2u; pm++) {
			const uint64_t n = index_face(c, (uint32_t)a, direction, pm==0u ? Ndir(c, direction)-1u : 0u);
			const uint32_t side = 2u*direction+pm;
			uint64_t j[19]; neighbors(c, n, j);
			for(uint32_t b=0u; b<5u; b++) {
				const uint32_t i = INDEX_TRANSFER[side*5u+b];
				const uint64_t idx = (uint64_t)(t%2ull ? i : (i%2u ? i+1u : i-1u))*N+(i%2u ? n : j[i-1u]);
				copy_ddf(c, fi, idx, pm==0u ? buf_p : buf_m, b*A+a);
			}
		}
	}
}

/* thermal lattice: one population per face cell and side, i = side+1 (FX/kernel.cpp:2338-2363) */
void luwo_transfer_extract_gi(const LuwOracleCfg* c, const uint32_t direction, const uint64_t t, void* buf_p, void* buf_m, const void* gi) {
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	const uint64_t A = luwo_get_area(c, direction);
	for(uint64_t a=0; a<A; a++) for(uint32_t pm=0u; pm<2u; pm++) {
		const uint64_t n = index_face(c, (uint32_t)a, direction, pm==0u ? Ndir(c, direction)-2u : 1u);
		uint64_t j[19]; neighbors(c, n, j);
		const uint32_t i = 2u*direction+pm+1u;
		copy_ddf(c, pm==0u ? buf_p : buf_m, a, gi, (uint64_t)(t%2ull ? (i%2u ? i+1u : i-1u) : i)*N+(i%2u ? j[i] : n));
	}
}
void luwo_transfer_insert_gi(const LuwOracleCfg* c, const uint32_t direction, const uint64_t t, const void* buf_p, const void* buf_m, void* gi) {
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	const uint64_t A = luwo_get_area(c, direction);
	for(uint64_t a=0; a<A; a++) for(uint32_t pm=0u; pm<2u; pm++) {
		const uint64_t n = index_face(c, (uint32_t)a, direction, pm==0u ? Ndir(c, direction)-1u : 0u);
		uint64_t j[19]; neighbors(c, n, j);
		const uint32_t i = 2u*direction+pm+1u;
		copy_ddf(c, gi, (uint64_t)(t%2ull ? i : (i%2u ? i+1u : i-1u))*N+(i%2u ? n : j[i-1u]), pm==0u ? buf_p : buf_m, a);
	}
}

/* vk_inlet_apply, FX/kernel.cpp:2495-2571 (u in the reference layout, plane stride N) */
void luwo_vk_inlet_apply(const uint64_t N, const uint32_t use_interp, const float t0, const float t1, const float alpha, const uint64_t P, const uint64_t M,
		const uint64_t* point_cell, const uint8_t* point_face, const float* point_data, const float* mode_data, float* u) {
	const uint64_t V = 5ull*M;
	#pragma omp parallel for schedule(static)
	for(int64_t ii=0; ii<(int64_t)P; ii++) {
		const uint64_t i = (uint64_t)ii, n = point_cell[i];
		const uint64_t fid = (uint64_t)(point_face[i]&0x07u);
		const float px = point_data[i], py = point_data[P+i], pz = point_data[2ull*P+i];
		const float ubx = point_data[3ull*P+i], uby = point_data[4ull*P+i], ubz = point_data[5ull*P+i], sigma = point_data[6ull*P+i];
		if(fid>=5ull||!(sigma>0.0f)) { u[n] = ubx; u[N+n] = uby; u[2ull*N+n] = ubz; continue; }
		const uint64_t fbase = fid*M;
		float qx = 0.0f, qy = 0.0f, qz = 0.0f;
		for(uint64_t m=0ull; m<M; ++m) {
			const uint64_t idx = fbase+m;
			const float kx = mode_data[idx], ky = mode_data[V+idx], kz = mode_data[2ull*V+idx], omega = mode_data[3ull*V+idx];
			const float Ax = mode_data[4ull*V+idx], Ay = mode_data[5ull*V+idx], Az = mode_data[6ull*V+idx];
			const float phix = mode_data[7ull*V+idx], phiy = mode_data[8ull*V+idx], phiz = mode_data[9ull*V+idx];
			const float phase0 = fmaf(kx, px, fmaf(ky, py, fmaf(kz, pz, omega*t0)));
			float vx = Ax*cosf(phase0+phix), vy = Ay*cosf(phase0+phiy), vz = Az*cosf(phase0+phiz);
			if(use_interp!=0u) {
				const float phase1 = fmaf(kx, px, fmaf(ky, py, fmaf(kz, pz, omega*t1)));
				const float vx1 = Ax*cosf(phase1+phix), vy1 = Ay*cosf(phase1+phiy), vz1 = Az*cosf(phase1+phiz);
				vx = fmaf(alpha, vx1-vx, vx); vy = fmaf(alpha, vy1-vy, vy); vz = fmaf(alpha, vz1-vz, vz);
			}
			qx += vx; qy += vy; qz += vz;
		}
		u[n] = fmaf(sigma, qx, ubx);
		u[N+n] = fmaf(sigma, qy, uby);
		u[2ull*N+n] = fmaf(sigma, qz, ubz);
	}
}

/* accumulate_from_buffers, FX/setup.cpp:4441-4488: Welford mean / M2 of u (avg_u is AoS [3n+c]), running mean of rho.
 * count is the sample number AFTER the increment (avg_count), inv_n = 1/count. */
void luwo_accumulate_stats(const uint64_t N, const uint64_t count, const float* rho, const float* u, float* avg_u, float* avg_rho, float* m2_u, float* m2_v, float* m2_w) {
	const float inv_n = 1.0f/(float)count;
	#pragma omp parallel for schedule(static)
	for(int64_t nn=0; nn<(int64_t)N; nn++) {
		const uint64_t n = (uint64_t)nn, i3 = 3ull*n;
		const float ux = u[n], uy = u[N+n], uz = u[2ull*N+n];
		float mean_u = avg_u[i3], mean_v = avg_u[i3+1ull], mean_w = avg_u[i3+2ull];
		const float delta_u = ux-mean_u;
		mean_u += delta_u*inv_n;
		const float delta2_u = ux-mean_u;
		m2_u[n] += delta_u*delta2_u;
		avg_u[i3] = mean_u;
		const float delta_v = uy-mean_v;
		mean_v += delta_v*inv_n;
		const float delta2_v = uy-mean_v;
		m2_v[n] += delta_v*delta2_v;
		avg_u[i3+1ull] = mean_v;
		const float delta_w = uz-mean_w;
		mean_w += delta_w*inv_n;
		const float delta2_w = uz-mean_w;
		m2_w[n] += delta_w*delta2_w;
		avg_u[i3+2ull] = mean_w;
		const float r = rho[n];
		avg_rho[n] += (r-avg_rho[n])*inv_n;
	}
}

/* OpenMP thread control for the cpu_baseline leg of bench.py (no-ops when built without OpenMP) */
#ifdef _OPENMP
#include <omp.h>
void luwo_set_threads(const int n) { if(n>0) omp_set_num_threads(n); }
int luwo_get_max_threads(void) { return omp_get_max_threads(); }
#else
void luwo_set_threads(const int n) { (void)n; }
int luwo_get_max_threads(void) { return 1; }
#endif

/* host copy bandwidth (GB/s, read + write counted) with the OpenMP threads in force: the denominator bench.py's cpu_baseline
 * reports the restatement's DRAM traffic against.  Buffers are touched by the threads that later copy them (first touch). */
#include <time.h>
double luwo_copy_bandwidth_gbps(const uint64_t bytes) {
	const uint64_t n = bytes/8u;
	double* a = (double*)malloc(n*8u); double* b = (double*)malloc(n*8u);
	if(!a||!b) { free(a); free(b); return 0.0; }
	#pragma omp parallel for schedule(static)
	for(int64_t i=0; i<(int64_t)n; i++) { a[i] = (double)i; b[i] = 0.0; }
	double best = 0.0;
	for(int rep=0; rep<4; rep++) {
		struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
		#pragma omp parallel for schedule(static)
		for(int64_t i=0; i<(int64_t)n; i++) b[i] = a[i];
		clock_gettime(CLOCK_MONOTONIC, &t1);
		const double dt = (double)(t1.tv_sec-t0.tv_sec)+1e-9*(double)(t1.tv_nsec-t0.tv_nsec);
		const double gbps = 2.0*(double)(n*8u)/dt*1e-9;
		if(gbps>best) best = gbps;
	}
	const double keep = b[n/2u];
	free(a); free(b);
	return keep<0.0 ? 0.0 : best;
}

/* update_fields equivalent for checks: rho,u of a cell straight from the stored DDFs at time t, no forcing.
 * (used by tests for conservation checks; mirrors load_f + calculate_rho_u) */
void luwo_moments(const LuwOracleCfg* c, const void* fi, const uint64_t t, float* rho_out, float* u_out) {
	const uint64_t N = (uint64_t)c->Nx*(uint64_t)c->Ny*(uint64_t)c->Nz;
	#pragma omp parallel for schedule(static)
	for(int64_t nn=0; nn<(int64_t)N; nn++) {
		const uint64_t n = (uint64_t)nn;
		uint64_t j[19]; neighbors(c, n, j);
		float fhn[19]; load_f(c, N, n, fhn, fi, j, t);
		float r, ux, uy, uz; luwo_calculate_rho_u(fhn, &r, &ux, &uy, &uz);
		rho_out[n] = r; u_out[n] = ux; u_out[N+n] = uy; u_out[2ull*N+n] = uz;
	}
}
