// valu_rates.hip -- issue cost of the instruction classes the scoring kernels are made of, on gfx950.
//
// VERDICT r3 (weak 3): bench.py priced EVERY VALU instruction at 4 cycles per wave64; MI355X_MICROARCH.md says a plain
// fp32 / integer instruction issues over 2.  This settles it per class, and measures what an MFMA-based screen could
// hide beside the matrix pipe: independent chains of ONE instruction (inline asm: the compiler neither folds nor
// reorders them), W waves per SIMD, cycles per wave-instruction per SIMD at the clock the chip actually ran at
// (s_memtime = shader cycles, s_memrealtime = 100 MHz: their ratio is the sustained clock of that test).
//
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/valu_rates.hip -o tools/ubench/valu_rates && tools/ubench/valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Stamp {
    unsigned long long cyc, real;
};

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP4(M) M(0) M(1) M(2) M(3)

enum Test {
    T_FMA_F32 = 0,
    T_PK_FMA_F32,
    T_PK_ADD_F32,
    T_ALIGNBIT,
    T_MIN3_F32,
    T_ADD_U32,
    T_BCNT,
    T_AND_OR,
    T_CNDMASK,
    T_CMP_F32_BCNT,      // v_cmp_lt_f32 -> sgpr pair, s_bcnt1_i32_b64, s_add_u32
    T_CMP_F32_ONLY,      // v_cmp_lt_f32 -> sgpr pair only
    T_CMP_ADDC,          // v_cmp_lt_f32 vcc + v_addc_co_u32 (count in the lane)
    T_SALU_BCNT,         // s_bcnt1 + s_add only
    T_MIX_PK_PLAIN,      // 8 v_pk_fma_f32 + 8 v_alignbit
    T_PLANE_LOOP,        // the plane screen's mix: 16 v_pk_fma + 8 v_alignbit + 4 v_min3 + 1 v_cmp
    T_FMA_F64,
    T_MUL_F64,
    T_ADD_F64,
    T_CMP_F64,
    T_PK_FMA_F16,
    T_CVT_F16,
    T_MFMA_F16_32,       // v_mfma_f32_32x32x16_f16, 4 independent accumulators
    T_MFMA_F16_16,       // v_mfma_f32_16x16x32_f16
    T_MFMA_F32_16,       // v_mfma_f32_16x16x4_f32
    T_MFMA32_VALU4,      // per v_mfma 32x32x16: + 4 plain VALU (v_fma_f32) in the same wave
    T_MFMA32_VALU8,
    T_MFMA32_VALU12,
    T_MFMA32_VALU16,
    T_MFMA32_VALU24,
    T_MFMA32_PK8,        // per MFMA: + 8 v_pk_fma_f32
    T_MFMA32_ALIGN16,    // per MFMA: + 16 v_alignbit (integer class beside the matrix pipe)
    T_MFMA32_POST,       // per MFMA: 16 v_alignbit ON ITS OUTPUT of the previous round + 8 v_min3 (the screen's post-processing)
    T_DS_READ_B128_BC,   // broadcast ds_read_b128 (all lanes one address)
    T_MIN_F32_E32,       // VOP2 (32-bit encoding) classes
    T_MUL_F32_E32,
    T_FMAC_F32_E32,
    T_SUB_F32_E32,
    T_AND_B32_E32,
    T_LSHR_B32_E32,
    T_MOV_B32,
    T_SUB_F32_SDWA_ABS,  // v_sub_f32_sdwa with |src0|
    T_CVT_PKRTZ,
    T_SCREEN2,           // the MFMA screen's trip: 2 MFMA + 16 v_min_f32 e32 + 16 v_alignbit + 8 v_min3 on the PREVIOUS accumulators
    T_SCREEN4,           // the same with 4 MFMA (K = 32: the cylinder's quadratic form)
    T_SCREEN1Q,          // 1 MFMA + 16 v_alignbit + 8 v_min3 (q straight out of the matrix pipe)
    T_COUNT
};

static const char* kNames[T_COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_alignbit_b32", "v_min3_f32 |abs|", "v_add_u32", "v_bcnt_u32_b32", "v_and_or_b32",
    "v_cndmask_b32", "v_cmp_lt_f32+s_bcnt1+s_add", "v_cmp_lt_f32 (sgpr dst)", "v_cmp_lt_f32+v_addc_co", "s_bcnt1+s_add (SALU only)",
    "mix 8 pk_fma + 8 alignbit", "plane loop mix (16 pk,8 ab,4 min3,1 cmp)", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cmp_lt_f64 (sgpr dst)",
    "v_pk_fma_f16", "v_cvt_f16_f32", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32",
    "mfma32x32x16 + 4 v_fma_f32", "mfma32x32x16 + 8 v_fma_f32", "mfma32x32x16 + 12 v_fma_f32", "mfma32x32x16 + 16 v_fma_f32",
    "mfma32x32x16 + 24 v_fma_f32", "mfma32x32x16 + 8 v_pk_fma_f32", "mfma32x32x16 + 16 v_alignbit", "mfma32x32x16 + post(16 ab + 8 min3 on acc)",
    "ds_read_b128 broadcast", "v_min_f32 e32", "v_mul_f32 e32", "v_fmac_f32 e32", "v_sub_f32 e32", "v_and_b32 e32", "v_lshrrev_b32 e32",
    "v_mov_b32 e32", "v_sub_f32_sdwa |abs|", "v_cvt_pkrtz_f16_f32", "screen trip: 2 mfma + 16 min e32 + 16 ab + 8 min3 (per 1024 pairs)",
    "screen trip: 4 mfma + 16 min e32 + 16 ab + 8 min3 (per 1024 pairs)", "screen trip: 1 mfma + 16 ab + 8 min3 (per 1024 pairs)"};
// wave-instructions counted per loop trip (what the cycles are divided by)
static const int kPerTrip[T_COUNT] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64 /* pairs */, 64, 64, 29 * 4, 64, 64, 64, 64, 64, 64,
                                      16, 16, 16, 4, 4, 4, 4, 4, 4, 4, 4, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 2, 2, 4};

template <int T>
__global__ __launch_bounds__(256) void k(Stamp* stamps, float* out, int iters, float seed) {
    __shared__ f32x4 lds[64];
    float r[16];
    f32x2 p[16];
    double d[8];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        r[i] = seed + (float)threadIdx.x * 1e-3f + (float)i;
        p[i] = f32x2{r[i], r[i] + 0.5f};
        u[i] = (uint32_t)threadIdx.x * 2654435761u + (uint32_t)i;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = (double)r[i];
    float c = 1.0000001f, e = 0.9999999f;
    f32x2 c2 = {c, c}, e2 = {e, e};
    double cd = 1.0000001, ed = 0.9999999;
    uint32_t sacc = 0;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
    f32x4 acc4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16x8 ha, hb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ha[j] = (_Float16)(0.001f * (float)(threadIdx.x + j));
        hb[j] = (_Float16)(0.002f * (float)(threadIdx.x - j));
    }
    if (threadIdx.x < 64) lds[threadIdx.x] = f32x4{seed, seed + 1, seed + 2, seed + 3};
    __syncthreads();
    asm volatile("" : "+v"(c), "+v"(e), "+v"(c2), "+v"(e2), "+v"(cd), "+v"(ed));
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(e));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(e2));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(e2));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(r[i]));
#define MIN3(i) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(r[i]) : "v"(c), "v"(e));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(u[i]) : "v"(r[i]));
#define ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(r[i]), "v"(c));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(r[i]) : );
#define CMPB(i)                                                                                      \
    {                                                                                                \
        unsigned long long m;                                                                        \
        uint32_t n;                                                                                  \
        asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(r[i]), "v"(c));                      \
        asm volatile("s_bcnt1_i32_b64 %0, %2\n\ts_add_u32 %1, %1, %0" : "=&s"(n), "+s"(sacc) : "s"(m) : "scc"); \
        (void)n;                                                                                     \
    }
#define CMPO(i)                                                                   \
    {                                                                             \
        unsigned long long m;                                                     \
        asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(r[i]), "v"(c));   \
        asm volatile("" ::"s"(m));                                                \
    }
#define CMPA(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[i]) : "v"(r[i]), "v"(c) : "vcc");
#define SBC(i)                                                                                                         \
    {                                                                                                                  \
        uint32_t n;                                                                                                    \
        asm volatile("s_bcnt1_i32_b64 %0, exec\n\ts_add_u32 %1, %1, %0" : "=&s"(n), "+s"(sacc) : : "scc");            \
        (void)n;                                                                                                       \
    }
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(cd), "v"(ed));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(ed));
#define CMP64(i)                                                                   \
    {                                                                              \
        unsigned long long m;                                                      \
        asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(d[i]), "v"(cd));   \
        asm volatile("" ::"s"(m));                                                 \
    }
#define PKF16(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c), "v"(e));
#define CVT16(i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(r[i]));
#define MF32(i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
#define MF16(i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[i], 0, 0, 0);
#define MF32F(i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(c, e, acc4[i], 0, 0, 0);
        if (T == T_FMA_F32) { REP16(FMA32) REP16(FMA32) REP16(FMA32) REP16(FMA32) }
        if (T == T_PK_FMA_F32) { REP16(PKFMA) REP16(PKFMA) REP16(PKFMA) REP16(PKFMA) }
        if (T == T_PK_ADD_F32) { REP16(PKADD) REP16(PKADD) REP16(PKADD) REP16(PKADD) }
        if (T == T_ALIGNBIT) { REP16(ALIGN) REP16(ALIGN) REP16(ALIGN) REP16(ALIGN) }
        if (T == T_MIN3_F32) { REP16(MIN3) REP16(MIN3) REP16(MIN3) REP16(MIN3) }
        if (T == T_ADD_U32) { REP16(ADDU) REP16(ADDU) REP16(ADDU) REP16(ADDU) }
        if (T == T_BCNT) { REP16(BCNT) REP16(BCNT) REP16(BCNT) REP16(BCNT) }
        if (T == T_AND_OR) { REP16(ANDOR) REP16(ANDOR) REP16(ANDOR) REP16(ANDOR) }
        if (T == T_CNDMASK) { REP16(CNDM) REP16(CNDM) REP16(CNDM) REP16(CNDM) }
        if (T == T_CMP_F32_BCNT) { REP16(CMPB) REP16(CMPB) REP16(CMPB) REP16(CMPB) }
        if (T == T_CMP_F32_ONLY) { REP16(CMPO) REP16(CMPO) REP16(CMPO) REP16(CMPO) }
        if (T == T_CMP_ADDC) { REP16(CMPA) REP16(CMPA) REP16(CMPA) REP16(CMPA) }
        if (T == T_SALU_BCNT) { REP16(SBC) REP16(SBC) REP16(SBC) REP16(SBC) }
        if (T == T_MIX_PK_PLAIN) {
            REP8(PKFMA) REP8(ALIGN) REP8(PKFMA) REP8(ALIGN) REP8(PKFMA) REP8(ALIGN) REP8(PKFMA) REP8(ALIGN)
        }
        if (T == T_PLANE_LOOP) {
#define PLANE1 REP16(PKFMA) REP8(ALIGN) REP4(MIN3) CMPO(0)
            PLANE1 PLANE1 PLANE1 PLANE1
        }
        if (T == T_FMA_F64) { REP8(FMA64) REP8(FMA64) REP8(FMA64) REP8(FMA64) REP8(FMA64) REP8(FMA64) REP8(FMA64) REP8(FMA64) }
        if (T == T_MUL_F64) { REP8(MUL64) REP8(MUL64) REP8(MUL64) REP8(MUL64) REP8(MUL64) REP8(MUL64) REP8(MUL64) REP8(MUL64) }
        if (T == T_ADD_F64) { REP8(ADD64) REP8(ADD64) REP8(ADD64) REP8(ADD64) REP8(ADD64) REP8(ADD64) REP8(ADD64) REP8(ADD64) }
        if (T == T_CMP_F64) { REP8(CMP64) REP8(CMP64) REP8(CMP64) REP8(CMP64) REP8(CMP64) REP8(CMP64) REP8(CMP64) REP8(CMP64) }
        if (T == T_PK_FMA_F16) { REP16(PKF16) REP16(PKF16) REP16(PKF16) REP16(PKF16) }
        if (T == T_CVT_F16) { REP16(CVT16) REP16(CVT16) REP16(CVT16) REP16(CVT16) }
        if (T == T_MFMA_F16_32) { REP4(MF32) REP4(MF32) REP4(MF32) REP4(MF32) }
        if (T == T_MFMA_F16_16) { REP4(MF16) REP4(MF16) REP4(MF16) REP4(MF16) }
        if (T == T_MFMA_F32_16) { REP4(MF32F) REP4(MF32F) REP4(MF32F) REP4(MF32F) }
        if (T == T_MFMA32_VALU4) { MF32(0) REP4(FMA32) MF32(1) REP4(FMA32) MF32(2) REP4(FMA32) MF32(3) REP4(FMA32) }
        if (T == T_MFMA32_VALU8) { MF32(0) REP8(FMA32) MF32(1) REP8(FMA32) MF32(2) REP8(FMA32) MF32(3) REP8(FMA32) }
        if (T == T_MFMA32_VALU12) {
            MF32(0) REP8(FMA32) REP4(FMA32) MF32(1) REP8(FMA32) REP4(FMA32) MF32(2) REP8(FMA32) REP4(FMA32) MF32(3) REP8(FMA32) REP4(FMA32)
        }
        if (T == T_MFMA32_VALU16) { MF32(0) REP16(FMA32) MF32(1) REP16(FMA32) MF32(2) REP16(FMA32) MF32(3) REP16(FMA32) }
        if (T == T_MFMA32_VALU24) {
            MF32(0) REP16(FMA32) REP8(FMA32) MF32(1) REP16(FMA32) REP8(FMA32) MF32(2) REP16(FMA32) REP8(FMA32) MF32(3) REP16(FMA32) REP8(FMA32)
        }
        if (T == T_MFMA32_PK8) { MF32(0) REP8(PKFMA) MF32(1) REP8(PKFMA) MF32(2) REP8(PKFMA) MF32(3) REP8(PKFMA) }
        if (T == T_MFMA32_ALIGN16) { MF32(0) REP16(ALIGN) MF32(1) REP16(ALIGN) MF32(2) REP16(ALIGN) MF32(3) REP16(ALIGN) }
        if (T == T_MFMA32_POST) {
            // the post-processing reads the accumulator the MFMA two slots back wrote (software-pipelined screen)
#define POST(a)                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[j & 7]) : "v"(acc[a][j])); \
    _Pragma("unroll") for (int j = 0; j < 16; j += 2) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(r[(j >> 1) & 7]) : "v"(acc[a][j]), "v"(acc[a][j + 1]));
            MF32(0) POST(2) MF32(1) POST(3) MF32(2) POST(0) MF32(3) POST(1)
        }
#define MINE(i) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(c));
#define MULE(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(c));
#define FMACE(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(e));
#define SUBE(i) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(c));
#define ANDE(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(r[i]));
#define LSHRE(i) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(u[i]));
#define MOVE(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(u[i]) : "v"(r[i]));
#define SUBSDWA(i) asm volatile("v_sub_f32_sdwa %0, |%0|, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(r[i]) : "v"(c));
#define PKRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[i]), "v"(c));
        if (T == T_MIN_F32_E32) { REP16(MINE) REP16(MINE) REP16(MINE) REP16(MINE) }
        if (T == T_MUL_F32_E32) { REP16(MULE) REP16(MULE) REP16(MULE) REP16(MULE) }
        if (T == T_FMAC_F32_E32) { REP16(FMACE) REP16(FMACE) REP16(FMACE) REP16(FMACE) }
        if (T == T_SUB_F32_E32) { REP16(SUBE) REP16(SUBE) REP16(SUBE) REP16(SUBE) }
        if (T == T_AND_B32_E32) { REP16(ANDE) REP16(ANDE) REP16(ANDE) REP16(ANDE) }
        if (T == T_LSHR_B32_E32) { REP16(LSHRE) REP16(LSHRE) REP16(LSHRE) REP16(LSHRE) }
        if (T == T_MOV_B32) { REP16(MOVE) REP16(MOVE) REP16(MOVE) REP16(MOVE) }
        if (T == T_SUB_F32_SDWA_ABS) { REP16(SUBSDWA) REP16(SUBSDWA) REP16(SUBSDWA) REP16(SUBSDWA) }
        if (T == T_CVT_PKRTZ) { REP16(PKRTZ) REP16(PKRTZ) REP16(PKRTZ) REP16(PKRTZ) }
        // the screen's post-processing of one 32 x 32 block pair: t = min(u1, u2) (VOP2), its sign bit into the lane's bit
        // string, min |t| over the lane's outputs -- on the accumulators the MFMAs of the trip BEFORE wrote
#define POST2(a, b)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                                  \
        float t_;                                                                                                     \
        asm volatile("v_min_f32_e32 %0, %1, %2" : "=v"(t_) : "v"(acc[a][j]), "v"(acc[b][j]));                         \
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[j & 7]) : "v"(t_));                                     \
        if (j & 1) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(r[(j >> 1) & 7]) : "v"(t_), "v"(r[8 + (j >> 1)])); \
        else r[8 + (j >> 1)] = t_;                                                                                    \
    }
        if (T == T_SCREEN2) { MF32(0) MF32(1) POST2(2, 3) MF32(2) MF32(3) POST2(0, 1) }
        if (T == T_SCREEN4) { MF32(0) MF32(1) MF32(0) MF32(1) POST2(2, 3) MF32(2) MF32(3) MF32(2) MF32(3) POST2(0, 1) }
        if (T == T_SCREEN1Q) { MF32(0) POST(2) MF32(1) POST(3) MF32(2) POST(0) MF32(3) POST(1) }
        if (T == T_DS_READ_B128_BC) {
#define DSR(i)                                                                                       \
    {                                                                                                \
        f32x4 v;                                                                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((uint32_t)((it + i) & 63) * 16u));       \
        asm volatile("" ::"v"(v));                                                                   \
    }
            REP16(DSR) REP16(DSR) REP16(DSR) REP16(DSR)
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + p[i].x + p[i].y + (float)u[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)d[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + (float)sacc;
    if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, w1 - w0};
}

struct Result {
    double wall_ms, cyc_per_inst, cyc_wall, ghz;
};

template <int T>
Result run(int waves_per_simd, int iters) {
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
    Stamp* stamps;
    float* out;
    hipMalloc(&stamps, sizeof(Stamp) * blocks * 4);
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<T><<<blocks, 256>>>(stamps, out, iters / 8, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T><<<blocks, 256>>>(stamps, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks * 4);
    hipMemcpy(h.data(), stamps, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (auto& s : h) {
        cyc += (double)s.cyc;
        real += (double)s.real;
    }
    cyc /= h.size();
    real /= h.size();
    hipFree(stamps);
    hipFree(out);
    Result r;
    r.wall_ms = ms;
    // a SIMD holds waves_per_simd waves that run side by side for `cyc` shader cycles each
    r.cyc_per_inst = cyc / ((double)waves_per_simd * iters * kPerTrip[T]);
    r.ghz = real > 0 ? cyc / (real / 100e6) / 1e9 : 0.0;   // s_memrealtime ticks at 100 MHz
    // the same from the launch's wall time at that clock (right also when the W waves do not fit a SIMD side by side:
    // the MFMA rows hold 144-192 registers, so W > 2-3 runs in rounds and the per-wave figure reads low)
    r.cyc_wall = ms * 1e-3 * r.ghz * 1e9 * 1024.0 / ((double)blocks * 4.0 * iters * kPerTrip[T]);
    return r;
}

template <int T>
void sweep(int iters) {
    printf("%-44s", kNames[T]);
    for (int w : {1, 2, 3, 4, 8}) {
        Result r = run<T>(w, iters);
        printf("  W=%d %6.2f|%6.2f @%.2f", w, r.cyc_per_inst, r.cyc_wall, r.ghz);
    }
    printf("\n");
    fflush(stdout);
}

template <int T>
void all(int iters) {
    sweep<T>(iters);
    if constexpr (T + 1 < T_COUNT) all<T + 1>(iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# %s, %d CUs, clockRate %.0f MHz; shader cycles per wave-instruction per SIMD (MFMA rows: per MFMA), W waves per SIMD: per-wave stamps | launch wall time @ sustained GHz\n",
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3);
    all<0>(iters);
    return 0;
}
