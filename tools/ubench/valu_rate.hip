// valu_rate.hip -- VALU issue-rate microbenchmark (development tool): how many cycles does a wave64
// instruction of each kind occupy a SIMD?  Each kernel runs 8 independent dependency chains so the
// result is throughput-bound; 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAIN8(OP)                                                                                 \
  asm volatile(OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")            \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)  \
               : "v"(x), "v"(y));

#define DEFK(NAME, OP)                                                                             \
  __global__ __launch_bounds__(256) void NAME(int* out, int iters, int x, int y) {                \
    int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    for (int i = 0; i < iters; ++i) {                                                              \
      CHAIN8(OP) CHAIN8(OP) CHAIN8(OP) CHAIN8(OP) CHAIN8(OP) CHAIN8(OP) CHAIN8(OP) CHAIN8(OP)      \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;           \
  }

#define OP_ADD(R) "v_add_u32 " R ", " R ", %8\n"
#define OP_MAX(R) "v_max_i32 " R ", " R ", %8\n"
#define OP_MAX3(R) "v_max3_i32 " R ", " R ", %8, %9\n"
#define OP_AND(R) "v_and_b32 " R ", " R ", %8\n"
#define OP_BFI(R) "v_bfi_b32 " R ", %8, " R ", %9\n"
#define OP_ALIGN(R) "v_alignbit_b32 " R ", " R ", %8, 4\n"
#define OP_ADD3(R) "v_add3_u32 " R ", " R ", %8, %9\n"
#define OP_PKADD16(R) "v_pk_add_i16 " R ", " R ", %8\n"
#define OP_PKMAX16(R) "v_pk_max_i16 " R ", " R ", %8\n"
#define OP_PKADDU16(R) "v_pk_add_u16 " R ", " R ", %8\n"
#define OP_ADDF(R) "v_add_f32 " R ", " R ", %8\n"
#define OP_MAXF(R) "v_max_f32 " R ", " R ", %8\n"
#define OP_MAX3F(R) "v_max3_f32 " R ", " R ", %8, %9\n"
#define OP_FMAF(R) "v_fma_f32 " R ", " R ", %8, %9\n"
#define OP_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define OP_SDWA(R) "v_add_u32_sdwa " R ", " R ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define OP_DPP(R) "v_mov_b32_dpp " R ", " R " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 1, %8\n"
#define OP_ADDI16(R) "v_add_i16 " R ", " R ", %8\n"
#define OP_MAXI16(R) "v_max_i16 " R ", " R ", %8\n"
#define OP_PERM(R) "v_perm_b32 " R ", " R ", %8, %9\n"
#define OP_MADI24(R) "v_mad_i32_i24 " R ", " R ", %8, %9\n"
#define OP_SUBREV(R) "v_subrev_u32 " R ", " R ", %8\n"
#define OP_MED3(R) "v_med3_i32 " R ", " R ", %8, %9\n"
#define OP_MIN3(R) "v_min3_i32 " R ", " R ", %8, %9\n"
#define OP_PKMINU16(R) "v_pk_min_u16 " R ", " R ", %8\n"
#define OP_PKSUBU16(R) "v_pk_sub_u16 " R ", " R ", %8 clamp\n"
#define OP_PKMAD16(R) "v_pk_mad_i16 " R ", " R ", %8, %9\n"

#define OP_SUB(R) "v_sub_u32 " R ", " R ", %8\n"
#define OP_OR(R) "v_or_b32 " R ", " R ", %8\n"
#define OP_XOR(R) "v_xor_b32 " R ", " R ", %8\n"
#define OP_LSHL(R) "v_lshlrev_b32 " R ", 1, " R "\n"
#define OP_ASHR(R) "v_ashrrev_i32 " R ", 1, " R "\n"
#define OP_MOV(R) "v_mov_b32 " R ", %8\n"
#define OP_CMPGT(R) "v_cmp_gt_i32 vcc, " R ", %8\n"
#define OP_CMPGT16(R) "v_cmp_gt_i16 vcc, " R ", %8\n"
#define OP_CMPE64(R) "v_cmp_gt_i32 s[20:21], " R ", %8\n"
#define OP_CNDMASK2(R) "v_cndmask_b32 " R ", " R ", %8, s[20:21]\n"
#define OP_ADDC(R) "v_addc_co_u32 " R ", vcc, " R ", " R ", vcc\n"
#define OP_ADDU16(R) "v_add_u16 " R ", " R ", %8\n"
#define OP_MAXU16(R) "v_max_u16 " R ", " R ", %8\n"
#define OP_MINI16(R) "v_min_i16 " R ", " R ", %8\n"
#define OP_MAX3I16(R) "v_max3_i16 " R ", " R ", %8, %9\n"
#define OP_LSHLOR(R) "v_lshl_or_b32 " R ", " R ", 4, %8\n"
#define OP_ANDOR(R) "v_and_or_b32 " R ", " R ", %8, %9\n"
#define OP_OR3(R) "v_or3_b32 " R ", " R ", %8, %9\n"
#define OP_MAXU32(R) "v_max_u32 " R ", " R ", %8\n"
#define OP_MINI32(R) "v_min_i32 " R ", " R ", %8\n"
#define OP_MAXF16(R) "v_max_f16 " R ", " R ", %8\n"
#define OP_ADDF16(R) "v_add_f16 " R ", " R ", %8\n"
#define OP_MULF32(R) "v_mul_f32 " R ", " R ", %8\n"
#define OP_CVTI32F32(R) "v_cvt_i32_f32 " R ", " R "\n"
#define OP_CVTF32I32(R) "v_cvt_f32_i32 " R ", " R "\n"
#define OP_FRACT(R) "v_fract_f32 " R ", " R "\n"
#define OP_FLOOR(R) "v_floor_f32 " R ", " R "\n"
#define OP_RNDNE(R) "v_rndne_f32 " R ", " R "\n"
#define OP_TRUNC(R) "v_trunc_f32 " R ", " R "\n"
#define OP_SUBF32(R) "v_sub_f32 " R ", " R ", %8\n"
#define OP_BFE(R) "v_bfe_i32 " R ", " R ", 0, 16\n"
#define OP_ADD_E64(R) "v_add_u32_e64 " R ", " R ", %8\n"
#define OP_SUBU16(R) "v_sub_u16 " R ", " R ", %8\n"
#define OP_MULLO16(R) "v_mul_lo_u16 " R ", " R ", %8\n"
#define OP_PKADDF16(R) "v_pk_add_f16 " R ", " R ", %8\n"
#define OP_PKMAXF16(R) "v_pk_max_f16 " R ", " R ", %8\n"
#define OP_PKMULF32(R) "v_pk_mul_f32 " R ", " R ", %10\n"
#define OP_PKADDF32(R) "v_pk_add_f32 " R ", " R ", %10\n"
#define CHAIN8D(OP)                                                                                \
  asm volatile(OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")            \
               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)  \
               : "v"(x), "v"(y), "v"(dz));
#define DEFKD(NAME, OP)                                                                            \
  __global__ __launch_bounds__(256) void NAME(int* out, int iters, int x, int y) {                \
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7, dz = 1.0; \
    for (int i = 0; i < iters; ++i) {                                                              \
      CHAIN8D(OP) CHAIN8D(OP) CHAIN8D(OP) CHAIN8D(OP) CHAIN8D(OP) CHAIN8D(OP) CHAIN8D(OP) CHAIN8D(OP) \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);    \
  }
DEFKD(k_pkmulf32, OP_PKMULF32) DEFKD(k_pkaddf32, OP_PKADDF32)
DEFK(k_sub, OP_SUB) DEFK(k_or, OP_OR) DEFK(k_xor, OP_XOR) DEFK(k_lshl, OP_LSHL) DEFK(k_ashr, OP_ASHR) DEFK(k_mov, OP_MOV)
DEFK(k_cmpgt, OP_CMPGT) DEFK(k_cmpgt16, OP_CMPGT16) DEFK(k_cmpe64, OP_CMPE64) DEFK(k_cndmask2, OP_CNDMASK2) DEFK(k_addc, OP_ADDC)
DEFK(k_addu16, OP_ADDU16) DEFK(k_maxu16, OP_MAXU16) DEFK(k_mini16, OP_MINI16) DEFK(k_max3i16, OP_MAX3I16) DEFK(k_lshlor, OP_LSHLOR)
DEFK(k_andor, OP_ANDOR) DEFK(k_or3, OP_OR3) DEFK(k_maxu32, OP_MAXU32) DEFK(k_mini32, OP_MINI32) DEFK(k_maxf16, OP_MAXF16) DEFK(k_addf16, OP_ADDF16)
DEFK(k_cvti32f32, OP_CVTI32F32) DEFK(k_cvtf32i32, OP_CVTF32I32) DEFK(k_fract, OP_FRACT) DEFK(k_floor, OP_FLOOR) DEFK(k_rndne, OP_RNDNE) DEFK(k_trunc, OP_TRUNC)
DEFK(k_mulf32, OP_MULF32) DEFK(k_subf32, OP_SUBF32) DEFK(k_bfe, OP_BFE) DEFK(k_adde64, OP_ADD_E64) DEFK(k_subu16, OP_SUBU16) DEFK(k_mullo16, OP_MULLO16)
DEFK(k_pkaddf16, OP_PKADDF16) DEFK(k_pkmaxf16, OP_PKMAXF16)
DEFK(k_add, OP_ADD) DEFK(k_max, OP_MAX) DEFK(k_max3, OP_MAX3) DEFK(k_and, OP_AND) DEFK(k_bfi, OP_BFI)
DEFK(k_align, OP_ALIGN) DEFK(k_add3, OP_ADD3) DEFK(k_pkadd16, OP_PKADD16) DEFK(k_pkmax16, OP_PKMAX16)
DEFK(k_pkaddu16, OP_PKADDU16) DEFK(k_addf, OP_ADDF) DEFK(k_maxf, OP_MAXF) DEFK(k_max3f, OP_MAX3F)
DEFK(k_fmaf, OP_FMAF) DEFK(k_cndmask, OP_CNDMASK) DEFK(k_sdwa, OP_SDWA) DEFK(k_dpp, OP_DPP)
DEFK(k_lshladd, OP_LSHLADD) DEFK(k_addi16, OP_ADDI16) DEFK(k_maxi16, OP_MAXI16) DEFK(k_perm, OP_PERM)
DEFK(k_madi24, OP_MADI24) DEFK(k_med3, OP_MED3) DEFK(k_min3, OP_MIN3) DEFK(k_pkminu16, OP_PKMINU16)
DEFK(k_pksubu16, OP_PKSUBU16) DEFK(k_pkmad16, OP_PKMAD16)

typedef void (*kern_t)(int*, int, int, int);

int main() {
  int* out;
  hipMalloc(&out, 256 * 4096 * sizeof(int));
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 4;  // 4 blocks x 4 waves = 16 waves per CU = 4 per SIMD
  const int iters = 20000;
  struct E { const char* n; kern_t k; };
  std::vector<E> es = {{"v_add_u32", k_add}, {"v_max_i32", k_max}, {"v_max3_i32", k_max3}, {"v_and_b32", k_and},
                       {"v_bfi_b32", k_bfi}, {"v_alignbit_b32", k_align}, {"v_add3_u32", k_add3},
                       {"v_pk_add_i16", k_pkadd16}, {"v_pk_max_i16", k_pkmax16}, {"v_pk_add_u16", k_pkaddu16},
                       {"v_add_f32", k_addf}, {"v_max_f32", k_maxf}, {"v_max3_f32", k_max3f}, {"v_fma_f32", k_fmaf},
                       {"v_cndmask_b32", k_cndmask}, {"v_add_u32_sdwa", k_sdwa}, {"v_mov_b32_dpp wave_shr", k_dpp},
                       {"v_lshl_add_u32", k_lshladd}, {"v_add_i16", k_addi16}, {"v_max_i16", k_maxi16},
                       {"v_perm_b32", k_perm}, {"v_mad_i32_i24", k_madi24}, {"v_med3_i32", k_med3},
                       {"v_min3_i32", k_min3}, {"v_pk_min_u16", k_pkminu16}, {"v_pk_sub_u16 clamp", k_pksubu16},
                       {"v_pk_mad_i16", k_pkmad16},
 {"v_sub_u32", k_sub}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_ashrrev_i32", k_ashr}, {"v_mov_b32", k_mov},
 {"v_cmp_gt_i32 vcc", k_cmpgt}, {"v_cmp_gt_i16 vcc", k_cmpgt16}, {"v_cmp_gt_i32 sgpr", k_cmpe64}, {"v_cndmask_b32 sgpr", k_cndmask2}, {"v_addc_co_u32", k_addc},
 {"v_add_u16", k_addu16}, {"v_max_u16", k_maxu16}, {"v_min_i16", k_mini16}, {"v_max3_i16", k_max3i16}, {"v_lshl_or_b32", k_lshlor}, {"v_and_or_b32", k_andor},
 {"v_or3_b32", k_or3}, {"v_max_u32", k_maxu32}, {"v_min_i32", k_mini32}, {"v_max_f16", k_maxf16}, {"v_add_f16", k_addf16}, {"v_mul_f32", k_mulf32}, {"v_sub_f32", k_subf32},
 {"v_bfe_i32", k_bfe}, {"v_add_u32_e64", k_adde64}, {"v_sub_u16", k_subu16}, {"v_mul_lo_u16", k_mullo16}, {"v_pk_add_f16", k_pkaddf16}, {"v_pk_max_f16", k_pkmaxf16},
 {"v_pk_mul_f32", k_pkmulf32}, {"v_pk_add_f32", k_pkaddf32},
 {"v_cvt_i32_f32", k_cvti32f32}, {"v_cvt_f32_i32", k_cvtf32i32}, {"v_fract_f32", k_fract}, {"v_floor_f32", k_floor}, {"v_rndne_f32", k_rndne}, {"v_trunc_f32", k_trunc}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("device %s, %d CUs, clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  for (auto& e : es) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 100, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)iters * 64 * 4;  // 64 instrs per iter, 4 waves per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd;
    const double tlane = (double)blocks * 4 * iters * 64 * 64 / (ms * 1e-3) / 1e12;
    printf("%-26s %8.3f ms  %5.2f cyc/wave-instr @2.4GHz  %6.1f T lane-ops/s\n", e.n, ms, cyc, tlane);
  }
  return 0;
}
