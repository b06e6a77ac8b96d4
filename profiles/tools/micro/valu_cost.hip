// Issue cost of single VALU instruction kinds on gfx950 with 4 waves per SIMD (every CU busy):
// ns per wave64 instruction per SIMD, relative to v_add_u32 -- and in shader CYCLES: every kernel stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop, the in-kernel clock is their
// ratio (MI355X_MICROARCH.md, DVFS give-back item 6), cycles = ns x that clock.
//   hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

#define REP8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)

#define K_ADD(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_AND_OR(r) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_ADD3K(r) asm volatile("v_add3_u32 %0, %0, %1, -8" : "+v"(r) : "v"(b0));
#define K_BFI(r) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_ALIGN(r) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(r) : "v"(b0));
#define K_ALIGNV(r) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_BFE(r) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(r));
#define K_CND32(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b0));
#define K_CND64(r) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "s"(sm));
#define K_CNDK(r) asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(r) : "s"(sm));
#define K_MIN(r) asm volatile("v_min_i32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_MIN3(r) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_LSHL(r) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
#define K_LSHLADD(r) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r) : "v"(b0));
#define K_LSHLOR(r) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r) : "v"(b0));
#define K_OR3(r) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_BITOP3(r) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xde" : "+v"(r) : "v"(b0), "v"(b1));
#define K_BITOP2(r) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0xde" : "+v"(r) : "v"(b0));
#define K_AND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_ANDK(r) asm volatile("v_and_b32 %0, 0xf0f0f0f, %0" : "+v"(r));
#define K_ANDS(r) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "s"(s0));
#define K_CMP(r) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(r), "v"(b0) : "vcc");
#define K_CMP64(r) asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(sm) : "v"(r), "v"(b0));
#define K_MOVDPP(r) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r));
#define K_MINDPP(r) asm volatile("v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(r));
#define K_ADDDPP(r) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));
#define K_SDWA(r) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r) : "v"(b0));
#define K_BCNT(r) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_FFBL(r) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r));
#define K_ADDCO(r) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r) : "v"(b0) : "vcc");
#define K_ADDCO64(r) asm volatile("v_add_co_u32 %0, %1, %0, %2" : "+v"(r), "=s"(sm) : "v"(b0));
#define K_MUL(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_MUL24(r) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_MAD24(r) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_PERM(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_FMAC(r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_FADD(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_FMAX(r) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_FMAX3(r) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
#define K_PKADD(r) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_PKMIN(r) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_READLANE(r) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(r));
#define K_SNOP(r) asm volatile("s_nop 0");
#define K_SAND(r) asm volatile("s_and_b64 %0, %0, %1" : "+s"(sm) : "s"(sm2));


#define K_OR(r) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_XOR(r) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_SUB(r) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_MOV(r) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(b0));
#define K_MOVK(r) asm volatile("v_mov_b32 %0, 0x12345" : "=v"(r));
#define K_ADDS(r) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r) : "s"(s0));
#define K_ADDI(r) asm volatile("v_add_u32 %0, 5, %0" : "+v"(r));
#define K_ADDL(r) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(r));
#define K_MAXU(r) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_LSHR(r) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r));
#define K_ASHR(r) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r));
#define K_FMUL(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_FSUB(r) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_FMIN(r) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_FMAS(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "s"(s0), "v"(b1));
#define K_FMAK(r) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(r) : "v"(b1));
#define K_FADDS(r) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "s"(s0));
#define K_FCMP(r) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r), "v"(b0) : "vcc");
#define K_FMED3(r) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b0), "v"(b1));
#define K_CVTFI(r) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r));
#define K_CVTIF(r) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r));
#define K_LOG(r) asm volatile("v_log_f32 %0, %0" : "+v"(r));
#define K_RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
#define K_PKFMA(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##r) : "v"(pb0), "v"(pb1));
#define K_PKFADD(r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##r) : "v"(pb0));
#define K_PKFMUL(r) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##r) : "v"(pb0));
#define K_ADD64(r) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p##r) : "v"(pb0));
#define K_BITOP3S(r) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xde" : "+v"(r) : "s"(s0), "v"(b1));
#define K_BITOP3L(r) asm volatile("v_bitop3_b32 %0, %0, 0xf0f0f0f, %1 bitop3:0xde" : "+v"(r) : "v"(b1));
#define K_SUBREV(r) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_XNOR(r) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_NOT(r) asm volatile("v_not_b32 %0, %0" : "+v"(r));
#define K_WRITELANE(r) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(r) : "s"(s0));
#define K_SWAP(r) asm volatile("v_swap_b32 %0, %1" : "+v"(r), "+v"(b0));
#define K_DSSWZ(r) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)" : "+v"(r));
#define K_BPERM(r) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(b0));
#define K_SMOV(r) asm volatile("s_mov_b32 %0, %0" : "+s"(s0));
#define K_SADD(r) asm volatile("s_add_u32 %0, %0, %0" : "+s"(s0) : : "scc");

#define K_CMPCND_VCC(r) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b0) : "vcc");
#define K_CMPCND_SG(r) asm volatile("v_cmp_lt_u32 %1, %0, %2\n v_cndmask_b32 %0, %0, %2, %1" : "+v"(r), "+s"(sm) : "v"(b0));
#define K_UNUSED_CND_VCC_SET(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b0), "{vcc}"(sm));
#define K_CMP64(r) asm volatile("v_cmp_le_u64 vcc, %0, %1" : : "v"(p##r), "v"(pb0) : "vcc");
#define K_CMP64S(r) asm volatile("v_cmp_le_u64 %0, %1, %2" : "=s"(sm) : "v"(p##r), "v"(pb0));
#define K_MINU(r) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(b0));
#define K_MINDPPU(r) asm volatile("v_min_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(b0));
#define K_MOV_DPP_BC(r) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(b0));

#define KERNEL(NAME, K)                                                                     \
  __global__ void NAME(unsigned *out, int iters, unsigned long long *stamps) {              \
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, \
             a6 = a0 + 17, a7 = a0 + 19, b0 = a0 ^ 0x55, b1 = a0 ^ 0x33, s0 = 7;           \
    unsigned long long sm = 0x5555, sm2 = 0x3333, pa0 = a0, pa1 = a1, pa2 = a2, pa3 = a3, pa4 = a4, pa5 = a5, pa6 = a6, pa7 = a7, pb0 = b0, pb1 = b1; \
    asm volatile("" : "+v"(pa0), "+v"(pa1), "+v"(pa2), "+v"(pa3), "+v"(pa4), "+v"(pa5), "+v"(pa6), "+v"(pa7), "+v"(pb0), "+v"(pb1));                                            \
    asm volatile("" : "+s"(sm), "+s"(sm2), "+s"(s0));                                       \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int i = 0; i < iters; ++i) {                                                        \
      REP8(K) REP8(K) REP8(K) REP8(K) REP8(K) REP8(K) REP8(K) REP8(K)                         \
    }                                                                                        \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + (unsigned)sm + (unsigned)(pa0 + pa1 + pa2 + pa3 + pa4 + pa5 + pa6 + pa7); \
  }

#define LIST(X) X(add, K_ADD) X(and_, K_AND) X(and_lit, K_ANDK) X(and_sgpr, K_ANDS) X(lshl, K_LSHL) X(min, K_MIN) X(bcnt, K_BCNT) X(ffbl, K_FFBL) \
  X(cndmask_vcc, K_CND32) X(cndmask_sgpr, K_CND64) X(cndmask_0_sgpr, K_CNDK) X(cmp_vcc, K_CMP) X(cmp_sgpr, K_CMP64) X(add_co_vcc, K_ADDCO) X(add_co_sgpr, K_ADDCO64) \
  X(and_or, K_AND_OR) X(add3, K_ADD3) X(add3_const, K_ADD3K) X(bfi, K_BFI) X(alignbit_const, K_ALIGN) X(alignbit_v, K_ALIGNV) X(bfe, K_BFE) X(min3, K_MIN3) \
  X(lshl_add, K_LSHLADD) X(lshl_or, K_LSHLOR) X(or3, K_OR3) X(bitop3_3src, K_BITOP3) X(bitop3_2src, K_BITOP2) X(perm, K_PERM) \
  X(mov_dpp, K_MOVDPP) X(min_dpp, K_MINDPP) X(add_dpp, K_ADDDPP) X(sdwa_shift, K_SDWA) X(mul_lo, K_MUL) X(mul_u24, K_MUL24) X(mad_u24, K_MAD24) \
  X(pk_add_u16, K_PKADD) X(pk_min_i16, K_PKMIN) X(fma, K_FMA) X(fmac, K_FMAC) X(fadd, K_FADD) X(fmax, K_FMAX) X(fmax3, K_FMAX3) X(exp, K_EXP) \
  X(readlane, K_READLANE) X(s_nop, K_SNOP) X(s_and_b64, K_SAND) \
  X(or_, K_OR) X(xor_, K_XOR) X(sub, K_SUB) X(subrev, K_SUBREV) X(xnor, K_XNOR) X(not_, K_NOT) X(mov, K_MOV) X(mov_lit, K_MOVK) X(add_sgpr, K_ADDS) X(add_inline, K_ADDI) X(add_lit, K_ADDL) \
  X(max_u32, K_MAXU) X(lshr, K_LSHR) X(ashr, K_ASHR) X(fmul, K_FMUL) X(fsub, K_FSUB) X(fmin, K_FMIN) X(fma_sgpr, K_FMAS) X(fma_inline, K_FMAK) X(fadd_sgpr, K_FADDS) \
  X(fcmp_vcc, K_FCMP) X(fmed3, K_FMED3) X(cvt_f32_i32, K_CVTFI) X(cvt_i32_f32, K_CVTIF) X(log, K_LOG) X(rcp, K_RCP) X(pk_fma_f32, K_PKFMA) X(pk_add_f32, K_PKFADD) X(pk_mul_f32, K_PKFMUL) \
  X(lshl_add_u64, K_ADD64) X(bitop3_sgpr, K_BITOP3S) X(writelane, K_WRITELANE) X(swap, K_SWAP) X(ds_swizzle_wait, K_DSSWZ) X(ds_bpermute_wait, K_BPERM) X(s_mov, K_SMOV) X(s_add, K_SADD) \
  X(cmp_cndmask_vcc_pair, K_CMPCND_VCC) X(cmp_cndmask_sgpr_pair, K_CMPCND_SG) X(cmp_u64_vcc, K_CMP64) X(cmp_u64_sgpr, K_CMP64S) X(min_u32, K_MINU) X(min_u32_dpp, K_MINDPPU) X(mov_dpp_bc, K_MOV_DPP_BC)

#define DEF(n, k) KERNEL(kern_##n, k)
LIST(DEF)

int main() {
  unsigned *out;
  if (hipMalloc(&out, 256 * 1024 * 4) != hipSuccess) return 1;
  unsigned long long *stamps, hst[512];
  if (hipMalloc(&stamps, sizeof(hst)) != hipSuccess) return 1;
  const int iters = 1000;
  double base = 0;
#define RUN(n, k)                                                                                  \
  for (int wps : {1, 4}) {                                                                         \
    hipEvent_t e0, e1;                                                                              \
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);                                           \
    kern_##n<<<256, 256 * wps>>>(out, iters, stamps);                                               \
    (void)hipEventRecord(e0);                                                                       \
    kern_##n<<<256, 256 * wps>>>(out, iters, stamps);                                               \
    (void)hipEventRecord(e1);                                                                       \
    (void)hipDeviceSynchronize();                                                                   \
    float ms;                                                                                       \
    (void)hipEventElapsedTime(&ms, e0, e1);                                                         \
    const double ns = ms * 1e6 / ((double)iters * 64 * wps);                                        \
    (void)hipMemcpy(hst, stamps, sizeof(hst), hipMemcpyDeviceToHost);                               \
    double ghz[256];                                                                                \
    for (int b = 0; b < 256; ++b) ghz[b] = (double)hst[2 * b] / (double)hst[2 * b + 1] * 0.1;         \
    std::sort(ghz, ghz + 256);                                                                      \
    if (wps == 4 && base == 0) base = ns;                                                           \
    printf("%-16s waves/SIMD %d: %6.3f ns per instruction per SIMD = %5.2f cycles at the %.3f GHz the kernel ran at%s\n", #n, wps, ns, ns * ghz[128], ghz[128], wps == 4 ? "" : "  (one wave alone)"); \
    if (wps == 4) printf("%-16s   relative to v_add_u32: %.2f\n", #n, ns / base);                   \
  }
  LIST(RUN)
  return 0;
}
