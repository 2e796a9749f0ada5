// Microbenchmark: VALU issue cost on gfx950 per wave64 instruction, by opcode, chain shape and waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 valu_issue.hip -o valu_issue.bin && ./valu_issue.bin
//
// Settles the question DESIGN section 8.1 of round 2 left open (and the round-2 review asked for): is a wave64 VALU
// instruction 2 or 4 cycles of a SIMD's issue on gfx950, what do the f64 / conversion / 64-bit-compare / packed
// instructions of the fine rasterizer's inner loop cost, and how much of a dependent chain's latency do 2 / 4 resident
// waves hide.  The SQ counters tick in quad-cycles (ACTIVE_INST_VALU / INSTS_VALU = 1.01) and cannot tell.
//
// Method: one workgroup of 256 * W threads per CU (4 * W waves = W waves on each of the CU's 4 SIMDs); every wave runs
// ITER iterations of a 64-instruction inline-asm block of ONE opcode, either as a dependent chain (every instruction reads
// the previous result) or as 8 independent chains (8 accumulators round-robin).  Reported per opcode and W:
//   cyc/wave  = kernel wall time (HIP events; 1.28 M instructions per wave, launch overhead < 1 %) x clockRate /
//               instructions of one wave: latency-bound when the chain is dependent and W = 1
//   cyc/SIMD  = cyc/wave / W: the issue cost per instruction when W waves share the SIMD (the throughput number)
//   tick      = s_memtime ticks per instruction (median over waves), as a cross-check of the clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

enum Op {
  FMA_F32, MUL_F32, ADD_F32, MAX_F32, MOV_B32, CNDMASK, CMP_F32, CMP_U64, PK_MUL_F32, PK_FMA_F32, PK_ADD_F32, PK_MOV_B32,
  CNDMASK_SGPR, CNDMASK_E64_VCC, CNDMASK_ALT_ADD, CNDMASK_ALT2, CNDMASK_VCCSET, CNDMASK_CONST, EXEC_MOV, EXEC_PKMOV, CMP_CNDMASK, MIN3_F32, MED3_F32, MUL_F64, FMA_F64, ADD_F64, CVT_F64_F32, CVT_F32_F64, CVT_ROUNDTRIP, RCP_F32, RCP_F64, MUL_LO_U32, LSHL_B64, BPERMUTE, READLANE, NOPS
};
static const char* kNames[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_mov_b32", "v_cndmask_b32", "v_cmp_lt_f32", "v_cmp_lt_u64",
                               "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mov_b32", "v_cndmask_b32 (sgpr mask)", "v_cndmask_b32_e64 (vcc)", "cndmask_e32 / v_add alternating", "2 cndmask_e32 / 2 v_add", "v_cndmask_b32 (vcc set)", "v_cndmask_b32 (inline 0)", "saveexec+v_mov+restore", "saveexec+v_pk_mov+restore", "v_cmp+v_cndmask pair", "v_min3_f32", "v_med3_f32", "v_mul_f64", "v_fma_f64", "v_add_f64",
                               "v_cvt_f64_f32", "v_cvt_f32_f64", "cvt f32->f64->f32 (pair)", "v_rcp_f32", "v_rcp_f64", "v_mul_lo_u32",
                               "v_lshlrev_b64", "ds_bpermute_b32", "v_readlane_b32", "(count)"};

// 8 instructions: dependent = all on accumulator 0; independent = one per accumulator.  %0..%7 accumulators (32-bit),
// %8..%15 accumulators (64-bit), %16 / %17 32-bit sources, %18 / %19 64-bit sources.
#define I8_32(INS, TAIL)                                                                                            \
  INS " %0, %0" TAIL "\n" INS " %1, %1" TAIL "\n" INS " %2, %2" TAIL "\n" INS " %3, %3" TAIL "\n" INS " %4, %4" TAIL "\n" INS \
      " %5, %5" TAIL "\n" INS " %6, %6" TAIL "\n" INS " %7, %7" TAIL "\n"
#define D8_32(INS, TAIL)                                                                                            \
  INS " %0, %0" TAIL "\n" INS " %0, %0" TAIL "\n" INS " %0, %0" TAIL "\n" INS " %0, %0" TAIL "\n" INS " %0, %0" TAIL "\n" INS \
      " %0, %0" TAIL "\n" INS " %0, %0" TAIL "\n" INS " %0, %0" TAIL "\n"
#define I8_64(INS, TAIL)                                                                                                  \
  INS " %8, %8" TAIL "\n" INS " %9, %9" TAIL "\n" INS " %10, %10" TAIL "\n" INS " %11, %11" TAIL "\n" INS " %12, %12" TAIL "\n" INS \
      " %13, %13" TAIL "\n" INS " %14, %14" TAIL "\n" INS " %15, %15" TAIL "\n"
#define D8_64(INS, TAIL)                                                                                            \
  INS " %8, %8" TAIL "\n" INS " %8, %8" TAIL "\n" INS " %8, %8" TAIL "\n" INS " %8, %8" TAIL "\n" INS " %8, %8" TAIL "\n" INS \
      " %8, %8" TAIL "\n" INS " %8, %8" TAIL "\n" INS " %8, %8" TAIL "\n"

#define RUN_BLOCK(BODY)                                                                                                  \
  asm volatile(".rept 8\n" BODY ".endr\n"                                                                                \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), \
                 "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)                                                        \
               : "v"(s0), "v"(s1), "v"(q0), "v"(q1)                                                                      \
               : "vcc", "scc", "s20", "s21")

template <int OP, bool DEP>
__global__ __launch_bounds__(1024) void k(int iters, unsigned long long* ticks, float* sink) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  float s0 = 1.0000001f, s1 = 1e-9f;
  double q0 = 1.0000000001, q1 = 1e-12;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == FMA_F32) { if (DEP) RUN_BLOCK(D8_32("v_fma_f32", ", %16, %17")); else RUN_BLOCK(I8_32("v_fma_f32", ", %16, %17")); }
    if constexpr (OP == MUL_F32) { if (DEP) RUN_BLOCK(D8_32("v_mul_f32", ", %16")); else RUN_BLOCK(I8_32("v_mul_f32", ", %16")); }
    if constexpr (OP == ADD_F32) { if (DEP) RUN_BLOCK(D8_32("v_add_f32", ", %17")); else RUN_BLOCK(I8_32("v_add_f32", ", %17")); }
    if constexpr (OP == MAX_F32) { if (DEP) RUN_BLOCK(D8_32("v_max_f32", ", %17")); else RUN_BLOCK(I8_32("v_max_f32", ", %17")); }
    if constexpr (OP == MOV_B32) {
      if (DEP) RUN_BLOCK("v_mov_b32 %0, %1\nv_mov_b32 %1, %0\nv_mov_b32 %0, %1\nv_mov_b32 %1, %0\nv_mov_b32 %0, %1\nv_mov_b32 %1, %0\nv_mov_b32 %0, %1\nv_mov_b32 %1, %0\n");
      else RUN_BLOCK("v_mov_b32 %0, %16\nv_mov_b32 %1, %16\nv_mov_b32 %2, %16\nv_mov_b32 %3, %16\nv_mov_b32 %4, %17\nv_mov_b32 %5, %17\nv_mov_b32 %6, %17\nv_mov_b32 %7, %17\n");
    }
    if constexpr (OP == CNDMASK) { if (DEP) RUN_BLOCK(D8_32("v_cndmask_b32", ", %16, vcc")); else RUN_BLOCK(I8_32("v_cndmask_b32", ", %16, vcc")); }
    if constexpr (OP == CMP_F32) {
      // a compare writes a lane mask (SGPR pair / vcc): "dependent" = compare feeding a v_cndmask that feeds the next compare
      if (DEP) RUN_BLOCK("v_cmp_lt_f32 vcc, %0, %16\nv_cndmask_b32 %0, %0, %17, vcc\nv_cmp_lt_f32 vcc, %0, %16\nv_cndmask_b32 %0, %0, %17, vcc\nv_cmp_lt_f32 vcc, %0, %16\nv_cndmask_b32 %0, %0, %17, vcc\nv_cmp_lt_f32 vcc, %0, %16\nv_cndmask_b32 %0, %0, %17, vcc\n");
      else RUN_BLOCK("v_cmp_lt_f32 vcc, %0, %16\nv_cmp_lt_f32 s[20:21], %1, %16\nv_cmp_lt_f32 vcc, %2, %16\nv_cmp_lt_f32 s[20:21], %3, %16\nv_cmp_lt_f32 vcc, %4, %16\nv_cmp_lt_f32 s[20:21], %5, %16\nv_cmp_lt_f32 vcc, %6, %16\nv_cmp_lt_f32 s[20:21], %7, %16\n");
    }
    if constexpr (OP == CMP_U64) {
      if (DEP) RUN_BLOCK("v_cmp_lt_u64 vcc, %8, %18\nv_cndmask_b32 %0, %0, %17, vcc\nv_cmp_lt_u64 vcc, %8, %18\nv_cndmask_b32 %0, %0, %17, vcc\nv_cmp_lt_u64 vcc, %8, %18\nv_cndmask_b32 %0, %0, %17, vcc\nv_cmp_lt_u64 vcc, %8, %18\nv_cndmask_b32 %0, %0, %17, vcc\n");
      else RUN_BLOCK("v_cmp_lt_u64 vcc, %8, %18\nv_cmp_lt_u64 s[20:21], %9, %18\nv_cmp_lt_u64 vcc, %10, %18\nv_cmp_lt_u64 s[20:21], %11, %18\nv_cmp_lt_u64 vcc, %12, %18\nv_cmp_lt_u64 s[20:21], %13, %18\nv_cmp_lt_u64 vcc, %14, %18\nv_cmp_lt_u64 s[20:21], %15, %18\n");
    }
    if constexpr (OP == PK_MUL_F32) { if (DEP) RUN_BLOCK(D8_64("v_pk_mul_f32", ", %18")); else RUN_BLOCK(I8_64("v_pk_mul_f32", ", %18")); }
    if constexpr (OP == PK_FMA_F32) { if (DEP) RUN_BLOCK(D8_64("v_pk_fma_f32", ", %18, %19")); else RUN_BLOCK(I8_64("v_pk_fma_f32", ", %18, %19")); }
    if constexpr (OP == PK_ADD_F32) { if (DEP) RUN_BLOCK(D8_64("v_pk_add_f32", ", %19")); else RUN_BLOCK(I8_64("v_pk_add_f32", ", %19")); }
    if constexpr (OP == PK_MOV_B32) {
      if (DEP) RUN_BLOCK("v_pk_mov_b32 %8, %9, %9 op_sel:[0,1]\nv_pk_mov_b32 %9, %8, %8 op_sel:[0,1]\nv_pk_mov_b32 %8, %9, %9 op_sel:[0,1]\nv_pk_mov_b32 %9, %8, %8 op_sel:[0,1]\nv_pk_mov_b32 %8, %9, %9 op_sel:[0,1]\nv_pk_mov_b32 %9, %8, %8 op_sel:[0,1]\nv_pk_mov_b32 %8, %9, %9 op_sel:[0,1]\nv_pk_mov_b32 %9, %8, %8 op_sel:[0,1]\n");
      else RUN_BLOCK("v_pk_mov_b32 %8, %18, %18 op_sel:[0,1]\nv_pk_mov_b32 %9, %18, %18 op_sel:[0,1]\nv_pk_mov_b32 %10, %18, %18 op_sel:[0,1]\nv_pk_mov_b32 %11, %18, %18 op_sel:[0,1]\nv_pk_mov_b32 %12, %19, %19 op_sel:[0,1]\nv_pk_mov_b32 %13, %19, %19 op_sel:[0,1]\nv_pk_mov_b32 %14, %19, %19 op_sel:[0,1]\nv_pk_mov_b32 %15, %19, %19 op_sel:[0,1]\n");
    }
    if constexpr (OP == CNDMASK_SGPR) {  // VOP3 form: the lane mask in an SGPR pair (what the compiler emits for all but one select)
      asm volatile("s_mov_b32 s20, 0x55555555\ns_mov_b32 s21, 0x55555555" ::: "s20", "s21");
      if (DEP) RUN_BLOCK(D8_32("v_cndmask_b32", ", %16, s[20:21]")); else RUN_BLOCK(I8_32("v_cndmask_b32", ", %16, s[20:21]"));
    }
    if constexpr (OP == CNDMASK_E64_VCC) {  // the same select in the VOP3 encoding, mask still in vcc
      if (DEP) RUN_BLOCK(D8_32("v_cndmask_b32_e64", ", %16, vcc")); else RUN_BLOCK(I8_32("v_cndmask_b32_e64", ", %16, vcc"));
    }
    if constexpr (OP == CNDMASK_ALT_ADD) {
      RUN_BLOCK("v_cndmask_b32_e32 %0, %0, %16, vcc\nv_add_f32 %1, %1, %17\nv_cndmask_b32_e32 %2, %2, %16, vcc\nv_add_f32 %3, %3, %17\nv_cndmask_b32_e32 %4, %4, %16, vcc\nv_add_f32 %5, %5, %17\nv_cndmask_b32_e32 %6, %6, %16, vcc\nv_add_f32 %7, %7, %17\n");
    }
    if constexpr (OP == CNDMASK_ALT2) {
      RUN_BLOCK("v_cndmask_b32_e32 %0, %0, %16, vcc\nv_cndmask_b32_e32 %1, %1, %16, vcc\nv_add_f32 %2, %2, %17\nv_add_f32 %3, %3, %17\nv_cndmask_b32_e32 %4, %4, %16, vcc\nv_cndmask_b32_e32 %5, %5, %16, vcc\nv_add_f32 %6, %6, %17\nv_add_f32 %7, %7, %17\n");
    }
    if constexpr (OP == CNDMASK_VCCSET) {
      asm volatile("s_mov_b32 vcc_lo, 0x33333333\ns_mov_b32 vcc_hi, 0x33333333" ::: "vcc");
      if (DEP) RUN_BLOCK(D8_32("v_cndmask_b32", ", %16, vcc")); else RUN_BLOCK(I8_32("v_cndmask_b32", ", %16, vcc"));
    }
    if constexpr (OP == CNDMASK_CONST) {
      asm volatile("s_mov_b32 vcc_lo, 0x33333333\ns_mov_b32 vcc_hi, 0x33333333" ::: "vcc");
      if (DEP) RUN_BLOCK(D8_32("v_cndmask_b32", ", 0, vcc")); else RUN_BLOCK(I8_32("v_cndmask_b32", ", 0, vcc"));
    }
    if constexpr (OP == EXEC_MOV) {  // 8 x (s_and_saveexec, v_mov, s_mov exec): counted as 8 "instructions" (one masked move each)
      asm volatile("s_mov_b32 vcc_lo, 0x33333333\ns_mov_b32 vcc_hi, 0x33333333" ::: "vcc");
      RUN_BLOCK("s_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %0, %16\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %1, %16\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %2, %16\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %3, %16\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %4, %16\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %5, %16\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %6, %16\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %7, %16\ns_mov_b64 exec, s[20:21]\n");
    }
    if constexpr (OP == EXEC_PKMOV) {  // one saveexec / restore around 6 v_pk_mov (the TopKPairs entry step): 8 "instructions" = 8 instructions
      asm volatile("s_mov_b32 vcc_lo, 0x33333333\ns_mov_b32 vcc_hi, 0x33333333" ::: "vcc");
      RUN_BLOCK("s_and_saveexec_b64 s[20:21], vcc\nv_pk_mov_b32 %8, %18, %18 op_sel:[0,1]\nv_pk_mov_b32 %9, %18, %18 op_sel:[0,1]\nv_pk_mov_b32 %10, %18, %18 op_sel:[0,1]\nv_pk_mov_b32 %11, %19, %19 op_sel:[0,1]\nv_pk_mov_b32 %12, %19, %19 op_sel:[0,1]\nv_pk_mov_b32 %13, %19, %19 op_sel:[0,1]\ns_mov_b64 exec, s[20:21]\n");
    }
    if constexpr (OP == CMP_CNDMASK) {  // the compiler's select idiom: compare into an SGPR pair, select on it
      if (DEP) RUN_BLOCK("v_cmp_lt_f32 s[20:21], %0, %16\nv_cndmask_b32 %0, %0, %17, s[20:21]\nv_cmp_lt_f32 s[20:21], %0, %16\nv_cndmask_b32 %0, %0, %17, s[20:21]\nv_cmp_lt_f32 s[20:21], %0, %16\nv_cndmask_b32 %0, %0, %17, s[20:21]\nv_cmp_lt_f32 s[20:21], %0, %16\nv_cndmask_b32 %0, %0, %17, s[20:21]\n");
      else RUN_BLOCK("v_cmp_lt_f32 s[20:21], %0, %16\nv_cndmask_b32 %1, %1, %17, s[20:21]\nv_cmp_lt_f32 vcc, %2, %16\nv_cndmask_b32 %3, %3, %17, vcc\nv_cmp_lt_f32 s[20:21], %4, %16\nv_cndmask_b32 %5, %5, %17, s[20:21]\nv_cmp_lt_f32 vcc, %6, %16\nv_cndmask_b32 %7, %7, %17, vcc\n");
    }
    if constexpr (OP == MIN3_F32) { if (DEP) RUN_BLOCK(D8_32("v_min3_f32", ", %16, %17")); else RUN_BLOCK(I8_32("v_min3_f32", ", %16, %17")); }
    if constexpr (OP == MED3_F32) { if (DEP) RUN_BLOCK(D8_32("v_med3_f32", ", %16, %17")); else RUN_BLOCK(I8_32("v_med3_f32", ", %16, %17")); }
    if constexpr (OP == MUL_F64) { if (DEP) RUN_BLOCK(D8_64("v_mul_f64", ", %18")); else RUN_BLOCK(I8_64("v_mul_f64", ", %18")); }
    if constexpr (OP == FMA_F64) { if (DEP) RUN_BLOCK(D8_64("v_fma_f64", ", %18, %19")); else RUN_BLOCK(I8_64("v_fma_f64", ", %18, %19")); }
    if constexpr (OP == ADD_F64) { if (DEP) RUN_BLOCK(D8_64("v_add_f64", ", %19")); else RUN_BLOCK(I8_64("v_add_f64", ", %19")); }
    if constexpr (OP == CVT_F64_F32) {  // no dependent form (type changes): both rows are independent
      RUN_BLOCK("v_cvt_f64_f32 %8, %0\nv_cvt_f64_f32 %9, %1\nv_cvt_f64_f32 %10, %2\nv_cvt_f64_f32 %11, %3\nv_cvt_f64_f32 %12, %4\nv_cvt_f64_f32 %13, %5\nv_cvt_f64_f32 %14, %6\nv_cvt_f64_f32 %15, %7\n");
    }
    if constexpr (OP == CVT_F32_F64) {
      RUN_BLOCK("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n");
    }
    if constexpr (OP == CVT_ROUNDTRIP) {  // f32 -> f64 -> f32 -> ...: 8 instructions, dependent or 4 chains of 2
      if (DEP) RUN_BLOCK("v_cvt_f64_f32 %8, %0\nv_cvt_f32_f64 %0, %8\nv_cvt_f64_f32 %8, %0\nv_cvt_f32_f64 %0, %8\nv_cvt_f64_f32 %8, %0\nv_cvt_f32_f64 %0, %8\nv_cvt_f64_f32 %8, %0\nv_cvt_f32_f64 %0, %8\n");
      else RUN_BLOCK("v_cvt_f64_f32 %8, %0\nv_cvt_f64_f32 %9, %1\nv_cvt_f64_f32 %10, %2\nv_cvt_f64_f32 %11, %3\nv_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n");
    }
    if constexpr (OP == RCP_F32) { if (DEP) RUN_BLOCK(D8_32("v_rcp_f32", "")); else RUN_BLOCK(I8_32("v_rcp_f32", "")); }
    if constexpr (OP == RCP_F64) { if (DEP) RUN_BLOCK(D8_64("v_rcp_f64", "")); else RUN_BLOCK(I8_64("v_rcp_f64", "")); }
    if constexpr (OP == MUL_LO_U32) { if (DEP) RUN_BLOCK(D8_32("v_mul_lo_u32", ", %16")); else RUN_BLOCK(I8_32("v_mul_lo_u32", ", %16")); }
    if constexpr (OP == LSHL_B64) {
      if (DEP) RUN_BLOCK("v_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %8, 1, %8\n");
      else RUN_BLOCK("v_lshlrev_b64 %8, 1, %8\nv_lshlrev_b64 %9, 1, %9\nv_lshlrev_b64 %10, 1, %10\nv_lshlrev_b64 %11, 1, %11\nv_lshlrev_b64 %12, 1, %12\nv_lshlrev_b64 %13, 1, %13\nv_lshlrev_b64 %14, 1, %14\nv_lshlrev_b64 %15, 1, %15\n");
    }
    if constexpr (OP == BPERMUTE) {
      if (DEP) RUN_BLOCK("ds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 %0, %16, %0\ns_waitcnt lgkmcnt(0)\n");
      else RUN_BLOCK("ds_bpermute_b32 %0, %16, %0\nds_bpermute_b32 %1, %16, %1\nds_bpermute_b32 %2, %16, %2\nds_bpermute_b32 %3, %16, %3\nds_bpermute_b32 %4, %16, %4\nds_bpermute_b32 %5, %16, %5\nds_bpermute_b32 %6, %16, %6\nds_bpermute_b32 %7, %16, %7\ns_waitcnt lgkmcnt(0)\n");
    }
    if constexpr (OP == READLANE) {
      RUN_BLOCK("v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s20, %2, 3\nv_readlane_b32 s21, %3, 5\nv_readlane_b32 s20, %4, 3\nv_readlane_b32 s21, %5, 5\nv_readlane_b32 s20, %6, 3\nv_readlane_b32 s21, %7, 5\n");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  // one record per wave
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  const float f = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if (f == 12345.678f) sink[0] = f;
}

struct Result {
  double wall_ms;
  double med_ticks;
};

template <int OP, bool DEP>
Result run_one(int waves_per_simd, int iters, unsigned long long* d_ticks, float* d_sink, int cus) {
  const int threads = 256 * waves_per_simd;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  k<OP, DEP><<<cus, threads>>>(4, d_ticks, d_sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  k<OP, DEP><<<cus, threads>>>(iters, d_ticks, d_sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h((size_t)cus * 16);
  CHECK(hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> v;
  for (int c = 0; c < cus; ++c)
    for (int w = 0; w < 4 * waves_per_simd; ++w) v.push_back(h[(size_t)c * 16 + w]);
  std::sort(v.begin(), v.end());
  Result r;
  r.wall_ms = ms;
  r.med_ticks = (double)v[v.size() / 2];
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
  return r;
}

static double g_ghz = 2.4;  // hipDeviceProp clockRate; the sustained clock under these loops is within a few % of it

template <int OP>
void report(unsigned long long* d_ticks, float* d_sink, int cus) {
  const int iters = 20000;
  const double n = (double)iters * 64;  // instructions per wave
  for (int dep = 1; dep >= 0; --dep) {
    printf("%-26s %-10s", kNames[OP], dep ? "dependent" : "8 chains");
    for (int w : {1, 2, 4}) {
      Result r = dep ? run_one<OP, true>(w, iters, d_ticks, d_sink, cus) : run_one<OP, false>(w, iters, d_ticks, d_sink, cus);
      const double cyc_wave = r.wall_ms * 1e-3 * g_ghz * 1e9 / n;  // shader cycles per instruction of one wave (wall clock)
      printf(" | W=%d %6.2f cyc/wave %5.2f cyc/SIMD (%5.3f tick)", w, cyc_wave, cyc_wave / w, r.med_ticks / n);
    }
    printf("\n");
  }
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned long long* d_ticks;
  float* d_sink;
  CHECK(hipMalloc(&d_ticks, (size_t)cus * 16 * 8));
  CHECK(hipMalloc(&d_sink, 64));
  g_ghz = p.clockRate / 1e6;
  printf("# %s, %d CUs, clockRate %.0f MHz, wave64; one workgroup of 256*W threads per CU (W waves per SIMD), 1.28M instructions per wave\n",
         p.gcnArchName, cus, p.clockRate / 1e3);
  {
    Result r = run_one<FMA_F32, false>(1, 20000, d_ticks, d_sink, cus);
    printf("# s_memtime: %.1f ticks per microsecond of kernel wall time in a one-wave-per-SIMD v_fma_f32 loop (a constant 100 MHz counter would read 100)\n",
           r.med_ticks / (r.wall_ms * 1e3));
  }
  printf("# cyc/wave = kernel wall time x clockRate / instructions of one wave; cyc/SIMD = cyc/wave / W (issue cost per instruction\n"
         "# when W waves share the SIMD); tick = s_memtime ticks per instruction of one wave (median over waves)\n");
  report<FMA_F32>(d_ticks, d_sink, cus);
  report<MUL_F32>(d_ticks, d_sink, cus);
  report<ADD_F32>(d_ticks, d_sink, cus);
  report<MAX_F32>(d_ticks, d_sink, cus);
  report<MOV_B32>(d_ticks, d_sink, cus);
  report<CNDMASK>(d_ticks, d_sink, cus);
  report<CMP_F32>(d_ticks, d_sink, cus);
  report<CMP_U64>(d_ticks, d_sink, cus);
  report<PK_MUL_F32>(d_ticks, d_sink, cus);
  report<PK_FMA_F32>(d_ticks, d_sink, cus);
  report<PK_ADD_F32>(d_ticks, d_sink, cus);
  report<PK_MOV_B32>(d_ticks, d_sink, cus);
  report<CNDMASK_SGPR>(d_ticks, d_sink, cus);
  report<CNDMASK_E64_VCC>(d_ticks, d_sink, cus);
  report<CNDMASK_ALT_ADD>(d_ticks, d_sink, cus);
  report<CNDMASK_ALT2>(d_ticks, d_sink, cus);
  report<CNDMASK_VCCSET>(d_ticks, d_sink, cus);
  report<CNDMASK_CONST>(d_ticks, d_sink, cus);
  report<EXEC_MOV>(d_ticks, d_sink, cus);
  report<EXEC_PKMOV>(d_ticks, d_sink, cus);
  report<CMP_CNDMASK>(d_ticks, d_sink, cus);
  report<MIN3_F32>(d_ticks, d_sink, cus);
  report<MED3_F32>(d_ticks, d_sink, cus);
  report<MUL_F64>(d_ticks, d_sink, cus);
  report<FMA_F64>(d_ticks, d_sink, cus);
  report<ADD_F64>(d_ticks, d_sink, cus);
  report<CVT_F64_F32>(d_ticks, d_sink, cus);
  report<CVT_F32_F64>(d_ticks, d_sink, cus);
  report<CVT_ROUNDTRIP>(d_ticks, d_sink, cus);
  report<RCP_F32>(d_ticks, d_sink, cus);
  report<RCP_F64>(d_ticks, d_sink, cus);
  report<MUL_LO_U32>(d_ticks, d_sink, cus);
  report<LSHL_B64>(d_ticks, d_sink, cus);
  report<BPERMUTE>(d_ticks, d_sink, cus);
  report<READLANE>(d_ticks, d_sink, cus);
  return 0;
}
