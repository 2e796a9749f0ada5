// Microbenchmark: WHICH VALU instructions of gfx950 issue at the fast rate?
//
//   hipcc --offload-arch=gfx950 -O2 valu_classes.hip -o valu_classes.bin && ./valu_classes.bin
//
// valu_issue.hip (round 3) found two classes at four waves per SIMD: v_mul_f32 / v_add_f32 / v_mov_b32 (and v_cndmask_b32
// in its VOP2 form when mixed with them) cost ~2.3-2.5 nominal cycles of a SIMD per wave64 instruction, everything else it
// tried (v_fma_f32, v_max_f32, compares, VOP3 selects, every v_pk_*, every f64 instruction, conversions, v_readlane) ~4.1-
// 4.6.  The hot kernels of this repository are bound by VALU issue, so the class of an instruction is
// worth as much as its count.  This file sorts the rest of the instructions those kernels use -- or could use instead --
// into the two classes: subtraction, integer add / logic / shifts, the VOP2 vs VOP3 encoding of the same operation
// (modifiers, SGPR and literal operands), v_fmac vs v_fma, compares into vcc vs an SGPR pair, DPP forms, 64-bit moves
// (v_mov_b64 vs v_pk_mov_b32: the pair queues of topk.h are made of the latter), v_swap_b32, the IEEE-division helpers,
// transcendentals, and mixes of the two classes.
//
// Method as in valu_issue.hip: one workgroup of 256 * W threads per CU (W waves per SIMD), every wave runs ITER iterations
// of a 64-instruction block (8 x 8 instructions on 8 independent accumulators: throughput, not latency); reported is
// kernel wall time x clockRate / instructions of one wave / W = nominal cycles of a SIMD per instruction, and the same in
// nanoseconds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// one instruction on accumulator r: A "%r" B "%r" C
#define G1(A, B, C, r) A "%" #r B "%" #r C "\n"
#define G8(A, B, C) G1(A, B, C, 0) G1(A, B, C, 1) G1(A, B, C, 2) G1(A, B, C, 3) G1(A, B, C, 4) G1(A, B, C, 5) G1(A, B, C, 6) G1(A, B, C, 7)
#define G8D(A, B, C) G1(A, B, C, 8) G1(A, B, C, 9) G1(A, B, C, 10) G1(A, B, C, 11) G1(A, B, C, 12) G1(A, B, C, 13) G1(A, B, C, 14) G1(A, B, C, 15)
// two instructions alternating over the accumulators (class mixes)
#define G2(A1, B1, C1, A2, B2, C2) \
  G1(A1, B1, C1, 0) G1(A2, B2, C2, 1) G1(A1, B1, C1, 2) G1(A2, B2, C2, 3) G1(A1, B1, C1, 4) G1(A2, B2, C2, 5) G1(A1, B1, C1, 6) G1(A2, B2, C2, 7)

// %0..%7 32-bit accumulators, %8..%15 64-bit accumulators, %16 / %17 32-bit sources, %18 / %19 64-bit sources; s20..s23 free
#define OPS(X)                                                                                                        \
  X("v_mul_f32 (VOP2)               ", G8("v_mul_f32 ", ", ", ", %16"))                                                \
  X("v_add_f32 (VOP2)               ", G8("v_add_f32 ", ", ", ", %17"))                                                \
  X("v_sub_f32 (VOP2)               ", G8("v_sub_f32 ", ", ", ", %17"))                                                \
  X("v_subrev_f32                   ", G8("v_subrev_f32 ", ", ", ", %17"))                                             \
  X("v_mul_f32_e64 (forced VOP3)    ", G8("v_mul_f32_e64 ", ", ", ", %16"))                                            \
  X("v_mul_f32 -src (neg modifier)  ", G8("v_mul_f32_e64 ", ", -", ", %16"))                                           \
  X("v_add_f32 |src| (abs modifier) ", G8("v_add_f32_e64 ", ", |", "|, %17"))                                          \
  X("v_mul_f32 sgpr operand (VOP2)  ", G8("v_mul_f32 ", ", s20, ", ""))                                                \
  X("v_mul_f32 literal (VOP2)       ", G8("v_mul_f32 ", ", 0x3f800001, ", ""))                                         \
  X("v_fma_f32                      ", G8("v_fma_f32 ", ", ", ", %16, %17"))                                           \
  X("v_fmac_f32 (VOP2)              ", G8("v_fmac_f32 ", ", %16, %17 ; ", ""))                                          \
  X("v_max_f32 (VOP2)               ", G8("v_max_f32 ", ", ", ", %17"))                                                \
  X("v_min_f32 (VOP2)               ", G8("v_min_f32 ", ", ", ", %17"))                                                \
  X("v_and_b32                      ", G8("v_and_b32 ", ", ", ", %16"))                                                \
  X("v_or_b32                       ", G8("v_or_b32 ", ", ", ", %16"))                                                 \
  X("v_xor_b32                      ", G8("v_xor_b32 ", ", ", ", %16"))                                                \
  X("v_add_u32                      ", G8("v_add_u32 ", ", ", ", %16"))                                                \
  X("v_sub_u32                      ", G8("v_sub_u32 ", ", ", ", %16"))                                                \
  X("v_add_co_u32 (carry out)       ", G8("v_add_co_u32 ", ", vcc, ", ", %16"))                                        \
  X("v_lshlrev_b32                  ", G8("v_lshlrev_b32 ", ", 1, ", ""))                                              \
  X("v_lshrrev_b32                  ", G8("v_lshrrev_b32 ", ", 1, ", ""))                                              \
  X("v_mul_u32_u24                  ", G8("v_mul_u32_u24 ", ", ", ", %16"))                                            \
  X("v_mad_u32_u24                  ", G8("v_mad_u32_u24 ", ", ", ", %16, %17"))                                       \
  X("v_add3_u32                     ", G8("v_add3_u32 ", ", ", ", %16, %17"))                                          \
  X("v_lshl_add_u32                 ", G8("v_lshl_add_u32 ", ", ", ", 2, %17"))                                        \
  X("v_and_or_b32                   ", G8("v_and_or_b32 ", ", ", ", %16, %17"))                                        \
  X("v_bfe_u32                      ", G8("v_bfe_u32 ", ", ", ", 1, 5"))                                               \
  X("v_perm_b32                     ", G8("v_perm_b32 ", ", ", ", %16, %17"))                                          \
  X("v_cvt_f32_i32                  ", G8("v_cvt_f32_i32 ", ", ", ""))                                                 \
  X("v_cvt_i32_f32                  ", G8("v_cvt_i32_f32 ", ", ", ""))                                                 \
  X("v_cmp_lt_f32 -> vcc (VOPC)     ", G8("v_cmp_lt_f32 vcc, ", ", %16 ; ", ""))                                        \
  X("v_cmp_lt_f32 -> sgpr pair      ", G8("v_cmp_lt_f32 s[20:21], ", ", %16 ; ", ""))                                   \
  X("v_cmp_gt_u32 -> vcc            ", G8("v_cmp_gt_u32 vcc, ", ", %16 ; ", ""))                                        \
  X("v_cmp_class_f32 -> vcc         ", G8("v_cmp_class_f32 vcc, ", ", %16 ; ", ""))                                     \
  X("v_cndmask_b32 e32 (vcc)        ", G8("v_cndmask_b32_e32 ", ", ", ", %16, vcc"))                                   \
  X("v_cndmask_b32 e64 (sgpr pair)  ", G8("v_cndmask_b32_e64 ", ", ", ", %16, s[22:23]"))                              \
  X("cndmask e32 / v_mul alternating", G2("v_cndmask_b32_e32 ", ", ", ", %16, vcc", "v_mul_f32 ", ", ", ", %16"))       \
  X("cndmask e64 / v_mul alternating", G2("v_cndmask_b32_e64 ", ", ", ", %16, s[22:23]", "v_mul_f32 ", ", ", ", %16"))  \
  X("v_mul_f32 / v_fma_f32 alternat.", G2("v_mul_f32 ", ", ", ", %16", "v_fma_f32 ", ", ", ", %16, %17"))                \
  X("v_mul_f32 / v_cmp (vcc) altern.", G2("v_mul_f32 ", ", ", ", %16", "v_cmp_lt_f32 vcc, ", ", %16 ; ", ""))            \
  X("v_mul_f32 / v_max_f32 alternat.", G2("v_mul_f32 ", ", ", ", %16", "v_max_f32 ", ", ", ", %17"))                     \
  X("v_add_f32_dpp row_shr:1        ", G8("v_add_f32_dpp ", ", ", ", %17 row_shr:1 row_mask:0xf bank_mask:0xf"))       \
  X("v_mov_b32_dpp row_shr:1        ", G8("v_mov_b32_dpp ", ", ", " row_shr:1 row_mask:0xf bank_mask:0xf"))            \
  X("v_mov_b32_dpp quad_perm        ", G8("v_mov_b32_dpp ", ", ", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))  \
  X("v_mov_b32                      ", G8("v_mov_b32 ", ", %16 ; ", ""))                                                \
  X("v_mov_b64                      ", G8D("v_mov_b64 ", ", %18 ; ", ""))                                               \
  X("v_pk_mov_b32                   ", G8D("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", ""))                          \
  X("v_lshlrev_b64                  ", G8D("v_lshlrev_b64 ", ", 1, ", ""))                                             \
  X("v_swap_b32 (acc <-> source)    ", G8("v_swap_b32 ", ", %16 ; ", ""))                                               \
  X("v_pk_add_f32                   ", G8D("v_pk_add_f32 ", ", ", ", %19"))                                            \
  X("v_add_f64                      ", G8D("v_add_f64 ", ", ", ", %19"))                                               \
  X("v_mul_f64                      ", G8D("v_mul_f64 ", ", ", ", %18"))                                               \
  X("v_rcp_f32                      ", G8("v_rcp_f32 ", ", ", ""))                                                     \
  X("v_rsq_f32                      ", G8("v_rsq_f32 ", ", ", ""))                                                     \
  X("v_sqrt_f32                     ", G8("v_sqrt_f32 ", ", ", ""))                                                    \
  X("v_exp_f32                      ", G8("v_exp_f32 ", ", ", ""))                                                     \
  X("v_div_scale_f32                ", G8("v_div_scale_f32 ", ", vcc, ", ", %16, %17"))                                \
  X("v_div_fmas_f32                 ", G8("v_div_fmas_f32 ", ", ", ", %16, %17"))                                      \
  X("v_div_fixup_f32                ", G8("v_div_fixup_f32 ", ", ", ", %16, %17"))                                     \
  X("v_mbcnt_lo_u32_b32             ", G8("v_mbcnt_lo_u32_b32 ", ", -1, ", ""))                                        \
  X("v_readlane_b32                 ", G8("v_readlane_b32 s20, ", ", 3 ; ", ""))                                        \
  X("v_readfirstlane_b32            ", G8("v_readfirstlane_b32 s20, ", " ; ", ""))                                      \
  X("exec-masked: saveexec + 6 v_mul + restore (8 instr.)",                                                            \
    "s_and_saveexec_b64 s[20:21], s[22:23]\n" G1("v_mul_f32 ", ", ", ", %16", 0) G1("v_mul_f32 ", ", ", ", %16", 1)      \
        G1("v_mul_f32 ", ", ", ", %16", 2) G1("v_mul_f32 ", ", ", ", %16", 3) G1("v_mul_f32 ", ", ", ", %16", 4)         \
            G1("v_mul_f32 ", ", ", ", %16", 5) "s_mov_b64 exec, s[20:21]\n")                                            \
  X("exec-masked: saveexec + 6 v_mov_b64 + restore       ",                                                            \
    "s_and_saveexec_b64 s[20:21], s[22:23]\n" G1("v_mov_b64 ", ", %18 ; ", "", 8) G1("v_mov_b64 ", ", %18 ; ", "", 9)    \
        G1("v_mov_b64 ", ", %18 ; ", "", 10) G1("v_mov_b64 ", ", %19 ; ", "", 11) G1("v_mov_b64 ", ", %19 ; ", "", 12)   \
            G1("v_mov_b64 ", ", %19 ; ", "", 13) "s_mov_b64 exec, s[20:21]\n")                                          \
  X("exec-masked: saveexec + 6 v_pk_mov_b32 + restore    ",                                                            \
    "s_and_saveexec_b64 s[20:21], s[22:23]\n" G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 8)                  \
        G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 9) G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 10) \
            G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 11) G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 12) \
                G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 13) "s_mov_b64 exec, s[20:21]\n")                  \
  X("queue entry step: 6 v_pk_mov_b32 under two masks (9)",                                                           \
    "s_and_saveexec_b64 s[20:21], s[22:23]\n" G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 8)                  \
        G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 9) G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 10) \
            "s_and_b64 exec, exec, vcc\n" G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 8)                      \
                G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 9) G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 10) \
                    "s_mov_b64 exec, s[20:21]\n")                                                                      \
  X("queue entry step: 4 v_pk_mov_b32 + 4 v_mov_b32 interleaved (11)",                                                 \
    "s_and_saveexec_b64 s[20:21], s[22:23]\n" G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 8)                  \
        G1("v_mov_b32 ", ", %16 ; ", "", 0) G1("v_pk_mov_b32 ", ", %18, %18 op_sel:[0,1] ; ", "", 9)                     \
            G1("v_mov_b32 ", ", %16 ; ", "", 1) "s_and_b64 exec, exec, vcc\n"                                           \
                G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 8) G1("v_mov_b32 ", ", %17 ; ", "", 0)             \
                    G1("v_pk_mov_b32 ", ", %19, %19 op_sel:[0,1] ; ", "", 9) G1("v_mov_b32 ", ", %17 ; ", "", 1)         \
                        "s_mov_b64 exec, s[20:21]\n")                                                                  \
  X("queue entry step: 12 v_mov_b32 under two masks (15)",                                                             \
    "s_and_saveexec_b64 s[20:21], s[22:23]\n" G1("v_mov_b32 ", ", %16 ; ", "", 0) G1("v_mov_b32 ", ", %16 ; ", "", 1)   \
        G1("v_mov_b32 ", ", %16 ; ", "", 2) G1("v_mov_b32 ", ", %16 ; ", "", 3) G1("v_mov_b32 ", ", %16 ; ", "", 4)      \
            G1("v_mov_b32 ", ", %16 ; ", "", 5) "s_and_b64 exec, exec, vcc\n" G1("v_mov_b32 ", ", %17 ; ", "", 0)       \
                G1("v_mov_b32 ", ", %17 ; ", "", 1) G1("v_mov_b32 ", ", %17 ; ", "", 2) G1("v_mov_b32 ", ", %17 ; ", "", 3) \
                    G1("v_mov_b32 ", ", %17 ; ", "", 4) G1("v_mov_b32 ", ", %17 ; ", "", 5) "s_mov_b64 exec, s[20:21]\n") \
  X("exec-masked: saveexec + 12 v_mov_b32 + restore (14) ",                                                            \
    "s_and_saveexec_b64 s[20:21], s[22:23]\n" G8("v_mov_b32 ", ", %16 ; ", "") G1("v_mov_b32 ", ", %17 ; ", "", 0)       \
        G1("v_mov_b32 ", ", %17 ; ", "", 1) G1("v_mov_b32 ", ", %17 ; ", "", 2) G1("v_mov_b32 ", ", %17 ; ", "", 3)      \
            "s_mov_b64 exec, s[20:21]\n")

enum { kBase = __COUNTER__ + 1 };
#define X_COUNT(name, body) +1
constexpr int kOps = 0 OPS(X_COUNT);
#define X_NAME(name, body) name,
static const char* kNames[] = {OPS(X_NAME)};
// instructions per block of each entry (for the per-instruction figure): 8 unless stated in the name
static int block_len(int op) {
  // "(N)" in the name: N instructions per block; else 8
  const char* n = kNames[op];
  for (const char* p = n; *p; ++p)
    if (p[0] == '(' && p[1] >= '0' && p[1] <= '9') {
      int v = 0;
      const char* q = p + 1;
      while (*q >= '0' && *q <= '9') v = v * 10 + (*q++ - '0');
      if (*q == ')') return v;
    }
  return 8;
}

#define RUN_BLOCK(BODY)                                                                                                  \
  asm volatile(".rept 8\n" BODY ".endr\n"                                                                                \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), \
                 "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(q0), "+v"(q1)                 \
               :                                                                                                         \
               : "vcc", "scc", "s20", "s21")

template <int OP>
__global__ __launch_bounds__(1024) void k(int iters, float* sink) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  float s0 = 1.0000001f, s1 = 1e-9f;
  double q0 = 1.0000000001, q1 = 1e-12;
  // s[22:23]: a lane mask with half of the lanes on (selects and exec-masked blocks); s20: an SGPR operand
  asm volatile("s_mov_b32 s22, 0x55555555\ns_mov_b32 s23, 0x33333333\ns_mov_b32 s20, 0x3f800001\ns_mov_b32 vcc_lo, 0x0f0f0f0f\ns_mov_b32 vcc_hi, 0x00ff00ff" ::
                   : "s20", "s22", "s23", "vcc");
  for (int it = 0; it < iters; ++it) {
#define X_RUN(name, body) \
  if constexpr (OP == __COUNTER__ - kBase) RUN_BLOCK(body);
    OPS(X_RUN)
  }
  const float f = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + s0 + s1 + (float)(q0 + q1);
  if (f == 12345.678f) sink[0] = f;
}

static double g_ghz = 2.4;

template <int OP>
double run_one(int waves_per_simd, int iters, float* d_sink, int cus) {
  const int threads = 256 * waves_per_simd;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  k<OP><<<cus, threads>>>(4, d_sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  k<OP><<<cus, threads>>>(iters, d_sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
  return ms;
}

template <int OP>
void report_from(float* d_sink, int cus) {
  if constexpr (OP < kOps) {
    const int iters = 10000;
    const double n = (double)iters * 8 * block_len(OP);  // instructions per wave
    printf("%-56s", kNames[OP]);
    for (int w : {1, 2, 4}) {
      const double ms = run_one<OP>(w, iters, d_sink, cus);
      const double ns = ms * 1e6 / n / w;  // per instruction and SIMD
      printf(" | W=%d %5.2f cyc %5.3f ns", w, ns * g_ghz, ns);
    }
    printf("\n");
    fflush(stdout);
    report_from<OP + 1>(d_sink, cus);
  }
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float* d_sink;
  CHECK(hipMalloc(&d_sink, 64));
  g_ghz = p.clockRate / 1e6;
  printf("# %s, %d CUs, clockRate %.0f MHz, wave64; one workgroup of 256*W threads per CU (W waves per SIMD); per entry: nominal\n"
         "# cycles (wall time x clockRate) and nanoseconds of one SIMD per wave64 instruction, 8 independent accumulators\n",
         p.gcnArchName, cus, p.clockRate / 1e3);
  report_from<0>(d_sink, cus);
  return 0;
}
