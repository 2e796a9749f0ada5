// raster_mesh_bwd.hip -- SoftRas backward of mesh rasterization for gfx950.
//
// Replaces RasterizeMeshesBackwardCudaKernel (pytorch3d/csrc/rasterize_meshes/rasterize_meshes.cu:433-625):
// per (pixel, k) sample with a face, recompute the forward through p3d_geom.h and scatter nine
// partials to grad_face_verts[f].  The reference issues nine global float atomics per sample; on
// MI355X device-scope atomics resolve beyond the per-XCD L2, so that design is atomic-bound (the
// first version of this kernel, with a wave-level segmented reduction, still spent 1.4 GB of HBM
// write traffic per launch on 11.5 MB of output).
//
// Design (numbers from profiles/microbench/lds_atomic.hip and profiles/ablate.py on MI355X):
//   * gfx950 executes ds_add_f32 one lane at a time (~190-260 CU-cycles per wave instruction whatever
//     the conflict pattern; ds_add_u32 / ds_wrxchg take ~4), so accumulating partials with LDS float
//     atomics costs twice the whole rest of the kernel.  This kernel uses NO float atomics in LDS.
//   * one wave owns a 16x16-pixel area of one image (a 256-thread workgroup = a 32x32 region), walked
//     as four 8x8 tiles, one pixel per lane; a lane reads its pixel's K-row of pix_to_face /
//     grad_zbuf / grad_dists / grad_bary with 16-byte loads, once, and only if the pixel has a face
//     (background rows are never fetched);
//   * per K slot, lanes that hit the same face are linked into a list with ONE integer LDS exchange
//     per lane on the face's hash-table slot (ds_wrxchg_rtn returns the previous visitor), the nine
//     partials are summed along the lists by pointer jumping (ds_bpermute, log2(group) steps), and
//     each list head adds its group's total to the wave-private table with plain LDS loads/stores
//     (no two heads of one wave instruction share a slot);
//   * a face's contributions from all pixels and all K slots of the 16x16 area meet in that table;
//     it is flushed with nine global atomics per (area, face) -- instead of nine per sample -- when
//     the area is done or the table runs full.
// Accumulation order is not deterministic (float atomics on the final flush), as in the reference
// (rasterize_meshes.cu:587 alertNotDeterministic).
#include "p3d_common.h"
#include "p3d_geom.h"
#include "wave_table.h"

#include <stdlib.h>

#include <type_traits>

namespace p3d {

namespace {

constexpr int kRegion = 32;  // pixels per workgroup-region side (four 16x16 wave areas)
// 4 waves x 182 slots x (8 + 12*4) B = 40768 B of LDS -> 4 workgroups per CU
constexpr int kBwdSlots = 182;
using FaceTable = WaveTable<9, kBwdSlots, kRows>;
using FaceTableV = WaveTable<9, kBwdSlots, kCorners>;  // flushed straight to grad_verts through faces_packed

struct BwdArgs {
  const float* face_verts;
  const int64_t* p2f;
  const float* grad_zbuf;
  const float* grad_bary;
  const float* grad_dists;
  float* grad_fv;           // (F,3,3), or (V,3) with `faces`
  const int64_t* faces;     // (F,3) or null: send the partials of face f to grad_verts[faces[f]] instead of grad_face_verts[f]
  int64_t V;  // vertices behind `faces` (index check of the fused scatter), -1 without
  int N, H, W, K;
  int RY, RX;  // regions per image
  unsigned scatter, nblocks;  // work item b is region / area (b * scatter) % nblocks (launch_mesh_backward)
  const int* area_list;       // rows kernel: the 16 x 16 areas that hold a face (area_list_kernel), or null: all of them
  const int* area_count;
  int persp, clip;
  const int* cover;  // row cover written by the forward (p3d_rasterize_meshes_with_cover) or null; (N, CY, CX) words
  int CY, CX;
  const float4* face_pre;  // (F) per-face reciprocals (p3d_geom.h: BwdFacePre, written by p3d_gather_face_verts_pre) or null
};

// Rows of the wave's 16 x 16 area that may hold a sample (bit r: row ay + r); all of them without a cover.
__device__ __forceinline__ unsigned area_rows(const BwdArgs& a, int n, int ay, int ax) {
  if (a.cover == nullptr) return 0xffffu;
  return (unsigned)a.cover[((int64_t)n * a.CY + (ay >> 4)) * a.CX + (ax >> 4)] & 0xffffu;
}

// A wave-uniform pointer the optimizer may not take apart: loads through it are [scalar base][32-bit lane offset] (global_load
// with an SGPR pair), not 64-bit per-lane address arithmetic.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* uniform_ptr(const T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (const __attribute__((address_space(1))) T*)(((unsigned long long)hi << 32) | (unsigned long long)lo);  // (global memory: not a flat access)
}

// Row loaders: KT contiguous elements starting at a (KT * elemsize)-aligned address.
template <int KT>
__device__ __forceinline__ void load_idx_row(const int64_t* p, int (&out)[KT]) {
  if constexpr (KT % 2 == 0) {
#pragma unroll
    for (int k = 0; k < KT; k += 2) {
      const longlong2 t = *reinterpret_cast<const longlong2*>(p + k);
      out[k] = (int)t.x;
      out[k + 1] = (int)t.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) out[k] = (int)p[k];
  }
}

template <int M>
__device__ __forceinline__ void load_f32_row(const float* p, float (&out)[M]) {
  if constexpr (M % 4 == 0) {
#pragma unroll
    for (int k = 0; k < M; k += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + k);
      out[k] = t.x;
      out[k + 1] = t.y;
      out[k + 2] = t.z;
      out[k + 3] = t.w;
    }
  } else if constexpr (M % 2 == 0) {
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      const float2 t = *reinterpret_cast<const float2*>(p + k);
      out[k] = t.x;
      out[k + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < M; ++k) out[k] = p[k];
  }
}

// The same rows with the slot PAIRS permuted: out pair j <- memory pair j ^ m (m < KT / 2).  C floats per slot.
template <int KT>
__device__ __forceinline__ void load_idx_row_pairs(const int64_t* p, int m, int (&out)[KT]) {
  if constexpr (KT % 2 == 0) {
#pragma unroll
    for (int j = 0; j < KT / 2; ++j) {
      const longlong2 t = *reinterpret_cast<const longlong2*>(p + 2 * (j ^ m));
      out[2 * j] = (int)t.x;
      out[2 * j + 1] = (int)t.y;
    }
  } else {
    load_idx_row<KT>(p, out);
  }
}

template <int KT, int C>
__device__ __forceinline__ void load_f32_row_pairs(const float* p, int m, float (&out)[KT * C]) {
  if constexpr (KT >= 4) {
#pragma unroll
    for (int j = 0; j < KT / 2; ++j) {
      const float* src = p + 2 * C * (j ^ m);  // 8-byte aligned: rows start on 16-byte boundaries, pairs are 8 * C bytes
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float2 t = *reinterpret_cast<const float2*>(src + 2 * c);
        out[2 * C * j + 2 * c] = t.x;
        out[2 * C * j + 2 * c + 1] = t.y;
      }
    }
  } else {
    load_f32_row<KT * C>(p, out);
  }
}

// KT > 0: K == KT, rows read with vector loads.  KT == 0: any K, per-slot scalar loads.
// TO_VERTS: the table is flushed to grad_verts through faces_packed (the scatter of `verts[faces]`'s backward fused in).
template <int KT, bool TO_VERTS>
__global__ __launch_bounds__(256, 4) void mesh_backward_kernel(BwdArgs a) {
  using Table = std::conditional_t<TO_VERTS, FaceTableV, FaceTable>;
  __shared__ __align__(16) int s_table[4][Table::kLdsInts];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  // region of this workgroup, 16x16 area of this wave
  long long t = (long long)(((unsigned long long)blockIdx.x * a.scatter) % a.nblocks);
  const int rx = (int)(t % a.RX);
  t /= a.RX;
  const int ry = (int)(t % a.RY);
  const int n = (int)(t / a.RY);
  const int ay = ry * kRegion + (w >> 1) * 16;
  const int ax = rx * kRegion + (w & 1) * 16;
  const int H = a.H, W = a.W, K = a.K;
  if (ay >= H || ax >= W) return;  // wave-uniform; no workgroup barriers in this kernel
  const unsigned rowmask = area_rows(a, n, ay, ax);
  if (rowmask == 0) return;  // the forward wrote no face into this area (uniform)

  Table tab;
  tab.init(s_table[w], lane);
  tab.index = a.faces;
  tab.index_limit = a.V;
  const bool persp = a.persp != 0, clip = a.clip != 0;

#pragma unroll 1
  for (int tile = 0; tile < 4; ++tile) {
    if (((rowmask >> ((tile >> 1) * 8)) & 0xffu) == 0) continue;  // uniform
    const int yo = ay + (tile >> 1) * 8 + (lane >> 3);
    const int xo = ax + (tile & 1) * 8 + (lane & 7);
    const bool ok = yo < H && xo < W;
    const int yi = H - 1 - yo, xi = W - 1 - xo;  // rasterize_meshes.cu:458-462
    const f2 p = mk2(pix_to_ndc(xi, W, H), pix_to_ndc(yi, H, W));
    const int64_t base = (((int64_t)n * H + yo) * W + xo) * K;

    if constexpr (KT > 0) {
      int f[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) f[k] = -1;
      // Neighbouring pixels hold (nearly) the same faces in the same depth order, so at a common slot k many lanes of
      // the wave hit ONE face and the table step has to sum long lists (log2(group) rounds of pointer jumping, the most
      // expensive part of this kernel).  Each lane therefore walks its K slots in its own order: slot PAIR j ^ m, with
      // m distinct within every 2x2 pixel block -- the same (pixel, slot) samples, spread so that a step sees ~4x fewer
      // lanes per face.  The permutation is done by the loads (pairs are 16 / 8 / 8 / 24 bytes of the four rows): it
      // costs no VALU and no registers (a butterfly of conditional swaps over the 48 row registers spilled 76 of them).
      const int m = KT >= 4 ? (((lane & 1) | (((lane >> 3) & 1) << 1)) & (KT / 2 - 1)) : 0;
      if (ok) load_idx_row_pairs<KT>(a.p2f + base, m, f);
      bool any = false;
#pragma unroll
      for (int k = 0; k < KT; ++k) any |= f[k] >= 0;
      if (__ballot(any) == 0) continue;  // wave-uniform: nothing rendered in this 8x8 tile
      float gz[KT], gd[KT], gb[3 * KT];
      if (any) {
        load_f32_row_pairs<KT, 1>(a.grad_zbuf + base, m, gz);
        load_f32_row_pairs<KT, 1>(a.grad_dists + base, m, gd);
        load_f32_row_pairs<KT, 3>(a.grad_bary + base * 3, m, gb);
      }
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (__ballot(f[k] >= 0) == 0) continue;  // wave-uniform
        FaceGrad r;
        if (f[k] >= 0) {
          const float* g = a.face_verts + (int64_t)f[k] * 9;
          const f3 v0 = mk3(g[0], g[1], g[2]);
          const f3 v1 = mk3(g[3], g[4], g[5]);
          const f3 v2 = mk3(g[6], g[7], g[8]);
          r = face_sample_bwd(v0, v1, v2, p, gz[k], mk3(gb[3 * k], gb[3 * k + 1], gb[3 * k + 2]), gd[k], persp, clip, false);
        }
        tab.add(a.grad_fv, lane, f[k], r.g);
      }
    } else {
      // any K: pix_to_face is read four slots (one 32-byte sector) at a time -- a lane per pixel reading 8 bytes at a
      // stride of 8*K fetches a whole sector per slot (K = 100: 3.3 ms, most of it this) -- and chunks without a face,
      // i.e. nearly all of a row's -1 padding, are skipped whole
      const bool vec = (K & 1) == 0;  // K-rows start on 16-byte boundaries
#pragma unroll 1
      for (int k0 = 0; k0 < K; k0 += 4) {
        int f4[4] = {-1, -1, -1, -1};
        if (ok) {
          const int64_t* src = a.p2f + base + k0;
          if (vec) {
            const longlong2 t0 = *reinterpret_cast<const longlong2*>(src);
            f4[0] = (int)t0.x;
            f4[1] = (int)t0.y;
            if (k0 + 2 < K) {
              const longlong2 t1 = *reinterpret_cast<const longlong2*>(src + 2);
              f4[2] = (int)t1.x;
              f4[3] = (int)t1.y;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (k0 + j < K) f4[j] = (int)src[j];
          }
        }
        if (__ballot((f4[0] >= 0) | (f4[1] >= 0) | (f4[2] >= 0) | (f4[3] >= 0)) == 0) continue;  // wave-uniform
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
          const int f = j == 0 ? f4[0] : (j == 1 ? f4[1] : (j == 2 ? f4[2] : f4[3]));
          if (__ballot(f >= 0) == 0) continue;  // wave-uniform
          const int64_t i = base + k0 + j;
          FaceGrad r;
          if (f >= 0) {
            const float* g = a.face_verts + (int64_t)f * 9;
            const f3 v0 = mk3(g[0], g[1], g[2]);
            const f3 v1 = mk3(g[3], g[4], g[5]);
            const f3 v2 = mk3(g[6], g[7], g[8]);
            const f3 gb = mk3(a.grad_bary[i * 3 + 0], a.grad_bary[i * 3 + 1], a.grad_bary[i * 3 + 2]);
            r = face_sample_bwd(v0, v1, v2, p, a.grad_zbuf[i], gb, a.grad_dists[i], persp, clip, false);
          }
          tab.add(a.grad_fv, lane, f, r.g);
        }
      }
    }
  }
  if (tab.used > 0) tab.flush(a.grad_fv, lane);
}


// ---------------------------------------------------------------------------------------------------------------------
// Sample-major form for K = 4, 8, 16 (round 3) and 32 (round 4: two pixels per 64-sample step; K = 32 on 8 bench meshes 0.95 -> 0.45 ms).  The kernel above gives a lane a PIXEL and walks its K slots: 56 row
// registers per lane (127 VGPRs, four waves per SIMD) and, per slot, a face-vertex gather whose latency nothing hides --
// the wave sits in s_waitcnt 47 % of its time (profiles/r02_fine_v2_rocprof.md) while VALU and LDS are each under half
// busy.  Here a lane owns one SAMPLE per step: a 16-pixel row segment of the four operands is 16 K contiguous samples,
// walked 64 at a time (K = 8: two steps per row; lane L of step t -> sample 64 t + L = pixel (64 t + L) / K, slot
// (64 t + L) % K).
//   * every load is one fully contiguous piece per wave (512 B of pix_to_face, 256 B of grad_zbuf / grad_dists, 768 B of
//     grad_bary per step) and a lane holds six operand registers instead of 7 K: the kernel fits 80 VGPRs, and with a
//     118-slot table (26 KB per workgroup, spill mode of wave_table.h) SIX waves per SIMD are resident instead of four --
//     occupancy is what hides the gather and the LDS round trips of the table (measured, K = 8: 1.26 -> 1.19 ms; a
//     software-pipelined variant of the same layout at four waves: 1.33; gathers issued a row ahead: no gain);
//   * a lane's pixel column takes K / 4 values over the steps of a row (their NDC x are computed once), the row's y is
//     uniform; pix_to_face of the next step is requested before the current one is computed.
// Table machinery (wave_table.h) unchanged: per step every lane contributes one sample.
// ---------------------------------------------------------------------------------------------------------------------
// The areas (words of the cover) that hold a face, in roughly ascending order (one atomic per wave of 64 words).
__global__ __launch_bounds__(256) void area_list_kernel(const int* __restrict__ cover, int nareas, int* __restrict__ list,
                                                         int* __restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < nareas && (cover[i] & 0xffff) != 0;
  const unsigned long long m = __ballot(on);
  if (m == 0) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(count, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (on) list[base + mask_rank(m)] = i;
}

// p3d_rasterize_meshes_cover_check: one thread per 16-pixel row segment; raises the flag when the segment holds a face
// (pix_to_face[n, y, x, 0] >= 0: entries are sorted, slot 0 is the first to fill) that the cover does not know of.
__global__ __launch_bounds__(256) void cover_check_kernel(const int64_t* __restrict__ p2f, const int* __restrict__ cover, int N, int H, int W,
                                                          int K, int CY, int CX, int* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * H * CX) return;
  const int cx = (int)(i % CX);
  const int64_t t = i / CX;
  const int y = (int)(t % H), n = (int)(t / H);
  bool any = false;
  for (int x = cx * 16; x < min(W, cx * 16 + 16); ++x) any |= p2f[(((int64_t)n * H + y) * W + x) * K] >= 0;
  const int word = cover[((int64_t)n * CY + (y >> 4)) * CX + cx];
  if (any && !((word >> (y & 15)) & 1)) atomicOr(flag, 1);
}

// The per-face reciprocals (p3d_geom.h: BwdFacePre) from face_verts, a thread per face: what p3d_gather_face_verts_pre writes beside its
// gather, for callers that arrive with face_verts already made (the reference's own signature: p3d_rasterize_meshes_backward_pre).
__global__ __launch_bounds__(256) void face_pre_kernel(const float* __restrict__ face_verts, int64_t F, float4* __restrict__ face_pre) {
  for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < F; f += (int64_t)gridDim.x * 256) {
    const float* q = face_verts + f * 9;
    const BwdFacePre r = bwd_face_pre_make(mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), mk3(q[6], q[7], q[8]));
    face_pre[f] = make_float4(r.inv_area, r.inv_l01, r.inv_l02, r.inv_l12);
  }
}

template <int KT, bool PRE = false>
struct RowsCfg {
  // Round 6: with the per-face reciprocals gathered (PRE) the K = 8 / K = 4 kernels fit 73 registers, and seven waves per SIMD with
  // 100-slot tables (7 workgroups x 22.4 KB of LDS per CU) run the bench launch 5.4 % faster than six with 116 (0.885 / 0.899 / 0.894
  // -> 0.837 / 0.853 / 0.844 ms alternating on one box, profiles/r06/c42; the light batch 0.555 -> 0.532).  Round 4 had tried seven
  // waves on the kernel of its day: +3..5 % on config 3, -6 % on the light batch, not taken.
  static constexpr bool kSeven = PRE && KT <= 8;
  static constexpr int kWaves = KT >= 16 ? 5 : (kSeven ? 7 : 6);     // waves per SIMD the kernel is built for (512 / kWaves registers)
  // 4 waves x kSlots x 56 B of LDS per workgroup (a multiple of 4: bucket probing).  The LDS, not the 70 registers, is what
  // holds K = 8 at six waves per SIMD: with 100 slots (seven waves) the launch measured 1.036 -> 0.98 / 1.00 ms on config 3 in
  // two runs but 0.654 -> 0.694 on the light batch, 104 and 92 slots no change (profiles/r04/r04c10/bwd_occ*.txt): not taken.
  // At eight waves (88 slots, 64 registers) the kernel spills and returns NaN -- the build refuses it.
  static constexpr int kSlots = KT >= 16 ? 136 : (kSeven ? 100 : 116);
  static constexpr int kPix = 64 / KT;                 // pixels per 64-sample step
};

// Lane <-> sample inside a step: lane = slot * kPix + pixel, i.e. the kPix neighbouring pixels of ONE slot sit in adjacent
// lanes (the loads address sample pixel * KT + slot of the same contiguous 64: same cache lines, permuted).  Neighbouring
// pixels hold the same face at the same slot more often than not, so a face's samples form RUNS of adjacent lanes, and
// runs can be summed with DPP row shifts -- VALU operations at ~3 cycles each, where a round of the table's list summation
// is ten ds_bpermute at ~24 (profiles/microbench/valu_issue_mi355x.txt).  run_reduce is a segmented inclusive scan
// (distances 1, 2, 4, 8 inside the group of kPix lanes); the last lane of a run ends up with the run's total and stays,
// the others leave the step (f = -1).  The table then sees each face once per run instead of once per sample.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);  // lanes shifted in from outside the row read 0
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int D, int PIX>
__device__ __forceinline__ void run_level(bool e, float (&g)[9]) {
  if constexpr (D < PIX) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const float sum = g[j] + dpp_f<0x110 + D>(g[j]);  // row_shr:D -- the partial sum D lanes to the left (one v_add_f32_dpp)
      g[j] = e ? sum : g[j];
    }
  }
}

template <int PIX>
__device__ __forceinline__ void run_reduce(int& f, float (&g)[9], int lane) {
  const int pix = lane & (PIX - 1);
  // Every shift is its own statement, executed by all 64 lanes: a DPP operand read from a lane that EXEC has switched off
  // comes back as 0 on gfx9, so none of them may end up behind the short circuit of an && (the first version did: runs
  // that started at a group's first pixel lost that pixel).
  const int f_left = dpp_i<0x111>(f);   // row_shr:1
  const int f_right = dpp_i<0x101>(f);  // row_shl:1
  // e_d: lanes i - d .. i hold the same face (and are inside the group)
  const int e1 = (pix >= 1) & (f >= 0) & (f_left == f);
  const int e1_left = dpp_i<0x111>(e1);
  const int e2 = PIX > 2 ? (e1 & e1_left) : 0;
  const int e2_left = dpp_i<0x112>(e2);
  const int e4 = PIX > 4 ? (e2 & e2_left) : 0;
  const int e4_left = dpp_i<0x114>(e4);
  const int e8 = PIX > 8 ? (e4 & e4_left) : 0;
  const bool continues = (pix < PIX - 1) & (f_right == f);  // the lane to the right carries the run on
  run_level<1, PIX>(e1 != 0, g);
  run_level<2, PIX>(e2 != 0, g);
  run_level<4, PIX>(e4 != 0, g);
  run_level<8, PIX>(e8 != 0, g);
  if (continues) f = -1;
}

// PC: perspective_correct && clip_barycentric_coords are known to be set (what the renderer uses for perspective cameras with blur,
// as in the forward's PC kernels): the step's arithmetic is one basic block instead of five behind uniform flag branches.
// PRE: the per-face reciprocals come from a.face_pre (one 16-byte gather per sample) instead of five v_rcp_f32 per sample.
template <int KT, bool TO_VERTS, bool PC = false, bool PRE = false>
__global__ __launch_bounds__(256, (RowsCfg<KT, PRE>::kWaves)) void mesh_backward_rows_kernel(BwdArgs a) {
  constexpr int SPR = KT / 4;  // steps per 16-pixel row segment
  constexpr int PIX = RowsCfg<KT>::kPix;
  static_assert(KT == 4 || KT == 8 || KT == 16 || KT == 32, "16 K samples per row segment, 64 per step");
  using Table = WaveTable<9, RowsCfg<KT, PRE>::kSlots, TO_VERTS ? kCorners : kRows, true, true>;
  __shared__ __align__(16) int s_table[4][Table::kLdsInts];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // One wave per 16 x 16 area.  With a cover the areas come from the list of those that hold a face (every workgroup
  // then carries four working waves: workgroups reach the CUs round robin, not by load, and a mix of empty and full ones
  // left half of the wave slots unused, profiles/r03/bwd_timeline.txt); without one, all areas in scattered order.
  const unsigned item = blockIdx.x * 4u + (unsigned)w;
  unsigned area;
  if (a.area_list != nullptr) {
    if (item >= (unsigned)*a.area_count) return;
    area = (unsigned)a.area_list[item];
  } else {
    if (item >= a.nblocks) return;
    area = (unsigned)(((unsigned long long)item * a.scatter) % a.nblocks);
  }
  const int cx = (int)(area % (unsigned)a.CX);
  const unsigned t = area / (unsigned)a.CX;
  const int cy = (int)(t % (unsigned)a.CY);
  const int n = (int)(t / (unsigned)a.CY);
  const int ay = cy * 16, ax = cx * 16;
  const int H = a.H, W = a.W;
  const int rows = min(16, H - ay);
  // Rows to walk: with the forward's row cover only those that hold a face -- at the bench workload 68 % of the 64-sample
  // steps hold none, and reading their pix_to_face to find that out was a third of this kernel (profiles/r03/bwd_ablate.txt).
  unsigned todo = area_rows(a, n, ay, ax) & ((1u << rows) - 1u);
  if (todo == 0) return;  // uniform

  Table tab;
  tab.init(s_table[w], lane);
  tab.index = a.faces;
  tab.index_limit = a.V;
  const bool persp = PC || a.persp != 0, clip = PC || a.clip != 0;

  const int seg = min(16, W - ax) * KT;               // samples of a row segment inside the image
  const int e = (lane & (PIX - 1)) * KT + lane / PIX;  // this lane's sample within a step's 64
  // Pixel centres, once per wave.  px[s]: NDC x of this lane's pixel in step s of a row (rasterize_meshes.cu:458-462); py_rows:
  // lane r holds the NDC y of row r of the area, a step fetches its row's with one v_readlane.  Both pass through an empty asm:
  // left to itself the compiler recomputed them inside the step loop -- two IEEE divisions (v_div_scale / v_rcp / 5 fma /
  // v_div_fmas / v_div_fixup each) per step, ~45 of the ~530 VALU instructions of a step (round 4, seen in the ISA).
  float px[SPR];
#pragma unroll
  for (int s = 0; s < SPR; ++s) {
    px[s] = pix_to_ndc(W - 1 - (ax + s * PIX + (lane & (PIX - 1))), W, H);
    asm volatile("" : "+v"(px[s]));
  }
  float py_rows = pix_to_ndc(H - 1 - (ay + (lane & 15)), H, W);
  asm volatile("" : "+v"(py_rows));
  const int64_t area_base = (((int64_t)n * H + ay) * W + ax) * KT;
  const int64_t row_pitch = (int64_t)W * KT;

  // Addresses: a step's operands are [uniform row pointer][this lane's fixed sample offset] -- the row pointer is scalar arithmetic,
  // the lane offsets (e, 3 e) never change: written with 64-bit per-lane indices each step cost nine vector instructions of
  // address arithmetic (round 6, seen in the ISA).
  const unsigned eu = (unsigned)e, eu3 = 3u * (unsigned)e;
  // pix_to_face one step ahead: the only load every step needs (steps without a sample cost nothing else)
  int rn = __builtin_ctz(todo), sn = 0;  // the next step: row, part of the row
  todo &= todo - 1;
  int f_nxt = e < seg ? (int)uniform_ptr(a.p2f + (area_base + (int64_t)rn * row_pitch))[eu] : -1;
#pragma unroll 1
  while (rn >= 0) {
    const int r = rn, s = sn;
    int f = f_nxt;
    if (++sn == SPR) {
      sn = 0;
      rn = todo ? __builtin_ctz(todo) : -1;
      todo &= todo - 1;  // (0 stays 0)
    }
    {
      const int en = 64 * sn + e;
      const auto* prow = uniform_ptr(a.p2f + (area_base + (int64_t)(rn < 0 ? 0 : rn) * row_pitch + 64 * sn));
      f_nxt = (rn >= 0 && en < seg) ? (int)prow[eu] : -1;
    }
    if (__ballot(f >= 0) == 0) continue;  // wave-uniform: nothing rendered in these 64 samples
    FaceGrad g;
    if (f >= 0) {
      const int64_t rb = area_base + (int64_t)r * row_pitch + 64 * s;  // uniform
      const auto* gzp = uniform_ptr(a.grad_zbuf + rb);
      const auto* gdp = uniform_ptr(a.grad_dists + rb);
      const auto* gbp = uniform_ptr(a.grad_bary + 3 * rb);
      const float gz = gzp[eu], gd = gdp[eu];
      const f3 gb = mk3(gbp[eu3], gbp[eu3 + 1u], gbp[eu3 + 2u]);
      const float* q = a.face_verts + (int64_t)f * 9;
      float pxs = px[0];
#pragma unroll
      for (int c = 1; c < SPR; ++c) pxs = s == c ? px[c] : pxs;
      const f2 p = mk2(pxs, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py_rows), r)));
      BwdFacePre pre = BwdFacePre();
      if constexpr (PRE) {
        const float4 r = a.face_pre[f];
        pre.inv_area = r.x;
        pre.inv_l01 = r.y;
        pre.inv_l02 = r.z;
        pre.inv_l12 = r.w;
      }
      g = face_sample_bwd<PRE>(mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), mk3(q[6], q[7], q[8]), p, gz, gb, gd, persp, clip, false, pre);
    } else {
#pragma unroll
      for (int j = 0; j < 9; ++j) g.g[j] = 0.0f;  // read by the neighbours' shifts, never added
    }
    run_reduce<PIX>(f, g.g, lane);
    tab.add(a.grad_fv, lane, f, g.g);
  }
  if (tab.used > 0) tab.flush(a.grad_fv, lane);
}

}  // namespace

}  // namespace p3d

using namespace p3d;

P3D_API size_t p3d_rasterize_meshes_backward_workspace_bytes(int N, int H, int W);

namespace {
unsigned scatter_multiplier(uint64_t items) {
  if (items <= 1) return 1;
  auto gcd = [](uint64_t x, uint64_t y) {
    while (y) {
      const uint64_t r = x % y;
      x = y;
      y = r;
    }
    return x;
  };
  uint64_t m = (uint64_t)((double)items * 0.6180339887) | 1u;
  while (gcd(m, items) != 1) m += 2;
  return (unsigned)(m % items);
}

int launch_mesh_backward(const float* face_verts, const int64_t* faces, int64_t V, const int64_t* p2f, const float* grad_zbuf,
                         const float* grad_bary, const float* grad_dists, int N, int H, int W, int K, int persp, int clip,
                         float* grad_out, const int32_t* cover, void* workspace, size_t workspace_bytes, hipStream_t s,
                         bool cover_has_list = false, const float* face_pre = nullptr) {
  BwdArgs a;
  a.face_pre = reinterpret_cast<const float4*>(face_pre);
  a.V = V;
  a.face_verts = face_verts;
  a.p2f = p2f;
  a.grad_zbuf = grad_zbuf;
  a.grad_bary = grad_bary;
  a.grad_dists = grad_dists;
  a.grad_fv = grad_out;
  a.faces = faces;
  a.N = N;
  a.H = H;
  a.W = W;
  a.K = K;
  a.RY = (int)ceil_div(H, kRegion);
  a.RX = (int)ceil_div(W, kRegion);
  a.persp = persp;
  a.clip = clip;
  a.cover = cover;
  a.CY = (H + 15) / 16;
  a.CX = (W + 15) / 16;
  a.area_list = nullptr;
  a.area_count = nullptr;
  const bool rows_kernel = K == 4 || K == 8 || K == 16 || K == 32;  // (32: round 4, with the on-chip forward queues for K > 16)
  // Work items in scattered order: workgroups reach XCDs and CUs round robin by index, and in (image, row, column) order
  // the index says where in the image the item is -- the CUs that drew the borders ran empty while the ones with the image
  // centres queued work.  b -> (b * scatter) mod items, odd multiplier near items / golden ratio, coprime: a bijection.
  const int64_t items = rows_kernel ? (int64_t)N * a.CY * a.CX : (int64_t)N * a.RY * a.RX;
  if (items > 0x7fffffffll) return P3D_ERR_INVALID_ARG;
  a.nblocks = (unsigned)items;
  a.scatter = scatter_multiplier((uint64_t)items);
  if (rows_kernel && cover != nullptr && cover_has_list) {
    // the forward listed the areas itself (p3d_rasterize_meshes_with_cover_list): no pass over the cover, no counter to clear
    const int* count = reinterpret_cast<const int*>(cover) + items;
    a.area_list = count + 16;
    a.area_count = count;
  } else if (rows_kernel && cover != nullptr && workspace != nullptr &&
      workspace_bytes >= p3d_rasterize_meshes_backward_workspace_bytes(N, H, W)) {
    int* count = static_cast<int*>(workspace);
    int* list = count + 16;
    if (hipMemsetAsync(count, 0, sizeof(int), s) != hipSuccess) return P3D_ERR_LAUNCH;
    LaunchScope ls("mesh_backward_areas", s);
    area_list_kernel<<<(unsigned)ceil_div(items, 256), 256, 0, s>>>(cover, (int)items, list, count);
    a.area_list = list;
    a.area_count = count;
  }
  LaunchScope ls("mesh_backward", s);
  const unsigned grid = rows_kernel ? (unsigned)ceil_div(items, 4) : (unsigned)items;
  const bool pc = persp && clip;
#define P3D_LAUNCH_MESH_BWD(TV)                                                  \
  switch (K) {                                                                   \
    case 1: mesh_backward_kernel<1, TV><<<grid, 256, 0, s>>>(a); break;          \
    case 2: mesh_backward_kernel<2, TV><<<grid, 256, 0, s>>>(a); break;          \
    case 4:                                                                      \
      if (pc && a.face_pre) mesh_backward_rows_kernel<4, TV, true, true><<<grid, 256, 0, s>>>(a); \
      else if (pc) mesh_backward_rows_kernel<4, TV, true><<<grid, 256, 0, s>>>(a); \
      else mesh_backward_rows_kernel<4, TV><<<grid, 256, 0, s>>>(a);             \
      break;                                                                     \
    case 8:                                                                      \
      if (pc && a.face_pre) mesh_backward_rows_kernel<8, TV, true, true><<<grid, 256, 0, s>>>(a); \
      else if (pc) mesh_backward_rows_kernel<8, TV, true><<<grid, 256, 0, s>>>(a); \
      else mesh_backward_rows_kernel<8, TV><<<grid, 256, 0, s>>>(a);             \
      break;                                                                     \
    case 16: mesh_backward_rows_kernel<16, TV><<<grid, 256, 0, s>>>(a); break;   \
    case 32: mesh_backward_rows_kernel<32, TV><<<grid, 256, 0, s>>>(a); break;   \
    default: mesh_backward_kernel<0, TV><<<grid, 256, 0, s>>>(a); break;         \
  }
  if (faces) {
    P3D_LAUNCH_MESH_BWD(true)
  } else {
    P3D_LAUNCH_MESH_BWD(false)
  }
#undef P3D_LAUNCH_MESH_BWD
  return launch_status();
}
}  // namespace

P3D_API size_t p3d_rasterize_meshes_backward_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return ((size_t)N * (size_t)((H + 15) / 16) * (size_t)((W + 15) / 16) + 16) * sizeof(int);
}

P3D_API int p3d_rasterize_meshes_backward_with_cover(const float* face_verts, const int64_t* p2f, const float* grad_zbuf,
                                                     const float* grad_bary, const float* grad_dists, const int32_t* cover,
                                                     int64_t F, int N, int H, int W, int K, int persp, int clip,
                                                     float* grad_face_verts, void* workspace, size_t workspace_bytes,
                                                     p3d_stream_t stream) {
  if (F < 0 || N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (F == 0) return P3D_OK;
  if (!grad_face_verts || !face_verts) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_face_verts, 0, (size_t)F * 9 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if (!p2f || !grad_zbuf || !grad_bary || !grad_dists) return P3D_ERR_INVALID_ARG;
  return launch_mesh_backward(face_verts, nullptr, -1, p2f, grad_zbuf, grad_bary, grad_dists, N, H, W, K, persp, clip,
                              grad_face_verts, cover, workspace, workspace_bytes, s);
}

P3D_API int p3d_rasterize_meshes_backward(const float* face_verts, const int64_t* p2f, const float* grad_zbuf,
                                          const float* grad_bary, const float* grad_dists, int64_t F, int N, int H,
                                          int W, int K, int persp, int clip, float* grad_face_verts,
                                          p3d_stream_t stream) {
  return p3d_rasterize_meshes_backward_with_cover(face_verts, p2f, grad_zbuf, grad_bary, grad_dists, nullptr, F, N, H, W, K,
                                                  persp, clip, grad_face_verts, nullptr, 0, stream);
}

P3D_API int p3d_rasterize_meshes_backward_verts_with_cover(const float* face_verts, const int64_t* faces, const int64_t* p2f,
                                                           const float* grad_zbuf, const float* grad_bary,
                                                           const float* grad_dists, const int32_t* cover, int64_t F,
                                                           int64_t V, int N, int H, int W, int K, int persp, int clip,
                                                           float* grad_verts, void* workspace, size_t workspace_bytes,
                                                           p3d_stream_t stream) {
  if (F < 0 || V < 0 || N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (V == 0) return P3D_OK;
  if (!grad_verts) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_verts, 0, (size_t)V * 3 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if (F == 0 || (int64_t)N * H * W * K == 0) return P3D_OK;
  if (!face_verts || !faces || !p2f || !grad_zbuf || !grad_bary || !grad_dists) return P3D_ERR_INVALID_ARG;
  return launch_mesh_backward(face_verts, faces, V, p2f, grad_zbuf, grad_bary, grad_dists, N, H, W, K, persp, clip,
                              grad_verts, cover, workspace, workspace_bytes, s);
}

P3D_API int p3d_rasterize_meshes_backward_verts(const float* face_verts, const int64_t* faces, const int64_t* p2f,
                                                const float* grad_zbuf, const float* grad_bary, const float* grad_dists,
                                                int64_t F, int64_t V, int N, int H, int W, int K, int persp, int clip,
                                                float* grad_verts, p3d_stream_t stream) {
  return p3d_rasterize_meshes_backward_verts_with_cover(face_verts, faces, p2f, grad_zbuf, grad_bary, grad_dists, nullptr, F, V,
                                                        N, H, W, K, persp, clip, grad_verts, nullptr, 0, stream);
}

P3D_API int p3d_rasterize_meshes_cover_check(const int64_t* p2f, const int32_t* cover, int N, int H, int W, int K, int32_t* stale,
                                             p3d_stream_t stream) {
  if (N < 0 || H < 0 || W < 0 || K < 0 || !stale) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(stale, 0, sizeof(int32_t), s) != hipSuccess) return P3D_ERR_LAUNCH;
  const int CY = (H + 15) / 16, CX = (W + 15) / 16;
  const int64_t items = (int64_t)N * H * CX;
  if (items == 0 || K == 0) return P3D_OK;
  if (!p2f || !cover) return P3D_ERR_INVALID_ARG;
  const int64_t blocks = ceil_div(items, 256);
  if (blocks > 0x7fffffffll) return P3D_ERR_INVALID_ARG;
  LaunchScope ls("mesh_cover_check", s);
  cover_check_kernel<<<(unsigned)blocks, 256, 0, s>>>(p2f, reinterpret_cast<const int*>(cover), N, H, W, K, CY, CX, reinterpret_cast<int*>(stale));
  return launch_status();
}

P3D_API int p3d_rasterize_meshes_backward_with_cover_list(const float* face_verts, const int64_t* p2f, const float* grad_zbuf,
                                                          const float* grad_bary, const float* grad_dists, const int32_t* cover_and_list,
                                                          int64_t F, int N, int H, int W, int K, int persp, int clip,
                                                          float* grad_face_verts, p3d_stream_t stream) {
  if (F < 0 || N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (F == 0) return P3D_OK;
  if (!grad_face_verts || !face_verts) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_face_verts, 0, (size_t)F * 9 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if (!p2f || !grad_zbuf || !grad_bary || !grad_dists) return P3D_ERR_INVALID_ARG;
  return launch_mesh_backward(face_verts, nullptr, -1, p2f, grad_zbuf, grad_bary, grad_dists, N, H, W, K, persp, clip, grad_face_verts,
                              cover_and_list, nullptr, 0, s, cover_and_list != nullptr);
}

P3D_API int p3d_rasterize_meshes_backward_verts_with_cover_list(const float* face_verts, const int64_t* faces, const int64_t* p2f,
                                                                const float* grad_zbuf, const float* grad_bary, const float* grad_dists,
                                                                const int32_t* cover_and_list, int64_t F, int64_t V, int N, int H, int W,
                                                                int K, int persp, int clip, float* grad_verts, p3d_stream_t stream) {
  if (F < 0 || V < 0 || N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (V == 0) return P3D_OK;
  if (!grad_verts) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_verts, 0, (size_t)V * 3 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if (F == 0 || (int64_t)N * H * W * K == 0) return P3D_OK;
  if (!face_verts || !faces || !p2f || !grad_zbuf || !grad_bary || !grad_dists) return P3D_ERR_INVALID_ARG;
  return launch_mesh_backward(face_verts, faces, V, p2f, grad_zbuf, grad_bary, grad_dists, N, H, W, K, persp, clip, grad_verts,
                              cover_and_list, nullptr, 0, s, cover_and_list != nullptr);
}

P3D_API int p3d_rasterize_meshes_backward_verts_pre(const float* face_verts, const float* face_pre, const int64_t* faces, const int64_t* p2f,
                                                    const float* grad_zbuf, const float* grad_bary, const float* grad_dists,
                                                    const int32_t* cover_and_list, int64_t F, int64_t V, int N, int H, int W, int K,
                                                    int persp, int clip, float* grad_verts, p3d_stream_t stream) {
  if (F < 0 || V < 0 || N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (V == 0) return P3D_OK;
  if (!grad_verts) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_verts, 0, (size_t)V * 3 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if (F == 0 || (int64_t)N * H * W * K == 0) return P3D_OK;
  if (!face_verts || !faces || !p2f || !grad_zbuf || !grad_bary || !grad_dists || ((uintptr_t)face_pre & 15u)) return P3D_ERR_INVALID_ARG;
  return launch_mesh_backward(face_verts, faces, V, p2f, grad_zbuf, grad_bary, grad_dists, N, H, W, K, persp, clip, grad_verts,
                              cover_and_list, nullptr, 0, s, cover_and_list != nullptr, face_pre);
}

P3D_API int p3d_rasterize_meshes_backward_pre(const float* face_verts, const int64_t* p2f, const float* grad_zbuf, const float* grad_bary,
                                              const float* grad_dists, const int32_t* cover, int cover_has_list, int64_t F, int N, int H,
                                              int W, int K, int persp, int clip, float* grad_face_verts, float* face_pre_scratch,
                                              void* workspace, size_t workspace_bytes, p3d_stream_t stream) {
  if (F < 0 || N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (F == 0) return P3D_OK;
  if (!grad_face_verts || !face_verts || ((uintptr_t)face_pre_scratch & 15u)) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_face_verts, 0, (size_t)F * 9 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if (!p2f || !grad_zbuf || !grad_bary || !grad_dists) return P3D_ERR_INVALID_ARG;
  const bool use_pre = face_pre_scratch != nullptr && persp && clip && (K == 4 || K == 8);  // (the kernels that read the records)
  if (use_pre) {
    int64_t blocks = ceil_div(F, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    LaunchScope ls("face_pre", s);
    face_pre_kernel<<<(unsigned)blocks, 256, 0, s>>>(face_verts, F, reinterpret_cast<float4*>(face_pre_scratch));
    const int st = launch_status();
    if (st != P3D_OK) return st;
  }
  const bool list = cover != nullptr && cover_has_list != 0;
  return launch_mesh_backward(face_verts, nullptr, -1, p2f, grad_zbuf, grad_bary, grad_dists, N, H, W, K, persp, clip, grad_face_verts, cover,
                              list ? nullptr : workspace, list ? 0 : workspace_bytes, s, list, use_pre ? face_pre_scratch : nullptr);
}
