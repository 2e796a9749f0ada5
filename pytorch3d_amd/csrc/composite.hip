// composite.hip -- alpha / normalised-weighted-sum / weighted-sum compositing for gfx950.
//
// Replaces alphaCompositeCuda{Forward,Backward}Kernel (pytorch3d/csrc/compositing/
// alpha_composite.cu:24-141), weightedSumNormCuda* (norm_weighted_sum.cu:24-154) and
// weightedSumCuda* (weighted_sum.cu:22-113).
//
// The reference runs one thread per (channel, pixel): each of the C threads of a pixel re-reads
// the pixel's K (index, alpha) pairs, accumulates into its own output element with atomicAdd and
// forces contiguous (N,K,H,W) copies of what arrive as permuted (N,H,W,K) views
// (alpha_composite.h:63-65).  Here one thread owns a pixel: it loads the K pairs once into
// VGPRs (through the caller's strides -- no copy), walks the channels, and writes each output
// element exactly once; grad_alphas is accumulated in registers and written once, only
// grad_features (a genuine scatter) uses f32 atomics.
#include "p3d_common.h"
#include "wave_table.h"

namespace p3d {
namespace {

struct CompArgs {
  const float* features;     // logical (C, P), element strides fs0 (channel), fs1 (point)
  int64_t fs0, fs1;          // (P, 1): planar, what the reference's operators take; (1, C): the renderers' transposed view of (P, C)
  int64_t gs0, gs1;          // the same for grad_features
  const float* alphas;       // logical (N,K,H,W)
  const int64_t* idx;        // logical (N,K,H,W)
  const float* grad_out;     // (N,C,H,W)
  int N, C, K, H, W;
  int64_t P;
  int64_t as[4], is[4];      // element strides
  float* result;             // (N,C,H,W)
  float* grad_features;      // (C,P)
  float* grad_alphas;        // (N,K,H,W) contiguous
  // alphas and idx are the renderers' permuted views of contiguous (N,H,W,K) tensors (element strides (H W K, 1, W K, K)): the K
  // entries of a pixel are adjacent and the pixels of an image row follow each other -- set by the launcher (sample_major_of)
  int sample_major;
};

// 12 / 16 adjacent bytes at 4-byte alignment: one global_load_dwordx3 / x4
struct __attribute__((packed, aligned(4))) Feat3 {
  float x, y, z;
};
struct __attribute__((packed, aligned(4))) Feat4 {
  float x, y, z, w;
};

constexpr float kEpsAlpha = 1e-9f;  // alpha_composite.cu:20
constexpr float kEpsNorm = 1e-4f;   // norm_weighted_sum.cu:20

// KT > 0: the K pairs are cached in registers (K <= KT).  KT == 0: re-read per channel.
// IL: the features are the transposed view of a (P, C) tensor (fs0 == 1, fs1 == C), else (C, P) planes (fs0 == P, fs1 == 1) -- a
// template flag because with run-time strides every one of the 4 x KT gathers in flight keeps its own 64-bit address (the KT = 16
// kernel then needs 282 registers: one wave per SIMD, refused by the build); here a gather is pointer-of-the-point + constant
// offset, or uniform plane pointer + the point's index.
template <int MODE, int KT, bool IL = false>
__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs a) {
  // sample-major inputs (round 5): a lane that reads ITS pixel's K entries touches 8 (4) bytes at a stride of 8 K (4 K) -- every
  // load instruction of the wave spreads over 64 K / 16 cache lines and the ten of them re-fetch the same lines from L2 (measured on
  // BASELINE configs[3]: FETCH_SIZE 3.6 x the algorithmic bytes, 0.043 ms for 47 MB).  The wave's 64 pixels x K entries are ONE
  // contiguous run: it is loaded 64 consecutive entries per instruction and transposed through a wave-private LDS slab.
  __shared__ int s_id[KT > 0 ? 4 : 1][KT > 0 ? 64 * KT : 1];
  __shared__ float s_al[KT > 0 ? 4 : 1][KT > 0 ? 64 * KT : 1];
  const int64_t npix = (int64_t)a.N * a.H * a.W;
  const int K = a.K, C = a.C;
  const int64_t HW = (int64_t)a.H * a.W;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x + w * 64; t0 < npix; t0 += (int64_t)gridDim.x * blockDim.x) {  // wave-uniform
    const int64_t t = t0 + lane;
    const bool valid = t < npix;
    const int64_t tc = valid ? t : npix - 1;
    const int n = (int)(tc / HW);
    const int64_t yx = tc % HW;
    const int y = (int)(yx / a.W), x = (int)(yx % a.W);
    const int64_t abase = n * a.as[0] + y * a.as[2] + x * a.as[3];
    const int64_t ibase = n * a.is[0] + y * a.is[2] + x * a.is[3];
    float* out = a.result + ((int64_t)n * C) * HW + yx;

    if constexpr (KT > 0) {
      int id[KT];
      float al[KT];
      if (a.sample_major) {  // uniform
        const int64_t e0 = t0 * K;
        const int64_t e1 = (t0 + 64 < npix ? t0 + 64 : npix) * K;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int sidx = j * 64 + lane;
          if (j < K && e0 + sidx < e1) {
            s_id[w][sidx] = (int)a.idx[e0 + sidx];
            s_al[w][sidx] = a.alphas[e0 + sidx];
          }
        }
        __builtin_amdgcn_wave_barrier();  // the slab is this wave's own: its LDS operations execute in order
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          id[k] = -1;
          al[k] = 0.0f;
          if (k < K && valid) {
            id[k] = s_id[w][lane * K + k];
            al[k] = s_al[w][lane * K + k];
          }
        }
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          id[k] = -1;
          al[k] = 0.0f;
          if (k < K && valid) {
            id[k] = (int)a.idx[ibase + k * a.is[1]];
            al[k] = a.alphas[abase + k * a.as[1]];
          }
        }
      }
      float norm = 0.0f;
      if (MODE == P3D_COMPOSITE_NORM_SUM) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (id[k] >= 0) norm += al[k];
        if (norm < kEpsNorm) norm = kEpsNorm;
      }
      // Feature gathers (round 5).  (1) All gathers of a group of four channels are requested TOGETHER, ahead of the accumulation:
      // inside the `id >= 0` branches each one waited for the one before it.  (2) What bounds this kernel on BASELINE configs[3] is
      // the gathers themselves -- 2.6 M entries x C random 4-byte reads, each a 64-byte request to the fabric: FETCH_SIZE 3.6 x the
      // algorithmic bytes -- so the features are read through their strides: PointsRenderer hands over the transposed view of a
      // (P, C) tensor, in which a point's C channels are ADJACENT (one request instead of C; the reference, and this package until
      // round 4, first copied it to (C, P) planes).  k outer, channel inner: a point's channels are requested back to back.  An
      // empty slot reads point 0 and never uses it; the arithmetic, and with it every bit of the result, is unchanged.
      constexpr int CG = IL ? 4 : 1;  // channels gathered together: a point's adjacent channels, or one plane at a time
      for (int c0 = 0; c0 < C; c0 += CG) {
        const int nc = C - c0 < CG ? C - c0 : CG;
        float fvs[CG][KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const int idc = id[k] < 0 ? 0 : id[k];
          if constexpr (IL) {
            // a point's channels with ONE request: what bounds these gathers is the number of requests the L2s take per second
            // (2.6 M entries x C at one request per lane), not bytes -- a 12 / 16-byte load per point instead of three / four
            const float* fp = a.features + (int64_t)idc * C + c0;
            if (nc == 3) {  // uniform
              const Feat3 v = k < K ? *reinterpret_cast<const Feat3*>(fp) : Feat3{0.0f, 0.0f, 0.0f};
              fvs[0][k] = v.x;
              fvs[1][k] = v.y;
              fvs[2][k] = v.z;
              fvs[3][k] = 0.0f;
            } else if (nc == 4) {
              const Feat4 v = k < K ? *reinterpret_cast<const Feat4*>(fp) : Feat4{0.0f, 0.0f, 0.0f, 0.0f};
              fvs[0][k] = v.x;
              fvs[1][k] = v.y;
              fvs[2][k] = v.z;
              fvs[3][k] = v.w;
            } else {
#pragma unroll
              for (int j = 0; j < CG; ++j) fvs[j][k] = (k < K && j < nc) ? fp[j] : 0.0f;
            }
          } else {
            fvs[0][k] = k < K ? (a.features + (int64_t)c0 * a.P)[idc] : 0.0f;
          }
        }
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          if (j < nc) {  // uniform
            float res = 0.0f;
            float cum = 1.0f;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
              if (id[k] >= 0) {
                const float fv = fvs[j][k];
                if (MODE == P3D_COMPOSITE_ALPHA) {
                  res += fv * cum * al[k];
                  cum = cum * (1 - al[k]);
                } else if (MODE == P3D_COMPOSITE_NORM_SUM) {
                  res += fv * al[k] / norm;
                } else {
                  res += fv * al[k];
                }
              }
            }
            if (valid) out[(int64_t)(c0 + j) * HW] = res;
          }
        }
      }
    } else if (valid) {
      float norm = 0.0f;
      if (MODE == P3D_COMPOSITE_NORM_SUM) {
        for (int k = 0; k < K; ++k)
          if ((int)a.idx[ibase + k * a.is[1]] >= 0) norm += a.alphas[abase + k * a.as[1]];
        if (norm < kEpsNorm) norm = kEpsNorm;
      }
      for (int c = 0; c < C; ++c) {
        const float* f = a.features + (int64_t)c * a.fs0;
        float res = 0.0f;
        float cum = 1.0f;
        for (int k = 0; k < K; ++k) {
          const int id = (int)a.idx[ibase + k * a.is[1]];
          if (id < 0) continue;
          const float al = a.alphas[abase + k * a.as[1]];
          const float fv = f[(int64_t)id * a.fs1];
          if (MODE == P3D_COMPOSITE_ALPHA) {
            res += fv * cum * al;
            cum = cum * (1 - al);
          } else if (MODE == P3D_COMPOSITE_NORM_SUM) {
            res += fv * al / norm;
          } else {
            res += fv * al;
          }
        }
        out[(int64_t)c * HW] = res;
      }
    }
  }
}

// Backward, K <= KT: a wave owns an 8x8 pixel tile, a lane one pixel.  grad_alphas is accumulated in registers in
// the (channel, k) order of a serial loop and written once.  grad_features is a scatter in which neighbouring
// pixels name the same points (a splat covers ~pi*r^2 pixels), so the per-(pixel, k) terms of four channels at
// a time are merged per point in a wave-private LDS table (wave_table.h) and reach memory as one atomic per
// (tile, point, channel) instead of one per (pixel, k, channel).
using FeatTable = WaveTable<4, 320, true>;  // 4 waves x 320 x 24 B = 30 KB

template <int MODE, int KT>
__global__ __launch_bounds__(256) void composite_bwd_tile_kernel(CompArgs a, int tiles_y, int tiles_x) {
  __shared__ __align__(16) int s_table[4][FeatTable::kLdsInts];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * 4 + w;
  const int64_t per_image = (int64_t)tiles_y * tiles_x;
  if (tile >= (int64_t)a.N * per_image) return;  // wave-uniform; no workgroup barrier in this kernel
  const int n = (int)(tile / per_image);
  const int t = (int)(tile - (int64_t)n * per_image);
  const int y = (t / tiles_x) * 8 + (lane >> 3), x = (t % tiles_x) * 8 + (lane & 7);
  const bool ok = y < a.H && x < a.W;
  const int K = a.K, C = a.C;
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t yx = (int64_t)y * a.W + x;
  const int64_t abase = n * a.as[0] + y * a.as[2] + x * a.as[3];
  const int64_t ibase = n * a.is[0] + y * a.is[2] + x * a.is[3];
  const float* go_p = a.grad_out + ((int64_t)n * C) * HW + yx;
  float* ga_p = a.grad_alphas + ((int64_t)n * K) * HW + yx;  // + k*HW

  int id[KT];
  float al[KT], ga[KT];
  if (a.sample_major && 2 * 64 * K <= FeatTable::kLdsInts) {  // uniform: the tile's eight rows of 8 K adjacent entries, transposed
    // through the wave's table memory (composite_fwd_kernel has the why; the table is not in use yet -- tab.init comes after the
    // grad_alphas part; 1920 ints: K <= 15)
    int* sid = s_table[w];
    float* sal = reinterpret_cast<float*>(s_table[w]) + 64 * K;
    const int ty0 = (t / tiles_x) * 8, tx0 = (t % tiles_x) * 8;
    const int rows = min(8, a.H - ty0), cols = min(8, a.W - tx0);
    const int run = cols * K;
    for (int r = 0; r < rows; ++r) {
      const int64_t g0 = ((int64_t)n * a.H + ty0 + r) * a.W * K + (int64_t)tx0 * K;  // element offset in the (N,H,W,K) memory
      for (int e = lane; e < run; e += 64) {
        sid[r * 8 * K + e] = (int)a.idx[g0 + e];
        sal[r * 8 * K + e] = a.alphas[g0 + e];
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      id[k] = -1;
      al[k] = 0.0f;
      ga[k] = 0.0f;
      if (k < K && ok) {
        id[k] = sid[lane * K + k];
        al[k] = sal[lane * K + k];
      }
    }
    __builtin_amdgcn_wave_barrier();
  } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      id[k] = -1;
      al[k] = 0.0f;
      ga[k] = 0.0f;
      if (k < K && ok) {
        id[k] = (int)a.idx[ibase + k * a.is[1]];
        al[k] = a.alphas[abase + k * a.as[1]];
      }
    }
  }
  float sum_alpha = 0.0f;
  if (MODE == P3D_COMPOSITE_NORM_SUM) {
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (id[k] >= 0) sum_alpha += al[k];
    if (sum_alpha < kEpsNorm) sum_alpha = kEpsNorm;
  }

  // grad_alphas: registers only
  float inv[MODE == P3D_COMPOSITE_ALPHA ? KT : 1];
  if (MODE == P3D_COMPOSITE_ALPHA) {
#pragma unroll
    for (int k = 0; k < KT; ++k) inv[k] = 1.0f / (1 - al[k] + kEpsAlpha);
  }
  if (ok) {
    // one channel's share of grad_alphas, its K feature values given
    auto channel = [&](int c, const float (&fvs)[KT]) {
      const float go = go_p[(int64_t)c * HW];
      if (MODE == P3D_COMPOSITE_ALPHA) {
        // alpha_composite.cu:120-139: entry k adds -go f_k cum_k alpha_k / (1 - alpha_t + eps) to grad_alpha[t] for every valid
        // t < k.  The reference (and this kernel until round 4) forms each of those K (K - 1) / 2 quotients per channel with an
        // IEEE division -- 135 divisions of 11 instructions per pixel at K = 10, C = 3, a dependent chain that made the
        // kernel latency-bound (0.113 ms for 69 MB).  The divisor depends on t alone: grad_alpha[t] += inv_t * (sum of the
        // `back` terms of the valid entries behind t), one reciprocal per entry (outside the channel loop) and one running sum.
        // The sums are re-associated (tolerance-gated: 1e-6 absolute in the reference's own test, tests/test_compositing.py:207).
        float cum = 1.0f;
        float back[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          back[k] = 0.0f;
          if (id[k] >= 0) {
            const float fv = fvs[k];
            ga[k] += cum * fv * go;
            back[k] = -go * fv * cum * al[k];
            cum = cum * (1 - al[k]);
          }
        }
        float behind = 0.0f;
#pragma unroll
        for (int k = KT - 1; k >= 0; --k) {
          if (id[k] >= 0) ga[k] += behind * inv[k];
          behind += back[k];
        }
      } else if (MODE == P3D_COMPOSITE_NORM_SUM) {
        float sum_af = 0.0f;
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (id[k] >= 0) sum_af += al[k] * fvs[k];
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (id[k] >= 0) ga[k] += (fvs[k] * sum_alpha - sum_af) / (sum_alpha * sum_alpha) * go;
      } else {
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (id[k] >= 0) ga[k] += fvs[k] * go;
      }
    };
    if (KT <= 16 && a.fs0 == 1 && C == 3) {  // uniform
      // the renderers' (P, 3) features: a point's three channels with ONE 12-byte request (the gathers are bound by the number of
      // requests the L2s take, see composite_fwd_kernel); channels in the serial order 0, 1, 2 as below
      float f0[KT], f1[KT], f2[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const Feat3 v = k < K ? *reinterpret_cast<const Feat3*>(a.features + (int64_t)(id[k] < 0 ? 0 : id[k]) * 3) : Feat3{0.0f, 0.0f, 0.0f};
        f0[k] = v.x;
        f1[k] = v.y;
        f2[k] = v.z;
      }
      channel(0, f0);
      channel(1, f1);
      channel(2, f2);
    } else {
      for (int c = 0; c < C; ++c) {
        const float* f = a.features + (int64_t)c * a.fs0;
        // all K gathers of the channel in flight at once (see composite_fwd_kernel); empty slots read point 0 unused
        float fvs[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) fvs[k] = k < K ? f[(int64_t)(id[k] < 0 ? 0 : id[k]) * a.fs1] : 0.0f;
        channel(c, fvs);
      }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < K) ga_p[(int64_t)k * HW] = ga[k];
  }

  // grad_features: four channels per pass through the table (wave-uniform control flow from here on)
  FeatTable tab;
  tab.init(s_table[w], lane);
  tab.plane = a.gs0;
  tab.fstride = a.gs1;
  for (int c0 = 0; c0 < C; c0 += 4) {
    const int nc = min(4, C - c0);
    tab.nlive = nc;
    float* gf = a.grad_features + (int64_t)c0 * a.gs0;
    float go[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) go[j] = (ok && j < nc) ? go_p[(int64_t)(c0 + j) * HW] : 0.0f;
    float cum = 1.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k < K) {
        float wgt;  // d result[c] / d features[c, id[k]]
        if (MODE == P3D_COMPOSITE_ALPHA)
          wgt = cum * al[k];
        else if (MODE == P3D_COMPOSITE_NORM_SUM)
          wgt = al[k];
        else
          wgt = al[k];
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MODE == P3D_COMPOSITE_NORM_SUM)
            g[j] = wgt * go[j] / sum_alpha;
          else
            g[j] = wgt * go[j];
        }
        tab.add(gf, lane, id[k], g);
        if (MODE == P3D_COMPOSITE_ALPHA && id[k] >= 0) cum = cum * (1 - al[k]);
      }
    }
    if (tab.used > 0) tab.flush(gf, lane);
  }
}

// Backward, any K: one thread per pixel, nothing cached; this thread owns grad_alphas[n, :, y, x] and accumulates
// there without atomics.
template <int MODE>
__global__ __launch_bounds__(256) void composite_bwd_generic_kernel(CompArgs a) {
  const int64_t npix = (int64_t)a.N * a.H * a.W;
  const int K = a.K, C = a.C;
  const int64_t HW = (int64_t)a.H * a.W;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npix; t += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(t / HW);
    const int64_t yx = t % HW;
    const int y = (int)(yx / a.W), x = (int)(yx % a.W);
    const int64_t abase = n * a.as[0] + y * a.as[2] + x * a.as[3];
    const int64_t ibase = n * a.is[0] + y * a.is[2] + x * a.is[3];
    const float* go_p = a.grad_out + ((int64_t)n * C) * HW + yx;
    float* ga_p = a.grad_alphas + ((int64_t)n * K) * HW + yx;  // + k*HW
    for (int k = 0; k < K; ++k) ga_p[(int64_t)k * HW] = 0.0f;
    float sum_alpha = 0.0f;
    if (MODE == P3D_COMPOSITE_NORM_SUM) {
      for (int k = 0; k < K; ++k)
        if ((int)a.idx[ibase + k * a.is[1]] >= 0) sum_alpha += a.alphas[abase + k * a.as[1]];
      if (sum_alpha < kEpsNorm) sum_alpha = kEpsNorm;
    }
    for (int c = 0; c < C; ++c) {
      const float* f = a.features + (int64_t)c * a.fs0;
      float* gf = a.grad_features + (int64_t)c * a.gs0;
      const float go = go_p[(int64_t)c * HW];
      float cum = 1.0f;
      float sum_af = 0.0f;
      if (MODE == P3D_COMPOSITE_NORM_SUM) {
        for (int k = 0; k < K; ++k) {
          const int id = (int)a.idx[ibase + k * a.is[1]];
          if (id >= 0) sum_af += a.alphas[abase + k * a.as[1]] * f[(int64_t)id * a.fs1];
        }
      }
      for (int k = 0; k < K; ++k) {
        const int id = (int)a.idx[ibase + k * a.is[1]];
        if (id < 0) continue;
        const float al = a.alphas[abase + k * a.as[1]];
        const float fv = f[(int64_t)id * a.fs1];
        if (MODE == P3D_COMPOSITE_ALPHA) {
          ga_p[(int64_t)k * HW] += cum * fv * go;
          unsafeAtomicAdd(gf + (int64_t)id * a.gs1, cum * al * go);
          const float back = -go * fv * cum * al;
          for (int tt = 0; tt < k; ++tt) {
            if ((int)a.idx[ibase + tt * a.is[1]] < 0) continue;
            ga_p[(int64_t)tt * HW] += back / (1 - a.alphas[abase + tt * a.as[1]] + kEpsAlpha);
          }
          cum = cum * (1 - al);
        } else if (MODE == P3D_COMPOSITE_NORM_SUM) {
          ga_p[(int64_t)k * HW] += (fv * sum_alpha - sum_af) / (sum_alpha * sum_alpha) * go;
          unsafeAtomicAdd(gf + (int64_t)id * a.gs1, al * go / sum_alpha);
        } else {
          ga_p[(int64_t)k * HW] += fv * go;
          unsafeAtomicAdd(gf + (int64_t)id * a.gs1, al * go);
        }
      }
    }
  }
}

template <int MODE>
int launch_fwd(const CompArgs& a, unsigned grid, hipStream_t s) {
  const bool il = a.fs0 == 1 && a.C > 1;
  if (a.K <= 8) {
    if (il)
      composite_fwd_kernel<MODE, 8, true><<<grid, 256, 0, s>>>(a);
    else
      composite_fwd_kernel<MODE, 8, false><<<grid, 256, 0, s>>>(a);
  } else if (a.K <= 16) {
    if (il)
      composite_fwd_kernel<MODE, 16, true><<<grid, 256, 0, s>>>(a);
    else
      composite_fwd_kernel<MODE, 16, false><<<grid, 256, 0, s>>>(a);
  } else {
    composite_fwd_kernel<MODE, 0><<<grid, 256, 0, s>>>(a);
  }
  return launch_status();
}

// The two feature layouts the kernels address (include/p3d_amd.h): (P, 1) planes or (1, C) rows.  A dimension of size 1 may
// carry any stride -- but the kernels of K <= 16 read the layout off (fs0, fs1) themselves, so what is accepted is NORMALISED
// to one of the two before the launch: a single channel (its P values contiguous) is one plane, a single point (its C values
// contiguous) one row.  Anything else is refused, whatever K.
bool feature_strides_ok(const int64_t st[2], int C, int64_t P, int64_t out[2]) {
  const bool planar = (st[0] == P || C <= 1) && (st[1] == 1 || P <= 1);
  const bool rows = (st[0] == 1 || C <= 1) && (st[1] == C || P <= 1);
  if (!planar && !rows) return false;
  out[0] = planar ? P : 1;
  out[1] = planar ? 1 : C;
  return true;
}

template <int MODE>
int launch_bwd(const CompArgs& a, unsigned grid, hipStream_t s) {
  const int tiles_y = (int)ceil_div(a.H, 8), tiles_x = (int)ceil_div(a.W, 8);
  const int64_t tile_blocks = ceil_div((int64_t)a.N * tiles_y * tiles_x, 4);
  if (a.K <= 32 && tile_blocks > 0x7fffffff) return P3D_ERR_INVALID_ARG;
  if (a.K <= 8)
    composite_bwd_tile_kernel<MODE, 8><<<(unsigned)tile_blocks, 256, 0, s>>>(a, tiles_y, tiles_x);
  else if (a.K <= 16)
    composite_bwd_tile_kernel<MODE, 16><<<(unsigned)tile_blocks, 256, 0, s>>>(a, tiles_y, tiles_x);
  else if (a.K <= 32)  // 96 registers of (id, alpha, grad_alpha) rows: still the table path, not per-sample atomics
    composite_bwd_tile_kernel<MODE, 32><<<(unsigned)tile_blocks, 256, 0, s>>>(a, tiles_y, tiles_x);
  else
    composite_bwd_generic_kernel<MODE><<<grid, 256, 0, s>>>(a);
  return launch_status();
}

// alphas and idx both laid out as contiguous (N,H,W,K) memory seen as (N,K,H,W) (dimensions of size 1 may carry any stride)
int sample_major_of(const int64_t as[4], const int64_t is[4], int N, int K, int H, int W) {
  const int64_t wk = (int64_t)W * K, hwk = (int64_t)H * wk;
  auto ok = [&](const int64_t* st) {
    return (K == 1 || st[1] == 1) && (W == 1 || st[3] == K) && (H == 1 || st[2] == wk) && (N == 1 || st[0] == hwk);
  };
  return (ok(as) && ok(is)) ? 1 : 0;
}

unsigned pick_grid(int64_t npix) {
  int64_t blocks = ceil_div(npix, 256);
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_composite_forward_strided(int mode, const float* features, const int64_t feature_strides[2], const float* alphas,
                                          const int64_t* points_idx, int N, int C, int64_t P, int K, int H, int W,
                                          const int64_t alphas_strides[4], const int64_t idx_strides[4], float* result,
                                          p3d_stream_t stream) {
  if (mode < 0 || mode > 2 || N < 0 || C < 0 || K < 0 || H < 0 || W < 0 || P < 0) return P3D_ERR_INVALID_ARG;
  const int64_t nout = (int64_t)N * C * H * W;
  if (nout == 0) return P3D_OK;
  if (!result || !alphas_strides || !idx_strides || !feature_strides) return P3D_ERR_INVALID_ARG;
  int64_t fst[2];
  if (!feature_strides_ok(feature_strides, C, P, fst)) return P3D_ERR_INVALID_ARG;
  if (K > 0 && (!alphas || !points_idx || !features)) return P3D_ERR_INVALID_ARG;
  if (P == 0) {  // no point: every slot is empty and every compositor returns zeros (the kernels' empty slots gather point 0)
    return hipMemsetAsync(result, 0, (size_t)nout * sizeof(float), (hipStream_t)stream) == hipSuccess ? P3D_OK : P3D_ERR_LAUNCH;
  }
  CompArgs a{};
  a.features = features;
  a.fs0 = fst[0];
  a.fs1 = fst[1];
  a.alphas = alphas;
  a.idx = points_idx;
  a.N = N;
  a.C = C;
  a.K = K;
  a.H = H;
  a.W = W;
  a.P = P;
  for (int i = 0; i < 4; ++i) {
    a.as[i] = alphas_strides[i];
    a.is[i] = idx_strides[i];
  }
  a.result = result;
  a.sample_major = sample_major_of(a.as, a.is, N, K, H, W);
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = pick_grid((int64_t)N * H * W);
  LaunchScope ls(mode == 0 ? "alpha_composite_fwd" : (mode == 1 ? "norm_weighted_sum_fwd" : "weighted_sum_fwd"), s);
  if (mode == P3D_COMPOSITE_ALPHA) return launch_fwd<P3D_COMPOSITE_ALPHA>(a, grid, s);
  if (mode == P3D_COMPOSITE_NORM_SUM) return launch_fwd<P3D_COMPOSITE_NORM_SUM>(a, grid, s);
  return launch_fwd<P3D_COMPOSITE_SUM>(a, grid, s);
}

P3D_API int p3d_composite_forward(int mode, const float* features, const float* alphas, const int64_t* points_idx, int N,
                                  int C, int64_t P, int K, int H, int W, const int64_t alphas_strides[4],
                                  const int64_t idx_strides[4], float* result, p3d_stream_t stream) {
  const int64_t planar[2] = {P, 1};
  return p3d_composite_forward_strided(mode, features, planar, alphas, points_idx, N, C, P, K, H, W, alphas_strides, idx_strides, result,
                                       stream);
}

P3D_API int p3d_composite_backward_strided(int mode, const float* grad_outputs, const float* features, const int64_t feature_strides[2],
                                           const float* alphas, const int64_t* points_idx, int N, int C, int64_t P, int K, int H, int W,
                                           const int64_t alphas_strides[4], const int64_t idx_strides[4], float* grad_features,
                                           const int64_t grad_feature_strides[2], float* grad_alphas, p3d_stream_t stream) {
  if (mode < 0 || mode > 2 || N < 0 || C < 0 || K < 0 || H < 0 || W < 0 || P < 0) return P3D_ERR_INVALID_ARG;
  if (!feature_strides || !grad_feature_strides) return P3D_ERR_INVALID_ARG;
  int64_t fst[2], gst[2];
  if (!feature_strides_ok(feature_strides, C, P, fst) || !feature_strides_ok(grad_feature_strides, C, P, gst)) return P3D_ERR_INVALID_ARG;
  // grad_features: C * P floats of ONE allocation in either layout ((P, 1) planes or (1, C) rows): zeroed as a block
  hipStream_t s = (hipStream_t)stream;
  if ((int64_t)C * P > 0) {
    if (!grad_features) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_features, 0, (size_t)C * P * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  const int64_t nga = (int64_t)N * K * H * W;
  if (nga == 0) return P3D_OK;
  if (!grad_alphas || !alphas || !points_idx || !alphas_strides || !idx_strides) return P3D_ERR_INVALID_ARG;
  if (C == 0) {
    if (hipMemsetAsync(grad_alphas, 0, (size_t)nga * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
    return P3D_OK;
  }
  if (!grad_outputs || !features) return P3D_ERR_INVALID_ARG;
  if (P == 0) {  // no point: every slot is empty, nothing flows to the alphas
    if (hipMemsetAsync(grad_alphas, 0, (size_t)nga * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
    return P3D_OK;
  }
  CompArgs a{};
  a.features = features;
  a.fs0 = fst[0];
  a.fs1 = fst[1];
  a.gs0 = gst[0];
  a.gs1 = gst[1];
  a.alphas = alphas;
  a.idx = points_idx;
  a.grad_out = grad_outputs;
  a.N = N;
  a.C = C;
  a.K = K;
  a.H = H;
  a.W = W;
  a.P = P;
  for (int i = 0; i < 4; ++i) {
    a.as[i] = alphas_strides[i];
    a.is[i] = idx_strides[i];
  }
  a.grad_features = grad_features;
  a.grad_alphas = grad_alphas;
  a.sample_major = sample_major_of(a.as, a.is, N, K, H, W);
  const unsigned grid = pick_grid((int64_t)N * H * W);
  LaunchScope ls(mode == 0 ? "alpha_composite_bwd" : (mode == 1 ? "norm_weighted_sum_bwd" : "weighted_sum_bwd"), s);
  if (mode == P3D_COMPOSITE_ALPHA) return launch_bwd<P3D_COMPOSITE_ALPHA>(a, grid, s);
  if (mode == P3D_COMPOSITE_NORM_SUM) return launch_bwd<P3D_COMPOSITE_NORM_SUM>(a, grid, s);
  return launch_bwd<P3D_COMPOSITE_SUM>(a, grid, s);
}

P3D_API int p3d_composite_backward(int mode, const float* grad_outputs, const float* features, const float* alphas,
                                   const int64_t* points_idx, int N, int C, int64_t P, int K, int H, int W,
                                   const int64_t alphas_strides[4], const int64_t idx_strides[4], float* grad_features,
                                   float* grad_alphas, p3d_stream_t stream) {
  const int64_t planar[2] = {P, 1};
  return p3d_composite_backward_strided(mode, grad_outputs, features, planar, alphas, points_idx, N, C, P, K, H, W, alphas_strides,
                                        idx_strides, grad_features, planar, grad_alphas, stream);
}
