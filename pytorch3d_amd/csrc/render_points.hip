// render_points.hip -- the backward of PointsRenderer's chain as ONE kernel (round 6; forward: raster_points.hip,
// p3d_rasterize_points_composite).
//
// The chain (pytorch3d/renderer/points/renderer.py:56-76): fragments = rasterize_points(...); weights = 1 - dists / r^2;
// images = alpha_composite(idx, weights, features).  Its backward in the reference is alphaCompositeCudaBackwardKernel
// (alpha_composite.cu:72-141: grad_features by atomics, grad_alphas), two torch element-wise kernels (grad_dists = -grad_alphas / r^2)
// and RasterizePointsBackwardCudaKernel (rasterize_points.cu:366-411: grad_points by atomics).  Both scatters name the SAME point per
// (pixel, k) entry, so here a wave owns an 8x8 pixel tile, forms grad_alphas in registers (composite.hip: composite_bwd_tile_kernel's
// arithmetic), turns it into the entry's (d/dx, d/dy) and feature gradients and merges all 2 + C values per point in ONE wave-private
// LDS table (wave_table.h, kSplit): one atomic group per (tile, point) into grad_points and grad_features, no (N,K,H,W) intermediate.
#include "p3d_common.h"
#include "p3d_geom.h"
#include "wave_table.h"

namespace p3d {
namespace {

constexpr float kEpsAlpha = 1e-9f;  // alpha_composite.cu:20
constexpr float kEpsNorm = 1e-4f;   // norm_weighted_sum.cu:20
constexpr int kSplatKT = 16;        // entries per pixel held in registers
constexpr int kSplatSlots = 240;    // 4 waves x 240 x 40 B = 38.4 KB: four workgroups per CU (one 512^2 image is 1024 of them)

struct __attribute__((packed, aligned(4))) RFeat3 {
  float x, y, z;
};
struct __attribute__((packed, aligned(4))) RFeat4 {
  float x, y, z, w;
};
struct __attribute__((packed, aligned(4))) RXY {
  float x, y;
};

struct SplatBwdArgs {
  const float* points;       // (P, 3) NDC
  const float* features;     // (P, C)
  const int32_t* idxs;       // (N, H, W, K)
  const float* dists;        // (N, H, W, K)
  const float* grad_images;  // (N, H, W, C)
  int N, H, W, K;
  float inv_r2;
  int tiles_y, tiles_x;
  float* grad_points;        // (P, 3), zeroed by the launcher
  float* grad_features;      // (P, C), zeroed by the launcher
};

// KT: entries per pixel held in registers (K <= KT); up to 12 the kernel is held to 128 registers: four workgroups per CU
// MODE: P3D_COMPOSITE_ALPHA, or P3D_COMPOSITE_NORM_SUM (NormWeightedCompositor: norm_weighted_sum.cu:82-154 in composite.hip's form)
template <int C, int KT, int MODE>
__global__ __launch_bounds__(256, KT <= 12 ? 4 : 2) void splat_backward_kernel(SplatBwdArgs a) {
  using Tab = WaveTable<2 + C, kSplatSlots, kSplit>;
  // (the tile's entries are transposed through the table's memory before the table is in use)
  constexpr int kInts = Tab::kLdsInts > 2 * 64 * KT ? Tab::kLdsInts : 2 * 64 * KT;
  __shared__ __align__(16) int s_table[4][kInts];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * 4 + w;
  const int64_t per_image = (int64_t)a.tiles_y * a.tiles_x;
  if (tile >= (int64_t)a.N * per_image) return;  // wave-uniform; no workgroup barrier in this kernel
  const int n = (int)(tile / per_image);
  const int t = (int)(tile - (int64_t)n * per_image);
  const int ty0 = (t / a.tiles_x) * 8, tx0 = (t % a.tiles_x) * 8;
  const int y = ty0 + (lane >> 3), x = tx0 + (lane & 7);
  const bool ok = y < a.H && x < a.W;
  const int K = a.K, H = a.H, W = a.W;
  const int64_t pixel = ((int64_t)n * H + y) * W + x;

  // ---- the tile's entries: eight rows of 8 K adjacent entries, loaded 64 consecutive entries per instruction and transposed
  // through the wave's table memory (composite.hip: composite_fwd_kernel has the why); the table is not in use yet
  int id[KT];
  float al[KT], ga[KT];
  {
    int* sid = s_table[w];
    float* sd = reinterpret_cast<float*>(s_table[w]) + 64 * K;
    const int rows = min(8, H - ty0), cols = min(8, W - tx0);
    const int run = cols * K;
    for (int r = 0; r < rows; ++r) {
      const int64_t g0 = (((int64_t)n * H + ty0 + r) * W + tx0) * K;
      for (int e = lane; e < run; e += 64) {
        sid[r * 8 * K + e] = a.idxs[g0 + e];
        sd[r * 8 * K + e] = a.dists[g0 + e];
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
        al[k] = 1.0f - sd[lane * K + k] * a.inv_r2;  // renderer.py:62-64, as torch evaluates it (raster_points.hip: SplatPixel)
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  float go[C];
  {
    const float* gp = a.grad_images + pixel * C;
#pragma unroll
    for (int c = 0; c < C; ++c) go[c] = 0.0f;
    if (ok) {
      if constexpr (C == 3) {
        const RFeat3 v = *reinterpret_cast<const RFeat3*>(gp);
        go[0] = v.x, go[1] = v.y, go[2] = v.z;
      } else if constexpr (C == 4) {
        const RFeat4 v = *reinterpret_cast<const RFeat4*>(gp);
        go[0] = v.x, go[1] = v.y, go[2] = v.z, go[3] = v.w;
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) go[c] = gp[c];
      }
    }
  }

  float sum_alpha = 0.0f;
  if constexpr (MODE == P3D_COMPOSITE_NORM_SUM) {
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (id[k] >= 0) sum_alpha += al[k];
    if (sum_alpha < kEpsNorm) sum_alpha = kEpsNorm;
  }

  // ---- grad_alphas in registers: alpha_composite.cu:120-139 in composite_bwd_tile_kernel's form (one reciprocal per entry, the
  // terms behind an entry as one running sum; tolerance-gated, tests/test_compositing.py:207)
  {
    float fv[C][KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {  // all gathers requested together; an empty slot reads point 0 and never uses it
      const float* fp = a.features + (int64_t)(id[k] < 0 ? 0 : id[k]) * C;
      if constexpr (C == 3) {
        const RFeat3 v = k < K ? *reinterpret_cast<const RFeat3*>(fp) : RFeat3{0.0f, 0.0f, 0.0f};
        fv[0][k] = v.x, fv[1][k] = v.y, fv[2][k] = v.z;
      } else if constexpr (C == 4) {
        const RFeat4 v = k < K ? *reinterpret_cast<const RFeat4*>(fp) : RFeat4{0.0f, 0.0f, 0.0f, 0.0f};
        fv[0][k] = v.x, fv[1][k] = v.y, fv[2][k] = v.z, fv[3][k] = v.w;
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) fv[c][k] = k < K ? fp[c] : 0.0f;
      }
    }
    if constexpr (MODE == P3D_COMPOSITE_ALPHA) {
      float inv[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) inv[k] = 1.0f / (1 - al[k] + kEpsAlpha);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float cum = 1.0f;
        float back[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          back[k] = 0.0f;
          if (id[k] >= 0) {
            ga[k] += cum * fv[c][k] * go[c];
            back[k] = -go[c] * fv[c][k] * cum * al[k];
            cum = cum * (1 - al[k]);
          }
        }
        float behind = 0.0f;
#pragma unroll
        for (int k = KT - 1; k >= 0; --k) {
          if (id[k] >= 0) ga[k] += behind * inv[k];
          behind += back[k];
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float sum_af = 0.0f;
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (id[k] >= 0) sum_af += al[k] * fv[c][k];
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (id[k] >= 0) ga[k] += (fv[c][k] * sum_alpha - sum_af) / (sum_alpha * sum_alpha) * go[c];
      }
    }
  }

  // ---- per entry: the point's (d/dx, d/dy) (rasterize_points.cu:389-405 with grad_dists = -grad_alpha * inv_r2; the stored pixel
  // is un-flipped for its centre) and its C feature gradients (alpha_composite.cu:113), merged per point in the wave's table
  float qx[KT], qy[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    qx[k] = qy[k] = 0.0f;
    if (k < K && id[k] >= 0) {
      const RXY q = *reinterpret_cast<const RXY*>(a.points + (int64_t)id[k] * 3);
      qx[k] = q.x;
      qy[k] = q.y;
    }
  }
  const float xf = pix_to_ndc(W - 1 - x, W, H), yf = pix_to_ndc(H - 1 - y, H, W);
  Tab tab;
  tab.init(s_table[w], lane);
  tab.out2 = a.grad_features;
  tab.split_at = 2;
  tab.split_row = 3;
  float cum = 1.0f;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < K) {  // uniform
      float g[2 + C];
      const float gd = -ga[k] * a.inv_r2;
      g[0] = 2.0f * gd * (qx[k] - xf);
      g[1] = 2.0f * gd * (qy[k] - yf);
      if constexpr (MODE == P3D_COMPOSITE_ALPHA) {
        const float wgt = cum * al[k];
#pragma unroll
        for (int c = 0; c < C; ++c) g[2 + c] = wgt * go[c];
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) g[2 + c] = al[k] * go[c] / sum_alpha;
      }
      tab.add(a.grad_points, lane, id[k], g);
      if (MODE == P3D_COMPOSITE_ALPHA && id[k] >= 0) cum = cum * (1 - al[k]);
    }
  }
  if (tab.used > 0) tab.flush(a.grad_points, lane);
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_rasterize_points_composite_backward(int mode, const float* points, const float* features, const int32_t* idxs,
                                                    const float* dists, const float* grad_images, int64_t P, int C, int N, int H, int W,
                                                    int K, float inv_r2, float* grad_points, float* grad_features, p3d_stream_t stream) {
  if (P < 0 || N < 0 || H < 0 || W < 0 || K < 0 || C < 1 || C > 4 || K > kSplatKT) return P3D_ERR_INVALID_ARG;
  if (mode != P3D_COMPOSITE_ALPHA && mode != P3D_COMPOSITE_NORM_SUM) return P3D_ERR_INVALID_ARG;
  if (P == 0) return P3D_OK;
  if (!grad_points || !grad_features) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_points, 0, (size_t)P * 3 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if (hipMemsetAsync(grad_features, 0, (size_t)P * C * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if (!points || !features || !idxs || !dists || !grad_images) return P3D_ERR_INVALID_ARG;
  SplatBwdArgs a;
  a.points = points;
  a.features = features;
  a.idxs = idxs;
  a.dists = dists;
  a.grad_images = grad_images;
  a.N = N, a.H = H, a.W = W, a.K = K;
  a.inv_r2 = inv_r2;
  a.tiles_y = (int)ceil_div(H, 8), a.tiles_x = (int)ceil_div(W, 8);
  a.grad_points = grad_points;
  a.grad_features = grad_features;
  const int64_t blocks = ceil_div((int64_t)N * a.tiles_y * a.tiles_x, 4);
  if (blocks > 0x7fffffff) return P3D_ERR_INVALID_ARG;
  LaunchScope ls("points_composite_bwd", s);
#define P3D_SPLAT_BWD_M(C_, M_)                                                                  \
  if (K <= 4) splat_backward_kernel<C_, 4, M_><<<(unsigned)blocks, 256, 0, s>>>(a);              \
  else if (K <= 8) splat_backward_kernel<C_, 8, M_><<<(unsigned)blocks, 256, 0, s>>>(a);         \
  else if (K <= 12) splat_backward_kernel<C_, 12, M_><<<(unsigned)blocks, 256, 0, s>>>(a);       \
  else splat_backward_kernel<C_, 16, M_><<<(unsigned)blocks, 256, 0, s>>>(a);
#define P3D_SPLAT_BWD(C_)                                                         \
  if (mode == P3D_COMPOSITE_ALPHA) {                                              \
    P3D_SPLAT_BWD_M(C_, P3D_COMPOSITE_ALPHA)                                      \
  } else {                                                                        \
    P3D_SPLAT_BWD_M(C_, P3D_COMPOSITE_NORM_SUM)                                   \
  }
  switch (C) {
    case 1: P3D_SPLAT_BWD(1) break;
    case 2: P3D_SPLAT_BWD(2) break;
    case 3: P3D_SPLAT_BWD(3) break;
    default: P3D_SPLAT_BWD(4) break;
  }
#undef P3D_SPLAT_BWD
#undef P3D_SPLAT_BWD_M
  return launch_status();
}
