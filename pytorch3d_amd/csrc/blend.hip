// blend.hip -- fragment blending for gfx950 (the step right after rasterization).
//
//   sigmoid_alpha_blend forward / backward  replaces SigmoidAlphaBlend{Forward,Backward}Kernel
//       (pytorch3d/csrc/blending/sigmoid_alpha_blend.cu:16-67, 109-166; `pytorch3d._C.sigmoid_alpha_blend[_backward]`)
//   softmax_rgb_blend forward / backward    replaces ~20 elementwise torch ops over (N,H,W,K) tensors
//       (pytorch3d/renderer/blending.py:147-244) and their autograd graph with one streaming kernel each.
//
// All four are pure HBM streams: one lane per pixel reads its K-rows of pix_to_face / dists / zbuf / colors
// with 16-byte loads (28 B per (pixel, k)), keeps everything in registers and writes each output once.
// The arithmetic follows the reference's float chain step by step -- (z_inv - z_inv_max) / gamma amplifies
// rounding by 1/gamma (1e4 at the default BlendParams), so a "more accurate" evaluation would NOT match.
#include "p3d_common.h"

#include <math.h>

namespace p3d {
namespace {

constexpr int kBlendBlock = 256;

struct BlendArgs {
  const float* colors;   // (P,K,3)
  const int64_t* p2f;    // (P,K)
  const float* dists;    // (P,K)
  const float* zbuf;     // (P,K)
  const float* grad_out; // (P,4)            backward only
  float sigma, gamma;
  float bg0, bg1, bg2;
  float znear, zfar;                 // used when the per-image arrays are null
  const float* znear_n;              // (N) or null
  const float* zfar_n;               // (N) or null
  int64_t npix, pix_per_image;
  int K;
  float* out;       // (P,4)                 forward
  float* g_colors;  // (P,K,3)               backward
  float* g_dists;   // (P,K)
  float* g_zbuf;    // (P,K)
};

// ---- row loaders: KT elements, vectorised when the row is a multiple of 16 bytes ------------------
template <int KT>
__device__ __forceinline__ void load_i64_row(const int64_t* p, int K, bool (&valid)[KT]) {
  if (K == KT && KT % 2 == 0) {
#pragma unroll
    for (int k = 0; k < KT; k += 2) {
      const longlong2 t = *reinterpret_cast<const longlong2*>(p + k);
      valid[k] = t.x >= 0;
      valid[k + 1] = t.y >= 0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) valid[k] = k < K ? p[k] >= 0 : false;
  }
}

template <int M>
__device__ __forceinline__ void load_f32_row(const float* p, int m, float (&v)[M]) {
  if (m == M && M % 4 == 0) {
#pragma unroll
    for (int k = 0; k < M; k += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + k);
      v[k] = t.x;
      v[k + 1] = t.y;
      v[k + 2] = t.z;
      v[k + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < M; ++k) v[k] = k < m ? p[k] : 0.0f;
  }
}

template <int M>
__device__ __forceinline__ void store_f32_row(float* p, int m, const float (&v)[M]) {
  if (m == M && M % 4 == 0) {
#pragma unroll
    for (int k = 0; k < M; k += 4) *reinterpret_cast<float4*>(p + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < M; ++k)
      if (k < m) p[k] = v[k];
  }
}

// ---- sigmoid_alpha_blend ---------------------------------------------------------------------------
// prob = float(1. / (1. + double(expf(-dist / sigma)))), alpha = float(alpha * (1.0 - prob)): the double
// literals of the reference promote each step (sigmoid_alpha_blend.cu:57-63).
__device__ __forceinline__ float sig_prob(float d, float sigma) {
  const float dist = -1.0f * d;
  return (float)(1. / (1. + (double)expf(-dist / sigma)));
}

template <int KT>
__global__ __launch_bounds__(kBlendBlock) void sigmoid_alpha_fwd_kernel(const float* __restrict__ dists,
                                                                       const int64_t* __restrict__ p2f, float sigma,
                                                                       int64_t npix, int K, float* __restrict__ alphas) {
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kBlendBlock) {
    bool valid[KT];
    float d[KT];
    load_i64_row<KT>(p2f + i * K, K, valid);
    load_f32_row<KT>(dists + i * K, K, d);
    float alpha = 1.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (valid[k]) alpha = (float)(alpha * (1.0 - sig_prob(d[k], sigma)));
    alphas[i] = (float)(1.0 - alpha);
  }
}

template <int KT>
__global__ __launch_bounds__(kBlendBlock) void sigmoid_alpha_bwd_kernel(const float* __restrict__ grad_alphas,
                                                                       const float* __restrict__ alphas,
                                                                       const float* __restrict__ dists,
                                                                       const int64_t* __restrict__ p2f, float sigma,
                                                                       int64_t npix, int K,
                                                                       float* __restrict__ grad_dists) {
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kBlendBlock) {
    bool valid[KT];
    float d[KT], g[KT];
    load_i64_row<KT>(p2f + i * K, K, valid);
    load_f32_row<KT>(dists + i * K, K, d);
    const float alpha = (float)(1.0 - alphas[i]);
    const float ga = grad_alphas[i];
#pragma unroll
    for (int k = 0; k < KT; ++k)
      g[k] = valid[k] ? (float)(ga * (-1.0 / sigma) * sig_prob(d[k], sigma) * alpha) : 0.0f;
    store_f32_row<KT>(grad_dists + i * K, K, g);
  }
}

// ---- softmax_rgb_blend -----------------------------------------------------------------------------
// Per-pixel state shared by forward and backward (blending.py:195-232, evaluated in float like torch does).
template <int KT>
struct SoftmaxPixel {
  float m[KT], s[KT], p[KT], zi[KT], e[KT], w[KT];
  float zraw, zmax, delta, denom;
  int kstar;
  bool z_clamped, d_clamped;

  __device__ __forceinline__ void eval(const bool (&valid)[KT], const float (&d)[KT], const float (&z)[KT], float sigma,
                                       float gamma, float zn, float zf) {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      m[k] = valid[k] ? 1.0f : 0.0f;
      s[k] = 1.0f / (1.0f + expf(d[k] / sigma));  // torch.sigmoid(-dists / sigma)
      p[k] = s[k] * m[k];
      zi[k] = (zf - z[k]) / (zf - zn) * m[k];
    }
    // padded slots beyond K hold m = 0, d = z = 0, i.e. zi = 0 like a masked slot; they must not take part in
    // the max (torch.max runs over the K real slots): softmax_state() stops at K
  }
};

template <int KT>
__device__ __forceinline__ void softmax_state(SoftmaxPixel<KT>& q, int K, float gamma) {
  const float eps = 1e-10f;
  q.zraw = -INFINITY;
  q.kstar = 0;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < K && q.zi[k] > q.zraw) {
      q.zraw = q.zi[k];
      q.kstar = k;
    }
  }
  q.z_clamped = q.zraw < eps;
  q.zmax = q.z_clamped ? eps : q.zraw;
  q.delta = expf((eps - q.zmax) / gamma);
  q.d_clamped = q.delta < eps;
  if (q.d_clamped) q.delta = eps;
  q.denom = 0.0f;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    q.e[k] = expf((q.zi[k] - q.zmax) / gamma);
    q.w[k] = k < K ? q.p[k] * q.e[k] : 0.0f;
    q.denom += q.w[k];
  }
  q.denom += q.delta;
}

template <int KT>
__global__ __launch_bounds__(kBlendBlock) void softmax_blend_fwd_kernel(BlendArgs a) {
  const int K = a.K;
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < a.npix; i += (int64_t)gridDim.x * kBlendBlock) {
    const int64_t n = i / a.pix_per_image;
    const float zn = a.znear_n ? a.znear_n[n] : a.znear;
    const float zf = a.zfar_n ? a.zfar_n[n] : a.zfar;
    bool valid[KT];
    float d[KT], z[KT], c[3 * KT];
    load_i64_row<KT>(a.p2f + i * K, K, valid);
    bool any = false;
#pragma unroll
    for (int k = 0; k < KT; ++k) any |= valid[k];
    if (!any && zf != zn) {
      // A pixel without a face (3 of 5 at the bench workload): every probability and weight below is exactly 0, the
      // masked inverse depths are 0 < eps, so delta = exp(0) = 1 = denom and the result is the background with alpha 0
      // -- known without reading the pixel's distances, depths and colours (160 of its 224 input bytes at K = 8).
      *reinterpret_cast<float4*>(a.out + i * 4) = make_float4(a.bg0, a.bg1, a.bg2, 0.0f);
      continue;
    }
    load_f32_row<KT>(a.dists + i * K, K, d);
    load_f32_row<KT>(a.zbuf + i * K, K, z);
    load_f32_row<3 * KT>(a.colors + i * K * 3, 3 * K, c);
    SoftmaxPixel<KT> q;
    q.eval(valid, d, z, a.sigma, a.gamma, zn, zf);
    softmax_state<KT>(q, K, a.gamma);
    float alpha = 1.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k < K) {
        alpha *= 1.0f - q.p[k];
        r0 += q.w[k] * c[3 * k];
        r1 += q.w[k] * c[3 * k + 1];
        r2 += q.w[k] * c[3 * k + 2];
      }
    }
    float4 o;
    o.x = (r0 + q.delta * a.bg0) / q.denom;
    o.y = (r1 + q.delta * a.bg1) / q.denom;
    o.z = (r2 + q.delta * a.bg2) / q.denom;
    o.w = 1.0f - alpha;
    *reinterpret_cast<float4*>(a.out + i * 4) = o;
  }
}

template <int KT>
__global__ __launch_bounds__(kBlendBlock) void softmax_blend_bwd_kernel(BlendArgs a) {
  const int K = a.K;
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < a.npix; i += (int64_t)gridDim.x * kBlendBlock) {
    const int64_t n = i / a.pix_per_image;
    const float zn = a.znear_n ? a.znear_n[n] : a.znear;
    const float zf = a.zfar_n ? a.zfar_n[n] : a.zfar;
    bool valid[KT];
    float d[KT], z[KT], c[3 * KT];
    load_i64_row<KT>(a.p2f + i * K, K, valid);
    bool any = false;
#pragma unroll
    for (int k = 0; k < KT; ++k) any |= valid[k];
    if (!any && zf != zn) {
      // no face: the image does not depend on this pixel's fragments (see the forward kernel) -- zero rows
      float zr[3 * KT];
#pragma unroll
      for (int k = 0; k < 3 * KT; ++k) zr[k] = 0.0f;
      store_f32_row<3 * KT>(a.g_colors + i * K * 3, 3 * K, zr);
      float zk[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) zk[k] = 0.0f;
      store_f32_row<KT>(a.g_dists + i * K, K, zk);
      store_f32_row<KT>(a.g_zbuf + i * K, K, zk);
      continue;
    }
    load_f32_row<KT>(a.dists + i * K, K, d);
    load_f32_row<KT>(a.zbuf + i * K, K, z);
    load_f32_row<3 * KT>(a.colors + i * K * 3, 3 * K, c);
    const float4 g = *reinterpret_cast<const float4*>(a.grad_out + i * 4);
    SoftmaxPixel<KT> q;
    q.eval(valid, d, z, a.sigma, a.gamma, zn, zf);
    softmax_state<KT>(q, K, a.gamma);
    float r0 = q.delta * a.bg0, r1 = q.delta * a.bg1, r2 = q.delta * a.bg2;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      r0 += q.w[k] * c[3 * k];
      r1 += q.w[k] * c[3 * k + 1];
      r2 += q.w[k] * c[3 * k + 2];
    }
    const float inv_denom = 1.0f / q.denom;
    r0 *= inv_denom;
    r1 *= inv_denom;
    r2 *= inv_denom;
    const float G_delta = (g.x * (a.bg0 - r0) + g.y * (a.bg1 - r1) + g.z * (a.bg2 - r2)) * inv_denom;
    float G_zmax = q.d_clamped ? 0.0f : G_delta * (-q.delta / a.gamma);
    float Gw[KT], gc[3 * KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      Gw[k] = (g.x * (c[3 * k] - r0) + g.y * (c[3 * k + 1] - r1) + g.z * (c[3 * k + 2] - r2)) * inv_denom;
      const float wd = q.w[k] * inv_denom;
      gc[3 * k] = g.x * wd;
      gc[3 * k + 1] = g.y * wd;
      gc[3 * k + 2] = g.z * wd;
      G_zmax += Gw[k] * (-q.w[k] / a.gamma);
    }
    // prefix / suffix products of (1 - p): others[k] = prod_{j != k} (1 - p_j)
    float pre[KT], gd[KT], gz[KT];
    float run = 1.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      pre[k] = run;
      run *= 1.0f - q.p[k];
    }
    run = 1.0f;
    const float inv_range = 1.0f / (zf - zn);
#pragma unroll
    for (int k = KT - 1; k >= 0; --k) {
      const float others = pre[k] * run;
      run *= 1.0f - q.p[k];
      float G_zi = Gw[k] * q.w[k] / a.gamma;
      if (k == q.kstar && !q.z_clamped) G_zi += G_zmax;
      gz[k] = G_zi * (-q.m[k] * inv_range);
      const float G_p = Gw[k] * q.e[k] + g.w * others;
      gd[k] = G_p * (-(1.0f / a.sigma) * q.s[k] * (1.0f - q.s[k]) * q.m[k]);
    }
    store_f32_row<3 * KT>(a.g_colors + i * K * 3, 3 * K, gc);
    store_f32_row<KT>(a.g_dists + i * K, K, gd);
    store_f32_row<KT>(a.g_zbuf + i * K, K, gz);
  }
}

// ---- any K (> 32): the same arithmetic with loops over memory instead of register arrays ------------
__global__ __launch_bounds__(kBlendBlock) void sigmoid_alpha_fwd_generic(const float* __restrict__ dists,
                                                                        const int64_t* __restrict__ p2f, float sigma,
                                                                        int64_t npix, int K, float* __restrict__ alphas) {
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kBlendBlock) {
    float alpha = 1.0f;
    for (int k = 0; k < K; ++k)
      if (p2f[i * K + k] >= 0) alpha = (float)(alpha * (1.0 - sig_prob(dists[i * K + k], sigma)));
    alphas[i] = (float)(1.0 - alpha);
  }
}

__global__ __launch_bounds__(kBlendBlock) void sigmoid_alpha_bwd_generic(const float* __restrict__ grad_alphas,
                                                                        const float* __restrict__ alphas,
                                                                        const float* __restrict__ dists,
                                                                        const int64_t* __restrict__ p2f, float sigma,
                                                                        int64_t npix, int K,
                                                                        float* __restrict__ grad_dists) {
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kBlendBlock) {
    const float alpha = (float)(1.0 - alphas[i]);
    const float ga = grad_alphas[i];
    for (int k = 0; k < K; ++k)
      grad_dists[i * K + k] =
          p2f[i * K + k] >= 0 ? (float)(ga * (-1.0 / sigma) * sig_prob(dists[i * K + k], sigma) * alpha) : 0.0f;
  }
}

struct SoftK {
  float m, s, p, zi;
};
__device__ __forceinline__ SoftK soft_k(const BlendArgs& a, int64_t j, float zn, float zf) {
  SoftK r;
  r.m = a.p2f[j] >= 0 ? 1.0f : 0.0f;
  r.s = 1.0f / (1.0f + expf(a.dists[j] / a.sigma));
  r.p = r.s * r.m;
  r.zi = (zf - a.zbuf[j]) / (zf - zn) * r.m;
  return r;
}

template <bool BACKWARD>
__global__ __launch_bounds__(kBlendBlock) void softmax_blend_generic(BlendArgs a) {
  const int K = a.K;
  const float eps = 1e-10f;
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < a.npix; i += (int64_t)gridDim.x * kBlendBlock) {
    const int64_t n = i / a.pix_per_image;
    const float zn = a.znear_n ? a.znear_n[n] : a.znear;
    const float zf = a.zfar_n ? a.zfar_n[n] : a.zfar;
    float zraw = -INFINITY, alpha = 1.0f;
    int kstar = 0;
    for (int k = 0; k < K; ++k) {
      const SoftK q = soft_k(a, i * K + k, zn, zf);
      alpha *= 1.0f - q.p;
      if (q.zi > zraw) {
        zraw = q.zi;
        kstar = k;
      }
    }
    const bool z_clamped = zraw < eps;
    const float zmax = z_clamped ? eps : zraw;
    float delta = expf((eps - zmax) / a.gamma);
    const bool d_clamped = delta < eps;
    if (d_clamped) delta = eps;
    float denom = 0.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
    for (int k = 0; k < K; ++k) {
      const SoftK q = soft_k(a, i * K + k, zn, zf);
      const float w = q.p * expf((q.zi - zmax) / a.gamma);
      denom += w;
      const float* c = a.colors + (i * K + k) * 3;
      r0 += w * c[0];
      r1 += w * c[1];
      r2 += w * c[2];
    }
    denom += delta;
    r0 = (r0 + delta * a.bg0) / denom;
    r1 = (r1 + delta * a.bg1) / denom;
    r2 = (r2 + delta * a.bg2) / denom;
    if (!BACKWARD) {
      *reinterpret_cast<float4*>(a.out + i * 4) = make_float4(r0, r1, r2, 1.0f - alpha);
      continue;
    }
    const float4 g = *reinterpret_cast<const float4*>(a.grad_out + i * 4);
    const float inv_denom = 1.0f / denom;
    const float G_delta = (g.x * (a.bg0 - r0) + g.y * (a.bg1 - r1) + g.z * (a.bg2 - r2)) * inv_denom;
    float G_zmax = d_clamped ? 0.0f : G_delta * (-delta / a.gamma);
    for (int k = 0; k < K; ++k) {
      const SoftK q = soft_k(a, i * K + k, zn, zf);
      const float w = q.p * expf((q.zi - zmax) / a.gamma);
      const float* c = a.colors + (i * K + k) * 3;
      const float Gw = (g.x * (c[0] - r0) + g.y * (c[1] - r1) + g.z * (c[2] - r2)) * inv_denom;
      G_zmax += Gw * (-w / a.gamma);
    }
    for (int k = 0; k < K; ++k) {
      const SoftK q = soft_k(a, i * K + k, zn, zf);
      const float e = expf((q.zi - zmax) / a.gamma);
      const float w = q.p * e;
      const float* c = a.colors + (i * K + k) * 3;
      const float Gw = (g.x * (c[0] - r0) + g.y * (c[1] - r1) + g.z * (c[2] - r2)) * inv_denom;
      float* gc = a.g_colors + (i * K + k) * 3;
      gc[0] = g.x * w * inv_denom;
      gc[1] = g.y * w * inv_denom;
      gc[2] = g.z * w * inv_denom;
      float G_zi = Gw * w / a.gamma;
      if (k == kstar && !z_clamped) G_zi += G_zmax;
      a.g_zbuf[i * K + k] = G_zi * (-q.m / (zf - zn));
      float others = 1.0f;  // prod_{j != k} (1 - p_j)
      for (int j = 0; j < K; ++j)
        if (j != k) others *= 1.0f - soft_k(a, i * K + j, zn, zf).p;
      const float G_p = Gw * e + g.w * others;
      a.g_dists[i * K + k] = G_p * (-(1.0f / a.sigma) * q.s * (1.0f - q.s) * q.m);
    }
  }
}

unsigned blend_grid(int64_t npix) {
  int64_t b = ceil_div(npix, kBlendBlock);
  if (b > 256 * 64) b = 256 * 64;
  return (unsigned)(b < 1 ? 1 : b);
}

// ---- hard_rgb_blend (blending.py:54-88): colour of the closest face, the background colour where there is none,
// alpha = 1 / 0.  A thread per pixel reads pix_to_face[..., 0] and the first of the K colour slots.
__global__ __launch_bounds__(kBlendBlock) void hard_blend_fwd_kernel(const float* __restrict__ colors,
                                                                      const int64_t* __restrict__ p2f, float bg0, float bg1,
                                                                      float bg2, int64_t npix, int K,
                                                                      float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kBlendBlock) {
    const bool hit = p2f[i * K] >= 0;
    const float* c = colors + i * K * 3;
    float4 o;
    o.x = hit ? c[0] : bg0;
    o.y = hit ? c[1] : bg1;
    o.z = hit ? c[2] : bg2;
    o.w = hit ? 1.0f : 0.0f;
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

// grad_colors (npix, K, 3) fully written: slot 0 of covered pixels takes grad_out[..., :3], everything else is zero
__global__ __launch_bounds__(kBlendBlock) void hard_blend_bwd_kernel(const float* __restrict__ grad_out,
                                                                      const int64_t* __restrict__ p2f, int64_t npix, int K,
                                                                      float* __restrict__ grad_colors) {
  const int64_t per_pix = (int64_t)K * 3, total = npix * per_pix;
  for (int64_t e = (int64_t)blockIdx.x * kBlendBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlendBlock) {
    const int64_t pix = e / per_pix;
    const int r = (int)(e - pix * per_pix);
    float g = 0.0f;
    if (r < 3 && p2f[pix * K] >= 0) g = grad_out[pix * 4 + r];
    grad_colors[e] = g;
  }
}

// smallest instantiated capacity >= K (0 when K is too large for the register kernels)
int blend_capacity(int K) { return K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : 0; }

#define P3D_BLEND_DISPATCH(KERNEL, GENERIC, K, ...)                                                     \
  switch (blend_capacity(K)) {                                                                           \
    case 1: KERNEL<1><<<grid, kBlendBlock, 0, s>>>(__VA_ARGS__); break;                                  \
    case 2: KERNEL<2><<<grid, kBlendBlock, 0, s>>>(__VA_ARGS__); break;                                  \
    case 4: KERNEL<4><<<grid, kBlendBlock, 0, s>>>(__VA_ARGS__); break;                                  \
    case 8: KERNEL<8><<<grid, kBlendBlock, 0, s>>>(__VA_ARGS__); break;                                  \
    case 16: KERNEL<16><<<grid, kBlendBlock, 0, s>>>(__VA_ARGS__); break;                                \
    case 32: KERNEL<32><<<grid, kBlendBlock, 0, s>>>(__VA_ARGS__); break;                                \
    default: GENERIC<<<grid, kBlendBlock, 0, s>>>(__VA_ARGS__); break;                                   \
  }

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_sigmoid_alpha_blend_forward(const float* dists, const int64_t* pix_to_face, float sigma, int64_t npix,
                                            int K, float* alphas, p3d_stream_t stream) {
  if (npix < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (npix == 0) return P3D_OK;
  if (!alphas || (K > 0 && (!dists || !pix_to_face))) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = blend_grid(npix);
  LaunchScope ls("sigmoid_alpha_fwd", s);
  P3D_BLEND_DISPATCH(sigmoid_alpha_fwd_kernel, sigmoid_alpha_fwd_generic, K, dists, pix_to_face, sigma, npix, K, alphas)
  return launch_status();
}

P3D_API int p3d_sigmoid_alpha_blend_backward(const float* grad_alphas, const float* alphas, const float* dists,
                                             const int64_t* pix_to_face, float sigma, int64_t npix, int K,
                                             float* grad_dists, p3d_stream_t stream) {
  if (npix < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (npix * K == 0) return P3D_OK;
  if (!grad_alphas || !alphas || !dists || !pix_to_face || !grad_dists) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = blend_grid(npix);
  LaunchScope ls("sigmoid_alpha_bwd", s);
  P3D_BLEND_DISPATCH(sigmoid_alpha_bwd_kernel, sigmoid_alpha_bwd_generic, K, grad_alphas, alphas, dists, pix_to_face, sigma, npix, K, grad_dists)
  return launch_status();
}

static int fill_blend_args(BlendArgs* a, const float* colors, const int64_t* p2f, const float* dists, const float* zbuf,
                           float sigma, float gamma, const float* background, float znear, float zfar,
                           const float* znear_n, const float* zfar_n, int64_t N, int64_t pix_per_image, int K) {
  if (N < 0 || pix_per_image < 0 || K < 0 || !background) return P3D_ERR_INVALID_ARG;
  a->colors = colors;
  a->p2f = p2f;
  a->dists = dists;
  a->zbuf = zbuf;
  a->sigma = sigma;
  a->gamma = gamma;
  a->bg0 = background[0];
  a->bg1 = background[1];
  a->bg2 = background[2];
  a->znear = znear;
  a->zfar = zfar;
  a->znear_n = znear_n;
  a->zfar_n = zfar_n;
  a->npix = N * pix_per_image;
  a->pix_per_image = pix_per_image > 0 ? pix_per_image : 1;
  a->K = K;
  return P3D_OK;
}

P3D_API int p3d_softmax_rgb_blend_forward(const float* colors, const int64_t* pix_to_face, const float* dists,
                                          const float* zbuf, float sigma, float gamma, const float background[3],
                                          float znear, float zfar, const float* znear_per_image,
                                          const float* zfar_per_image, int64_t N, int64_t pix_per_image, int K,
                                          float* out, p3d_stream_t stream) {
  BlendArgs a{};
  const int rc = fill_blend_args(&a, colors, pix_to_face, dists, zbuf, sigma, gamma, background, znear, zfar,
                                 znear_per_image, zfar_per_image, N, pix_per_image, K);
  if (rc != P3D_OK) return rc;
  if (a.npix == 0) return P3D_OK;
  if (!out || (K > 0 && (!colors || !pix_to_face || !dists || !zbuf))) return P3D_ERR_INVALID_ARG;
  a.out = out;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = blend_grid(a.npix);
  LaunchScope ls("softmax_blend_fwd", s);
  P3D_BLEND_DISPATCH(softmax_blend_fwd_kernel, softmax_blend_generic<false>, K, a)
  return launch_status();
}

P3D_API int p3d_softmax_rgb_blend_backward(const float* grad_out, const float* colors, const int64_t* pix_to_face,
                                           const float* dists, const float* zbuf, float sigma, float gamma,
                                           const float background[3], float znear, float zfar,
                                           const float* znear_per_image, const float* zfar_per_image, int64_t N,
                                           int64_t pix_per_image, int K, float* grad_colors, float* grad_dists,
                                           float* grad_zbuf, p3d_stream_t stream) {
  BlendArgs a{};
  const int rc = fill_blend_args(&a, colors, pix_to_face, dists, zbuf, sigma, gamma, background, znear, zfar,
                                 znear_per_image, zfar_per_image, N, pix_per_image, K);
  if (rc != P3D_OK) return rc;
  if (a.npix * K == 0) return P3D_OK;
  if (!grad_out || !colors || !pix_to_face || !dists || !zbuf || !grad_colors || !grad_dists || !grad_zbuf)
    return P3D_ERR_INVALID_ARG;
  a.grad_out = grad_out;
  a.g_colors = grad_colors;
  a.g_dists = grad_dists;
  a.g_zbuf = grad_zbuf;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = blend_grid(a.npix);
  LaunchScope ls("softmax_blend_bwd", s);
  P3D_BLEND_DISPATCH(softmax_blend_bwd_kernel, softmax_blend_generic<true>, K, a)
  return launch_status();
}

P3D_API int p3d_hard_rgb_blend_forward(const float* colors, const int64_t* pix_to_face, const float background[3],
                                       int64_t npix, int K, float* out, p3d_stream_t stream) {
  if (npix < 0 || K < 1 || !background) return P3D_ERR_INVALID_ARG;
  if (npix == 0) return P3D_OK;
  if (!colors || !pix_to_face || !out || ((uintptr_t)out & 15) != 0) return P3D_ERR_INVALID_ARG;  // float4 stores
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("hard_blend_fwd", s);
  hard_blend_fwd_kernel<<<blend_grid(npix), kBlendBlock, 0, s>>>(colors, pix_to_face, background[0], background[1],
                                                                 background[2], npix, K, out);
  return launch_status();
}

P3D_API int p3d_hard_rgb_blend_backward(const float* grad_out, const int64_t* pix_to_face, int64_t npix, int K,
                                        float* grad_colors, p3d_stream_t stream) {
  if (npix < 0 || K < 1) return P3D_ERR_INVALID_ARG;
  if (npix == 0) return P3D_OK;
  if (!grad_out || !pix_to_face || !grad_colors) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("hard_blend_bwd", s);
  hard_blend_bwd_kernel<<<blend_grid(npix * K * 3), kBlendBlock, 0, s>>>(grad_out, pix_to_face, npix, K,
                                                                               grad_colors);
  return launch_status();
}
