// shade.hip -- fused per-pixel Phong shading of rasterization fragments for gfx950.
//
// Replaces the torch-op chain of phong_shading (pytorch3d/renderer/mesh/shading.py:17-112):
//   pixel_coords  = interpolate_face_attributes(pix_to_face, bary, verts[faces])
//   pixel_normals = interpolate_face_attributes(pix_to_face, bary, verts_normals[faces])
//   ambient, diffuse, specular = _apply_lighting(...)       (lighting.py:17-159: F.normalize eps 1e-6, relu, pow)
//   colors = (ambient + diffuse) * texels + specular
// ~45 elementwise / reduction kernels over (N,H,W,K,3) tensors plus their autograd twins, each re-reading the
// fragments.  Here: one forward kernel (reads pix_to_face + bary [+ texels], gathers one 72/108-byte face record,
// writes colors) and one backward kernel that recomputes the forward per sample, applies the chain rule in
// registers and hands the per-face partials to the same wave-private LDS table as the mesh / interp backward
// (wave_table.h).  Face records are (F, 3, D): D = 6 [vertex xyz | vertex normal] with texels given per sample
// (the reference's signature), or D = 9 [.. | vertex colour] with the texture interpolation fused in as well.
//
// Per-image parameters (25 floats, `P3D_SHADE_PARAM_FLOATS`): light ambient / diffuse / specular colour,
// light location (point light) or direction (directional light), material ambient / diffuse / specular colour,
// shininess, camera centre.  Ambient-only lights are a directional light with zero diffuse and specular colour.
#include <stdlib.h>

#include "p3d_common.h"
#include "shade_sample.h"
#include "wave_table.h"

namespace p3d {
namespace {

struct ShadeArgs {
  const int64_t* p2f;
  const float* bary;
  const float* attrs;    // (F, 3, D)
  const float* texels;   // (N,H,W,K,3) when D == 6
  const float* params;   // (N, 25)
  const float* gcolors;  // (N,H,W,K,3)
  float* colors;         // (N,H,W,K,3)
  float* gbary;          // (N,H,W,K,3)
  float* gattrs;         // (F, 3, D)
  float* gtexels;        // (N,H,W,K,3) when D == 6
  float* gparams;        // (N, 25) or null: gradient of the per-image parameters (zeroed by the launcher, accumulated)
  int N, H, W, K, RY, RX, AW;
  int64_t HWK;
};

// ---- forward: one thread per sample, blockIdx.y = image ----------------------------------------------------------
template <int D, bool POINT>
__global__ __launch_bounds__(256) void phong_fwd_kernel(ShadeArgs a) {
  const int n = blockIdx.y;
  const ShadeConst c = load_params(a.params + (int64_t)n * P3D_SHADE_PARAM_FLOATS);
  float amb[3], kd[3], ks[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    amb[j] = c.ma[j] * c.la[j];  // shading.py:41
    kd[j] = c.md[j] * c.ld[j];
    ks[j] = c.ms[j] * c.ls[j];
  }
  const float pw0 = powf(0.0f, c.shin);
  const int64_t img = (int64_t)n * a.HWK;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.HWK; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = img + i;
    const int f = (int)a.p2f[p];
    float P[3] = {0.f, 0.f, 0.f}, Nn[3] = {0.f, 0.f, 0.f}, tex[3] = {0.f, 0.f, 0.f};
    if (f >= 0) {
      const float b0 = a.bary[p * 3], b1 = a.bary[p * 3 + 1], b2 = a.bary[p * 3 + 2];
      const float* r = a.attrs + (int64_t)f * 3 * D;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // interp_face_attrs.cu:39-41: (w0*a0 + w1*a1) + w2*a2
        P[j] = (b0 * r[j] + b1 * r[D + j]) + b2 * r[2 * D + j];
        Nn[j] = (b0 * r[3 + j] + b1 * r[D + 3 + j]) + b2 * r[2 * D + 3 + j];
        if (D == 9) tex[j] = (b0 * r[6 + j] + b1 * r[D + 6 + j]) + b2 * r[2 * D + 6 + j];
      }
    }
    if (D == 6) {
#pragma unroll
      for (int j = 0; j < 3; ++j) tex[j] = a.texels[p * 3 + j];
    }
    // Background samples interpolate to P = N = 0 in the reference and go through the same arithmetic; its outcome is
    // known: normalize(0) = 0, so the cosine is 0, there is no diffuse term and alpha = 0, pow(0, shininess) = pw0.
    float angle = 0.0f, pw = pw0;
    if (f >= 0) {
      const Lit s = light_sample<POINT>(c, P, Nn);
      angle = s.angle;
      pw = s.pw;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) a.colors[p * 3 + j] = (amb[j] + kd[j] * angle) * tex[j] + ks[j] * pw;  // shading.py:96
  }
}

// ---- backward: wave per area of 16 rows x AW pixels, lanes over consecutive (pixel, k) samples ------------------------
template <int D>
struct ShadeTable {
  static constexpr int NV = 3 * D;
  // LDS per workgroup = 4 waves x slots x (8 + 4 * stride) B and the kernel is latency-bound, so the tables are sized
  // for 5 / 4 workgroups per CU (D = 6: 32 KB, D = 9: 40 KB) rather than for the worst case of a step (64 consecutive
  // samples can name 64 faces): they run in SPILL mode (wave_table.h).  Measured on the config-3 fragments, D = 6:
  // 182 slots (2 WG/CU) 4.3 ms, 144 (3 WG/CU) 3.3 ms, 106 (4 WG/CU) 2.9 ms, 90 (5 WG/CU) 2.6 ms, 75 (6 WG/CU) 2.65 ms; D = 9: 110 slots (3 WG/CU) 3.6 ms, 83 (4 WG/CU) 3.2 ms.
  static constexpr int kSlots = D == 6 ? 88 : 80;  // multiples of 4: the keys are probed a bucket of four at a time
  using T = WaveTable<NV, kSlots, false, true, true>;
};

// PG: also accumulate the gradient of the per-image parameters (lights, materials, camera centre): 25 per-lane sums over
// the wave's samples, one butterfly reduction and 25 global atomics per wave at the end.
template <int D, bool POINT, bool PG>
__global__ __launch_bounds__(256) void phong_bwd_kernel(ShadeArgs a) {
#pragma clang fp contract(fast)  // gradient-only arithmetic (tolerance-gated): let mul+add fuse into FMA
  using Tab = typename ShadeTable<D>::T;
  constexpr int NV = 3 * D;
  __shared__ __align__(16) int s_table[4][Tab::kLdsInts];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // wave -> (image, strip of 16 rows, span of AW pixels); RY strips x RX spans per image
  const int64_t wid = (int64_t)blockIdx.x * 4 + w;
  const int64_t per_image = (int64_t)a.RY * a.RX;
  if (wid >= (int64_t)a.N * per_image) return;  // wave-uniform; no workgroup barrier in this kernel
  const int n = (int)(wid / per_image);
  const int t = (int)(wid - (int64_t)n * per_image);
  const int H = a.H, W = a.W, K = a.K;
  const int y0 = (t / a.RX) * 16, x0 = (t % a.RX) * a.AW;
  const int rows = min(16, H - y0), cols = min(a.AW, W - x0);
  const int run = cols * K;        // contiguous samples per row of the area
  const int total = rows * run;
  const float inv_run = 1.0f / (float)run;
  const ShadeConst c = load_params(a.params + (int64_t)n * P3D_SHADE_PARAM_FLOATS);
  float amb[3], kd[3], ks[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    amb[j] = c.ma[j] * c.la[j];
    kd[j] = c.md[j] * c.ld[j];
    ks[j] = c.ms[j] * c.ls[j];
  }
  Tab tab;
  tab.init(s_table[w], lane);
  float pg[PG ? P3D_SHADE_PARAM_FLOATS : 1];
#pragma unroll
  for (int j = 0; j < (PG ? P3D_SHADE_PARAM_FLOATS : 1); ++j) pg[j] = 0.0f;
  const float pw0 = c.shin == 0.0f ? 1.0f : 0.0f;  // pow(0, shininess): the specular term of samples without a highlight
  // Lanes take 64 consecutive samples of a row ((pixel, k) pairs in memory order): every load and store of the
  // per-sample arrays is then one contiguous 768-byte piece per wave instruction, for any K.  (A lane per pixel
  // stepping through k writes 12 bytes at a stride of 12*K: measured 2.9 ms of 5.5 on partial-line stores.)
  {
#pragma unroll 1
    for (int e0 = 0; e0 < total; e0 += 64) {
      const int e = e0 + lane;
      const bool ok = e < total;
      // exact for these small operands: (e + 0.5) / run is at least 0.5 / run away from an integer
      const int r = (int)(((float)e + 0.5f) * inv_run);
      const int64_t p = (((int64_t)n * H + y0 + r) * W + x0) * K + (e - r * run);
      int f = ok ? (int)a.p2f[p] : -1;
      float g[NV];
      float gb[3] = {0.f, 0.f, 0.f};
      float go[3] = {0.f, 0.f, 0.f};
      if (ok && (D == 6 || PG || f >= 0)) {  // with vertex colours a background sample's colour gradient goes nowhere
#pragma unroll
        for (int j = 0; j < 3; ++j) go[j] = a.gcolors[p * 3 + j];
      }
      if (D == 6 && ok && f < 0) {
        // background sample with caller-supplied texels: colour = ambient * texel (+ a constant)
#pragma unroll
        for (int j = 0; j < 3; ++j) a.gtexels[p * 3 + j] = amb[j] * go[j];
      }
      if (PG && ok && f < 0) {
        // colour_j = ma_j * la_j * tex_j + ms_j * ls_j * pow(0, shininess)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float tg = D == 6 ? a.texels[p * 3 + j] * go[j] : 0.0f;
          pg[j] += c.ma[j] * tg;
          pg[12 + j] += c.la[j] * tg;
          pg[6 + j] += c.ms[j] * pw0 * go[j];
          pg[18 + j] += c.ls[j] * pw0 * go[j];
        }
      }
      if (f >= 0) {
        const float b[3] = {a.bary[p * 3], a.bary[p * 3 + 1], a.bary[p * 3 + 2]};
        float tex_in[3] = {0.f, 0.f, 0.f};
        if (D == 6) {
#pragma unroll
          for (int j = 0; j < 3; ++j) tex_in[j] = a.texels[p * 3 + j];
        }
        float dtex[3];
        shade_sample_bwd<D, POINT, PG>(c, amb, kd, ks, pw0, a.attrs + (int64_t)f * NV, b, tex_in, go, g, gb, dtex, pg);
        if (D == 6) {
#pragma unroll
          for (int j = 0; j < 3; ++j) a.gtexels[p * 3 + j] = dtex[j];
        }
      }
      if (ok) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a.gbary[p * 3 + j] = gb[j];
      }
      if (__ballot(f >= 0) == 0) continue;  // wave-uniform
      tab.add(a.gattrs, lane, f, g);
    }
  }
  if (tab.used > 0) tab.flush(a.gattrs, lane);
  if constexpr (PG) {
#pragma unroll
    for (int j = 0; j < P3D_SHADE_PARAM_FLOATS; ++j) {
      float v = pg[j];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) unsafeAtomicAdd(a.gparams + (int64_t)n * P3D_SHADE_PARAM_FLOATS + j, v);
    }
  }
}

int check_shape(int N, int H, int W, int K, int64_t F, int D, int light_kind) {
  if (N < 0 || H < 0 || W < 0 || K < 0 || F < 0) return P3D_ERR_INVALID_ARG;
  if (D != 6 && D != 9) return P3D_ERR_INVALID_ARG;
  if (light_kind != P3D_LIGHT_DIRECTIONAL && light_kind != P3D_LIGHT_POINT) return P3D_ERR_INVALID_ARG;
  return P3D_OK;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_phong_shade_forward(const int64_t* pix_to_face, const float* bary, const float* face_attrs, int D,
                                    const float* texels, const float* params, int light_kind, int N, int H, int W, int K,
                                    int64_t F, float* colors, p3d_stream_t stream) {
  const int rc = check_shape(N, H, W, K, F, D, light_kind);
  if (rc != P3D_OK) return rc;
  const int64_t HWK = (int64_t)H * W * K;
  if ((int64_t)N * HWK == 0) return P3D_OK;
  if (!pix_to_face || !bary || !params || !colors || (F > 0 && !face_attrs) || (D == 6 && !texels)) return P3D_ERR_INVALID_ARG;
  if (N > 65535) return P3D_ERR_INVALID_ARG;
  ShadeArgs a{};
  a.p2f = pix_to_face;
  a.bary = bary;
  a.attrs = face_attrs;
  a.texels = texels;
  a.params = params;
  a.colors = colors;
  a.N = N;
  a.H = H;
  a.W = W;
  a.K = K;
  a.HWK = HWK;
  hipStream_t s = (hipStream_t)stream;
  int64_t bx = ceil_div(HWK, 256 * 4);  // four samples per thread
  if (bx > 65535) bx = 65535;
  const dim3 grid((unsigned)bx, (unsigned)N);
  LaunchScope ls("phong_fwd", s);
  const bool point = light_kind == P3D_LIGHT_POINT;
  if (D == 6) {
    if (point)
      phong_fwd_kernel<6, true><<<grid, 256, 0, s>>>(a);
    else
      phong_fwd_kernel<6, false><<<grid, 256, 0, s>>>(a);
  } else {
    if (point)
      phong_fwd_kernel<9, true><<<grid, 256, 0, s>>>(a);
    else
      phong_fwd_kernel<9, false><<<grid, 256, 0, s>>>(a);
  }
  return launch_status();
}

P3D_API int p3d_phong_shade_backward(const float* grad_colors, const int64_t* pix_to_face, const float* bary,
                                     const float* face_attrs, int D, const float* texels, const float* params,
                                     int light_kind, int N, int H, int W, int K, int64_t F, float* grad_bary,
                                     float* grad_face_attrs, float* grad_texels, float* grad_params,
                                     p3d_stream_t stream) {
  const int rc = check_shape(N, H, W, K, F, D, light_kind);
  if (rc != P3D_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (F > 0) {
    if (!grad_face_attrs) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_face_attrs, 0, (size_t)F * 3 * D * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  if (grad_params && N > 0 &&
      hipMemsetAsync(grad_params, 0, (size_t)N * P3D_SHADE_PARAM_FLOATS * sizeof(float), s) != hipSuccess)
    return P3D_ERR_LAUNCH;
  const int64_t HWK = (int64_t)H * W * K;
  if ((int64_t)N * HWK == 0) return P3D_OK;
  if (!grad_colors || !pix_to_face || !bary || !params || !grad_bary || (F > 0 && !face_attrs)) return P3D_ERR_INVALID_ARG;
  if (D == 6 && (!texels || !grad_texels)) return P3D_ERR_INVALID_ARG;
  ShadeArgs a{};
  a.gcolors = grad_colors;
  a.p2f = pix_to_face;
  a.bary = bary;
  a.attrs = face_attrs;
  a.texels = texels;
  a.params = params;
  a.gbary = grad_bary;
  a.gattrs = grad_face_attrs;
  a.gtexels = grad_texels;
  a.gparams = grad_params;
  a.N = N;
  a.H = H;
  a.W = W;
  a.K = K;
  a.HWK = HWK;
  a.AW = K >= 4 ? 16 : (K == 3 ? 24 : (K == 2 ? 32 : 64));  // >= 64 samples per row of the area
  a.RY = (int)ceil_div(H, 16);
  a.RX = (int)ceil_div(W, a.AW);
  const int64_t blocks = ceil_div((int64_t)N * a.RY * a.RX, 4);
  if (blocks > 0x7fffffff) return P3D_ERR_INVALID_ARG;
  LaunchScope ls("phong_bwd", s);
  const bool point = light_kind == P3D_LIGHT_POINT;
  const unsigned grid = (unsigned)blocks;
#define P3D_LAUNCH_PHONG_BWD(DD)                                                 \
  do {                                                                           \
    if (grad_params) {                                                           \
      if (point)                                                                 \
        phong_bwd_kernel<DD, true, true><<<grid, 256, 0, s>>>(a);                \
      else                                                                       \
        phong_bwd_kernel<DD, false, true><<<grid, 256, 0, s>>>(a);               \
    } else {                                                                     \
      if (point)                                                                 \
        phong_bwd_kernel<DD, true, false><<<grid, 256, 0, s>>>(a);               \
      else                                                                       \
        phong_bwd_kernel<DD, false, false><<<grid, 256, 0, s>>>(a);              \
    }                                                                            \
  } while (0)
  if (D == 6)
    P3D_LAUNCH_PHONG_BWD(6);
  else
    P3D_LAUNCH_PHONG_BWD(9);
#undef P3D_LAUNCH_PHONG_BWD
  return launch_status();
}
