// soft_phong.hip -- SoftPhongShader in one kernel each way for gfx950: Phong shading fused with softmax_rgb_blend
// ("fused interp + light + blend kernel").
//
// Replaces SoftPhongShader.forward (pytorch3d/renderer/mesh/shader.py:113-147):
//     colors = phong_shading(meshes, fragments, texels, lights, cameras, materials)     (shading.py:100-125)
//     images = softmax_rgb_blend(colors, fragments, blend_params, znear, zfar)         (blending.py:147-244)
// With shade.hip + blend.hip the per-sample colours (N,H,W,K,3) make a round trip through HBM each way: 1.6 GB written
// and read in the forward, and again as their gradient in the backward (config-3 fragments), between kernels that are
// both HBM streams.  Here the colours never leave registers.
//
// Layout: a lane owns one SAMPLE, lanes of a wave are 64 consecutive (pixel, k) samples in memory order -- for K in
// {1, 2, 4, 8, 16} a pixel's K slots are K adjacent lanes of one DPP row.  The per-pixel reductions of the blend (max of
// the inverse depths and its first arg max, sum of the weights, weighted colour sums, product / prefix / suffix products
// of 1 - p) are DPP butterflies inside those K lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror; 3 VALU each),
// every load and store of the per-sample arrays is one contiguous piece per wave instruction, and the per-face partials
// of the backward go through the same wave-private LDS table as shade.hip (wave_table.h).
// Arithmetic: the per-sample shading is shade_sample.h's, the blend follows blend.hip (i.e. blending.py's float chain)
// step by step; the sums over k are butterfly sums instead of serial ones (a rounding-level difference).
#include <stdlib.h>

#include "p3d_common.h"
#include "shade_sample.h"
#include "wave_table.h"

namespace p3d {
namespace {

struct SoftPhongArgs {
  const int64_t* p2f;
  const float* bary;
  const float* dists;
  const float* zbuf;
  const float* attrs;    // (F, 3, D)
  const float* texels;   // (N,H,W,K,3) when D == 6
  const float* params;   // (N, 25)
  const float* grad_out; // (N,H,W,4)
  float sigma, gamma, bg[3];
  float znear, zfar;
  const float* znear_n;
  const float* zfar_n;
  float* out;            // (N,H,W,4)
  float* gbary;          // (N,H,W,K,3)
  float* gdists;         // (N,H,W,K)
  float* gzbuf;          // (N,H,W,K)
  float* gattrs;         // (F, 3, D)
  float* gtexels;        // (N,H,W,K,3) when D == 6
  float* gparams;        // (N, 25) or null
  int N, H, W, RY, RX, AW;
  int64_t HWK;
};

// ---- reductions over the KT lanes of a pixel (KT | 16, groups aligned inside a 16-lane DPP row) -------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int x) {
  return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false);
}
constexpr int kXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kHalfMirror = 0x141; // lane i <-> 7 - i inside 8 lanes
constexpr int kRowMirror = 0x140;  // lane i <-> 15 - i inside 16 lanes

#define P3D_GRP_REDUCE(NAME, TYPE, DPPF, OP)                 \
  template <int KT>                                          \
  __device__ __forceinline__ TYPE NAME(TYPE x) {             \
    if constexpr (KT >= 2) x = OP(x, DPPF<kXor1>(x));        \
    if constexpr (KT >= 4) x = OP(x, DPPF<kXor2>(x));        \
    if constexpr (KT >= 8) x = OP(x, DPPF<kHalfMirror>(x));  \
    if constexpr (KT >= 16) x = OP(x, DPPF<kRowMirror>(x));  \
    return x;                                                \
  }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
__device__ __forceinline__ float op_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float op_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ int op_imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int op_ior(int a, int b) { return a | b; }
P3D_GRP_REDUCE(grp_sum, float, dpp, op_add)
P3D_GRP_REDUCE(grp_prod, float, dpp, op_mul)
P3D_GRP_REDUCE(grp_max, float, dpp, op_max)
P3D_GRP_REDUCE(grp_imin, int, dppi, op_imin)
P3D_GRP_REDUCE(grp_ior, int, dppi, op_ior)
#undef P3D_GRP_REDUCE

// product of x over the OTHER lanes of the group: exclusive prefix x exclusive suffix (Hillis-Steele over row_shr / row_shl;
// k = the lane's slot).  Exact also when some x is 0 (a division by the lane's own factor would not be).
template <int KT>
__device__ __forceinline__ float grp_prod_others(float x, int k) {
  float pre = x, suf = x;  // inclusive scans
  if constexpr (KT >= 2) {
    const float a = dpp<0x111>(pre), b = dpp<0x101>(suf);  // row_shr:1, row_shl:1
    pre = k >= 1 ? pre * a : pre;
    suf = k + 1 < KT ? suf * b : suf;
  }
  if constexpr (KT >= 4) {
    const float a = dpp<0x112>(pre), b = dpp<0x102>(suf);
    pre = k >= 2 ? pre * a : pre;
    suf = k + 2 < KT ? suf * b : suf;
  }
  if constexpr (KT >= 8) {
    const float a = dpp<0x114>(pre), b = dpp<0x104>(suf);
    pre = k >= 4 ? pre * a : pre;
    suf = k + 4 < KT ? suf * b : suf;
  }
  if constexpr (KT >= 16) {
    const float a = dpp<0x118>(pre), b = dpp<0x108>(suf);
    pre = k >= 8 ? pre * a : pre;
    suf = k + 8 < KT ? suf * b : suf;
  }
  // exclusive = the neighbour's inclusive value
  float ep = 1.0f, es = 1.0f;
  if constexpr (KT >= 2) {
    const float a = dpp<0x111>(pre), b = dpp<0x101>(suf);
    ep = k >= 1 ? a : 1.0f;
    es = k + 1 < KT ? b : 1.0f;
  }
  return ep * es;
}

// per-pixel state of softmax_rgb_blend as one of the pixel's KT lanes sees it (blending.py:195-232, blend.hip SoftmaxPixel)
struct SoftLane {
  float m, s, p, zi, e, w;        // this sample
  float zmax, delta, denom;        // the pixel
  bool z_clamped, d_clamped, is_kstar;
};

template <int KT>
__device__ __forceinline__ SoftLane soft_lane(bool valid, float d, float z, int k, float sigma, float gamma, float zn, float zf) {
  const float eps = 1e-10f;
  SoftLane q;
  q.m = valid ? 1.0f : 0.0f;
  q.s = 1.0f / (1.0f + expf(d / sigma));  // torch.sigmoid(-dists / sigma)
  q.p = q.s * q.m;
  q.zi = (zf - z) / (zf - zn) * q.m;
  const float zraw = grp_max<KT>(q.zi);
  q.is_kstar = grp_imin<KT>(q.zi == zraw ? k : KT) == k;  // torch.max: the first maximal slot takes the gradient
  q.z_clamped = zraw < eps;
  q.zmax = q.z_clamped ? eps : zraw;
  q.delta = expf((eps - q.zmax) / gamma);
  q.d_clamped = q.delta < eps;
  if (q.d_clamped) q.delta = eps;
  q.e = expf((q.zi - q.zmax) / gamma);
  q.w = q.p * q.e;
  q.denom = grp_sum<KT>(q.w) + q.delta;
  return q;
}

// ---- forward: grid-stride over the samples of image blockIdx.y, a lane per sample ------------------------------------
template <int D, bool POINT, int KT>
__global__ __launch_bounds__(256) void soft_phong_fwd_kernel(SoftPhongArgs a) {
  const int n = blockIdx.y;
  const ShadeConst c = load_params(a.params + (int64_t)n * P3D_SHADE_PARAM_FLOATS);
  float amb[3], kd[3], ks[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    amb[j] = c.ma[j] * c.la[j];  // shading.py:41
    kd[j] = c.md[j] * c.ld[j];
    ks[j] = c.ms[j] * c.ls[j];
  }
  const float zn = a.znear_n ? a.znear_n[n] : a.znear;
  const float zf = a.zfar_n ? a.zfar_n[n] : a.zfar;
  const int64_t img = (int64_t)n * a.HWK;
  const int k = (int)(threadIdx.x & (KT - 1));
  // HWK is a multiple of KT and the stride a multiple of 64: a pixel's lanes always run together
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.HWK; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = img + i;
    const int f = (int)a.p2f[p];
    const bool valid = f >= 0;
    const bool any = grp_ior<KT>(valid ? 1 : 0) != 0;
    float4 o = make_float4(a.bg[0], a.bg[1], a.bg[2], 0.0f);
    if (any || zf == zn) {  // (a pixel without a face is the background with alpha 0: blend.hip, same shortcut)
      float col[3] = {0.f, 0.f, 0.f};
      float d = 0.0f, z = 0.0f;
      if (valid) {
        d = a.dists[p];
        z = a.zbuf[p];
        const float b[3] = {a.bary[p * 3], a.bary[p * 3 + 1], a.bary[p * 3 + 2]};
        float tex_in[3] = {0.f, 0.f, 0.f};
        if (D == 6) {
#pragma unroll
          for (int j = 0; j < 3; ++j) tex_in[j] = a.texels[p * 3 + j];
        }
        shade_sample_fwd<D, POINT>(c, amb, kd, ks, a.attrs + (int64_t)f * 3 * D, b, tex_in, col);
      } else if (any) {
        // an empty slot of a covered pixel: its colour has weight 0, but dists / zbuf take part as the reference's masked
        // values do (zi = 0 enters the max): read what the unfused kernels read
        d = a.dists[p];
        z = a.zbuf[p];
      }
      const SoftLane q = soft_lane<KT>(valid, d, z, k, a.sigma, a.gamma, zn, zf);
      const float r0 = grp_sum<KT>(q.w * col[0]), r1 = grp_sum<KT>(q.w * col[1]), r2 = grp_sum<KT>(q.w * col[2]);
      const float alpha = grp_prod<KT>(1.0f - q.p);
      o.x = (r0 + q.delta * a.bg[0]) / q.denom;
      o.y = (r1 + q.delta * a.bg[1]) / q.denom;
      o.z = (r2 + q.delta * a.bg[2]) / q.denom;
      o.w = 1.0f - alpha;
    }
    if (k == 0) *reinterpret_cast<float4*>(a.out + (p / KT) * 4) = o;
  }
}

// ---- backward: wave per area of 16 rows x AW pixels, lanes over consecutive (pixel, k) samples (shade.hip's layout) -----
template <int D>
struct SoftTable {
  static constexpr int NV = 3 * D;
  static constexpr int kSlots = D == 6 ? 88 : 80;  // shade.hip: ShadeTable (5 / 4 workgroups per CU, spill mode, bucket probing)
  using T = WaveTable<NV, kSlots, false, true, true>;
};

constexpr int kSoftPhongBwdWaves = 3;  // measured (D = 9, K = 8): 146 VGPRs at 3 waves 3.84 ms; capped at 128 (72 B of scratch) 4.80 ms
template <int D, bool POINT, bool PG, int KT>
// PG (gradients to lights / materials / camera: 25 more per-lane sums) needs more than the 168 registers of three waves per
// SIMD: it spilled 2..19 VGPRs there, and kernels with VGPR spills are refused by the build since round 4 (build.py).
__global__ __launch_bounds__(256, PG ? 2 : kSoftPhongBwdWaves) void soft_phong_bwd_kernel(SoftPhongArgs a) {
#pragma clang fp contract(fast)  // gradient-only arithmetic (tolerance-gated): let mul+add fuse into FMA
  using Tab = typename SoftTable<D>::T;
  constexpr int NV = 3 * D;
  __shared__ __align__(16) int s_table[4][Tab::kLdsInts];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 4 + w;
  const int64_t per_image = (int64_t)a.RY * a.RX;
  if (wid >= (int64_t)a.N * per_image) return;  // wave-uniform; no workgroup barrier in this kernel
  const int n = (int)(wid / per_image);
  const int t = (int)(wid - (int64_t)n * per_image);
  const int H = a.H, W = a.W;
  const int y0 = (t / a.RX) * 16, x0 = (t % a.RX) * a.AW;
  const int rows = min(16, H - y0), cols = min(a.AW, W - x0);
  const int run = cols * KT;  // contiguous samples per row of the area: a multiple of KT, pixels never straddle steps
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
  const float zn = a.znear_n ? a.znear_n[n] : a.znear;
  const float zf = a.zfar_n ? a.zfar_n[n] : a.zfar;
  const float inv_range = 1.0f / (zf - zn);
  Tab tab;
  tab.init(s_table[w], lane);
  float pg[PG ? P3D_SHADE_PARAM_FLOATS : 1];
#pragma unroll
  for (int j = 0; j < (PG ? P3D_SHADE_PARAM_FLOATS : 1); ++j) pg[j] = 0.0f;
  const float pw0 = c.shin == 0.0f ? 1.0f : 0.0f;  // pow(0, shininess)
  const int k = lane & (KT - 1);
  // Streams two steps deep (round 4).  The kernel is bound by latency, not by issue: 11,400 VALU instructions per wave in
  // ~200 us, 52 % of the wave cycles waiting with three waves per SIMD (profiles/r04/soft_phong_pmc.md) -- every step was the
  // chain pix_to_face -> (face record gather, per-sample operands) -> arithmetic -> table, each link a memory round trip.
  // Now pix_to_face of step t + 2 and the per-sample operands of step t + 1 (distance, depth, barycentrics, the pixel's
  // upstream gradient -- still only for samples / pixels that hold a face) are requested before step t is computed: 12
  // registers, and the only load a step still waits for is its face-record gather.
  auto sample_pos = [&](int e0, bool* ok) -> int64_t {
    const int e = e0 + lane;
    *ok = e < total;  // (run is a multiple of KT and 64 is too: a pixel's lanes are ok together)
    const int r = (int)(((float)e + 0.5f) * inv_run);  // exact for these small operands (shade.hip)
    return (((int64_t)n * H + y0 + r) * W + x0) * KT + (e - r * run);
  };
  struct Ops {
    float d, z, b[3];
    float4 go;
  };
  auto load_ops = [&](bool ok, int64_t p, int f, Ops* o) {
    const bool valid = f >= 0;
    const bool any = grp_ior<KT>(valid ? 1 : 0) != 0;
    o->d = o->z = 0.0f;
    o->b[0] = o->b[1] = o->b[2] = 0.0f;
    o->go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && (any || zf == zn)) {
      o->d = a.dists[p];
      o->z = a.zbuf[p];
      o->go = *reinterpret_cast<const float4*>(a.grad_out + (p / KT) * 4);
    }
    if (valid) {
#pragma unroll
      for (int j = 0; j < 3; ++j) o->b[j] = a.bary[p * 3 + j];
    }
  };
  bool ok_c, ok_n = false, ok_nn = false;
  int64_t p_c = sample_pos(0, &ok_c), p_n = 0, p_nn = 0;
  int f_c = ok_c ? (int)a.p2f[p_c] : -1, f_n = -1, f_nn = -1;
  if (64 < total) {
    p_n = sample_pos(64, &ok_n);
    f_n = ok_n ? (int)a.p2f[p_n] : -1;
  }
  Ops cur, nxt;
  load_ops(ok_c, p_c, f_c, &cur);
#pragma unroll 1
  for (int e0 = 0; e0 < total; e0 += 64) {
    // requests for the steps ahead
    f_nn = -1;
    ok_nn = false;
    if (e0 + 128 < total) {
      p_nn = sample_pos(e0 + 128, &ok_nn);
      f_nn = ok_nn ? (int)a.p2f[p_nn] : -1;
    }
    if (e0 + 64 < total) load_ops(ok_n, p_n, f_n, &nxt);
    const bool ok = ok_c;
    const int64_t p = p_c;
    const int f = f_c;
    const bool valid = f >= 0;
    const bool any = grp_ior<KT>(valid ? 1 : 0) != 0;
    float gb[3] = {0.f, 0.f, 0.f};
    float gdist = 0.0f, gz = 0.0f;
    float g[NV];
    if (any || (ok && zf == zn)) {
      const float d = cur.d, z = cur.z;
      float b[3] = {cur.b[0], cur.b[1], cur.b[2]}, col[3] = {0.f, 0.f, 0.f}, tex_in[3] = {0.f, 0.f, 0.f};
      ShadeState<D> st;
      if (valid) {
        if (D == 6) {
#pragma unroll
          for (int j = 0; j < 3; ++j) tex_in[j] = a.texels[p * 3 + j];
        }
        shade_sample_prepare<D, POINT>(c, a.attrs + (int64_t)f * NV, b, tex_in, st);
        shade_sample_color<D>(amb, kd, ks, pw0, st, col);
      }
      // ---- softmax blend backward (blend.hip: softmax_blend_bwd_kernel), the pixel's K slots side by side in the wave
      const SoftLane q = soft_lane<KT>(valid, d, z, k, a.sigma, a.gamma, zn, zf);
      const float4 go4 = cur.go;
      const float inv_denom = 1.0f / q.denom;
      const float r0 = (grp_sum<KT>(q.w * col[0]) + q.delta * a.bg[0]) * inv_denom;
      const float r1 = (grp_sum<KT>(q.w * col[1]) + q.delta * a.bg[1]) * inv_denom;
      const float r2 = (grp_sum<KT>(q.w * col[2]) + q.delta * a.bg[2]) * inv_denom;
      const float G_delta = (go4.x * (a.bg[0] - r0) + go4.y * (a.bg[1] - r1) + go4.z * (a.bg[2] - r2)) * inv_denom;
      const float Gw = (go4.x * (col[0] - r0) + go4.y * (col[1] - r1) + go4.z * (col[2] - r2)) * inv_denom;
      const float G_zmax = (q.d_clamped ? 0.0f : G_delta * (-q.delta / a.gamma)) + grp_sum<KT>(Gw * (-q.w / a.gamma));
      float G_zi = Gw * q.w / a.gamma;
      if (q.is_kstar && !q.z_clamped) G_zi += G_zmax;
      gz = G_zi * (-q.m * inv_range);
      const float others = grp_prod_others<KT>(1.0f - q.p, k);
      const float G_p = Gw * q.e + go4.w * others;
      gdist = G_p * (-(1.0f / a.sigma) * q.s * (1.0f - q.s) * q.m);
      // gradient of this sample's colour, handed straight to the shading backward
      const float wd = q.w * inv_denom;
      const float go[3] = {go4.x * wd, go4.y * wd, go4.z * wd};
      if (valid) {
        float dtex[3];
        shade_sample_finish<D, POINT, PG>(c, amb, kd, ks, pw0, st, a.attrs + (int64_t)f * NV, b, go, g, gb, dtex, pg);
        if (D == 6) {
#pragma unroll
          for (int j = 0; j < 3; ++j) a.gtexels[p * 3 + j] = dtex[j];
        }
      } else if (D == 6 && ok) {
        // an empty slot: colour = ambient * texel + constant with weight w = 0, so its texel gets no gradient
#pragma unroll
        for (int j = 0; j < 3; ++j) a.gtexels[p * 3 + j] = 0.0f;
      }
    } else if (D == 6 && ok) {
#pragma unroll
      for (int j = 0; j < 3; ++j) a.gtexels[p * 3 + j] = 0.0f;
    }
    if (ok) {
#pragma unroll
      for (int j = 0; j < 3; ++j) a.gbary[p * 3 + j] = gb[j];
      a.gdists[p] = gdist;
      a.gzbuf[p] = gz;
    }
    // rotate the pipeline (every path below this point continues with the next step)
    ok_c = ok_n;
    p_c = p_n;
    f_c = f_n;
    cur = nxt;
    ok_n = ok_nn;
    p_n = p_nn;
    f_n = f_nn;
    if (__ballot(valid) == 0) continue;  // wave-uniform
    tab.add(a.gattrs, lane, f, g);
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

bool k_supported(int K) { return K == 1 || K == 2 || K == 4 || K == 8 || K == 16; }

int fill(SoftPhongArgs* a, const int64_t* p2f, const float* bary, const float* dists, const float* zbuf, const float* attrs, int D,
         const float* texels, const float* params, int light_kind, float sigma, float gamma, const float background[3], float znear,
         float zfar, const float* znear_n, const float* zfar_n, int N, int H, int W, int K, int64_t F) {
  if (N < 0 || H < 0 || W < 0 || K < 0 || F < 0) return P3D_ERR_INVALID_ARG;
  if (D != 6 && D != 9) return P3D_ERR_INVALID_ARG;
  if (light_kind != P3D_LIGHT_DIRECTIONAL && light_kind != P3D_LIGHT_POINT) return P3D_ERR_INVALID_ARG;
  if (!k_supported(K) && (int64_t)N * H * W * K != 0) return P3D_ERR_INVALID_ARG;
  if (!(sigma > 0.0f) || !(gamma > 0.0f) || !background) return P3D_ERR_INVALID_ARG;
  a->p2f = p2f;
  a->bary = bary;
  a->dists = dists;
  a->zbuf = zbuf;
  a->attrs = attrs;
  a->texels = texels;
  a->params = params;
  a->sigma = sigma;
  a->gamma = gamma;
  a->bg[0] = background[0];
  a->bg[1] = background[1];
  a->bg[2] = background[2];
  a->znear = znear;
  a->zfar = zfar;
  a->znear_n = znear_n;
  a->zfar_n = zfar_n;
  a->N = N;
  a->H = H;
  a->W = W;
  a->HWK = (int64_t)H * W * K;
  return P3D_OK;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_soft_phong_supported_k(int K) { return k_supported(K) ? 1 : 0; }

#define P3D_SOFT_K(KERNEL_CALL) \
  switch (K) {                  \
    case 1: KERNEL_CALL(1); break;   \
    case 2: KERNEL_CALL(2); break;   \
    case 4: KERNEL_CALL(4); break;   \
    case 8: KERNEL_CALL(8); break;   \
    default: KERNEL_CALL(16); break; \
  }

P3D_API int p3d_soft_phong_forward(const int64_t* pix_to_face, const float* bary, const float* dists, const float* zbuf,
                                   const float* face_attrs, int D, const float* texels, const float* params, int light_kind,
                                   float sigma, float gamma, const float background[3], float znear, float zfar,
                                   const float* znear_per_image, const float* zfar_per_image, int N, int H, int W, int K,
                                   int64_t F, float* out, p3d_stream_t stream) {
  SoftPhongArgs a{};
  const int rc = fill(&a, pix_to_face, bary, dists, zbuf, face_attrs, D, texels, params, light_kind, sigma, gamma, background, znear,
                      zfar, znear_per_image, zfar_per_image, N, H, W, K, F);
  if (rc != P3D_OK) return rc;
  if ((int64_t)N * a.HWK == 0) return P3D_OK;
  if (!pix_to_face || !bary || !dists || !zbuf || !params || !out || (F > 0 && !face_attrs) || (D == 6 && !texels)) return P3D_ERR_INVALID_ARG;
  if (N > 65535) return P3D_ERR_INVALID_ARG;
  a.out = out;
  hipStream_t s = (hipStream_t)stream;
  int64_t bx = ceil_div(a.HWK, 256 * 4);
  if (bx > 65535) bx = 65535;
  const dim3 grid((unsigned)bx, (unsigned)N);
  LaunchScope ls("soft_phong_fwd", s);
  const bool point = light_kind == P3D_LIGHT_POINT;
#define P3D_CALL(KT_)                                                                   \
  do {                                                                                  \
    if (D == 6) {                                                                       \
      if (point) soft_phong_fwd_kernel<6, true, KT_><<<grid, 256, 0, s>>>(a);           \
      else soft_phong_fwd_kernel<6, false, KT_><<<grid, 256, 0, s>>>(a);                \
    } else {                                                                            \
      if (point) soft_phong_fwd_kernel<9, true, KT_><<<grid, 256, 0, s>>>(a);           \
      else soft_phong_fwd_kernel<9, false, KT_><<<grid, 256, 0, s>>>(a);                \
    }                                                                                   \
  } while (0)
  P3D_SOFT_K(P3D_CALL)
#undef P3D_CALL
  return launch_status();
}

P3D_API int p3d_soft_phong_backward(const float* grad_out, const int64_t* pix_to_face, const float* bary, const float* dists,
                                    const float* zbuf, const float* face_attrs, int D, const float* texels, const float* params,
                                    int light_kind, float sigma, float gamma, const float background[3], float znear, float zfar,
                                    const float* znear_per_image, const float* zfar_per_image, int N, int H, int W, int K,
                                    int64_t F, float* grad_bary, float* grad_dists, float* grad_zbuf, float* grad_face_attrs,
                                    float* grad_texels, float* grad_params, p3d_stream_t stream) {
  SoftPhongArgs a{};
  const int rc = fill(&a, pix_to_face, bary, dists, zbuf, face_attrs, D, texels, params, light_kind, sigma, gamma, background, znear,
                      zfar, znear_per_image, zfar_per_image, N, H, W, K, F);
  if (rc != P3D_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (F > 0) {
    if (!grad_face_attrs) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_face_attrs, 0, (size_t)F * 3 * D * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  if (grad_params && N > 0 && hipMemsetAsync(grad_params, 0, (size_t)N * P3D_SHADE_PARAM_FLOATS * sizeof(float), s) != hipSuccess)
    return P3D_ERR_LAUNCH;
  if ((int64_t)N * a.HWK == 0) return P3D_OK;
  if (!grad_out || !pix_to_face || !bary || !dists || !zbuf || !params || !grad_bary || !grad_dists || !grad_zbuf || (F > 0 && !face_attrs))
    return P3D_ERR_INVALID_ARG;
  if (D == 6 && (!texels || !grad_texels)) return P3D_ERR_INVALID_ARG;
  a.grad_out = grad_out;
  a.gbary = grad_bary;
  a.gdists = grad_dists;
  a.gzbuf = grad_zbuf;
  a.gattrs = grad_face_attrs;
  a.gtexels = grad_texels;
  a.gparams = grad_params;
  a.AW = K >= 4 ? 16 : (K == 2 ? 32 : 64);  // >= 64 samples per row of the area
  a.RY = (int)ceil_div(H, 16);
  a.RX = (int)ceil_div(W, a.AW);
  const int64_t blocks = ceil_div((int64_t)N * a.RY * a.RX, 4);
  if (blocks > 0x7fffffff) return P3D_ERR_INVALID_ARG;
  LaunchScope ls("soft_phong_bwd", s);
  const bool point = light_kind == P3D_LIGHT_POINT;
  const unsigned grid = (unsigned)blocks;
#define P3D_CALL2(DD, KT_)                                                                   \
  do {                                                                                       \
    if (grad_params) {                                                                       \
      if (point) soft_phong_bwd_kernel<DD, true, true, KT_><<<grid, 256, 0, s>>>(a);         \
      else soft_phong_bwd_kernel<DD, false, true, KT_><<<grid, 256, 0, s>>>(a);              \
    } else {                                                                                 \
      if (point) soft_phong_bwd_kernel<DD, true, false, KT_><<<grid, 256, 0, s>>>(a);        \
      else soft_phong_bwd_kernel<DD, false, false, KT_><<<grid, 256, 0, s>>>(a);             \
    }                                                                                        \
  } while (0)
#define P3D_CALL(KT_)          \
  do {                         \
    if (D == 6) P3D_CALL2(6, KT_); \
    else P3D_CALL2(9, KT_);    \
  } while (0)
  P3D_SOFT_K(P3D_CALL)
#undef P3D_CALL
#undef P3D_CALL2
  return launch_status();
}
