// texture.hip -- fused UV texture sampling of rasterization fragments for gfx950 (TexturesUV).
//
// Replaces TexturesUV.sample_textures (pytorch3d/renderer/mesh/textures.py:1190-1268) for one map per mesh:
//   pixel_uvs = interpolate_face_attributes(pix_to_face, bary, verts_uvs[faces_uvs])     (N,H,W,K,2)
//   grid      = lerp([-1, 1], [1, -1], pixel_uvs)                                        x = 2u - 1, y = 1 - 2v
//   texels    = F.grid_sample(maps as (N*K, C, Hm, Wm), grid, mode, align_corners, padding_mode) -> (N,H,W,K,C)
// i.e. an interpolation kernel, two permute / expand copies of the maps (K copies of every map), the grid_sample
// kernel and a permute back, plus their autograd twins.  Here one kernel each way reads the maps in their own
// (N, Hm, Wm, C) layout: a thread per sample interpolates uv, applies the grid_sample arithmetic
// (ATen/native/GridSampler.h: unnormalize, clip / zero padding, bilinear or nearest) and gathers C contiguous floats
// per corner.  Backward: gradient of the maps with global atomics (as grid_sampler_2d_backward does), of the per-face
// uvs through a wave-private LDS table (wave_table.h, 6 partials per face), of the barycentrics written per sample.
// Background samples (pix_to_face < 0) interpolate to uv = (0, 0) and sample the map there -- like the reference.
#include "p3d_common.h"
#include "wave_table.h"

namespace p3d {
namespace {

struct TexArgs {
  const int64_t* p2f;
  const float* bary;
  const float* fuv;    // (F, 3, 2)
  const float* maps;   // (N, Hm, Wm, C)
  const float* gtex;   // (N,H,W,K,C)
  float* texels;       // (N,H,W,K,C)
  float* gbary;        // (N,H,W,K,3)
  float* gfuv;         // (F, 3, 2)
  float* gmaps;        // (N, Hm, Wm, C)
  int N, Hm, Wm, C;
  int align, border, nearest;
  int64_t HWK;
};

// torch.lerp(start, end, w) (ATen/native/Lerp.h): start + w * (end - start) where |w| < 0.5, end - (end - start) * (1 - w) elsewhere
__device__ __forceinline__ float lerp_t(float start, float end, float w) {
  const float diff = end - start;
  return fabsf(w) < 0.5f ? start + w * diff : end - diff * (1.0f - w);
}

// GridSampler.h: grid_sampler_compute_source_index_set_grad for padding zeros / border.  mult = d index / d coord.
__device__ __forceinline__ float source_index(float coord, int size, bool align, bool border, float* mult) {
  float x, m;
  if (align) {
    m = (float)(size - 1) / 2.0f;
    x = ((coord + 1.0f) / 2.0f) * (float)(size - 1);
  } else {
    m = (float)size / 2.0f;
    x = ((coord + 1.0f) * (float)size - 1.0f) / 2.0f;
  }
  if (border) {  // clip_coordinates_set_grad
    if (x <= 0.0f) {
      x = 0.0f;
      m = 0.0f;
    } else {
      const float mx = (float)(size - 1);
      if (x >= mx) {
        x = mx;
        m = 0.0f;
      }
    }
  }
  *mult = m;
  return x;
}

struct Corners {
  int x0, y0;            // north-west corner
  float w[4];            // nw, ne, sw, se
  float ix, iy, mx, my;  // source index and d index / d grid coordinate
};

__device__ __forceinline__ Corners corners_of(float u, float v, int Hm, int Wm, bool align, bool border) {
  Corners c;
  const float gx = lerp_t(-1.0f, 1.0f, u), gy = lerp_t(1.0f, -1.0f, v);
  c.ix = source_index(gx, Wm, align, border, &c.mx);
  c.iy = source_index(gy, Hm, align, border, &c.my);
  const float fx = floorf(c.ix), fy = floorf(c.iy);
  c.x0 = (int)fx;
  c.y0 = (int)fy;
  const float x1 = fx + 1.0f, y1 = fy + 1.0f;
  c.w[0] = (x1 - c.ix) * (y1 - c.iy);
  c.w[1] = (c.ix - fx) * (y1 - c.iy);
  c.w[2] = (x1 - c.ix) * (c.iy - fy);
  c.w[3] = (c.ix - fx) * (c.iy - fy);
  return c;
}

__device__ __forceinline__ bool inside(int x, int y, int Hm, int Wm) { return x >= 0 && x < Wm && y >= 0 && y < Hm; }

template <bool NEAREST>
__global__ __launch_bounds__(256) void sample_uv_fwd_kernel(TexArgs a) {
  const int n = blockIdx.y;
  const int C = a.C, Hm = a.Hm, Wm = a.Wm;
  const float* map = a.maps + (int64_t)n * Hm * Wm * C;
  const int64_t img = (int64_t)n * a.HWK;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.HWK; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = img + i;
    const int f = (int)a.p2f[p];
    float u = 0.0f, v = 0.0f;
    if (f >= 0) {
      const float b0 = a.bary[p * 3], b1 = a.bary[p * 3 + 1], b2 = a.bary[p * 3 + 2];
      const float* r = a.fuv + (int64_t)f * 6;
      u = (b0 * r[0] + b1 * r[2]) + b2 * r[4];  // interp_face_attrs.cu:39-41
      v = (b0 * r[1] + b1 * r[3]) + b2 * r[5];
    }
    const Corners c = corners_of(u, v, Hm, Wm, a.align != 0, a.border != 0);
    float* out = a.texels + p * C;
    if (NEAREST) {
      const int xn = (int)nearbyintf(c.ix), yn = (int)nearbyintf(c.iy);
      const bool in = inside(xn, yn, Hm, Wm);
      const float* src = map + ((int64_t)yn * Wm + xn) * C;
      for (int ch = 0; ch < C; ++ch) out[ch] = in ? src[ch] : 0.0f;
    } else {
      const bool in[4] = {inside(c.x0, c.y0, Hm, Wm), inside(c.x0 + 1, c.y0, Hm, Wm), inside(c.x0, c.y0 + 1, Hm, Wm),
                          inside(c.x0 + 1, c.y0 + 1, Hm, Wm)};
      const float* s0 = map + ((int64_t)c.y0 * Wm + c.x0) * C;
      const float* s2 = s0 + (int64_t)Wm * C;
      for (int ch = 0; ch < C; ++ch) {
        float acc = 0.0f;  // GridSampler.cpp: nw, ne, sw, se in this order
        if (in[0]) acc += s0[ch] * c.w[0];
        if (in[1]) acc += s0[C + ch] * c.w[1];
        if (in[2]) acc += s2[ch] * c.w[2];
        if (in[3]) acc += s2[C + ch] * c.w[3];
        out[ch] = acc;
      }
    }
  }
}

// Per wave: 128 slots for the per-face uv partials (6 values, 40 B) + 320 slots for the map gradient keyed by texel (up to
// 4 channels, 24 B): 12.5 KB -> 50 KB per workgroup, three workgroups per CU.
using UvTable = WaveTable<6, 128, kRows>;
using TexelTable = WaveTable<4, 320, kChunk>;

template <bool NEAREST>
__global__ __launch_bounds__(256) void sample_uv_bwd_kernel(TexArgs a, int64_t span) {
  __shared__ __align__(16) int s_table[4][UvTable::kLdsInts];
  __shared__ __align__(16) int s_texels[4][TexelTable::kLdsInts];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int C = a.C, Hm = a.Hm, Wm = a.Wm;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w;
  const int64_t begin = wave * span;
  if (begin >= a.HWK) return;  // wave-uniform; no workgroup barrier in this kernel
  const int64_t end = begin + span < a.HWK ? begin + span : a.HWK;
  const float* map = a.maps + (int64_t)n * Hm * Wm * C;
  float* gmap = a.gmaps + (int64_t)n * Hm * Wm * C;
  const int64_t img = (int64_t)n * a.HWK;
  UvTable tab;
  tab.init(s_table[w], lane);
  // Map gradient of the samples with a face: neighbouring pixels read overlapping 2x2 texel footprints, so lanes of one
  // instruction would hit the same addresses with atomics, which serialises them (6.4 ms for 34 M samples).  For
  // C <= 4 the four corner contributions go through a second wave table keyed by the texel instead, and reach
  // memory as one atomic per (wave span, texel, channel).
  TexelTable ttab;
  ttab.init(s_texels[w], lane);
  ttab.plane = C;
  ttab.pitch = 0;
  ttab.nlive = C;
  const bool merge = C <= 4;
  // Background samples (most of an image) all sample the map at uv = (0, 0): their gradient is summed per lane and
  // leaves the wave as ONE atomic per corner and channel at the end -- scattered per sample, millions of atomics would
  // serialise on a single texel (measured: 209 ms for 34 M samples).  Channels beyond the fourth take the slow road.
  constexpr int kBgCh = 4;
  float bgacc[kBgCh] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t base = begin; base < end; base += 64) {
    const int64_t i = base + lane;
    const bool ok = i < end;
    const int64_t p = img + i;
    const int f = ok ? (int)a.p2f[p] : -1;
    float b[3] = {0.f, 0.f, 0.f}, r[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float u = 0.0f, v = 0.0f;
    if (f >= 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = a.bary[p * 3 + j];
      const float* rp = a.fuv + (int64_t)f * 6;
#pragma unroll
      for (int j = 0; j < 6; ++j) r[j] = rp[j];
      u = (b[0] * r[0] + b[1] * r[2]) + b[2] * r[4];
      v = (b[0] * r[1] + b[1] * r[3]) + b[2] * r[5];
    }
    float gix = 0.0f, giy = 0.0f;
    Corners c;
    int tkey[4] = {-1, -1, -1, -1};  // texel of each corner (merge path), -1: none
    float tval[4][4];
    if (ok && f < 0) {
      const float* g = a.gtex + p * C;
#pragma unroll
      for (int ch = 0; ch < kBgCh; ++ch)
        if (ch < C) bgacc[ch] += g[ch];
    }
    if (ok && (f >= 0 || C > kBgCh)) {
      const int ch0 = f >= 0 ? 0 : kBgCh;
      c = corners_of(u, v, Hm, Wm, a.align != 0, a.border != 0);
      const float* g = a.gtex + p * C;
      if (NEAREST) {
        const int xn = (int)nearbyintf(c.ix), yn = (int)nearbyintf(c.iy);
        if (inside(xn, yn, Hm, Wm)) {
          if (merge) {
            tkey[0] = yn * Wm + xn;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) tval[0][ch] = ch < C ? g[ch] : 0.0f;
          } else {
            float* dst = gmap + ((int64_t)yn * Wm + xn) * C;
            for (int ch = ch0; ch < C; ++ch) unsafeAtomicAdd(dst + ch, g[ch]);
          }
        }
      } else {
        const bool in[4] = {inside(c.x0, c.y0, Hm, Wm), inside(c.x0 + 1, c.y0, Hm, Wm), inside(c.x0, c.y0 + 1, Hm, Wm),
                            inside(c.x0 + 1, c.y0 + 1, Hm, Wm)};
        const int64_t o0 = ((int64_t)c.y0 * Wm + c.x0) * C, o2 = o0 + (int64_t)Wm * C;
        const float fx = (float)c.x0, fy = (float)c.y0, x1 = fx + 1.0f, y1 = fy + 1.0f;
        if (merge) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (in[k]) tkey[k] = (c.y0 + (k >> 1)) * Wm + c.x0 + (k & 1);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) tval[k][ch] = ch < C ? c.w[k] * g[ch] : 0.0f;
          }
        }
        for (int ch = ch0; ch < C; ++ch) {
          const float gc = g[ch];
          // grid_sampler_2d_backward: scatter g * weight, and d out / d (ix, iy) from the corner values
          if (in[0]) {
            if (!merge) unsafeAtomicAdd(gmap + o0 + ch, c.w[0] * gc);
            const float val = map[o0 + ch];
            gix -= val * (y1 - c.iy) * gc;
            giy -= val * (x1 - c.ix) * gc;
          }
          if (in[1]) {
            if (!merge) unsafeAtomicAdd(gmap + o0 + C + ch, c.w[1] * gc);
            const float val = map[o0 + C + ch];
            gix += val * (y1 - c.iy) * gc;
            giy -= val * (c.ix - fx) * gc;
          }
          if (in[2]) {
            if (!merge) unsafeAtomicAdd(gmap + o2 + ch, c.w[2] * gc);
            const float val = map[o2 + ch];
            gix -= val * (c.iy - fy) * gc;
            giy += val * (x1 - c.ix) * gc;
          }
          if (in[3]) {
            if (!merge) unsafeAtomicAdd(gmap + o2 + C + ch, c.w[3] * gc);
            const float val = map[o2 + C + ch];
            gix += val * (c.iy - fy) * gc;
            giy += val * (c.ix - fx) * gc;
          }
        }
      }
    }
    // d grid / d (u, v) = (2, -2) (the lerp), times d index / d grid
    float g6[6];
    float gb[3] = {0.f, 0.f, 0.f};
    if (f >= 0) {
      const float du = NEAREST ? 0.0f : c.mx * gix * 2.0f;
      const float dv = NEAREST ? 0.0f : c.my * giy * -2.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        g6[2 * j] = b[j] * du;
        g6[2 * j + 1] = b[j] * dv;
        gb[j] = r[2 * j] * du + r[2 * j + 1] * dv;
      }
    }
    if (ok) {
#pragma unroll
      for (int j = 0; j < 3; ++j) a.gbary[p * 3 + j] = gb[j];
    }
    if (__ballot(f >= 0) == 0) continue;  // wave-uniform
    tab.add(a.gfuv, lane, f, g6);
    if (merge) {
#pragma unroll
      for (int k = 0; k < (NEAREST ? 1 : 4); ++k) {
        if (__ballot(tkey[k] >= 0) == 0) continue;  // wave-uniform
        ttab.add(gmap, lane, tkey[k], tval[k]);
      }
    }
  }
  if (tab.used > 0) tab.flush(a.gfuv, lane);
  if (ttab.used > 0) ttab.flush(gmap, lane);
  // the background's share of the map gradient
  const Corners cb = corners_of(0.0f, 0.0f, Hm, Wm, a.align != 0, a.border != 0);
#pragma unroll
  for (int ch = 0; ch < kBgCh; ++ch) {
    if (ch >= C) break;
    float t = bgacc[ch];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
    if (lane != 0 || t == 0.0f) continue;
    if (NEAREST) {
      const int xn = (int)nearbyintf(cb.ix), yn = (int)nearbyintf(cb.iy);
      if (inside(xn, yn, Hm, Wm)) unsafeAtomicAdd(gmap + ((int64_t)yn * Wm + xn) * C + ch, t);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = cb.x0 + (k & 1), y = cb.y0 + (k >> 1);
        if (inside(x, y, Hm, Wm)) unsafeAtomicAdd(gmap + ((int64_t)y * Wm + x) * C + ch, cb.w[k] * t);
      }
    }
  }
}

int check_tex(int N, int H, int W, int K, int64_t F, int Hm, int Wm, int C, int padding_mode, int sampling_mode) {
  if (N < 0 || H < 0 || W < 0 || K < 0 || F < 0 || Hm < 1 || Wm < 1 || C < 1) return P3D_ERR_INVALID_ARG;
  if (padding_mode != P3D_PAD_ZEROS && padding_mode != P3D_PAD_BORDER) return P3D_ERR_INVALID_ARG;
  if (sampling_mode != P3D_SAMPLE_BILINEAR && sampling_mode != P3D_SAMPLE_NEAREST) return P3D_ERR_INVALID_ARG;
  if (N > 65535 || (int64_t)Hm * Wm > 0x7fffffffll) return P3D_ERR_INVALID_ARG;
  return P3D_OK;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_sample_uv_forward(const int64_t* pix_to_face, const float* bary, const float* face_uvs, const float* maps,
                                  int N, int H, int W, int K, int64_t F, int Hm, int Wm, int C, int align_corners,
                                  int padding_mode, int sampling_mode, float* texels, p3d_stream_t stream) {
  const int rc = check_tex(N, H, W, K, F, Hm, Wm, C, padding_mode, sampling_mode);
  if (rc != P3D_OK) return rc;
  const int64_t HWK = (int64_t)H * W * K;
  if ((int64_t)N * HWK == 0) return P3D_OK;
  if (!pix_to_face || !bary || !maps || !texels || (F > 0 && !face_uvs)) return P3D_ERR_INVALID_ARG;
  TexArgs a{};
  a.p2f = pix_to_face;
  a.bary = bary;
  a.fuv = face_uvs;
  a.maps = maps;
  a.texels = texels;
  a.N = N;
  a.Hm = Hm;
  a.Wm = Wm;
  a.C = C;
  a.align = align_corners;
  a.border = padding_mode == P3D_PAD_BORDER;
  a.HWK = HWK;
  hipStream_t s = (hipStream_t)stream;
  int64_t bx = ceil_div(HWK, 256 * 4);
  if (bx > 65535) bx = 65535;
  const dim3 grid((unsigned)bx, (unsigned)N);
  LaunchScope ls("sample_uv_fwd", s);
  if (sampling_mode == P3D_SAMPLE_NEAREST)
    sample_uv_fwd_kernel<true><<<grid, 256, 0, s>>>(a);
  else
    sample_uv_fwd_kernel<false><<<grid, 256, 0, s>>>(a);
  return launch_status();
}

P3D_API int p3d_sample_uv_backward(const float* grad_texels, const int64_t* pix_to_face, const float* bary,
                                   const float* face_uvs, const float* maps, int N, int H, int W, int K, int64_t F, int Hm,
                                   int Wm, int C, int align_corners, int padding_mode, int sampling_mode, float* grad_bary,
                                   float* grad_face_uvs, float* grad_maps, p3d_stream_t stream) {
  const int rc = check_tex(N, H, W, K, F, Hm, Wm, C, padding_mode, sampling_mode);
  if (rc != P3D_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (F > 0) {
    if (!grad_face_uvs) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_face_uvs, 0, (size_t)F * 6 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  if (N > 0) {
    if (!grad_maps) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_maps, 0, (size_t)N * Hm * Wm * C * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  const int64_t HWK = (int64_t)H * W * K;
  if ((int64_t)N * HWK == 0) return P3D_OK;
  if (!grad_texels || !pix_to_face || !bary || !maps || !grad_bary || (F > 0 && !face_uvs)) return P3D_ERR_INVALID_ARG;
  TexArgs a{};
  a.gtex = grad_texels;
  a.p2f = pix_to_face;
  a.bary = bary;
  a.fuv = face_uvs;
  a.maps = maps;
  a.gbary = grad_bary;
  a.gfuv = grad_face_uvs;
  a.gmaps = grad_maps;
  a.N = N;
  a.Hm = Hm;
  a.Wm = Wm;
  a.C = C;
  a.align = align_corners;
  a.border = padding_mode == P3D_PAD_BORDER;
  a.HWK = HWK;
  int64_t waves = ceil_div(HWK, 4096);  // >= 4096 samples per wave amortise the final flush
  if (waves > 4 * 16384) waves = 4 * 16384;
  const int64_t bx = ceil_div(waves, 4);
  const int64_t span = ceil_div(ceil_div(HWK, bx * 4), 64) * 64;
  const dim3 grid((unsigned)bx, (unsigned)N);
  LaunchScope ls("sample_uv_bwd", s);
  if (sampling_mode == P3D_SAMPLE_NEAREST)
    sample_uv_bwd_kernel<true><<<grid, 256, 0, s>>>(a, span);
  else
    sample_uv_bwd_kernel<false><<<grid, 256, 0, s>>>(a, span);
  return launch_status();
}
