// texture_multi.hip -- UV texture sampling with several maps per mesh (TexturesUV with maps_ids) for gfx950.
//
// Replaces the maps_ids branch of TexturesUV.sample_textures (pytorch3d/renderer/mesh/textures.py:1270-1313): a gather
// of the per-face map index, its normalisation to a grid coordinate, a cat, a permute of the (N, M, Hm, Wm, C) maps to
// (N, C, M, Hm, Wm) and a 3-D F.grid_sample, plus the autograd graph of all of it, by one kernel each way that reads
// the maps where they lie.  The per-sample arithmetic is uvm_sample.h (the 3-D sampler restated, including its blend
// of neighbouring maps when the un-normalised map coordinate is not an integer).  A thread per sample.  Backward:
// the map gradient goes out as float atomics like grid_sampler_3d_backward's; background samples (pix_to_face < 0) of
// one image all read the same footprint -- face 0's map at uv = (0, 0), textures.py:1279-1281 -- so their gradient is
// summed across the wave first and leaves as one set of atomics per 64 samples.  This is the plain form of
// texture.hip's backward (no wave tables): the single-map case is the one tuned for the bench-size fragments.
#include "p3d_common.h"
#include "uvm_sample.h"

namespace p3d {
namespace {

struct UvmArgs {
  const int64_t* p2f;  // (N, HWK)
  const float* bary;   // (N, HWK, 3)
  const float* fuv;    // (F, 3, 2)
  const float* maps;   // (N, M, Hm, Wm, C)
  const int64_t* ids;  // (L) flattened maps_ids_padded, indexed by packed face index
  const float* gtex;   // (N, HWK, C)
  float* texels;       // (N, HWK, C)
  float* gbary;        // (N, HWK, 3)
  float* gfuv;         // (F, 3, 2)
  float* gmaps;        // (N, M, Hm, Wm, C)
  int64_t L, F, HWK;
  uvm::Volume v;
};

struct Sample {
  int64_t f;      // pix_to_face
  bool valid;     // the face has a map index
  int64_t id;     // its map
  float u, v;     // interpolated uv
  float b[3], r[6];
};

__device__ __forceinline__ Sample load_sample(const UvmArgs& a, int64_t p, bool ok) {
  Sample s;
  s.f = ok ? a.p2f[p] : -1;
  s.u = s.v = 0.0f;
#pragma unroll
  for (int j = 0; j < 3; ++j) s.b[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < 6; ++j) s.r[j] = 0.0f;
  if (s.f >= 0 && s.f < a.F) {
#pragma unroll
    for (int j = 0; j < 3; ++j) s.b[j] = a.bary[p * 3 + j];
    const float* rp = a.fuv + s.f * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) s.r[j] = rp[j];
    s.u = (s.b[0] * s.r[0] + s.b[1] * s.r[2]) + s.b[2] * s.r[4];  // interp_face_attrs.cu:39-41
    s.v = (s.b[0] * s.r[1] + s.b[1] * s.r[3]) + s.b[2] * s.r[5];
  }
  const int64_t fi = s.f < 0 ? 0 : s.f;  // textures.py:1279-1281: background samples look up face 0's map
  s.valid = ok && fi < a.L && (s.f < 0 || s.f < a.F);
  s.id = s.valid ? a.ids[fi] : 0;
  return s;
}

__global__ __launch_bounds__(256) void sample_uvm_fwd_kernel(UvmArgs a) {
  const int n = blockIdx.y;
  const int C = a.v.C;
  const float* vol = a.maps + (int64_t)n * a.v.M * a.v.Hm * a.v.Wm * C;
  const int64_t img = (int64_t)n * a.HWK;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.HWK; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = img + i;
    const Sample s = load_sample(a, p, true);
    float* out = a.texels + p * C;
    if (!s.valid) {
      for (int ch = 0; ch < C; ++ch) out[ch] = 0.0f;
      continue;
    }
    const uvm::Coords c = uvm::coords_of(a.v, s.u, s.v, s.id);
    const uvm::Footprint fp = uvm::footprint_of(a.v, c);
    uvm::sample_forward(a.v, vol, fp, out);
  }
}

struct AtomicAdd {
  __device__ __forceinline__ void operator()(float* dst, float v) const { unsafeAtomicAdd(dst, v); }
};

__global__ __launch_bounds__(256) void sample_uvm_bwd_kernel(UvmArgs a) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.y;
  const int C = a.v.C;
  const int64_t per = (int64_t)a.v.M * a.v.Hm * a.v.Wm * C;
  const float* vol = a.maps + (int64_t)n * per;
  float* gvol = a.gmaps + (int64_t)n * per;
  const int64_t img = (int64_t)n * a.HWK;
  // the footprint every background sample of this image reads
  const bool bg_valid = a.L > 0;
  uvm::Footprint bgfp;
  {
    const uvm::Coords cb = uvm::coords_of(a.v, 0.0f, 0.0f, bg_valid ? a.ids[0] : 0);
    bgfp = uvm::footprint_of(a.v, cb);
  }
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * 64; base < a.HWK; base += nwaves * 64) {  // wave-uniform trip count
    const int64_t i = base + lane;
    const bool ok = i < a.HWK;
    const int64_t p = img + i;
    const Sample s = load_sample(a, p, ok);
    const float* g = a.gtex + p * C;
    float gb[3] = {0.0f, 0.0f, 0.0f};
    const bool bg = ok && s.f < 0 && bg_valid;
    if (s.valid && s.f >= 0) {
      const uvm::Coords c = uvm::coords_of(a.v, s.u, s.v, s.id);
      const uvm::Footprint fp = uvm::footprint_of(a.v, c);
      float du, dv;
      uvm::sample_backward(a.v, vol, gvol, fp, c, g, AtomicAdd(), &du, &dv);
      if (!a.v.nearest) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          unsafeAtomicAdd(a.gfuv + s.f * 6 + 2 * j, s.b[j] * du);
          unsafeAtomicAdd(a.gfuv + s.f * 6 + 2 * j + 1, s.b[j] * dv);
          gb[j] = s.r[2 * j] * du + s.r[2 * j + 1] * dv;
        }
      }
    }
    if (ok) {
#pragma unroll
      for (int j = 0; j < 3; ++j) a.gbary[p * 3 + j] = gb[j];
    }
    // background lanes: one reduction per channel, then lane 0 scatters the sum through the shared footprint
    if (__ballot(bg) == 0) continue;  // wave-uniform
    for (int ch = 0; ch < C; ++ch) {
      float t = bg ? g[ch] : 0.0f;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
      if (lane != 0 || t == 0.0f) continue;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (bgfp.mask >> k & 1u) unsafeAtomicAdd(gvol + bgfp.off[k] + ch, bgfp.w[k] * t);
    }
  }
}

int check_uvm(int N, int H, int W, int K, int64_t F, int64_t L, int M, int Hm, int Wm, int C, int padding_mode,
              int sampling_mode) {
  if (N < 0 || H < 0 || W < 0 || K < 0 || F < 0 || L < 0 || Hm < 1 || Wm < 1 || C < 1) return P3D_ERR_INVALID_ARG;
  if (M < 2) return P3D_ERR_INVALID_ARG;  // the reference divides by M - 1 (textures.py:1291)
  if (padding_mode != P3D_PAD_ZEROS && padding_mode != P3D_PAD_BORDER) return P3D_ERR_INVALID_ARG;
  if (sampling_mode != P3D_SAMPLE_BILINEAR && sampling_mode != P3D_SAMPLE_NEAREST) return P3D_ERR_INVALID_ARG;
  if (N > 65535) return P3D_ERR_INVALID_ARG;
  return P3D_OK;
}

dim3 uvm_grid(int64_t HWK, int N) {
  int64_t bx = ceil_div(HWK, 256 * 4);
  if (bx < 1) bx = 1;
  if (bx > 16384) bx = 16384;
  return dim3((unsigned)bx, (unsigned)N);
}

UvmArgs uvm_args(const int64_t* p2f, const float* bary, const float* fuv, const float* maps, const int64_t* ids, int64_t L,
                 int64_t F, int64_t HWK, int M, int Hm, int Wm, int C, int align, int padding_mode, int sampling_mode) {
  UvmArgs a{};
  a.p2f = p2f;
  a.bary = bary;
  a.fuv = fuv;
  a.maps = maps;
  a.ids = ids;
  a.L = L;
  a.F = F;
  a.HWK = HWK;
  a.v.M = M;
  a.v.Hm = Hm;
  a.v.Wm = Wm;
  a.v.C = C;
  a.v.align = align != 0;
  a.v.border = padding_mode == P3D_PAD_BORDER;
  a.v.nearest = sampling_mode == P3D_SAMPLE_NEAREST;
  return a;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_sample_uv_multi_forward(const int64_t* pix_to_face, const float* bary, const float* face_uvs,
                                        const float* maps, const int64_t* maps_ids, int64_t L, int N, int H, int W, int K,
                                        int64_t F, int M, int Hm, int Wm, int C, int align_corners, int padding_mode,
                                        int sampling_mode, float* texels, p3d_stream_t stream) {
  const int rc = check_uvm(N, H, W, K, F, L, M, Hm, Wm, C, padding_mode, sampling_mode);
  if (rc != P3D_OK) return rc;
  const int64_t HWK = (int64_t)H * W * K;
  if ((int64_t)N * HWK == 0) return P3D_OK;
  if (!pix_to_face || !bary || !maps || !texels || (F > 0 && !face_uvs) || (L > 0 && !maps_ids)) return P3D_ERR_INVALID_ARG;
  UvmArgs a = uvm_args(pix_to_face, bary, face_uvs, maps, maps_ids, L, F, HWK, M, Hm, Wm, C, align_corners, padding_mode,
                       sampling_mode);
  a.texels = texels;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("sample_uv_multi_fwd", s);
  sample_uvm_fwd_kernel<<<uvm_grid(HWK, N), 256, 0, s>>>(a);
  return launch_status();
}

P3D_API int p3d_sample_uv_multi_backward(const float* grad_texels, const int64_t* pix_to_face, const float* bary,
                                         const float* face_uvs, const float* maps, const int64_t* maps_ids, int64_t L, int N,
                                         int H, int W, int K, int64_t F, int M, int Hm, int Wm, int C, int align_corners,
                                         int padding_mode, int sampling_mode, float* grad_bary, float* grad_face_uvs,
                                         float* grad_maps, p3d_stream_t stream) {
  const int rc = check_uvm(N, H, W, K, F, L, M, Hm, Wm, C, padding_mode, sampling_mode);
  if (rc != P3D_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (F > 0) {
    if (!grad_face_uvs) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_face_uvs, 0, (size_t)F * 6 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  if (N > 0) {
    if (!grad_maps) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_maps, 0, (size_t)N * M * Hm * Wm * C * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  const int64_t HWK = (int64_t)H * W * K;
  if ((int64_t)N * HWK == 0) return P3D_OK;
  if (!grad_texels || !pix_to_face || !bary || !maps || !grad_bary || (F > 0 && !face_uvs) || (L > 0 && !maps_ids))
    return P3D_ERR_INVALID_ARG;
  UvmArgs a = uvm_args(pix_to_face, bary, face_uvs, maps, maps_ids, L, F, HWK, M, Hm, Wm, C, align_corners, padding_mode,
                       sampling_mode);
  a.gtex = grad_texels;
  a.gbary = grad_bary;
  a.gfuv = grad_face_uvs;
  a.gmaps = grad_maps;
  LaunchScope ls("sample_uv_multi_bwd", s);
  sample_uvm_bwd_kernel<<<uvm_grid(HWK, N), 256, 0, s>>>(a);
  return launch_status();
}
