// atlas.hip -- per-face texture atlas sampling of rasterization fragments for gfx950 (TexturesAtlas, the second
// texture class of SURVEY 8(f) row 4).
//
// Replaces TexturesAtlas.sample_textures (pytorch3d/renderer/mesh/textures.py:565-612): a where, a multiply, an int64
// cast, a clamp, two sums, a compare, two more wheres, one advanced-indexing gather of the (F, R, R, C) atlas and a
// mask multiply -- eleven elementwise kernels over (N,H,W,K[,2]) tensors plus the index_put_ of their autograd graph --
// by one kernel each way.  A thread per sample: 8 B of pix_to_face + 8 of the 12 B of barycentrics in, one R x R cell
// (atlas_cell.h) picked, C contiguous floats gathered and written.  Nearest-cell sampling has no gradient to the
// barycentrics (the reference's docstring says so); the backward is the scatter-add of grad_texels into the atlas with
// the hardware float atomic (-munsafe-fp-atomics), as index_put_(accumulate=True) does.
// Both kernels are bound by the streamed fragments: 20 + 4 C bytes per sample each way.
#include "atlas_cell.h"
#include "p3d_common.h"

namespace p3d {
namespace {

constexpr int kAtlasBlock = 256;

struct AtlasArgs {
  const int64_t* p2f;  // (P)
  const float* bary;   // (P, 3)
  const float* atlas;  // (F, R, R, C)
  const float* gtex;   // (P, C)
  float* texels;       // (P, C)
  float* gatlas;       // (F, R, R, C)
  int64_t P, F;
  int R, C;
};

// -> offset of the sample's cell in the atlas, or -1 for background / indices the reference cannot address
__device__ __forceinline__ int64_t cell_offset(const AtlasArgs& a, int64_t i) {
  const int64_t f = a.p2f[i];
  if (f < 0 || f >= a.F) return -1;
  int row, col;
  if (!atlas_cell(a.bary[i * 3], a.bary[i * 3 + 1], a.R, &row, &col)) return -1;
  return ((f * a.R + row) * a.R + col) * a.C;
}

template <int CT>  // CT > 0: channel count known at compile time
__global__ __launch_bounds__(kAtlasBlock) void atlas_fwd_kernel(AtlasArgs a) {
  const int C = CT > 0 ? CT : a.C;
  for (int64_t i = (int64_t)blockIdx.x * kAtlasBlock + threadIdx.x; i < a.P; i += (int64_t)gridDim.x * kAtlasBlock) {
    const int64_t off = cell_offset(a, i);
    float* out = a.texels + i * C;
    if (off < 0) {
      for (int c = 0; c < C; ++c) out[c] = 0.0f;
    } else {
      for (int c = 0; c < C; ++c) out[c] = a.atlas[off + c];
    }
  }
}

template <int CT>
__global__ __launch_bounds__(kAtlasBlock) void atlas_bwd_kernel(AtlasArgs a) {
  const int C = CT > 0 ? CT : a.C;
  for (int64_t i = (int64_t)blockIdx.x * kAtlasBlock + threadIdx.x; i < a.P; i += (int64_t)gridDim.x * kAtlasBlock) {
    const int64_t off = cell_offset(a, i);
    if (off < 0) continue;
    const float* g = a.gtex + i * C;
    for (int c = 0; c < C; ++c) atomicAdd(a.gatlas + off + c, g[c]);
  }
}

unsigned atlas_grid(int64_t P) {
  int64_t g = ceil_div(P, (int64_t)kAtlasBlock * 4);
  if (g < 1) g = 1;
  if (g > 16384) g = 16384;
  return (unsigned)g;
}

int check_atlas(int64_t P, int64_t F, int R, int C) {
  if (P < 0 || F < 0 || R < 1 || C < 1) return P3D_ERR_INVALID_ARG;
  return P3D_OK;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_sample_atlas_forward(const int64_t* pix_to_face, const float* bary, const float* atlas, int64_t P,
                                     int64_t F, int R, int C, float* texels, p3d_stream_t stream) {
  const int rc = check_atlas(P, F, R, C);
  if (rc != P3D_OK) return rc;
  if (P == 0) return P3D_OK;
  if (!pix_to_face || !bary || !texels || (F > 0 && !atlas)) return P3D_ERR_INVALID_ARG;
  AtlasArgs a{};
  a.p2f = pix_to_face;
  a.bary = bary;
  a.atlas = atlas;
  a.texels = texels;
  a.P = P;
  a.F = F;
  a.R = R;
  a.C = C;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("sample_atlas_fwd", s);
  if (C == 3)
    atlas_fwd_kernel<3><<<atlas_grid(P), kAtlasBlock, 0, s>>>(a);
  else
    atlas_fwd_kernel<0><<<atlas_grid(P), kAtlasBlock, 0, s>>>(a);
  return launch_status();
}

P3D_API int p3d_sample_atlas_backward(const float* grad_texels, const int64_t* pix_to_face, const float* bary, int64_t P,
                                      int64_t F, int R, int C, float* grad_atlas, p3d_stream_t stream) {
  const int rc = check_atlas(P, F, R, C);
  if (rc != P3D_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (F > 0) {
    if (!grad_atlas) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(grad_atlas, 0, (size_t)F * R * R * C * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  if (P == 0 || F == 0) return P3D_OK;
  if (!grad_texels || !pix_to_face || !bary) return P3D_ERR_INVALID_ARG;
  AtlasArgs a{};
  a.gtex = grad_texels;
  a.p2f = pix_to_face;
  a.bary = bary;
  a.gatlas = grad_atlas;
  a.P = P;
  a.F = F;
  a.R = R;
  a.C = C;
  LaunchScope ls("sample_atlas_bwd", s);
  if (C == 3)
    atlas_bwd_kernel<3><<<atlas_grid(P), kAtlasBlock, 0, s>>>(a);
  else
    atlas_bwd_kernel<0><<<atlas_grid(P), kAtlasBlock, 0, s>>>(a);
  return launch_status();
}
