// interp.hip -- interpolate_face_attributes for gfx950.
//
// Replaces InterpFaceAttrs{Forward,Backward}Kernel (pytorch3d/csrc/interp_face_attrs/
// interp_face_attrs.cu:15-40, 86-115).  64-bit element indexing throughout (the reference's
// `int pd` loop counter overflows past 2^31 elements, interp_face_attrs.cu:25); every output
// element is written (zeros where pix_to_face < 0), so callers pass uninitialised memory;
// grad_barycentric_coords is reduced over D in registers and written once, only
// grad_face_attrs (a scatter over faces) uses atomics.
#include "p3d_common.h"

namespace p3d {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void interp_fwd_kernel(const int64_t* __restrict__ p2f, const T* __restrict__ bary,
                                                         const T* __restrict__ attrs, int64_t P, int64_t D,
                                                         T* __restrict__ out) {
  const int64_t total = P * D;
  for (int64_t pd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pd < total; pd += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = pd / D;
    const int64_t d = pd - p * D;
    const int64_t f = p2f[p];
    T v = T(0);
    if (f >= 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) v += bary[p * 3 + i] * attrs[f * 3 * D + i * D + d];
    }
    out[pd] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void interp_bwd_kernel(const int64_t* __restrict__ p2f, const T* __restrict__ bary,
                                                         const T* __restrict__ attrs, const T* __restrict__ gout,
                                                         int64_t P, int64_t D, T* __restrict__ gbary,
                                                         T* __restrict__ gattrs) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = p2f[p];
    T g0 = T(0), g1 = T(0), g2 = T(0);
    if (f >= 0) {
      const T w0 = bary[p * 3 + 0], w1 = bary[p * 3 + 1], w2 = bary[p * 3 + 2];
      const T* a = attrs + f * 3 * D;
      T* ga = gattrs + f * 3 * D;
      for (int64_t d = 0; d < D; ++d) {
        const T up = gout[p * D + d];
        g0 += a[d] * up;
        g1 += a[D + d] * up;
        g2 += a[2 * D + d] * up;
        unsafeAtomicAdd(ga + d, w0 * up);
        unsafeAtomicAdd(ga + D + d, w1 * up);
        unsafeAtomicAdd(ga + 2 * D + d, w2 * up);
      }
    }
    gbary[p * 3 + 0] = g0;
    gbary[p * 3 + 1] = g1;
    gbary[p * 3 + 2] = g2;
  }
}

unsigned pick_grid(int64_t n) {
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 32768) blocks = 32768;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_interp_face_attrs_forward(int dtype, const int64_t* p2f, const void* bary, const void* attrs, int64_t P,
                                          int64_t F, int64_t D, void* out, p3d_stream_t stream) {
  if (P < 0 || F < 0 || D < 0 || (dtype != 0 && dtype != 1)) return P3D_ERR_INVALID_ARG;
  if (P * D == 0) return P3D_OK;
  if (!p2f || !bary || !out || (F > 0 && !attrs)) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("interp_fwd", s);
  if (dtype == 0)
    interp_fwd_kernel<float><<<pick_grid(P * D), 256, 0, s>>>(p2f, (const float*)bary, (const float*)attrs, P, D,
                                                             (float*)out);
  else
    interp_fwd_kernel<double><<<pick_grid(P * D), 256, 0, s>>>(p2f, (const double*)bary, (const double*)attrs, P, D,
                                                              (double*)out);
  return launch_status();
}

P3D_API int p3d_interp_face_attrs_backward(int dtype, const int64_t* p2f, const void* bary, const void* attrs,
                                           const void* gout, int64_t P, int64_t F, int64_t D, void* gbary, void* gattrs,
                                           p3d_stream_t stream) {
  if (P < 0 || F < 0 || D < 0 || (dtype != 0 && dtype != 1)) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype == 0 ? sizeof(float) : sizeof(double);
  if (F * D > 0) {
    if (!gattrs) return P3D_ERR_INVALID_ARG;
    if (hipMemsetAsync(gattrs, 0, (size_t)F * 3 * D * esz, s) != hipSuccess) return P3D_ERR_LAUNCH;
  }
  if (P == 0) return P3D_OK;
  if (!p2f || !bary || !gbary || (D > 0 && !gout) || (F > 0 && D > 0 && !attrs)) return P3D_ERR_INVALID_ARG;
  LaunchScope ls("interp_bwd", s);
  if (dtype == 0)
    interp_bwd_kernel<float><<<pick_grid(P), 256, 0, s>>>(p2f, (const float*)bary, (const float*)attrs,
                                                         (const float*)gout, P, D, (float*)gbary, (float*)gattrs);
  else
    interp_bwd_kernel<double><<<pick_grid(P), 256, 0, s>>>(p2f, (const double*)bary, (const double*)attrs,
                                                          (const double*)gout, P, D, (double*)gbary, (double*)gattrs);
  return launch_status();
}
