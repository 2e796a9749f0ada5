// atlas_cell.h -- the texel of a per-face texture atlas that a barycentric sample reads.
//
// TexturesAtlas.sample_textures (pytorch3d/renderer/mesh/textures.py:565-612) restated as one function of the two first
// barycentric coordinates; plain C++ so that tests/hostgeom compiles the very same code for the CPU checks.
//   w_xy       = (bary[:2] * R).to(int64).clamp(max = R - 1)          truncation toward zero, no lower clamp
//   below_diag = (bary[0] + bary[1]) * R - (float(w_x) + float(w_y)) <= 1
//   w          = below_diag ? w : R - 1 - w
//   texel      = atlas[face, w_y, w_x]                                 negative indices wrap once, as torch indexing does
// Every product and sum is a separate float32 operation there (torch ops); the library is built with -ffp-contract=off.
// Where torch raises an IndexError (an index outside [-R, R-1], only reachable with barycentrics below -1 or NaN) the
// sample is reported as not addressable and reads as zero.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define P3D_HD __host__ __device__ inline
#else
#define P3D_HD inline
#endif

namespace p3d {

// float -> int64 as torch's .to(int64) on the device: truncation; out-of-range and NaN inputs saturate to a value that
// fails the range check below (static_cast would be undefined behaviour for them)
P3D_HD int64_t atlas_trunc(float v) {
  if (!(v > -9.0e18f && v < 9.0e18f)) return INT64_MIN / 2;
  return (int64_t)v;
}

// -> true and the (row, column) inside the R x R grid of the face, or false when the reference would fail to index
P3D_HD bool atlas_cell(float b0, float b1, int R, int* row, int* col) {
  const float r = (float)R;
  int64_t wx = atlas_trunc(b0 * r), wy = atlas_trunc(b1 * r);
  if (wx > R - 1) wx = R - 1;
  if (wy > R - 1) wy = R - 1;
  const float lhs = (b0 + b1) * r;
  const float rhs = (float)wx + (float)wy;
  const bool below = (lhs - rhs) <= 1.0f;
  if (!below) {
    wx = R - 1 - wx;
    wy = R - 1 - wy;
  }
  if (wx < 0) wx += R;
  if (wy < 0) wy += R;
  if (wx < 0 || wx >= R || wy < 0 || wy >= R) return false;
  *row = (int)wy;
  *col = (int)wx;
  return true;
}

}  // namespace p3d
