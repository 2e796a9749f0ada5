// uvm_sample.h -- one sample of TexturesUV.sample_textures with maps_ids (several texture maps per mesh).
//
// pytorch3d/renderer/mesh/textures.py:1270-1313 hands (u, v, map index) to a 3-D F.grid_sample over the mesh's maps
// (N, C, M, Hm, Wm): the map index is normalised like a coordinate (z = 2 id / (M - 1) - 1) and un-normalised again by
// the sampler, so with align_corners=False -- or wherever rounding leaves z off an integer -- the "bilinear" mode blends
// neighbouring maps (ATen/native/GridSampler.cpp: grid_sampler_3d_cpu, corners tnw tne tsw tse bnw bne bsw bse).  That
// arithmetic is restated here per sample, in plain C++ so that tests/hostgeom runs the very same code on the CPU; the
// kernels (texture_multi.hip) add only the indexing and the atomics.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define P3D_UVM_HD __host__ __device__ inline
#else
#define P3D_UVM_HD inline
#endif

namespace p3d {
namespace uvm {

// torch.lerp (ATen/native/Lerp.h)
P3D_UVM_HD float lerp_t(float start, float end, float w) {
  const float diff = end - start;
  return fabsf(w) < 0.5f ? start + w * diff : end - diff * (1.0f - w);
}

// GridSampler.h: grid_sampler_compute_source_index_set_grad, padding zeros / border.  *mult = d index / d coordinate
P3D_UVM_HD float source_index(float coord, int size, bool align, bool border, float* mult) {
  float x, m;
  if (align) {
    m = (float)(size - 1) / 2.0f;
    x = ((coord + 1.0f) / 2.0f) * (float)(size - 1);
  } else {
    m = (float)size / 2.0f;
    x = ((coord + 1.0f) * (float)size - 1.0f) / 2.0f;
  }
  if (border) {
    if (x <= 0.0f) {
      x = 0.0f;
      m = 0.0f;
    } else if (x >= (float)(size - 1)) {
      x = (float)(size - 1);
      m = 0.0f;
    }
  }
  *mult = m;
  return x;
}

struct Volume {
  int M, Hm, Wm, C;
  bool align, border, nearest;
};

struct Coords {
  float ix, iy, iz;  // un-normalised sample position (x: map column, y: map row, z: map)
  float mx, my;      // d ix / d grid x, d iy / d grid y
};

P3D_UVM_HD Coords coords_of(const Volume& v, float u, float vv, int64_t map_id) {
  Coords c;
  float mz;
  const float gz = ((2.0f * (float)map_id) / (float)(v.M - 1)) - 1.0f;  // textures.py:1291
  c.ix = source_index(lerp_t(-1.0f, 1.0f, u), v.Wm, v.align, v.border, &c.mx);
  c.iy = source_index(lerp_t(1.0f, -1.0f, vv), v.Hm, v.align, v.border, &c.my);
  c.iz = source_index(gz, v.M, v.align, v.border, &mz);
  return c;
}

P3D_UVM_HD bool inside(const Volume& v, int x, int y, int z) {
  return x >= 0 && x < v.Wm && y >= 0 && y < v.Hm && z >= 0 && z < v.M;
}

P3D_UVM_HD int64_t texel_offset(const Volume& v, int x, int y, int z) {
  return (((int64_t)z * v.Hm + y) * v.Wm + x) * v.C;
}

// The (up to) 8 texels a sample touches and their weights; nearest mode: corner 0 only, weight 1.  Fixed-size arrays
// walked by fully unrolled loops under a validity mask: nothing is indexed dynamically (no private-memory arrays).
struct Footprint {
  unsigned mask;   // bit k: corner k lies inside the volume
  int64_t off[8];  // texel offsets inside the mesh's (M, Hm, Wm, C) volume
  float w[8];      // interpolation weights
  float dx[8];     // d weight / d ix  (0 in nearest mode)
  float dy[8];     // d weight / d iy
};

P3D_UVM_HD Footprint footprint_of(const Volume& v, const Coords& c) {
  Footprint f;
  f.mask = 0u;
  for (int k = 0; k < 8; ++k) {
    f.off[k] = 0;
    f.w[k] = f.dx[k] = f.dy[k] = 0.0f;
  }
  if (v.nearest) {
    const int x = (int)nearbyintf(c.ix), y = (int)nearbyintf(c.iy), z = (int)nearbyintf(c.iz);
    if (inside(v, x, y, z)) {
      f.off[0] = texel_offset(v, x, y, z);
      f.w[0] = 1.0f;
      f.mask = 1u;
    }
    return f;
  }
  const float fx = floorf(c.ix), fy = floorf(c.iy), fz = floorf(c.iz);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const float wx0 = fx + 1.0f - c.ix, wx1 = c.ix - fx, wy0 = fy + 1.0f - c.iy, wy1 = c.iy - fy, wz0 = fz + 1.0f - c.iz,
              wz1 = c.iz - fz;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < 8; ++k) {
    const int kx = k & 1, ky = (k >> 1) & 1, kz = k >> 2;
    const int x = x0 + kx, y = y0 + ky, z = z0 + kz;
    if (!inside(v, x, y, z)) continue;
    const float wxk = kx ? wx1 : wx0, wyk = ky ? wy1 : wy0, wzk = kz ? wz1 : wz0;
    f.mask |= 1u << k;
    f.off[k] = texel_offset(v, x, y, z);
    f.w[k] = wxk * wyk * wzk;
    f.dx[k] = (kx ? 1.0f : -1.0f) * wyk * wzk;
    f.dy[k] = (ky ? 1.0f : -1.0f) * wxk * wzk;
  }
  return f;
}

// texel of one sample: out[0..C) (vol = the mesh's maps)
P3D_UVM_HD void sample_forward(const Volume& v, const float* vol, const Footprint& f, float* out) {
  for (int ch = 0; ch < v.C; ++ch) {
    float acc = 0.0f;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k)
      if (f.mask >> k & 1u) acc += vol[f.off[k] + ch] * f.w[k];
    out[ch] = acc;
  }
}

// backward of one sample: scatters g * weight into gvol through `add` and returns d loss / d (u, v)
template <typename Add>
P3D_UVM_HD void sample_backward(const Volume& v, const float* vol, float* gvol, const Footprint& f, const Coords& c,
                                const float* g, Add add, float* du, float* dv) {
  float gix = 0.0f, giy = 0.0f;
  for (int ch = 0; ch < v.C; ++ch) {
    const float gc = g[ch];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k) {
      if (!(f.mask >> k & 1u)) continue;
      add(gvol + f.off[k] + ch, f.w[k] * gc);
      const float val = vol[f.off[k] + ch];
      gix += val * f.dx[k] * gc;
      giy += val * f.dy[k] * gc;
    }
  }
  // d grid / d (u, v) = (2, -2) (the lerp onto [-1, 1] with the y axis flipped), times d index / d grid
  *du = c.mx * gix * 2.0f;
  *dv = c.my * giy * -2.0f;
}

}  // namespace uvm
}  // namespace p3d
