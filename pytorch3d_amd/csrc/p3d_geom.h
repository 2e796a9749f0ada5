// p3d_geom.h -- the arithmetic contract of the rasterization hot path.
//
// Every function here is a per-(pixel, primitive) scalar routine shared by all
// gfx950 kernels in this directory.  The expression trees follow the CUDA
// variant of the reference (pytorch3d/csrc/utils/geometry_utils.cuh:37-462,
// pytorch3d/csrc/rasterize_points/rasterization_utils.cuh:16-42) operation by
// operation, because pix_to_face must come out bit-exact: no FMA contraction
// (the translation units are compiled with -ffp-contract=off), IEEE division and
// sqrt, and the reference's mixed float/double epsilon arithmetic
// (`const auto kEpsilon = 1e-8` is a double, geometry_utils.cuh:18).
//
// The header is also host-compilable (plain C++), so tests/ can build it with
// g++ and check the very same functions against oracle/ without a GPU.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define P3D_HD __host__ __device__ __forceinline__
#define P3D_HDM __host__ __device__ __forceinline__
#else
#define P3D_HD static inline
#define P3D_HDM inline
#endif

namespace p3d {

// geometry_utils.cuh:18 -- a double; comparisons against it promote the float side.
#define P3D_KEPS 1e-8

struct f2 {
  float x, y;
};
struct f3 {
  float x, y, z;
};

P3D_HD f2 mk2(float x, float y) {
  f2 r;
  r.x = x;
  r.y = y;
  return r;
}
P3D_HD f3 mk3(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}

// Division for GRADIENT arithmetic only (tolerance-gated, rtol 2e-3 in the reference's own tests):
// on the device one v_rcp_f32 (1 ulp) + one multiply instead of the 11-instruction IEEE sequence.
// Everything that decides pix_to_face / zbuf / bary / dists keeps IEEE division.
template <bool FAST>
P3D_HD float qdiv(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (FAST) return a * __builtin_amdgcn_rcpf(b);
#endif
  return a / b;
}

P3D_HD float min3(float a, float b, float c) { return fminf(a, fminf(b, c)); }
P3D_HD float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
// __saturatef semantics: clamp to [+0, 1], NaN -> +0.
P3D_HD float sat01(float t) { return fminf(fmaxf(t, 0.0f), 1.0f); }

// ---------------------------------------------------------------------------
// Pixel <-> NDC (rasterization_utils.cuh:16-42).  The short image side spans
// [-1, 1]; the long side is scaled by the aspect ratio.  Pixel centres.
// ---------------------------------------------------------------------------
P3D_HD float ndc_range(int S1, int S2) {
  float range = 2.0f;
  if (S1 > S2) {
    range = ((float)S1 * range) / (float)S2;
  }
  return range;
}

P3D_HD float pix_to_ndc(int i, int S1, int S2) {
  const float range = ndc_range(S1, S2);
  const float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S1;
}

// Half a pixel in NDC units along the S1 axis (rasterize_coarse.cu:99-105).
P3D_HD float half_pixel(int S1, int S2) { return (ndc_range(S1, S2) / 2.0f) / (float)S1; }

// Lower/upper NDC edge of bin b along an axis (rasterize_coarse.cu:148-160).
P3D_HD float bin_lo(int b, int bin_size, int S1, int S2) {
  return pix_to_ndc(b * bin_size, S1, S2) - half_pixel(S1, S2);
}
P3D_HD float bin_hi(int b, int bin_size, int S1, int S2) {
  return pix_to_ndc((b + 1) * bin_size - 1, S1, S2) + half_pixel(S1, S2);
}

// ---------------------------------------------------------------------------
// 2D edge function and barycentrics (geometry_utils.cuh:37-86).
// ---------------------------------------------------------------------------
P3D_HD float edge_fn(f2 p, f2 a, f2 b) { return (p.x - a.x) * (b.y - a.y) - (p.y - a.y) * (b.x - a.x); }

// area = float(double(edge) + 1e-8): one double add, then narrowing.
P3D_HD float bary_area(f2 v0, f2 v1, f2 v2) { return (float)((double)edge_fn(v2, v0, v1) + P3D_KEPS); }

template <bool FAST = false>
P3D_HD f3 bary_coords(f2 p, f2 v0, f2 v1, f2 v2) {
  const float area = bary_area(v0, v1, v2);
  const float w0 = qdiv<FAST>(edge_fn(p, v1, v2), area);
  const float w1 = qdiv<FAST>(edge_fn(p, v2, v0), area);
  const float w2 = qdiv<FAST>(edge_fn(p, v0, v1), area);
  return mk3(w0, w1, w2);
}

// Perspective correction (geometry_utils.cuh:172-185).  Product order is the
// CUDA one: (b.x*z1)*z2, (z0*b.y)*z2, (z0*z1)*b.z.
template <bool FAST = false>
P3D_HD f3 bary_perspective(f3 b, float z0, float z1, float z2) {
  const float t0 = b.x * z1 * z2;
  const float t1 = z0 * b.y * z2;
  const float t2 = z0 * z1 * b.z;
  const float denom = fmaxf(t0 + t1 + t2, (float)P3D_KEPS);
  return mk3(qdiv<FAST>(t0, denom), qdiv<FAST>(t1, denom), qdiv<FAST>(t2, denom));
}

// Clip to >= 0 and renormalise (geometry_utils.cuh:246-259).
template <bool FAST = false>
P3D_HD f3 bary_clip(f3 b) {
  float w0 = b.x > 0.0f ? b.x : 0.0f;
  float w1 = b.y > 0.0f ? b.y : 0.0f;
  float w2 = b.z > 0.0f ? b.z : 0.0f;
  float s = w0 + w1 + w2;
  s = fmaxf(s, 1e-5f);
  return mk3(qdiv<FAST>(w0, s), qdiv<FAST>(w1, s), qdiv<FAST>(w2, s));
}

// Squared distance from p to segment (a, b) (geometry_utils.cuh:340-352).  (FAST: reciprocal estimate and float predicate; the
// backward forms its ranking distances inline, see face_sample_bwd.)
template <bool FAST = false>
P3D_HD float seg_dist2(f2 p, f2 a, f2 b) {
  const float bax = b.x - a.x;
  const float bay = b.y - a.y;
  const float l2 = bax * bax + bay * bay;
  float t = qdiv<FAST>(bax * (p.x - a.x) + bay * (p.y - a.y), l2);
  // degenerate edge (geometry_utils.cuh:345): distance to b.  Both results are formed and one is selected --
  // a branch here costs more than the five operations it would skip, and it sits in the hottest loop.
  const float ex = p.x - b.x;
  const float ey = p.y - b.y;
  const float d_point = ex * ex + ey * ey;
  t = sat01(t);
  const float dx = (a.x + t * bax) - p.x;
  const float dy = (a.y + t * bay) - p.y;
  const float d_seg = dx * dx + dy * dy;
  // FAST (backward): the same predicate on floats -- 1e-8 lies between the floats 9.99999994e-9 (= 1e-8f, the nearer one) and
  // 1.00000008e-8, so `l2 <= 1e-8` in double and `l2 <= 1e-8f` in float select the same floats -- without the v_cvt_f64_f32 +
  // v_cmp_f64 per edge and sample
  if (FAST) return (l2 <= 1e-8f) ? d_point : d_seg;
  return ((double)l2 <= P3D_KEPS) ? d_point : d_seg;
}

// Squared distance to the triangle boundary (geometry_utils.cuh:397-408).
P3D_HD float tri_dist2(f2 p, f2 v0, f2 v1, f2 v2) {
  const float e01 = seg_dist2(p, v0, v1);
  const float e02 = seg_dist2(p, v0, v2);
  const float e12 = seg_dist2(p, v1, v2);
  return fminf(fminf(e01, e02), e12);
}

// ---------------------------------------------------------------------------
// Pixel-independent face setup (what CheckPixelInsideFace recomputes per pixel,
// rasterize_meshes.cu:138-150).  `reject` collects every test that does not
// depend on the pixel: zmax < 0, culled back face, zero area, zmin < 1e-8.
// ---------------------------------------------------------------------------
struct FaceSetup {
  float xlo, xhi, ylo, yhi;  // bbox expanded by sqrt(blur_radius)
  bool reject;
};

P3D_HD FaceSetup face_setup(f3 v0, f3 v1, f3 v2, float sqrt_blur, bool cull_backfaces) {
  FaceSetup s;
  s.xlo = min3(v0.x, v1.x, v2.x) - sqrt_blur;
  s.ylo = min3(v0.y, v1.y, v2.y) - sqrt_blur;
  s.xhi = max3(v0.x, v1.x, v2.x) + sqrt_blur;
  s.yhi = max3(v0.y, v1.y, v2.y) + sqrt_blur;
  const float zmin = min3(v0.z, v1.z, v2.z);
  const float zmax = max3(v0.z, v1.z, v2.z);
  const float area = edge_fn(mk2(v0.x, v0.y), mk2(v1.x, v1.y), mk2(v2.x, v2.y));
  const bool back_face = area < 0.0f;
  const bool zero_area = ((double)area <= P3D_KEPS) && ((double)area >= -1.0 * P3D_KEPS);
  const bool z_invalid = (double)zmin < P3D_KEPS;
  s.reject = (zmax < 0.0f) || (cull_backfaces && back_face) || zero_area || z_invalid;
  return s;
}

// Per-pixel bbox reject (rasterize_meshes.cu:94-97), strict comparisons.
P3D_HD bool outside_box(const FaceSetup& s, f2 p) { return p.x > s.xhi || p.x < s.xlo || p.y > s.yhi || p.y < s.ylo; }

// Result of testing one pixel against one face.
struct FaceHit {
  float z;
  float dist;  // signed: negative inside
  f3 bary;     // clipped when clip_barycentric_coords
};

// The pixel-dependent part of CheckPixelInsideFace (rasterize_meshes.cu:152-177).
// Returns false when the face does not contribute to this pixel.
P3D_HD bool face_hit(f3 v0, f3 v1, f3 v2, f2 p, float blur_radius, bool perspective_correct, bool clip_bary,
                     FaceHit* out) {
  const f2 a = mk2(v0.x, v0.y);
  const f2 b = mk2(v1.x, v1.y);
  const f2 c = mk2(v2.x, v2.y);
  const f3 bw = bary_coords(p, a, b, c);
  const f3 bp = perspective_correct ? bary_perspective(bw, v0.z, v1.z, v2.z) : bw;
  const f3 bc = clip_bary ? bary_clip(bp) : bp;
  const float pz = bc.x * v0.z + bc.y * v1.z + bc.z * v2.z;
  const float dist = tri_dist2(p, a, b, c);
  const bool inside = (bp.x > 0.0f) & (bp.y > 0.0f) & (bp.z > 0.0f);
  // branch-free accept test (rasterize_meshes.cu:162-177): behind the camera, or outside the face and beyond the
  // blur radius -> no contribution
  const bool hit = !(pz < 0.0f) & (inside | !(dist >= blur_radius));
  out->z = pz;
  out->dist = inside ? -dist : dist;
  out->bary = bc;
  return hit;
}

// ---------------------------------------------------------------------------
// The same per-(pixel, face) test with SHARED RECIPROCALS -- bit-identical results, fewer instructions.
//
// An IEEE float division costs 11 instructions on gfx950 (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup),
// twelve of them per (pixel, face) = 132 of ~300.  Nine share a denominator with two others (the face area, the
// perspective denominator, the clip sum), the other three divide by per-FACE squared edge lengths.  Replacing
// `n / d` by a multiplication with a reciprocal is not exact in float arithmetic -- but it is in DOUBLE:
//
//     (float)((double)n * rd)  ==  n / d   (the correctly rounded float quotient)   whenever |rd * d - 1| <= 2^-52
//
// Proof.  Scale n, d to integers in [2^23, 2^24); q = n/d lies in (1/2, 2).  The rounding boundaries of the float
// grid around q are M = j / 2^25 (j odd; 2^24 for q >= 1).  |q - M| = |n 2^25 - j d| / (d 2^25) >= 1 / (d 2^25):
// the numerator is a non-zero integer (n 2^25 = j d would need 2^25 | d).  So q is at least 2^-49 (relative) away
// from every boundary, while (double)n * rd is within 2^-52 (rd) + 2^-53 (the product's rounding) of q: rounding it
// to float lands on the same float as rounding q.  n = 0, d = inf and NaNs propagate as in IEEE division; for
// results below 2^-126 (float denormals) the argument does not hold and the last denormal bit may differ.
//
// rd: a hardware reciprocal estimate refined in double (recip_newton).  Per pixel the seed is v_rcp_f32 (1 ulp =
// 2^-23 -> 2^-46 -> below 2^-52), which needs 2^-126 <= d <= 2^126; that holds for the per-pixel denominators of every
// face whose coordinates are of ordinary magnitude (`FaceRec::wide` false).  For the other faces, and for the per-face
// reciprocals (computed once per face), the seed is v_rcp_f64, valid over the whole float range.  On the host 1.0 / d.
// All of them give the same float quotients, by the argument above.
// ---------------------------------------------------------------------------
// One cubic step instead of two Newton steps (round 4: three v_fma_f64 instead of four, twice per (pixel, face)): with
// e = 1 - d r the true reciprocal is r (1 + e + e^2 + e^3 + ...); r (1 + e + e^2) is off by e^3 <= 2^-69 for a seed of 23
// bits (v_rcp_f32: 1 ulp; v_rcp_f64 on gfx950 measures 2^-24.4, profiles/microbench/rcp_accuracy_mi355x.txt).  Roundings:
// e and t = e + e^2 are of magnitude 2^-23, their rounding errors (2^-53 relative to them) vanish; the last fma rounds
// once, 2^-53 relative -- together below the 2^-52 the argument above needs.
P3D_HD double recip_newton(double dd, double r) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double e = __builtin_fma(-dd, r, 1.0);
  const double t = __builtin_fma(e, e, e);
  r = __builtin_fma(r, t, r);
#endif
  return r;
}

P3D_HD double recip_for_div(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return recip_newton((double)d, (double)__builtin_amdgcn_rcpf(d));
#else
  return 1.0 / (double)d;
#endif
}

P3D_HD double recip_for_div_wide(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return recip_newton((double)d, __builtin_amdgcn_rcp((double)d));
#else
  return 1.0 / (double)d;
#endif
}

P3D_HD float exact_div(float n, double rd) { return (float)((double)n * rd); }

// Per-face record for `face_hit_rec`: vertices + the per-face reciprocals.  A degenerate edge (squared length <= 1e-8,
// geometry_utils.cuh:345: the distance to the edge is the distance to its end point) is marked by a NEGATIVE reciprocal.
struct FaceRec {
  f3 v0, v1, v2;
  double rd_area;                // 1 / bary_area(v0, v1, v2)
  double rd_l01, rd_l02, rd_l12;  // 1 / |v1 - v0|^2 etc., < 0 when the edge is degenerate
  bool wide;                      // magnitudes beyond the ordinary: per-pixel reciprocals need the f64 seed
  // The edge vectors the three edge functions and the three segment distances subtract out of the vertices for every pixel
  // (round 6: six v_sub per evaluation, now once per face).  d01 = v1 - v0, d12 = v2 - v1, d20 = v0 - v2: the very differences
  // of edge_fn(p, a, b) / edge_fn(p, b, c) / edge_fn(p, c, a); seg_dist2(p, v0, v2) wants v2 - v0 = -d20, and a negation is
  // exact and commutes with every rounding after it: the same bits.
  f2 d01, d12, d20;
};

// edge_fn(p, a, b) with d = b - a formed beforehand
P3D_HD float edge_fn_d(f2 p, f2 a, f2 d) { return (p.x - a.x) * d.y - (p.y - a.y) * d.x; }

P3D_HD double edge_recip(f2 a, f2 b) {
  const float bax = b.x - a.x;
  const float bay = b.y - a.y;
  const float l2 = bax * bax + bay * bay;
  return ((double)l2 <= P3D_KEPS) ? -1.0 : recip_for_div_wide(l2);
}

P3D_HD void face_rec_make(f3 v0, f3 v1, f3 v2, FaceRec* r) {
  const f2 a = mk2(v0.x, v0.y), b = mk2(v1.x, v1.y), c = mk2(v2.x, v2.y);
  const float area = bary_area(a, b, c);
  r->v0 = v0;
  r->v1 = v1;
  r->v2 = v2;
  r->rd_area = recip_for_div_wide(area);
  r->rd_l01 = edge_recip(a, b);
  r->rd_l02 = edge_recip(a, c);
  r->rd_l12 = edge_recip(b, c);
  r->d01 = mk2(b.x - a.x, b.y - a.y);
  r->d12 = mk2(c.x - b.x, c.y - b.y);
  r->d20 = mk2(a.x - c.x, a.y - c.y);
  const float big = fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v1.x), fabsf(v1.y))),
                          fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), max3(fabsf(v0.z), fabsf(v1.z), fabsf(v2.z))));
  // Ordinary: |x|, |y|, z <= 1024 and |area| >= 1e-9 bound the perspective denominator by 1e22 and the clip sum by
  // 3e30 (both are >= 1e-8 by construction), inside v_rcp_f32's range.  !(x <= y) also catches NaN.
  r->wide = !((big <= 1024.0f) & (fabsf(area) >= 1e-9f));
}

// seg_dist2 with the edge's reciprocal and its vector ba = b - a: same value as seg_dist2(p, a, b)
// PROPER: the caller knows the edge is not degenerate (rd_l2 > 0): the end-point distance and the select are not formed
// (five instructions per edge in the fine kernel's hottest block; round 6).
template <bool PROPER = false>
P3D_HD float seg_dist2_rec(f2 p, f2 a, f2 b, f2 ba, double rd_l2) {
  const float bax = ba.x;
  const float bay = ba.y;
  float t = exact_div(bax * (p.x - a.x) + bay * (p.y - a.y), rd_l2);
  t = sat01(t);
  const float dx = (a.x + t * bax) - p.x;
  const float dy = (a.y + t * bay) - p.y;
  const float d_seg = dx * dx + dy * dy;
  if (PROPER) return d_seg;
  const float ex = p.x - b.x;
  const float ey = p.y - b.y;
  const float d_point = ex * ex + ey * ey;
  return (rd_l2 < 0.0) ? d_point : d_seg;
}

// a face with a degenerate edge (FaceRec::rd_l.. < 0): the fine kernel's fast nest hands such faces to its general nest
P3D_HD bool face_rec_degenerate(const FaceRec& r) { return (r.rd_l01 < 0.0) | (r.rd_l02 < 0.0) | (r.rd_l12 < 0.0); }

// face_hit on a FaceRec: identical outputs, 12 divisions -> 12 (cvt, mul_f64, cvt) + 2 reciprocals.  In two halves so that
// the fine kernel can stop after the depth: a sample whose depth cannot enter a full queue needs no distance
// (half of all evaluations at the bench workload, profiles/r03/probe_counts.txt).
//
// First half: barycentrics (bp: before the clip, decides `inside`; out->bary: what is stored) and the depth.
P3D_HD f3 face_depth_rec(const FaceRec& r, f2 p, bool perspective_correct, bool clip_bary, FaceHit* out) {
  const f2 a = mk2(r.v0.x, r.v0.y);
  const f2 b = mk2(r.v1.x, r.v1.y);
  const f2 c = mk2(r.v2.x, r.v2.y);
  const f3 bw = mk3(exact_div(edge_fn_d(p, b, r.d12), r.rd_area), exact_div(edge_fn_d(p, c, r.d20), r.rd_area),
                    exact_div(edge_fn_d(p, a, r.d01), r.rd_area));
  f3 bp = bw;
  if (perspective_correct) {
    const float t0 = bw.x * r.v1.z * r.v2.z;
    const float t1 = r.v0.z * bw.y * r.v2.z;
    const float t2 = r.v0.z * r.v1.z * bw.z;
    const float denom = fmaxf(t0 + t1 + t2, (float)P3D_KEPS);
    const double rd = r.wide ? recip_for_div_wide(denom) : recip_for_div(denom);
    bp = mk3(exact_div(t0, rd), exact_div(t1, rd), exact_div(t2, rd));
  }
  f3 bc = bp;
  if (clip_bary) {
    const float w0 = bp.x > 0.0f ? bp.x : 0.0f;
    const float w1 = bp.y > 0.0f ? bp.y : 0.0f;
    const float w2 = bp.z > 0.0f ? bp.z : 0.0f;
    float s = w0 + w1 + w2;
    s = fmaxf(s, 1e-5f);
    const double rd = r.wide ? recip_for_div_wide(s) : recip_for_div(s);
    bc = mk3(exact_div(w0, rd), exact_div(w1, rd), exact_div(w2, rd));
  }
  out->z = bc.x * r.v0.z + bc.y * r.v1.z + bc.z * r.v2.z;
  out->bary = bc;
  return bp;
}

// Second half: the signed squared distance and the hit test (bp, out->z from face_depth_rec).  PROPER: no edge of the face is
// degenerate (face_rec_degenerate is false).
template <bool PROPER = false>
P3D_HD bool face_dist_rec(const FaceRec& r, f2 p, float blur_radius, f3 bp, FaceHit* out) {
  const f2 a = mk2(r.v0.x, r.v0.y);
  const f2 b = mk2(r.v1.x, r.v1.y);
  const f2 c = mk2(r.v2.x, r.v2.y);
  const float e01 = seg_dist2_rec<PROPER>(p, a, b, r.d01, r.rd_l01);
  const float e02 = seg_dist2_rec<PROPER>(p, a, c, mk2(-r.d20.x, -r.d20.y), r.rd_l02);
  const float e12 = seg_dist2_rec<PROPER>(p, b, c, r.d12, r.rd_l12);
  const float dist = fminf(fminf(e01, e02), e12);
  const bool inside = (bp.x > 0.0f) & (bp.y > 0.0f) & (bp.z > 0.0f);
  out->dist = inside ? -dist : dist;
  return !(out->z < 0.0f) & (inside | !(dist >= blur_radius));
}

P3D_HD bool face_hit_rec(const FaceRec& r, f2 p, float blur_radius, bool perspective_correct, bool clip_bary,
                         FaceHit* out) {
  const f3 bp = face_depth_rec(r, p, perspective_correct, clip_bary, out);
  return face_dist_rec(r, p, blur_radius, bp, out);
}

// ---------------------------------------------------------------------------
// Pixel masks of a bounding box (fine rasterizer).  The reference tests every pixel against the blur-expanded box of
// every candidate face (rasterize_meshes.cu:94-97: outside if px > xhi || px < xlo || py > yhi || py < ylo, strict).
// Pixel centres are strictly monotone in the pixel index, so along one axis the pixels inside are the index range
// [#(centre < lo), n - 1 - #(centre > hi)] -- two counts per axis, taken once per (face, tile) instead of four compares per
// (face, pixel).  A NaN edge compares false both ways: every pixel is inside, as in the reference's expression.
// ---------------------------------------------------------------------------
// bits [below, 15 - above] of a 16-pixel axis (below = centres under the low edge, above = centres over the high edge)
P3D_HD unsigned range_mask16(int below, int above) { return (0xffffu >> above) & (0xffffu << below) & 0xffffu; }

// An 8-column mask and an 8-row mask as the 64-bit mask of an 8x8 block, bit 8 * row + column, in two halves: the column
// byte replicated into the bytes of the set rows.  A 4-bit row group times 0x00204081 puts bit i at position 8 i (the
// shifted copies x, x << 7, x << 14, x << 21 do not overlap), times the column byte (< 256) fills those bytes.
P3D_HD void block_mask_8x8(unsigned cols8, unsigned rows8, unsigned* lo, unsigned* hi) {
  *lo = cols8 * (((rows8 & 15u) * 0x00204081u) & 0x01010101u);
  *hi = cols8 * ((((rows8 >> 4) & 15u) * 0x00204081u) & 0x01010101u);
}

// ---------------------------------------------------------------------------
// Conservative rectangle-vs-face reject for the fine rasterizers' culling stages: true only if NO pixel centre
// in [x0, x1] x [y0, y1] can be hit by the face, i.e. every point of the rectangle is outside the triangle AND
// farther than sqrt(blur) from it.  Two sufficient conditions: (1) the rectangle is farther than r from the
// triangle's bounding box (Euclidean: rounds the corners of the blur-expanded box); (2) the rectangle lies beyond
// one edge LINE by more than r.  `m` absorbs the float error of these tests and of the reference's own inside /
// distance arithmetic (both ~1e-6 for coordinates of a few units); faces with larger coordinates are never
// rejected here.  A reject only skips work; results never depend on it.
// ---------------------------------------------------------------------------
P3D_HD bool rect_cannot_hit(f2 a, f2 b, f2 c, float x0, float x1, float y0, float y1, float r) {
  const float big = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(b.x), fabsf(b.y))), fmaxf(fabsf(c.x), fabsf(c.y)));
  const float rm = r + (2e-5f + 1e-4f * r);
  const float rm2 = rm * rm;
  // (1) distance to the bounding box
  const float gx = fmaxf(fmaxf(min3(a.x, b.x, c.x) - x1, x0 - max3(a.x, b.x, c.x)), 0.0f);
  const float gy = fmaxf(fmaxf(min3(a.y, b.y, c.y) - y1, y0 - max3(a.y, b.y, c.y)), 0.0f);
  const bool miss_box = gx * gx + gy * gy > rm2;
  bool miss = false;
  // (2) beyond an edge line.  s * edge_fn(p, u, v) > 0 on the inner side of edge (u, v), s = sign of the area.
  const float area = edge_fn(c, a, b);
  const float s = area < 0.0f ? -1.0f : 1.0f;
#define P3D_EDGE_MISS(u, v)                                                                      \
  {                                                                                              \
    const float A = s * (v.y - u.y), B = -s * (v.x - u.x);                                       \
    const float f = A * ((A > 0.0f ? x1 : x0) - u.x) + B * ((B > 0.0f ? y1 : y0) - u.y);         \
    miss = miss | ((f < 0.0f) & (f * f > rm2 * (A * A + B * B)));                                \
  }
  P3D_EDGE_MISS(b, c)
  P3D_EDGE_MISS(c, a)
  P3D_EDGE_MISS(a, b)
#undef P3D_EDGE_MISS
  return (big <= 8.0f) & (miss_box | (miss & (fabsf(area) >= 1e-7f)));
}

// ---------------------------------------------------------------------------
// Backward pieces (geometry_utils.cuh:54-64, 101-161, 200-228, 273-329,
// 365-385, 421-462).  Gradients are tolerance-gated, the expression order is
// kept anyway; pow(x, 2.0f) is written x*x.
// ---------------------------------------------------------------------------
struct EdgeGrad {
  f2 dp, da, db;
};

P3D_HD EdgeGrad edge_fn_bwd(f2 p, f2 a, f2 b, float g) {
#if defined(__clang__)
#pragma clang fp contract(fast)  // gradient-only arithmetic (tolerance-gated): let mul+add fuse into FMA
#endif
  EdgeGrad r;
  r.dp = mk2(g * (b.y - a.y), g * (a.x - b.x));
  r.da = mk2(g * (p.y - b.y), g * (b.x - p.x));
  r.db = mk2(g * (a.y - p.y), g * (p.x - a.x));
  return r;
}

struct TriGrad {
  f2 d0, d1, d2;
};

P3D_HD f2 add2(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }

// e: the three edge functions of p, inv_area = 1 / bary_area, inv_area2 = 1 / area^2 -- formed by the caller (face_sample_bwd), once
P3D_HD TriGrad bary_coords_bwd(f2 p, f2 v0, f2 v1, f2 v2, f3 g, f3 e, float inv_area, float inv_area2) {
#if defined(__clang__)
#pragma clang fp contract(fast)  // gradient-only arithmetic (tolerance-gated): let mul+add fuse into FMA
#endif
  const float e0 = e.x, e1 = e.y, e2 = e.z;

  // w_k = e_k / area with e0 = edge_fn(p, v1, v2), e1 = edge_fn(p, v2, v0), e2 = edge_fn(p, v0, v1), area = edge_fn(v2, v0, v1)
  // (geometry_utils.cuh:101-161).  Numerators: one edge_fn_bwd each.  Denominator: the reference runs edge_fn_bwd(v2, v0, v1, .)
  // three times with upstream g_k * (-e_k / area^2); it is linear in the upstream, so the three are summed first.
  const EdgeGrad n0 = edge_fn_bwd(p, v1, v2, g.x * inv_area);
  const EdgeGrad n1 = edge_fn_bwd(p, v2, v0, g.y * inv_area);
  const EdgeGrad n2 = edge_fn_bwd(p, v0, v1, g.z * inv_area);
  const EdgeGrad ar = edge_fn_bwd(v2, v0, v1, -(g.x * e0 + g.y * e1 + g.z * e2) * inv_area2);

  TriGrad r;
  r.d0 = add2(add2(n1.db, n2.da), ar.da);
  r.d1 = add2(add2(n0.da, n2.db), ar.db);
  r.d2 = add2(add2(n0.db, n1.da), ar.dp);
  return r;
}

P3D_HD f3 bary_clip_bwd(f3 b, f3 g) {
#if defined(__clang__)
#pragma clang fp contract(fast)  // gradient-only arithmetic (tolerance-gated): let mul+add fuse into FMA
#endif
  const float w0 = b.x > 0.0f ? b.x : 0.0f;
  const float w1 = b.y > 0.0f ? b.y : 0.0f;
  const float w2 = b.z > 0.0f ? b.z : 0.0f;
  float s = w0 + w1 + w2;
  float live = 1.0f;
  if ((double)s < 1e-5) {  // reference compares the float sum with a double literal
    live = 0.0f;
    s = 1e-5f;
  }
  const float m0 = b.x < 0.0f ? 0.0f : 1.0f;
  const float m1 = b.y < 0.0f ? 0.0f : 1.0f;
  const float m2 = b.z < 0.0f ? 0.0f : 1.0f;
  const float inv = qdiv<true>(1.0f, s);
  const float inv_s2 = inv * inv * live;
  const float q0 = -w0 * inv_s2;
  const float q1 = -w1 * inv_s2;
  const float q2 = -w2 * inv_s2;
  // d clip_k / d w_k = 1 / s - w_k / s^2.  The reference forms it as that DIFFERENCE of two correctly rounded quotients
  // (geometry_utils.cuh:313-327), which is exactly 0 when w_k is the only coordinate left after clipping (the clipped
  // barycentrics are then the constant (0, 0, 1)): 1 / s and w_k / s^2 round to the same float.  With a reciprocal
  // estimate the two terms no longer cancel, and what is left (1e-7 / s) meets the 1e16 of a perspective denominator
  // clamped at 1e-8 in the blur band of faces seen nearly edge-on: gradients of 1e15 where the reference has 0 (found in
  // round 3 on 0.4 % of the faces of the bench batch, once the comparison was made per face).  Written as the sum of the
  // OTHER two coordinates over s^2 the term is exact in that case and carries no cancellation in any other.
  const float c0 = live != 0.0f ? (w1 + w2) * inv_s2 : inv;
  const float c1 = live != 0.0f ? (w0 + w2) * inv_s2 : inv;
  const float c2 = live != 0.0f ? (w0 + w1) * inv_s2 : inv;
  return mk3(m0 * (g.x * c0 + g.y * q1 + g.z * q2), m1 * (g.y * c1 + g.x * q0 + g.z * q2), m2 * (g.z * c2 + g.x * q0 + g.y * q1));
}

// Nine partials of one (pixel, k) sample wrt its face's vertices
// (rasterize_meshes.cu:486-561).  The CUDA variant feeds the *pre-perspective*
// barycentrics to the clip backward (rasterize_meshes.cu:528); the CPU variant
// uses the post-perspective ones (rasterize_meshes_cpu.cpp:499).  We follow
// CUDA; `clip_bwd_on_corrected` selects the CPU behaviour (used only to pin the
// oracle against the reference's CPU build).
struct FaceGrad {
  float g[9];
};

// What a sample's gradient needs of its FACE alone (round 6): the reciprocal of the barycentric area and of the three squared edge
// lengths (negative: the edge is degenerate, squared length <= 1e-8, geometry_utils.cuh:345).  The backward formed them per SAMPLE
// -- five v_rcp_f32 (quarter rate), the area's double-precision epsilon add, three compares: a tenth of a step's vector time at the
// bench workload, where a face is sampled ~230 times.  The face gather of the forward writes one 16-byte record per face
// (p3d_gather_face_verts_pre), the backward gathers it beside the 36 bytes of vertices.  Same instructions on the same operands as
// the per-sample form (v_rcp_f32 is deterministic); 1 / area^2 becomes (1 / area)^2 (gradient arithmetic, tolerance-gated).
struct BwdFacePre {
  float inv_area, inv_l01, inv_l02, inv_l12;
};

P3D_HD BwdFacePre bwd_face_pre_make(f3 v0, f3 v1, f3 v2) {
  const f2 a = mk2(v0.x, v0.y), b = mk2(v1.x, v1.y), c = mk2(v2.x, v2.y);
  BwdFacePre r;
  r.inv_area = qdiv<true>(1.0f, bary_area(a, b, c));
  const f2 es[3] = {a, a, b}, ee[3] = {b, c, c};
  float inv[3];
  for (int i = 0; i < 3; ++i) {
    const float bax = ee[i].x - es[i].x, bay = ee[i].y - es[i].y;
    const float l2 = bax * bax + bay * bay;
    inv[i] = (l2 <= 1e-8f) ? -1.0f : qdiv<true>(1.0f, l2);
  }
  r.inv_l01 = inv[0];
  r.inv_l02 = inv[1];
  r.inv_l12 = inv[2];
  return r;
}

template <bool PRE = false>
P3D_HD FaceGrad face_sample_bwd(f3 v0, f3 v1, f3 v2, f2 p, float g_zbuf, f3 g_bary, float g_dist,
                                bool perspective_correct, bool clip_bary, bool clip_bwd_on_corrected,
                                BwdFacePre pre = BwdFacePre()) {
#if defined(__clang__)
#pragma clang fp contract(fast)  // gradient-only arithmetic (tolerance-gated): let mul+add fuse into FMA
#endif
  // One fused function (round 6) instead of a forward recompute followed by the reference's five backward routines called one
  // after the other: each shared quantity -- the perspective numerators and their denominator, a segment's parameter and
  // closest point -- is formed ONCE.  The backward kernel follows its instruction count, and the separate routines rebuilt these
  // with other roundings (fused or not), which the compiler cannot merge.  Formulas: geometry_utils.cuh:54-64 (edge function),
  // 101-161 (barycentrics), 200-228 (perspective), 273-329 (clip), 365-385 / 421-462 (point-segment / point-triangle distance).
  const f2 a = mk2(v0.x, v0.y);
  const f2 b = mk2(v1.x, v1.y);
  const f2 c = mk2(v2.x, v2.y);
  const float z0 = v0.z, z1 = v1.z, z2 = v2.z;
  // ---- forward recompute: only signs of bw / bp decide anything here (inside test, clip masks), and the sign of a quotient does
  // not depend on how the division rounds
  const f3 e = mk3(edge_fn(p, b, c), edge_fn(p, c, a), edge_fn(p, a, b));
  float inv_area, inv_area2;
  if (PRE) {
    inv_area = pre.inv_area;
    inv_area2 = inv_area * inv_area;
  } else {
    const float area = bary_area(a, b, c);
    inv_area = qdiv<true>(1.0f, area);
    inv_area2 = qdiv<true>(1.0f, area * area);
  }
  const f3 bw = mk3(e.x * inv_area, e.y * inv_area, e.z * inv_area);
  f3 bp = bw;
  float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, inv_denom = 0.0f;
  if (perspective_correct) {
    t0 = bw.x * z1 * z2;
    t1 = z0 * bw.y * z2;
    t2 = z0 * z1 * bw.z;
    const float denom = fmaxf(t0 + t1 + t2, (float)P3D_KEPS);
    inv_denom = qdiv<true>(1.0f, denom);
    bp = mk3(t0 * inv_denom, t1 * inv_denom, t2 * inv_denom);
  }
  const f3 bc = clip_bary ? bary_clip<true>(bp) : bp;
  const bool inside = (bp.x > 0.0f) & (bp.y > 0.0f) & (bp.z > 0.0f);
  const float gd = inside ? -g_dist : g_dist;

  // ---- squared distance to the boundary: only the closest edge gets a gradient, ties resolved e01, e02, e12
  // (geometry_utils.cuh:441-459).  Per edge (s, e): t = clamp(<e - s, p - s> / |e - s|^2), q = s + t (e - s) - p; the distances only
  // RANK the edges (reciprocal estimate, float predicate for the degenerate edge: 1e-8 lies between the floats 9.99999994e-9 and
  // 1.00000008e-8); two edges that tie to within an ulp may swap: they tie where the nearest point is their common vertex, and
  // there both give that vertex the same gradient and the other end point none (t saturates at 0 or 1).
  float et[3], ex[3], ey[3], ed[3];
  const f2 es[3] = {a, a, b}, ee[3] = {b, c, c};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float bax = ee[i].x - es[i].x, bay = ee[i].y - es[i].y;
    const float dot = bax * (p.x - es[i].x) + bay * (p.y - es[i].y);
    float t;
    bool degenerate;
    if (PRE) {
      const float inv = i == 0 ? pre.inv_l01 : (i == 1 ? pre.inv_l02 : pre.inv_l12);
      degenerate = inv < 0.0f;
      t = sat01(dot * inv);
    } else {
      const float l2 = bax * bax + bay * bay;
      degenerate = l2 <= 1e-8f;
      t = sat01(qdiv<true>(dot, l2));
    }
    const float qx = (es[i].x + t * bax) - p.x, qy = (es[i].y + t * bay) - p.y;
    const float px = p.x - ee[i].x, py = p.y - ee[i].y;
    et[i] = t;
    ex[i] = qx;
    ey[i] = qy;
    ed[i] = degenerate ? px * px + py * py : qx * qx + qy * qy;
  }
  // which edge is closest; 3 = none (NaN distances).  The three candidate branches of the reference are folded into ONE
  // evaluation on selected values: lanes of a wave pick different edges, and divergent branches would run it three times.
  const int sel = ((ed[0] <= ed[1]) & (ed[0] <= ed[2])) ? 0 : (((ed[1] <= ed[0]) & (ed[1] <= ed[2])) ? 1 : (((ed[2] <= ed[0]) & (ed[2] <= ed[1])) ? 2 : 3));
  const float st = sel == 0 ? et[0] : (sel == 1 ? et[1] : et[2]);
  const float sx = sel == 0 ? ex[0] : (sel == 1 ? ex[1] : ex[2]);
  const float sy = sel == 0 ? ey[0] : (sel == 1 ? ey[1] : ey[2]);
  const float g2 = sel == 3 ? 0.0f : gd + gd;  // d |q|^2 / d q = 2 q
  const float ga = g2 * (1.0f - st), gb_ = g2 * st;  // q = (1 - t) s + t e - p: weights of the segment's start and end point
  const f2 da = mk2(ga * sx, ga * sy), db = mk2(gb_ * sx, gb_ * sy);
  const f2 zero = mk2(0.0f, 0.0f);
  const f2 dd0 = sel <= 1 ? da : zero;                        // v0 starts e01 and e02
  const f2 dd1 = sel == 0 ? db : (sel == 2 ? da : zero);      // v1 ends e01, starts e12
  const f2 dd2 = (sel == 1 || sel == 2) ? db : zero;          // v2 ends e02 and e12

  // ---- barycentrics: zbuf = sum bc_i z_i, so its upstream joins the barycentrics' (rasterize_meshes.cu:486-561)
  f3 gb = mk3(g_bary.x + g_zbuf * z0, g_bary.y + g_zbuf * z1, g_bary.z + g_zbuf * z2);
  if (clip_bary) gb = bary_clip_bwd(clip_bwd_on_corrected ? bp : bw, gb);
  float dz0 = 0.0f, dz1 = 0.0f, dz2 = 0.0f;
  if (perspective_correct) {
    // geometry_utils.cuh:200-228 on the numerators and the reciprocal formed above
    const float gd_top = -t0 * gb.x - t1 * gb.y - t2 * gb.z;
    const float gdn = gd_top * (inv_denom * inv_denom);
    const float g0 = gdn + gb.x * inv_denom;
    const float g1 = gdn + gb.y * inv_denom;
    const float g2p = gdn + gb.z * inv_denom;
    dz0 = g1 * bw.y * z2 + g2p * bw.z * z1;
    dz1 = g0 * bw.x * z2 + g2p * bw.z * z0;
    dz2 = g0 * bw.x * z1 + g1 * bw.y * z0;
    gb = mk3(g0 * z1 * z2, g1 * z0 * z2, g2p * z0 * z1);
  }
  const TriGrad dbw = bary_coords_bwd(p, a, b, c, gb, e, inv_area, inv_area2);

  FaceGrad r;
  r.g[0] = dbw.d0.x + dd0.x;
  r.g[1] = dbw.d0.y + dd0.y;
  r.g[2] = g_zbuf * bc.x + dz0;
  r.g[3] = dbw.d1.x + dd1.x;
  r.g[4] = dbw.d1.y + dd1.y;
  r.g[5] = g_zbuf * bc.y + dz1;
  r.g[6] = dbw.d2.x + dd2.x;
  r.g[7] = dbw.d2.y + dd2.y;
  r.g[8] = g_zbuf * bc.z + dz2;
  return r;
}


}  // namespace p3d
