// shade_sample.h -- the per-sample Phong arithmetic shared by shade.hip (phong_shading as its own operator) and
// soft_phong.hip (shading fused with the softmax blend).  Device code only.
//
// Restates, per (pixel, k) sample, pytorch3d/renderer/mesh/shading.py:17-112 (interpolate position / normal / colour,
// _apply_lighting) and renderer/lighting.py:17-159 (F.normalize eps 1e-6, relu, cos > 0 mask, pow(alpha, shininess)).
#pragma once

#include "p3d_common.h"

namespace p3d {
namespace {

constexpr float kNormEps = 1e-6f;  // lighting.py:76-77,143-144: F.normalize(..., eps=1e-6)

struct ShadeConst {
  float la[3], ld[3], ls[3], lvec[3], ma[3], md[3], ms[3], shin, cam[3];
};

__device__ __forceinline__ ShadeConst load_params(const float* __restrict__ p) {
  ShadeConst c;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    c.la[j] = p[j];
    c.ld[j] = p[3 + j];
    c.ls[j] = p[6 + j];
    c.lvec[j] = p[9 + j];
    c.ma[j] = p[12 + j];
    c.md[j] = p[15 + j];
    c.ms[j] = p[18 + j];
    c.cam[j] = p[22 + j];
  }
  c.shin = p[21];
  return c;
}

__device__ __forceinline__ float dot3(const float (&a)[3], const float (&b)[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// x / max(|x|, eps); returns |x|
__device__ __forceinline__ float normalize3(const float (&x)[3], float (&out)[3]) {
  const float len = sqrtf(dot3(x, x));
  const float d = fmaxf(len, kNormEps);
#pragma unroll
  for (int j = 0; j < 3; ++j) out[j] = x[j] / d;
  return len;
}

// gradient of normalize3 at x (xh = its output, len = |x|) for an upstream dxh.  Gradient arithmetic is
// tolerance-gated (rtol 1e-3): v_rcp_f32 instead of the IEEE division sequence.
__device__ __forceinline__ void normalize3_bwd(const float (&xh)[3], float len, const float (&dxh)[3], float (&dx)[3]) {
  // clamp_min passes the gradient of the norm through only when the norm is the larger operand
  const bool thru = len >= kNormEps;
  const float t = thru ? dot3(xh, dxh) : 0.0f;
  const float inv = __builtin_amdgcn_rcpf(thru ? len : kNormEps);
#pragma unroll
  for (int j = 0; j < 3; ++j) dx[j] = (dxh[j] - xh[j] * t) * inv;
}

// Everything the backward needs again from the lighting of one sample.
struct Lit {
  float nh[3], lh[3], vh[3], R[3];
  float nlen, llen, vlen, cosv, d, alpha, angle, pw;
};

// FAST (backward only): reciprocal-multiply normalisation and exp2(s * log2(alpha)) for the power.
template <bool FAST>
__device__ __forceinline__ float normalize3_any(const float (&x)[3], float (&out)[3]) {
  if constexpr (!FAST) return normalize3(x, out);
  // v_rsq_f32 (1 ulp) instead of the IEEE sqrt + division sequences
  const float d2 = dot3(x, x);
  const float rs = __builtin_amdgcn_rsqf(d2);
  const float len = d2 > 0.0f ? d2 * rs : 0.0f;
  const float inv = len >= kNormEps ? rs : 1.0f / kNormEps;
#pragma unroll
  for (int j = 0; j < 3; ++j) out[j] = x[j] * inv;
  return len;
}

template <bool POINT, bool FAST = false>
__device__ __forceinline__ Lit light_sample(const ShadeConst& c, const float (&P)[3], const float (&Nn)[3]) {
  Lit s;
  s.nlen = normalize3_any<FAST>(Nn, s.nh);
  float L[3], V[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    L[j] = POINT ? c.lvec[j] - P[j] : c.lvec[j];  // lighting.py:283-285 / :196-204
    V[j] = c.cam[j] - P[j];                       // lighting.py:151
  }
  s.llen = normalize3_any<FAST>(L, s.lh);
  s.vlen = normalize3_any<FAST>(V, s.vh);
  s.cosv = dot3(s.nh, s.lh);
  s.angle = fmaxf(s.cosv, 0.0f);  // lighting.py:78
#pragma unroll
  for (int j = 0; j < 3; ++j) s.R[j] = -s.lh[j] + 2.0f * (s.cosv * s.nh[j]);  // lighting.py:153
  s.d = dot3(s.vh, s.R);
  s.alpha = s.cosv > 0.0f ? fmaxf(s.d, 0.0f) : 0.0f;  // lighting.py:147,156
  if constexpr (!FAST) s.pw = powf(s.alpha, c.shin);  // lighting.py:157
  return s;
}


// Colour of one sample with a face (forward, exact arithmetic): rec = the face's (3, D) record, b = barycentrics,
// tex_in = the caller's texel (D == 6; with D == 9 the vertex colours of the record are interpolated instead).
template <int D, bool POINT>
__device__ __forceinline__ void shade_sample_fwd(const ShadeConst& c, const float (&amb)[3], const float (&kd)[3],
                                                 const float (&ks)[3], const float* __restrict__ rec, const float (&b)[3],
                                                 const float (&tex_in)[3], float (&color)[3]) {
  float P[3], Nn[3], tex[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // interp_face_attrs.cu:39-41: (w0*a0 + w1*a1) + w2*a2
    P[j] = (b[0] * rec[j] + b[1] * rec[D + j]) + b[2] * rec[2 * D + j];
    Nn[j] = (b[0] * rec[3 + j] + b[1] * rec[D + 3 + j]) + b[2] * rec[2 * D + 3 + j];
    tex[j] = D == 9 ? (b[0] * rec[6 + j] + b[1] * rec[D + 6 + j]) + b[2] * rec[2 * D + 6 + j] : tex_in[j];
  }
  const Lit s = light_sample<POINT>(c, P, Nn);
#pragma unroll
  for (int j = 0; j < 3; ++j) color[j] = (amb[j] + kd[j] * s.angle) * tex[j] + ks[j] * s.pw;  // shading.py:96
}

// State of one sample between the forward recompute and the backward chain (FAST arithmetic: reciprocal / rsqrt /
// exp2-log2 forms; gradients are tolerance-gated).
template <int D>
struct ShadeState {
  float tex[3];
  Lit s;
  float apow;  // alpha^(shininess - 1)
  // (the face's record is NOT part of the state: shade_sample_finish reads it again -- an L1 hit -- instead of holding
  // 3 D registers across whatever the caller does in between; soft_phong.hip's backward went from 173 to ~120 VGPRs)
};

template <int D, bool POINT>
__device__ __forceinline__ void shade_sample_prepare(const ShadeConst& c, const float* __restrict__ rec, const float (&b)[3],
                                                     const float (&tex_in)[3], ShadeState<D>& st) {
#pragma clang fp contract(fast)
  float P[3], Nn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    P[j] = (b[0] * rec[j] + b[1] * rec[D + j]) + b[2] * rec[2 * D + j];
    Nn[j] = (b[0] * rec[3 + j] + b[1] * rec[D + 3 + j]) + b[2] * rec[2 * D + 3 + j];
    st.tex[j] = D == 9 ? (b[0] * rec[6 + j] + b[1] * rec[D + 6 + j]) + b[2] * rec[2 * D + 6 + j] : tex_in[j];
  }
  st.s = light_sample<POINT, true>(c, P, Nn);
  // alpha^(s-1) as exp2((s-1) * log2(alpha)): alpha = 0 gives exp2(-+inf) = 0 / inf and 0^0 = exp2(NaN) needs the select
  const float em1 = c.shin - 1.0f;
  st.apow = em1 == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(em1 * __builtin_amdgcn_logf(st.s.alpha));
}

// the sample's forward colour as the backward arithmetic sees it (the fused softmax blend needs it)
template <int D>
__device__ __forceinline__ void shade_sample_color(const float (&amb)[3], const float (&kd)[3], const float (&ks)[3], float pw0,
                                                   const ShadeState<D>& st, float (&color)[3]) {
  const float pw = st.s.alpha > 0.0f ? st.apow * st.s.alpha : pw0;
#pragma unroll
  for (int j = 0; j < 3; ++j) color[j] = (amb[j] + kd[j] * st.s.angle) * st.tex[j] + ks[j] * pw;
}

// Backward of one prepared sample for the upstream colour gradient `go`: g = the 3 D partials of the face record, gb =
// gradient of the barycentrics, dtex = gradient of the texel (D == 6: the caller stores it; D == 9: already folded into
// g), pg (PG) += gradient of the 25 per-image parameters.
template <int D, bool POINT, bool PG>
__device__ __forceinline__ void shade_sample_finish(const ShadeConst& c, const float (&amb)[3], const float (&kd)[3],
                                                    const float (&ks)[3], float pw0, const ShadeState<D>& st,
                                                    const float* __restrict__ rec, const float (&b)[3], const float (&go)[3],
                                                    float (&g)[3 * D],
                                                    float (&gb)[3], float (&dtex)[3],
                                                    float (&pg)[PG ? P3D_SHADE_PARAM_FLOATS : 1]) {
#pragma clang fp contract(fast)  // gradient-only arithmetic (tolerance-gated): let mul+add fuse into FMA
  float r[3 * D];
#pragma unroll
  for (int j = 0; j < 3 * D; ++j) r[j] = rec[j];
  const float (&tex)[3] = st.tex;
  const Lit& s = st.s;
  const float apow = st.apow;
    // pow backward (torch: 0 where the exponent is 0), relu and the cos > 0 mask
    // colour_j = (amb_j + kd_j * angle) * tex_j + ks_j * pw
    float dangle = 0.0f, dpw = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dtex[j] = (amb[j] + kd[j] * s.angle) * go[j];
      dangle += kd[j] * tex[j] * go[j];
      dpw += ks[j] * go[j];
    }
    const float dalpha = c.shin == 0.0f ? 0.0f : c.shin * apow * dpw;
    if constexpr (PG) {
      const float pw = s.alpha > 0.0f ? apow * s.alpha : pw0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float tg = tex[j] * go[j];
        pg[j] += c.ma[j] * tg;                 // light ambient
        pg[12 + j] += c.la[j] * tg;            // material ambient
        pg[3 + j] += c.md[j] * s.angle * tg;   // light diffuse
        pg[15 + j] += c.ld[j] * s.angle * tg;  // material diffuse
        pg[6 + j] += c.ms[j] * pw * go[j];     // light specular
        pg[18 + j] += c.ls[j] * pw * go[j];    // material specular
      }
      // d pow(alpha, s) / d s = pow * ln(alpha), 0 at alpha = 0 (torch's pow backward for s >= 0)
      if (s.alpha > 0.0f) pg[21] += dpw * pw * (__builtin_amdgcn_logf(s.alpha) * 0.6931471805599453f);
    }
    const float dd = (s.cosv > 0.0f && s.d > 0.0f) ? dalpha : 0.0f;
    float dvh[3], dR[3], dlh[3], dnh[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dvh[j] = dd * s.R[j];
      dR[j] = dd * s.vh[j];
    }
    const float dcos = (s.cosv > 0.0f ? dangle : 0.0f) + 2.0f * dot3(dR, s.nh);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dnh[j] = 2.0f * s.cosv * dR[j] + dcos * s.lh[j];
      dlh[j] = -dR[j] + dcos * s.nh[j];
    }
    float dNn[3], dL[3], dV[3], dP[3];
    normalize3_bwd(s.nh, s.nlen, dnh, dNn);
    normalize3_bwd(s.lh, s.llen, dlh, dL);
    normalize3_bwd(s.vh, s.vlen, dvh, dV);
#pragma unroll
    for (int j = 0; j < 3; ++j) dP[j] = (POINT ? -dL[j] : 0.0f) - dV[j];
    if constexpr (PG) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        pg[9 + j] += dL[j];   // light location / direction
        pg[22 + j] += dV[j];  // camera centre
      }
    }
    // interpolation backward (interp_face_attrs.cu:100-118)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        g[i * D + j] = b[i] * dP[j];
        g[i * D + 3 + j] = b[i] * dNn[j];
        acc += r[i * D + j] * dP[j];
        acc += r[i * D + 3 + j] * dNn[j];
        if (D == 9) {
          g[i * D + 6 + j] = b[i] * dtex[j];
          acc += r[i * D + 6 + j] * dtex[j];
        }
      }
      gb[i] = acc;
    }
}

template <int D, bool POINT, bool PG>
__device__ __forceinline__ void shade_sample_bwd(const ShadeConst& c, const float (&amb)[3], const float (&kd)[3],
                                                 const float (&ks)[3], float pw0, const float* __restrict__ rec,
                                                 const float (&b)[3], const float (&tex_in)[3], const float (&go)[3],
                                                 float (&g)[3 * D], float (&gb)[3], float (&dtex)[3],
                                                 float (&pg)[PG ? P3D_SHADE_PARAM_FLOATS : 1]) {
  ShadeState<D> st;
  shade_sample_prepare<D, POINT>(c, rec, b, tex_in, st);
  shade_sample_finish<D, POINT, PG>(c, amb, kd, ks, pw0, st, rec, b, go, g, gb, dtex, pg);
}

}  // namespace
}  // namespace p3d
