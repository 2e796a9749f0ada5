// Brute-force check of csrc/p3d_geom.h: exact_div(n, recip_for_div(d)) == n / d (IEEE) and the cubic refinement's residual.
//
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I pytorch3d_amd/csrc profiles/microbench/exact_div_check.hip -o /tmp/edc && /tmp/edc
//
// Round 4 replaced the two Newton steps of recip_newton by one cubic step (three v_fma_f64 instead of four).  Reported:
// max |rd * d - 1| (needs <= 2^-52) for both seeds, and the number of (n, d) pairs -- 2^31 random float pairs of ordinary
// magnitude (what FaceRec::wide == false guarantees) + 2^29 pairs with n an exact multiple of d plus or minus one ulp (the
// quotients closest to rounding boundaries) -- where the float quotient differs from the IEEE division.  Expected: 0.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "p3d_geom.h"

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ unsigned long long xs(unsigned long long& s) {
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  return s;
}

// out: [0] worst residual f32 seed (double bits), [1] worst residual f64 seed, [2] mismatches random, [3] mismatches near-boundary
__global__ void probe(unsigned long long seed, unsigned long long* out) {
  unsigned long long s = seed + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  double w32 = 0, w64 = 0;
  unsigned long long bad = 0, bad_nb = 0;
  for (int it = 0; it < 8192; ++it) {
    unsigned long long r = xs(s);
    const unsigned md = (unsigned)(r & 0x7fffff), ed = 127u - 40u + (unsigned)((r >> 23) % 81);
    const float d = __uint_as_float((ed << 23) | md | ((unsigned)(r >> 63) << 31));
    r = xs(s);
    const unsigned mn = (unsigned)(r & 0x7fffff), en = 127u - 40u + (unsigned)((r >> 23) % 81);
    const float n = __uint_as_float((en << 23) | mn | ((unsigned)(r >> 63) << 31));
    const double rd = p3d::recip_for_div(d), rw = p3d::recip_for_div_wide(d);
    w32 = fmax(w32, fabs(__builtin_fma(rd, (double)d, -1.0)));
    w64 = fmax(w64, fabs(__builtin_fma(rw, (double)d, -1.0)));
    const float want = n / d;
    bad += (__float_as_uint(p3d::exact_div(n, rd)) != __float_as_uint(want)) + (__float_as_uint(p3d::exact_div(n, rw)) != __float_as_uint(want));
    if ((it & 3) == 0) {
      // n2 = (q * d) +- 1 ulp for a random float q: quotients next to representable values / midpoints
      const float q = __uint_as_float(((127u - 8u + (unsigned)((r >> 40) % 17)) << 23) | (unsigned)((r >> 17) & 0x7fffff));
      const float prod = q * d;
      const float n2 = __uint_as_float(__float_as_uint(prod) + ((r >> 62) & 1 ? 1u : 0xffffffffu));
      const float want2 = n2 / d;
      bad_nb += (__float_as_uint(p3d::exact_div(n2, rd)) != __float_as_uint(want2)) + (__float_as_uint(p3d::exact_div(n2, rw)) != __float_as_uint(want2));
    }
  }
  atomicMax(&out[0], (unsigned long long)__double_as_longlong(w32));
  atomicMax(&out[1], (unsigned long long)__double_as_longlong(w64));
  atomicAdd(&out[2], bad);
  atomicAdd(&out[3], bad_nb);
}

int main() {
  unsigned long long* d_out;
  CHECK(hipMalloc(&d_out, 4 * 8));
  CHECK(hipMemset(d_out, 0, 4 * 8));
  probe<<<1024, 256>>>(987654321, d_out);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[4];
  CHECK(hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost));
  double a, b;
  memcpy(&a, &h[0], 8);
  memcpy(&b, &h[1], 8);
  printf("# csrc/p3d_geom.h recip_newton (one cubic step) on gfx950: 2^31 random (n, d) float pairs, exponents -40..40, both signs\n");
  printf("max |rd * d - 1|, v_rcp_f32 seed: %.3e (2^%.2f)   v_rcp_f64 seed: %.3e (2^%.2f)   [exact_div needs <= 2^-52]\n", a, log2(a), b, log2(b));
  printf("exact_div(n, rd) != n / d: %llu of 2^32 random quotients (both seeds), %llu of 2^30 next to products q * d +- 1 ulp\n", h[2], h[3]);
  return (h[2] | h[3]) ? 1 : 0;
}
