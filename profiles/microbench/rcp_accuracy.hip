// Probe: how accurate are the hardware reciprocal seeds of gfx950, and what do Newton steps in double make of them?
//
//   hipcc --offload-arch=gfx950 -O2 rcp_accuracy.hip -o rcp_accuracy.bin && ./rcp_accuracy.bin
//
// csrc/p3d_geom.h needs per-pixel reciprocals rd with |rd * d - 1| <= 2^-52 (exact_div's argument).  Today: v_rcp_f32 seed
// (1 ulp = 2^-23) + TWO Newton steps in double (four f64 FMAs); the per-face ones: v_rcp_f64 seed + two steps.  If v_rcp_f64
// alone is good to ~2^-27, ONE step after it reaches 2^-54 -- two instructions fewer per reciprocal in the fine kernel's
// inner loop.  Reported: the largest |r * d - 1| (evaluated with an exact FMA in double) over 2^28 operands per seed kind
// and number of steps, as log2.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ double newton(double d, double r) {
  const double e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(e, r, r);
}

// out[kind * 3 + steps] = max |r d - 1| as a double bit pattern (atomicMax on the bits: all values are >= 0)
__global__ void probe(unsigned long long seed, unsigned long long* out) {
  unsigned long long s = seed + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  double worst[6] = {0, 0, 0, 0, 0, 0};
  for (int it = 0; it < 1024; ++it) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    // a float operand (what the kernels divide by): random mantissa, exponent in [-60, 60]
    const unsigned m = (unsigned)(s & 0x7fffff);
    const unsigned e = 127u - 60u + (unsigned)((s >> 23) % 121);
    const float df = __uint_as_float((e << 23) | m);
    const double d = (double)df;
    double r32 = (double)__builtin_amdgcn_rcpf(df);
    double r64 = __builtin_amdgcn_rcp(d);
    for (int st = 0; st < 3; ++st) {
      const double e32 = fabs(__builtin_fma(r32, d, -1.0)), e64 = fabs(__builtin_fma(r64, d, -1.0));
      worst[st] = fmax(worst[st], e32);
      worst[3 + st] = fmax(worst[3 + st], e64);
      r32 = newton(d, r32);
      r64 = newton(d, r64);
    }
  }
  for (int k = 0; k < 6; ++k) atomicMax(&out[k], (unsigned long long)__double_as_longlong(worst[k]));
}

int main() {
  unsigned long long* d_out;
  CHECK(hipMalloc(&d_out, 6 * 8));
  CHECK(hipMemset(d_out, 0, 6 * 8));
  probe<<<1024, 256>>>(12345, d_out);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[6];
  CHECK(hipMemcpy(h, d_out, 48, hipMemcpyDeviceToHost));
  const char* kind[2] = {"v_rcp_f32 seed", "v_rcp_f64 seed"};
  printf("# max |r * d - 1| over 2^28 float operands d (exponents -60..60), log2; exact_div needs <= -52\n");
  for (int k = 0; k < 2; ++k)
    for (int st = 0; st < 3; ++st) {
      double v;
      memcpy(&v, &h[k * 3 + st], 8);
      printf("%s + %d Newton step(s) in double: %.3e  (2^%.1f)\n", kind[k], st, v, v > 0 ? log2(v) : -1074.0);
    }
  return 0;
}
