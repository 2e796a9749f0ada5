// Microbenchmark: global float atomics (global_atomic_add_f32, no return) on gfx950 -- does it matter which lanes carry the
// values of one output row?  A scatter-type backward flushes entries of NV floats to random rows of a (P, NV) array:
//   mode 0  a lane per ROW, NV instructions (instruction j adds value j of 64 different rows)          -- wave_table.h until round 5
//   mode 1  a lane per VALUE, lanes of one row adjacent (lane = entry * NV + j): ceil(64 / (64 / NV)) instructions for 64 rows
// Same number of lane-atomics either way.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics global_atomic.hip -o global_atomic.bin && ./global_atomic.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

template <int NV, int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned rows, unsigned window, int rounds) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  // a wave's rows come from a window of `window` rows (window == rows: anywhere)
  const unsigned w0 = window >= rows ? 0u : (mix(wave) % (rows - window));
  for (int r = 0; r < rounds; ++r) {
    if (MODE == 0) {
      const unsigned row = w0 + mix(wave * 7919u + r * 64u + lane) % window;
#pragma unroll
      for (int j = 0; j < NV; ++j) unsafeAtomicAdd(out + (size_t)row * NV + j, 1.0f);
    } else {
      constexpr int G = 64 / NV;
      const int sub = lane / NV, j = lane - sub * NV;
      for (int e0 = 0; e0 < 64; e0 += G) {
        const int e = e0 + sub;
        const unsigned row = w0 + mix(wave * 7919u + r * 64u + e) % window;
        if (sub < G && e < 64) unsafeAtomicAdd(out + (size_t)row * NV + j, 1.0f);
      }
    }
  }
}

template <int NV, int MODE>
void run(float* d, unsigned rows, unsigned window) {
  const int blocks = 4096, rounds = 20;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<NV, MODE><<<blocks, 256>>>(d, rows, window, 2);
  hipEventRecord(a);
  k<NV, MODE><<<blocks, 256>>>(d, rows, window, rounds);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double entries = (double)blocks * 4 * rounds * 64;
  printf("NV=%d  %-16s window=%8u rows  %8.3f ms  %7.2f G rows/s  %7.2f G lane-atomics/s\n", NV,
         MODE == 0 ? "lane per row" : "values adjacent", window, ms, entries / ms / 1e6, entries * NV / ms / 1e6);
}

int main() {
  const unsigned rows = 1u << 20;
  float* d;
  hipMalloc(&d, (size_t)rows * 12 * 4);
  hipMemset(d, 0, (size_t)rows * 12 * 4);
  for (unsigned window : {rows, 65536u, 4096u}) {
    run<3, 0>(d, rows, window);
    run<3, 1>(d, rows, window);
    run<4, 0>(d, rows, window);
    run<4, 1>(d, rows, window);
    run<9, 0>(d, rows, window);
    run<9, 1>(d, rows, window);
  }
  return 0;
}
