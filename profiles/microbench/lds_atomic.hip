// Microbenchmark: LDS float-atomic throughput on gfx950 as a function of same-address conflicts.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>  // 0: ds_add_f32, 1: ds_add_u32, 2: plain ds_write (no atomic), 3: ds_add_rtn_f32
__global__ __launch_bounds__(256) void k(int conflict, int iters, float* out) {
  __shared__ float s[9 * 1024];
  for (int i = threadIdx.x; i < 9 * 1024; i += 256) s[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  // lanes are grouped in `conflict`-sized groups that share an address; groups hit distinct banks
  int slot = ((lane / conflict) * 37 + w * 251) & 1023;
  float v = 1.0f + lane;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      if (MODE == 0) unsafeAtomicAdd(&s[j * 1024 + slot], v);
      if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&s[j * 1024 + slot]), 1u);
      if (MODE == 2) s[j * 1024 + slot] = v;
      if (MODE == 3) acc += atomicAdd(&s[j * 1024 + slot], v);
    }
    slot = (slot + 97) & 1023;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[slot] + acc;
}

template <int MODE>
void run(const char* name, float* d) {
  for (int conflict : {1, 2, 4, 8, 16, 64}) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(conflict, 10, d);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(conflict, iters, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)blocks * 256 * iters * 9;
    // per CU: 4 blocks resident; cycles at 2.4 GHz
    const double cyc_per_wave_instr = ms * 1e-3 * 2.4e9 / ((double)blocks / 256 * 4 * iters * 9);
    printf("%-14s conflict=%2d  %8.3f ms  %7.1f Glane-op/s  %6.1f CU-cycles per wave-instr\n", name, conflict, ms,
           lane_ops / ms / 1e6, cyc_per_wave_instr);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 4096 * 4);
  run<0>("ds_add_f32", d);
  run<3>("ds_add_rtn_f32", d);
  run<1>("ds_add_u32", d);
  run<2>("ds_write_b32", d);
  return 0;
}
