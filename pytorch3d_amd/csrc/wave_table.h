// wave_table.h -- wave-private LDS accumulation of per-primitive float partials without float atomics.
//
// Used by the scatter-type backward kernels (mesh rasterization: 9 partials per face;
// interpolate_face_attributes: 3*D partials per face).  gfx950 executes ds_add_f32 one lane at a
// time (~190-260 CU-cycles per wave instruction, profiles/microbench/lds_atomic.hip), and
// device-scope global float atomics resolve beyond the per-XCD L2, so neither may sit in the
// per-sample path.  Instead, per step (one sample per lane):
//   1. every lane finds / claims the hash-table slot of its primitive (plain LDS ld/st: lanes of
//      one primitive probe in lockstep and see the same thing; races between different primitives
//      for one empty slot are resolved by re-reading the key);
//   2. the visitors of a slot are chained with ONE integer LDS exchange per lane (ds_wrxchg_rtn
//      returns the previous visitor); the last visitor heads the list;
//   3. partials are summed along the lists by pointer jumping over ds_bpermute (log2(group) steps);
//   4. list heads fold their totals into the table with plain LDS loads / stores (distinct
//      primitives -> distinct slots within one wave instruction).
// The table is flushed with global f32 atomics (NV per occupied slot) when it runs full and at
// the end.  Everything here must be called by all 64 lanes of the wave (wave-uniform control flow).
#pragma once

#include "p3d_common.h"

namespace p3d {

constexpr int kEmptyKey = -1;

template <int NV, int SLOTS>
struct WaveTable {
  static constexpr int kFlushAt = SLOTS - 64;  // a step adds at most 64 primitives: the table cannot overflow
  static constexpr int kLdsInts = SLOTS * (2 + NV);

  volatile int* keys;   // [SLOTS] primitive id or kEmptyKey (volatile: other lanes write between my store and re-load)
  volatile int* owner;  // [SLOTS] scratch for the per-step visitor lists, -1 between steps
  float* vals;          // [NV][SLOTS]
  int used;             // occupied slots (wave-uniform)
  bool no_atomics = false;  // ablation only (profiles/ablate.py): drop the global atomics of flush()

  __device__ __forceinline__ void init(int* lds, int lane) {
    keys = lds;
    owner = lds + SLOTS;
    vals = reinterpret_cast<float*>(lds + 2 * SLOTS);
    used = 0;
    for (int i = lane; i < SLOTS; i += 64) {
      keys[i] = kEmptyKey;
      owner[i] = -1;
    }
  }

  static __device__ __forceinline__ int hash(int f) {
    return (int)(((unsigned long long)((unsigned)f * 2654435761u) * (unsigned)SLOTS) >> 32);
  }

  static __device__ __forceinline__ float lane_read(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
  }

  // out[f * NV + j] += vals[j][slot] for every occupied slot; empties the table.
  __device__ __forceinline__ void flush(float* __restrict__ out, int lane) {
    for (int s = lane; s < SLOTS; s += 64) {
      const int f = keys[s];
      if (f != kEmptyKey) {
        float* o = out + (int64_t)f * NV;
#pragma unroll
        for (int j = 0; j < NV; ++j)
          if (!no_atomics) unsafeAtomicAdd(o + j, vals[j * SLOTS + s]);
        keys[s] = kEmptyKey;
      }
    }
    used = 0;
  }

  // One step: every lane with f >= 0 contributes g[0..NV) to primitive f.  g is clobbered.
  __device__ __forceinline__ void add(float* __restrict__ out, int lane, int f, float (&g)[NV]) {
    if (used > kFlushAt) flush(out, lane);
    const bool active = f >= 0;
    int slot = -1;
    bool fresh = false;
    if (active) {
      int h = hash(f);
      for (;;) {
        const int cur = keys[h];
        if (cur == f) {
          slot = h;
          break;
        }
        if (cur == kEmptyKey) {
          keys[h] = f;  // several primitives may race for one empty slot: the last store wins
          if (keys[h] == f) {
            slot = h;
            fresh = true;
            break;
          }
        }
        h = (h + 1 == SLOTS) ? 0 : h + 1;
      }
    }
    int prev = -1;
    if (active) prev = atomicExch(const_cast<int*>(&owner[slot]), lane);
    const bool head = active && owner[slot] == lane;  // the last visitor heads the list
    // after step s every lane holds the sum of the 2^s list entries starting at itself
    while (__ballot(prev >= 0)) {
      const int src = prev >= 0 ? prev : lane;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float o = lane_read(g[j], src);
        if (prev >= 0) g[j] += o;
      }
      const int pp = __builtin_amdgcn_ds_bpermute(src << 2, prev);
      prev = prev >= 0 ? pp : -1;
    }
    if (head) {
      owner[slot] = -1;
      if (fresh) {
#pragma unroll
        for (int j = 0; j < NV; ++j) vals[j * SLOTS + slot] = g[j];
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) vals[j * SLOTS + slot] += g[j];
      }
    }
    used += __popcll(__ballot(head && fresh));
  }
};

}  // namespace p3d
