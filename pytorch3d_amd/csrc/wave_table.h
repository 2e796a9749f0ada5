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
//   4. list heads fold their totals into the table with plain 16-byte LDS loads / stores (distinct
//      primitives -> distinct slots within one wave instruction).
// The callers are bound by LDS instruction issue, so a step in which no two lanes share a primitive (detected
// from the exchange itself: owners carry a step stamp, nothing is ever reset) skips 2.-3. entirely.
// The table is flushed with global f32 atomics (NV per occupied slot) when it runs full and at
// the end.  Everything here must be called by all 64 lanes of the wave (wave-uniform control flow).
#pragma once

#include "p3d_common.h"

namespace p3d {

constexpr int kEmptyKey = -1;

// LAYOUT of the output the table is flushed to:
//   kRows   out[f * NV + j]                                   (P, NV) rows
//   kPlanar out[j * plane + f], j < nlive                     NV planes of `plane` floats, e.g. (C, P) features
//   kChunk  out[f * plane + (j / 4) * pitch + j % 4], j % 4 < nlive
//           a 4-wide column chunk of (P, NV/4, pitch) records, e.g. channels c0..c0+3 of (F, 3, D) face attributes
//   kCorners out[index[f * (NV/3) + j / 3] * 3 + j % 3]         per-corner xyz partials of face f sent to its vertices
//   kSplit  j < split_at: out[f * split_row + j]; else out2[f * (NV - split_at) + j - split_at]
//           one table for two outputs of the same primitive, e.g. a point's xy gradient (P, 3) and its feature gradient (P, C)
// SPILL: see kFlushAt.
enum { kRows = 0, kPlanar = 1, kChunk = 2, kCorners = 3, kSplit = 4 };
// BUCKET: the keys are probed four at a time (one 16-byte LDS load per bucket of four slots instead of one load per slot;
// SLOTS % 4 == 0).  A step's probe is a chain of dependent LDS round trips whose length is that of the longest chain among
// the wave's primitives -- a fresh primitive in a table that is 3/4 full walks ~8 slots -- and in the rasterizer's
// backward that chain was the largest single part of the table's cost (profiles/r03/bwd_ablate.txt).
template <int NV, int SLOTS, int LAYOUT = kRows, bool SPILL = false, bool BUCKET = false>
struct WaveTable {
  static_assert(!BUCKET || SLOTS % 4 == 0, "bucket probing reads the keys as aligned groups of four");
  static constexpr int kStride = (NV + 3) / 4 * 4;  // floats per slot: values are moved as 16-byte chunks
  // The table is emptied before a step when it is fuller than this.  A step adds up to 64 primitives, so by default 64
  // slots stay in reserve and the table cannot overflow.  SPILL: no reserve -- the table fills up to 3/4 between
  // steps, a lane that finds no free slot for its primitive during a step sends its values straight to memory with
  // atomics, and the table is flushed after that step.  For small tables (3*D = 18..27 values per slot, ~100 slots in
  // the LDS budget of 4 workgroups per CU) that doubles the usable capacity.
  static constexpr int kFlushAt = SPILL ? SLOTS * 3 / 4 : SLOTS - 64;
  static constexpr int kLdsInts = SLOTS * (2 + kStride);

  // keys / owner are LDS pointers BY TYPE: through generic pointers their volatile accesses compile to flat_load / flat_store
  // with system scope (sc0 sc1), which take the vector-memory path to find out that the address is LDS.
  typedef __attribute__((address_space(3))) volatile int LdsInt;
  LdsInt* keys;   // [SLOTS] primitive id or kEmptyKey (volatile: other lanes write between my store and re-load)
  LdsInt* owner;  // [SLOTS] last visitor of the slot, stamped with the step number (never needs resetting)
  float* vals;          // [SLOTS][kStride]
  int used;             // occupied slots (wave-uniform)
  int gen;              // step counter (wave-uniform), > 0
  int64_t plane = 0;        // kPlanar: floats per output plane; kChunk: floats per primitive record
  int64_t fstride = 1;      // kPlanar: floats between the entries of consecutive primitives inside a plane (interleaved outputs: plane 1, fstride NV)
  int pitch = 0;            // kChunk: floats between the NV/4 sub-rows of a record
  int nlive = NV;           // kPlanar: planes that exist; kChunk: live columns of the chunk (others are never flushed)
  float* out2 = nullptr;    // kSplit: the second output
  int split_at = 0;         // kSplit: values [0, split_at) go to `out`, the rest to `out2`
  int split_row = 0;        // kSplit: floats per primitive in `out`
  const int64_t* index = nullptr;  // kCorners: (P, NV/3) vertex ids
  int64_t index_limit = -1;        // kCorners: number of vertices; ids outside [0, limit) have no destination (negative
                                   // ids wrap once, as torch indexing does).  -1: unchecked

  // address of value j of primitive f, or nullptr when that value has no destination
  __device__ __forceinline__ float* dest(float* __restrict__ out, int f, int j) const {
    if constexpr (LAYOUT == kPlanar) return j < nlive ? out + j * plane + (int64_t)f * fstride : nullptr;
    if constexpr (LAYOUT == kSplit)
      return j < split_at ? out + (int64_t)f * split_row + j : out2 + (int64_t)f * (NV - split_at) + (j - split_at);
    if constexpr (LAYOUT == kChunk) return (j & 3) < nlive ? out + (int64_t)f * plane + (j >> 2) * pitch + (j & 3) : nullptr;
    if constexpr (LAYOUT == kCorners) {
      int64_t v = index[(int64_t)f * (NV / 3) + j / 3];
      if (index_limit >= 0) {
        if (v < 0) v += index_limit;
        if (v < 0 || v >= index_limit) return nullptr;
      }
      return out + v * 3 + j % 3;
    }
    return out + (int64_t)f * NV + j;
  }

  __device__ __forceinline__ void init(int* lds, int lane) {
    vals = reinterpret_cast<float*>(lds);  // first: keeps the 16-byte alignment of the workgroup's array
    keys = (LdsInt*)(lds + SLOTS * kStride);
    owner = (LdsInt*)(lds + SLOTS * kStride + SLOTS);
    used = 0;
    gen = 0;
    for (int i = lane; i < SLOTS; i += 64) {
      keys[i] = kEmptyKey;
      owner[i] = 0;
    }
  }

  static __device__ __forceinline__ int hash(int f) {
    return (int)(((unsigned long long)((unsigned)f * 2654435761u) * (unsigned)SLOTS) >> 32);
  }

  static __device__ __forceinline__ float lane_read(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
  }

  // out[f * NV + j] += vals[slot][j] for every occupied slot; empties the table.
  __device__ __forceinline__ void flush(float* __restrict__ out, int lane) {
    // The NV values of an entry leave from ADJACENT lanes (lane = entry * NV + value).  Global float atomics are served per
    // REQUEST, not per lane (profiles/microbench/global_atomic_mi355x.txt: a lane per row 20 G lane-atomics/s whatever NV; the
    // values of a row in adjacent lanes 56 / 84 / 126 G for NV = 3 / 4 / 9), and until round 5 instruction j carried value j of
    // 64 different rows.  Occupied slots are compacted first (a permutation through ds_permute), so a sparse table issues
    // ceil(used * NV / 64) atomic instructions.  No memory read sits between the atomics (kCorners: the vertex ids are loaded
    // by the slots' own lanes up front): a wait for a load would also wait for the atomics before it.
    constexpr int G = 64 / NV;
    constexpr int NC = LAYOUT == kCorners ? NV / 3 : 1;
    const int sub = lane / NV, j = lane - sub * NV;
    for (int s0 = 0; s0 < SLOTS; s0 += 64) {
      const int s = s0 + lane;
      const int f = s < SLOTS ? keys[s] : kEmptyKey;
      const bool occ = f != kEmptyKey;
      const unsigned long long m = __ballot(occ);
      if (m == 0) continue;  // uniform
      const int n = __popcll(m);
      int vid[NC];
      if constexpr (LAYOUT == kCorners) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          int64_t v = occ ? index[(int64_t)f * NC + c] : -1;
          if (index_limit >= 0) {
            if (v < 0) v += index_limit;
            if (v < 0 || v >= index_limit) v = -1;
          }
          vid[c] = (int)v;
        }
      }
      // lane r of `holder` = the lane whose slot is the r-th occupied one (the others fill the tail: a full permutation)
      const int dst = occ ? mask_rank(m) : n + mask_rank(~m);
      const int holder = __builtin_amdgcn_ds_permute(dst << 2, lane);
      for (int e0 = 0; e0 < n; e0 += G) {
        const int r = e0 + sub;
        const int sl = __builtin_amdgcn_ds_bpermute((r & 63) << 2, holder);
        float* o = nullptr;
        if constexpr (LAYOUT == kCorners) {
          int v = __builtin_amdgcn_ds_bpermute(sl << 2, vid[0]);
#pragma unroll
          for (int c = 1; c < NC; ++c) {
            const int t = __builtin_amdgcn_ds_bpermute(sl << 2, vid[c]);
            v = j / 3 == c ? t : v;
          }
          if (v >= 0 || index_limit < 0) o = out + (int64_t)v * 3 + j % 3;
        } else {
          o = dest(out, __builtin_amdgcn_ds_bpermute(sl << 2, f), j);
        }
        if (sub < G && r < n && o) unsafeAtomicAdd(o, vals[(s0 + sl) * kStride + j]);
      }
      if (occ) keys[s] = kEmptyKey;
    }
    used = 0;
  }

  // One step: every lane with f >= 0 contributes g[0..NV) to primitive f.  g is clobbered.
  // LDS instruction count is what bounds the callers, so the common step -- no two lanes share a primitive -- costs
  // one key load, one exchange and kStride/4 16-byte loads + stores per lane, nothing else.
  __device__ __forceinline__ void add(float* __restrict__ out, int lane, int f, float (&g)[NV]) {
    if (used > kFlushAt) flush(out, lane);
    ++gen;
    const bool active = f >= 0;
    int slot = -1;
    bool fresh = false;
    bool spill = false;  // no free slot left for this lane's primitive
    if (BUCKET && active) {
      // Invariant (no deletions between flushes): a key lives in the first bucket of its probe sequence that had an empty
      // slot when it was inserted, and every bucket before it is full for good -- so a lookup that tests a bucket for the
      // key BEFORE it looks for an empty slot finds it.
      constexpr int NB = SLOTS / 4;
      int b = (int)(((unsigned long long)((unsigned)f * 2654435761u) * (unsigned)NB) >> 32);
      int tries = 0;
      for (;;) {
        typedef int KeyQuad __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) volatile KeyQuad LdsQuad;
        LdsInt* kb = keys + 4 * b;
        const KeyQuad k = *(LdsQuad*)kb;  // one ds_read_b128
        const int hit = k.x == f ? 0 : (k.y == f ? 1 : (k.z == f ? 2 : (k.w == f ? 3 : -1)));
        if (hit >= 0) {
          slot = 4 * b + hit;
          break;
        }
        const int e = k.x == kEmptyKey ? 0 : (k.y == kEmptyKey ? 1 : (k.z == kEmptyKey ? 2 : (k.w == kEmptyKey ? 3 : -1)));
        if (e >= 0) {
          kb[e] = f;  // several primitives may race for one empty slot: the last store wins, the losers read the bucket again
          if (kb[e] == f) {
            slot = 4 * b + e;
            fresh = true;
            break;
          }
          continue;
        }
        b = (b + 1 == NB) ? 0 : b + 1;
        if (SPILL && ++tries == NB) {
          spill = true;
          break;
        }
      }
    } else if (active) {
      int h = hash(f);
      int tries = 0;
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
        if (SPILL && ++tries == SLOTS) {
          spill = true;
          break;
        }
      }
    }
    if (SPILL && spill) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float* o = dest(out, f, j);
        if (o) unsafeAtomicAdd(o, g[j]);
      }
    }
    const bool linked = active && !spill;
    // chain the visitors of each slot: prev = the lane that visited before me in THIS step (stamp check)
    const int stamp = (gen << 6) | lane;
    int prev = -1;
    if (linked) {
      const int old = __hip_atomic_exchange((__attribute__((address_space(3))) int*)(owner + slot), stamp, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
      prev = (old >> 6) == gen ? (old & 63) : -1;
    }
    bool head = linked;
    if (__ballot(prev >= 0)) {  // wave-uniform: some primitive is hit by more than one lane
      head = linked && owner[slot] == stamp;  // the last visitor heads the list
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
    }
    if (head) {
      float4* row = reinterpret_cast<float4*>(vals + slot * kStride);
#pragma unroll
      for (int c = 0; c < kStride / 4; ++c) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!fresh) t = row[c];
        t.x += g[4 * c];
        if (4 * c + 1 < NV) t.y += g[4 * c + 1];
        if (4 * c + 2 < NV) t.z += g[4 * c + 2];
        if (4 * c + 3 < NV) t.w += g[4 * c + 3];
        row[c] = t;
      }
    }
    used += __popcll(__ballot(head && fresh));
    if (SPILL && __ballot(spill)) flush(out, lane);
  }
};

}  // namespace p3d
