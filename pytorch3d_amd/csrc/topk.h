// topk.h -- per-pixel "K nearest" queues for the fine / naive rasterizers.
//
// Target semantics ("Top-K"): the K smallest candidates under the total
// order (z ascending, primitive index ascending), emitted in that order -- what the reference's
// CPU queue (rasterize_meshes_cpu.cpp:281-285) and Python implementation do and what
// test_order_of_ties pins.  The reference's CUDA queue (rasterize_meshes.cu:216-237, an unsorted
// array with a tracked maximum + final BubbleSort) yields the same set except when several queued
// entries tie exactly at the maximum z while the queue overflows.
//
// TopKReg keeps the queue sorted in VGPRs (compile-time capacity KT, fully unrolled
// compare-exchange insertion: no scratch, no divergence inside an insertion).  TopKMem is the
// K <= 150 fallback in private memory.  NP = number of float payload words per entry.
// Host-compilable like p3d_geom.h, so tests/ can exercise the queue logic without a GPU.
#pragma once

#include <float.h>
#include <math.h>
#include <stdint.h>

#include "p3d_geom.h"  // P3D_HDM
#if defined(__HIP_DEVICE_COMPILE__)
#include "topk_insert_asm.h"
#endif

namespace p3d {

constexpr int kEmptyIdx = 0x7fffffff;

template <int KT, int NP>
struct TopKReg {
  static constexpr bool kKeyOrder = false;  // entries are ordered by float compares: -0.0 == +0.0
  static constexpr int kPayload = NP;       // payload words an entry carries (0: the consumer recomputes them from the index)
  static constexpr int NPS = NP > 0 ? NP : 1;  // storage rows; NP = 0: no payload (the one row is never touched and costs no register)
  float z[KT];
  int idx[KT];
  float pl[NPS][KT];

  // (kz, ki): the K-th entry of the live queue -- what a candidate has to beat -- cached so that the
  // admission test is one compare also when K < KT (+inf / kEmptyIdx while the queue has room).
  float kz;
  int ki;

  P3D_HDM void init() {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      z[k] = INFINITY;
      idx[k] = kEmptyIdx;
#pragma unroll
      for (int p = 0; p < NP; ++p) pl[p][k] = -1.0f;
    }
    kz = INFINITY;
    ki = kEmptyIdx;
  }

  P3D_HDM void refresh_kth(int K) {
    if (K >= KT) {
      kz = z[KT - 1];
      ki = idx[KT - 1];
    } else {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k == K - 1) {
          kz = z[k];
          ki = idx[k];
        }
      }
    }
  }

  // Sorted insertion; the largest entry falls off the end.  `K` <= KT is the live capacity.
  // Shift formulation: lt[k] = "the candidate sorts before entry k" is monotone in k (the queue is
  // sorted), so entry k becomes  lt[k] ? (lt[k-1] ? old entry k-1 : candidate) : itself.  All KT
  // slots update independently of each other (select depth 2) -- a compare-exchange chain that
  // carries the displaced entry from slot to slot has the same instruction count but a dependency
  // chain 2*KT long, which a SIMD with 3-4 resident waves cannot hide.
  P3D_HDM void insert(int K, float cz, int cidx, const float (&cpl)[NPS]) {
    bool lt[KT];
#pragma unroll
    // bitwise, not short-circuit: `||` / `&&` compile to exec-masked branches (12 instructions per slot instead of 5)
    for (int k = 0; k < KT; ++k) lt[k] = (cz < z[k]) | ((cz == z[k]) & (cidx < idx[k]));
#pragma unroll
    for (int k = KT - 1; k >= 1; --k) {
      const float tz = lt[k - 1] ? z[k - 1] : cz;
      const int ti = lt[k - 1] ? idx[k - 1] : cidx;
      z[k] = lt[k] ? tz : z[k];
      idx[k] = lt[k] ? ti : idx[k];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float tp = lt[k - 1] ? pl[p][k - 1] : cpl[p];
        pl[p][k] = lt[k] ? tp : pl[p][k];
      }
    }
    z[0] = lt[0] ? cz : z[0];
    idx[0] = lt[0] ? cidx : idx[0];
#pragma unroll
    for (int p = 0; p < NP; ++p) pl[p][0] = lt[0] ? cpl[p] : pl[p][0];
    if (K < KT) {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k >= K) {
          z[k] = INFINITY;
          idx[k] = kEmptyIdx;
        }
      }
    }
    refresh_kth(K);
  }

  // Exact pre-test: can (cz, cidx) enter the queue at all?
  P3D_HDM bool admits(int /*K*/, float cz, int cidx) const { return (cz < kz) | ((cz == kz) & (cidx < ki)); }

  // Depth a candidate must not exceed to enter the queue: the K-th entry's z, +inf while the queue has room.
  P3D_HDM float kth_z(int /*K*/) const { return kz; }
  P3D_HDM int kth_i(int /*K*/) const { return ki; }  // index of the K-th entry (kEmptyIdx while the queue has room)

  // Position of primitive `want` in the queue, or -1.
  P3D_HDM int find(int want) const {
    int at = -1;
#pragma unroll
    for (int k = KT - 1; k >= 0; --k) {
      if (idx[k] == want) at = k;
    }
    return at;
  }

  P3D_HDM float payload_at(int p, int at) const {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k == at) v = pl[p][k];
    }
    return v;
  }

  // Remove entry `at` (keeps order).
  P3D_HDM void erase(int at) {
#pragma unroll
    for (int k = 0; k < KT - 1; ++k) {
      if (k >= at) {
        z[k] = z[k + 1];
        idx[k] = idx[k + 1];
#pragma unroll
        for (int p = 0; p < NP; ++p) pl[p][k] = pl[p][k + 1];
      }
    }
    z[KT - 1] = INFINITY;
    idx[KT - 1] = kEmptyIdx;
    kz = INFINITY;  // the queue has room again
    ki = kEmptyIdx;
  }

  P3D_HDM bool valid(int k) const { return idx[k] != kEmptyIdx; }

  // entry accessors (the kernels read entries through these, so that queue layouts can differ: see TopKPairs)
  P3D_HDM float zf(int k) const { return z[k]; }
  P3D_HDM int ix(int k) const { return idx[k]; }
  P3D_HDM float pay(int p, int k) const { return pl[p][k]; }
};

template <int KMAX, int NP>
struct TopKMem {
  static constexpr bool kKeyOrder = false;
  static constexpr int kPayload = NP;
  float z[KMAX];
  int idx[KMAX];
  float pl[NP][KMAX];
  int n;

  P3D_HDM void init() { n = 0; }

  P3D_HDM void insert(int K, float cz, int cidx, const float (&cpl)[NP]) {
    int j = n;
    while (j > 0 && ((cz < z[j - 1]) || (cz == z[j - 1] && cidx < idx[j - 1]))) --j;
    if (j >= K) return;
    const int last = n < K ? n : K - 1;
    for (int m = last; m > j; --m) {
      z[m] = z[m - 1];
      idx[m] = idx[m - 1];
      for (int p = 0; p < NP; ++p) pl[p][m] = pl[p][m - 1];
    }
    z[j] = cz;
    idx[j] = cidx;
    for (int p = 0; p < NP; ++p) pl[p][j] = cpl[p];
    if (n < K) ++n;
  }

  P3D_HDM bool admits(int K, float cz, int cidx) const {
    if (n < K) return true;
    return (cz < z[K - 1]) || (cz == z[K - 1] && cidx < idx[K - 1]);
  }

  P3D_HDM float kth_z(int K) const { return n < K ? INFINITY : z[K - 1]; }
  P3D_HDM int kth_i(int K) const { return n < K ? kEmptyIdx : idx[K - 1]; }

  P3D_HDM int find(int want) const {
    for (int k = 0; k < n; ++k)
      if (idx[k] == want) return k;
    return -1;
  }

  P3D_HDM float payload_at(int p, int at) const { return pl[p][at]; }

  P3D_HDM void erase(int at) {
    for (int k = at; k < n - 1; ++k) {
      z[k] = z[k + 1];
      idx[k] = idx[k + 1];
      for (int p = 0; p < NP; ++p) pl[p][k] = pl[p][k + 1];
    }
    --n;
  }

  P3D_HDM bool valid(int k) const { return k < n; }

  P3D_HDM float zf(int k) const { return z[k]; }
  P3D_HDM int ix(int k) const { return idx[k]; }
  P3D_HDM float pay(int p, int k) const { return pl[p][k]; }
};

// ---------------------------------------------------------------------------------------------------------------
// TopKPairs -- TopKReg<KT, 4> with every entry held as three 64-bit register PAIRS (z | idx, payload 0 | 1, payload
// 2 | 3) and the sorted insertion done with v_pk_mov_b32 under the lane mask instead of v_cndmask_b32.  The fine
// rasterizers are bound by VALU issue (profiles/microbench/valu_issue.hip: v_cndmask / v_cmp / v_pk_mov all cost ~3.1
// cycles of a SIMD with four waves resident) and the 8-entry insertion network is 97 v_cndmask + 24 v_cmp of the 278 VALU
// instructions of the K = 8 inner loop; in pairs it is 45 v_pk_mov + 22 compares (measured, round 3: mesh_fine 1.49 ->
// 1.31 ms at K = 8 with the 64-bit key, points K = 100 9.1 -> 3.0 ms).  lt[k] = "the candidate sorts before entry k" is
// monotone in k, so top-down
//     lanes with lt[k]:    entry k <- candidate            (3 pair moves)
//     lanes with lt[k-1]:  entry k <- entry k-1            (3 pair moves; lt[k-1] implies lt[k])
// leaves entry k = lt[k-1] ? old k-1 : (lt[k] ? candidate : itself) -- TopKReg::insert's result -- in 6 instructions
// per entry instead of 12.  Same interface and semantics as TopKReg<KT, 4> for K == KT (the exact-K kernels).  On the
// host the two masked moves are plain ifs, so tests/hostgeom can check the scheme against TopKReg.
// ---------------------------------------------------------------------------------------------------------------
#if defined(__clang__)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#else
typedef unsigned int u32x2 __attribute__((vector_size(8)));
#endif

P3D_HDM unsigned f32_bits(float v) {
  union {
    float f;
    unsigned u;
  } c;
  c.f = v;
  return c.u;
}
P3D_HDM float bits_f32(unsigned v) {
  union {
    float f;
    unsigned u;
  } c;
  c.u = v;
  return c.f;
}
// A wave-uniform integer the optimizer may not reason about (no instruction is emitted for it).
P3D_HDM int opaque_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (!__builtin_constant_p(v)) asm volatile("" : "+s"(v));
#endif
  return v;
}

P3D_HDM u32x2 mk_pair(unsigned lo, unsigned hi) {
  u32x2 r;
  r[0] = lo;
  r[1] = hi;
  return r;
}

// KEY64 (kernels whose depths are never -0.0: perspective-correct + clipped barycentrics give z = sum of products of
// non-negative numbers; point depths are staged with -0.0 canonicalised): the pair is laid out idx | z so that, as one unsigned 64-bit number, it
// orders like (z, idx) for z >= +0 -- "sorts before" is then ONE 64-bit compare instead of three 32-bit ones.  NaN
// depths compare above +inf (the empty entry) as bit patterns and are never admitted, as with the float compares.
// NP: 4 payload words (meshes) or none (long point queues: the entry is the one pair).
template <int KT, bool KEY64 = false, int NP = 4>
struct TopKPairs {
  static_assert(NP == 4 || NP == 0, "payload pairs: two or none");
  static constexpr bool kKeyOrder = KEY64;  // ordered by the bit pattern of z: the caller supplies depths >= +0
  static constexpr int kPayload = NP;
  static constexpr int KP = NP == 4 ? KT : 1;
  static constexpr int NPS = NP > 0 ? NP : 1;
  u32x2 zi[KT];  // z bits, idx (KEY64: idx, z bits)
  u32x2 pa[KP];  // payload 0, 1
  u32x2 pb[KP];  // payload 2, 3
  float kz;
  int ki;
  static constexpr int kZ = KEY64 ? 1 : 0, kI = KEY64 ? 0 : 1;  // which half holds what

  P3D_HDM static u32x2 mk_entry(float z, int idx) { return KEY64 ? mk_pair((unsigned)idx, f32_bits(z)) : mk_pair(f32_bits(z), (unsigned)idx); }
  P3D_HDM static unsigned long long key_of(u32x2 e) { return ((unsigned long long)e[1] << 32) | (unsigned long long)e[0]; }

  P3D_HDM void init() {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      zi[k] = mk_entry(INFINITY, kEmptyIdx);
      if (NP == 4) {
        pa[k % KP] = mk_pair(f32_bits(-1.0f), f32_bits(-1.0f));
        pb[k % KP] = pa[k % KP];
      }
    }
    kz = INFINITY;
    ki = kEmptyIdx;
  }

  P3D_HDM float zf(int k) const { return bits_f32(zi[k][kZ]); }
  P3D_HDM int ix(int k) const { return (int)zi[k][kI]; }
  P3D_HDM float pay(int p, int k) const {
    const int j = k % KP;
    return bits_f32(p == 0 ? pa[j][0] : (p == 1 ? pa[j][1] : (p == 2 ? pb[j][0] : pb[j][1])));
  }
  P3D_HDM bool valid(int k) const { return ix(k) != kEmptyIdx; }
  P3D_HDM bool admits(int /*K*/, float cz, int cidx) const {
    if (KEY64) return key_of(mk_entry(cz, cidx)) < key_of(mk_entry(kz, ki));
    return (cz < kz) | ((cz == kz) & (cidx < ki));
  }
  P3D_HDM float kth_z(int /*K*/) const { return kz; }
  P3D_HDM int kth_i(int /*K*/) const { return ki; }  // index of the K-th entry (kEmptyIdx while the queue has room)

#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned long long LaneMask;
  // (the masked-move blocks below change SCC -- s_and_saveexec / s_and write it -- and say so: since round 4 scalar compares
  // on a runtime K sit right next to them, and a compare result the compiler kept in SCC across a block came back wrong)
  // three compares straight into lane masks, combined on the scalar unit (a ballot of the combined bool goes through
  // a v_cndmask + v_cmp pair)
  __device__ __forceinline__ LaneMask sorts_before(float cz, int cidx, int k) const {
    if (KEY64) return __builtin_amdgcn_ballot_w64(key_of(mk_entry(cz, cidx)) < key_of(zi[k]));
    const LaneMask lt = __builtin_amdgcn_ballot_w64(cz < zf(k));
    const LaneMask eq = __builtin_amdgcn_ballot_w64(cz == zf(k));
    const LaneMask il = __builtin_amdgcn_ballot_w64(cidx < ix(k));
    return lt | (eq & il);
  }
  // does any lane of this insertion hold entry k?  (wave-uniform)
  __device__ __forceinline__ bool any_holds(int k) const { return __builtin_amdgcn_ballot_w64(ix(k) != kEmptyIdx) != 0; }
  // entry k <- candidate where m, then entry k <- entry k-1 where m1
  __device__ __forceinline__ void place_and_shift(int k, LaneMask m, LaneMask m1, u32x2 czi, u32x2 cpa, u32x2 cpb) {
    LaneMask saved;
    if constexpr (NP == 0) {
      asm volatile(
          "s_and_saveexec_b64 %[sv], %[m]\n\t"
          "v_pk_mov_b32 %[z], %[cz], %[cz] op_sel:[0,1]\n\t"
          "s_and_b64 exec, exec, %[m1]\n\t"
          "v_pk_mov_b32 %[z], %[pz], %[pz] op_sel:[0,1]\n\t"
          "s_mov_b64 exec, %[sv]"
          : [z] "+v"(zi[k]), [sv] "=&s"(saved)
          : [m] "s"(m), [m1] "s"(m1), [cz] "v"(czi), [pz] "v"(zi[k - 1])
          : "scc");
      return;
    }
    asm volatile(
        "s_and_saveexec_b64 %[sv], %[m]\n\t"
        "v_pk_mov_b32 %[z], %[cz], %[cz] op_sel:[0,1]\n\t"
        "v_pk_mov_b32 %[a], %[ca], %[ca] op_sel:[0,1]\n\t"
        "v_pk_mov_b32 %[b], %[cb], %[cb] op_sel:[0,1]\n\t"
        "s_and_b64 exec, exec, %[m1]\n\t"
        "v_pk_mov_b32 %[z], %[pz], %[pz] op_sel:[0,1]\n\t"
        "v_pk_mov_b32 %[a], %[pa_], %[pa_] op_sel:[0,1]\n\t"
        "v_pk_mov_b32 %[b], %[pb_], %[pb_] op_sel:[0,1]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [z] "+v"(zi[k]), [a] "+v"(pa[k % KP]), [b] "+v"(pb[k % KP]), [sv] "=&s"(saved)
        : [m] "s"(m), [m1] "s"(m1), [cz] "v"(czi), [ca] "v"(cpa), [cb] "v"(cpb), [pz] "v"(zi[k - 1]), [pa_] "v"(pa[(k - 1) % KP]),
          [pb_] "v"(pb[(k - 1) % KP])
          : "scc");
  }
  __device__ __forceinline__ void place(int k, LaneMask m, u32x2 czi, u32x2 cpa, u32x2 cpb) {
    LaneMask saved;
    if constexpr (NP == 0) {
      asm volatile(
          "s_and_saveexec_b64 %[sv], %[m]\n\t"
          "v_pk_mov_b32 %[z], %[cz], %[cz] op_sel:[0,1]\n\t"
          "s_mov_b64 exec, %[sv]"
          : [z] "+v"(zi[k]), [sv] "=&s"(saved)
          : [m] "s"(m), [cz] "v"(czi)
          : "scc");
      return;
    }
    asm volatile(
        "s_and_saveexec_b64 %[sv], %[m]\n\t"
        "v_pk_mov_b32 %[z], %[cz], %[cz] op_sel:[0,1]\n\t"
        "v_pk_mov_b32 %[a], %[ca], %[ca] op_sel:[0,1]\n\t"
        "v_pk_mov_b32 %[b], %[cb], %[cb] op_sel:[0,1]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [z] "+v"(zi[k]), [a] "+v"(pa[k % KP]), [b] "+v"(pb[k % KP]), [sv] "=&s"(saved)
        : [m] "s"(m), [cz] "v"(czi), [ca] "v"(cpa), [cb] "v"(cpb)
          : "scc");
  }
#else
  typedef bool LaneMask;
  LaneMask sorts_before(float cz, int cidx, int k) const {
    if (KEY64) return key_of(mk_entry(cz, cidx)) < key_of(zi[k]);
    return (cz < zf(k)) | ((cz == zf(k)) & (cidx < ix(k)));
  }
  bool any_holds(int k) const { return ix(k) != kEmptyIdx; }
  void place_and_shift(int k, LaneMask m, LaneMask m1, u32x2 czi, u32x2 cpa, u32x2 cpb) {
    if (m) {
      zi[k] = czi;
      if (NP == 4) {
        pa[k % KP] = cpa;
        pb[k % KP] = cpb;
      }
    }
    if (m1) {
      zi[k] = zi[k - 1];
      if (NP == 4) {
        pa[k % KP] = pa[(k - 1) % KP];
        pb[k % KP] = pb[(k - 1) % KP];
      }
    }
  }
  void place(int k, LaneMask m, u32x2 czi, u32x2 cpa, u32x2 cpb) {
    if (m) {
      zi[k] = czi;
      if (NP == 4) {
        pa[k % KP] = cpa;
        pb[k % KP] = cpb;
      }
    }
  }
#endif

  // K <= KT live entries, K uniform over the wave (the exact-K kernels pass the constant KT and every test on K folds away).
  // Entries K .. KT-1 are never written: their steps of the network are skipped by scalar branches, so a queue of
  // capacity KT serves every K below it at the cost of K entries -- the live capacity must be exactly K, not KT, because
  // the clipped-neighbour rule looks the other half of a split face up in the CURRENT K nearest (rasterize_meshes.cu:186-215).
  P3D_HDM void insert(int K, float cz, int cidx, const float (&cpl)[NPS]) {
    const u32x2 czi = mk_entry(cz, cidx);
    const u32x2 cpa = NP == 4 ? mk_pair(f32_bits(cpl[0]), f32_bits(cpl[1 % NPS])) : czi;
    const u32x2 cpb = NP == 4 ? mk_pair(f32_bits(cpl[2 % NPS]), f32_bits(cpl[3 % NPS])) : czi;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (KEY64 && NP == 4 && (KT == 8 || KT == 4 || KT == 2)) {
      // the exact-K queues of the perspective + clip kernels: the whole network as one asm statement, its lane masks in the exec
      // register (topk_insert_asm.h)
      if (__builtin_constant_p(K) && K == KT) {
        const unsigned long long ek = ((unsigned long long)f32_bits(INFINITY) << 32) | (unsigned)kEmptyIdx;  // the key of an empty entry
        unsigned long long sv;
        if constexpr (KT == 8) P3D_TOPK_INSERT_ASM_8(zi, pa, pb, czi, cpa, cpb, ek, sv);
        if constexpr (KT == 4) P3D_TOPK_INSERT_ASM_4(zi, pa, pb, czi, cpa, cpb, ek, sv);
        if constexpr (KT == 2) P3D_TOPK_INSERT_ASM_2(zi, pa, pb, czi, cpa, cpb, ek, sv);
        kz = zf(KT - 1);
        ki = ix(KT - 1);
        return;
      }
    }
    if constexpr (KT % 2 == 0 && KT >= 4 && KT <= 8) {
      if (__builtin_constant_p(K) && K == KT) {
        insert_segments(czi, cpa, cpb, cz, cidx);
        return;
      }
    }
#endif
    LaneMask mk = 0;
#pragma unroll
    for (int k = KT - 1; k >= 1; --k) {
      // (a runtime K is re-read through an empty asm per step: left alone, the compiler evaluates all 2 KT tests on K ahead of
      // the loop nest as 64-bit masks, spills them into VGPR lanes and fetches them back with two v_readlane per step)
      const int Kq = opaque_uniform(K);
      if (k < Kq) {  // uniform
        if (k == Kq - 1) mk = sorts_before(cz, cidx, k);     // the first live step has no predecessor that computed its mask
        const LaneMask mk1 = sorts_before(cz, cidx, k - 1);  // reads entry k-1 before it can change
        place_and_shift(k, mk, mk1, czi, cpa, cpb);
        mk = mk1;
        if (k == Kq - 1) {
          kz = zf(k);
          ki = ix(k);
        }
      }
    }
    if (K == 1) mk = sorts_before(cz, cidx, 0);
    place(0, mk, czi, cpa, cpb);
    if (K == 1) {
      kz = zf(0);
      ki = ix(0);
    }
  }

#if defined(__HIP_DEVICE_COMPILE__)
  // The exact-K network (K == KT) in SEGMENTS of two steps, entered and left where it can change something (round 6).
  // A step costs its nine instructions whether or not a lane's mask is set, the network is a quarter of the fine kernel's
  // candidate pass, and at the bench workload 57 % of its steps change nothing in any lane (profiles/r06/probe_insert.txt):
  //   * top: entries fill from the front, so while NO inserting lane holds entry k - 1, entry k stays empty in all of them
  //     (the candidate lands at k - 1 or before): the segment {k + 1, k} is skipped when nobody holds entry k - 1;
  //   * bottom: lt[k] is monotone in k, so once no lane's candidate sorts before entry k - 1 nothing below k changes.
  // The kernel's time follows its instruction COUNT, scalar ones included (a version with both tests at every step ran 57 %
  // fewer steps and 3 % slower, profiles/r06/): one test per segment and side, nothing per step.
  // The K-th entry (kz, ki) changes only if the top step ran.
  static constexpr int SL = 2;  // steps per segment (1: the tests cost what the steps save; 4: the same time as 2, profiles/r06/README.md)
  // segment S holds steps S * SL + SL - 1 ... S * SL (step 0 = the placement into entry 0)
  template <int S>
  __device__ __forceinline__ LaneMask enter_segments(float cz, int cidx, int* seg) {
    if constexpr (S > 0) {
      if (!any_holds(S * SL - 1)) return enter_segments<S - 1>(cz, cidx, seg);  // uniform: nobody reaches entry S * SL -- this segment's steps change nothing
    }
    *seg = S;
    return sorts_before(cz, cidx, S * SL + SL - 1);
  }
  __device__ __forceinline__ void insert_segments(u32x2 czi, u32x2 cpa, u32x2 cpb, float cz, int cidx) {
    static_assert(KT % SL == 0, "whole segments");
    constexpr int NS = KT / SL;
    int seg = 0;
    LaneMask mk = enter_segments<NS - 1>(cz, cidx, &seg);
    // one chain of segments, entered at `seg` (no copies of the chain: with one copy per entry point the register allocator shuffles the queue)
#pragma unroll
    for (int S = NS - 1; S >= 0; --S) {
      if (S <= seg) {  // uniform
#pragma unroll
        for (int k = S * SL + SL - 1; k >= S * SL; --k) {
          if (k > 0) {
            const LaneMask m1 = sorts_before(cz, cidx, k - 1);
            place_and_shift(k, mk, m1, czi, cpa, cpb);
            if (k == KT - 1) {
              kz = zf(k);
              ki = ix(k);
            }
            mk = m1;
          } else {
            place(0, mk, czi, cpa, cpb);
          }
        }
        if (S > 0 && mk == 0) break;  // uniform
      }
    }
  }
#endif

  P3D_HDM int find(int want) const {
    int at = -1;
#pragma unroll
    for (int k = KT - 1; k >= 0; --k) {
      if (ix(k) == want) at = k;
    }
    return at;
  }

  P3D_HDM float payload_at(int p, int at) const {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k == at) v = pay(p, k);
    }
    return v;
  }

  // Remove entry `at` (keeps order).
  P3D_HDM void erase(int at) {
#pragma unroll
    for (int k = 0; k < KT - 1; ++k) {
      if (k >= at) {
        zi[k] = zi[k + 1];
        if (NP == 4) {
          pa[k % KP] = pa[(k + 1) % KP];
          pb[k % KP] = pb[(k + 1) % KP];
        }
      }
    }
    zi[KT - 1] = mk_entry(INFINITY, kEmptyIdx);
    kz = INFINITY;
    ki = kEmptyIdx;
  }
};

// the queue a kernel with compile-time perspective-correct + clipped barycentrics may use in place of Q.
// Why the 64-bit key order is safe there without canonicalising -0.0: the depth of a sample is
// bc.x * z0 + bc.y * z1 + bc.z * z2 with clipped barycentrics bc >= +0 (max(b, 0) / s, s >= 1e-5 > 0) and vertex depths
// z >= 1e-8: a face with a vertex depth below 1e-8 -- zero and negative zero included -- never reaches a queue
// (p3d_geom.h: face_setup, `z_invalid`, the reference's rasterize_meshes.cu:147 `zmin < kEpsilon`).  Products of a value
// >= +0 with a positive one are >= +0, and so is their sum in round-to-nearest: the bit pattern of z orders like z.
// (The point rasterizer, whose depths may be exactly zero of either sign, adds +0.0f when it stages them.)
template <typename Q>
struct PcQueue {
  typedef Q type;
};
template <int KT>
struct PcQueue<TopKPairs<KT, false, 4>> {
  typedef TopKPairs<KT, true> type;
};
template <int KT>
struct PcQueue<TopKPairs<KT, false, 0>> {
  typedef TopKPairs<KT, true, 0> type;
};

}  // namespace p3d
