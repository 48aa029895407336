/*
 * exp_lds_merge_tree.hip -- PROBE, not product.
 *
 * Question: would the N-way tile kernel (k_nway_merge, genometester4_amd/csrc/gt4hip_nway_tile.h) be cheaper if a tile's
 * records were RANKED BY MERGING instead of by interpolated buckets?  A tile holds <= 8 sorted runs in LDS; three
 * levels of pairwise merges order them: at every level a thread owns four consecutive output positions of one pair,
 * finds its merge-path split (a branch-free binary search over the cross diagonal: two 8-byte LDS reads, five VALU
 * per step), loads four candidates of either run, and takes the four smallest through a register network (four
 * minima + a bitonic merge of four: 32 VALU), written back with two 16-byte LDS stores.  Keys travel as 64-bit
 * COMPOSITES (key - tile's smallest key) << 12 | position, so the payload costs nothing and equal keys end up adjacent
 * (lower position first).  Behind the third level the (optional) fold: counts of equal neighbours are summed by a
 * segmented scan (thread-serial + DPP across the wavefront + a look-back over the previous wavefront's last seven
 * records), the LAST record of a group applies the cutoff and is staged (ballot-free: prefix of kept per thread,
 * DPP scan, wave totals through LDS).
 *
 * What is NOT in it (all of it would come on top in the product): fetch from HBM with its descriptors and slot
 * tables, the chained scan of tile totals, the write-out of the staged tile, tickets, count tables.  The tiles come
 * from a small pool in global memory (L2-resident), already laid out in the kernel's position space.
 *
 * Reported: microseconds and clock cycles per tile (persistent workgroups, one per CU, HIP events), for the copy
 * phase alone (LEVELS = 0), the three levels (LEVELS = 3) and levels + fold (FOLD); exactness against the host.
 * PMC: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS ... -- ./exp_lds_merge_tree <dist> pmc
 * (one launch per variant; counts / (workgroups x 16 x tiles per workgroup) = per wavefront and tile).
 *
 * Gate (VERDICT): <= 300 VALU per wavefront and tile, or <= 10 k cycles per tile -- against k_nway_merge's
 * 449 VALU / 128 SALU / 74 LDS and ~13.6 k cycles (profiles/round5/r5_nsub_ab.log).
 *
 * build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_lds_merge_tree.hip -o tools/exp/exp_lds_merge_tree
 * run:   tools/exp/exp_lds_merge_tree [stride|iid|genomic|clustered] [iters per workgroup] [pmc]
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__ ((ext_vector_type (4)));

constexpr int WAVE = 64;
constexpr int NT = 1024;
constexpr int NW = NT / WAVE;
constexpr int VT = 4;
constexpr int NRUN = 8;
constexpr int GAP = 4;                       /* all-ones slots behind every run: candidates past a run's end read them */
constexpr int CAPQ = NT * VT;                /* positions incl. gaps */
constexpr int CAP = CAPQ - NRUN * GAP;       /* records (runs rounded up to 4) a tile may hold */
constexpr u64 INF = ~0ull;
constexpr int STAGE_MAX = CAP;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf (stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString (e_)); exit (2); } } while (0)

/* a tile as the kernel reads it: position space with gaps (run r starts at q[r], q[r] multiple of 4, GAP all-ones slots
 * behind it), p[r] = records in front of run r (gapless: the thread numbering), steps[l] = search steps of level l */
struct TileHdr {
  u32 q[NRUN + 1];
  u32 p[NRUN + 1];
  u32 steps[3];
  u32 n_real;
  u64 min_key;
};

typedef __attribute__ ((address_space (3))) u64 lds_u64;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ u64 lds_ld64 (u32 byte_off) { return *(lds_u64 *) byte_off; }
#pragma clang diagnostic pop
template <class T> __device__ __forceinline__ u32 lds_off (T *p) { return (u32) (uintptr_t) p; }

__device__ __forceinline__ u32 dpp_incl_scan (u32 v)
{
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x111, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x112, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x114, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x118, 0xf, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x142, 0xa, 0xf, false);
  v += (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x143, 0xc, 0xf, false);
  return v;
}

__device__ __forceinline__ void cex (u64 &a, u64 &b) /* a <= b afterwards */
{
  const bool s = b < a;
  const u64 lo = s ? b : a, hi = s ? a : b;
  a = lo;
  b = hi;
}

struct Split { u32 a_addr, b_addr; }; /* byte addresses of A[a0] and B[b0] */

/* One merge level.  L = 1, 2, 3: pairs of runs of 2^(L-1) original runs each.  x: input buffer (byte offset in LDS), the
 * thread's four outputs come back in m[]; returns false for threads behind the tile's last record (m = INF). */
template <int L>
__device__ __forceinline__ bool merge_level (const TileHdr *h, u32 x, int tid, u64 (&m)[VT], u32 &out_q, Split &sp)
{
  constexpr int RS = 1 << L, NP = NRUN >> L;
  const u32 d4 = (u32) tid * VT;
  /* the pair this thread works for (thread numbering: gapless positions) */
  u32 pi = 0;
#pragma unroll
  for (int q = 1; q < NP; q++) pi += d4 >= h->p[q * RS] ? 1u : 0u;
  const u32 r0 = pi * RS;
  const u32 pa = h->p[r0], pm = h->p[r0 + RS / 2], pe = h->p[r0 + RS];
  const u32 qa = h->q[r0], qm = h->q[r0 + RS / 2];
  const bool active = d4 < h->p[NRUN];
  const u32 d = active ? d4 - pa : 0u, nA = pm - pa, nB = pe - pm;
  const u32 lo = d > nB ? d - nB : 0u, hi = d < nA ? d : nA;
  const int n = (int) (hi - lo);
  const u32 steps = (u32) __builtin_amdgcn_readfirstlane ((int) h->steps[L - 1]);
  /* byte addresses: A[a] at A8 + 8a, B[b] at B8 + 8b; on the diagonal a + b = d - 1 the two addresses add up to K.
    * (a merged run of level L - 1 lies where its first run lay) */
  const u32 A8 = x + 8u * qa, B8 = x + 8u * qm;
  const int loA = (int) (A8 + 8u * lo);
  const int K = (int) (A8 + B8 + 8u * (d - 1u));
  /* lower bound of "A[a] < B[d-1-a] fails" over a in [lo, hi) by fixed strides P/2 .. 1 from the virtual start
   * hi - (P - 1): probes below lo read A[lo] instead (if that fails every probe fails and the clamp returns lo) */
  int r = loA + 8 * (n - (int) (1u << (steps - 1u)));
  {
    u32 vh = 8u << (steps - 1u);              /* twice the next stride in bytes (a vector register: v_cndmask takes no scalar beside its mask) */
    int sneg = -(int) (4u << (steps - 1u));
    for (u32 i = steps; i > 1u; i--) {
      const int rc = r > loA ? r : loA;
      const u64 ka = lds_ld64 ((u32) rc), kb = lds_ld64 ((u32) (K - rc));
      const u32 t = ka < kb ? vh : 0u;
      r = r + sneg + (int) t;
      vh >>= 1;
      sneg >>= 1;
    }
    const int rc = r > loA ? r : loA;
    const u64 ka = lds_ld64 ((u32) rc), kb = lds_ld64 ((u32) (K - rc));
    r += ka < kb ? 8 : 0;
  }
  const int hiA = loA + 8 * n;
  const int ra = r < loA ? loA : (r > hiA ? hiA : r);
  sp.a_addr = (u32) ra;
  sp.b_addr = (u32) (K + 8 - ra);
  u64 a[VT], b[VT];
#pragma unroll
  for (int i = 0; i < VT; i++) a[i] = lds_ld64 (sp.a_addr + 8u * i);
#pragma unroll
  for (int i = 0; i < VT; i++) b[i] = lds_ld64 (sp.b_addr + 8u * i);
  /* the four smallest of a[0..3] (ascending) and b[0..3] (ascending): min (a[i], b[3-i]) is a bitonic sequence of them */
#pragma unroll
  for (int i = 0; i < VT; i++) m[i] = a[i] < b[VT - 1 - i] ? a[i] : b[VT - 1 - i];
  cex (m[0], m[2]);
  cex (m[1], m[3]);
  cex (m[0], m[1]);
  cex (m[2], m[3]);
  out_q = qa + d;
  if (!active) {
#pragma unroll
    for (int i = 0; i < VT; i++) m[i] = INF;
  }
  return active;
}

template <int L>
__device__ __forceinline__ void store_level (const TileHdr *h, u32 y, int tid, const u64 (&m)[VT], u32 out_q, bool active)
{
  if (active) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
    __attribute__ ((address_space (3))) u32x4 *o = (__attribute__ ((address_space (3))) u32x4 *) (y + 8u * out_q);
#pragma clang diagnostic pop
    u32x4 v0 = { (u32) m[0], (u32) (m[0] >> 32), (u32) m[1], (u32) (m[1] >> 32) };
    u32x4 v1 = { (u32) m[2], (u32) (m[2] >> 32), (u32) m[3], (u32) (m[3] >> 32) };
    o[0] = v0;
    o[1] = v1;
    /* the thread that writes a merged run's last records also writes the all-ones gap behind it */
    constexpr int RS = 1 << L, NP = NRUN >> L;
    bool last = false;
#pragma unroll
    for (int q = 1; q <= NP; q++) last |= (u32) tid * VT + VT == h->p[q * RS];
    if (last) {
      u32x4 ones = { ~0u, ~0u, ~0u, ~0u };
      o[2] = ones;
      o[3] = ones;
    }
  }
}

struct Shared {
  alignas (16) u64 pad0[8];
  alignas (16) u64 buf0[CAPQ + 8];
  alignas (16) u64 buf1[CAPQ + 8];
  alignas (16) u32 cnt[CAPQ];
  alignas (16) u32 stage[3 * STAGE_MAX + 16];
  TileHdr hdr;
  u32 wkept[NW];
};

/* LEVELS: 0 copy only, 3 the merge tree; FOLD: counts folded, kept records staged; VERIFY: results to global memory */
template <int LEVELS, bool FOLD, bool VERIFY>
__global__ __launch_bounds__ (NT, 4) void k_tree (const TileHdr *__restrict__ hdrs, const u64 *__restrict__ keys, const u32 *__restrict__ cnts,
                                                  u32 n_pool, u32 iters, u32 cutoff, u64 *__restrict__ out_sorted, u32 *__restrict__ out_recs,
                                                  u32 *__restrict__ out_n, u64 *__restrict__ sink)
{
  __shared__ Shared sh;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane (tid / WAVE);
  const u32 b0 = lds_off (&sh.buf0[0]), b1 = lds_off (&sh.buf1[0]);
  u64 acc = 0;
  u64 pk[VT];
  u32 pc[VT];
  auto fetch = [&] (u32 tile) {
#pragma unroll
    for (int j = 0; j < VT; j++) {
      pk[j] = __builtin_nontemporal_load (&keys[(u64) tile * CAPQ + (u32) (j * NT + tid)]);
      pc[j] = __builtin_nontemporal_load (&cnts[(u64) tile * CAPQ + (u32) (j * NT + tid)]);
    }
  };
  u32 tile = blockIdx.x % n_pool;
  fetch (tile);
  for (u32 it = 0; it < iters; it++) {
    /* ---- phase 0: records -> composites (position space = the pool's layout) */
    if (tid < (int) (sizeof (TileHdr) / 4)) ((u32 *) &sh.hdr)[tid] = ((const u32 *) &hdrs[tile])[tid];
    const u64 mk = hdrs[tile].min_key;
#pragma unroll
    for (int j = 0; j < VT; j++) {
      const u32 pos = (u32) (j * NT + tid);
      const u64 c = pk[j] == INF ? INF : (((pk[j] - mk) << 12) | pos);
      sh.buf0[pos] = c;
      sh.cnt[pos] = pc[j];
    }
    const u32 this_tile = tile;
    tile = (tile + gridDim.x) % n_pool;
    fetch (tile);
    __syncthreads ();
    u64 m[VT];
    u32 oq = 0;
    Split sp = { 0, 0 };
    bool active = true;
    if (LEVELS == 3) {
      active = merge_level<1> (&sh.hdr, b0, tid, m, oq, sp);
      store_level<1> (&sh.hdr, b1, tid, m, oq, active);
      __syncthreads ();
      active = merge_level<2> (&sh.hdr, b1, tid, m, oq, sp);
      store_level<2> (&sh.hdr, b0, tid, m, oq, active);
      __syncthreads ();
      active = merge_level<3> (&sh.hdr, b0, tid, m, oq, sp);
    } else {
#pragma unroll
      for (int j = 0; j < VT; j++) m[j] = sh.buf0[tid * VT + j];
    }
    if (!FOLD) {
      if (VERIFY) {
        if (active && LEVELS == 3)
          for (int j = 0; j < VT; j++) out_sorted[(u64) this_tile * CAPQ + (u32) (tid * VT + j)] = m[j];
      } else {
#pragma unroll
        for (int j = 0; j < VT; j++) acc += m[j];
      }
    } else {
      /* ---- fold: key = composite >> 12, position = low 12 bits; groups of equal keys are adjacent */
      u64 k[VT];
      u32 c[VT];
      bool valid[VT];
#pragma unroll
      for (int j = 0; j < VT; j++) {
        valid[j] = m[j] != INF;
        k[j] = m[j] >> 12;
        c[j] = valid[j] ? sh.cnt[(u32) m[j] & 4095u] : 0u;
      }
      /* the key in front of the thread's first: the previous lane's last (lane 0: the larger of A[a0-1], B[b0-1]) */
      u64 prev;
      {
        const u32 plo = (u32) __builtin_amdgcn_update_dpp (0, (int) (u32) k[VT - 1], 0x138, 0xf, 0xf, false);        /* wave_shr:1 */
        const u32 phi = (u32) __builtin_amdgcn_update_dpp (0, (int) (u32) (k[VT - 1] >> 32), 0x138, 0xf, 0xf, false);
        prev = (u64) plo | ((u64) phi << 32);
      }
      /* look-back of the wavefront (lanes 0..15): the <= 7 records in front of the wavefront's first in either input run
       * of level 3 that carry the same key -- their counts are the carry into lane 0's first group */
      const u32 wa = (u32) __builtin_amdgcn_readfirstlane ((int) sp.a_addr), wb = (u32) __builtin_amdgcn_readfirstlane ((int) sp.b_addr);
      const u64 k0 = (u64) (u32) __builtin_amdgcn_readfirstlane ((int) (u32) k[0]) | ((u64) (u32) __builtin_amdgcn_readfirstlane ((int) (u32) (k[0] >> 32)) << 32);
      u32 carry_in = 0;
      bool first_is_head = true;
      {
        const u32 a_first = b0 + 8u * sh.hdr.q[0], b_first = b0 + 8u * sh.hdr.q[4];
        const u32 back = 8u * (1u + ((u32) lane & 7u));
        const bool is_b = (lane & 8) != 0;
        const u32 base = is_b ? wb : wa, first = is_b ? b_first : a_first;
        u32 cc = 0;
        if (lane < 16 && base >= first + back) {
          const u64 v = lds_ld64 (base - back);
          if ((v >> 12) == k0) cc = sh.cnt[(u32) v & 4095u] | 0x80000000u; /* (bit 31: a predecessor with the same key exists; counts < 2^31 in the probe) */
        }
        u32 any = cc >> 31;
        cc &= 0x7fffffffu;
        /* sum over the row of 16 */
        cc += (u32) __builtin_amdgcn_update_dpp (0, (int) cc, 0x111, 0xf, 0xf, false);
        cc += (u32) __builtin_amdgcn_update_dpp (0, (int) cc, 0x112, 0xf, 0xf, false);
        cc += (u32) __builtin_amdgcn_update_dpp (0, (int) cc, 0x114, 0xf, 0xf, false);
        cc += (u32) __builtin_amdgcn_update_dpp (0, (int) cc, 0x118, 0xf, 0xf, false);
        any |= (u32) __builtin_amdgcn_update_dpp (0, (int) any, 0x111, 0xf, 0xf, false);
        any |= (u32) __builtin_amdgcn_update_dpp (0, (int) any, 0x112, 0xf, 0xf, false);
        any |= (u32) __builtin_amdgcn_update_dpp (0, (int) any, 0x114, 0xf, 0xf, false);
        any |= (u32) __builtin_amdgcn_update_dpp (0, (int) any, 0x118, 0xf, 0xf, false);
        carry_in = (u32) __builtin_amdgcn_readlane ((int) cc, 15);
        first_is_head = __builtin_amdgcn_readlane ((int) any, 15) == 0;
      }
      bool head[VT];
      head[0] = lane == 0 ? first_is_head : k[0] != prev;
#pragma unroll
      for (int j = 1; j < VT; j++) head[j] = k[j] != k[j - 1];
      /* thread-serial inclusive segmented sums, then the segmented scan of (any head, open sum) across the lanes */
      u32 s[VT];
      s[0] = c[0];
#pragma unroll
      for (int j = 1; j < VT; j++) s[j] = head[j] ? c[j] : s[j - 1] + c[j];
      u32 F = (head[0] | head[1] | head[2] | head[3]) ? 1u : 0u, V = s[VT - 1];
#define SEG_STEP(ctrl, row_mask) do { \
        const u32 v2 = (u32) __builtin_amdgcn_update_dpp (0, (int) V, ctrl, row_mask, 0xf, false); \
        const u32 f2 = (u32) __builtin_amdgcn_update_dpp (0, (int) F, ctrl, row_mask, 0xf, false); \
        V = F ? V : V + v2; \
        F |= f2; \
      } while (0)
      SEG_STEP (0x111, 0xf);
      SEG_STEP (0x112, 0xf);
      SEG_STEP (0x114, 0xf);
      SEG_STEP (0x118, 0xf);
      SEG_STEP (0x142, 0xa);
      SEG_STEP (0x143, 0xc);
      u32 cin = (u32) __builtin_amdgcn_update_dpp (0, (int) V, 0x138, 0xf, 0xf, false); /* the open sum in front of this thread */
      const u32 fin = (u32) __builtin_amdgcn_update_dpp (0, (int) F, 0x138, 0xf, 0xf, false);
      cin = lane == 0 ? carry_in : (fin ? cin : cin + carry_in);
      /* (lanes whose predecessors hold no head at all continue the look-back's group: cin + carry_in above) */
      bool open = true;
#pragma unroll
      for (int j = 0; j < VT; j++) {
        open = open && !head[j];
        s[j] += open ? cin : 0u;
      }
      /* the last record of a group decides: the next key differs (the next lane's first head flag; lane 63: the smaller
       * of the two records behind the wavefront's split) */
      bool tail[VT];
#pragma unroll
      for (int j = 0; j + 1 < VT; j++) tail[j] = head[j + 1];
      {
        const u32 nh = (u32) __builtin_amdgcn_update_dpp (0, (int) (head[0] ? 1u : 0u), 0x130, 0xf, 0xf, false); /* wave_shl:1 */
        /* records of A taken by this thread = those of its four that came from A: composites of A are those below B's first position */
        u32 ta = 0;
        const u32 bpos0 = sh.hdr.q[4];
#pragma unroll
        for (int j = 0; j < VT; j++) ta += ((u32) m[j] & 4095u) < bpos0 && valid[j] ? 1u : 0u;
        tail[VT - 1] = nh != 0;
        if (lane == WAVE - 1) {
          const u64 na = lds_ld64 (sp.a_addr + 8u * ta), nb = lds_ld64 (sp.b_addr + 8u * (VT - ta));
          const u64 nx = na < nb ? na : nb;
          tail[VT - 1] = (nx >> 12) != k[VT - 1];
        }
      }
      u32 keep[VT], nk = 0;
#pragma unroll
      for (int j = 0; j < VT; j++) {
        keep[j] = valid[j] && tail[j] && s[j] >= cutoff ? 1u : 0u;
        nk += keep[j];
      }
      const u32 incl = dpp_incl_scan (nk);
      if (lane == WAVE - 1) sh.wkept[wid] = incl;
      __syncthreads ();
      u32 before = incl - nk, total = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const u32 t = sh.wkept[w];
        before += w < wid ? t : 0u;
        total += t;
      }
#pragma unroll
      for (int j = 0; j < VT; j++) {
        if (keep[j]) {
          const u64 key = k[j] + mk;
          u32 *o = &sh.stage[3 * before];
          o[0] = (u32) key;
          o[1] = (u32) (key >> 32);
          o[2] = s[j];
          before++;
        }
      }
      if (VERIFY) {
        __syncthreads ();
        for (u32 i = (u32) tid; i < 3 * total; i += NT) out_recs[(u64) this_tile * 3 * STAGE_MAX + i] = sh.stage[i];
        if (tid == 0) out_n[this_tile] = total;
      } else {
        acc += total;
      }
    }
    __syncthreads ();
  }
  if (!VERIFY) {
    acc += sh.stage[tid]; /* (keeps the staging stores alive) */
    if (acc == 0x123456789abcdefull) sink[0] = acc;
  }
}

/* ------------------------------------------------------------------ host */

struct Pool {
  std::vector<TileHdr> hdr;
  std::vector<u64> keys;
  std::vector<u32> cnts;
  std::vector<std::vector<std::pair<u64, u32>>> runs; /* per tile x run: (key, count) */
};

static u32 bitlen (u32 v) { u32 b = 0; while (v) { b++; v >>= 1; } return b; }

static Pool make_pool (const std::string &dist, u32 n_pool, u32 target, u64 seed)
{
  Pool P;
  std::mt19937_64 rng (seed);
  P.hdr.resize (n_pool);
  P.keys.assign ((size_t) n_pool * CAPQ, INF);
  P.cnts.assign ((size_t) n_pool * CAPQ, 0);
  P.runs.resize ((size_t) n_pool * NRUN);
  for (u32 t = 0; t < n_pool; t++) {
    /* distinct keys of the tile, then membership per list */
    std::vector<u64> uni;
    std::vector<std::vector<std::pair<u64, u32>>> run (NRUN);
    const u64 base = (rng () >> 16) & ~0xfffffffffull;
    u32 total = 0;
    auto member = [&] (u64 key, u32 mask) {
      for (int r = 0; r < NRUN; r++)
        if (mask >> r & 1) {
          run[r].push_back ({ key, 1u + (u32) (rng () % 8) });
          total++;
        }
    };
    u64 key = base;
    while (total + NRUN <= target) {
      u32 mask;
      if (dist == "iid") {
        key += 1 + rng () % (1u << 19);
        mask = 1u << (rng () % NRUN);
      } else if (dist == "stride") {
        /* the bench's lists: even lists hold the same keys, odd lists keys of their own (1.6 records per distinct key) */
        key += 1 + rng () % (1u << 19);
        mask = (rng () % 5 == 0) ? 0x55u : 1u << (1 + 2 * (rng () % 4));
      } else if (dist == "genomic") {
        key += 1 + rng () % (1u << 19);
        const u32 r = (u32) (rng () % 100);
        mask = r < 70 ? 0xffu : (r < 85 ? (u32) (rng () % 255 + 1) : 1u << (rng () % NRUN));
      } else { /* clustered: stretches of nearly adjacent keys, far apart */
        key += (rng () % 3000 == 0) ? (1ull << 40) + rng () % (1ull << 30) : 1 + rng () % 3;
        mask = 1u << (rng () % NRUN);
        if (rng () % 4 == 0) mask |= 1u << (rng () % NRUN);
      }
      member (key, mask);
    }
    TileHdr &h = P.hdr[t];
    memset (&h, 0, sizeof h);
    u32 q = 0, p = 0;
    u64 mn = INF;
    for (int r = 0; r < NRUN; r++) {
      h.q[r] = q;
      h.p[r] = p;
      for (size_t i = 0; i < run[r].size (); i++) {
        P.keys[(size_t) t * CAPQ + q + i] = run[r][i].first;
        P.cnts[(size_t) t * CAPQ + q + i] = run[r][i].second;
        mn = std::min (mn, run[r][i].first);
      }
      const u32 len4 = ((u32) run[r].size () + 3u) & ~3u;
      q += len4 + GAP;
      p += len4;
      P.runs[(size_t) t * NRUN + r] = run[r];
    }
    h.q[NRUN] = q;
    h.p[NRUN] = p;
    if (q > (u32) CAPQ) { fprintf (stderr, "tile overflows\n"); exit (2); }
    h.n_real = total;
    h.min_key = mn;
    for (int L = 1; L <= 3; L++) {
      const int RS = 1 << L;
      u32 mx = 1;
      for (int pi = 0; pi < NRUN / RS; pi++) {
        const u32 nA = h.p[pi * RS + RS / 2] - h.p[pi * RS], nB = h.p[(pi + 1) * RS] - h.p[pi * RS + RS / 2];
        mx = std::max (mx, std::min (nA, nB));
      }
      h.steps[L - 1] = std::max (1u, bitlen (mx));
    }
  }
  return P;
}

int main (int argc, char **argv)
{
  const std::string dist = argc > 1 ? argv[1] : "stride";
  const u32 iters = argc > 2 ? (u32) atoi (argv[2]) : 2000u;
  const bool pmc = argc > 3 && !strcmp (argv[3], "pmc");
  const u32 target = argc > 4 ? (u32) atoi (argv[4]) : 3500u;
  const u32 n_pool = 512;
  hipDeviceProp_t prop;
  CHECK (hipGetDeviceProperties (&prop, 0));
  const u32 grid = (u32) prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  Pool P = make_pool (dist, n_pool, target, 12345);
  double avg_real = 0;
  for (auto &h : P.hdr) avg_real += h.n_real;
  avg_real /= n_pool;
  TileHdr *d_hdr;
  u64 *d_keys, *d_sorted, *d_sink;
  u32 *d_cnts, *d_recs, *d_n;
  CHECK (hipMalloc (&d_hdr, sizeof (TileHdr) * n_pool));
  CHECK (hipMalloc (&d_keys, 8ull * n_pool * CAPQ));
  CHECK (hipMalloc (&d_cnts, 4ull * n_pool * CAPQ));
  CHECK (hipMalloc (&d_sorted, 8ull * n_pool * CAPQ));
  CHECK (hipMalloc (&d_recs, 12ull * n_pool * STAGE_MAX));
  CHECK (hipMalloc (&d_n, 4ull * n_pool));
  CHECK (hipMalloc (&d_sink, 8));
  CHECK (hipMemcpy (d_hdr, P.hdr.data (), sizeof (TileHdr) * n_pool, hipMemcpyHostToDevice));
  CHECK (hipMemcpy (d_keys, P.keys.data (), 8ull * n_pool * CAPQ, hipMemcpyHostToDevice));
  CHECK (hipMemcpy (d_cnts, P.cnts.data (), 4ull * n_pool * CAPQ, hipMemcpyHostToDevice));
  CHECK (hipMemset (d_sorted, 0xff, 8ull * n_pool * CAPQ));
  const u32 cutoff = 1;

  /* ---- exactness: every pool tile once (grid = pool, one iteration) */
  int bad = 0;
  if (!pmc) {
    hipLaunchKernelGGL ((k_tree<3, false, true>), dim3 (n_pool), dim3 (NT), 0, 0, d_hdr, d_keys, d_cnts, n_pool, 1u, cutoff, d_sorted, d_recs, d_n, d_sink);
    CHECK (hipDeviceSynchronize ());
    std::vector<u64> got ((size_t) n_pool * CAPQ);
    CHECK (hipMemcpy (got.data (), d_sorted, 8ull * n_pool * CAPQ, hipMemcpyDeviceToHost));
    for (u32 t = 0; t < n_pool && bad < 5; t++) {
      std::vector<u64> want;
      const TileHdr &h = P.hdr[t];
      for (int r = 0; r < NRUN; r++)
        for (size_t i = 0; i < P.runs[(size_t) t * NRUN + r].size (); i++) want.push_back (((P.runs[(size_t) t * NRUN + r][i].first - h.min_key) << 12) | (h.q[r] + (u32) i));
      std::sort (want.begin (), want.end ());
      for (size_t i = 0; i < want.size (); i++)
        if (got[(size_t) t * CAPQ + i] != want[i]) {
          fprintf (stderr, "levels: tile %u position %zu: got %llx want %llx\n", t, i, got[(size_t) t * CAPQ + i], want[i]);
          bad++;
          break;
        }
    }
    hipLaunchKernelGGL ((k_tree<3, true, true>), dim3 (n_pool), dim3 (NT), 0, 0, d_hdr, d_keys, d_cnts, n_pool, 1u, cutoff, d_sorted, d_recs, d_n, d_sink);
    CHECK (hipDeviceSynchronize ());
    std::vector<u32> recs ((size_t) n_pool * 3 * STAGE_MAX), nn (n_pool);
    CHECK (hipMemcpy (recs.data (), d_recs, 12ull * n_pool * STAGE_MAX, hipMemcpyDeviceToHost));
    CHECK (hipMemcpy (nn.data (), d_n, 4ull * n_pool, hipMemcpyDeviceToHost));
    for (u32 t = 0; t < n_pool && bad < 5; t++) {
      std::vector<std::pair<u64, u32>> all;
      for (int r = 0; r < NRUN; r++) all.insert (all.end (), P.runs[(size_t) t * NRUN + r].begin (), P.runs[(size_t) t * NRUN + r].end ());
      std::sort (all.begin (), all.end ());
      std::vector<std::pair<u64, u32>> want;
      for (auto &kc : all) {
        if (!want.empty () && want.back ().first == kc.first) want.back ().second += kc.second;
        else want.push_back (kc);
      }
      if (nn[t] != want.size ()) {
        fprintf (stderr, "fold: tile %u keeps %u records, want %zu\n", t, nn[t], want.size ());
        bad++;
        continue;
      }
      for (size_t i = 0; i < want.size (); i++) {
        const u32 *r = &recs[(size_t) t * 3 * STAGE_MAX + 3 * i];
        const u64 key = (u64) r[0] | ((u64) r[1] << 32);
        if (key != want[i].first || r[2] != want[i].second) {
          fprintf (stderr, "fold: tile %u record %zu: got (%llx, %u) want (%llx, %u)\n", t, i, key, r[2], want[i].first, want[i].second);
          bad++;
          break;
        }
      }
    }
    printf ("exact: %s (%u tiles, %.0f records per tile on average, dist %s)\n", bad ? "NO" : "yes", n_pool, avg_real, dist.c_str ());
  }

  /* ---- timing */
  hipEvent_t e0, e1;
  CHECK (hipEventCreate (&e0));
  CHECK (hipEventCreate (&e1));
  auto run = [&] (const char *name, auto kernel) {
    float best = 1e30f;
    const int reps = pmc ? 1 : 4;
    for (int rep = 0; rep < reps; rep++) {
      CHECK (hipEventRecord (e0, 0));
      hipLaunchKernelGGL (kernel, dim3 (grid), dim3 (NT), 0, 0, d_hdr, d_keys, d_cnts, n_pool, iters, cutoff, d_sorted, d_recs, d_n, d_sink);
      CHECK (hipEventRecord (e1, 0));
      CHECK (hipEventSynchronize (e1));
      float ms;
      CHECK (hipEventElapsedTime (&ms, e0, e1));
      best = std::min (best, ms);
    }
    const double us_tile = best * 1e3 / iters;
    printf ("%-14s %8.3f ms  %7.3f us per tile = %6.0f cycles at %.2f GHz  (%.2f cycles per record; %u workgroups x %u tiles)\n", name, best, us_tile,
            us_tile * ghz * 1e3, ghz, us_tile * ghz * 1e3 / avg_real, grid, iters);
  };
  run ("copy only", k_tree<0, false, false>);
  run ("three levels", k_tree<3, false, false>);
  run ("levels + fold", k_tree<3, true, false>);
  return bad ? 1 : 0;
}
