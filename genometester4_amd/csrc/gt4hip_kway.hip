/*
 * gt4hip_kway.hip -- N-way union of up to eight sorted lists in ONE pass over HBM.
 *
 * What it restates: union_multi (reference src/glistcompare.c:500-603; hot loop :545-591) and
 * gt4_write_union (src/set-operations.c:40-129): for every distinct key ascending, the count is
 * the sum / maximum / override over the lists that hold the key, kept iff count >= cutoff.
 *
 * Why a second kernel: the pairwise tree of k_pair_merge moves every record log2(N) times through
 * HBM (246 GB for eight 5e8-record lists whose algorithmic traffic is 78 GB).  Here a workgroup
 * owns one TILE of the merged key sequence -- a key range cut out of all the lists at once -- loads
 * the tile's up to eight sorted runs into LDS, merges them pairwise INSIDE LDS (three passes for
 * eight runs, each pass a merge-path split per thread followed by a short serial merge, records
 * ping-ponging between two LDS buffers), combines equal keys, compacts, and writes the tile out
 * once.  HBM sees every input record once and every output record once.
 *
 *   K5 k_kway_sample     every S-th key of every list -> "sample lists" (1/S of the data)
 *   K6 k_kway_partition  tile boundaries: the merged samples' every G-th key, located in every list
 *                        by binary search (all records with a key <= the boundary key go left);
 *      k_kway_check      no tile may exceed the LDS capacity (else the host takes the pairwise tree)
 *   K7 k_kway_merge      the tile merge; MODE_DUPS keeps every record (it is how the sample lists
 *                        themselves are merged, one level up: the recursion ends when a level fits
 *                        one tile), MODE_LOOKBACK / MODE_COUNT combine equal keys and apply the rule.
 *
 * Global output offsets come from the same chained scan as the pair kernel (scanner wavefront +
 * per-row carries, gt4hip_device.h).
 */
#include "gt4hip_device.h"
#include "gt4hip_host.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

namespace gt4 {

namespace {

constexpr int KWAY_MAX = 8;
constexpr int KWAY_SAMPLE = 128; /* S: one sample per 128 records (1.5 KB): the strided gather costs ~8 % of a streaming read of the inputs */

enum : int { KWAY_COUNT = 0, KWAY_UNION = 1, KWAY_DUPS = 2 };

struct KwayParams {
  const u32 *list[KWAY_MAX];
  u64 n[KWAY_MAX];
  u32 k;
  u32 rule;            /* 1 ADD, 4 MAX, 7 NUMBER */
  u32 cutoff;
  u32 count_override;
  u32 filter;          /* FILTER_RAW: keep every key; FILTER_RESULT: count >= cutoff */
  u32 spin_limit;
  u32 num_tiles;
  u32 vt_min;          /* positions per thread in a merge pass, at least (the split search is paid per thread) */
};

/* ------------------------------------------------------------------ K5 / K6: samples and tile boundaries */

__global__ void k_kway_sample (const u32 *__restrict__ list, u64 n_samples, u32 *__restrict__ out)
{
  const u64 step = (u64) gridDim.x * blockDim.x;
  for (u64 j = (u64) blockIdx.x * blockDim.x + threadIdx.x; j < n_samples; j += step) {
    const u64 src = (j + 1) * KWAY_SAMPLE - 1; /* the last key of every full block of S records */
    out[3 * j] = list[3 * src];
    out[3 * j + 1] = list[3 * src + 1];
    out[3 * j + 2] = 0;
  }
}

/* part[t][i] = first record of list i that belongs to tile t or a later one.  Tile t > 0 starts
 * behind the boundary key x_t = merged_samples[t * G - 1] (the last boundary is the very last
 * sample, so that the final tile holds only the lists' tails behind their last samples):
 * records with key <= x_t belong to earlier tiles (upper bound), equal keys of different lists
 * therefore always meet in one tile. */
__global__ void k_kway_partition (KwayParams p, const u32 *__restrict__ merged, u64 m_total, u32 G, u64 *__restrict__ part)
{
  const u64 id = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  const u64 t = id / KWAY_MAX;
  const u32 i = (u32) (id % KWAY_MAX);
  if (t > p.num_tiles) return;
  u64 v;
  if (i >= p.k || t == 0) {
    v = 0;
  } else if (t == p.num_tiles) {
    v = p.n[i];
  } else {
    const u64 sidx = (t == p.num_tiles - 1) ? m_total - 1 : t * (u64) G - 1;
    const u64 x = load_key (merged, sidx);
    const u32 *__restrict__ L = p.list[i];
    u64 lo = 0, hi = p.n[i];
    while (lo < hi) {
      const u64 mid = (lo + hi) >> 1;
      if (load_key (L, mid) <= x) lo = mid + 1;
      else hi = mid;
    }
    v = lo;
  }
  part[t * KWAY_MAX + i] = v;
}

__global__ void k_kway_check (const u64 *__restrict__ part, u32 num_tiles, u32 cap, u32 *flag)
{
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= num_tiles) return;
  u64 s = 0;
  bool mono = true;
  for (int i = 0; i < KWAY_MAX; i++) {
    const u64 a = part[t * KWAY_MAX + i], b = part[(t + 1) * KWAY_MAX + i];
    mono &= b >= a;
    s += b - a;
  }
  if (s > cap || !mono) atomicOr (flag, 1u);
}

/* ------------------------------------------------------------------ K7: tile merge in LDS */

typedef u32 u32x3 __attribute__ ((ext_vector_type (3)));

template <int NT, int CAP>
struct KwayShared {
  /* two record buffers for the merge passes: 16 bytes per record {key lo, key hi, count, -}, so that
   * a record is one aligned ds_read_b128 / ds_write_b128 and a key one ds_read_b64 */
  alignas (16) u32x4 rec[2][CAP + 2]; /* (+2: the merge steps read up to two records ahead) */
  /* the tile's kept records, packed 12-byte as in the output list; written out during the NEXT tile */
  alignas (16) u32 stage[3 * CAP + 4];
  /* run table of the tile being loaded / merged, two deep */
  u64 tab_start[2][KWAY_MAX];  /* first record of the run in its list */
  u32 tab_len[2][KWAY_MAX];    /* records */
  u32 tab_total[2];
  u64 tab_base[2];             /* sum of the first records (MODE_DUPS: where the tile's output starts) */
  u64 rng[3][2][KWAY_MAX];     /* part[] entries of the next tiles, three deep */
  u32 wave_tot[WAVE];         /* kept records per (row of NT positions, wavefront) */
  u64 excl;
  u32 tick[2];
};

/* P = number of pairwise passes = log2 (run slots): 2 for three or four lists, 3 for five to eight */
template <int NT, int CAP, int MODE, int P>
__global__ __launch_bounds__ (NT, (NT / 256)) void
k_kway_merge (KwayParams p, const u64 *__restrict__ part, u32 *__restrict__ out, u64 *desc, PairControl *ctl)
{
  constexpr int NW = NT / WAVE;
  constexpr int SLOTS = 1 << P;
  constexpr int J = (CAP / WAVE + KWAY_MAX + NW - 1) / NW; /* 64-record wave slots a wavefront fetches per tile, at most */
  typedef KwayShared<NT, CAP> Shared;
  __shared__ Shared sh;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane (tid / WAVE);
  const u64 n_rows = ((u64) p.num_tiles + WAVE - 1) / WAVE;
  const u32 spin_limit = p.spin_limit ? p.spin_limit : SPIN_LIMIT;
  u32 *const agg = reinterpret_cast<u32 *> (desc);
  u64 *const carry = desc + 2 * n_rows * WAVE;

  u32 role = 0;
  if (MODE == KWAY_UNION) {
    if (tid == 0) sh.tick[0] = atomicAdd (&ctl->role, 1u);
    __syncthreads ();
    role = sh.tick[0];
    __syncthreads ();
    if (role == 0) {
      if (wid < 8) scanner_part (agg, carry + 4 * (n_rows + 1), carry, p.num_tiles, ctl, lane, spin_limit, (u32) wid, NW < 8 ? (u32) NW : 8u);
      return;
    }
  }
  const u32 n_workers = MODE == KWAY_UNION ? gridDim.x - 1 : gridDim.x;
  const u32 wk = MODE == KWAY_UNION ? role - 1 : blockIdx.x;
  const u64 ntl = p.num_tiles;
  auto tile_at = [&] (int j) -> u64 { return (u64) wk + (u64) j * n_workers; };

  /* threads 0 .. 7: the run table of the tile whose part[] entries lie in ring slot r */
  auto build_table = [&] (int r, int slot) {
    if (tid < KWAY_MAX) {
      const u64 s = sh.rng[r][0][tid], e = sh.rng[r][1][tid];
      const bool live = (u32) tid < p.k;
      sh.tab_start[slot][tid] = live ? s : 0;
      sh.tab_len[slot][tid] = live ? (u32) (e - s) : 0u;
    }
    if (tid == 0) {
      u32 total = 0;
      u64 base = 0;
      for (u32 i = 0; i < p.k; i++) {
        total += (u32) (sh.rng[r][1][i] - sh.rng[r][0][i]);
        base += sh.rng[r][0][i];
      }
      sh.tab_total[slot] = total;
      sh.tab_base[slot] = base;
    }
  };

  /* The tile's records, fetched one tile ahead into registers.  A wavefront fetches 64 consecutive
   * records of ONE run per instruction (wave slot w of the tile: the runs' slots are numbered one
   * run after the other, every run rounded up to whole slots), so run, descriptor and addresses are
   * scalar and the range-checked descriptor zero-fills past the run's end: no per-lane bounds. */
  u32x3 pre[J];
  u32 pre_dst[J]; /* record position in buffer 0, or ~0 */
  auto fetch = [&] (int slot, int part) { /* part j of J, or -1: all */
    u32 len[KWAY_MAX];
#pragma unroll
    for (int i = 0; i < KWAY_MAX; i++) len[i] = uniform32 (sh.tab_len[slot][i]);
#pragma unroll
    for (int j = 0; j < J; j++) {
      if (part >= 0 && part != j) continue;
      const u32 w = (u32) wid + (u32) j * NW; /* wave slot */
      /* run of this slot, its first slot and its first record position in the buffer (scalar) */
      u32 i = 0, w0 = 0, r0 = 0, acc_w = 0, acc_r = 0;
#pragma unroll
      for (int m = 0; m < KWAY_MAX; m++) {
        const u32 nw = (len[m] + WAVE - 1) / WAVE;
        if (nw && w >= acc_w) {
          i = m;
          w0 = acc_w;
          r0 = acc_r;
        }
        acc_w += nw;
        acc_r += len[m];
      }
      pre_dst[j] = 0xffffffffu;
      pre[j] = u32x3 { 0, 0, 0 };
      if (w < acc_w) {
        const u32 r = (w - w0) * WAVE + (u32) lane; /* record of the run */
        const u64 s = uniform64 (sh.tab_start[slot][i]);
        const u32 *base = p.list[0];
#pragma unroll
        for (int m = 1; m < KWAY_MAX; m++) base = i == (u32) m ? p.list[m] : base; /* scalar selects: no dynamic indexing of the kernel arguments */
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) (base + 3 * s), 0, (int) (12 * len[i]), 0x00020000);
        pre[j] = __builtin_amdgcn_raw_buffer_load_b96 (rs, 12 * r, 0, 0);
        if (r < len[i]) pre_dst[j] = r0 + r;
      }
    }
  };

  /* ---- prologue: part[] of the first two tiles, table and fetch of the first */
  if (tid < 2 * KWAY_MAX) {
    for (int r = 0; r < 2; r++) {
      const u64 t = tile_at (r);
      if (t < ntl) sh.rng[r][tid / KWAY_MAX][tid % KWAY_MAX] = part[(t + tid / KWAY_MAX) * KWAY_MAX + tid % KWAY_MAX];
    }
  }
  __syncthreads ();
  if (tile_at (0) < ntl) build_table (0, 0);
  __syncthreads ();
  if (tile_at (0) < ntl) fetch (0, -1);

  u64 acc_sum = 0; /* per-thread sum of kept counts */
  u64 blk_cnt = 0; /* thread 0: records kept */
  /* deferred write-out: the previous tile's kept records wait in sh.stage; its global offset is
   * resolved at the top of this iteration, a whole tile after its total was published */
  u32 pend_tot = 0;
  u64 pend_tile = 0, pend_base = 0;
  bool pend = false;
  int it = 0;
#ifdef GT4_PROFILE_PHASES
  u64 ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  u64 t_last;
  asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last) :: "memory");
#endif
  for (u64 tile = tile_at (0); tile < ntl; tile = tile_at (++it)) {
    const int slot = it & 1;
    const u64 nxt = tile_at (it + 1), nn = tile_at (it + 2);
    u32x4 *X = sh.rec[0], *Y = sh.rec[1];
    /* ---- registers -> LDS: the runs lie one after the other in buffer 0 */
#pragma unroll
    for (int j = 0; j < J; j++)
      if (pre_dst[j] != 0xffffffffu) X[pre_dst[j]] = u32x4 { pre[j].x, pre[j].y, pre[j].z, 0u };
    /* part[] entries two tiles ahead (consumed at the end of the iteration), table of the next tile */
    u64 hk = 0;
    if (tid < 2 * KWAY_MAX && nn < ntl) hk = __hip_atomic_load (&part[(nn + tid / KWAY_MAX) * KWAY_MAX + tid % KWAY_MAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nxt < ntl) build_table ((it + 1) % 3, slot ^ 1);
    if (MODE == KWAY_UNION && pend && wid == NW - 1) {
      const u64 x = resolve_offset (agg, carry, pend_tile, lane, 0, 0, ctl, spin_limit);
      if (lane == 0) sh.excl = x;
    }
    PHASE_STAMP (0); /* wait for the prefetched records, LDS stores, next table, resolve of the previous tile */
    __syncthreads ();
    /* The next tile's records are fetched and the previous tile's kept records leave for HBM in J
     * resp. three portions, one behind each barrier of the passes: a CU gets ~12 bytes per cycle
     * from HBM, a tile moves ~55 KB -- issued all at once the wavefronts would queue in front of the
     * memory pipeline for thousands of cycles with nothing else to do. */
    const u64 w_excl = MODE == KWAY_UNION ? uniform64 (sh.excl) : pend_base;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc ((void *) (out + 3 * w_excl), 0, (int) (12 * pend_tot), 0x00020000);
    auto write_part = [&] (int k) {
      const u32 c = (u32) tid + (u32) k * NT;
      if (MODE != KWAY_COUNT && pend && c < ((3 * pend_tot + 3) >> 2)) __builtin_amdgcn_raw_buffer_store_b128 (*reinterpret_cast<const u32x4 *> (sh.stage + 4 * c), w_rs, 16 * c, 0, 0);
    };
    static_assert ((3 * CAP / 4 + NT - 1) / NT <= 3, "the write-out has three portions");
    if (nxt < ntl) fetch (slot ^ 1, 0);
    write_part (0);
    PHASE_STAMP (1); /* barrier, first fetch and write-out portions */

    u32 rlen[SLOTS], roff[SLOTS]; /* wave-uniform: current runs (records, first record in the current buffer) */
    {
      u32 acc = 0;
#pragma unroll
      for (int i = 0; i < SLOTS; i++) {
        rlen[i] = i < KWAY_MAX ? uniform32 (sh.tab_len[slot][i < KWAY_MAX ? i : 0]) : 0u;
        roff[i] = acc;
        acc += rlen[i];
      }
    }
    const u32 total = uniform32 (sh.tab_total[slot]);
    const u64 out_base = uniform64 (sh.tab_base[slot]);
    if (total > (u32) CAP) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }

    /* ---- P pairwise passes inside LDS: runs (2m, 2m+1) of buffer X -> run m of buffer Y */
#pragma unroll
    for (int pass = 0; pass < P; pass++) {
      const int M = SLOTS >> (pass + 1); /* merges of this pass */
      /* positions per thread; every merge rounds its thread count up, hence the M spare threads */
      u32 VT = total ? (total + (u32) NT - (u32) M - 1u) / ((u32) NT - (u32) M) : 1u;
      VT = VT < p.vt_min ? p.vt_min : VT;
      u32 thr[SLOTS / 2 + 1], oo[SLOTS / 2 + 1], iters = 0;
      thr[0] = 0;
      oo[0] = 0;
#pragma unroll
      for (int m = 0; m < SLOTS / 2; m++) {
        if (m >= M) continue;
        const u32 lx = rlen[2 * m], ly = rlen[2 * m + 1], s = lx + ly;
        thr[m + 1] = thr[m] + (s + VT - 1) / VT;
        oo[m + 1] = oo[m] + s;
        const u32 mn = lx < ly ? lx : ly;
        const u32 itn = mn ? 32u - (u32) __builtin_clz (mn) : 0u;
        iters = itn > iters ? itn : iters;
      }
      /* this thread's merge and its descriptor (per-lane selects over at most four merges) */
      u32 xo = roff[0], lx = rlen[0], yo = roff[1], ly = rlen[1], ob = 0, t0 = 0;
#pragma unroll
      for (int m = 1; m < SLOTS / 2; m++) {
        if (m >= M) continue;
        const bool g = (u32) tid >= thr[m];
        xo = g ? roff[2 * m] : xo;
        lx = g ? rlen[2 * m] : lx;
        yo = g ? roff[2 * m + 1] : yo;
        ly = g ? rlen[2 * m + 1] : ly;
        ob = g ? oo[m] : ob;
        t0 = g ? thr[m] : t0;
      }
      const u32 s = lx + ly;
      const u32 d0 = ((u32) tid - t0) * VT;
      const bool work = (u32) tid < thr[M] && d0 < s;
      if (__builtin_amdgcn_ballot_w64 (work)) { /* wavefronts without a single position skip the pass */
        /* merge-path split: lo = records of X among the first d0 of merge (X, Y), X first on ties
         * (measured: a four-way split search with six LDS reads in flight per round and a serial merge
         * that prefetches the successors of both heads were both SLOWER than this plain form -- the
         * passes are bound by instruction issue and LDS cycles, not by the length of the chain) */
        u32 lo = d0 > ly ? d0 - ly : 0u, hi = d0 < lx ? d0 : lx;
        if (!work) lo = hi = 0;
        const u64 *const kx_base = reinterpret_cast<const u64 *> (X + xo);
        const u64 *const ky_base = reinterpret_cast<const u64 *> (X + yo);
        for (u32 itn = 0; itn < iters; itn++) {
          const bool act = lo < hi;
          const u32 mid = (lo + hi) >> 1;
          const u32 ia = act ? mid : 0u, ib = act ? d0 - 1u - mid : 0u;
          const u64 kx = kx_base[2 * ia], ky = ky_base[2 * ib];
          const bool c = kx <= ky;
          lo = (act && c) ? mid + 1u : lo;
          hi = (act && !c) ? mid : hi;
        }
        u32 pa = xo + lo, pb = yo + (d0 - lo);
        const u32 ea = xo + lx, eb = yo + ly;
        u32 po = ob + d0;
        const u32 pend_o = ob + (d0 + VT < s ? d0 + VT : s);
        if (work) {
          for (u32 j = 0; j < VT; j++) {
            if (po >= pend_o) break;
            const bool ax = pa < ea, bx = pb < eb;
            const u32x4 ra = X[ax ? pa : xo], rb = X[bx ? pb : yo]; /* exhausted side: any record inside the buffer */
            const u64 kx = (u64) ra.x | ((u64) ra.y << 32), ky = (u64) rb.x | ((u64) rb.y << 32);
            const bool take = ax && (!bx || kx <= ky);
            Y[po] = take ? ra : rb;
            pa += take ? 1u : 0u;
            pb += take ? 0u : 1u;
            po += 1;
          }
        }
      }
      __syncthreads ();
      /* the merged runs are the next pass's inputs */
#pragma unroll
      for (int m = 0; m < SLOTS / 2; m++) {
        if (m >= M) continue;
        rlen[m] = oo[m + 1] - oo[m];
        roff[m] = oo[m];
      }
      u32x4 *const tmp = X;
      X = Y;
      Y = tmp;
      /* next portions of the fetch and of the write-out, behind this pass's barrier */
      if (nxt < ntl) {
        if (pass + 1 < J) fetch (slot ^ 1, pass + 1);
        if (pass == P - 1)
          for (int j = P + 1; j < J; j++) fetch (slot ^ 1, j);
      }
      if (pass < 2) write_part (pass + 1);
      if (pass == 0) PHASE_STAMP (2);
      if (pass == 1) PHASE_STAMP (3);
      if (pass == 2) PHASE_STAMP (4);
    }
    /* X: the tile's records in key order, equal keys of different lists next to each other.  The
     * write-out of the previous tile (issued before the passes) has read sh.stage by now: every
     * wavefront passed at least one barrier behind its LDS reads. */

    u32 tile_total = 0;
    if (MODE == KWAY_DUPS) {
      __syncthreads (); /* every wavefront has read its last write-out portion from sh.stage */
      for (u32 q = (u32) tid; q < total; q += NT) {
        const u32x4 r = X[q];
        sh.stage[3 * q] = r.x;
        sh.stage[3 * q + 1] = r.y;
        sh.stage[3 * q + 2] = r.z;
      }
      tile_total = total;
    } else {
      /* ---- combine equal keys (union_multi :548-571), keep test (:574), compaction.  Lane-consecutive
       * positions (p = tid + i * NT): every LDS read is 64 consecutive records.  A position whose
       * left neighbour has another key is the head of its key's run (at most eight records: one per
       * list) and folds the counts of the positions behind it. */
      constexpr int NR = (CAP + NT - 1) / NT; /* rows of NT positions */
      u64 hkey[NR];
      u32 hf[NR];
      u32 kept_mask = 0;
#pragma unroll
      for (int i = 0; i < NR; i++) {
        const u32 q = (u32) tid + (u32) i * NT;
        const bool in = q < total;
        const u32x4 r = X[in ? q : 0u];
        const u64 key = (u64) r.x | ((u64) r.y << 32);
        const u64 prev = *reinterpret_cast<const u64 *> (X + (in && q ? q - 1u : 0u));
        const bool head = in && (q == 0 || prev != key);
        u32 f = p.rule == 7u ? p.count_override : r.z;
        bool more = head;
#pragma unroll
        for (int d = 1; d < KWAY_MAX; d++) {
          if (!__builtin_amdgcn_ballot_w64 (more)) break;
          const bool ok = more && q + (u32) d < total;
          const u32x4 r2 = X[ok ? q + (u32) d : 0u];
          more = ok && (((u64) r2.x | ((u64) r2.y << 32)) == key);
          f = !more ? f : (p.rule == 1u ? f + r2.z : (p.rule == 4u ? (r2.z > f ? r2.z : f) : p.count_override));
        }
        const bool keep = head && (p.filter == FILTER_RAW || f >= p.cutoff);
        hkey[i] = key;
        hf[i] = f;
        kept_mask |= keep ? 1u << i : 0u;
        acc_sum += keep ? f : 0u;
        const u64 m = __builtin_amdgcn_ballot_w64 (keep);
        if (lane == 0) sh.wave_tot[i * NW + wid] = (u32) __popcll (m);
      }
      PHASE_STAMP (5); /* combine */
      __syncthreads ();
      /* exclusive prefix over (row, wavefront) in position order: NR * NW <= 64 values, one DPP scan */
      static_assert (NR * NW <= WAVE, "one wavefront pass scans the row / wavefront counts");
      const u32 wt = lane < NR * NW ? sh.wave_tot[lane] : 0u;
      const u32 wincl = dpp_inclusive_scan_u32 (wt);
      tile_total = (u32) __builtin_amdgcn_readlane ((int) wincl, WAVE - 1);
      if (tid == 0) {
        blk_cnt += tile_total;
        if (MODE == KWAY_UNION) publish_u32 (&agg[tile], AGG_READY | tile_total);
      }
      if (MODE == KWAY_UNION) {
#pragma unroll
        for (int i = 0; i < NR; i++) {
          const bool keep = (kept_mask >> i) & 1u;
          const u64 m = __builtin_amdgcn_ballot_w64 (keep);
          const u32 before = (u32) __builtin_amdgcn_readlane ((int) (wincl - wt), i * NW + wid); /* kept in earlier rows / wavefronts */
          const u32 slot_o = before + __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u));
          if (keep) {
            sh.stage[3 * slot_o] = (u32) hkey[i];
            sh.stage[3 * slot_o + 1] = (u32) (hkey[i] >> 32);
            sh.stage[3 * slot_o + 2] = hf[i];
          }
        }
      }
      PHASE_STAMP (6); /* barrier behind the combine, totals, staging */
    }
    pend = MODE != KWAY_COUNT;
    pend_tot = tile_total;
    pend_tile = tile;
    pend_base = out_base;
    if (tid < 2 * KWAY_MAX && nn < ntl) sh.rng[(it + 2) % 3][tid / KWAY_MAX][tid % KWAY_MAX] = hk;
    __syncthreads (); /* the next tile's records may overwrite the buffers; sh.stage is complete */
    PHASE_STAMP (7); /* end barrier */
  }
#ifdef GT4_PROFILE_PHASES
  if (tid == GT4_STAMP_TID)
    for (int i = 0; i < 8; i++) atomicAdd (&ctl->phase_cycles[i], ph[i]);
#endif
  /* drain: the last tile is still staged */
  if (MODE != KWAY_COUNT && pend) {
    if (MODE == KWAY_UNION && wid == 0) {
      const u64 x = resolve_offset (agg, carry, pend_tile, lane, 0, 0, ctl, spin_limit);
      if (lane == 0) sh.excl = x;
    }
    __syncthreads ();
    write_out_tile<NT> (out, MODE == KWAY_UNION ? uniform64 (sh.excl) : pend_base, pend_tot, sh.stage, tid);
  }

  if (MODE != KWAY_DUPS) {
    const u64 v = wave_sum (acc_sum);
    if (lane == 0 && v) atomicAdd (&ctl->total_count[0], v);
    if (tid == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt);
  }
}

constexpr int KWAY_NT = 1024;
constexpr int KWAY_CAP = 3520;

template <int MODE, int P>
hipError_t launch_kway (hipStream_t s, int grid, const KwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  hipLaunchKernelGGL ((k_kway_merge<KWAY_NT, KWAY_CAP, MODE, P>), dim3 (grid), dim3 (KWAY_NT), 0, s, p, part, out, desc, ctl);
  return hipGetLastError ();
}

hipError_t launch_kway_mode (hipStream_t s, int mode, int grid, const KwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  const bool p3 = p.k > 4;
  if (mode == KWAY_DUPS) return p3 ? launch_kway<KWAY_DUPS, 3> (s, grid, p, part, out, desc, ctl) : launch_kway<KWAY_DUPS, 2> (s, grid, p, part, out, desc, ctl);
  if (mode == KWAY_COUNT) return p3 ? launch_kway<KWAY_COUNT, 3> (s, grid, p, part, out, desc, ctl) : launch_kway<KWAY_COUNT, 2> (s, grid, p, part, out, desc, ctl);
  return p3 ? launch_kway<KWAY_UNION, 3> (s, grid, p, part, out, desc, ctl) : launch_kway<KWAY_UNION, 2> (s, grid, p, part, out, desc, ctl);
}

}  // namespace

}  // namespace gt4

using namespace gt4;

/* ------------------------------------------------------------------ host orchestration */

namespace {

struct Level {
  KwayParams p;            /* lists of this level (level 0: the caller's; above: sample lists) */
  gt4hip_list *owned[KWAY_MAX];
  u64 total;
};

size_t kway_desc_bytes (u64 tiles)
{
  const u64 rows = (tiles + 63) / 64;
  return (((size_t) rows * 64 * 16 + (size_t) (rows + 1) * 32 + (size_t) rows * 32) + 255) & ~(size_t) 255; /* agg, carry, rowsum */
}

int kway_grow (gt4hip_context *ctx, void **p, size_t *have, size_t need)
{
  if (*have >= need) return GT4HIP_OK;
  if (*p) {
    HIPCHK (ctx, hipStreamSynchronize (ctx->stream));
    HIPCHK (ctx, hipFree (*p));
    *p = NULL;
    *have = 0;
  }
  need += need / 8;
  if (gt4hip_dev_alloc (ctx, p, need) != hipSuccess) return gt4hip_fail (ctx, GT4HIP_ENOMEM, "workspace hipMalloc of %zu bytes failed", need);
  *have = need;
  return GT4HIP_OK;
}

}  // namespace

/* N-way union of 3..8 non-empty lists in one pass.  *used = 0 when the call must take the pairwise
 * tree instead (a tile would not fit LDS: adversarial key distributions; or the single-pass chain
 * gave up on a shared device).  `out`: capacity >= sum of the lists (unless count_only). */
int gt4hip_kway_union (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                       uint32_t filter, bool count_only, gt4hip_list *out, uint64_t *n_words, uint64_t *total_count, double *device_ms,
                       int *used)
{
  *used = 0;
  if (k < 3 || k > KWAY_MAX) return GT4HIP_OK;
  hipStream_t st = ctx->stream;
  std::vector<Level> levels;
  Level l0;
  memset (&l0, 0, sizeof l0);
  l0.p.k = k;
  for (uint32_t i = 0; i < k; i++) {
    l0.p.list[i] = (const u32 *) lists[i]->dev;
    l0.p.n[i] = lists[i]->n_words;
    l0.total += lists[i]->n_words;
  }
  levels.push_back (l0);
  int rc = GT4HIP_OK;
  auto cleanup = [&] () {
    for (Level &lv : levels)
      for (int i = 0; i < KWAY_MAX; i++)
        if (lv.owned[i]) gt4hip_list_free (lv.owned[i]);
  };
  HIPCHK (ctx, hipEventRecord (ctx->ev[0], st));
  /* sample levels until one fits a single tile */
  while (levels.back ().total > (u64) KWAY_CAP) {
    const Level &lo = levels.back ();
    Level up;
    memset (&up, 0, sizeof up);
    up.p.k = k;
    for (uint32_t i = 0; i < k && !rc; i++) {
      const u64 m = lo.p.n[i] / KWAY_SAMPLE;
      rc = gt4hip_list_new (ctx, m ? m : 1, lists[0]->word_length, &up.owned[i]);
      if (rc) break;
      up.p.list[i] = (const u32 *) up.owned[i]->dev;
      up.p.n[i] = m;
      up.total += m;
      if (m) {
        u64 g = (m + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL (k_kway_sample, dim3 ((unsigned) g), dim3 (256), 0, st, lo.p.list[i], m, (u32 *) up.owned[i]->dev);
      }
    }
    levels.push_back (up);
    if (rc) {
      cleanup ();
      return rc;
    }
  }
  /* top-down: the merged samples of level l+1 cut level l into tiles */
  gt4hip_list *merged = NULL; /* merged sample records of the level above */
  const double margin = 5.0 * KWAY_SAMPLE * sqrt ((double) k / 6.0);
  u32 G = (u32) (((double) KWAY_CAP - margin) / KWAY_SAMPLE);
  if (G < 1) G = 1;
  if (ctx->kway_g > 0) G = (u32) ctx->kway_g;
  for (int l = (int) levels.size () - 1; l >= 0 && !rc; l--) {
    Level &lv = levels[l];
    const u64 m_total = merged ? merged->n_words : 0;
    const u64 tiles = m_total ? m_total / G + 2 : 1;
    if (tiles >= 0xfffffff0ull) {
      rc = gt4hip_fail (ctx, GT4HIP_EINVAL, "lists too long: %llu tiles", (unsigned long long) tiles);
      break;
    }
    lv.p.num_tiles = (u32) tiles;
    lv.p.rule = rule;
    lv.p.cutoff = cutoff;
    lv.p.count_override = ovr;
    lv.p.filter = filter;
    lv.p.spin_limit = ctx->spin_limit;
    lv.p.vt_min = ctx->kway_vt > 0 ? (u32) ctx->kway_vt : 6u;
    if ((rc = kway_grow (ctx, (void **) &ctx->kway_part, &ctx->kway_part_bytes, (size_t) (tiles + 1) * KWAY_MAX * 8))) break;
    {
      const u64 threads = (tiles + 1) * KWAY_MAX;
      hipLaunchKernelGGL (k_kway_partition, dim3 ((unsigned) ((threads + 255) / 256)), dim3 (256), 0, st, lv.p, merged ? (const u32 *) merged->dev : NULL,
                          m_total, G, (u64 *) ctx->kway_part);
      hipMemsetAsync (ctx->scratch, 0, 64, st);
      hipLaunchKernelGGL (k_kway_check, dim3 ((unsigned) ((tiles + 255) / 256)), dim3 (256), 0, st, (const u64 *) ctx->kway_part, (u32) tiles, (u32) KWAY_CAP,
                          (u32 *) ctx->scratch);
      hipError_t e = hipMemcpyAsync (ctx->scratch_host, ctx->scratch, 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) {
        rc = gt4hip_fail (ctx, GT4HIP_EHIP, "k-way partition failed: %s", hipGetErrorString (e));
        break;
      }
      if (ctx->scratch_host[0] & 0xffffffffu) { /* a tile would overflow LDS: the caller takes the pairwise tree */
        if (merged) gt4hip_list_free (merged);
        cleanup ();
        ctx->kway_overflows++;
        return GT4HIP_OK;
      }
    }
    if (merged) {
      gt4hip_list_free (merged);
      merged = NULL;
    }
    const int mode = l > 0 ? KWAY_DUPS : (count_only ? KWAY_COUNT : KWAY_UNION);
    u32 *dst = NULL;
    if (l > 0) {
      if ((rc = gt4hip_list_new (ctx, lv.total ? lv.total : 1, lists[0]->word_length, &merged))) break;
      merged->n_words = lv.total;
      dst = (u32 *) merged->dev;
    } else if (!count_only) {
      dst = (u32 *) out->dev;
    }
    int grid = ctx->n_cus;
    if (ctx->grid_override > 0) grid = (int) ctx->grid_override;
    if (mode == KWAY_UNION) {
      if ((rc = kway_grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, kway_desc_bytes (tiles)))) break;
      hipMemsetAsync (ctx->desc, 0, kway_desc_bytes (tiles), st);
      if ((u64) grid > tiles + 1) grid = (int) tiles + 1;
    } else if ((u64) grid > tiles) {
      grid = (int) tiles;
    }
    hipMemsetAsync (ctx->ctl, 0, sizeof (PairControl), st);
    hipError_t e = launch_kway_mode (st, mode, grid, lv.p, (const u64 *) ctx->kway_part, dst, (u64 *) ctx->desc, ctx->ctl);
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "k-way merge launch failed: %s", hipGetErrorString (e));
      break;
    }
    if (l == 0) {
      HIPCHK (ctx, hipEventRecord (ctx->ev[3], st));
      e = hipMemcpyAsync (ctx->ctl_host, ctx->ctl, sizeof (PairControl), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) {
        rc = gt4hip_fail (ctx, GT4HIP_EHIP, "k-way merge failed: %s", hipGetErrorString (e));
        break;
      }
      if (ctx->ctl_host->error) {
        /* a bounded wait gave up (shared device) or a consistency check tripped: the tree redoes the call */
        ctx->single_pass_fallbacks++;
        cleanup ();
        return GT4HIP_OK;
      }
#ifdef GT4_PROFILE_PHASES
      {
        static const char *names[8] = { "wait+store+resolve", "B0+fetch+writeout", "pass0", "pass1", "pass2", "combine", "B+move", "end barrier" };
        unsigned long long tot = 0;
        for (int i = 0; i < 8; i++) tot += ctx->ctl_host->phase_cycles[i];
        fprintf (stderr, "[kway phases] tiles %llu:", (unsigned long long) tiles);
        for (int i = 0; i < 8; i++) fprintf (stderr, " %s %.1f%%", names[i], tot ? 100.0 * ctx->ctl_host->phase_cycles[i] / tot : 0.0);
        fprintf (stderr, " | avg cycles/tile %.0f\n", tiles ? (double) tot / tiles : 0.0);
      }
#endif
      *n_words = ctx->ctl_host->n_words[0];
      *total_count = ctx->ctl_host->total_count[0];
      float ms = 0;
      if (hipEventElapsedTime (&ms, ctx->ev[0], ctx->ev[3]) == hipSuccess) *device_ms = ms;
      *used = 1;
    }
  }
  if (merged) gt4hip_list_free (merged);
  cleanup ();
  return rc;
}
