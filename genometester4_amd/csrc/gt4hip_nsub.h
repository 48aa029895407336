/* gt4hip_nsub.h -- the N-way tile kernel with WAVE-PRIVATE SUB-TILES (round 5).  Included by gt4hip_nway.hip
 * (inside its anonymous namespace, behind k_nway_merge, whose helpers, parameters and partition it shares).
 *
 * What it restates: union_multi (reference src/glistcompare.c:500-603; hot loop :545-591) and gt4_write_union
 * (src/set-operations.c:40-129), modes NWAY_UNION and NWAY_COUNT of k_nway_merge -- same tiles, same results.
 *
 * Why: k_nway_merge keeps its sixteen wavefronts in lock-step (six workgroup barriers per tile; 17 - 34 % of a
 * tile inside them, no unit of the CU saturated: profiles/round4).  Here a tile still belongs to one workgroup,
 * but behind ONE hand-off the wavefronts never wait for each other again:
 *
 *   1. every wavefront fetches its 64-record slots of the tile after next into registers (as before) and, in the
 *      MIDDLE of the tile it is working on, stores them to LDS as the sorted runs they are ("raw": keys and counts
 *      apart, so that keys are 8-byte aligned);
 *   2. the SERVICE wavefront (the sixteenth; it owns no records) cuts the runs at NSUB - 1 = 14 splitter keys -- the
 *      keys of the tile's longest run at equal distances -- by binary searches in LDS (14 x 8 lanes), while the
 *      fifteen WORKERS are still busy with the tile before.  Records with a key <= the splitter go left: equal keys
 *      of different lists stay in one sub-range;
 *   3. worker w owns sub-range w (at most 256 records, about 200): it gathers them from the runs, buckets them by
 *      interpolation inside the sub-range's key range (256 buckets, 16-bit counters, LDS atomics that return the
 *      arrival number), scans the counters itself (four per lane, one DPP scan), groups the keys by bucket, ranks every
 *      key by walking its bucket, folds equal keys by ONE returning 64-bit LDS atomic on {sum, arrivals} at the
 *      key's position (first arrival writes the key), reads the positions back in order (16 bytes per lane) and
 *      keeps the surviving records IN REGISTERS -- all of it in its own 5 KB of LDS, ordered only by the fact that
 *      one wavefront's LDS operations complete in order: no s_barrier, no s_waitcnt between wavefronts;
 *   4. the kept records leave from the registers one tile later (12-byte stores, consecutive lanes consecutive
 *      records) at the offset the chained scan (gt4hip_device.h) has published by then; the service wavefront
 *      adds up the workers' totals, publishes the tile's, resolves its offset.
 *
 *   Hand-offs are monotonic counters in LDS (arrive = one LDS add, wait = poll + s_sleep, all bounded): raw
 *   written (16 wavefronts), cuts ready, gathered (15 workers: the raw area may be overwritten), tile ended
 *   (15: totals are in), offset known.  In steady state nobody waits: each is asked for half a tile after it
 *   was signalled.
 *
 * A sub-range beyond 256 records (lists of very different density inside one tile) sends the TILE through the
 * slow path: every record adds up its lower bounds in all the runs, positions are folded in one tile-wide array
 * (three software barriers).  A bucket beyond SUB_LIMIT keys (clustered keys) sends the WORKER through the same
 * searches inside its own sub-runs.  Results are identical on every path (tests force each: option "kway_vt" 99 / 98).
 */

constexpr int SUB_NT = 1024;
constexpr int SUB_NW = SUB_NT / WAVE;     /* wavefronts: fifteen workers and the service wavefront */
constexpr int SUB_NSUB = SUB_NW - 1;      /* sub-ranges of a tile = workers */
constexpr int SUB_RW = 4;                 /* records per lane of a worker */
constexpr int SUB_CAPW = SUB_RW * WAVE;   /* records a worker takes */
constexpr int SUB_NCH = 64;               /* 64-record slots of a tile: four per wavefront */
constexpr int SUB_POS = SUB_NCH * WAVE;   /* raw positions (runs rounded up to whole slots) */
constexpr int SUB_MAXREC = SUB_NSUB * SUB_CAPW; /* records of a tile (3840) */
constexpr int SUB_NBW = 256;              /* buckets of a worker */
#ifndef GT4_SUB_LIMIT
#define GT4_SUB_LIMIT 32
#endif
constexpr int SUB_LIMIT = GT4_SUB_LIMIT;  /* keys per bucket the walks handle */
static_assert (SUB_LIMIT % 2 == 0 && SUB_LIMIT <= NWAY_LIMIT, "bucket walks go two steps at a time, by nway_rank_steps");

/* positions of the fold: 16 bytes each {key, sum, arrivals}, skewed (index p lives at p + p / 8): records of one
 * list in consecutive lanes lie about as many positions apart as there are lists */
__host__ __device__ constexpr u32 sub_phys (u32 p) { return p + (p >> 3); }
constexpr int SUB_GKN = SUB_CAPW + SUB_CAPW / 32 + NWAY_LIMIT + 4; /* grouped keys (skewed as in k_nway_merge) + the longest walk behind the last */
constexpr int SUB_PN = (int) sub_phys (SUB_CAPW - 1) + 1 + 1;       /* positions of a worker (+1: even) */
constexpr int SUB_TPN = (int) sub_phys (SUB_MAXREC - 1) + 1 + 3;   /* positions of the slow path's tile-wide array */

struct SubPriv {
  alignas (16) u32 cntw[SUB_NBW / 2];  /* 16-bit bucket counters, then bucket starts */
  alignas (16) u32 wtab[8];            /* raw index - lane slot of each of the worker's sub-runs */
  union {
    alignas (16) u64 gk[SUB_GKN];      /* keys grouped by bucket; all-ones wherever no key is */
    alignas (16) u32x4 pos[SUB_PN];    /* ... then the positions of the fold (the walks are over) */
  };
};

struct SubShared {
  alignas (16) u64 rawk[SUB_POS];      /* the tile as sorted runs: keys */
  alignas (16) u32 rawc[SUB_POS];      /* ... counts */
  union {
    SubPriv priv[SUB_NSUB];
    alignas (16) u32x4 tpos[SUB_TPN];  /* slow path: the positions of the whole tile */
  };
  u64 slot_addr[3][SUB_NCH];
  alignas (16) u32 slot_cnt[3][SUB_NCH];
  u32 tab_pbase[3][NWAY_MAX];          /* first raw position of each run */
  u32 tab_len[3][NWAY_MAX];
  alignas (16) u32 hdr[3][12];         /* tile (0xffffffff: none), records, slots, -, -, smallest possible key (2), -, -, longest run: first position, records */
  unsigned short cuts[2][SUB_NW][NWAY_MAX]; /* [tile parity][boundary b][run]: records of the run in sub-ranges below b */
  u64 cutkey[2][SUB_NW];               /* largest key of the sub-ranges below b (b = 0: the tile's smallest possible key - 1) */
  u32 slow[2];                         /* the tile takes the slow path */
  u32 kept[2][SUB_NW];                 /* records every worker keeps */
  u32 wpre[2][SUB_NW];                 /* ... kept by the workers before it */
  u64 excl[2];                         /* records in front of the tile in the output */
  /* hand-offs: monotonic counters */
  u32 c_raw, c_cuts, c_gath, c_end, c_x, c_tab, c_sb, c_abort, c_kept, c_pad[3];
  u32 tick;
};

__device__ __forceinline__ u32 sub_peek (u32 *p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sub_poke (u32 *p, u32 v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
/* one wavefront's LDS operations complete in order: what it stored before the add is there when the add is seen.
 * The clobbers keep the compiler from moving LDS accesses across */
__device__ __forceinline__ void sub_arrive (u32 *p, int lane)
{
  asm volatile ("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add (p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile ("" ::: "memory");
}
__device__ __forceinline__ void sub_signal (u32 *p, u32 v, int lane)
{
  asm volatile ("" ::: "memory");
  if (lane == 0) sub_poke (p, v);
  asm volatile ("" ::: "memory");
}
/* wait until *p >= target (wrap-safe); false: gave up (the launch is over: ctl->error) */
#ifndef GT4_SUB_SLEEP
#define GT4_SUB_SLEEP 2
#endif
__device__ __forceinline__ bool sub_wait (u32 *p, u32 target, u32 *abort_word, u32 limit)
{
  /* the CU has ONE scalar unit: a polling wavefront's loop is paid by everybody (measured: 625 scalar instructions per
   * wavefront and tile with a ten-instruction loop and s_sleep 1, five times k_nway_merge's).  The loop is the look,
   * one compare, the sleep; the bound and the abort word are looked at every 256th round only. */
  asm volatile ("" ::: "memory");
  bool ok = true;
#ifdef GT4_SUB_OLDWAIT
  u32 spins = 0;
  for (;;) {
    const u32 v = uniform32 (sub_peek (p));
    if ((int) (v - target) >= 0) break;
    if (++spins > limit || ((spins & 31u) == 0 && uniform32 (sub_peek (abort_word)))) {
      ok = false;
      break;
    }
    __builtin_amdgcn_s_sleep (1);
  }
  asm volatile ("" ::: "memory");
  return ok;
#endif
  u32 rounds = 0;
  for (;;) {
    u32 v = 0;
#pragma unroll 1
    for (u32 i = 0; i < 256u; i++) {
      v = uniform32 (sub_peek (p));
      if ((int) (v - target) >= 0) break;
      __builtin_amdgcn_s_sleep (GT4_SUB_SLEEP);
    }
    if ((int) (v - target) >= 0) break;
    if (++rounds > (limit >> 8) || uniform32 (sub_peek (abort_word))) {
      ok = false;
      break;
    }
  }
  asm volatile ("" ::: "memory");
  return ok;
}

/* The worker's LDS area is written and read under several types (u64 keys, 16-byte positions, 64-bit atomics on
 * their halves, 16-bit counters): type-based alias analysis would let the compiler move, say, the zeroing stores of
 * the positions behind the atomics that fold into them (it did).  A compiler fence -- no instruction -- at every
 * phase boundary keeps the program order the hardware then honours (one wavefront's LDS operations complete in order). */
__device__ __forceinline__ void sub_fence () { asm volatile ("" ::: "memory"); }

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ u64 sub_lds_add_u64 (u32 byte_offset, u64 v)
{
  return __hip_atomic_fetch_add ((lds_u64 *) byte_offset, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); /* ds_add_rtn_u64 */
}
#pragma clang diagnostic pop

/* a constant made where it is used (one v_mov): the compiler otherwise keeps fill patterns in registers across the
 * whole tile loop -- and spilled them */
__device__ __forceinline__ u32 sub_const (u32 v)
{
  u32 r;
  asm volatile ("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
  return r;
}

template <int MODE>
__global__ __launch_bounds__ (SUB_NT, 4) void
k_nway_sub (NwayParams p, const u64 *__restrict__ part, u32 *__restrict__ out, u64 *desc, PairControl *ctl)
{
  static_assert (MODE == NWAY_UNION || MODE == NWAY_COUNT, "the other modes stay with k_nway_merge");
  constexpr int NW = SUB_NW, NSUB = SUB_NSUB, RW = SUB_RW, NCH = SUB_NCH;
  __shared__ SubShared sh;
  const int tid = threadIdx.x;
  int lane = tid & (WAVE - 1); /* (not const: made opaque once per tile, see the workers' loop) */
  const int wid = __builtin_amdgcn_readfirstlane (tid / WAVE);
  const u64 n_rows = ((u64) p.num_tiles + WAVE - 1) / WAVE;
  const u32 spin_limit = p.spin_limit ? p.spin_limit : SPIN_LIMIT;
  u32 *const agg = reinterpret_cast<u32 *> (desc);
  u64 *const carry = desc + 2 * n_rows * WAVE;

  u32 role = 0;
  if (MODE == NWAY_UNION) {
    if (tid == 0) sh.tick = atomicAdd (&ctl->role, 1u);
    __syncthreads ();
    role = sh.tick;
    __syncthreads ();
    if (role == 0) {
      const u32 n_sub = p.scan_group ? 8u : 1u;
      if ((u32) wid < n_sub) scanner_part (agg, carry + 4 * (n_rows + 1), carry, p.num_tiles, ctl, lane, spin_limit, (u32) wid, n_sub);
      return;
    }
  }
  const u32 n_workers = MODE == NWAY_UNION ? gridDim.x - 1 : gridDim.x;
  const u32 wk = MODE == NWAY_UNION ? role - 1 : blockIdx.x;
  const u32 ntl = p.num_tiles;
  const bool service = wid == NW - 1;
  if (service) __builtin_amdgcn_s_setprio (GT4_NWAY_SVPRIO);

  auto deal = [&] (int j) -> u32 { /* lane 0 of the service wavefront */
    if (p.dynamic) {
      const u32 t = atomicAdd (&ctl->ticket, 1u);
      return t < ntl ? t : 0xffffffffu;
    }
    const u64 t = (u64) wk + (u64) j * n_workers;
    return t < (u64) ntl ? (u32) t : 0xffffffffu;
  };
  auto load_row = [&] (u32 tile) -> u64 { /* lane i: entry i of the tile's two partition rows (its start and its end) */
    u64 v = 0;
    if (tile < ntl && lane < 2 * NWAY_PSTRIDE)
      v = __hip_atomic_load (&part[(u64) tile * NWAY_PSTRIDE + (u64) lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  /* the slot table of `tile` into table tb (k_nway_merge's, less what only its buckets needed) */
  auto build_table = [&] (u64 row, u32 tile, int tb) {
    if (tile >= ntl) {
      if (lane == 0) sh.hdr[tb][0] = 0xffffffffu;
      return;
    }
    u64 lbv = 0;
#pragma unroll
    for (int m = 0; m < NWAY_MAX; m++) lbv = lane == m ? (u64) p.list[m] : lbv;
    const u32 rlo = (u32) row, rhi = (u32) (row >> 32);
    const u32 elo = __shfl_down (rlo, NWAY_PSTRIDE, WAVE);
    const u32 len = (u32) lane < p.k ? elo - rlo : 0u;
    const u32 nw = (len + WAVE - 1) / WAVE;
    const u32 incl = dpp_inclusive_scan_u32 (nw), excl = incl - nw;
    const u32 total = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
    const u32 n = dpp_wave_sum_u32 (len);
    if (lane < NWAY_MAX) {
      sh.tab_pbase[tb][lane] = excl * WAVE;
      sh.tab_len[tb][lane] = len;
    }
    {
      const u32 slot = (u32) lane;
      u32 run = 0;
#pragma unroll
      for (int q = 0; q < NWAY_MAX - 1; q++) run += slot >= (u32) __builtin_amdgcn_readlane ((int) incl, q) ? 1u : 0u;
      const u32 len_r = __shfl (len, run, WAVE), excl_r = __shfl (excl, run, WAVE);
      const u64 s_r = (u64) __shfl (rlo, run, WAVE) | ((u64) __shfl (rhi, run, WAVE) << 32);
      const u64 lb_r = (u64) __shfl ((u32) lbv, run, WAVE) | ((u64) __shfl ((u32) (lbv >> 32), run, WAVE) << 32);
      const bool in = slot < total;
      const u32 first = in ? (slot - excl_r) * WAVE : 0u;
      sh.slot_cnt[tb][slot] = in ? (len_r - first < (u32) WAVE ? len_r - first : (u32) WAVE) : 0u;
      sh.slot_addr[tb][slot] = lb_r + 12ull * (s_r + first);
    }
    const u32 lo_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX), lo_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX);
    u32 pv_len = 0, pv_base = 0;
#pragma unroll
    for (int q = 0; q < NWAY_MAX; q++) {
      const u32 lq = (u32) __builtin_amdgcn_readlane ((int) len, q), bq = (u32) __builtin_amdgcn_readlane ((int) excl, q) * WAVE;
      const bool better = lq > pv_len; /* uniform */
      pv_base = better ? bq : pv_base;
      pv_len = better ? lq : pv_len;
    }
    u32 h = tile;
    h = lane == 1 ? n : h;
    h = lane == 2 ? total : h;
    h = lane == 5 ? lo_lo : h;
    h = lane == 6 ? lo_hi : h;
    h = lane == 9 ? pv_base : h;
    h = lane == 10 ? pv_len : h;
    if (lane < 11) sh.hdr[tb][lane] = h;
  };

  /* the wavefront's four slots of a tile, into registers (scalar addresses, range-checked descriptor: zeros behind a run's end) */
  u32x3 pre[RW];
  auto fetch = [&] (int tb) {
#pragma unroll
    for (int k = 0; k < RW; k++) {
      const int chunk = wid * RW + k;
      const u64 addr = uniform64 (sh.slot_addr[tb][chunk]);
      const u32 c = uniform32 (sh.slot_cnt[tb][chunk]);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) addr, 0, (int) (12 * c), 0x00020000);
      pre[k] = __builtin_amdgcn_raw_buffer_load_b96 (rs, 12 * lane, 0, 0);
    }
  };
  auto fetch_or_not = [&] (int tb) { /* the tile of table tb, if there is one and this wavefront has slots of it */
    const u32 t = uniform32 (sh.hdr[tb][0]);
    if (t < ntl && (u32) (wid * RW) < uniform32 (sh.hdr[tb][2])) {
      fetch (tb);
    } else {
#pragma unroll
      for (int k = 0; k < RW; k++) asm volatile ("" : "=v"(pre[k].x), "=v"(pre[k].y), "=v"(pre[k].z));
    }
  };
  auto write_raw = [&] (int tb) { /* ... from the registers to the raw area (slot s = positions 64 s ..) */
    if ((u32) (wid * RW) < uniform32 (sh.hdr[tb][2])) {
#pragma unroll
      for (int k = 0; k < RW; k++) {
        const u32 q = (u32) (wid * RW + k) * WAVE + (u32) lane;
        sh.rawk[q] = (u64) pre[k].x | ((u64) pre[k].y << 32);
        sh.rawc[q] = pre[k].z;
      }
    }
  };
  u32 *const abortw = &sh.c_abort;
  bool dead = false; /* a wait gave up */
  auto give_up = [&] () {
    dead = true;
    if (lane == 0) {
      sub_poke (abortw, 1u);
      atomicOr (&ctl->error, 1u);
    }
  };
#define SUB_WAIT(ctr, target) do { if (!sub_wait (&sh.ctr, (target), abortw, spin_limit)) give_up (); } while (0)

  /* ---- prologue */
  u32 sv_t2 = 0xffffffffu, sv_tk = 0xffffffffu;
  u64 sv_row = 0;
  if (tid < 12) (&sh.c_raw)[tid] = 0;
  if (service) {
    u32 d[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
    if (lane == 0)
      for (int q = 0; q < 4; q++) d[q] = deal (q);
    const u32 d0 = uniform32 (d[0]), d1 = uniform32 (d[1]);
    sv_t2 = uniform32 (d[2]);
    sv_tk = d[3];
    const u64 r0 = load_row (d0), r1 = load_row (d1);
    sv_row = load_row (sv_t2);
    build_table (r0, d0, 0);
    build_table (r1, d1, 1);
  } else {
    SubPriv &pv = sh.priv[wid];
    pv.cntw[lane] = 0;
    pv.cntw[WAVE + lane] = 0;
    for (int i = lane; i < SUB_GKN; i += WAVE) pv.gk[i] = ~0ull;
  }
  __syncthreads ();
  fetch_or_not (0);
  if (uniform32 (sh.hdr[0][0]) < ntl) write_raw (0);
  fetch_or_not (1);
  __syncthreads (); /* raw (0) is complete, tables 0 and 1 are built */

  u64 acc_sum = 0; /* per-thread sum of kept counts */
  u64 blk_cnt = 0; /* records kept by this wavefront (the same in every lane) */
  int tb = 0, tb1 = 1, tb2 = 2;
  u32 it = 0;
#ifdef GT4_PROFILE_PHASES
  u64 ph[24];
  for (int i = 0; i < 24; i++) ph[i] = 0;
  u64 t_last;
  asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last) :: "memory");
#define SUB_STAMPS_OUT() do { if (tid == GT4_STAMP_TID) for (int i = 0; i < 24; i++) atomicAdd (&ctl->phase_cycles[i], ph[i]); } while (0)
#else
#define SUB_STAMPS_OUT() do { } while (0)
#endif

  if (service) {
    /* =============================================================== the service wavefront */
    /* the runs of the tile of table tbx (in the raw area) cut at NSUB - 1 keys of its longest run */
    auto compute_cuts = [&] (int tbx, u32 par) {
      const u32 r = (u32) lane & 7u;
      const u32 len_r = sh.tab_len[tbx][r], pb_r = sh.tab_pbase[tbx][r];
      const u32 pv_base = uniform32 (sh.hdr[tbx][9]), pv_len = uniform32 (sh.hdr[tbx][10]);
      const u64 key_lo = (u64) uniform32 (sh.hdr[tbx][5]) | ((u64) uniform32 (sh.hdr[tbx][6]) << 32);
      /* boundaries 1 .. 8 in round A, 9 .. 16 in round B (those up to NSUB - 1 exist) */
      const u32 bA = 1u + ((u32) lane >> 3), bB = 9u + ((u32) lane >> 3);
      const bool inB = bB < (u32) NSUB;
      const u32 pidxA = (u32) (((u64) bA * pv_len) / (u32) NSUB), pidxB = inB ? (u32) (((u64) bB * pv_len) / (u32) NSUB) : 0u;
      const u64 sA = sh.rawk[pv_base + (pidxA ? pidxA - 1u : 0u)], sB = sh.rawk[pv_base + (pidxB ? pidxB - 1u : 0u)];
      u32 loA = 0, hiA = len_r, loB = 0, hiB = len_r;
      const u32 steps = pv_len ? 32u - (u32) __builtin_clz (pv_len) : 0u; /* (the longest run: uniform) */
      for (u32 s = 0; s < steps; s++) {
        const bool actA = loA < hiA, actB = loB < hiB;
        const u32 midA = (loA + hiA) >> 1, midB = (loB + hiB) >> 1;
        const u64 kA = sh.rawk[pb_r + (actA ? midA : 0u)], kB = sh.rawk[pb_r + (actB ? midB : 0u)];
        const bool cA = kA <= sA, cB = kB <= sB;
        loA = (actA && cA) ? midA + 1u : loA;
        hiA = (actA && !cA) ? midA : hiA;
        loB = (actB && cB) ? midB + 1u : loB;
        hiB = (actB && !cB) ? midB : hiB;
      }
      const u32 cutA = pidxA ? loA : 0u, cutB = pidxB ? loB : 0u;
      sh.cuts[par][bA][r] = (unsigned short) cutA;
      if (inB) sh.cuts[par][bB][r] = (unsigned short) cutB;
      if (lane < NWAY_MAX) {
        sh.cuts[par][0][lane] = 0;
        sh.cuts[par][NSUB][lane] = (unsigned short) len_r;
      }
      if (r == 0) {
        sh.cutkey[par][bA] = pidxA ? sA : key_lo - 1ull;
        if (inB) sh.cutkey[par][bB] = pidxB ? sB : key_lo - 1ull;
      }
      /* the tile's largest key: the largest of the runs' last keys */
      u64 last = (lane < NWAY_MAX && len_r) ? sh.rawk[pb_r + len_r - 1u] : 0ull;
#pragma unroll
      for (int m = 1; m < NWAY_MAX; m <<= 1) {
        const u64 o = shfl_xor_u64 (last, m);
        last = o > last ? o : last;
      }
      if (lane == 0) {
        sh.cutkey[par][0] = key_lo - 1ull;
        sh.cutkey[par][NSUB] = last;
      }
      /* does every sub-range fit a worker?  (the table is this wavefront's own: its LDS operations complete in order) */
      asm volatile ("" ::: "memory");
      const u32 b0 = (u32) lane >> 3, b1 = 8u + ((u32) lane >> 3);
      u32 d0 = (u32) sh.cuts[par][b0 + 1][r] - (u32) sh.cuts[par][b0][r];
      u32 d1 = b1 < (u32) NSUB ? (u32) sh.cuts[par][b1 + 1][r] - (u32) sh.cuts[par][b1][r] : 0u;
#pragma unroll
      for (int m = 1; m < NWAY_MAX; m <<= 1) {
        d0 += __shfl_xor (d0, m, WAVE);
        d1 += __shfl_xor (d1, m, WAVE);
      }
      const bool too_long = __any (d0 > (u32) SUB_CAPW || d1 > (u32) SUB_CAPW);
      if (lane == 0) sh.slow[par] = (too_long || p.force_fallback == 1u) ? 1u : 0u;
    };
    /* the workers' totals of the tile of iteration j: kept before each, the tile's total published */
    auto finish_tile = [&] (u32 j, u32 tile) {
      const u32 x = lane < NSUB ? sh.kept[j & 1u][lane] : 0u;
      const u32 incl = dpp_inclusive_scan_u32 (x);
      const u32 total = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
      if (lane < NSUB) sh.wpre[j & 1u][lane] = incl - x;
      if (MODE == NWAY_UNION && lane == 0) publish_u32 (&agg[tile], AGG_READY | total);
      if (MODE == NWAY_COUNT && lane == 0 && p.tile_totals) p.tile_totals[tile] = total;
    };
    /* cuts of the first tile */
    if (uniform32 (sh.hdr[0][0]) < ntl) compute_cuts (0, 0u);
    sub_signal (&sh.c_tab, 2u, lane);
    sub_signal (&sh.c_cuts, 1u, lane);
    /* Order of an iteration (tile `it` is with the workers): table of it + 2; the records of it + 1 to the raw area
     * once everybody has gathered it; the cuts of it + 1; the offset of it - 1 (its chain words were asked for an
     * iteration ago); then the workers' totals of tile `it` -- known behind their folds, well before the tile ends --
     * are added up and published, and the chain words of `it` asked for. */
    u32 prev_tile = 0xffffffffu;
    u32 xagg = 0;
    u64 xcarry = 0;
    for (;; it++) {
      PHASE_STAMP (23);
      const u32 cur = uniform32 (sh.hdr[tb][0]);
      if (cur >= ntl || dead) break;
      /* the table of the tile after next, the ticket behind it, its partition entries */
      build_table (sv_row, sv_t2, tb2);
      sub_signal (&sh.c_tab, it + 3u, lane);
      {
        const u32 t3 = uniform32 (sv_tk);
        sv_t2 = t3;
        sv_row = load_row (t3);
        if (lane == 0) sv_tk = deal ((int) it + 4);
      }
      if (MODE == NWAY_UNION && it > 0) { /* (asked for again: the look an iteration ago was early) */
        const u64 prow = prev_tile / WAVE;
        const bool mine = (u32) lane < prev_tile % WAVE;
        if (mine && !(xagg & AGG_READY)) xagg = peek_u32 (&agg[prow * WAVE + lane]);
        if (!(xcarry & CARRY_READY)) xcarry = peek_u64 (&carry[prow]);
      }
      PHASE_STAMP (13);
      /* everybody has gathered this tile: the next one's records go to the raw area */
      SUB_WAIT (c_gath, (u32) NSUB * (it + 1u));
      PHASE_STAMP (16);
      const u32 nxt = uniform32 (sh.hdr[tb1][0]);
      if (nxt < ntl) {
        write_raw (tb1);
        sub_arrive (&sh.c_raw, lane);
        fetch_or_not (tb2);
      }
      PHASE_STAMP (17);
      /* the offset of the tile before, if the chain has it by now (the workers ask for it behind their folds) */
      bool x_done = it == 0;
#ifdef GT4_SUB_LATEX
      if (false) {
#else
      if (it > 0) {
#endif
        if (MODE == NWAY_UNION) {
          const bool mine = (u32) lane < prev_tile % WAVE;
          if (__all (!mine || (xagg & AGG_READY) != 0) && (xcarry & CARRY_READY)) {
            const u64 x = (xcarry & ~CARRY_READY) + dpp_wave_sum_u32 (mine ? (xagg & ~AGG_READY) : 0u);
            if (lane == 0) sh.excl[(it - 1u) & 1u] = x;
            x_done = true;
          } else { /* asked for once more; looked at behind the cuts */
            const u64 prow = prev_tile / WAVE;
            if (mine && !(xagg & AGG_READY)) xagg = peek_u32 (&agg[prow * WAVE + lane]);
            if (!(xcarry & CARRY_READY)) xcarry = peek_u64 (&carry[prow]);
          }
        } else {
          x_done = true;
        }
        if (x_done) sub_signal (&sh.c_x, it, lane);
      }
      PHASE_STAMP (18);
      /* the next tile's cuts, as soon as all sixteen wavefronts have stored their slots */
      if (nxt < ntl) {
        SUB_WAIT (c_raw, (u32) NW * (it + 1u));
        PHASE_STAMP (19);
        compute_cuts (tb1, (it + 1u) & 1u);
        sub_signal (&sh.c_cuts, it + 2u, lane);
        PHASE_STAMP (20);
      }
      if (!x_done) {
        const u64 x = resolve_offset (agg, carry, prev_tile, lane, xagg, xcarry, ctl, spin_limit);
        if (lane == 0) sh.excl[(it - 1u) & 1u] = x;
        sub_signal (&sh.c_x, it, lane);
        PHASE_STAMP (21);
      }
      /* this tile: every worker knows what it keeps */
      SUB_WAIT (c_kept, (u32) NSUB * (it + 1u));
      PHASE_STAMP (14);
      finish_tile (it, cur);
      if (MODE == NWAY_UNION) {
        const u64 prow = cur / WAVE;
        xagg = (u32) lane < cur % WAVE ? peek_u32 (&agg[prow * WAVE + lane]) : 0u;
        xcarry = peek_u64 (&carry[prow]);
      }
      PHASE_STAMP (15);
      prev_tile = cur;
      const int t0 = tb;
      tb = tb1;
      tb1 = tb2;
      tb2 = t0;
    }
    if (it > 0 && !dead) { /* the last tile's offset */
      if (MODE == NWAY_UNION) {
        const u64 x = resolve_offset (agg, carry, prev_tile, lane, xagg, xcarry, ctl, spin_limit);
        if (lane == 0) sh.excl[(it - 1u) & 1u] = x;
      }
      sub_signal (&sh.c_x, it, lane);
    }
    SUB_STAMPS_OUT ();
    return;
  }

  /* =============================================================== the workers */
  SubPriv &pv = sh.priv[wid];
  const u32 cnt_off = lds_offset (&pv.cntw[0]), gk_off = lds_offset (&pv.gk[0]), pp_off = lds_offset (&pv.pos[0]);
  const u32 rawk_off = lds_offset (&sh.rawk[0]), rawc_off = lds_offset (&sh.rawc[0]), tpos_off = lds_offset (&sh.tpos[0]);
  /* the kept records of the tile before, in output order (lane l: the wavefront's positions l, 64 + l, ..) */
  u64 okey[RW];
  u32 ocnt[RW];
  u32 oslots = 0; /* the records' places among the wavefront's kept ones, eight bits each */
  u32 okeep = 0, oprev_kept = 0;
  bool have_prev = false;
#pragma unroll
  for (int i = 0; i < RW; i++) {
    okey[i] = 0;
    ocnt[i] = 0;
  }
  u32 sb_epoch = 0;
  auto soft_barrier = [&] () { /* among the workers (slow tiles only) */
    sb_epoch++;
    sub_arrive (&sh.c_sb, lane);
    SUB_WAIT (c_sb, (u32) NSUB * sb_epoch);
  };
  /* the middle of a tile: the raw area is free once every worker has gathered; the next tile's records go there,
   * the fetch of the one after starts */
  auto mid = [&] () {
    SUB_WAIT (c_gath, (u32) NSUB * (it + 1u));
    PHASE_STAMP (5);
    const u32 nxt = uniform32 (sh.hdr[tb1][0]);
    if (nxt < ntl) {
      write_raw (tb1);
      sub_arrive (&sh.c_raw, lane);
      PHASE_STAMP (6);
      SUB_WAIT (c_tab, it + 3u);
      fetch_or_not (tb2);
    }
  };
  /* the tile before leaves the registers */
  auto write_out_prev = [&] () {
    if (MODE == NWAY_UNION && have_prev) {
      SUB_WAIT (c_x, it);
      PHASE_STAMP (9);
      const u64 base = uniform64 (sh.excl[(it - 1u) & 1u]) + (u64) uniform32 (sh.wpre[(it - 1u) & 1u][wid]);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) (out + 3 * base), 0, (int) (12 * uniform32 (oprev_kept)), 0x00020000); /* (said to be uniform: a waterfall loop around every store otherwise) */
#pragma unroll
      for (int i = 0; i < RW; i++) {
        if ((okeep >> i) & 1u) {
          const u32x3 rec = { (u32) okey[i], (u32) (okey[i] >> 32), ocnt[i] };
          __builtin_amdgcn_raw_buffer_store_b96 (rec, rs, 12 * ((oslots >> (8 * i)) & 0xffu), 0, 0);
        }
      }
    }
  };
  /* positions p0 .. p0 + c of the array at rb, in order: cutoff, ballots; the kept records stay in the registers */
  auto ordered = [&] (u32 rb, u32 p0, u32 c, bool publish) {
    u32 wave_kept = 0;
    okeep = 0;
    oslots = 0;
    static_assert (RW == 4 && SUB_CAPW <= 256, "four slot numbers below 256 share a dword");
    const u32 least = p.filter == FILTER_RAW ? 0u : p.cutoff;
#pragma unroll
    for (int i = 0; i < RW; i++) {
      if ((u32) (WAVE * i) >= c) break; /* uniform */
      const u32 q = (u32) (WAVE * i) + (u32) lane;
      const bool in = q < c;
      const u32x4 e = lds_load<u32x4> (rb + 16u * sub_phys (p0 + (in ? q : 0u)));
      const u32 f = p.rule == 7u ? p.count_override : e.z;
      const bool keep = in && e.w != 0u && f >= least;
      const u64 m = __builtin_amdgcn_ballot_w64 (keep);
      oslots |= ((wave_kept + __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u))) & 0xffu) << (8 * i);
      wave_kept += (u32) __popcll (m);
      okey[i] = (u64) e.x | ((u64) e.y << 32);
      ocnt[i] = f;
      okeep |= keep ? 1u << i : 0u;
      acc_sum += keep ? f : 0u;
    }
    oprev_kept = wave_kept;
    blk_cnt += wave_kept;
    have_prev = true;
    if (publish) { /* (slow tiles: the fast path knows what it keeps behind its folds) */
      if (lane == 0) sh.kept[it & 1u][wid] = wave_kept;
      sub_arrive (&sh.c_kept, lane);
    }
    /* (the caller arrives at c_end once its own area is ready for the next tile: a slow tile lies over it) */
  };
  /* {sum, arrivals} of a position += {count, 1}; the first arrival leaves the key */
  auto fold = [&] (u32 rb, u32 pos, bool valid, u64 key, u32 cnt, u32 spare) -> u64 {
    const u32 at = rb + 16u * sub_phys (valid ? pos : spare);
    return sub_lds_add_u64 (at + 8u, valid ? ((1ull << 32) | (u64) (p.rule == 1u ? cnt : 0u)) : 0ull);
  };

  for (;; it++) {
    /* the lane number is made opaque once per tile: addresses and masks derived from it are recomputed where they
     * are used instead of living in registers across the whole loop (hoisted by the compiler) */
    asm volatile ("" : "+v"(lane));
    const u32 cur = uniform32 (sh.hdr[tb][0]);
    if (cur >= ntl || dead) break;
    const u32 n = uniform32 (sh.hdr[tb][1]), slots = uniform32 (sh.hdr[tb][2]);
    if (n > (u32) SUB_MAXREC || slots > (u32) NCH) {
      if (lane == 0) atomicOr (&ctl->error, 2u);
      give_up ();
      break;
    }
    const u32 par = it & 1u;
    PHASE_STAMP (12);
    SUB_WAIT (c_cuts, it + 1u);
    PHASE_STAMP (0);
    if (dead) break;
    const bool slow_tile = uniform32 (sh.slow[par]) != 0u;

    if (__builtin_expect (!slow_tile, 1)) {
      /* ---- my sub-range: records [lo, hi) of every run */
      u32 lo_c = 0, hi_c = 0, pb = 0;
      if (lane < NWAY_MAX) {
        lo_c = sh.cuts[par][wid][lane];
        hi_c = sh.cuts[par][wid + 1][lane];
        pb = sh.tab_pbase[tb][lane];
      }
      const u32 len = hi_c - lo_c;
      const u32 incl = dpp_inclusive_scan_u32 (len);
      const u32 n_w = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
      const u32 pre_r = incl - len;
      if (lane < NWAY_MAX) pv.wtab[lane] = pb + lo_c - pre_r; /* raw position of lane slot j of run r: j + this */
      sub_fence ();
      u32 pf[NWAY_MAX];
#pragma unroll
      for (int r = 1; r < NWAY_MAX; r++) pf[r] = (u32) __builtin_amdgcn_readlane ((int) pre_r, r);
      /* ---- gather: lane slot j = 64 k + lane is record j of the sub-runs laid end to end */
      u64 key[RW];
      u32 cnt[RW], ba[RW]; /* ba: bucket | arrival number << 16 | valid << 31 */
      {
        u32 idx[RW], vmask[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) {
          const u32 j = (u32) (WAVE * k) + (u32) lane;
          u32 run = 0;
#pragma unroll
          for (int r = 1; r < NWAY_MAX; r++) run += j >= pf[r] ? 1u : 0u;
          idx[k] = lds_load<u32> (lds_offset (&pv.wtab[0]) + 4u * run);
          vmask[k] = j < n_w ? ~0u : 0u;
        }
#pragma unroll
        for (int k = 0; k < RW; k++) {
          const u32 j = (u32) (WAVE * k) + (u32) lane;
          idx[k] = (idx[k] + j) & vmask[k];
        }
#pragma unroll
        for (int k = 0; k < RW; k++) {
          key[k] = lds_load<u64> (rawk_off + 8u * idx[k]);
          cnt[k] = lds_load<u32> (rawc_off + 4u * idx[k]);
        }
      }
      /* the raw area is not looked at again (the arrival is seen behind the reads: in order) */
#ifdef GT4_SUB_LATEGATH
      asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      sub_arrive (&sh.c_gath, lane);
      PHASE_STAMP (1);
      /* ---- buckets: interpolation inside the sub-range's key range */
      const u64 k_lo = uniform64 (sh.cutkey[par][wid]) + 1ull, k_hi = uniform64 (sh.cutkey[par][wid + 1]);
      u32 bk_sh = 0, bk_mul = 0xffffffffu;
      {
        const u64 D = k_hi >= k_lo ? k_hi - k_lo : 0ull;
        const u32 bl = D ? 64u - (u32) __builtin_clzll (D) : 0u;
        bk_sh = bl > 32u ? bl - 32u : 0u;
        const u32 vmax = (u32) (D >> bk_sh);
        /* (a range below the number of buckets: the value itself, less one -- the same multiply, no branch) */
        if (vmax >= (u32) SUB_NBW) bk_mul = (u32) ((float) SUB_NBW * 4294967296.0f * 0.999999f * __builtin_amdgcn_rcpf ((float) vmax + 1.0f));
      }
      u32 mx = 0;
      u32 pos[RW]; /* position | valid << 31 */
      {
        u32 old[RW], b[RW], vmask[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) vmask[k] = (u32) (WAVE * k) + (u32) lane < n_w ? ~0u : 0u;
#pragma unroll
        for (int k = 0; k < RW; k++) {
          const u32 v = __umulhi ((u32) ((key[k] - k_lo) >> bk_sh), bk_mul);
          b[k] = v < (u32) SUB_NBW ? v : (u32) SUB_NBW - 1u;
        }
#pragma unroll
        for (int k = 0; k < RW; k++) /* (a lane without a record adds 0 to a word of its own) */
          old[k] = atomicAdd (&pv.cntw[nway_pick (vmask[k], b[k] >> 1, (u32) lane)], (1u << ((b[k] & 1u) * 16u)) & vmask[k]);
#pragma unroll
        for (int k = 0; k < RW; k++) ba[k] = (b[k] | (((old[k] >> ((b[k] & 1u) * 16u)) & 0x7fffu) << 16) | 0x80000000u) & vmask[k];
      }
      sub_fence ();
      PHASE_STAMP (2);
      /* ---- scan of the counters: four per lane */
      {
        const u64 w2 = lds_load<u64> (cnt_off + 8u * (u32) lane);
        const u32 c0 = (u32) w2 & 0xffffu, c1 = ((u32) w2) >> 16, c2 = (u32) (w2 >> 32) & 0xffffu, c3 = (u32) (w2 >> 48);
        const u32 tsum = c0 + c1 + c2 + c3;
        const u32 m01 = c0 > c1 ? c0 : c1, m23 = c2 > c3 ? c2 : c3;
        const u32 inc = dpp_inclusive_scan_u32 (tsum);
        mx = dpp_wave_max_u32 (m01 > m23 ? m01 : m23);
        const u32 s0 = inc - tsum, s1 = s0 + c0, s2 = s1 + c1, s3 = s2 + c2;
        *reinterpret_cast<u64 *> (&pv.cntw[2 * lane]) = (u64) (s0 | (s1 << 16)) | ((u64) (s2 | (s3 << 16)) << 32);
      }
      sub_fence ();
      PHASE_STAMP (3);
      const bool walk = mx <= (u32) SUB_LIMIT && p.force_fallback != 2u;
      if (__builtin_expect (walk, 1)) {
        /* ---- keys grouped by bucket, rank = bucket start + smaller keys in the bucket */
        u32 h0[RW], st[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) h0[k] = lds_load<unsigned short> (cnt_off + 2u * (ba[k] & 0xffffu));
#pragma unroll
        for (int k = 0; k < RW; k++) {
          st[k] = h0[k] & nway_valid_mask (ba[k]);
          if (ba[k] >> 31) pv.gk[nway_skew (st[k]) + ((ba[k] >> 16) & 0x7fffu)] = key[k];
        }
        sub_fence ();
        *reinterpret_cast<u64 *> (&pv.cntw[2 * lane]) = 0ull; /* the next tile's counters */
        sub_fence ();
        u32 lt[RW], ga[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) {
          lt[k] = 0;
          ga[k] = gk_off + 8u * nway_skew (st[k]);
        }
        static_assert (RW == 4, "the walks take four records at a time");
        nway_rank_steps<0> (mx, ga[0], ga[1], ga[2], ga[3], key, lt);
#pragma unroll
        for (int k = 0; k < RW; k++) pos[k] = (st[k] + lt[k]) | (ba[k] & 0x80000000u);
      } else {
        /* clustered keys: every record adds up its lower bounds in the worker's eight sub-runs -- which are the
         * worker's own records in the order it gathered them: lane slot j of run r is key prefix_r + j of its area */
        *reinterpret_cast<u64 *> (&pv.cntw[2 * lane]) = 0ull;
#pragma unroll
        for (int k = 0; k < RW; k++)
          if (ba[k] >> 31) pv.gk[(u32) (WAVE * k) + (u32) lane] = key[k];
        sub_fence ();
#pragma unroll
        for (int k = 0; k < RW; k++) pos[k] = 0;
        for (int r = 0; r < NWAY_MAX; r++) {
          const u32 len_r = (u32) __builtin_amdgcn_readlane ((int) len, r), start_r = (u32) __builtin_amdgcn_readlane ((int) pre_r, r);
          if (!len_r) continue;
          const u32 steps = 32u - (u32) __builtin_clz (len_r);
          u32 lo[RW], hi[RW];
#pragma unroll
          for (int k = 0; k < RW; k++) {
            lo[k] = 0;
            hi[k] = len_r;
          }
          for (u32 s = 0; s < steps; s++) {
#pragma unroll
            for (int k = 0; k < RW; k++) {
              const bool act = lo[k] < hi[k];
              const u32 mid_ = (lo[k] + hi[k]) >> 1;
              const u64 km = lds_load<u64> (gk_off + 8u * (start_r + (act ? mid_ : 0u)));
              const bool c = km < key[k];
              lo[k] = (act && c) ? mid_ + 1u : lo[k];
              hi[k] = (act && !c) ? mid_ : hi[k];
            }
          }
#pragma unroll
          for (int k = 0; k < RW; k++) pos[k] += lo[k];
        }
#pragma unroll
        for (int k = 0; k < RW; k++) pos[k] = (pos[k] & nway_valid_mask (ba[k])) | (ba[k] & 0x80000000u);
      }
      sub_fence ();
      PHASE_STAMP (4);
      /* ---- the next tile's records to the raw area, the fetch of the one after */
      mid ();
      PHASE_STAMP (7);
      /* ---- fold: the positions start from zero (the grouped keys lay there) */
      static_assert (SUB_PN <= 5 * WAVE, "the positions are zeroed in five rounds");
      {
        const u32 z = sub_const (0u);
#pragma unroll
        for (int r = 0; r < 5; r++)
          if (r * WAVE + lane < SUB_PN) pv.pos[r * WAVE + lane] = u32x4 { z, z, z, z };
      }
      sub_fence ();
      {
        u64 old[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) old[k] = fold (pp_off, pos[k] & 0x7fffffffu, (pos[k] >> 31) != 0u, key[k], cnt[k], (u32) lane);
        sub_fence ();
#pragma unroll
        for (int k = 0; k < RW; k++)
          if ((pos[k] >> 31) && (u32) (old[k] >> 32) == 0u) *reinterpret_cast<u64 *> (&pv.pos[sub_phys (pos[k] & 0x7fffffffu)]) = key[k];
        if (p.rule == 4u) {
#pragma unroll
          for (int k = 0; k < RW; k++)
            if (pos[k] >> 31) atomicMax (&reinterpret_cast<u32 *> (&pv.pos[sub_phys (pos[k] & 0x7fffffffu)])[2], cnt[k]);
        }
        sub_fence ();
        /* ---- what the worker keeps is known here already: the first arrivals look at the folded counts (this
         * wavefront's LDS operations complete in order).  The tile's total goes to the chained scan half a tile
         * before the records are put in order. */
        u32 fs[RW];
#pragma unroll
        for (int k = 0; k < RW; k++) fs[k] = lds_load<u32> (pp_off + 16u * sub_phys (pos[k] & 0x7fffffffu) + 8u);
        const u32 least = p.filter == FILTER_RAW ? 0u : p.cutoff;
        u32 kept_w = 0;
#pragma unroll
        for (int k = 0; k < RW; k++) {
          const u32 f = p.rule == 7u ? p.count_override : fs[k];
          kept_w += (u32) __popcll (__builtin_amdgcn_ballot_w64 ((pos[k] >> 31) && (u32) (old[k] >> 32) == 0u && f >= least));
        }
        if (lane == 0) sh.kept[it & 1u][wid] = kept_w;
        sub_arrive (&sh.c_kept, lane);
      }
      PHASE_STAMP (8);
      write_out_prev ();
      PHASE_STAMP (10);
      sub_fence ();
      ordered (pp_off, 0u, n_w, false);
      sub_fence ();
      PHASE_STAMP (11);
      /* the grouped keys of the next tile start from all-ones */
      static_assert (SUB_GKN % 2 == 0 && SUB_GKN / 2 <= 3 * WAVE, "the grouped keys are filled 16 bytes at a time, in three rounds");
      {
        const u32 o = sub_const (~0u);
#pragma unroll
        for (int r = 0; r < 3; r++)
          if (r * WAVE + lane < SUB_GKN / 2) *reinterpret_cast<u32x4 *> (&pv.gk[2 * (r * WAVE + lane)]) = u32x4 { o, o, o, o };
      }
      sub_arrive (&sh.c_end, lane);
    } else {
      /* ---- slow tile: positions by lower bounds in all the runs, folded in one array for the whole tile (it lies over
       * every worker's own area: nobody may still be working on the tile before) */
      if (it > 0) SUB_WAIT (c_end, (u32) NSUB * it);
#pragma unroll
      for (int i = 0; i < RW; i++) reinterpret_cast<u64 *> (&sh.tpos[sub_phys ((u32) (wid * SUB_CAPW + i * WAVE + lane))])[1] = 0ull;
      soft_barrier ();
      for (u32 s = (u32) wid; s < slots; s += (u32) NSUB) {
        const u32 q = s * WAVE + (u32) lane;
        const bool valid = (u32) lane < uniform32 (sh.slot_cnt[tb][s]);
        const u64 key = sh.rawk[q];
        const u32 cnt = sh.rawc[q];
        u32 pos = 0;
        for (u32 r = 0; r < p.k; r++) {
          const u32 pb_r = uniform32 (sh.tab_pbase[tb][r]), len_r = uniform32 (sh.tab_len[tb][r]);
          if (!len_r) continue;
          const u32 steps = 32u - (u32) __builtin_clz (len_r);
          u32 lo = 0, hi = len_r;
          for (u32 t = 0; t < steps; t++) {
            const bool act = lo < hi;
            const u32 mid_ = (lo + hi) >> 1;
            const u64 km = lds_load<u64> (rawk_off + 8u * (pb_r + (act ? mid_ : 0u)));
            const bool c = km < key;
            lo = (act && c) ? mid_ + 1u : lo;
            hi = (act && !c) ? mid_ : hi;
          }
          pos += lo;
        }
        sub_fence ();
        const u64 old = fold (tpos_off, pos, valid, key, cnt, (u32) lane);
        sub_fence ();
        if (valid && (u32) (old >> 32) == 0u) *reinterpret_cast<u64 *> (&sh.tpos[sub_phys (pos)]) = key;
        if (valid && p.rule == 4u) atomicMax (&reinterpret_cast<u32 *> (&sh.tpos[sub_phys (pos)])[2], cnt);
        sub_fence ();
      }
      sub_arrive (&sh.c_gath, lane);
      soft_barrier (); /* every record is folded */
      mid ();
      write_out_prev ();
      {
        const u32 p0 = (u32) (wid * SUB_CAPW);
        ordered (tpos_off, p0, n > p0 ? (n - p0 < (u32) SUB_CAPW ? n - p0 : (u32) SUB_CAPW) : 0u, true);
      }
      soft_barrier (); /* everybody has read its positions: the workers' own areas again */
      pv.cntw[lane] = 0;
      pv.cntw[WAVE + lane] = 0;
      for (int i = lane; i < SUB_GKN; i += WAVE) pv.gk[i] = ~0ull;
      sub_arrive (&sh.c_end, lane);
    }
    const int t0 = tb;
    tb = tb1;
    tb1 = tb2;
    tb2 = t0;
  }
  /* the last tile leaves the registers */
  if (!dead) write_out_prev ();
  SUB_STAMPS_OUT ();
  {
    const u64 v = wave_sum (acc_sum);
    if (lane == 0 && v) atomicAdd (&ctl->total_count[0], v);
    if (lane == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt);
  }
#undef SUB_WAIT
}
