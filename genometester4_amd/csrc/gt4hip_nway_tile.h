/* gt4hip_nway_tile.h -- N-way: the tile kernel k_nway_merge (K7).  Included by gt4hip_nway_body.h inside namespace
 * gt4::<anon>::km8 / km32; no include guard. */
/* ------------------------------------------------------------------ K7: the tile kernel */

#define GT4_NWAY_NT 1024
#define GT4_NWAY_RPT 4
#define GT4_NWAY_NBF 1
#define GT4_NWAY_SVPRIO 3
#define GT4_TABLE_STORE_AUX 0 /* cache policy of the count tables' row stores (see gt4hip_device.h) */
#define GT4_NWAY_ROWW 14336 /* words of the count tables' row area in LDS (56 KB: 159 of 160 KB with it; 12288: 1 - 2 % slower, 10240: 2 - 3 %) */
/* Round 4: the per-tile work that does not depend on the number of records is 58 % of a tile (time per tile against
 * samples per tile: 14.1 ns + 0.42 ns x G on 256 CUs, profiles/round4/r4_nway_experiments.log), part of it
 * instructions every one of the sixteen wavefronts executes.  Three cuts, each A/B-measured (31.4 -> 30.1 ms together;
 * GT4_NWAY_FILL / _SCAN4 / _LEAN = 0 restore the old forms):
 *   FILL   the grouped-key area is filled with all-ones once per tile (two 16-byte stores per thread, behind the
 *          walks of the previous tile) instead of every thread working out which skewed slots its buckets leave free
 *          (-1.0 ms);
 *   SCAN4  the bucket counters are scanned by four wavefronts (one per SIMD, eight words = sixteen counters per lane,
 *          16-byte LDS accesses) instead of sixteen (two words per lane): twelve wavefronts skip two DPP scans, two
 *          DPP maxima and their LDS traffic (-0.3 ms with FILL; +1.5 ms without it: sixteen slot tests per lane);
 *   LEAN   one DPP scan behind B6 instead of a scan and a sum. */
static_assert (GT4_NWAY_NT == 1024 && GT4_NWAY_RPT == 4 && GT4_NWAY_NBF == 1, "the scan of the bucket counters is written for 2048 counter words and sixteen wavefronts");

typedef u32 u32x3 __attribute__ ((ext_vector_type (3)));

__device__ __forceinline__ u32 dpp_wave_max_u32 (u32 v)
{
  auto mx = [] (u32 a, u32 b) { return a > b ? a : b; };
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x111, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x112, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x114, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x118, 0xf, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x142, 0xa, 0xf, false));
  v = mx (v, (u32) __builtin_amdgcn_update_dpp (0, (int) v, 0x143, 0xc, 0xf, false));
  return (u32) __builtin_amdgcn_readlane ((int) v, WAVE - 1);
}

/* buckets of a tile: about one per position, a multiple of 2048 (the counters are scanned 16 bytes at a time by 256 lanes) */
__host__ __device__ constexpr int nway_buckets (int positions) { return (positions + 2047) / 2048 * 2048; }

template <int NT, int RPT, int NBF, int MODE>
struct NwayShared {
  static constexpr int CAP = NT * RPT;     /* positions = records a tile may hold (runs rounded up to 64) */
  static constexpr int NCH = CAP / WAVE;   /* wave slots */
  static constexpr int NB = nway_buckets (NBF * CAP); /* buckets */
  static constexpr int NW = NT / WAVE;
  /* the tile in key order: key and folded count per position -- or, in the fallback, the records
   * as sorted runs (packed 12 bytes at their positions) */
  /* Records of one list in consecutive lanes lie about as many positions apart in the ordered tile as
   * there are lists: a power-of-two stride puts 32 lanes on 4 LDS banks.  Grouped keys and ordered tile
   * are therefore SKEWED: index i lives at i + i / 32 (nway_skew), which spreads every power-of-two
   * stride over all banks (measured before: 60 % of all LDS cycles were bank conflicts). */
  static constexpr int CAPS = CAP + CAP / 32;                 /* skewed positions */
  static constexpr int GSZ = (CAPS + NWAY_LIMIT + 5) & ~1;    /* grouped keys: + the longest bucket walk behind the last key */
  static constexpr bool LEAD = nway_lead (MODE);
  /* LEAD: the bitmap holds GT4_NWAY_LEAD_BITS positions per 32-bit word.  The 64 records of a wave-instruction come
   * from one sorted list and lie about as many positions apart as there are lists, so with 32 positions per word four
   * lanes claim bits of the SAME word in one atomic instruction, which the LDS serialises; fewer positions per word
   * make the claims cheaper and the scan behind B6 longer.  Measured (8 x 5e8 stride lists, tile kernel ms): 32 bits
   * 28.72, 16 bits 28.59, 8 bits 28.84, 4 bits 29.98; independent / genomic keys gain 1 - 2 % from 8 against 32. */
  static constexpr int LBP = GT4_NWAY_LEAD_BITS;              /* positions per bitmap word (a power of two) */
  static constexpr int LWL = (CAP / LBP + WAVE - 1) / WAVE;   /* bitmap words per lane of a scanning wavefront */
  static constexpr int LW = LWL * WAVE;                       /* ... bitmap words (padded) */
  union {
    struct {
      alignas (16) u64 skey[LEAD ? GSZ + WAVE : CAPS];        /* LEAD: the grouped keys live here (+ a row nobody reads: see the trash rows) */
      u32 scnt[CAPS];
    } s;
    u32 raw[3 * CAP];
  };
  alignas (16) u64 g_own[LEAD ? 2 : GSZ + WAVE]; /* keys grouped by bucket; all-ones wherever no key is */
  __device__ __forceinline__ u64 *g () { return LEAD ? s.skey : g_own; }
  /* TRASH ROWS: the per-record steps are straight-line code -- every LDS read of a thread's RPT records is issued
   * before the first one is waited for, no exec-mask bookkeeping, no branch between them -- so a lane whose record
   * is not there (or is not kept) does its store or atomic too, into a row of WAVE words / keys / records behind the
   * array, one per lane, that nobody reads */
  alignas (16) u32 cnt[NB / 2 + 4 + WAVE]; /* 16-bit bucket counters, then bucket starts, in pairs (+ the total) (+ a trash row) */
  alignas (16) u32 live[LEAD ? 4 : (CAPS + 3) / 4]; /* one byte per position: a key was stored there */
  alignas (16) u32 lead[2][LEAD ? LW : 4]; /* LEAD: the positions that have a (kept) leader, tiles alternating */
  alignas (16) unsigned short wpre[LEAD && nway_staged (MODE) ? NT / WAVE : 1][LEAD && nway_staged (MODE) ? LW : 4]; /* LEAD: kept leaders in front of every bitmap word, per wavefront */
  /* the kept records, packed, written out during the NEXT tile (+ a trash row); the count tables: ROWW words of the
   * tile's rows at a time (see table_rows) */
  static constexpr int ROWW = GT4_NWAY_ROWW;
  static constexpr int ROW_COLS_MAX = 384;  /* wider tables: rows straight to global memory, as before round 5 */
  alignas (16) u32 stage[nway_staged (MODE) ? 3 * CAP + 4 + 3 * WAVE + 8 : ((MODE == NWAY_TABLE || MODE == NWAY_PROBE) ? ROWW + 4 : 4)];
  alignas (16) u32 wtot[NW], wmax[NW], wkept[NW];
  /* the tiles of this iteration, the next one (being fetched) and the one after (being described),
   * three deep: one 64-record wave slot per wave-instruction */
#if GT4_KM > 8
  static constexpr int NSL = 4;            /* (many lists: no slot table) */
  static constexpr int NHM = 2 * NCH;      /* stretches of 32 positions */
  u64 hmask[3][NHM];                       /* per stretch: bit i = a run starts at its position i | runs that start in front of the stretch << 32 */
  u64 rtab[3][NWAY_MAX];                   /* per NON-EMPTY run, in order: address of list record (tile position 0 - first position of the run) */
  u32 rlist[3][NWAY_MAX];                  /* ... the list it is a run of */
#else
  static constexpr int NSL = NCH;
#endif
  u64 slot_addr[3][NSL];
  alignas (16) u32 slot_cnt[3][NSL];
  u32 slot_run[3][NSL];                    /* (NWAY_TABLE, NWAY_PROBE, NWAY_DUPS) the list a slot's records come from */
  u32 tab_pbase[3][NWAY_MAX];              /* first position of each run */
  u32 tab_len[3][NWAY_MAX];
  /* tile number (0xffffffff: none), records, wave slots, shift | direct << 8, multiplier, smallest
   * possible key (2), NWAY_DUPS: where the tile's output starts (2) */
  alignas (16) u32 hdr[3][12];
  u64 excl;
  u32 tick;
};

#define GT4_NWAY_WAVES 4
__host__ __device__ constexpr int nway_waves_per_simd (int nt) { return GT4_NWAY_WAVES; }

__device__ __forceinline__ u32 nway_skew (u32 i) { return i + (i >> 5); }
/* a where the mask is all ones, b where it is zero -- one bit-field insert; `c ? a : b` on a per-lane condition became
 * exec-mask bookkeeping (four scalar instructions each on the CU's one scalar unit) */
__device__ __forceinline__ u32 nway_pick (u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ u32 nway_valid_mask (u32 ba) { return (u32) ((int) ba >> 31); }

/* LDS accesses by byte offset through address-space-3 pointers: the compiler keeps generic pointers for
 * loop-invariant per-thread addresses otherwise (flat loads, two registers per address) */
typedef __attribute__ ((address_space (3))) u32 lds_u32;
typedef __attribute__ ((address_space (3))) u64 lds_u64;
typedef __attribute__ ((address_space (3))) unsigned char lds_u8;
template <class T> __device__ __forceinline__ u32 lds_offset (T *p) { return (u32) (uintptr_t) p; }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <class T> __device__ __forceinline__ T lds_load (u32 byte_offset) { return *(__attribute__ ((address_space (3))) T *) byte_offset; }
#pragma clang diagnostic pop

/* Two steps of four bucket walks: eight INDEPENDENT 8-byte reads, one wait, eight compares.  Inline
 * assembly because the compiler merges two reads of one walk into a ds_read2_b64 (twice the LDS cycles
 * of two ds_read_b64: MI355X_MICROARCH.md, LDS table) or, told not to (volatile), waits for every
 * single read.  The wait is part of the statement: the outputs are valid behind it. */
template <int J>
__device__ __forceinline__ void nway_rank_pair (u32 a0, u32 a1, u32 a2, u32 a3, const u64 (&key)[4], u32 (&lt)[4])
{
  u64 r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile ("ds_read_b64 %0, %8 offset:%12\n\t"
                "ds_read_b64 %1, %9 offset:%12\n\t"
                "ds_read_b64 %2, %10 offset:%12\n\t"
                "ds_read_b64 %3, %11 offset:%12\n\t"
                "ds_read_b64 %4, %8 offset:%13\n\t"
                "ds_read_b64 %5, %9 offset:%13\n\t"
                "ds_read_b64 %6, %10 offset:%13\n\t"
                "ds_read_b64 %7, %11 offset:%13\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(8 * J), "n"(8 * J + 8)
                : "memory");
  lt[0] += (r0 < key[0] ? 1u : 0u) + (r4 < key[0] ? 1u : 0u);
  lt[1] += (r1 < key[1] ? 1u : 0u) + (r5 < key[1] ? 1u : 0u);
  lt[2] += (r2 < key[2] ? 1u : 0u) + (r6 < key[2] ? 1u : 0u);
  lt[3] += (r3 < key[3] ? 1u : 0u) + (r7 < key[3] ? 1u : 0u);
}

template <int J>
__device__ __forceinline__ void nway_rank_steps (u32 mx, u32 a0, u32 a1, u32 a2, u32 a3, const u64 (&key)[4], u32 (&lt)[4])
{
  if ((u32) J >= mx) return; /* uniform */
  nway_rank_pair<J> (a0, a1, a2, a3, key, lt);
  if constexpr (J + 2 < NWAY_LIMIT) nway_rank_steps<J + 2> (mx, a0, a1, a2, a3, key, lt);
}

__device__ __forceinline__ u64 readlane_u64 (u64 v, int l)
{
  return (u64) (u32) __builtin_amdgcn_readlane ((int) (u32) v, l) | ((u64) (u32) __builtin_amdgcn_readlane ((int) (u32) (v >> 32), l) << 32);
}

template <int NT, int RPT, int NBF, int MODE>
__global__ __launch_bounds__ (NT, nway_waves_per_simd (NT)) void
k_nway_merge (NwayParams p, const u64 *__restrict__ part, u32 *__restrict__ out, u64 *desc, PairControl *ctl)
{
  typedef NwayShared<NT, RPT, NBF, MODE> Shared;
  constexpr int CAP = Shared::CAP, NW = Shared::NW, NCH = Shared::NCH, NB = Shared::NB;
  constexpr int NWORDS = NB / 2, WPT = NWORDS / NT;
  constexpr int CAPS = Shared::CAPS;
  constexpr bool LEAD = Shared::LEAD;
  static_assert (NCH <= 2 * WAVE, "one lane per wave slot builds the slot table, in two rounds at most");
  static_assert (WPT * NT == NWORDS && WPT >= 1, "every thread scans the same number of counter words");
  static_assert (NW <= 16 && NW >= 2, "wave totals are reduced by one DPP row");
  static_assert (NWAY_LIMIT % 2 == 0 && NWAY_TRY0 <= NWAY_LIMIT, "bucket walks go two steps at a time");
  static_assert (CAP <= 32767 && NB <= 65536, "16-bit bucket counters and starts; bucket, arrival number and a flag share a dword");
  static_assert (NWAY_MAX > 8 || 2 * NWAY_PSTRIDE <= WAVE, "one lane per partition entry of a tile");
  static_assert (NWAY_PSTRIDE <= WAVE, "many lists: one lane per entry of ONE partition row");
  __shared__ Shared sh;
  int tid = threadIdx.x, lane = tid & (WAVE - 1); /* (not const: see the top of the tile loop) */
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
      /* one wavefront sums and chains the rows while the launch has few enough of them (one hop less
       * between a tile's total and its successors' offsets: with a single staging area the chain's
       * latency bounds the time per tile); summers + chainer beyond that */
      const u32 n_sub = p.scan_group ? (NW < 8 ? (u32) NW : 8u) : 1u;
      if ((u32) wid < n_sub) scanner_part (agg, carry + 4 * (n_rows + 1), carry, p.num_tiles, ctl, lane, spin_limit, (u32) wid, n_sub);
      return;
    }
  }
  const u32 n_workers = MODE == NWAY_UNION ? gridDim.x - 1 : gridDim.x;
  const u32 wk = MODE == NWAY_UNION ? role - 1 : blockIdx.x;
  const u32 ntl = p.num_tiles;
  /* The LAST wavefront is the service wavefront.  Wave slots are dealt to the wavefronts in order (RPT
   * consecutive slots each), a tile fills 85 % of them on average, so the last wavefront usually has
   * no records and does what must not sit in front of everybody's barrier: tile numbers (by ticket or
   * round-robin), partition entries, the slot table two tiles ahead (from registers: no LDS round
   * trips), the chain words of the tile being written out, the publication of the tile total. */
  const bool service = wid == NW - 1;
  /* the SIMD issues oldest-first and the last wavefront is the youngest of its SIMD: without a raised
   * priority its few instructions crawl behind three ranking wavefronts (measured: 5.7 k cycles for the
   * slot table alone) and everybody waits for it at the next barrier */
  if (service) __builtin_amdgcn_s_setprio (GT4_NWAY_SVPRIO);
  auto deal = [&] (int j) -> u32 { /* lane 0 of the service wavefront */
    if (p.dynamic) {
      const u32 t = atomicAdd (&ctl->ticket, 1u);
      return t < ntl ? t : 0xffffffffu;
    }
    const u64 t = (u64) wk + (u64) j * n_workers;
    return t < (u64) ntl ? (u32) t : 0xffffffffu;
  };
#if GT4_KM > 8
  /* (many lists: a partition row is KM + 2 entries -- one lane per entry of the tile's START row; the END row is the
   * next tile's start row, asked for with it) */
  struct RowPair { u64 a, b; };
  auto load_row = [&] (u32 tile) -> RowPair {
    RowPair v = { 0, 0 };
    if (tile < ntl && lane < NWAY_PSTRIDE) {
      v.a = __hip_atomic_load (&part[(u64) tile * NWAY_PSTRIDE + (u64) lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v.b = __hip_atomic_load (&part[((u64) tile + 1) * NWAY_PSTRIDE + (u64) lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
  };
  /* the table of `tile`: the runs end to end in the position space, a mask of run starts per stretch of 32 positions */
  auto build_table = [&] (RowPair row, u32 tile, int tb) {
    if (tile >= ntl) {
      if (lane == 0) sh.hdr[tb][0] = 0xffffffffu;
      return;
    }
    const u64 lbv = (u32) lane < p.k ? (u64) p.list[lane < NWAY_MAX ? lane : 0] : 0ull; /* lane q: base address of list q */
    const u32 len = (u32) lane < p.k ? (u32) row.b - (u32) row.a : 0u;
    const u32 incl = dpp_inclusive_scan_u32 (len), excl = incl - len; /* excl: first position of the run */
    const u32 n = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
    if (lane < NWAY_MAX) {
      sh.tab_pbase[tb][lane] = excl;
      sh.tab_len[tb][lane] = len;
    }
    /* non-empty runs, in order */
    const u64 nz = __builtin_amdgcn_ballot_w64 (len != 0u);
    const u32 rank = __builtin_amdgcn_mbcnt_hi ((u32) (nz >> 32), __builtin_amdgcn_mbcnt_lo ((u32) nz, 0u));
    u32 *const hm32 = reinterpret_cast<u32 *> (&sh.hmask[tb][0]);
    static_assert (Shared::NHM == 2 * WAVE, "two stretch masks per lane are zeroed and scanned");
    hm32[2 * lane] = 0;                /* (the stretches' masks; 2 x 64 = NHM of them) */
    hm32[2 * (lane + WAVE)] = 0;
    asm volatile ("" ::: "memory");
    if (len) {
      sh.rtab[tb][rank] = lbv + 12ull * row.a - 12ull * (u64) excl;
      sh.rlist[tb][rank] = (u32) lane;
      atomicOr (&hm32[2 * (excl >> 5)], 1u << (excl & 31u));
    }
    asm volatile ("" ::: "memory");
    {
      const u32 m0 = hm32[2 * lane], m1 = hm32[2 * (lane + WAVE)];
      const u32 c0 = (u32) __popc (m0), c1 = (u32) __popc (m1);
      const u32 i0 = dpp_inclusive_scan_u32 (c0);
      const u32 t0 = (u32) __builtin_amdgcn_readlane ((int) i0, WAVE - 1);
      const u32 i1 = dpp_inclusive_scan_u32 (c1);
      hm32[2 * lane + 1] = i0 - c0;
      hm32[2 * (lane + WAVE) + 1] = t0 + i1 - c1;
    }
    u64 base = 0;
    /* the records in front of the tile: where a level of merged samples starts its output, and a count table the tile's rows */
    if (MODE == NWAY_DUPS || MODE == NWAY_TABLE) base = wave_sum ((u32) lane < p.k ? row.a : 0ull);
    if (MODE == NWAY_PROBE) base = readlane_u64 (row.a, 0); /* the tile's first record of list 0 = its first row */
    const u32 rlo = (u32) row.a, rhi = (u32) (row.a >> 32);
    const u32 lo_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX), lo_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX);
    const u32 bk_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX + 1), bk_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX + 1);
    /* the longest run (the lowest list among equals): one wave maximum over length << 6 | 63 - lane */
    const u32 best = dpp_wave_max_u32 ((len << 6) | (63u - (u32) lane));
    const u32 pl = 63u - (best & 63u);
    const u32 pv_len = best >> 6, pv_base = (u32) __builtin_amdgcn_readlane ((int) excl, (int) pl);
    u32 h = tile;
    h = lane == 1 ? n : h;
    h = lane == 2 ? (n + (u32) WAVE - 1u) / (u32) WAVE : h; /* wave slots */
    h = lane == 3 ? bk_lo : h;
    h = lane == 4 ? bk_hi : h;
    h = lane == 5 ? lo_lo : h;
    h = lane == 6 ? lo_hi : h;
    h = lane == 7 ? (u32) base : h;
    h = lane == 8 ? (u32) (base >> 32) : h;
    h = lane == 9 ? pv_base : h;
    h = lane == 10 ? pv_len : h;
    if (lane < 11) sh.hdr[tb][lane] = h;
  };
#else
  auto load_row = [&] (u32 tile) -> u64 { /* lane i: entry i of the tile's two partition rows (its start and its end) */
    u64 v = 0;
    if (tile < ntl && lane < 2 * NWAY_PSTRIDE)
      v = __hip_atomic_load (&part[(u64) tile * NWAY_PSTRIDE + (u64) lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  /* the slot table of `tile` (partition entries in `row`, one per lane) into table tb: branch-free
   * vector code -- run lengths by a lane shift, their prefix by a DPP scan, the run of a slot by
   * eight compares against broadcast prefixes, the run's data by lane permutes */
  auto build_table = [&] (u64 row, u32 tile, int tb) {
    if (tile >= ntl) {
      if (lane == 0) sh.hdr[tb][0] = 0xffffffffu;
      return;
    }
    u64 lbv = 0; /* lane q: base address of list q (rebuilt here: two registers less in every wavefront's loop) */
#pragma unroll
    for (int m = 0; m < NWAY_MAX; m++) lbv = lane == m ? (u64) p.list[m] : lbv;
    const u32 rlo = (u32) row, rhi = (u32) (row >> 32);
    const u32 elo = __shfl_down (rlo, NWAY_PSTRIDE, WAVE);
    const u32 len = (u32) lane < p.k ? elo - rlo : 0u; /* (p.k <= 8; a run is shorter than 2^32 records) */
    const u32 nw = (len + WAVE - 1) / WAVE;
    const u32 incl = dpp_inclusive_scan_u32 (nw), excl = incl - nw;
    const u32 total = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
    const u32 n = dpp_wave_sum_u32 (len);
    if (lane < NWAY_MAX) {
      sh.tab_pbase[tb][lane] = excl * WAVE;
      sh.tab_len[tb][lane] = len;
    }
#pragma unroll
    for (int sb = 0; sb < NCH; sb += WAVE) { /* one lane per wave slot (two rounds when a tile has more than 64) */
      const u32 slot = (u32) (sb + lane);
      u32 run = 0;
#pragma unroll
      for (int q = 0; q < NWAY_MAX - 1; q++) run += slot >= (u32) __builtin_amdgcn_readlane ((int) incl, q) ? 1u : 0u;
      const u32 len_r = __shfl (len, run, WAVE), excl_r = __shfl (excl, run, WAVE);
      const u64 s_r = (u64) __shfl (rlo, run, WAVE) | ((u64) __shfl (rhi, run, WAVE) << 32);
      const u64 lb_r = (u64) __shfl ((u32) lbv, run, WAVE) | ((u64) __shfl ((u32) (lbv >> 32), run, WAVE) << 32);
      const bool in = slot < total;
      const u32 first = in ? (slot - excl_r) * WAVE : 0u;
      if (slot < (u32) NCH) {
        sh.slot_cnt[tb][slot] = in ? (len_r - first < (u32) WAVE ? len_r - first : (u32) WAVE) : 0u;
        sh.slot_addr[tb][slot] = lb_r + 12ull * (s_r + first);
        if (MODE == NWAY_TABLE || MODE == NWAY_PROBE || MODE == NWAY_DUPS) sh.slot_run[tb][slot] = run;
      }
    }
    u64 base = 0;
    if (MODE == NWAY_DUPS || MODE == NWAY_TABLE) {
      /* the records in front of the tile: where a level of merged samples starts the tile's output -- and where
       * the count table starts the tile's rows (a tile has at most as many distinct keys as records: the table is
       * RAGGED, see gt4hip_count_table) */
#pragma unroll
      for (int q = 0; q < NWAY_MAX; q++) base += (u32) q < p.k ? readlane_u64 (row, q) : 0ull;
    }
    if (MODE == NWAY_PROBE) base = readlane_u64 (row, 0); /* the tile's first record of list 0 = its first row */
    const u32 lo_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX), lo_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX);
    const u32 bk_lo = (u32) __builtin_amdgcn_readlane ((int) rlo, NWAY_MAX + 1), bk_hi = (u32) __builtin_amdgcn_readlane ((int) rhi, NWAY_MAX + 1);
    u32 h = tile;
    h = lane == 1 ? n : h;
    h = lane == 2 ? total : h;
    h = lane == 3 ? bk_lo : h;
    h = lane == 4 ? bk_hi : h;
    h = lane == 5 ? lo_lo : h;
    h = lane == 6 ? lo_hi : h;
    h = lane == 7 ? (u32) base : h;
    h = lane == 8 ? (u32) (base >> 32) : h;
    /* the longest run: the pivot of the second bucketing attempt (first position, records) */
    u32 pv_len = 0, pv_base = 0;
#pragma unroll
    for (int q = 0; q < NWAY_MAX; q++) {
      const u32 lq = (u32) __builtin_amdgcn_readlane ((int) len, q), bq = (u32) __builtin_amdgcn_readlane ((int) excl, q) * WAVE;
      const bool better = lq > pv_len; /* uniform */
      pv_base = better ? bq : pv_base;
      pv_len = better ? lq : pv_len;
    }
    h = lane == 9 ? pv_base : h;
    h = lane == 10 ? pv_len : h;
    if (lane < 11) sh.hdr[tb][lane] = h;
  };

#endif /* GT4_KM > 8 */

  /* The tile's records, fetched one tile ahead into registers.  A wavefront fetches 64 consecutive
   * records of ONE run per instruction, so descriptor and addresses are scalar and the range-checked
   * descriptor zero-fills past the run's end: no per-lane bounds. */
  u32x3 pre[RPT];
#if GT4_KM > 8
  /* (many lists: position p = 64 x chunk + lane of the tile is a record of the run whose start is the last one at or
   * in front of p -- the stretch's mask of run starts and the number of starts in front of the stretch come in ONE
   * 64-bit LDS read, the run's address entry in another; positions behind the tile's last record are not loaded) */
  auto run_of = [&] (int tb, u32 pos) -> u32 { /* the non-empty run position `pos` belongs to, counted from 0 */
    const u64 hm = sh.hmask[tb][pos >> 5];
    return (u32) (hm >> 32) + (u32) __popc ((u32) hm & ((2u << (pos & 31u)) - 1u)) - 1u;
  };
  auto fetch = [&] (int tb) {
    const u32 n_t = uniform32 (sh.hdr[tb][1]);
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const u32 pos = (u32) (wid * RPT + k) * WAVE + (u32) lane;
      if (pos < n_t) {
        const u64 addr = sh.rtab[tb][run_of (tb, pos)] + 12ull * pos;
        pre[k] = *reinterpret_cast<const u32x3 *> (addr); /* (plain: non-temporal per-lane loads measured 0.4 % slower, r5_cache_policy.log) */
      }
    }
  };
#else
  auto fetch = [&] (int tb) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int chunk = wid * RPT + k;
      const u64 addr = uniform64 (sh.slot_addr[tb][chunk]);
      const u32 c = uniform32 (sh.slot_cnt[tb][chunk]);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) addr, 0, (int) (12 * c), 0x00020000);
      /* (non-temporal where the union streams its records; the count tables' launches measured 9 % SLOWER with it -- 6.2
       * against 5.7 ms on six lists -- and keep plain loads: profiles/round5/r5_cache_policy.log) */
      pre[k] = __builtin_amdgcn_raw_buffer_load_b96 (rs, 12 * lane, 0, (MODE == NWAY_TABLE || MODE == NWAY_PROBE) ? 0 : GT4_LOAD_AUX);
    }
  };
#endif

  /* ---- prologue: tiles of iterations 0 .. 3, tables of the first two, entries of the third */
  u32 sv_t2 = 0xffffffffu; /* service wavefront: tile of iteration it + 2 (uniform) */
  u32 sv_tk = 0xffffffffu; /* ... of iteration it + 3, in lane 0 (a ticket drawn one iteration ago) */
#if GT4_KM > 8
  RowPair sv_row = { 0, 0 }; /* partition entries of tile sv_t2, asked for one iteration ago */
#else
  u64 sv_row = 0;          /* partition entries of tile sv_t2, asked for one iteration ago */
#endif
  if (service) {
    u32 d[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
    if (lane == 0)
      for (int q = 0; q < 4; q++) d[q] = deal (q);
    const u32 d0 = uniform32 (d[0]), d1 = uniform32 (d[1]);
    sv_t2 = uniform32 (d[2]);
    sv_tk = d[3];
    const auto r0 = load_row (d0), r1 = load_row (d1);
    sv_row = load_row (sv_t2);
    build_table (r0, d0, 0);
    build_table (r1, d1, 1);
  }
#pragma unroll
  for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = 0;
  auto fill_g = [&] () { /* all-ones wherever no key is: what a bucket walk meets behind its bucket must not be smaller than any key */
    static_assert (Shared::GSZ % 2 == 0, "the grouped keys are filled 16 bytes at a time");
#pragma unroll
    for (int r = 0; r < Shared::GSZ / 2 / NT; r++) *reinterpret_cast<u32x4 *> (&sh.g ()[2 * (r * NT + tid)]) = u32x4 { ~0u, ~0u, ~0u, ~0u };
    if (tid < Shared::GSZ / 2 - Shared::GSZ / 2 / NT * NT) *reinterpret_cast<u32x4 *> (&sh.g ()[2 * (Shared::GSZ / 2 / NT * NT + tid)]) = u32x4 { ~0u, ~0u, ~0u, ~0u };
  };
  fill_g ();
  if (MODE == NWAY_TABLE || MODE == NWAY_PROBE) /* the row area of the count tables starts as zeros (see table_rows) */
    for (int c = 4 * tid; c < Shared::ROWW + 4; c += 4 * NT) *reinterpret_cast<u32x4 *> (&sh.stage[c]) = u32x4 { 0, 0, 0, 0 };
  __syncthreads ();
  if (uniform32 (sh.hdr[0][0]) < ntl && (u32) (wid * RPT) < uniform32 (sh.hdr[0][2])) fetch (0);

  u64 acc_sum = 0; /* per-thread sum of kept counts */
  u64 blk_cnt = 0; /* records kept (the same in every thread) */
  u32 pend_tot = 0, pend_tile = 0;
  u64 pend_base = 0;
  bool pend = false;
  int it = 0;
  int tb = 0, tb1 = 1, tb2 = 2; /* tables of this tile, the next, the one after */
  /* the list record k of this thread comes from (many lists: per lane; else the same for the whole wave slot) */
  auto list_of = [&] (int k) -> u32 {
#if GT4_KM > 8
    return sh.rlist[tb][run_of (tb, (u32) (wid * RPT + k) * WAVE + (u32) lane)];
#else
    return uniform32 (sh.slot_run[tb][wid * RPT + k]);
#endif
  };
  /* The count tables' rows leave through LDS (round 5): as many of the tile's rows as fit ROWW words are zeroed there, the
   * records drop their counts in, and the rows go out whole, 16 bytes per lane -- instead of zeros stored to global memory
   * and 4-byte stores scattered over them (32 columns: 67 GB written for 46 GB of table, half the rows went to HBM twice).
   * row[k]: the tile's row record k belongs in (anything >= n_rows: none). */
  auto table_rows = [&] (const u32 (&row)[RPT], const u32 (&val)[RPT], u32 n_rows, u64 first_row) {
    const u32 cols = p.table_cols;
    const u32 rb = (u32) Shared::ROWW / cols;
    u32 col[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) col[k] = p.table_col[list_of (k)];
    /* (the row area is all zeros here: zeroed once in front of the first tile, and every thread zeroes the 16 bytes it has
     * just sent out -- no zeroing pass and no barrier in front of the records' stores) */
    for (u32 r0 = 0; r0 < n_rows; r0 += rb) {
      const u32 nr = n_rows - r0 < rb ? n_rows - r0 : rb, words = nr * cols;
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const u32 rr = row[k] - r0;
        if (rr < nr) sh.stage[rr * cols + col[k]] = val[k];
      }
      __syncthreads ();
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc ((void *) (p.table_counts + (first_row + r0) * cols), 0, (int) (4 * words), 0x00020000);
      for (u32 c = (u32) tid; 4u * c < words; c += NT) {
        u32x4 *const q = reinterpret_cast<u32x4 *> (&sh.stage[4u * c]);
        __builtin_amdgcn_raw_buffer_store_b128 (*q, rs, 16 * c, 0, GT4_TABLE_STORE_AUX);
        *q = u32x4 { 0, 0, 0, 0 };
      }
      if (r0 + rb < n_rows) __syncthreads (); /* (the next rows' counts go where these lay) */
    }
  };
PROF (
  u64 ph[24];
  for (int i = 0; i < 24; i++) ph[i] = 0;
  u64 t_last;
  asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last) :: "memory");
)

  for (;;) {
    /* the thread number is made opaque once per tile: addresses and masks derived from it are then
     * recomputed where they are used (a few VALU each) instead of living in ~25 registers across the
     * whole loop (hoisted by the compiler), which had the rank walk's registers spill */
    asm volatile ("" : "+v"(tid));
    lane = tid & (WAVE - 1);
    PHASE_STAMP (23); /* (diagnostics: the back edge) */
    u32 cur, n, slots, bk0, bk_mul;
    u64 key_lo, out_base;
    {
      const u32x4 h0 = *reinterpret_cast<const u32x4 *> (&sh.hdr[tb][0]), h1 = *reinterpret_cast<const u32x4 *> (&sh.hdr[tb][4]);
      const u32 h8 = sh.hdr[tb][8];
      cur = uniform32 (h0.x);
      n = uniform32 (h0.y);
      slots = uniform32 (h0.z);
      bk0 = uniform32 (h0.w);
      bk_mul = uniform32 (h1.x);
      key_lo = (u64) uniform32 (h1.y) | ((u64) uniform32 (h1.z) << 32);
      out_base = (u64) uniform32 (h1.w) | ((u64) uniform32 (h8) << 32);
    }
    if (cur >= ntl) break;
    if (n > (u32) CAP || slots > (u32) NCH) {
      if (tid == 0) atomicOr (&ctl->error, 2u);
      break;
    }
    const u32 bk_sh = bk0 & 0xffu;
    const bool bk_direct = (bk0 >> 8) & 1u;
    const bool has_rec = (u32) (wid * RPT) < slots;          /* this wavefront holds records of the tile */
    const bool has_pos = (u32) (wid * RPT * WAVE) < n;       /* ... positions of the ordered tile */

    /* ---- phase 0: the prefetched records leave the fetch registers */
    u64 key[RPT];
    u32 cnt[RPT], ba[RPT]; /* ba: bucket | arrival number << 16 | valid << 31 */
#pragma unroll
    for (int k = 0; k < RPT; k++) ba[k] = 0;
    /* (a wavefront without records of this tile keeps whatever the registers hold: its lanes are not valid) */
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      key[k] = (u64) pre[k].x | ((u64) pre[k].y << 32);
      cnt[k] = pre[k].z;
    }
    PHASE_STAMP (19); /* (diagnostics: the tile's header) */
PROF (
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); /* (the diagnostics build takes the wait for the prefetched records here) */
)
    PHASE_STAMP (20); /* (diagnostics: the wait for the prefetched records) */
    u32 xagg = 0;
    u64 xcarry = 0;
    u32 st[RPT];
    u32 mx = 0;
    bool accepted = false; /* the buckets of the last attempt are walked (else: the search path) */
    /* one bucketing pass over bucket numbers bk[]: count (arrival numbers), scan, group the keys */
    auto count_pass = [&] (const u32 (&bk)[RPT]) {
      if (has_rec) {
        u32 c[RPT], old[RPT];
#if GT4_KM > 8
        const u32 lane_ = (u32) lane; /* (many lists: a position holds a record iff it lies in front of the tile's end) */
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const u32 first = (u32) (wid * RPT + k) * WAVE;
          c[k] = n > first ? n - first : 0u;
        }
#else
        const u32 lane_ = (u32) lane;
        if constexpr (RPT == 4) { /* (the wavefront's four slots: one 16-byte read; compared per lane, no scalar copy) */
          const u32x4 c4 = *reinterpret_cast<const u32x4 *> (&sh.slot_cnt[tb][wid * RPT]);
          c[0] = c4.x, c[1] = c4.y, c[2] = c4.z, c[3] = c4.w;
        } else {
#pragma unroll
          for (int k = 0; k < RPT; k++) c[k] = sh.slot_cnt[tb][wid * RPT + k];
        }
#endif
#pragma unroll
        for (int k = 0; k < RPT; k++) { /* the atomics one behind the other: one wait for all of them */
          const u32 b = bk[k] < (u32) NB ? bk[k] : (u32) NB - 1u;
          const u32 vm = lane_ < c[k] ? ~0u : 0u;
          old[k] = atomicAdd (&sh.cnt[nway_pick (vm, b >> 1, (u32) (NB / 2 + 4) + (u32) lane)], 1u << ((b & 1u) * 16u));
        }
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const u32 b = bk[k] < (u32) NB ? bk[k] : (u32) NB - 1u;
          ba[k] = lane_ < c[k] ? b | (((old[k] >> ((b & 1u) * 16u)) & 0x7fffu) << 16) | 0x80000000u : 0u;
        }
      }
    };
    auto bucket_pass = [&] (u32 limit) {
      PHASE_STAMP (0);
      __syncthreads (); /* B1: every record is counted */
      PHASE_STAMP (1);

      {
        /* ---- scan of the bucket counters by the first four wavefronts (one per SIMD: a workgroup's wavefronts go to
         * the SIMDs in turn): WPL words = 2 WPL counters per lane, read and written 16 bytes at a time */
        constexpr int WPL = NWORDS / (4 * WAVE);
        u32 ex[2 * WPL];
        u32 tsum = 0, incl = 0;
        const int sl = (wid & 3) * WAVE + lane; /* the lane's place among the 256 scanning lanes */
        if (wid < 4) {
          u32 tmax = 0;
          u32 w[WPL];
#pragma unroll
          for (int i = 0; i < WPL / 4; i++) {
            const u32x4 q = *reinterpret_cast<const u32x4 *> (&sh.cnt[sl * WPL + 4 * i]);
            w[4 * i] = q.x;
            w[4 * i + 1] = q.y;
            w[4 * i + 2] = q.z;
            w[4 * i + 3] = q.w;
          }
#pragma unroll
          for (int i = 0; i < WPL; i++) {
            const u32 a = w[i] & 0xffffu, b = w[i] >> 16;
            ex[2 * i] = tsum;
            tsum += a;
            ex[2 * i + 1] = tsum;
            tsum += b;
            tmax = a > tmax ? a : tmax;
            tmax = b > tmax ? b : tmax;
          }
          incl = dpp_inclusive_scan_u32 (tsum);
          const u32 wmx = dpp_wave_max_u32 (tmax);
          if (lane == WAVE - 1) {
            sh.wtot[wid] = incl;
            sh.wmax[wid] = wmx;
          }
        }
        PHASE_STAMP (2);
        __syncthreads (); /* B2: the four wavefronts' totals */
        PHASE_STAMP (3);
        if (wid < 4) {
          const u32x4 t4 = *reinterpret_cast<const u32x4 *> (&sh.wtot[0]);
          const u32 wbase = (wid > 0 ? t4.x : 0u) + (wid > 1 ? t4.y : 0u) + (wid > 2 ? t4.z : 0u); /* (wid is uniform: scalar selects) */
          const u32 tbase = wbase + incl - tsum;
#pragma unroll
          for (int i = 0; i < WPL / 4; i++) {
            u32x4 q;
            q.x = (tbase + ex[8 * i]) | ((tbase + ex[8 * i + 1]) << 16);
            q.y = (tbase + ex[8 * i + 2]) | ((tbase + ex[8 * i + 3]) << 16);
            q.z = (tbase + ex[8 * i + 4]) | ((tbase + ex[8 * i + 5]) << 16);
            q.w = (tbase + ex[8 * i + 6]) | ((tbase + ex[8 * i + 7]) << 16);
            *reinterpret_cast<u32x4 *> (&sh.cnt[sl * WPL + 4 * i]) = q;
          }
          if (sl == 4 * WAVE - 1) sh.cnt[NWORDS] = tbase + tsum; /* start of the bucket behind the last = the tile's records */
        }
      }
      PHASE_STAMP (4);
      __syncthreads (); /* B3: bucket starts */
      PHASE_STAMP (5);
      {
        const u32x4 m4 = *reinterpret_cast<const u32x4 *> (&sh.wmax[0]);
        const u32 m01 = m4.x > m4.y ? m4.x : m4.y, m23 = m4.z > m4.w ? m4.z : m4.w;
        mx = uniform32 (m01 > m23 ? m01 : m23);
      }

      /* ---- the keys grouped by bucket */
#pragma unroll
      for (int k = 0; k < RPT; k++) st[k] = 0;
      accepted = mx <= limit;
      if (has_rec && accepted) {
        u32 w0[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) w0[k] = sh.cnt[(ba[k] & 0xffffu) >> 1]; /* (word 0 where no record is) */
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const u32 b = ba[k] & 0xffffu;
          const u32 s = (b & 1u) ? w0[k] >> 16 : w0[k]; /* start of the bucket */
          const u32 vm = nway_valid_mask (ba[k]);
          st[k] = s & 0xffffu & vm;
          sh.g ()[nway_pick (vm, nway_skew (st[k]) + ((ba[k] >> 16) & 0x7fffu), (u32) Shared::GSZ + (u32) lane)] = key[k]; /* a bucket's keys stay together */
        }
      }
      PHASE_STAMP (6);
      __syncthreads (); /* B4: keys grouped */
      PHASE_STAMP (7);
#pragma unroll
      for (int i = 0; i < WPT; i++) sh.cnt[tid * WPT + i] = 0; /* the next pass's / the next tile's counters */
    };

    /* ---- buckets.  First by interpolation inside the tile's key range (no search at all).  If a bucket
     * then holds more than NWAY_TRY0 keys -- clustered keys: stretches of adjacent keys with wide gaps
     * put a whole stretch into one bucket -- the tile is bucketed again, by RANK IN ITS LONGEST RUN (one
     * binary search per record in that run's keys, copied to LDS): a bucket then holds what the other
     * runs have between two neighbours of the pivot run, whatever the keys' values.  Only a tile that
     * defeats that too (more than NWAY_LIMIT keys in a bucket) takes the full search path below. */
    const bool pivot_first = ((bk0 >> 9) & 1u) && p.force_fallback == 0; /* the partition found the tile's samples clustered */
    auto pivot_buckets = [&] (u32 (&bk)[RPT]) {
      const u32 pv_base = uniform32 (sh.hdr[tb][9]), pv_len = uniform32 (sh.hdr[tb][10]);
      /* the pivot run's keys to LDS, in order (its records sit at positions pv_base ..; sh.s.skey is free until the fold) */
      if (has_rec) {
#pragma unroll
        for (int k = 0; k < RPT; k++) {
#if GT4_KM > 8
          const bool here = (u32) (wid * RPT + k) * WAVE + (u32) lane < n;
#else
          const bool here = (u32) lane < uniform32 (sh.slot_cnt[tb][wid * RPT + k]);
#endif
          const u32 q = (u32) (wid * RPT + k) * WAVE + (u32) lane - pv_base;
          if (here && q < pv_len) sh.s.skey[q] = key[k];
        }
      }
      __syncthreads (); /* pivot keys complete (and the counters are zero) */
      /* sub-buckets per gap between two pivot keys, by interpolation inside the gap (what lies between
       * two neighbours of the longest run is spread evenly far more often than the tile as a whole) */
      u32 sub_bits = 0;
      while (sub_bits < 3 && ((pv_len + 1u) << (sub_bits + 1)) <= (u32) NB) sub_bits++;
      const float sub_n = (float) (1u << sub_bits);
      u32 lb[RPT]; /* lower bounds in the pivot run: the searches of a thread's records in step */
#pragma unroll
      for (int k = 0; k < RPT; k++) lb[k] = 0;
      for (u32 h = 1u << (31 - __builtin_clz (pv_len | 1u)); h; h >>= 1) { /* uniform trip count */
        u64 pk[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) pk[k] = sh.s.skey[(lb[k] + h <= pv_len ? lb[k] + h : 1u) - 1u];
#pragma unroll
        for (int k = 0; k < RPT; k++) lb[k] = (lb[k] + h <= pv_len && pk[k] < key[k]) ? lb[k] + h : lb[k];
      }
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const u32 b = lb[k];
        /* key in (pk[b-1], pk[b]]: its place in the gap, monotone in the key (float conversions and
         * products by positive constants are), the same for equal keys */
        u32 sub = 0;
        if (b > 0 && b < pv_len) {
          const u64 lo_k = sh.s.skey[b - 1u], d = sh.s.skey[b] - lo_k, x = key[k] - lo_k;
          const u32 shf = d >> 24 ? 40u - (u32) __builtin_clzll (d) : 0u; /* d >> shf below 2^24: exact in a float */
          const float q = (float) (u32) (x >> shf) * sub_n / (float) ((u32) (d >> shf) + 1u);
          sub = (u32) q;
          sub = sub < (1u << sub_bits) ? sub : (1u << sub_bits) - 1u;
        }
        bk[k] = (b << sub_bits) + sub;
      }
      if (LEAD) { /* the pivot keys lay where the grouped keys go */
        __syncthreads ();
        fill_g ();
      }
    };
    {
      u32 bk[RPT];
      if (__builtin_expect (pivot_first, 0)) {
        pivot_buckets (bk);
      } else {
        /* (a tile whose shifted key range is below the number of buckets: v itself, less one -- the same multiply, no branch) */
        const u32 mul = bk_direct ? 0xffffffffu : bk_mul;
#pragma unroll
        for (int k = 0; k < RPT; k++) bk[k] = __umulhi ((u32) ((key[k] - key_lo) >> bk_sh), mul);
      }
      count_pass (bk); /* (the atomics first: their round trip overlaps what follows) */
    }
    PHASE_STAMP (21); /* (bucket numbers, counting atomics returned) */
    /* the next tile's records: asked for as soon as this tile's have left the registers, a whole
     * iteration before they are looked at (its table was written during the previous iteration).  One
     * tile per workgroup is all that is in flight. */
    {
      const u32 nxt = uniform32 (sh.hdr[tb1][0]);
      if (nxt < ntl && (u32) (wid * RPT) < uniform32 (sh.hdr[tb1][2])) {
        fetch (tb1);
      } else {
        /* nothing to fetch: the fetch registers may hold anything (said so, or the compiler keeps their old values alive
         * through this arm and copies all twelve on both arms: 24 moves per wavefront and tile) */
#pragma unroll
        for (int k = 0; k < RPT; k++) asm volatile ("" : "=v"(pre[k].x), "=v"(pre[k].y), "=v"(pre[k].z));
      }
    }
    PHASE_STAMP (22); /* (the next tile's fetch issued) */
    /* the ordered tile: counts 0, nothing live */
    {
      static_assert (CAPS % 4 == 0 && CAPS >= 4 * NT, "the ordered tile's counts are zeroed 16 bytes at a time: whole rounds and a part of one");
#pragma unroll
      for (int r = 0; r < CAPS / (4 * NT); r++) *reinterpret_cast<u32x4 *> (&sh.s.scnt[4 * (r * NT + tid)]) = u32x4 { 0, 0, 0, 0 };
      if (tid < (CAPS - CAPS / (4 * NT) * (4 * NT)) / 4) *reinterpret_cast<u32x4 *> (&sh.s.scnt[CAPS / (4 * NT) * (4 * NT) + 4 * tid]) = u32x4 { 0, 0, 0, 0 };
    }
    if (!LEAD) for (int i = tid; i < (CAPS + 3) / 4; i += NT) sh.live[i] = 0;
    if (LEAD) { /* (last read two tiles ago) */
      static_assert (!LEAD || Shared::LW % 4 == 0, "the bitmap is zeroed 16 bytes at a time");
      for (int i = tid; i < Shared::LW / 4; i += NT) *reinterpret_cast<u32x4 *> (&sh.lead[it & 1][4 * i]) = u32x4 { 0, 0, 0, 0 };
    }
    /* service: the chain words of the tile staged one iteration ago are asked for; they are looked at
     * behind B4 at the earliest (the memory counter retires in order: a look waits for every older
     * operation of this wavefront, the previous write-out's stores included) */
    if (service && MODE == NWAY_UNION && pend) {
      const u64 prow = pend_tile / WAVE;
      if ((u32) lane < pend_tile % WAVE) xagg = peek_u32 (&agg[prow * WAVE + lane]);
      xcarry = peek_u64 (&carry[prow]);
    }

    /* ---- buckets.  Attempt 0: by interpolation inside the tile's key range (no search at all).  If a
     * bucket holds more than NWAY_TRY0 keys -- clustered keys: stretches of adjacent keys with wide
     * gaps put a whole stretch into one bucket -- attempt 1 buckets by RANK IN THE TILE'S LONGEST RUN
     * (one binary search per record in that run's keys, copied to LDS): buckets then hold what the other
     * runs have between two neighbours of the pivot run, whatever the keys' values.  Only a tile that
     * defeats that too (more than NWAY_LIMIT keys in a bucket) takes the full search path below. */
    bucket_pass (p.force_fallback ? 0u : (pivot_first ? (u32) NWAY_LIMIT : (u32) NWAY_TRY0));
    if (__builtin_expect (!accepted, 0)) { /* (cold: laid out behind the loop; ONE test on the usual path) */
      if (p.force_fallback && mx == 0) {
        accepted = true; /* (an empty tile) */
      } else if (!pivot_first && p.force_fallback != 1) {
        u32 bk[RPT];
        pivot_buckets (bk);
        count_pass (bk);
        bucket_pass ((u32) NWAY_LIMIT);
      }
    }

    /* ---- position of every record = number of smaller keys in the tile */
    u32 pos[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) pos[k] = 0;
    if (__builtin_expect (accepted, 1)) {
      if (has_rec) {
        u32 lt[RPT], ga[RPT];
        const u32 g0 = lds_offset (&sh.g ()[0]);
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          lt[k] = 0;
          ga[k] = g0 + 8u * nway_skew (st[k]);
        }
        /* every lane runs the longest bucket's length (rounded up to even): behind its own bucket a lane
         * meets larger keys or all-ones */
#pragma unroll
        for (int q = 0; q + 4 <= RPT; q += 4)
          nway_rank_steps<0> (mx, ga[q], ga[q + 1], ga[q + 2], ga[q + 3], *reinterpret_cast<const u64 (*)[4]> (&key[q]), *reinterpret_cast<u32 (*)[4]> (&lt[q]));
        if constexpr (RPT % 4 != 0) { /* (a fifth position per thread: two steps at a time as well) */
          for (u32 j = 0; j < mx; j += 2) {
#pragma unroll
            for (int k = RPT / 4 * 4; k < RPT; k++) {
              const u64 r0 = lds_load<u64> (ga[k] + 8u * j), r1 = lds_load<u64> (ga[k] + 8u * j + 8u);
              lt[k] += (r0 < key[k] ? 1u : 0u) + (r1 < key[k] ? 1u : 0u);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < RPT; k++) pos[k] = st[k] + lt[k];
      }
    } else {
      /* clustered keys: the records back to LDS as the sorted runs they came as, and every record adds
       * up its lower bounds in all the runs */
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const u32 q = (u32) (wid * RPT + k) * WAVE + (u32) lane;
        if (ba[k] >> 31) {
          sh.raw[3 * q] = (u32) key[k];
          sh.raw[3 * q + 1] = (u32) (key[k] >> 32);
          sh.raw[3 * q + 2] = cnt[k];
        }
      }
      __syncthreads ();
      for (u32 q = 0; q < p.k; q++) {
        const u32 pb = uniform32 (sh.tab_pbase[tb][q]), len = uniform32 (sh.tab_len[tb][q]);
        const u32 steps = len ? 32u - (u32) __builtin_clz (len) : 0u;
        u32 lo[RPT], hi[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          lo[k] = 0;
          hi[k] = len;
        }
        for (u32 s = 0; s < steps; s++) {
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            const bool act = lo[k] < hi[k];
            const u32 mid = (lo[k] + hi[k]) >> 1;
            const u32 at = 3 * (pb + (act ? mid : 0u));
            const u64 km = (u64) sh.raw[at] | ((u64) sh.raw[at + 1] << 32);
            const bool c = km < key[k];
            lo[k] = (act && c) ? mid + 1u : lo[k];
            hi[k] = (act && !c) ? mid : hi[k];
          }
        }
#pragma unroll
        for (int k = 0; k < RPT; k++) pos[k] += lo[k];
      }
#pragma unroll
      for (int k = 0; k < RPT; k++) pos[k] = (ba[k] >> 31) ? pos[k] : 0u; /* (lanes without a record searched with whatever their registers held) */
      __syncthreads ();
      for (int i = 4 * tid; i < CAPS; i += 4 * NT) *reinterpret_cast<u32x4 *> (&sh.s.scnt[i]) = u32x4 { 0, 0, 0, 0 }; /* (the runs lay over the counts) */
      __syncthreads ();
    }
    PHASE_STAMP (8);

    /* ---- the key once per position, the counts folded by LDS atomics */
    u32 lead_bits = 0;     /* LEAD: record k is the first of its key to arrive at its position (and, behind B5, is kept) */
    u32 lead_before[RPT];  /* ... its bitmap word as the record found it */
    if (LEAD && has_rec) {
      /* straight-line: the folds, then the claims (a lane without a record adds 0 to a word of its own and claims nothing) */
      if (p.rule == 1u) {
#pragma unroll
        for (int k = 0; k < RPT; k++) atomicAdd (&sh.s.scnt[nway_pick (nway_valid_mask (ba[k]), nway_skew (pos[k]), (u32) lane)], cnt[k] & nway_valid_mask (ba[k]));
      } else if (p.rule == 4u) {
#pragma unroll
        for (int k = 0; k < RPT; k++) atomicMax (&sh.s.scnt[nway_pick (nway_valid_mask (ba[k]), nway_skew (pos[k]), (u32) lane)], cnt[k] & nway_valid_mask (ba[k]));
      }
#pragma unroll
      for (int k = 0; k < RPT; k++)
        lead_before[k] = atomicOr (&sh.lead[it & 1][nway_pick (nway_valid_mask (ba[k]), pos[k] / (u32) Shared::LBP, (u32) lane)], (1u << (pos[k] % (u32) Shared::LBP)) & nway_valid_mask (ba[k]));
    }
    if (!LEAD && has_rec) {
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        if (!(ba[k] >> 31)) continue;
        u32 q = nway_skew (pos[k]);
        if (MODE == NWAY_DUPS) q = nway_skew (pos[k] + atomicAdd (&sh.s.scnt[q], 1u)); /* equal sample keys: one position each */
        else if (MODE == NWAY_TABLE) { /* (counts go to the table, below) */ }
        else if (MODE == NWAY_PROBE) {
          /* a record of list 0 (the first run: its slots are the tile's first) leaves its index + 1 */
          const u32 idx = (u32) (wid * RPT + k) * WAVE + (u32) lane;
#if GT4_KM > 8
          if (idx < sh.tab_len[tb][0]) {
#else
          if (uniform32 (sh.slot_run[tb][wid * RPT + k]) == 0u) {
#endif
            sh.s.scnt[q] = idx + 1u;
            p.table_keys[out_base + idx] = key[k];
          }
          continue;
        }
        else if (p.rule == 1u) atomicAdd (&sh.s.scnt[q], cnt[k]);
        else if (p.rule == 4u) atomicMax (&sh.s.scnt[q], cnt[k]);
        sh.s.skey[q] = key[k];
        reinterpret_cast<unsigned char *> (sh.live)[q] = MODE == NWAY_DUPS ? (unsigned char) (1u + list_of (k)) : (unsigned char) 1;
      }
    }
    PHASE_STAMP (9);
    /* ---- service window (the other wavefronts are ranking): the table of the tile two iterations
     * ahead from the entries asked for one iteration ago, the ticket drawn then, new requests -- and
     * the previous tile leaves its staging area as soon as the chain has its offset: only this
     * wavefront ever waits for the chain, and not before everybody else stands at B6 */
    bool wo_done = !(nway_staged (MODE) && pend);
    auto write_out = [&] (u64 excl_bytes) {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc ((void *) (reinterpret_cast<char *> (out) + excl_bytes), 0, (int) (12 * pend_tot), 0x00020000);
      const u32 chunks = (3 * pend_tot + 3) >> 2;
      for (u32 c0 = 0; c0 < chunks; c0 += 4 * WAVE) {
        u32x4 w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const u32 c = c0 + (u32) u * WAVE + (u32) lane;
          w[u] = *reinterpret_cast<const u32x4 *> (sh.stage + 4 * (c < chunks ? c : chunks - 1u));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128 (w[u], r, 16 * (c0 + (u32) u * WAVE + (u32) lane), 0, GT4_STORE_AUX);
      }
      wo_done = true;
    };
    if (service) {
      PHASE_STAMP (16);
      build_table (sv_row, sv_t2, tb2);
      PHASE_STAMP (17);
      const u32 t3 = uniform32 (sv_tk);
      sv_t2 = t3;
      sv_row = load_row (t3);
      if (lane == 0) sv_tk = deal (it + 4);
      PHASE_STAMP (18);
      if (!wo_done) {
        if (MODE == NWAY_UNION) {
          const bool mine = (u32) lane < pend_tile % WAVE;
          if (__all (!mine || (xagg & AGG_READY) != 0) && (xcarry & CARRY_READY)) {
            write_out (12 * ((xcarry & ~CARRY_READY) + dpp_wave_sum_u32 (mine ? (xagg & ~AGG_READY) : 0u)));
          } else {
            /* not yet: ask again, look again behind B5 */
            const u64 prow = pend_tile / WAVE;
            if (mine && !(xagg & AGG_READY)) xagg = peek_u32 (&agg[prow * WAVE + lane]);
            if (!(xcarry & CARRY_READY)) xcarry = peek_u64 (&carry[prow]);
          }
        } else {
          write_out (12 * pend_base);
        }
      }
      PHASE_STAMP (19);
    }
    PHASE_STAMP (10);
    __syncthreads (); /* B5: the tile in key order */
    PHASE_STAMP (11);
    if (service && !wo_done) write_out (12 * resolve_offset (agg, carry, pend_tile, lane, xagg, xcarry, ctl, spin_limit));
    fill_g (); /* every walk of this tile is behind B5: the grouped keys of the next tile start from all-ones */
    PHASE_STAMP (12);

    u32 tile_total;
    if constexpr (LEAD) {
      /* ---- the leaders look at the folded counts: cutoff, sum of the kept counts; a leader that is not kept gives
       * its bit back */
      u32 lf[RPT];
#pragma unroll
      for (int k = 0; k < RPT; k++) lf[k] = 0;
      if (has_rec) {
        /* (every lane reads: position 0 where no record is) */
        if (p.rule == 7u) {
#pragma unroll
          for (int k = 0; k < RPT; k++) lf[k] = p.count_override;
        } else {
#pragma unroll
          for (int k = 0; k < RPT; k++) lf[k] = lds_load<u32> (lds_offset (&sh.s.scnt[0]) + 4u * nway_skew (pos[k]));
        }
        const u32 least = p.filter == FILTER_RAW ? 0u : p.cutoff; /* kept iff the folded count reaches it */
        u32 drop = 0;
#pragma unroll
        for (int k = 0; k < RPT; k++) { /* (masks, not conditions: see nway_pick) */
          const u32 leads = nway_valid_mask (ba[k]) & (((lead_before[k] >> (pos[k] % (u32) Shared::LBP)) & 1u) - 1u);
          const u32 enough = lf[k] >= least ? ~0u : 0u;
          acc_sum += lf[k] & leads & enough;
          lead_bits |= leads & enough & (1u << k);
          drop |= leads & ~enough & (1u << k);
        }
        if (drop) { /* (rare: a cutoff above the counts) */
#pragma unroll
          for (int k = 0; k < RPT; k++)
            if ((drop >> k) & 1u) atomicAnd (&sh.lead[it & 1][pos[k] / (u32) Shared::LBP], ~(1u << (pos[k] % (u32) Shared::LBP)));
        }
      }
      PHASE_STAMP (13);
      __syncthreads (); /* B6: the bitmap holds the kept leaders; the staging area is free */
      PHASE_STAMP (14);
      /* kept leaders in front of every bitmap word: every wavefront scans the bitmap itself (LWL words per lane) */
      constexpr int LWL = Shared::LWL;
      u32 w[LWL], c = 0;
#pragma unroll
      for (int j = 0; j < LWL; j++) w[j] = sh.lead[it & 1][lane * LWL + j]; /* (consecutive: 16-byte reads) */
#pragma unroll
      for (int j = 0; j < LWL; j++) c += (u32) __popc (w[j]);
      const u32 incl = dpp_inclusive_scan_u32 (c);
      tile_total = (u32) __builtin_amdgcn_readlane ((int) incl, WAVE - 1);
      blk_cnt += tile_total;
      if (MODE == NWAY_UNION && service) {
        if (lane == 0) publish_u32 (&agg[cur], AGG_READY | tile_total);
      }
      if (MODE == NWAY_COUNT && tid == 0 && p.tile_totals) p.tile_totals[cur] = tile_total;
      PHASE_STAMP (16); /* (diagnostics, wavefronts other than the service one: the bitmap scan) */
      if (nway_staged (MODE) && has_rec) {
        u32 before = incl - c;
        u32 pre16[LWL]; /* (every lane its words, whether it holds a leader or not) */
#pragma unroll
        for (int j = 0; j < LWL; j++) {
          pre16[j] = before;
          before += (u32) __popc (w[j]);
        }
        if constexpr (LWL == 8) { /* eight 16-bit prefixes: one 16-byte store */
          *reinterpret_cast<u32x4 *> (&sh.wpre[wid][lane * LWL]) = u32x4 { pre16[0] | (pre16[1] << 16), pre16[2] | (pre16[3] << 16), pre16[4] | (pre16[5] << 16), pre16[6] | (pre16[7] << 16) };
        } else {
#pragma unroll
          for (int j = 0; j < LWL; j++) sh.wpre[wid][lane * LWL + j] = (unsigned short) pre16[j];
        }
        /* (the table is this wavefront's own: LDS operations of one wavefront complete in order) */
        u32 pw[RPT], lw[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          pw[k] = sh.wpre[wid][pos[k] / (u32) Shared::LBP];
          lw[k] = sh.lead[it & 1][pos[k] / (u32) Shared::LBP];
        }
        PHASE_STAMP (17); /* (diagnostics: prefix table written, words asked for) */
#pragma unroll
        for (int k = 0; k < RPT; k++) { /* (what is not kept goes to the lane's trash record) */
          const u32 slot = nway_pick (0u - ((lead_bits >> k) & 1u), pw[k] + (u32) __popc (lw[k] & ((1u << (pos[k] % (u32) Shared::LBP)) - 1u)), (u32) CAP + 2u + (u32) lane);
          sh.stage[3 * slot] = (u32) key[k];
          sh.stage[3 * slot + 1] = (u32) (key[k] >> 32);
          sh.stage[3 * slot + 2] = lf[k];
        }
      }
    } else {
    /* ---- positions in order, one per lane (a wavefront walks its RPT chunks of 64): keep test, ballots */
    u64 okey[RPT];
    u32 ocnt[RPT];
    u32 keep_bits = 0, wave_kept = 0;
    u32 kpre[RPT]; /* kept in the wavefront's earlier chunks (uniform) */
#pragma unroll
    for (int i = 0; i < RPT; i++) {
      okey[i] = 0;
      ocnt[i] = 0;
      kpre[i] = 0;
    }
    if (MODE == NWAY_PROBE && p.table_cols <= (u32) Shared::ROW_COLS_MAX) {
      /* (behind B5: every record of list 0 has left its index + 1 where its key's records look) */
      u32 row[RPT], val[RPT];
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        row[k] = has_rec && (ba[k] >> 31) ? sh.s.scnt[nway_skew (pos[k])] - 1u : 0xffffffffu; /* (0: list 0 does not hold the key) */
        val[k] = p.rule == 7u ? p.count_override : cnt[k];
      }
      table_rows (row, val, uniform32 (sh.tab_len[tb][0]), out_base);
    } else if (MODE == NWAY_PROBE && has_rec) {
      /* (wide tables: the host has zeroed the table) */
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        if (!(ba[k] >> 31)) continue;
        const u32 r = sh.s.scnt[nway_skew (pos[k])];
        if (r) p.table_counts[(out_base + r - 1u) * p.table_cols + p.table_col[list_of (k)]] = p.rule == 7u ? p.count_override : cnt[k];
      }
    }
    if (MODE != NWAY_PROBE && has_pos) {
#pragma unroll
      for (int i = 0; i < RPT; i++) {
        const u32 q = nway_skew ((u32) (wid * RPT + i) * WAVE + (u32) lane);
        const u32 lv = lds_load<unsigned char> (lds_offset (&sh.live[0]) + q);
        const bool on = lv != 0;
        okey[i] = lds_load<u64> (lds_offset (&sh.s.skey[0]) + 8u * q);
        u32 f = lds_load<u32> (lds_offset (&sh.s.scnt[0]) + 4u * q);
        if (MODE == NWAY_DUPS) f = lv - 1u; /* a merged sample keeps the list it came from: the partition counts them */
        else if (MODE == NWAY_TABLE) f = 0;
        else if (p.rule == 7u) f = p.count_override;
        ocnt[i] = f;
        const bool keep = on & (MODE == NWAY_DUPS || MODE == NWAY_TABLE || p.filter == FILTER_RAW || f >= p.cutoff);
        keep_bits |= keep ? 1u << i : 0u;
        acc_sum += keep ? f : 0u;
        kpre[i] = wave_kept;
        wave_kept += (u32) __popcll (__builtin_amdgcn_ballot_w64 (keep));
      }
    }
    if (lane == 0) sh.wkept[wid] = wave_kept;
    PHASE_STAMP (13);
    __syncthreads (); /* B6: kept per wavefront; the staging area is free */
    PHASE_STAMP (14);
    {
      const u32 x = lane < NW ? sh.wkept[lane] : 0u;
      const u32 incl2 = dpp_inclusive_scan_u32 (x);
      tile_total = (u32) __builtin_amdgcn_readlane ((int) incl2, WAVE - 1);
      const u32 wbase = wid ? (u32) __builtin_amdgcn_readlane ((int) incl2, wid - 1) : 0u;
      blk_cnt += tile_total;
      if (MODE == NWAY_UNION && service) {
        if (lane == 0) publish_u32 (&agg[cur], AGG_READY | tile_total);
      }
      if ((MODE == NWAY_COUNT || MODE == NWAY_TABLE) && tid == 0 && p.tile_totals) p.tile_totals[cur] = tile_total;
      if (nway_staged (MODE) && wave_kept) {
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          const bool keep = (keep_bits >> i) & 1u;
          const u64 m = __builtin_amdgcn_ballot_w64 (keep);
          const u32 slot = wbase + kpre[i] + __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u));
          if (keep) {
            sh.stage[3 * slot] = (u32) okey[i];
            sh.stage[3 * slot + 1] = (u32) (okey[i] >> 32);
            sh.stage[3 * slot + 2] = ocnt[i];
          }
        }
      }
      if (MODE == NWAY_TABLE) {
        const bool via_lds = p.table_cols <= (u32) Shared::ROW_COLS_MAX; /* uniform */
        /* Wide tables only: the tile's rows of the count matrix start as zeros, written here, 16 bytes per lane (round 4:
         * instead of a memset of the whole matrix in front of the launch); the records' own stores follow behind a wait
         * for these and the barrier below, so they land on the zeros. */
        if (!via_lds) {
          const u64 words = (u64) tile_total * p.table_cols;
          const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc ((void *) (p.table_counts + out_base * p.table_cols), 0, (int) (4 * words), 0x00020000);
          for (u32 c = (u32) tid; 4ull * c < words; c += NT) __builtin_amdgcn_raw_buffer_store_b128 (u32x4 { 0, 0, 0, 0 }, zr, 16 * c, 0, 0);
          asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        }
        /* the key column, and every position's row (over the ordered tile's counts, which this mode
         * does not fold) for the records to find */
        if (wave_kept) {
#pragma unroll
          for (int i = 0; i < RPT; i++) {
            const bool keep = (keep_bits >> i) & 1u;
            const u64 m = __builtin_amdgcn_ballot_w64 (keep);
            const u32 slot = wbase + kpre[i] + __builtin_amdgcn_mbcnt_hi ((u32) (m >> 32), __builtin_amdgcn_mbcnt_lo ((u32) m, 0u));
            if (keep) {
              p.table_keys[out_base + slot] = okey[i];
              sh.s.scnt[nway_skew ((u32) (wid * RPT + i) * WAVE + (u32) lane)] = slot;
            }
          }
        }
        __syncthreads ();
        if (via_lds) {
          u32 row[RPT];
#pragma unroll
          for (int k = 0; k < RPT; k++) row[k] = has_rec && (ba[k] >> 31) ? sh.s.scnt[nway_skew (pos[k])] : 0xffffffffu;
          table_rows (row, cnt, tile_total, out_base); /* (its barrier stands behind every thread's look at its rows) */
        } else {
          if (has_rec) {
#pragma unroll
            for (int k = 0; k < RPT; k++) {
              if (!(ba[k] >> 31)) continue;
              const u32 col = p.table_col[list_of (k)];
              const u64 row = out_base + sh.s.scnt[nway_skew (pos[k])];
              p.table_counts[row * p.table_cols + col] = cnt[k];
            }
          }
          __syncthreads (); /* (the rows lie where the next tile's counts are zeroed) */
        }
      }
    }
    } /* (!LEAD) */
    pend = nway_staged (MODE);
    pend_tot = tile_total;
    pend_tile = cur;
    pend_base = out_base;
    PHASE_STAMP (15);
    it++;
    {
      const int t0 = tb;
      tb = tb1;
      tb1 = tb2;
      tb2 = t0;
    }
  }
PROF (
  if (tid == GT4_STAMP_TID)
    for (int i = 0; i < 24; i++) atomicAdd (&ctl->phase_cycles[i], ph[i]);
)
  /* drain: the last tile is still staged */
  if (nway_staged (MODE) && pend) {
    __syncthreads ();
    if (MODE == NWAY_UNION && wid == 0) {
      const u64 x = resolve_offset (agg, carry, pend_tile, lane, 0, 0, ctl, spin_limit);
      if (lane == 0) sh.excl = x;
    }
    __syncthreads ();
    write_out_tile<NT> (out, MODE == NWAY_UNION ? uniform64 (sh.excl) : pend_base, pend_tot, sh.stage, tid);
  }
  if (MODE != NWAY_DUPS && MODE != NWAY_TABLE && MODE != NWAY_PROBE) {
    const u64 v = wave_sum (acc_sum);
    if (lane == 0 && v) atomicAdd (&ctl->total_count[0], v);
    if (tid == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt);
  }
  if (MODE == NWAY_TABLE && tid == 0 && blk_cnt) atomicAdd (&ctl->n_words[0], blk_cnt); /* the table's rows */
}
