/* gt4hip_nway_host.h -- N-way: launches and the host side of one call (sample levels, partition, retries, the tile
 * kernel, read-back).  Included by gt4hip_nway_body.h inside namespace gt4::<anon>::km8 / km32; no include guard. */
constexpr int NWAY_NT = GT4_NWAY_NT;
constexpr int NWAY_NBF = GT4_NWAY_NBF;
/* positions per thread: the modes that keep no ordered copy of the tile (GT4_NWAY_LEAD) have LDS for one more */
#define GT4_NWAY_RPT_LEAD GT4_NWAY_RPT
constexpr int nway_rpt (int mode) { return nway_lead (mode) ? GT4_NWAY_RPT_LEAD : GT4_NWAY_RPT; }
constexpr int nway_cap (int mode) { return NWAY_NT * nway_rpt (mode); }
constexpr int NWAY_CAP_MIN = NWAY_NT * (GT4_NWAY_RPT_LEAD < GT4_NWAY_RPT ? GT4_NWAY_RPT_LEAD : GT4_NWAY_RPT);

template <int MODE>
hipError_t launch_nway (hipStream_t s, int grid, const NwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  hipLaunchKernelGGL ((k_nway_merge<NWAY_NT, nway_rpt (MODE), NWAY_NBF, MODE>), dim3 (grid), dim3 (NWAY_NT), 0, s, p, part, out, desc, ctl);
  return hipGetLastError ();
}

hipError_t launch_nway_mode (hipStream_t s, int mode, int grid, const NwayParams &p, const u64 *part, u32 *out, u64 *desc, PairControl *ctl)
{
  if (mode == NWAY_DUPS) return launch_nway<NWAY_DUPS> (s, grid, p, part, out, desc, ctl);
  if (mode == NWAY_COUNT) return launch_nway<NWAY_COUNT> (s, grid, p, part, out, desc, ctl);
  if (mode == NWAY_TABLE) return launch_nway<NWAY_TABLE> (s, grid, p, part, out, desc, ctl);
  if (mode == NWAY_PROBE) return launch_nway<NWAY_PROBE> (s, grid, p, part, out, desc, ctl);
  return launch_nway<NWAY_UNION> (s, grid, p, part, out, desc, ctl);
}

int nway_blocks_per_cu (int mode)
{
  static int cache[5] = { 0, 0, 0, 0, 0 };
  if (!cache[mode]) {
    int n = 0;
    hipError_t e;
    if (mode == NWAY_DUPS) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_DUPS), NWAY_NBF, NWAY_DUPS>, NWAY_NT, 0);
    else if (mode == NWAY_COUNT) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_COUNT), NWAY_NBF, NWAY_COUNT>, NWAY_NT, 0);
    else if (mode == NWAY_TABLE) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_TABLE), NWAY_NBF, NWAY_TABLE>, NWAY_NT, 0);
    else if (mode == NWAY_PROBE) e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_PROBE), NWAY_NBF, NWAY_PROBE>, NWAY_NT, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&n, k_nway_merge<NWAY_NT, nway_rpt (NWAY_UNION), NWAY_NBF, NWAY_UNION>, NWAY_NT, 0);
    if (e != hipSuccess || n < 1) n = 1;
    const int by_regs = nway_waves_per_simd (NWAY_NT) * 4 / (NWAY_NT / 64);
    if (by_regs >= 1 && n > by_regs) n = by_regs;
    cache[mode] = n;
  }
  return cache[mode];
}


/* ------------------------------------------------------------------ host orchestration */

namespace host_part {

struct Level {
  NwayParams p;            /* lists of this level (level 0: the caller's; above: sample lists) */
  gt4hip_list *owned[NWAY_MAX];
  u64 total;
};

size_t nway_desc_bytes (u64 tiles)
{
  const u64 rows = (tiles + 63) / 64;
  return (((size_t) rows * 64 * 16 + (size_t) (rows + 1) * 32 + (size_t) rows * 32) + 255) & ~(size_t) 255; /* agg, carry, rowsum */
}

int nway_grow (gt4hip_context *ctx, void **p, size_t *have, size_t need)
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

/* samples per tile: a tile between two boundary keys G samples apart holds at most G + k - 1 samples
 * (ties at the boundaries), each list at most (its samples + 1) * S - 1 records, every run rounded up
 * to whole wavefronts.  `sure`: the G for which no tile can overflow; the first try takes the expected
 * tile (G * S records) plus GT4_NWAY_MARGIN (five) standard deviations of the lists' offsets against their sample grids; tiles beyond the
 * capacity are cut in two (k_nway_emit). */
void nway_samples_per_tile (u32 k, int positions, u32 *first_try, u32 *sure)
{
  const double cap = (double) positions - 0.5 * NWAY_HS * k; /* half a slot of padding per run, on average */
  const double margin = GT4_NWAY_MARGIN * NWAY_SAMPLE * sqrt ((double) k / 6.0);
  long g1 = (long) ((cap - margin) / NWAY_SAMPLE);
  long g0 = ((long) positions - (long) NWAY_HS * k) / NWAY_SAMPLE - (2L * k - 1);
  if (g0 < 1) g0 = 1;
  if (g1 < g0) g1 = g0;
  *first_try = (u32) g1;
  *sure = (u32) g0;
}

}  // namespace host_part
using namespace host_part;

int nway_run (gt4hip_context *ctx, const gt4hip_list *const lists[], uint32_t k, uint32_t rule, uint32_t cutoff, uint32_t ovr,
                     uint32_t filter, bool count_only, gt4hip_list *out, uint64_t *n_words, uint64_t *total_count, double *device_ms,
                     int *used, gt4hip_count_table *table, const uint32_t *cols, bool probe)
{
  *used = 0;
  if (k < 2 || k > NWAY_MAX) return GT4HIP_OK;
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
      for (int i = 0; i < NWAY_MAX; i++)
        if (lv.owned[i]) gt4hip_list_free (lv.owned[i]);
  };
  {
    const hipError_t e0 = hipEventRecord (ctx->ev[0], st);
    if (e0 != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "hipEventRecord failed: %s", hipGetErrorString (e0)); /* (nothing is owned yet) */
  }
  /* option "kway" = 1 (the default) lets the call decline clustered keys: *used = 0, the caller takes the tree */
  /* Host read-backs (round 5: nine per call of three levels -> four; each is a drained stream plus 20 - 30 us, 0.3 ms of
   * a 4.7 ms call on an eighth of the bench's lists, i.e. of one GPU's shard at 8 GPUs).  A top level of one tile reads
   * nothing back; the sample levels' control blocks are not read back (their error word stays set through the later
   * launches and is seen with the last one). */
  const bool may_decline = ctx->kway_enabled == 1 && !table && ctx->kway_vt == 0;
  u32 probe_windows = 0;
  bool probe_pending = false;
  hipMemsetAsync (ctx->ctl, 0, sizeof (PairControl), st);
  if (may_decline) {
    uint32_t longest = 0;
    for (uint32_t i = 1; i < k; i++)
      if (lists[i]->n_words > lists[longest]->n_words) longest = i;
    const u64 nl = lists[longest]->n_words;
    if (nl >= 16ull * NWAY_PROBE_KEYS) {
      const u32 windows = (u32) (nl / (4 * NWAY_PROBE_KEYS) < NWAY_PROBE_WINDOWS ? nl / (4 * NWAY_PROBE_KEYS) : NWAY_PROBE_WINDOWS);
      const hipError_t e = hipMemsetAsync ((char *) ctx->scratch + 32, 0, 8, st);
      if (e != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "N-way key probe failed: %s", hipGetErrorString (e));
      hipLaunchKernelGGL (k_nway_probe, dim3 (windows), dim3 (256), 0, st, (const u32 *) lists[longest]->dev, nl, windows, (u32) nway_buckets (NWAY_NBF * nway_cap (NWAY_UNION)), (u32 *) ctx->scratch + 8);
      probe_windows = windows;
      probe_pending = true;
      /* (read at once after all: read with the first partition read-back, a call that declines had sampled, merged
       * samples and partitioned for nothing -- 1.3 ms of a 53 ms tree on the clustered bench lists) */
      hipError_t e2 = hipMemcpyAsync (ctx->scratch_host + 4, (char *) ctx->scratch + 32, 8, hipMemcpyDeviceToHost, st);
      if (e2 == hipSuccess) e2 = hipStreamSynchronize (st);
      if (e2 != hipSuccess) return gt4hip_fail (ctx, GT4HIP_EHIP, "N-way key probe failed: %s", hipGetErrorString (e2));
      probe_pending = false;
      if (5ull * (u32) ctx->scratch_host[4] > probe_windows) {
        ctx->kway_declined++;
        return GT4HIP_OK; /* *used = 0 */
      }
    }
  }
  /* sample levels until one fits a single tile */
  const u64 one_tile = (u64) NWAY_CAP_MIN - (u64) NWAY_HS * k;
  while (levels.back ().total > one_tile) {
    const Level &lo = levels.back ();
    Level up;
    memset (&up, 0, sizeof up);
    up.p.k = k;
    for (uint32_t i = 0; i < k && !rc; i++) {
      const u64 m = lo.p.n[i] / NWAY_SAMPLE;
      rc = gt4hip_list_new (ctx, m ? m : 1, lists[0]->word_length, &up.owned[i]);
      if (rc) break;
      up.p.list[i] = (const u32 *) up.owned[i]->dev;
      up.p.n[i] = m;
      up.total += m;
    }
    if (!rc && up.total) {
      u64 g = (up.total + 255) / 256;
      if (g > 16384) g = 16384;
      hipLaunchKernelGGL (k_nway_sample, dim3 ((unsigned) g), dim3 (256), 0, st, lo.p, up.p);
    }
    levels.push_back (up);
    if (rc) {
      cleanup ();
      return rc;
    }
  }
  /* top-down: the merged samples of level l+1 cut level l into tiles */
  gt4hip_list *merged = NULL; /* merged sample records of the level above */
  for (int l = (int) levels.size () - 1; l >= 0 && !rc; l--) {
    Level &lv = levels[l];
    const u64 m_total = merged ? merged->n_words : 0;
    const int mode = l > 0 ? NWAY_DUPS : (table ? (probe ? NWAY_PROBE : NWAY_TABLE) : (count_only ? NWAY_COUNT : NWAY_UNION));
    const int cap = nway_cap (mode); /* positions of a tile */
    const u32 n_buckets = (u32) nway_buckets (NWAY_NBF * cap);
    u32 g_try, g_sure;
    nway_samples_per_tile (k, cap, &g_try, &g_sure);
    if (ctx->kway_g > 0) g_try = (u32) ctx->kway_g;
    u32 G = g_try;
    u64 tiles = 1;
    const u64 *part_final = NULL; /* the table the tile kernel reads: the partition's, or the one with the split tiles */
    for (;;) {
      tiles = m_total ? m_total / G + 2 : 1;
      if (tiles >= 0xfffffff0ull) {
        rc = gt4hip_fail (ctx, GT4HIP_EINVAL, "lists too long: %llu tiles", (unsigned long long) tiles);
        break;
      }
      lv.p.num_tiles = (u32) tiles;
      if ((rc = nway_grow (ctx, (void **) &ctx->kway_part, &ctx->kway_part_bytes, (size_t) (tiles + 1) * NWAY_PSTRIDE * 8))) break;
      const u64 threads = (tiles + 1) * NWAY_PSTRIDE;
      if (merged && ctx->kway_vt != 97 && G <= NWAY_G_MAX) {
        /* from the merged samples' list numbers (option "kway_vt" = 97 keeps the searches over whole brackets: tests) */
        const u64 n_br = (tiles + 1 + NWAY_BRACKET - 1) / NWAY_BRACKET;
        if ((rc = nway_grow (ctx, (void **) &ctx->kway_cnt, &ctx->kway_cnt_bytes, (size_t) n_br * NWAY_MAX * 4))) break;
        hipLaunchKernelGGL (k_nway_sample_counts, dim3 ((unsigned) ((n_br + 3) / 4)), dim3 (256), 0, st, (const u32 *) merged->dev, m_total, G, n_br, (u32 *) ctx->kway_cnt);
        hipLaunchKernelGGL (k_nway_bracket_bases, dim3 (1), dim3 (NWAY_MAX > 8 ? 1024 : 64 * NWAY_MAX), 0, st, (u32 *) ctx->kway_cnt, n_br);
        hipLaunchKernelGGL (k_nway_partition_rows, dim3 ((unsigned) n_br), dim3 (64), 0, st, lv.p, (const u32 *) merged->dev, m_total, G, n_buckets,
                            (const u32 *) ctx->kway_cnt, (u64 *) ctx->kway_part);
      } else {
        for (int pass = 0; pass < 2; pass++)
          hipLaunchKernelGGL (k_nway_partition, dim3 ((unsigned) ((threads + 255) / 256)), dim3 (256), 0, st, lv.p, merged ? (const u32 *) merged->dev : NULL,
                              m_total, G, n_buckets, (u64 *) ctx->kway_part, pass);
      }
      /* tiles that do not fit are cut in two (k_nway_need / _scan / _emit): flags = { more than two pieces,
       * clustered tiles, tiles cut, tiles of the final table } */
      const u64 n_blocks = (tiles + 1 + NWAY_SPLIT_BLOCK - 1) / NWAY_SPLIT_BLOCK;
      if ((rc = nway_grow (ctx, (void **) &ctx->kway_need, &ctx->kway_need_bytes, (size_t) (tiles + 1 + n_blocks + 4) * 4))) break;
      u32 *const need = (u32 *) ctx->kway_need, *const block_sums = need + tiles + 1;
      if (tiles == 1 && !merged && levels[l].total <= one_tile) {
        /* the top level: one tile that fits by construction -- nothing to cut, nothing to read back */
        part_final = (const u64 *) ctx->kway_part;
        if (l == 0) ctx->kway_splits = 0;
        break;
      }
      hipMemsetAsync (ctx->scratch, 0, 32, st);
      hipLaunchKernelGGL (k_nway_need, dim3 ((unsigned) n_blocks), dim3 (NWAY_SPLIT_BLOCK), 0, st, (const u64 *) ctx->kway_part, (u32) tiles, (u32) (cap / NWAY_HS), need, block_sums,
                          (u32 *) ctx->scratch);
      hipLaunchKernelGGL (k_nway_need_scan, dim3 (1), dim3 (1024), 0, st, block_sums, (u32) n_blocks, (u32 *) ctx->scratch + 3);
      hipError_t e = hipMemcpyAsync (ctx->scratch_host, ctx->scratch, 40, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) {
        rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way partition failed: %s", hipGetErrorString (e));
        break;
      }
      const u32 *const fl = (const u32 *) ctx->scratch_host;
      if (probe_pending) { /* (the probe ran in front of everything else on this stream) */
        probe_pending = false;
        if (5ull * fl[8] > probe_windows) {
          ctx->kway_declined++;
          if (merged) gt4hip_list_free (merged);
          cleanup ();
          return GT4HIP_OK; /* *used = 0 */
        }
      }
      bool overflow = fl[0] != 0;
      part_final = (const u64 *) ctx->kway_part;
      if (!overflow && fl[2]) {
        const u64 tiles2 = fl[3];
        if ((rc = nway_grow (ctx, (void **) &ctx->kway_part2, &ctx->kway_part2_bytes, (size_t) (tiles2 + 1) * NWAY_PSTRIDE * 8))) break;
        hipLaunchKernelGGL (k_nway_emit, dim3 ((unsigned) n_blocks), dim3 (NWAY_SPLIT_BLOCK), 0, st, lv.p, (const u64 *) ctx->kway_part, (u32) tiles, need, block_sums,
                            n_buckets, (u32) (cap / NWAY_HS), (u64 *) ctx->kway_part2, (u32 *) ctx->scratch);
        e = hipMemcpyAsync (ctx->scratch_host + 4, ctx->scratch, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize (st);
        if (e != hipSuccess) {
          rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way partition failed: %s", hipGetErrorString (e));
          break;
        }
        overflow = (u32) ctx->scratch_host[4] != 0;
        if (!overflow) {
          if (l == 0) ctx->kway_splits = fl[2];
          part_final = (const u64 *) ctx->kway_part2;
          tiles = tiles2;
          lv.p.num_tiles = (u32) tiles;
        }
      } else if (!overflow && l == 0) {
        ctx->kway_splits = 0;
      }
      if (!overflow) {
        if (l == 0 && may_decline && tiles >= 64 && 5ull * fl[1] > tiles) {
          /* the probe of the longest list did not see it, the tiles' own samples do: clustered keys */
          ctx->kway_declined++;
          if (merged) gt4hip_list_free (merged);
          cleanup ();
          return GT4HIP_OK; /* *used = 0 */
        }
        break;
      }
      /* a tile would overflow LDS: fewer samples per tile, down to the number that cannot overflow */
      ctx->kway_overflows++;
      if (merged) {
        /* The cuts come from the merged samples of the level above, whose launch is not read back on its own (see
         * below).  If a bounded wait gave up there (a shared device) the samples are incomplete and the partition built
         * on them is garbage: that is not a capacity problem -- the tree redoes the call, as a give-up in the last
         * launch does. */
        e = hipMemcpyAsync (ctx->ctl_host, ctx->ctl, sizeof (PairControl), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize (st);
        if (e != hipSuccess) {
          rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way partition failed: %s", hipGetErrorString (e));
          break;
        }
        if (ctx->ctl_host->error) {
          const unsigned flags = ctx->ctl_host->error;
          gt4hip_list_free (merged);
          cleanup ();
          if (flags & 2u) return gt4hip_fail (ctx, GT4HIP_EINTERNAL, "N-way merge kernel reported error flags 0x%x", flags);
          ctx->single_pass_fallbacks++;
          return GT4HIP_OK; /* *used = 0 */
        }
      }
      if (G <= g_sure) {
        rc = gt4hip_fail (ctx, GT4HIP_EINTERNAL, "N-way partition: a tile exceeds the capacity at %u samples per tile", G);
        break;
      }
      const u32 g2 = G - (G + 7) / 8;
      G = g2 > g_sure ? g2 : g_sure;
    }
    if (rc) break;
    if (merged) {
      gt4hip_list_free (merged);
      merged = NULL;
    }
    lv.p.rule = rule;
    lv.p.cutoff = cutoff;
    lv.p.count_override = ovr;
    lv.p.filter = filter;
    lv.p.spin_limit = ctx->spin_limit;
    lv.p.force_fallback = ctx->kway_vt == 99 ? 1u : (ctx->kway_vt == 98 ? 2u : 0u); /* option "kway_vt" = 99 / 98: every tile takes the search path / the pivot-run buckets (tests) */
    lv.p.scan_group = ctx->scan_group > 0 ? 1u : (ctx->scan_group < 0 ? 0u : (tiles > (48000ull << 6) ? 1u : 0u));
    lv.p.dynamic = ctx->dynamic > 0 ? 1u : (ctx->dynamic < 0 ? 0u : (mode == NWAY_UNION ? 1u : 0u));
    u32 *dst = NULL;
    if (l > 0) {
      if ((rc = gt4hip_list_new (ctx, lv.total ? lv.total : 1, lists[0]->word_length, &merged))) break;
      merged->n_words = lv.total;
      dst = (u32 *) merged->dev;
    } else if (!count_only) {
      dst = (u32 *) out->dev;
    }
    int grid = ctx->n_cus * nway_blocks_per_cu (mode);
    if (ctx->grid_override > 0) grid = (int) ctx->grid_override;
    if (mode == NWAY_UNION) {
      if ((rc = nway_grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, nway_desc_bytes (tiles)))) break;
      hipMemsetAsync (ctx->desc, 0, nway_desc_bytes (tiles), st);
      if ((u64) grid > tiles + 1) grid = (int) tiles + 1;
    } else if ((u64) grid > tiles) {
      grid = (int) tiles;
    }
    if (l == 0 && table && probe) { /* rows = the records of list 0: the table exists before the launch */
      if ((rc = gt4hip_table_alloc (ctx, table, lists[0]->n_words, table->n_lists))) break;
      table->n_keys = lists[0]->n_words;
      /* (up to ROW_COLS_MAX columns every row leaves the kernel whole, zeros included: no memset -- round 5) */
      if (table->n_lists > (uint32_t) NwayShared<NWAY_NT, nway_rpt (NWAY_PROBE), NWAY_NBF, NWAY_PROBE>::ROW_COLS_MAX)
        hipMemsetAsync (table->device_counts, 0, (size_t) table->n_keys * table->n_lists * 4, st);
      lv.p.table_keys = (u64 *) table->device_keys;
      lv.p.table_counts = (u32 *) table->device_counts;
      lv.p.table_cols = table->n_lists;
      for (uint32_t i = 0; i < k; i++) lv.p.table_col[i] = cols[i];
    } else if (l == 0 && table) {
      /* ONE launch (round 4): every tile writes its rows where its records start -- a tile has at most as many distinct
       * keys as records, so the table is allocated for the records and stays RAGGED (unused rows behind every tile's;
       * gt4hip_table_download and gt4hip_table_compact know, see gt4hip_count_table).  Round 3 counted every tile's
       * distinct keys in a launch of their own first: the records were read twice. */
      if ((rc = nway_grow (ctx, (void **) &ctx->desc, &ctx->desc_bytes, (size_t) tiles * 4 + 32 + (size_t) ((tiles + 1 + 1023) / 1024) * 8))) break; /* the tiles' totals, then their sums per block of 1024 */
      if ((rc = gt4hip_table_alloc (ctx, table, lv.total, table->n_lists))) break;
      lv.p.tile_totals = (u32 *) ctx->desc;
      lv.p.table_keys = (u64 *) table->device_keys;
      lv.p.table_counts = (u32 *) table->device_counts;
      lv.p.table_cols = table->n_lists;
      for (uint32_t i = 0; i < k; i++) lv.p.table_col[i] = cols[i];
    }
    hipMemsetAsync (ctx->ctl, 0, offsetof (PairControl, error), st); /* (totals, ticket; the error word stays) */
    hipMemsetAsync (&ctx->ctl->role, 0, sizeof (PairControl) - offsetof (PairControl, role), st);
    if (l == 0) hipEventRecord (ctx->ev[1], st);
    hipError_t e = launch_nway_mode (st, mode, grid, lv.p, part_final, dst, (u64 *) ctx->desc, ctx->ctl);
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way merge launch failed: %s", hipGetErrorString (e));
      break;
    }
    if (l == 0) hipEventRecord (ctx->ev[2], st);
    /* every level reads its control block back: a refused tile or a wait that gave up must not go unseen */
    e = l == 0 ? hipEventRecord (ctx->ev[3], st) : hipSuccess;
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "hipEventRecord failed: %s", hipGetErrorString (e));
      break;
    }
    if (l > 0) continue; /* (a sample level: its error word, if any, is still there behind the last launch) */
    e = hipMemcpyAsync (ctx->ctl_host, ctx->ctl, sizeof (PairControl), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize (st);
    if (e != hipSuccess) {
      rc = gt4hip_fail (ctx, GT4HIP_EHIP, "N-way merge failed: %s", hipGetErrorString (e));
      break;
    }
    if (ctx->ctl_host->error) {
      const unsigned flags = ctx->ctl_host->error;
      if (merged) gt4hip_list_free (merged);
      cleanup ();
      if (flags & 2u) return gt4hip_fail (ctx, GT4HIP_EINTERNAL, "N-way merge kernel reported error flags 0x%x", flags);
      /* a bounded wait gave up (shared device): the tree redoes the call */
      ctx->single_pass_fallbacks++;
      return GT4HIP_OK;
    }
    if (l == 0) {
PROF (
      {
        static const char *names[24] = { "p0: zeroing", "B1", "scan1", "B2", "scan2", "B3", "group", "B4", "rank", "fold", "service", "B5", "writeout+fill", "order", "B6", "stage+publish", "sv:-", "sv:table", "sv:ticket+row", "sv:try writeout | header", "p0: wait for records", "p0: buckets+atomics", "p0: fetch issue", "back edge" };
        unsigned long long tot = 0;
        for (int i = 0; i < 24; i++) tot += ctx->ctl_host->phase_cycles[i];
        fprintf (stderr, "[nway phases] tiles %llu:", (unsigned long long) tiles);
        for (int i = 0; i < 24; i++) fprintf (stderr, " %s %.1f%%", names[i], tot ? 100.0 * ctx->ctl_host->phase_cycles[i] / tot : 0.0);
        fprintf (stderr, " | avg cycles/tile %.0f\n", tiles ? (double) tot / tiles : 0.0);
      }
)
      *n_words = ctx->ctl_host->n_words[0];
      *total_count = ctx->ctl_host->total_count[0];
      if (table && !probe) {
        /* the ragged table's index: rows before every tile (compact) and where the tile's rows lie (padded) */
        table->n_keys = *n_words;
        if ((rc = gt4hip_table_set_ragged (ctx, table, tiles))) break;
        {
          const u64 nb = (tiles + 1 + 1023) / 1024;
          u64 *const bsum = (u64 *) ((char *) ctx->desc + (((size_t) tiles * 4 + 15) & ~(size_t) 15)); /* (behind the tiles' totals) */
          hipLaunchKernelGGL (k_nway_base_sums, dim3 ((unsigned) nb), dim3 (1024), 0, st, (const u32 *) ctx->desc, tiles, bsum);
          hipLaunchKernelGGL (k_nway_base_scan, dim3 (1), dim3 (1024), 0, st, bsum, nb);
          hipLaunchKernelGGL (k_nway_tile_bases, dim3 ((unsigned) nb), dim3 (1024), 0, st, (const u32 *) ctx->desc, tiles, (const u64 *) bsum, (u64 *) gt4hip_table_compact_bases (table));
        }
        hipLaunchKernelGGL (k_nway_padded_bases, dim3 ((unsigned) ((tiles + 256) / 256)), dim3 (256), 0, st, part_final, tiles, k, (u64 *) gt4hip_table_padded_bases (table));
        e = hipStreamSynchronize (st);
        if (e != hipSuccess) {
          rc = gt4hip_fail (ctx, GT4HIP_EHIP, "count table index failed: %s", hipGetErrorString (e));
          break;
        }
      }
      float ms = 0;
      if (hipEventElapsedTime (&ms, ctx->ev[0], ctx->ev[3]) == hipSuccess) *device_ms = ms;
      if (hipEventElapsedTime (&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) ctx->nway_kernel_ms = ms;
      ctx->nway_tiles = tiles;
      *used = 1;
    }
  }
  if (merged) gt4hip_list_free (merged);
  cleanup ();
  return rc;
}
