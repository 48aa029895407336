/*
 * gt4_glistcompare_cli.c -- `glistcompare`, the drop-in command line of the GPU set-operation
 * path.  Host C; every merge runs in the HIP kernels behind include/gt4hip.h.
 *
 * Same argv grammar, defaults, validation order, messages, output names (`<out>_<k>_union.list`
 * ...), tmp + rename discipline, stdout of --count_only / --print_operation / -v and exit codes
 * as the reference's main() (reference src/glistcompare.c:84-429, naming :814-834, :907-953).
 * Deliberate differences, all loud:
 *   - a file that cannot be opened is an error message + exit 1 (the reference dereferences NULL);
 *   - -mm and --subset are outside the GPU path: error + exit 1 (GT4I index inputs are read as
 *     the sorted k-mer lists they contain, as in the reference);
 *   - --stream and --disable_scouts are accepted and ignored (the whole list is uploaded to HBM;
 *     results are identical for well-formed files);
 *   - GT4HIP_VERBOSE=1 prints the device, kernel times and the chunk plan on stderr (-D prints exactly
 *     what the reference prints);
 *   - GT4HIP_CHECK_SORTED=1 rejects an input that is not strictly ascending (the reference trusts it);
 *   - --gpus N / GT4HIP_GPUS=N shards the job by key range over N GPUs (one worker process each, forked
 *     before any HIP call; every worker writes its own extents of the output files, or with
 *     GT4HIP_GATHER=rccl the shards are gathered on worker 0 over RCCL); GT4HIP_HBM_LIMIT=<bytes>, or
 *     inputs that do not fit the device memory, stream the job through the GPU in key-range chunks
 *     (gt4_shard.c).  Outputs are byte-identical to the single-pass run;
 *   - without a usable GPU the program fails: there is no CPU fallback.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "gt4_listfile.h"
#include "gt4_shard.h"
#include "gt4hip.h"

#define MAX_FILES 1024

enum { OPT_PLAIN, OPT_VERSION, OPT_HELP, OPT_OUT, OPT_CUTOFF, OPT_MM, OPT_UNION, OPT_INTRSEC, OPT_DIFF, OPT_DDIFF, OPT_DU,
       OPT_COUNT_ONLY, OPT_RULE, OPT_SUBSET, OPT_SEED, OPT_PRINT_OP, OPT_NOSCOUTS, OPT_STREAM, OPT_DEBUG, OPT_GPUS };

static const struct {
  const char *name;
  int opt;
} OPTIONS[] = {
  { "-v", OPT_VERSION }, { "--version", OPT_VERSION }, { "-h", OPT_HELP }, { "--help", OPT_HELP }, { "-?", OPT_HELP },
  { "-o", OPT_OUT }, { "--outputname", OPT_OUT }, { "-c", OPT_CUTOFF }, { "--cutoff", OPT_CUTOFF },
  { "--count_cutoff", OPT_CUTOFF }, /* alias used by the benchmark description; not in the reference */
  { "--gpus", OPT_GPUS },           /* not in the reference: key-range shards over several GPUs */
  { "-mm", OPT_MM }, { "--mismatch", OPT_MM }, { "-u", OPT_UNION }, { "--union", OPT_UNION },
  { "-i", OPT_INTRSEC }, { "--intersection", OPT_INTRSEC }, { "-d", OPT_DIFF }, { "--difference", OPT_DIFF },
  { "-dd", OPT_DDIFF }, { "--double_difference", OPT_DDIFF }, { "-du", OPT_DU }, { "--diff_union", OPT_DU },
  { "--count_only", OPT_COUNT_ONLY }, { "-r", OPT_RULE }, { "--rule", OPT_RULE }, { "-ss", OPT_SUBSET }, { "--subset", OPT_SUBSET },
  { "--seed", OPT_SEED }, { "--print_operation", OPT_PRINT_OP }, { "--disable_scouts", OPT_NOSCOUTS }, { "--stream", OPT_STREAM },
  { "-D", OPT_DEBUG },
};

static const char *const HELP_LINES[] = {
  "Usage: glistcompare INPUTLIST1 [INPUTLIST2...] METHOD [OPTIONS]",
  "Options:",
  "    -v, --version            - print version information and exit",
  "    -h, --help               - print this usage screen and exit",
  "    -u, --union              - union of input lists",
  "    -i, --intersection       - intersection of input lists",
  "    -d, --difference         - difference of input lists",
  "    -dd, --double_difference - double difference of input lists",
  "    -du, --diff_union        - subtract first list from the second and finds difference",
  "    -mm, --mismatch   NUMBER - specify number of mismatches (default 0, can be used with -diff and -ddiff)",
  "    -c, --cutoff NUMBER      - specify frequency cut-off (default 1)",
  "    -o, --outputname STRING  - specify output name (default \"out\")",
  "    -r, --rule STRING        - specify rule how final frequencies are calculated (default, add, subtract, min, max, first, second, 1, 2)",
  "                               NOTE: rules min, subtract, first and second can only be used with finding the intersection.",
  "    -ss, --subset METHOD SIZE - make subset with given method (rand, rand_unique, rand_weighted_unique)",
  "    --seed INTEGER           - Set seed of random number generator (default uses start time)",
  "    --count_only             - output count of k-mers instead of k-mers themself",
  "    --disable_scouts         - disable list read-ahead in background thread",
  "    --stream                 - read input as stream (do not memory map files)",
  "    -D                       - increase debug level",
};

static void print_version (void)
{
  fprintf (stdout, "glistcompare version %u.%u.%u (%s)\n", GT4_VERSION_MAJOR, GT4_VERSION_MINOR, GT4_VERSION_MICRO, GT4_VERSION_QUALIFIER);
}

static void print_help (int exit_value)
{
  print_version ();
  for (size_t i = 0; i < sizeof HELP_LINES / sizeof HELP_LINES[0]; i++) fprintf (stdout, "%s\n", HELP_LINES[i]);
  exit (exit_value);
}

static int lookup_option (const char *arg)
{
  for (size_t i = 0; i < sizeof OPTIONS / sizeof OPTIONS[0]; i++)
    if (!strcmp (arg, OPTIONS[i].name)) return OPTIONS[i].opt;
  return -1;
}

static double now_seconds (void)
{
  struct timeval tv;
  gettimeofday (&tv, NULL);
  return tv.tv_sec + tv.tv_usec * 1e-6;
}

#define DOWNLOAD_CHUNK (4u << 20) /* records per device -> host -> file step (48 MiB) */

/* device list -> "<final>.tmp" -> rename, header back-patched with the kernel's totals */
static int write_list_file (gt4hip_context *ctx, const gt4hip_list *list, unsigned int word_length, uint64_t n_words, uint64_t total_count,
                            const char *final_name, unsigned int mode)
{
  char tmp_name[2100];
  snprintf (tmp_name, sizeof tmp_name, "%.2048s.tmp", final_name);
  GT4ListWriter w;
  if (gt4_listwriter_begin (&w, tmp_name, word_length, mode)) {
    fprintf (stderr, "Error: Cannot create output file %s\n", tmp_name);
    return 1;
  }
  int bad = 0;
  void *buf = n_words ? malloc ((size_t) (n_words < DOWNLOAD_CHUNK ? n_words : DOWNLOAD_CHUNK) * 12u) : NULL;
  if (n_words && !buf) bad = 1;
  for (uint64_t first = 0; first < n_words && !bad; first += DOWNLOAD_CHUNK) {
    const uint64_t cnt = n_words - first < DOWNLOAD_CHUNK ? n_words - first : DOWNLOAD_CHUNK;
    if (gt4hip_list_download_range (ctx, list, first, cnt, buf)) {
      fprintf (stderr, "Error: reading results back from the GPU failed: %s\n", gt4hip_last_error (ctx));
      bad = 1;
    } else if (gt4_listwriter_append (&w, buf, cnt)) {
      fprintf (stderr, "Error: writing %s failed: %s\n", tmp_name, strerror (errno));
      bad = 1;
    }
  }
  free (buf);
  if (bad) {
    gt4_listwriter_abort (&w);
    unlink (tmp_name);
    return 1;
  }
  if (gt4_listwriter_finish (&w, n_words, total_count)) {
    fprintf (stderr, "Error: writing %s failed: %s\n", tmp_name, strerror (errno));
    unlink (tmp_name);
    return 1;
  }
  if (rename (tmp_name, final_name)) {
    fprintf (stderr, "Error: Cannot rename %s to %s\n", tmp_name, final_name);
    return 1;
  }
  return 0;
}


/* "<n>[K|M|G]" -> bytes */
static uint64_t parse_bytes (const char *s)
{
  if (!s || !*s) return 0;
  char *end;
  double v = strtod (s, &end);
  if (*end == 'K' || *end == 'k') v *= 1024.0;
  else if (*end == 'M' || *end == 'm') v *= 1024.0 * 1024.0;
  else if (*end == 'G' || *end == 'g') v *= 1024.0 * 1024.0 * 1024.0;
  return v > 0 ? (uint64_t) v : 0;
}

int main (int argc, const char *argv[])
{
  const char *fnames[MAX_FILES];
  unsigned int nfiles = 0;
  int rule = GT4HIP_RULE_DEFAULT;
  unsigned int cutoff = 1, nmm = 0, count_override = 1;
  int find_union = 0, find_intrsec = 0, find_diff = 0, find_ddiff = 0, subtraction = 0, countonly = 0, print_operation = 0;
  int find_subset = 0, stream = 0, debug = 0;
  int n_gpus = getenv ("GT4HIP_GPUS") ? atoi (getenv ("GT4HIP_GPUS")) : 0;
  /* lines of this implementation's own (device, kernel times, chunk plan): -D stays the reference's transcript */
  const int verbose = getenv ("GT4HIP_VERBOSE") && atoi (getenv ("GT4HIP_VERBOSE"));
  const char *outputname = "out";
  char *end;

  if (argc <= 1) print_help (1);

  /* ---- argv (reference :107-230; every quirk of its hand-rolled loop is kept) */
  for (int i = 1; i < argc; i++) {
    const char *arg = argv[i];
    if (arg[0] != '-') {
      if (nfiles >= MAX_FILES) {
        fprintf (stderr, "Too many file arguments (max %d)\n", MAX_FILES);
        print_help (1);
      }
      fnames[nfiles++] = arg;
      continue;
    }
    switch (lookup_option (arg)) {
      case OPT_VERSION:
        print_version ();
        return 0;
      case OPT_HELP:
        print_help (0);
        break;
      case OPT_OUT:
        if (!argv[i + 1] || argv[i + 1][0] == '-') {
          fprintf (stderr, "Warning: No output name specified!\n");
          i += 1; /* the reference skips the next argument here as well */
          break;
        }
        outputname = argv[++i];
        break;
      case OPT_CUTOFF:
        if (!argv[i + 1]) {
          fprintf (stderr, "Warning: No frequency cut-off specified! Using the default value: %d.\n", cutoff);
          break;
        }
        cutoff = (unsigned int) strtol (argv[i + 1], &end, 10);
        if (*end != 0) {
          fprintf (stderr, "Error: Invalid frequency cut-off: %s! Must be an integer.\n", argv[i + 1]);
          print_help (1);
        }
        i += 1;
        break;
      case OPT_MM:
        if (!argv[i + 1]) {
          fprintf (stderr, "Warning: No number of mismatches specified!");
          break;
        }
        nmm = (unsigned int) strtol (argv[i + 1], &end, 10);
        if (*end != 0) {
          fprintf (stderr, "Error: Invalid number of mismatches: %s! Must be an integer.\n", argv[i + 1]);
          print_help (1);
        }
        i += 1;
        break;
      case OPT_UNION: find_union = 1; break;
      case OPT_INTRSEC: find_intrsec = 1; break;
      case OPT_DIFF: find_diff = 1; break;
      case OPT_DDIFF: find_ddiff = 1; break;
      case OPT_DU:
        find_diff = 1;
        subtraction = 1;
        break;
      case OPT_COUNT_ONLY: countonly = 1; break;
      case OPT_RULE: {
        static const struct { const char *name; int rule; } RULES[] = {
          { "default", GT4HIP_RULE_DEFAULT }, { "add", GT4HIP_RULE_ADD }, { "sum", GT4HIP_RULE_ADD }, { "subtract", GT4HIP_RULE_SUBTRACT },
          { "min", GT4HIP_RULE_MIN }, { "max", GT4HIP_RULE_MAX }, { "first", GT4HIP_RULE_FIRST }, { "second", GT4HIP_RULE_SECOND },
        };
        i += 1;
        if (i >= argc) print_help (1);
        if (argv[i][0] >= '1' && argv[i][0] <= '9') {
          rule = GT4HIP_RULE_NUMBER;
          count_override = (unsigned int) strtol (argv[i], &end, 10);
        } else {
          for (size_t r = 0; r < sizeof RULES / sizeof RULES[0]; r++)
            if (!strcmp (argv[i], RULES[r].name)) rule = RULES[r].rule;
          /* an unknown rule name is silently ignored, as in the reference */
        }
        break;
      }
      case OPT_SUBSET:
        find_subset = 1;
        i += 1;
        if (i >= argc) print_help (1);
        if (strcmp (argv[i], "rand") && strcmp (argv[i], "rand_unique") && strcmp (argv[i], "rand_weighted_unique")) print_help (1);
        i += 1;
        if (i >= argc) print_help (1);
        (void) strtoll (argv[i], &end, 10);
        if (*end != 0) {
          fprintf (stderr, "Error: Invalid subset size: %s! Must be an integer.\n", argv[i]);
          print_help (1);
        }
        break;
      case OPT_SEED:
        i += 1;
        if (i >= argc) print_help (1);
        break; /* only --subset draws random numbers */
      case OPT_PRINT_OP: print_operation = 1; break;
      case OPT_NOSCOUTS: break;
      case OPT_STREAM: stream = 1; break;
      case OPT_DEBUG: debug += 1; break;
      case OPT_GPUS:
        i += 1;
        if (i >= argc) print_help (1);
        n_gpus = atoi (argv[i]);
        if (n_gpus < 1) {
          fprintf (stderr, "Error: Invalid number of GPUs: %s!\n", argv[i]);
          print_help (1);
        }
        break;
      default:
        fprintf (stderr, "Unknown argument: %s!\n", arg);
        print_help (1);
    }
  }
  if (debug) fprintf (stderr, "Rule: %d\n", rule);
  if (debug) fprintf (stderr, "Num files: %d\n", nfiles);
  if (nmm || find_subset) {
    if (stream) fprintf (stderr, "Warning: Subset and mismatches are incompatible with streaming, using mapping\n");
    stream = 0;
  }

  /* ---- open the inputs (reference :250-290) */
  static GT4ListFile files[MAX_FILES];
  unsigned int wlen = 0, err = 0;
  for (unsigned int f = 0; f < nfiles; f++) {
    uint32_t code;
    files[f].file_map = NULL;
    if (gt4_listfile_sniff (fnames[f], &code)) {
      fprintf (stderr, "Error: Cannot open %s\n", fnames[f]);
      err = 1;
      continue;
    }
    if (code != GT4_LIST_CODE_VALUE && code != GT4_INDEX_CODE_VALUE) {
      /* the reference reports both: no object was made, so the interface lookup fails as well (:272-279) */
      fprintf (stderr, "Error: File %s has unknown format\n", fnames[f]);
      fprintf (stderr, "Error: File %s is invalid or corrupted\n", fnames[f]);
      err = 1;
      continue;
    }
    /* a GT4I index is read as the sorted (k-mer, number of locations) list it contains (:269-270) */
    if (code == GT4_INDEX_CODE_VALUE ? gt4_indexfile_open (fnames[f], GT4_VERSION_MAJOR, &files[f])
                                     : gt4_listfile_open (fnames[f], GT4_VERSION_MAJOR, &files[f])) {
      fprintf (stderr, "Error: File %s is invalid or corrupted\n", fnames[f]);
      err = 1;
      continue;
    }
    if (!wlen) {
      wlen = files[f].header.word_length;
    } else if (files[f].header.word_length != wlen) {
      fprintf (stderr, "Error: File %s has different word length (%u != %u)\n", fnames[f], files[f].header.word_length, wlen);
      err = 1;
    }
  }
  if (err) {
    fprintf (stderr, "Stopping...\n");
    exit (1);
  }
  if (find_subset) {
    fprintf (stderr, "Error: --subset is not part of the GPU set-operation path\n");
    exit (1);
  }

  /* ---- validity checks, in the reference's order (:317-352) */
  if (nfiles < 2) {
    fprintf (stderr, "Error: At least 2 list/index files are needed\n");
    exit (1);
  }
  if (nfiles > 2) {
    if (!(find_union || find_intrsec) || find_diff || find_ddiff) {
      fprintf (stderr, "Error: Algorithm incompatible with multiple files!\n");
      print_help (1);
    }
    if (nmm) {
      fprintf (stderr, "Error: Multiple files are not compatible with mismatches!\n");
      print_help (1);
    }
  }
  if (find_ddiff) find_diff = 1;
  if (!find_diff && nmm) fprintf (stderr, "Warning: Number of mismatches are not used!\n");
  if (!find_diff && subtraction) fprintf (stderr, "Warning: Subtraction is not used!\n");
  if (strlen (outputname) > 200) {
    fprintf (stderr, "Error: Output name exceeds the 200 character limit.\n");
    exit (1);
  }
  if (!find_intrsec && (rule == GT4HIP_RULE_MIN || rule == GT4HIP_RULE_FIRST || rule == GT4HIP_RULE_SECOND)) {
    fprintf (stderr, "Error: Rules min, fist and second can only be used with finding the intersection.\n");
    exit (1);
  }
  if ((!find_intrsec && !find_diff) && (rule == GT4HIP_RULE_SUBTRACT)) {
    fprintf (stderr, "Error: Rule subtract can only be used with intersection and difference.\n");
    exit (1);
  }
  if (print_operation) {
    fprintf (stdout, "Operation\t%s%s%s%s\trule\t%u\nFiles\t%u\n", find_union ? "U" : "", find_intrsec ? "I" : "", find_diff ? "D" : "",
             find_ddiff ? "X" : "", rule, nfiles);
    for (unsigned int f = 0; f < nfiles; f++) fprintf (stdout, "%u\t%s\n", f, fnames[f]);
  }
  if (nmm) {
    fprintf (stderr, "Error: -mm (mismatch difference) is not part of the GPU set-operation path\n");
    exit (1);
  }


  /* ---- key-range shards: several GPUs and / or chunks streamed through the device memory */
  static const char *const SUFFIX[4] = { "union", "intrsec", "0_diff1", "0_diff2" };
  uint64_t hbm_limit = parse_bytes (getenv ("GT4HIP_HBM_LIMIT"));
  int use_shards = n_gpus >= 1 || hbm_limit != 0;
  int auto_budget = 0;
  gt4hip_context *ctx = NULL;
  if (!use_shards) {
    uint64_t in_records = 0;
    for (unsigned int f = 0; f < nfiles; f++) in_records += files[f].header.n_words;
    const int pipeline_off = getenv ("GT4HIP_PIPELINE") && !atoi (getenv ("GT4HIP_PIPELINE"));
    if (12 * in_records >= (4ull << 30) && !pipeline_off) {
      /* big inputs: key-range chunks through the loader / merger / writer pipeline, so that reading the next
       * chunk and writing the previous one overlap the merge (GT4HIP_PIPELINE=0 keeps everything in one
       * piece).  No context is created here: the worker that runs the job (in this process) creates the
       * only one, measures the device's free memory and chooses the chunk budget itself -- the HIP runtime
       * starts once per process. */
      use_shards = 1;
      auto_budget = 1;
    }
  }
  if (!use_shards) {
    const char *dev = getenv ("GT4HIP_DEVICE");
    if (gt4hip_create (dev ? atoi (dev) : 0, &ctx)) {
      fprintf (stderr, "Error: %s\n", gt4hip_last_error (NULL));
      exit (1);
    }
    if (verbose) fprintf (stderr, "Device: %s\n", gt4hip_device_info (ctx));
    /* inputs + worst-case outputs (+ the N-way tree's intermediates) must fit, else stream in chunks */
    uint64_t free_b = 0, total_b = 0, need = 0, in_records = 0;
    gt4hip_device_memory (ctx, &free_b, &total_b);
    for (unsigned int f = 0; f < nfiles; f++) in_records += files[f].header.n_words;
    if (nfiles == 2) need = 12 * in_records * (1 + (uint64_t) (find_union + find_intrsec + find_diff + find_ddiff));
    else need = 12 * in_records * 4;
    if (free_b && need > free_b / 100 * 85) {
      if (verbose) fprintf (stderr, "Inputs and outputs need %llu bytes, %llu are free: streaming in key-range chunks\n", (unsigned long long) need, (unsigned long long) free_b);
      gt4hip_destroy (ctx);
      ctx = NULL;
      use_shards = 1;
    }
  }
  if (use_shards) {
    GT4ShardJob job;
    memset (&job, 0, sizeof job);
    job.n_files = nfiles;
    job.files = files;
    job.word_length = wlen;
    job.n_ranks = n_gpus >= 1 ? n_gpus : 1;
    job.hbm_limit = hbm_limit;
    job.auto_budget = auto_budget;
    job.gather_rccl = getenv ("GT4HIP_GATHER") && !strcmp (getenv ("GT4HIP_GATHER"), "rccl");
    job.debug = verbose;
    job.prm.rule = rule;
    job.prm.cutoff = cutoff;
    job.prm.subtract = subtraction;
    job.prm.count_override = count_override;
    job.prm.count_only = countonly;
    char names[4][2048];
    int v = 0;
    if (nfiles == 2) {
      if (debug) {
        fprintf (stderr, "compare_wordmaps: methods %u/%u/%u/%u\n", find_union, find_intrsec, find_diff, find_ddiff);
        fprintf (stderr, "compare_wordmaps: List 1: %llu entries\n", (unsigned long long) files[0].header.n_words);
        fprintf (stderr, "compare_wordmaps; List 2: %llu entries\n", (unsigned long long) files[1].header.n_words);
      }
      job.mode = GT4_SHARD_PAIR;
      job.prm.ops = (find_union ? GT4HIP_OP_UNION : 0) | (find_intrsec ? GT4HIP_OP_INTRSEC : 0) | (find_diff ? GT4HIP_OP_DIFF1 : 0) |
                    (find_ddiff ? GT4HIP_OP_DIFF2 : 0);
      job.out_mode = 0666;
      for (int s = 0; s < 4; s++) {
        if (!((job.prm.ops >> s) & 1u) || countonly) continue;
        snprintf (names[s], sizeof names[s], "%s_%d_%s.list", outputname, wlen, SUFFIX[s]);
        job.out_name[s] = names[s];
      }
      GT4ShardResult res;
      if (job.prm.ops) {
        if (gt4_shard_run (&job, &res)) exit (1);
        if (verbose) fprintf (stderr, "Sharded run: %u chunks over %d GPU(s)\n", res.n_chunks, job.n_ranks);
        for (int s = 0; s < 4; s++) {
          if (!((job.prm.ops >> s) & 1u)) continue;
          if (countonly) fprintf (stdout, "NUnique\t%llu\nNTotal\t%llu\n", (unsigned long long) res.n_words[s], (unsigned long long) res.total_count[s]);
          else if (debug && s >= 2) fprintf (stderr, "Renaming %s.tmp to %s\n", names[s], names[s]);
        }
      }
    } else {
      for (int pass = 0; pass < 2; pass++) {
        const int is_union = pass == 0;
        if (is_union ? !find_union : !find_intrsec) continue;
        job.mode = is_union ? GT4_SHARD_UNION_MULTI : GT4_SHARD_INTERSECT_MULTI;
        job.out_mode = 0644;
        job.out_name[0] = NULL;
        if (!countonly) {
          snprintf (names[0], sizeof names[0], "%s_%d_%s.list", outputname, wlen, is_union ? "union" : "intrsec");
          job.out_name[0] = names[0];
        }
        GT4ShardResult res;
        const double t_s = now_seconds ();
        const int rc = gt4_shard_run (&job, &res);
        const double t_e = now_seconds ();
        if (rc && res.rule_rejected) {
          fprintf (stderr, "%s\n", res.message);
          v = 1;
          continue;
        }
        if (rc) exit (1);
        v = 0;
        if (debug) {
          unsigned long long total = 0;
          for (unsigned int f = 0; f < nfiles; f++) total += files[f].header.n_words;
          fprintf (stderr, "Combined %u maps: input %llu (%.3f Mwords/s) output %llu (%.3f Mwords/s)\n", nfiles, is_union ? total : 0ull,
                   (is_union ? total : 0ull) / (1000000 * (t_e - t_s)), (unsigned long long) res.n_words[0], res.n_words[0] / (1000000 * (t_e - t_s)));
        }
        if (countonly || debug) fprintf (stdout, "NUnique\t%llu\nNTotal\t%llu\n", (unsigned long long) res.n_words[0], (unsigned long long) res.total_count[0]);
      }
    }
    for (unsigned int f = 0; f < nfiles; f++) gt4_listfile_close (&files[f]);
    return v ? 1 : 0;
  }

  static gt4hip_list *lists[MAX_FILES];
  for (unsigned int f = 0; f < nfiles; f++) {
    if (files[f].index_kmers ? gt4hip_list_upload_index (ctx, files[f].index_kmers, files[f].header.n_words, files[f].index_locations, wlen, &lists[f])
                             : gt4hip_list_upload (ctx, files[f].records, files[f].header.n_words, wlen, &lists[f])) {
      fprintf (stderr, "Error: uploading %s to the GPU failed: %s\n", fnames[f], gt4hip_last_error (ctx));
      exit (1);
    }
    /* the reference trusts its inputs to be strictly ascending (results are undefined otherwise);
     * GT4HIP_CHECK_SORTED=1 verifies that on the device before merging */
    if (getenv ("GT4HIP_CHECK_SORTED") && atoi (getenv ("GT4HIP_CHECK_SORTED"))) {
      int sorted = 0;
      if (gt4hip_list_is_sorted (ctx, lists[f], &sorted) || !sorted) {
        fprintf (stderr, "Error: File %s is not sorted by k-mer (strictly ascending, unique)\n", fnames[f]);
        exit (1);
      }
    }
  }

  int v = 0;
  if (nfiles == 2) {
    /* ---- compare_wordmaps (reference :789-955) */
    if (debug) {
      fprintf (stderr, "compare_wordmaps: methods %u/%u/%u/%u\n", find_union, find_intrsec, find_diff, find_ddiff);
      fprintf (stderr, "compare_wordmaps: List 1: %llu entries\n", (unsigned long long) files[0].header.n_words);
      fprintf (stderr, "compare_wordmaps; List 2: %llu entries\n", (unsigned long long) files[1].header.n_words);
    }
    gt4hip_compare_params prm;
    memset (&prm, 0, sizeof prm);
    prm.ops = (find_union ? GT4HIP_OP_UNION : 0) | (find_intrsec ? GT4HIP_OP_INTRSEC : 0) | (find_diff ? GT4HIP_OP_DIFF1 : 0) |
              (find_ddiff ? GT4HIP_OP_DIFF2 : 0);
    prm.rule = rule;
    prm.cutoff = cutoff;
    prm.subtract = subtraction;
    prm.count_override = count_override;
    prm.count_only = countonly;
    gt4hip_compare_result res;
    memset (&res, 0, sizeof res);
    if (prm.ops) {
      if (gt4hip_compare (ctx, lists[0], lists[1], &prm, &res)) {
        fprintf (stderr, "Error: %s\n", gt4hip_last_error (ctx));
        exit (1);
      }
      if (verbose) fprintf (stderr, "GPU merge kernel: %.3f ms (%llu tiles), device total %.3f ms\n", res.merge_kernel_ms,
                          (unsigned long long) res.merge_tiles, res.device_ms);
    }
    if (countonly) {
      for (int s = 0; s < 4; s++)
        if ((prm.ops >> s) & 1u) fprintf (stdout, "NUnique\t%llu\nNTotal\t%llu\n", (unsigned long long) res.n_words[s], (unsigned long long) res.total_count[s]);
    } else {
      /* every requested output to its own "<name>.tmp" at once (the copy threads are dealt to the
       * files), then header back-patch and rename in the reference's order (:907-953) */
      char name[4][2048], tmp_name[4][2100];
      GT4ListWriter w[4];
      const gt4hip_list *wl[4];
      uint64_t wfirst[4], wcount[4], woff[4];
      int wfd[4], ws[4];
      uint32_t nw = 0;
      int bad = 0;
      for (int s = 0; s < 4 && !bad; s++) {
        if (!((prm.ops >> s) & 1u)) continue;
        snprintf (name[s], sizeof name[s], "%s_%d_%s.list", outputname, wlen, SUFFIX[s]);
        snprintf (tmp_name[s], sizeof tmp_name[s], "%s.tmp", name[s]);
        /* fopen (.., "w") in the reference: mode 0666 minus umask */
        if (gt4_listwriter_begin (&w[s], tmp_name[s], wlen, 0666)) {
          fprintf (stderr, "Error: Cannot create output file %s\n", tmp_name[s]);
          bad = 1;
          break;
        }
        wl[nw] = res.out[s];
        wfirst[nw] = 0;
        wcount[nw] = res.n_words[s];
        wfd[nw] = w[s].fd;
        woff[nw] = 48;
        ws[nw] = s;
        nw++;
      }
      if (!bad && nw && gt4hip_lists_write_fd (ctx, nw, wl, wfirst, wcount, wfd, woff)) {
        fprintf (stderr, "Error: writing the results failed: %s\n", gt4hip_last_error (ctx));
        bad = 1;
      }
      for (uint32_t q = 0; q < nw; q++) {
        const int s = ws[q];
        if (bad) {
          gt4_listwriter_abort (&w[s]);
          unlink (tmp_name[s]);
          continue;
        }
        if (debug && s >= 2) fprintf (stderr, "Renaming %s to %s\n", tmp_name[s], name[s]);
        if (gt4_listwriter_finish (&w[s], res.n_words[s], res.total_count[s])) {
          fprintf (stderr, "Error: writing %s failed: %s\n", tmp_name[s], strerror (errno));
          unlink (tmp_name[s]);
          bad = 1;
        } else if (rename (tmp_name[s], name[s])) {
          fprintf (stderr, "Error: Cannot rename %s to %s\n", tmp_name[s], name[s]);
          bad = 1;
        }
        gt4hip_list_free (res.out[s]);
      }
      if (bad) exit (1);
    }
  } else {
    /* ---- union_multi / intersect_multi (reference :366-422) */
    for (int pass = 0; pass < 2; pass++) {
      const int is_union = pass == 0;
      if (is_union ? !find_union : !find_intrsec) continue;
      gt4hip_multi_result res;
      memset (&res, 0, sizeof res);
      const double t_s = now_seconds ();
      int rc = is_union ? gt4hip_union_multi (ctx, (const gt4hip_list *const *) lists, nfiles, cutoff, rule, count_override, countonly, &res)
                        : gt4hip_intersect_multi (ctx, (const gt4hip_list *const *) lists, nfiles, cutoff, rule, count_override, countonly, &res);
      const double t_e = now_seconds ();
      if (rc == GT4HIP_ERULE) {
        fprintf (stderr, "%s\n", gt4hip_last_error (ctx));
        v = 1; /* the reference returns 1 from the merge and exits 1 without an output file */
        continue;
      }
      if (rc) {
        fprintf (stderr, "Error: %s\n", gt4hip_last_error (ctx));
        exit (1);
      }
      v = 0;
      if (debug) {
        unsigned long long total = 0;
        for (unsigned int f = 0; f < nfiles; f++) total += files[f].header.n_words;
        fprintf (stderr, "Combined %u maps: input %llu (%.3f Mwords/s) output %llu (%.3f Mwords/s)\n", nfiles, is_union ? total : 0ull,
                 (is_union ? total : 0ull) / (1000000 * (t_e - t_s)), (unsigned long long) res.n_words, res.n_words / (1000000 * (t_e - t_s)));
      }
      if (!countonly) {
        char name[2048];
        snprintf (name, sizeof name, "%s_%d_%s.list", outputname, wlen, is_union ? "union" : "intrsec");
        /* creat (.., 0644) in the reference */
        if (write_list_file (ctx, res.out, wlen, res.n_words, res.total_count, name, 0644)) exit (1);
        gt4hip_list_free (res.out);
      }
      if (countonly || debug) fprintf (stdout, "NUnique\t%llu\nNTotal\t%llu\n", (unsigned long long) res.n_words, (unsigned long long) res.total_count);
    }
  }

  for (unsigned int f = 0; f < nfiles; f++) {
    gt4hip_list_free (lists[f]);
    gt4_listfile_close (&files[f]);
  }
  gt4hip_destroy (ctx);
  return v ? 1 : 0;
}
