"""Summarises a tools/collect_profiles.sh output directory into small CSV/JSON files.

    python3 tools/summarize_profiles.py gpurun_out/prof_<tag>/<workload> <tag> [workload]

Writes next to the raw data (directory `summary/`): <tag>_bench.json, <tag>_kernel_stats.csv,
<tag>_pmc_fetch_size.csv, <tag>_pmc_write_size.csv and traffic_<workload>.json (what bench.py replays
as `roofline.traffic`, with the commit the passes ran on: GT4_COMMIT).  FETCH_SIZE / WRITE_SIZE
are in KiB; FETCH_SIZE is doubled on gfx950 as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes."""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "intersect"
ROUND = os.environ.get("GT4_ROUND", "round6")
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)


def csrc_sha16():
    """the hash bench.py ties the replayed traffic to (same function there)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "genometester4_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "Makefile"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


bench = None
try:
    with open(os.path.join(src, "bench.json")) as f:
        for line in f:
            if line.startswith("{"):
                bench = json.loads(line)
    with open(os.path.join(dst, tag + "_bench.json"), "w") as f:
        json.dump(bench, f, indent=1)
except Exception as e:  # noqa
    print("no bench line:", e)

p = find("stats", "_kernel_stats.csv")
if p:
    with open(p) as f, open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as g:
        g.write(f.read())
    print("kernel stats:", p)


def pmc(sub, counter):
    p = find(sub, "_counter_collection.csv")
    if not p:
        print("no counter file for", counter)
        return None, None
    per = {}
    with open(p) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("gt4::", "").split("(")[0]
            if short.startswith("km32::"):  # (the N-way sources are compiled twice: 8 and 32 lists per launch)
                short = short[len("km32::"):].replace("k_nway_", "k_nway32_", 1)
            elif short.startswith("km8::"):
                short = short[len("km8::"):]
            per.setdefault(short, {}).setdefault(row.get("Dispatch_Id"), 0.0)
            per[short][row.get("Dispatch_Id")] += float(row["Counter_Value"])
    out = os.path.join(dst, "%s_pmc_%s.csv" % (tag, counter.lower()))
    with open(out, "w") as g:
        g.write("kernel,counter,per_launch_values_KiB\n")
        for k, d in sorted(per.items()):
            g.write("%s,%s,%s\n" % (k.replace(",", ";"), counter, " ".join("%.1f" % v for v in d.values())))
    return per, out


fetch, fpath = pmc("fetch", "FETCH_SIZE")
write, wpath = pmc("write", "WRITE_SIZE")
if fetch and write:
    # dominant kernel = the merge kernel instantiation with the most fetched bytes in total
    cand = [k for k in fetch if k.startswith("k_pair_merge") or k.startswith("k_nway") or k.startswith("k_radix")]
    if workload == "sort":
        cand = [k for k in cand if k.startswith("k_radix")]
    if workload.startswith(("union", "table")) and any(k.startswith(("k_nway_merge", "k_nway32_merge")) for k in cand):
        # the N-way workloads build their lists with the pair kernel (class unions): the tile kernel is what is measured
        cand = [k for k in cand if k.startswith("k_nway")] or cand
    dom = max(cand, key=lambda k: sum(fetch[k].values()))
    fv = list(fetch[dom].values())
    wv = list(write.get(dom, {}).values())
    fb = 2.0 * 1024.0 * sum(fv) / len(fv)
    wb = 1024.0 * sum(wv) / len(wv) if wv else 0.0
    # every merge kernel launch of one bench step together (the N-way union is several launches per step)
    steps = 3  # --steps 2 --warmup 1
    all_f = 2.0 * 1024.0 * sum(sum(fetch[k].values()) for k in cand)
    all_w = 1024.0 * sum(sum(write.get(k, {}).values()) for k in cand)
    cfg = (bench or {}).get("config", {})
    tj = {
        "workload": workload,
        "commit": os.environ.get("GT4_COMMIT"),
        "csrc_sha16": csrc_sha16(),  # bench.py refuses the replay when the kernel sources differ
        "n_per_list": cfg.get("entries_per_list_per_gpu", cfg.get("entries_per_list", cfg.get("words"))),
        "merge_kernels_hbm_bytes_per_step_incl_list_generation": (all_f + all_w) / steps,
        "kernel": dom,
        "hbm_bytes_per_launch": fb + wb,
        "fetch_bytes_corrected": fb,
        "write_bytes": wb,
        "launches_averaged": [len(fv), len(wv)],
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py "
                  "--steps 2 --warmup 1 --no-cpu-baseline`; KiB units; FETCH_SIZE doubled per the gfx950 "
                  "correction of MI355X_MICROARCH.md (HBM section); WRITE_SIZE as is",
        "source": ["profiles/%s/" % ROUND + os.path.basename(fpath), "profiles/%s/" % ROUND + os.path.basename(wpath)],
    }
    if workload.startswith("union") and dom.startswith(("k_nway_merge", "k_nway32_merge")):
        # one launch of the dominant instantiation per union; the call's other kernels (key samples, their
        # merges, the tile partition) are counted into the per-union figure
        unions = max(1, len(fv))
        tj["unions_per_pmc_pass"] = unions
        tj["all_nway_kernels_hbm_bytes_per_union"] = (all_f + all_w) / unions
        tj["note"] = ("one launch of the one-pass N-way tile kernel per 8-way union (hbm_bytes_per_launch); the per-union figure adds "
                      "the sample, partition and sample-merge kernels of the call; the bench runs every step twice (with and without "
                      "the gather), so a PMC pass of --steps 2 --warmup 1 holds more unions than steps")
    elif workload.startswith("union8"):
        # 7 pair merges per union: count the unions of the pass from the launches themselves
        launches = sum(len(fetch[k]) for k in cand)
        unions = max(1, launches // 7)
        tj["unions_per_pmc_pass"] = unions
        tj["merge_kernels_hbm_bytes_per_union"] = (all_f + all_w) / unions
        tj["note"] = ("7 pair merges per 8-way union (4 + 2 raw levels, 1 final); hbm_bytes_per_launch is the average launch of "
                      "the dominant instantiation only; the bench runs every step twice (with and without the gather), so a "
                      "PMC pass of --steps 2 --warmup 1 holds more unions than steps")
    with open(os.path.join(dst, "traffic_%s.json" % workload), "w") as f:
        json.dump(tj, f, indent=1)
    print(json.dumps(tj))
