"""Prints the 'final state' table of profiles/<round>/README.md from the committed files (bench lines, rocprofv3 kernel
summaries, traffic records).   python tools/profiles_readme_table.py [round5]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "round6"
R = os.path.join(ROOT, "profiles", RND)
TAG = "r%sfinal" % RND[-1]
ORDER = [("intersect", "intersect (default line; config 1, 2 x 2e9 k=25, -i)", None), ("c2", "c2 (config 2, -u -d -c 3)", None),
         ("union8", "union8 (config 3 on one GPU, stride keys)", "call"), ("union8_iid", "union8 --dist iid", "call"),
         ("union8_genomic", "union8 --dist genomic", "call"), ("union8_clustered", "union8 --dist clustered (declined: pairwise tree)", "whole tree"),
         ("union32", "union32 (32 x 1.25e8, even lists the same: levels of eight)", "all levels"),
         ("union32_disjoint", "union32 --dist disjoint (ONE pass over 32 lists)", "call"), ("intersect8", "intersect8 (8 x 5e8, half shared)", "whole chain"),
         ("sort", "sort (N2, 1e9 words)", None), ("table", "table (N3, 6 x 1e8)", "whole call"),
         ("table32", "table --nt-lists 32 --nt 20000000 (N3, 32 x 2e7: ONE launch of km32)", "whole call")]
print("| workload (`bench.py --workload W [--dist D]`) | step | dominant kernel, average launch (rocprofv3) | algorithmic bytes | frac of 8 TB/s | PMC traffic per launch |")
print("|---|---|---|---|---|---|")
for n, label, what in ORDER:
    b = json.load(open(os.path.join(R, "%s_%s_bench.json" % (TAG, n))))
    ro = b["roofline"]
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_%s.json" % n)))
    dom = t["kernel"]
    want = dom.replace("k_nway32_", "k_nway_")
    avg = calls = 0
    for r in csv.DictReader(open(os.path.join(R, "%s_%s_kernel_stats.csv" % (TAG, n)))):
        short = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gt4::", "").split("(")[0]
        if short.replace("km8::", "").replace("km32::", "") == want and (("km32::" in short) == dom.startswith("k_nway32_")):
            avg, calls = float(r["AverageNs"]) / 1e6, int(r["Calls"])
            break
    unit = "words/s" if n == "sort" else "k-mers/s"
    step = "%.2f ms = %.0f G %s, verified %s" % (b["ms_per_step"], b["value"] / 1e9, unit, b.get("verified"))
    if b.get("self_check"):
        step += ", self-check %s" % b["self_check"]
    kname = ("km32::" + want) if dom.startswith("k_nway32_") else dom
    frac = "**%.3f**" % ro["frac"]
    if what == "call" and ro.get("whole_call_frac") is not None:
        frac += " (call %.3f)" % ro["whole_call_frac"]
    elif what:
        frac += " (%s)" % what
    print("| `%s` | %s | `%s` %.2f ms (x %d in the stats run) | %.1f GB | %s | %.2f GB |" % (label, step, kname, avg, calls, ro["algorithmic_bytes_per_launch"] / 1e9, frac, t["hbm_bytes_per_launch"] / 1e9))
    if n == "intersect" and b.get("union8"):
        u = b["union8"]
        print("| ... its `union8` record | %.2f ms per step (merge_only %.2f) | `km8::k_nway_merge<1024,4,1,1>` | 78.0 GB | %.3f (call %.3f) | -- |" % (u["ms_per_step_with_gather"], u["merge_only_ms_per_step"], u["roofline_frac"], u["whole_call_frac"]))
