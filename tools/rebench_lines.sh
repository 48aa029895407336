mkdir -p gpurun_out/r6final2
run() { name=$1; shift; timeout 600 python bench.py --steps 10 --warmup 3 "$@" 2> gpurun_out/r6final2/$name.err | grep '^{' | tail -1 > gpurun_out/r6final2/$name.json; }
run intersect --steps 20 --warmup 5
run c2 --workload c2
run union8 --workload union8
run union8_iid --workload union8 --dist iid
run union8_genomic --workload union8 --dist genomic
run union8_clustered --workload union8 --dist clustered
run union32 --workload union32
run union32_disjoint --workload union32 --dist disjoint
run intersect8 --workload intersect8
run sort --workload sort
run table --workload table
run table32 --workload table --nt-lists 32 --nt 20000000
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6final2/*.json")):
    try:
        b=json.load(open(f)); r=b.get("roofline",{})
        print(f.split("/")[-1], round(b["ms_per_step"],2), round(r.get("frac",0),3), r.get("traffic"), (r.get("traffic_source") or {}).get("note","")[:30], b.get("self_check"), b.get("verified"))
    except Exception as e:
        print(f, "FAILED", e)
PY
