#!/bin/bash
# Round 3: fp32 GEMM rasterisation A/B (DCLIP_GEMM_GROUP_M) on the default line, L2 hit rate (TCC_HIT / TCC_MISS) of the step,
# kernel stats of the default workload.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline --no-extra-legs"
for G in 8 4 16 2; do
  DCLIP_GEMM_GROUP_M=$G python3 bench.py $B > $O/bench_g$G.json 2> $O/bench_g$G.err
  python3 -c "import json;d=json.load(open('$O/bench_g$G.json'));print('GROUP_M=$G',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['gemm_ms_per_step'])"
done
E="--eager --no-cpu-baseline --no-extra-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py $E --steps 6 --warmup 1 > $O/stats.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -o t -- python3 bench.py $E --steps 2 --warmup 1 > $O/tcc.log 2>&1
echo "tcc done"
for G in 4 16; do
  DCLIP_GEMM_GROUP_M=$G timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_g$G -o f -- python3 bench.py $E --steps 2 --warmup 1 > $O/fetch_g$G.log 2>&1
done
echo "fetch done"
rm -f $O/stats/*trace.csv
python3 - <<'PY'
import csv, glob, collections, re
for d in ("gpurun_out/r03f/tcc", "gpurun_out/r03f/fetch_g4", "gpurun_out/r03f/fetch_g16"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        name = re.sub(r"<.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()[:40]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        k = (r.get("Dispatch_Id"), name)
        if k not in seen: seen.add(k); n[name] += 1
    for name, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:4]:
        print(d.split("/")[-1], name, n[name], {k: round(v / n[name], 1) for k, v in c.items()})
PY
