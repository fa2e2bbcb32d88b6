#!/bin/bash
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03final
rm -rf $O && mkdir -p $O
TRC=skipped
if [ "$1" != "--no-tests" ]; then
  timeout -k 10 1500 python3 -m pytest tests -q -m gpu > $O/tests.log 2>&1 && TRC=0 || TRC=$?
  grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -15 || true
  [ $TRC -eq 0 ] || grep -n "^E " $O/tests.log | head -30
  python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo "smoke ok" || tail -5 $O/smoke.log
fi
T0=$(date +%s); python3 bench.py > $O/bench.json 2> $O/bench.err; echo "default bench.py wall time $(( $(date +%s) - T0 )) s"
python3 -c "import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_step'],d['extra_configs']['c3_bf16']['value'],d['extra_configs']['c3_bf16']['ms_per_step'],d['extra_configs']['c3_bf16']['roofline_bf16'])"
python3 bench.py --workload c3 --tower-precision bf16 --student-precision bf16 --no-cpu-baseline > $O/bench_c3_bf16.json 2> $O/bench_c3.err
python3 bench.py --workload c5 --batch 512 --steps 4 --warmup 1 --student-precision bf16 --no-cpu-baseline --no-extra-legs > $O/bench_c5_b512.json 2> $O/bench_c5.err
C3="--eager --no-cpu-baseline --no-extra-legs --workload c3 --tower-precision bf16 --student-precision bf16"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3b -o s -- python3 bench.py $C3 --steps 4 --warmup 1 > $O/c3b.log 2>&1
rm -f $O/c3b/*trace.csv
python3 tools/summarize_profile.py $O/c3b /nonexistent /nonexistent $O/c3b_sum > /dev/null
python3 -c "
import json
for n in ('bench_c3_bf16','bench_c5_b512'):
    d=json.load(open('$O/'+n+'.json')); print(n,d['value'],d['ms_per_step'],d['roofline_bf16']['frac'],d['roofline_step']['frac'])"
echo "tests rc=$TRC"
