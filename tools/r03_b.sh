#!/bin/bash
# Round 3, second GPU call: full GPU suite (attention / LayerNorm kernels changed), c3 + default bench lines, c3 kernel stats,
# one bisecting 4-rank gloo rehearsal (no GEMM events, per-bucket wait trace).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03b
rm -rf $O && mkdir -p $O
timeout -k 10 1500 python3 -m pytest tests -q -s -m gpu > $O/tests.log 2>&1 && TRC=0 || TRC=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -15 || true
python3 bench.py --workload c3 --student-precision bf16 --tower-precision bf16 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python3 -c "import json;d=json.load(open('$O/bench_c3.json'));print('c3',d['value'],d['ms_per_step'],d['roofline_bf16']['frac'],d['roofline_bf16']['gemm_ms_per_step'])"
C3="--workload c3 --student-precision bf16 --tower-precision bf16 --eager --no-cpu-baseline --no-extra-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3stats -o s -- python3 bench.py $C3 --steps 4 --warmup 1 > $O/c3stats.log 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 -c "import json;d=json.load(open('$O/bench.json'));print('c2',d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('extra_configs'))"
DCLIP_SYNC_TRACE=1 DCLIP_DIST_BACKEND=gloo timeout -k 10 420 python3 bench.py --gpus 4 --batch 16 --steps 3 --warmup 2 --no-gemm-events > $O/gloo4.json 2> $O/gloo4.err && GRC=0 || GRC=$?
echo "gloo4 rc=$GRC"; grep -E "^\[rank 0|GradSync rank 0" $O/gloo4.err | tail -30 || true
echo "tests rc=$TRC"
rm -f $O/c3stats/*trace.csv
