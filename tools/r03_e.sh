#!/bin/bash
# Round 3: persistent ping-pong GEMM v2 (staged epilogue through the spare LDS window): parity test, A/B, c3 line.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03e
rm -rf $O && mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_bf16_gpu.py -q -s -m gpu > $O/tests.log 2>&1 && TRC=0 || TRC=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -15 || true
[ $TRC -eq 0 ] || { grep -n "^E " $O/tests.log | head -20; echo "tests rc=$TRC"; exit 1; }
timeout -k 10 300 python3 tools/bf16_persist_ab.py > $O/persist_ab.log 2>&1 || true
grep -v amdgpu.ids $O/persist_ab.log
AB_PERSIST_MIN=512 true
for P in 0 1; do
  DCLIP_BF16_PERSIST=$P python3 bench.py --workload c3 --student-precision bf16 --tower-precision bf16 --no-cpu-baseline > $O/bench_c3_p$P.json 2> $O/bench_c3_p$P.err
  python3 -c "import json;d=json.load(open('$O/bench_c3_p$P.json'));print('c3 persist=$P',d['value'],d['ms_per_step'],d['roofline_bf16']['frac'],d['roofline_bf16']['gemm_ms_per_step'],d['config']['loss'])"
done
echo "tests rc=$TRC"
