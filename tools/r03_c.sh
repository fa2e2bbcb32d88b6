#!/bin/bash
# Round 3, third GPU call: the bf16-path tests again, student-shape GEMM A/B (kernel choice by environment), 4-rank rehearsal
# over gloo with host-staged buckets.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c
rm -rf $O && mkdir -p $O
timeout -k 10 1200 python3 -m pytest tests/test_bf16_train_gpu.py tests/test_configs_gpu.py tests/test_dist_gpu.py tests/test_entrypoints_dp_gpu.py tests/test_bf16_gpu.py tests/test_bench_gpu.py -q -s -m gpu > $O/tests.log 2>&1 && TRC=0 || TRC=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -15 || true
python3 tools/bf16_student_shapes.py > $O/shapes_default.log 2>&1
DCLIP_BF16_BIG_MIN=151 python3 tools/bf16_student_shapes.py > $O/shapes_mid_reg128.log 2>&1
DCLIP_BF16_BIG_MIN=151 DCLIP_BF16_MID_DMA=1 python3 tools/bf16_student_shapes.py > $O/shapes_mid_dma128.log 2>&1
DCLIP_BF16_BIG_MIN=100000 DCLIP_BF16_MID_DMA=1 python3 tools/bf16_student_shapes.py > $O/shapes_all_dma128.log 2>&1
tail -n 10 $O/shapes_default.log $O/shapes_mid_reg128.log $O/shapes_mid_dma128.log $O/shapes_all_dma128.log
DCLIP_SYNC_TRACE=1 DCLIP_DIST_BACKEND=gloo timeout -k 10 420 python3 bench.py --gpus 4 --batch 16 --steps 4 --warmup 2 > $O/gloo4.json 2> $O/gloo4.err && GRC=0 || GRC=$?
echo "gloo4 rc=$GRC"; grep -E "^\[rank 0" $O/gloo4.err | tail -12 || true; cut -c1-1500 $O/gloo4.json
echo "tests rc=$TRC"
