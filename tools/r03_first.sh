#!/bin/bash
# Round 3, first GPU call: the new parity tests, counters over the c3 bf16 workload, the 4-rank gloo rehearsal (once).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a
rm -rf $O && mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_entrypoints_dp_gpu.py tests/test_dist_gpu.py tests/test_bf16_train_gpu.py tests/test_configs_gpu.py "tests/test_ops_gpu.py::test_contrastive_lse_and_grad" tests/test_bf16_gpu.py::test_bf16_weight_cache_follows_the_fused_optimizer -q -s -m gpu > $O/tests.log 2>&1 && TRC=0 || TRC=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -15 || true
C3="--workload c3 --student-precision bf16 --tower-precision bf16 --eager --no-cpu-baseline --no-extra-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3stats -o s -- python3 bench.py $C3 --steps 4 --warmup 1 > $O/c3stats.log 2>&1
echo "c3 stats done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c3sq -o q -- python3 bench.py $C3 --steps 2 --warmup 1 > $O/c3sq.log 2>&1
echo "c3 sq done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c3fetch -o f -- python3 bench.py $C3 --steps 2 --warmup 1 > $O/c3fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c3write -o w -- python3 bench.py $C3 --steps 2 --warmup 1 > $O/c3write.log 2>&1
echo "c3 fetch/write done"
python3 tools/summarize_pmc.py $O/c3sq $O/pmc_c3.json > $O/pmc_c3.txt
python3 bench.py --workload c3 --student-precision bf16 --tower-precision bf16 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
tail -c 600 $O/bench_c3.json
# the 4-rank rehearsal, ONCE: four ranks share the card over gloo, small batch, per-rank step log on stderr
DCLIP_DIST_BACKEND=gloo timeout -k 10 420 python3 bench.py --gpus 4 --batch 16 --steps 6 --warmup 2 > $O/gloo4.json 2> $O/gloo4.err || { echo "gloo4 rc=$?"; tail -30 $O/gloo4.err; exit 1; }
cat $O/gloo4.json | cut -c1-1200
grep -c "timed step" $O/gloo4.err
echo "tests rc=$TRC"
# keep the big counter CSVs out of the merge-back budget
find $O -name "*counter_collection.csv" -size +20M -delete
