#!/bin/bash
# Run on the GPU box (from the repo root): kernel-trace stats, an SQ counter pass, separate FETCH_SIZE / WRITE_SIZE passes
# (PMC passes never combined with other trace domains) for the default workload (c2) and for c3 in bf16, kernel stats of c5,
# and the bench lines.  Outputs under gpurun_out/refresh/; tools/summarize_profile.py, summarize_pmc.py, summarize_traffic.py
# and hbm_report.py condense them into profiles/.  Profiled runs launch every step eagerly (--eager): the default graph
# replay is timed by the last, unprofiled run.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
B="--eager --no-cpu-baseline --no-extra-legs"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py $B --steps 6 --warmup 1 > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc $SQ --output-format csv -d $O/sq -o q -- python3 bench.py $B --steps 2 --warmup 1 > $O/sq.log 2>&1
echo "sq done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py $B --steps 2 --warmup 1 > $O/write.log 2>&1
echo "fetch / write done"
C3="$B --workload c3 --tower-precision bf16 --student-precision bf16"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3b -o s -- python3 bench.py $C3 --steps 4 --warmup 1 > $O/c3b.log 2>&1
rocprofv3 --pmc $SQ --output-format csv -d $O/c3sq -o q -- python3 bench.py $C3 --steps 2 --warmup 1 > $O/c3sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c3fetch -o f -- python3 bench.py $C3 --steps 2 --warmup 1 > $O/c3fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c3write -o w -- python3 bench.py $C3 --steps 2 --warmup 1 > $O/c3write.log 2>&1
echo "c3 bf16 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5b -o s -- python3 bench.py $B --workload c5 --batch 64 --tower-precision bf16 --student-precision bf16 --steps 4 --warmup 1 > $O/c5b.log 2>&1
echo "c5 bf16 done"
rm -f $O/*/*trace.csv
python3 tools/summarize_profile.py $O/stats $O/fetch $O/write $O/c2 > /dev/null
python3 tools/summarize_pmc.py $O/sq $O/c2_pmc_step.json > /dev/null
python3 tools/summarize_pmc.py $O/c3sq $O/c3_pmc.json > /dev/null
python3 tools/summarize_traffic.py $O/c3fetch $O/c3write $O/c3_hbm.json > /dev/null
python3 tools/summarize_profile.py $O/c3b /nonexistent /nonexistent $O/c3b_sum > /dev/null
python3 tools/summarize_profile.py $O/c5b /nonexistent /nonexistent $O/c5b_sum > /dev/null
find $O -name "*counter_collection.csv" -delete
python3 bench.py --workload c3 --tower-precision bf16 --student-precision bf16 --no-cpu-baseline > $O/bench_c3_bf16.json 2> $O/bench_c3b.err
python3 bench.py --workload c5 --batch 512 --steps 4 --warmup 1 --student-precision bf16 --no-cpu-baseline --no-extra-legs > $O/bench_c5_b512.json 2> $O/bench_c5.err
python3 bench.py --model ViT-B/16 --batch 128 --no-cpu-baseline > $O/bench_c4_pergpu.json 2> $O/bench_c4.err
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json
