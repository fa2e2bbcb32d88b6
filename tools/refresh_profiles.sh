#!/bin/bash
# Run on the GPU box (from the repo root): kernel-trace stats, an SQ counter pass, separate FETCH_SIZE / WRITE_SIZE passes
# (PMC passes never combined with other trace domains), kernel stats of the c3 / c5 workloads, and the default bench line.
# Outputs under gpurun_out/refresh/; tools/summarize_profile.py + tools/summarize_pmc.py condense them into profiles/.
# Profiled runs launch every step eagerly (--eager): the default graph replay is timed by the last, unprofiled run.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
B="--eager --no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py $B --steps 6 --warmup 1 > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o q -- python3 bench.py $B --steps 2 --warmup 1 > $O/sq.log 2>&1
echo "sq done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py $B --steps 2 --warmup 1 > $O/write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3b -o s -- python3 bench.py $B --workload c3 --tower-precision bf16 --student-precision bf16 --steps 4 --warmup 1 > $O/c3b.log 2>&1
echo "c3 bf16 student done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5b -o s -- python3 bench.py $B --workload c5 --batch 64 --tower-precision bf16 --student-precision bf16 --steps 4 --warmup 1 > $O/c5b.log 2>&1
echo "c5 bf16 student done"
python3 bench.py --workload c3 --tower-precision bf16 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --workload c3 --tower-precision bf16 --student-precision bf16 --no-cpu-baseline > $O/bench_c3_bf16_student.json 2> $O/bench_c3b.err
python3 bench.py --workload c5 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5_b64.json 2> $O/bench_c5.err
python3 bench.py --workload c5 --batch 64 --steps 6 --warmup 2 --student-precision bf16 --no-cpu-baseline > $O/bench_c5_b64_bf16_student.json 2> $O/bench_c5b.err
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
