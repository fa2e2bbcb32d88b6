#!/bin/bash
# Run on the GPU box (from the repo root): kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes + the default
# bench line.  Outputs under gpurun_out/refresh/; tools/summarize_profile.py condenses them into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
echo "write done"
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
