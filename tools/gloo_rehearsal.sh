#!/bin/bash
# N data-parallel ranks on ONE GPU over gloo (device buckets staged through pinned host mirrors): the whole bench.py with
# per-rank step log and per-bucket wait trace on stderr.  usage (on the GPU box, from the repo root): tools/gloo_rehearsal.sh [N=4]
N=${1:-4}
DCLIP_SYNC_TRACE=1 DCLIP_DIST_BACKEND=gloo timeout -k 10 420 python3 bench.py --gpus $N --batch 16 --steps 4 --warmup 2
