#!/bin/bash
# AdamW non-temporal accesses A/B on the default line (same box, alternating)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03h
rm -rf $O && mkdir -p $O
B="--no-cpu-baseline"
for i in 1 2; do
  python3 bench.py $B > $O/nt_$i.json 2> $O/nt_$i.err
  python3 -c "import json;d=json.load(open('$O/nt_$i.json'));print('nt   ',d['value'],d['ms_per_step'],d['fwd_bwd_ms_per_step'],d['optimizer_ms_per_step'])"
  DCLIP_LIB_PATH=$PWD/tools/ab/libdclip_hip_adam_plain.so python3 bench.py $B > $O/plain_$i.json 2> $O/plain_$i.err
  python3 -c "import json;d=json.load(open('$O/plain_$i.json'));print('plain',d['value'],d['ms_per_step'],d['fwd_bwd_ms_per_step'],d['optimizer_ms_per_step'])"
done
python3 -m pytest tests/test_ops_gpu.py -q -m gpu -k "adam or Adam or optim" 2>&1 | tail -2
