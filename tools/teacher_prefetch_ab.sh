#!/bin/bash
# Same-box A/B of the cross-step meta-teacher pipelining (CLIPImageDistillation.prefetch_teacher): config c3 as quoted and
# one GPU's share of c5, DCLIP_TEACHER_PREFETCH=0 / 1 alternating, no per-launch events.  Run on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/teacher_prefetch_ab.sh'      -> gpurun_out/teacher_prefetch_ab.log
set -o pipefail
mkdir -p gpurun_out
C3="python bench.py --workload c3 --student-precision bf16 --tower-precision bf16 --no-cpu-baseline --no-extra-legs --no-gemm-events --steps 30 --warmup 8"
C5="python bench.py --workload c5 --batch 512 --student-precision bf16 --tower-precision bf16 --teacher-model ViT-L/14 --no-cpu-baseline --no-extra-legs --no-gemm-events --steps 6 --warmup 2"
for rep in 1 2; do
  for w in c3 c5; do
    for on in 0 1; do
      cmd="$C3"; [ $w = c5 ] && cmd="$C5"
      DCLIP_TEACHER_PREFETCH=$on $cmd > gpurun_out/tp_${w}_${on}_$rep.json 2> gpurun_out/tp_${w}_${on}_$rep.err || echo "$w prefetch=$on rc=$?"
    done
  done
done
python - <<'PY' | tee gpurun_out/teacher_prefetch_ab.log
import json, glob
for f in sorted(glob.glob('gpurun_out/tp_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], 'ms', d['value'], 'img/s loss', d['config'].get('loss'))
PY
