#!/bin/bash
# Does a long MFMA-bound run slow down, and do the clocks say why?  c5's per-GPU step (512 pairs, ViT-L/14 teacher) for 30 steps
# with the per-step log on, rocm-smi clocks / power / temperature sampled every 2 s beside it.  On the GPU box:
#   gpurun --timeout 600 -- 'bash tools/sustained_load_probe.sh'     -> gpurun_out/sustained/{bench.err,smi.log}
mkdir -p gpurun_out/sustained
( for i in $(seq 1 40); do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | head -8; sleep 2; done ) > gpurun_out/sustained/smi.log 2>&1 &
SMI=$!
DCLIP_BENCH_STEP_LOG=1 python bench.py --workload c5 --batch 512 --student-precision bf16 --tower-precision bf16 --teacher-model ViT-L/14 \
  --no-cpu-baseline --no-extra-legs --no-gemm-events --steps 30 --warmup 2 > gpurun_out/sustained/bench.json 2> gpurun_out/sustained/bench.err
kill $SMI 2>/dev/null
grep -E "timed step|warm-up step" gpurun_out/sustained/bench.err | head -40
tail -c 400 gpurun_out/sustained/bench.json
