#!/bin/bash
# Shader clock, socket power and junction temperature (rocm-smi, every 2 s) while a bench command runs: what clock does the
# MFMA roofline's 2.4 GHz actually get under this load?  usage (GPU box): bash tools/clock_probe.sh <tag> <bench.py args...>
TAG=$1; shift
mkdir -p gpurun_out/clock
( for i in $(seq 1 60); do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|junction" | head -4; sleep 2; done ) > gpurun_out/clock/$TAG.smi 2>&1 &
SMI=$!
DCLIP_BENCH_STEP_LOG=1 python bench.py "$@" > gpurun_out/clock/$TAG.json 2> gpurun_out/clock/$TAG.err
kill $SMI 2>/dev/null
python3 - "$TAG" <<'PY'
import re, sys, json
tag = sys.argv[1]
rec, cur = [], {}
for l in open(f"gpurun_out/clock/{tag}.smi"):
    l = l.strip()
    if re.match(r"^\d{10}\.\d+$", l):
        if cur: rec.append(cur)
        cur = {"t": float(l)}
    for key, pat in (("sclk", r"sclk clock level: \S+ \((\d+)Mhz\)"), ("w", r"Power \(W\): ([\d.]+)"), ("tj", r"junction\) \(C\): ([\d.]+)")):
        m = re.search(pat, l)
        if m: cur[key] = float(m.group(1))
if cur: rec.append(cur)
busy = [r for r in rec if r.get("w", 0) > 600]
d = json.loads(open(f"gpurun_out/clock/{tag}.json").read().strip().splitlines()[-1])
print(tag, "ms/step", d["ms_per_step"], "value", d["value"], "| samples under load:", len(busy),
      "sclk MHz", sorted(int(r["sclk"]) for r in busy), "power W", sorted(int(r["w"]) for r in busy), "Tj", sorted(int(r["tj"]) for r in busy))
PY
