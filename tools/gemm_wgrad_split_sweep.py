#!/usr/bin/env python3
"""Every tile x split-K count (2..24, not only powers of two) on the step's weight-gradient shapes; prints the best few per
shape beside what the planner picks.  Run on the GPU box."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops

dev = torch.device("cuda:0")
M = 12800
SHAPES = [("fc_wgrad", 0, 3072, 768, M), ("fc_wgrad'", 0, 768, 3072, M), ("qkv_wgrad", 0, 2304, 768, M), ("out_wgrad", 0, 768, 768, M)]


def timeit(a, b, layout, out, split, iters=12):
    for _ in range(2):
        ops.gemm(a, b, layout, out=out, split_k=split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(a, b, layout, out=out, split_k=split)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


if len(sys.argv) > 1 and sys.argv[1] == "child":
    res = {}
    for name, layout, m, n, k in SHAPES:
        a = torch.randn(k, m, device=dev)
        b = torch.randn(k, n, device=dev)
        out = torch.empty(m, n, device=dev)
        if os.environ.get("DCLIP_GEMM_TILE"):
            for s in range(2, 25):
                res[f"{name}|{s}"] = timeit(a, b, layout, out, s)
        else:
            res[f"{name}|planner"] = timeit(a, b, layout, out, 0)
    print("RESULT " + json.dumps(res))
else:
    allres = {}
    for tile in (None, "128x128", "128x64", "64x128", "64x64"):
        env = dict(os.environ)
        env.pop("DCLIP_GEMM_TILE", None)
        if tile:
            env["DCLIP_GEMM_TILE"] = tile
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        line = [l for l in o.splitlines() if l.startswith("RESULT ")][0]
        for k, v in json.loads(line[7:]).items():
            allres.setdefault(k.split("|")[0], {})[f"{tile or 'planner'}/s{k.split('|')[1]}"] = v
    fl = {n: 2.0 * m * nn * k for n, _, m, nn, k in SHAPES}
    for name, d in allres.items():
        pl = d.pop("planner/splanner")
        order = sorted(d.items(), key=lambda kv: kv[1])
        print(f"{name:11s} planner {pl:7.1f} us {fl[name] / pl / 1e6:6.1f} TF/s | best " +
              " ".join(f"{k}:{v:.1f}" for k, v in order[:8]), flush=True)
