#!/usr/bin/env python3
"""Times every tile / split-K candidate on the step's GEMM shapes (run on the GPU box)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops, _lib

dev = torch.device("cuda:0")
def shapes(B=256):
    M, Mt = B * 50, B * 77
    return [("v.qkv_fwd", 3, M, 2304, 768), ("v.out_fwd", 3, M, 768, 768), ("v.fc1_fwd", 3, M, 3072, 768),
            ("v.fc2_fwd", 3, M, 768, 3072), ("v.fc2_dgrad", 1, M, 3072, 768), ("v.fc1_dgrad", 1, M, 768, 3072),
            ("v.out_dgrad", 1, M, 768, 768), ("v.qkv_dgrad", 1, M, 768, 2304), ("v.fc_wgrad", 0, 3072, 768, M),
            ("v.qkv_wgrad", 0, 2304, 768, M), ("v.out_wgrad", 0, 768, 768, M), ("v.patch_fwd", 3, B * 49, 768, 3072),
            ("v.patch_wgrad", 0, 768, 3072, B * 49),
            ("t.qkv", 3, Mt, 1536, 512), ("t.out", 3, Mt, 512, 512), ("t.fc1", 3, Mt, 2048, 512), ("t.fc2", 3, Mt, 512, 2048)]

def timeit(a, b, layout, out, split, iters=10):
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
    tile = os.environ["DCLIP_GEMM_TILE"]
    res = {}
    for name, layout, m, n, k in shapes():
        a = torch.randn((m, k) if layout & 1 else (k, m), device=dev)
        b = torch.randn((n, k) if layout & 2 else (k, n), device=dev)
        out = torch.empty(m, n, device=dev)
        splits = [1] if k <= 4096 else [2, 4, 8, 16]
        for s in splits:
            res[f"{name}|{s}"] = timeit(a, b, layout, out, s)
    print("RESULT " + json.dumps(res))
else:
    allres = {}
    for tile in ("128x128", "128x64", "64x128", "64x64"):
        env = dict(os.environ, DCLIP_GEMM_TILE=tile)
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        line = [l for l in o.splitlines() if l.startswith("RESULT ")][0]
        for k, v in json.loads(line[7:]).items():
            allres.setdefault(k.split("|")[0], {})[f"{tile}/s{k.split('|')[1]}"] = v
    fl = {n: 2.0 * m * nn * k for n, _, m, nn, k in shapes()}
    for name, d in allres.items():
        best = min(d, key=d.get)
        print(f"{name:14s} best {best:12s} {d[best]:8.1f} us {fl[name]/d[best]/1e6:6.1f} TF | " +
              " ".join(f"{k}:{v:.0f}" for k, v in sorted(d.items(), key=lambda kv: kv[1])[:6]))
