#!/usr/bin/env python3
"""Achieved HBM rate of the bandwidth-bound kernels of the c2 step: algorithmic bytes per launch
(from the workload's shapes, stated below) / average launch duration from the rocprofv3 kernel-stats summary.
   usage: hbm_report.py profiles/r01_kernel_stats.csv profiles/r01_hbm_kernels.json"""
import csv, json, sys

stats, out = sys.argv[1:3]
B, S, D, H, I = 256, 50, 768, 12, 3072            # ViT-B/32 vision tower at bs 256
T, Dt = 77, 512                                   # text tower
rows_v, rows_t = B * S, B * T
F = 4
trainable = 87_456_000 + 393_216 + 1              # vision tower + visual_projection + logit_scale (north_star regime)
KERNELS = {
    "ln_fwd_kernel<3, true>": ("LayerNorm fwd, vision rows (x read, y written, 8 B/elem)", 2 * rows_v * D * F),
    "ln_fwd_kernel<2, true>": ("LayerNorm fwd, text rows", 2 * rows_t * Dt * F),
    "ln_bwd_kernel<3, true>": ("LayerNorm bwd + residual-gradient add (dy, x, dres read; dx written, 16 B/elem)", 4 * rows_v * D * F),
    "attn_fwd_rows_kernel<4, false, false>": ("vision attention fwd (q,k,v read once, out written)", 4 * rows_v * D * F),
    "attn_fwd_rows_kernel<5, true, false>": ("text causal attention fwd", 4 * rows_t * Dt * F),
    "attn_bwd_lean_kernel<false, false>": ("vision attention bwd (q,k,v,o,do read; dq,dk,dv written)", 8 * rows_v * D * F),
    "mt_adamw_kernel": ("multi-tensor AdamW (p,g,m,v read; p,m,v written, 28 B/param)", 28 * trainable),
    "mt_sumsq_kernel": ("global grad-norm partial sums (g read)", 4 * trainable),
    "im2col_vec_kernel<false>": ("patch gather (pixels read, columns written)", 2 * B * 3 * 224 * 224 * F),
}
PEAK = 8000.0  # GB/s, MI355X HBM3E (MI355X_MICROARCH.md)
rep = []
for r in csv.DictReader(open(stats)):
    for key, (what, nbytes) in KERNELS.items():
        if key in r["kernel"]:
            us = float(r["avg_us"])
            gbs = nbytes / us / 1e3
            rep.append({"kernel": key, "what": what, "launches": int(r["calls"]), "avg_us": us,
                        "algorithmic_bytes_per_launch": nbytes, "achieved_GBps": round(gbs, 1),
                        "frac_of_8TBps": round(gbs / PEAK, 3), "ms_total": float(r["total_ms"])})
json.dump({"source": stats, "peak_GBps": PEAK, "kernels": rep}, open(out, "w"), indent=1)
for k in rep:
    print(f"| `{k['kernel']}` | {k['what']} | {k['algorithmic_bytes_per_launch'] / 1e6:.0f} MB | {k['avg_us']:.1f} µs | "
          f"{k['achieved_GBps'] / 1e3:.2f} TB/s ({100 * k['frac_of_8TBps']:.0f} %) |")
