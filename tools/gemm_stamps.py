#!/usr/bin/env python3
"""Where does a GEMM launch spend its time?  Diagnostic (run on the GPU box; `make stamps` first).

Loads tools/ab/libdclip_hip_stamps.so — the product kernels plus in-kernel stamps in gemm_f32_kernel — launches each of
the step's GEMM shapes, and reports per shape
  * the launch span (first workgroup in -> last workgroup out, 100 MHz real-time clock) against the ideal MFMA time,
  * per-workgroup phase lengths in shader cycles: prologue (entry -> first barrier), K loop, epilogue,
  * per-CU occupancy: workgroups per CU, the time each CU sits without any workgroup inside the span (tail + head),
  * the K loop's cycles per K-tile against the 4 * MT * NT * 16 * 64 / waves-per-SIMD MFMA floor.
Stamp values go to a buffer of their own; the timed product path never runs this library."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libdclip_hip_stamps.so"))
ap.add_argument("--tile", default=None)
ap.add_argument("--shapes", default="step")
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
if args.tile:
    os.environ["DCLIP_GEMM_TILE"] = args.tile
import numpy as np
import torch
from dclip_amd import ops, _lib

_lib.LIB_PATH = os.path.abspath(args.lib)
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.dclip_debug_set_gemm_stamps.argtypes = [C.c_void_p]
raw.dclip_debug_gemm_plan.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)]
dev = torch.device("cuda:0")
M = 12800
SHAPES = {
    "step": [("qkv_fwd NT", 3, M, 2304, 768), ("out_fwd NT", 3, M, 768, 768), ("fc1_fwd NT", 3, M, 3072, 768),
             ("fc2_fwd NT", 3, M, 768, 3072), ("fc2_dgrad NN", 1, M, 3072, 768), ("fc1_dgrad NN", 1, M, 768, 3072),
             ("out_dgrad NN", 1, M, 768, 768), ("qkv_dgrad NN", 1, M, 768, 2304), ("fc1_wgrad TN", 0, 3072, 768, M),
             ("qkv_wgrad TN", 0, 2304, 768, M), ("out_wgrad TN", 0, 768, 768, M), ("txt_qkv NT", 3, 19712, 1536, 512),
             ("txt_fc1 NT", 3, 19712, 2048, 512), ("txt_fc2 NT", 3, 19712, 512, 2048)],
    "big": [("8192^3 NT", 3, 8192, 8192, 8192)],
}
MAXWG = 1 << 16
stamps = torch.zeros((MAXWG, 16), dtype=torch.int64, device=dev)
for name, layout, m, n, k in SHAPES[args.shapes]:
    a = torch.randn((m, k) if layout & 1 else (k, m), device=dev)
    b = torch.randn((n, k) if layout & 2 else (k, n), device=dev)
    out = torch.empty(m, n, device=dev)
    plan = (C.c_int * 4)()
    raw.dclip_debug_gemm_plan(m, n, k, layout, 0, plan)
    bm, bn, splits, kps = list(plan)
    raw.dclip_debug_set_gemm_stamps(None)
    for _ in range(3):
        ops.gemm(a, b, layout, out=out)
    torch.cuda.synchronize()
    stamps.zero_()
    raw.dclip_debug_set_gemm_stamps(stamps.data_ptr())
    # steady state: back-to-back launches (clock and caches as inside the step); every launch stamps the same slots,
    # the LAST one's values remain.  The event pair brackets that last launch only.
    for _ in range(args.reps - 1):
        ops.gemm(a, b, layout, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(a, b, layout, out=out)
    e1.record()
    torch.cuda.synchronize()
    raw.dclip_debug_set_gemm_stamps(None)
    nwg = -(-m // bm) * -(-n // bn) * splits
    s = stamps[:nwg].cpu().numpy().astype(np.int64)
    t0, t1, t2, t3, r0, r1, hw, tile = [s[:, i] for i in range(8)]
    pro, loop, epi = t1 - t0, t2 - t1, t3 - t2
    span_us = (r1.max() - r0.min()) / 100.0
    flops = 2.0 * m * n * k
    ideal_us = flops / 157.3e12 * 1e6
    nk = -(-kps // 32)
    cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF)
    cus = np.unique(cu_key)
    per_cu = np.array([(cu_key == c).sum() for c in cus])
    # per-CU: time with no workgroup resident inside the span (real-time clock, 10 ns ticks)
    idle = []
    for c in cus:
        sel = cu_key == c
        iv = sorted(zip(r0[sel], r1[sel]))
        cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for a_, b_ in iv[1:]:
            if a_ > cur_e:
                cov += cur_e - cur_s
                cur_s, cur_e = a_, b_
            else:
                cur_e = max(cur_e, b_)
        cov += cur_e - cur_s
        idle.append((r1.max() - r0.min()) - cov)
    idle = np.array(idle) / 100.0
    clk_ghz = np.median((t3 - t0) / np.maximum(r1 - r0, 1)) / 10.0
    waves_per_simd = {16384: 2, 8192: 3, 4096: 4}.get(bm * bn, 2)
    floor_cyc_per_ktile = (bm // 32) * (bn // 32) * 16 * 64 / 4           # MFMA cycles per K-tile on one CU's 4 SIMDs
    print(f"{name:16s} M={m:6d} N={n:5d} K={k:6d} tile {bm}x{bn} splits {splits} wgs {nwg} ({nwg / 256:.2f}/CU) "
          f"event {e0.elapsed_time(e1) * 1e3:7.1f} us span {span_us:7.1f} us ideal {ideal_us:7.1f} us "
          f"-> {ideal_us / span_us * 100:5.1f}% | clock {clk_ghz:.2f} GHz", flush=True)
    q = lambda x: f"{np.percentile(x, 10):8.0f}/{np.median(x):8.0f}/{np.percentile(x, 90):8.0f}"
    print(f"    cycles p10/median/p90: prologue {q(pro)}  loop {q(loop)} ({np.median(loop) / nk:7.0f}/K-tile, MFMA floor "
          f"{floor_cyc_per_ktile:.0f} x co-resident WGs)  epilogue {q(epi)}")
    s8, s9, s10, s11, s12 = [s[:, i] for i in range(8, 13)]
    ok = s11 > 0            # interior tiles only carry the epilogue sub-stamps
    print(f"    prologue split (median cycles): setup {np.median(s8 - t0):6.0f} | first tile issued + acc zeroed {np.median(s9 - s8):6.0f} | "
          f"landed / LDS store {np.median(s10 - s9):6.0f} | barrier {np.median(t1 - s10):6.0f}")
    if ok.any():
        print(f"    epilogue split (median cycles): acc->LDS (incl. MFMA drain) {np.median((s11 - t2)[ok]):6.0f} | barrier "
              f"{np.median((s12 - s11)[ok]):6.0f} | side loads + row stores {np.median((t3 - s12)[ok]):6.0f}")
    vmw, brw = (tile >> 20) & 0x3FFFFF, (tile >> 42) & 0x3FFFFF
    print(f"    wave 0, per K-tile: vmcnt(0) wait {np.median(vmw) / nk:6.0f} cyc  barrier wait {np.median(brw) / nk:6.0f} cyc "
          f"(p90 {np.percentile(vmw, 90) / nk:6.0f} / {np.percentile(brw, 90) / nk:6.0f})")
    print(f"    CUs seen {len(cus)}  WGs/CU min {per_cu.min()} max {per_cu.max()}  CU idle inside span: "
          f"median {np.median(idle):6.1f} us  p90 {np.percentile(idle, 90):6.1f} us  max {idle.max():6.1f} us  "
          f"mean {idle.mean():6.1f} us ({idle.mean() / span_us * 100:4.1f}% of span)")
    # per CU: how many of its resident workgroups are inside their K loop at a time?  (K-loop interval on the real-time
    # clock from the cycle stamps' share of the workgroup's life)  0 in the K loop = matrix pipes idle on that CU.
    life = np.maximum(t3 - t0, 1).astype(np.float64)
    k_beg = r0 + (t1 - t0) / life * (r1 - r0)
    k_end = r0 + (t2 - t0) / life * (r1 - r0)
    lo, hi = r0.min(), r1.max()
    grid = np.linspace(lo + 0.1 * (hi - lo), lo + 0.9 * (hi - lo), 400)
    hist = np.zeros(8)
    for c in cus:
        sel = cu_key == c
        kb, ke = k_beg[sel], k_end[sel]
        n = ((kb[None, :] <= grid[:, None]) & (grid[:, None] < ke[None, :])).sum(axis=1)
        hist += np.bincount(np.minimum(n, 7), minlength=8)
    hist /= hist.sum()
    print("    workgroups of a CU inside their K loop (middle 80 % of the span): " +
          "  ".join(f"{k}: {hist[k] * 100:4.1f} %" for k in range(5)) + f"   mean {sum(k * hist[k] for k in range(8)):.2f}")
    # concurrency-weighted efficiency: MFMA floor of all work on a CU / time the CU had at least one WG
    del a, b, out
