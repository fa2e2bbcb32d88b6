#!/usr/bin/env python3
"""DCLIP distillation-step benchmark (BASELINE.json metric: distill-step images/sec at ViT-B/32 bs=256).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch of 256 image-caption pairs per GPU
(BASELINE config c2): student ViT-B/32 forward + backward (all vision parameters trainable), frozen text tower
forward (shared by student and teacher: north_star regime), cosine distillation losses against a
given teacher image embedding, symmetric InfoNCE over the (all-gathered) similarity matrix, gradient all-reduce
for N > 1, and the AdamW update.  Inputs are resident in HBM before the timed region.  fp32 throughout.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# the host driver only supports dmabuf IPC: RCCL / device-tensor sharing across ranks needs this (already exported on the
# pool's boxes; set here as well so a bare launch behaves the same)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dclip_amd import config as dcfg, synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level table)


def vision_fwd_flops(v) -> float:
    S, D, L = v.seq_len, v.hidden_size, v.num_hidden_layers
    per_layer = 2.0 * S * D * (3 * D) + 2.0 * S * D * D + 2.0 * 2.0 * S * D * v.intermediate_size + 4.0 * S * S * D
    return L * per_layer + 2.0 * (S - 1) * v.patch_dim * D


def text_fwd_flops(t, T) -> float:
    D, L = t.hidden_size, t.num_hidden_layers
    per_layer = 2.0 * T * D * (3 * D) + 2.0 * T * D * D + 2.0 * 2.0 * T * D * t.intermediate_size + 4.0 * T * T * D
    return L * per_layer


def step_flops_per_image(cfg, T) -> float:
    """Algorithmic FLOPs of one distill step per image, north_star regime: 3 x F_vis (fwd + dgrad + wgrad) + 1 x
    F_txt (frozen) + projections (32.4 GFLOP/img for ViT-B/32)."""
    fv = vision_fwd_flops(cfg.vision) + 2.0 * cfg.vision.hidden_size * cfg.projection_dim
    ft = text_fwd_flops(cfg.text, T) + 2.0 * cfg.text.hidden_size * cfg.projection_dim
    return 3.0 * fv + ft


class GemmTimer:
    """HIP events around every dclip_gemm_f32 launch, on the stream the kernels are launched on (torch's current
    stream — torch.cuda.Event records there).  Sum(flops) / Sum(elapsed) over the timed region is the roofline
    figure for the dominant kernel."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def install(self):
        from dclip_amd import ops
        inner = ops.gemm
        timer = self

        def timed_gemm(a, b, layout, **kw):
            if not timer.enabled:
                return inner(a, b, layout, **kw)
            M, K = a.shape if layout & ops.A_KMAJOR else a.shape[::-1]
            N = b.shape[0] if layout & ops.B_KMAJOR else b.shape[1]
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner(a, b, layout, **kw)
            e1.record()
            timer.records.append((2.0 * M * N * K, e0, e1, (M, N, K, layout)))
            return out

        ops.gemm = timed_gemm

    def by_shape(self):
        """{(M,N,K,layout): (launches, total ms, TFLOP/s)} — tuning aid (DCLIP_BENCH_SHAPES=1 prints it to stderr)."""
        agg = {}
        for fl, e0, e1, key in self.records:
            n, ms, f = agg.get(key, (0, 0.0, 0.0))
            agg[key] = (n + 1, ms + e0.elapsed_time(e1), f + fl)
        return {k: (n, ms, f / (ms * 1e-3) / 1e12) for k, (n, ms, f) in agg.items()}

    def summary(self):
        flops = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return flops, ms, len(self.records)


def cpu_baseline(seconds: float = 12.0):
    """The reference's step arithmetic restated on the CPU (oracle/dclip_oracle.py, pinned to the reference's own
    outputs by tests/golden): BASELINE config c1 — ViT-B/32 + text tower, bs=8, forward + backward — timed on the
    host cores for a bounded sample."""
    from oracle import dclip_oracle as O
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 32))
    torch.set_num_threads(cores)
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and v.dim() > 0) for k, v in sd.items()}
    B = 8
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    ids = synth.synth_input_ids(B, cfg.text, seed=3)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1)

    def one():
        out = O.distill_step(p, cfg, pix, ids, t_img)
        out["loss"].backward()
        for v in p.values():
            v.grad = None

    one()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or n < 2:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"config c1: ViT-B/32 + text tower, bs={B}, fwd+bwd, {n} steps in {dt:.1f} s "
                      f"(oracle/dclip_oracle.py, torch {torch.__version__} CPU, all towers trainable as in the "
                      f"reference's as-written step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--model", default="ViT-B/32", choices=list(dcfg.NAMED))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+backward only")
    ap.add_argument("--no-gemm-events", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+backward from a captured HIP graph (dclip_amd/graph.py; N=1 only).  Per-launch "
                         "GEMM events do not exist inside a graph: `roofline` is then taken from an eager pass of the "
                         "same step before the timed region")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"],
                    help="c2 (default, the benched config): teacher image embedding given.  c3: the meta-teacher runs "
                         "inside the step on --regions crops per image (separate frozen teacher CLIP of --teacher-model)")
    ap.add_argument("--regions", type=int, default=8)
    ap.add_argument("--teacher-model", default=None, choices=list(dcfg.NAMED))
    ap.add_argument("--tower-precision", default="fp32", choices=["fp32", "bf16"],
                    help="c3 only: GEMM input precision of the FROZEN teacher towers (student is always fp32)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                             f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("DCLIP_DIST_BACKEND", "nccl")     # "gloo" lets N ranks share one GPU for a rehearsal
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"{world} ranks need {world} GPUs (found {ndev})")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    from dclip_amd import dist as ddist
    group = ddist.init_from_env(backend) if world > 1 else None

    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    from dclip_amd import optim

    cfg = dcfg.NAMED[args.model]()
    student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0), device=dev)   # same init on all ranks
    tcfg = None
    if args.workload == "c3":
        tcfg = dcfg.NAMED[args.teacher_model or args.model]()
        if tcfg.projection_dim != cfg.projection_dim:
            raise SystemExit("teacher and student must share the embedding width")
        teacher_clip = from_hf_state_dict(tcfg, synth.synth_clip_state_dict(tcfg, seed=7), device=dev)
        for p_ in teacher_clip.parameters():
            p_.requires_grad = False
        teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=cfg.projection_dim // 64,
                                       clip_model=teacher_clip, tower_precision=args.tower_precision).to(dev)
    else:
        teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=cfg.projection_dim // 64,
                                       clip_model=student).to(dev)
    hp = argparse.Namespace(learning_rate=1e-6, warmup_steps=0, total_steps=10 ** 6, train_batch_size=args.batch,
                            eval_batch_size=args.batch)
    module = CLIPImageDistillation(hp, student, None, teacher=teacher, freeze_mode="north_star",
                                   process_group=group).to(dev)
    trainable = [p for p in module.parameters() if p.requires_grad]
    opt = None if args.no_optimizer else optim.FusedAdamW(trainable, lr=hp.learning_rate, max_grad_norm=0.5)
    sync = ddist.GradSync(trainable, group) if world > 1 else None
    if sync is not None:            # gradient all-reduce overlapped with the backward pass
        from dclip_amd import functional
        functional.set_grad_ready_hook(sync.on_grads_ready)

    B, T = args.batch, cfg.text.max_position_embeddings
    batch = {                                                     # resident in HBM before the timed region
        "pixel_values": synth.synth_pixel_values(B, cfg.vision, seed=rank).to(dev),
        "input_ids": synth.synth_input_ids(B, cfg.text, seed=100 + rank).to(dev),
    }
    if args.workload == "c3":
        batch["regions"] = synth.synth_regions(B, args.regions, tcfg.vision, seed=2000 + rank).to(dev)
    else:
        batch["teacher_image_emb"] = synth.synth_embeddings(B, cfg.projection_dim, seed=1000 + rank).to(dev)

    timer = GemmTimer()
    if not args.no_gemm_events:
        timer.install()

    graphed = None
    if args.graph:
        if world > 1:
            raise SystemExit("--graph captures a single-process step")
        from dclip_amd.graph import GraphedStep
        timer.enabled = True                     # eager pass: per-launch GEMM events for the roofline figure
        for _ in range(2):
            for p_ in trainable:
                p_.grad = None
            module.training_step(batch).backward()
        torch.cuda.synchronize()
        timer.enabled = False
        eager_gemm = timer.summary()
        eager_gemm = (eager_gemm[0] / 2, eager_gemm[1] / 2, eager_gemm[2] // 2)      # per step
        for p_ in trainable:
            p_.grad = None
        graphed = GraphedStep(module, batch)

    def step():
        if graphed is not None:
            loss = graphed.step()                # inputs already sit in the captured buffers
            if opt is not None:
                opt.step()                       # gradients stay allocated in the graph's pool: no zero_grad
            return loss
        loss = module.training_step(batch)      # N > 1: this rank's share of the global loss (dist.py)
        loss.backward()
        if sync is not None:
            sync.finish()
        if opt is not None:
            opt.step()
            opt.zero_grad(set_to_none=True)
        else:
            for p in trainable:
                p.grad = None
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = graphed is None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * B * args.steps / elapsed
        step_flops = step_flops_per_image(cfg, T) * B
        teacher_flops = 0.0
        if args.workload == "c3":      # + R frozen region forwards + the teacher's own text forward + cross-attention
            E, R = cfg.projection_dim, args.regions
            teacher_flops = B * (R * (vision_fwd_flops(tcfg.vision) + 2.0 * tcfg.vision.hidden_size * E)
                                 + text_fwd_flops(tcfg.text, T) + 2.0 * T * tcfg.text.hidden_size * E
                                 + 8.0 * E * E * (T + R) + 8.0 * T * R * E)
        gflops, gms, glaunches = timer.summary() if not args.no_gemm_events else (0.0, 0.0, 0)
        if graphed is not None:                  # scale the eager per-step figures to the K timed steps
            gflops, gms, glaunches = eager_gemm[0] * args.steps, eager_gemm[1] * args.steps, eager_gemm[2] * args.steps
        achieved = gflops / (gms * 1e-3) / 1e12 if gms > 0 else None
        if os.environ.get("DCLIP_BENCH_SHAPES") and not args.no_gemm_events:
            for (M_, N_, K_, lay), (n, ms, tf) in sorted(timer.by_shape().items(), key=lambda kv: -kv[1][1]):
                print(f"gemm M={M_:6d} N={N_:5d} K={K_:6d} layout={lay} launches/step={n / args.steps:5.1f} "
                      f"ms/step={ms / args.steps:7.3f} {tf:6.1f} TF/s", file=sys.stderr)
        traffic = None
        pmc = os.path.join(REPO, "profiles", "gemm_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        line = {
            "metric": "distill-step images/sec at ViT-B/32 bs=256",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE config c2: {cfg.name} distill step, bs={B}/GPU, 224x224 + {T}-token "
                                    f"synthetic pairs, contrastive+cosine loss, fp32, vision trainable / text frozen "
                                    f"(north_star regime), " if args.workload == "c2" else
                                    f"BASELINE config c3 (extra, not the benched config): {cfg.name} student fp32 + "
                                    f"meta-teacher in the step ({args.regions} region crops/img through a frozen "
                                    f"{tcfg.name} tower with {args.tower_precision} GEMM inputs, token-level text, "
                                    f"cross-modal attention + aggregation), bs={B}/GPU, ")
                                   + ("fwd+bwd only" if opt is None else "fwd+bwd + clip-norm 0.5 + AdamW")
                                   + (", fwd+bwd replayed from a HIP graph" if graphed is not None else ""),
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "loss": float(last.detach())},
            "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                         "achieved": None if achieved is None else round(achieved, 2),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": None if achieved is None else round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": traffic,
                         "launches_per_step": glaunches // max(1, args.steps),
                         "gemm_ms_per_step": round(gms / max(1, args.steps), 3),
                         "gemm_flops_per_step": gflops / max(1, args.steps)},
            # whole step priced two ways: `frac` with the ALGORITHMIC flops of the reference's arithmetic (SURVEY §8d:
            # 3 F_vis + F_txt per image), `frac_executed` with the GEMM flops actually launched — lower because the last
            # vision layer is evaluated on the CLS rows only, which is exact ("dead-row elimination")
            "roofline_step": {"bound": "mfma", "achieved": round(step_flops / (ms_per_step * 1e-3) / 1e12, 2),
                              "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              "frac_executed": None if not gms else round(
                                  gflops / max(1, args.steps) / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              "flops_per_image": step_flops_per_image(cfg, T)},
        }
        if args.workload == "c3":
            # mixed-precision floor: student flops at the fp32 matrix peak + frozen-teacher flops at the peak of the
            # precision they were multiplied in (bf16 dense 2500 TF/s, MI355X_MICROARCH.md)
            tpeak = 2500.0 if args.tower_precision == "bf16" else PEAK_F32_MFMA_TFLOPS
            t_min_ms = (step_flops / PEAK_F32_MFMA_TFLOPS + teacher_flops / tpeak) / 1e9
            line["roofline_step"] = {"bound": "mfma", "unit": "ms", "floor_ms": round(t_min_ms, 3),
                                     "frac": round(t_min_ms / ms_per_step, 4),
                                     "student_flops_per_image": step_flops / B, "teacher_flops_per_image": teacher_flops / B,
                                     "teacher_peak_tflops": tpeak}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(float(os.environ.get("DCLIP_BENCH_CPU_SECONDS", "12")))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
