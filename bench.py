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

import faulthandler  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# Watchdog, default ON: a run that has not finished after this many seconds (counted from here, i.e. after `import torch`,
# which alone can take two minutes on a fresh box) dumps every thread's Python stack to stderr and exits non-zero — an
# N-rank run cannot die silently or sit in a collective until the driver's limit.  Ranks: 240 s (a healthy 8-rank run
# needs well under a minute after import); the single process, which also times the CPU baseline: 900 s.
WATCHDOG_S = int(os.environ.get("DCLIP_BENCH_WATCHDOG_S", os.environ.get("DCLIP_BENCH_TRACE_AFTER", "240" if "RANK" in os.environ else "900")))
if WATCHDOG_S > 0:
    faulthandler.dump_traceback_later(WATCHDOG_S, exit=True)
PG_TIMEOUT_S = float(os.environ.get("DCLIP_BENCH_PG_TIMEOUT_S", "180"))

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


class single_stream:
    """Steps whose GEMM launches are bracketed by HIP events run every kernel ALONE on one stream: the second streams of the
    step (the meta-teacher beside the student's forward and its text tower beside its region tower, the weight-gradient GEMMs
    beside the data-gradient chain, the frozen text tower) are switched off for them, so that a per-launch duration is that of the kernel, not of the kernel sharing the
    chip with another stream's work.  `value` comes from steps with the second streams on."""

    def __init__(self, module, on: bool = True):
        self.module, self.on = module, on

    def __enter__(self):
        if self.on:
            self.prev = (self.module.overlap_teacher, self.module.overlap_frozen_text,
                         {k: os.environ.get(k) for k in self.ENV})
            self.module.overlap_teacher = False
            self.module.overlap_frozen_text = False
            for k in self.ENV:
                os.environ[k] = "0"
        return self

    # second streams that are switched by environment variables read at call time (engine._SideWgrads, the meta-teacher's
    # text tower beside its region tower)
    ENV = ("DCLIP_BF16_WGRAD_STREAM", "DCLIP_TEACHER_TEXT_STREAM")

    def __exit__(self, *exc):
        if self.on:
            self.module.overlap_teacher, self.module.overlap_frozen_text = self.prev[0], self.prev[1]
            for k, v in self.prev[2].items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return False


class ClockSampler:
    """Shader clock (MHz) and socket power (W) of THIS process's GPU, read from the hwmon files of its PCI function every
    50 ms on a host thread while a region runs: the MFMA peaks of the roofline are quoted at 2.4 GHz, and at ~1.35 kW the card
    does not hold that clock under these steps (profiles/r03_clocks_under_load.log).  Purely informative: every failure
    (files absent or unreadable) leaves `summary()` at None."""
    NOMINAL_MHZ = 2400.0

    def __init__(self, device_index: int):
        import glob
        self.freq = self.power = None
        self.samples = []
        try:
            pr = torch.cuda.get_device_properties(device_index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for h in glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*"):
                if os.path.exists(os.path.join(h, "freq1_input")):
                    self.freq = os.path.join(h, "freq1_input")
                    self.power = os.path.join(h, "power1_input") if os.path.exists(os.path.join(h, "power1_input")) else None
        except Exception:
            self.freq = None
        self._stop = None
        self._thread = None

    def _run(self):
        while not self._stop.is_set():
            try:
                mhz = int(open(self.freq).read()) / 1e6
                w = int(open(self.power).read()) / 1e6 if self.power else None
                self.samples.append((mhz, w))
            except Exception:
                pass
            self._stop.wait(0.05)

    def __enter__(self):
        if self.freq:
            import threading
            self._stop = threading.Event()
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=1.0)
            self._thread = None
        return False

    def summary(self):
        good = sorted(m for m, _ in self.samples if m > 0)
        if len(good) < 3:
            return None
        good = good[len(good) // 4:]                 # the region starts from an idle card: drop the ramp
        pw = sorted(w for _, w in self.samples if w)
        return {"sclk_mhz": round(good[len(good) // 2], 0), "power_w": round(pw[len(pw) // 2], 0) if pw else None,
                "samples": len(self.samples), "nominal_mhz": self.NOMINAL_MHZ,
                "source": "hwmon freq1_input / power1_input of the device's PCI function, every 50 ms during the timed region"}


class GemmTimer:
    """HIP events around every dclip_gemm_f32 launch, on the stream the kernels are launched on (torch's current
    stream — torch.cuda.Event records there).  Sum(flops) / Sum(elapsed) over the timed region is the roofline
    figure for the dominant kernel."""

    def __init__(self):
        self.records = []          # dclip_gemm_f32 launches
        self.records16 = []        # dclip_gemm_bf16 launches (frozen towers / bf16 student)
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
        inner16 = ops.gemm_bf16

        def timed_gemm_bf16(a, w, **kw):
            if not timer.enabled:
                return inner16(a, w, **kw)
            M = a.shape[0]
            N = kw.get("n") or w.shape[0]
            K = kw.get("k") or min(a.shape[1], w.shape[1])
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner16(a, w, **kw)
            e1.record()
            timer.records16.append((2.0 * M * N * K, e0, e1, (M, N, K, "bf16")))
            return out

        ops.gemm_bf16 = timed_gemm_bf16

    def by_shape(self):
        """{(M,N,K,layout): (launches, total ms, TFLOP/s)} — tuning aid (DCLIP_BENCH_SHAPES=1 prints it to stderr)."""
        agg = {}
        for fl, e0, e1, key in self.records + self.records16:
            n, ms, f = agg.get(key, (0, 0.0, 0.0))
            agg[key] = (n + 1, ms + e0.elapsed_time(e1), f + fl)
        return {k: (n, ms, f / (ms * 1e-3) / 1e12) for k, (n, ms, f) in agg.items()}

    def summary(self, bf16: bool = False):
        recs = self.records16 if bf16 else self.records
        flops = sum(r[0] for r in recs)
        ms = sum(r[1].elapsed_time(r[2]) for r in recs)
        return flops, ms, len(recs)


def cpu_baseline(seconds: float = 12.0, regime: str = "north_star"):
    """The reference's step arithmetic restated on the CPU (oracle/dclip_oracle.py, pinned to the reference's own
    outputs by tests/golden), ViT-B/32 + text tower, forward + backward, timed on the host cores for a bounded sample.
      regime "north_star": the SAME regime as the GPU number — vision tower trainable, text tower frozen with one
                           forward shared by student and teacher, teacher image embedding given; bs 32;
      regime "as_written": BASELINE config c1 exactly — bs 8, every tower trainable as the reference's step leaves them."""
    from oracle import dclip_oracle as O
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 32))
    torch.set_num_threads(cores)
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0)
    if regime == "north_star":
        B = 32
        p = {k: v.clone().requires_grad_(v.is_floating_point() and v.dim() > 0
                                         and (k.startswith("vision_model.") or k == "visual_projection.weight"))
             for k, v in sd.items()}
    else:
        B = 8
        p = {k: v.clone().requires_grad_(v.is_floating_point() and v.dim() > 0) for k, v in sd.items()}
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
    what = ("north_star regime as in `value`: vision trainable, text frozen + shared forward, teacher image embedding given"
            if regime == "north_star" else
            "BASELINE config c1: all towers trainable as in the reference's as-written step")
    return {"value": round(B * n / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"ViT-B/32 + text tower, bs={B}, fwd+bwd, {n} steps in {dt:.1f} s; {what} "
                      f"(oracle/dclip_oracle.py, torch {torch.__version__} CPU)"}


def build_workload(workload, model, teacher_model, batch_size, regions, student_precision, tower_precision, dev, group, rank,
                   fast_teacher_init: bool = False):
    """The distillation module of one BASELINE config on synthetic inputs resident in HBM -> (module, student config, teacher
    config or None, batch).  c2: teacher image embedding given.  c3 / c5: the meta-teacher runs inside the step on `regions`
    crops per image through a separate frozen teacher CLIP."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    meta = workload in ("c3", "c5")
    cfg = dcfg.NAMED[model]()
    student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0), device=dev)   # same init on all ranks
    tcfg = None
    if meta:
        tcfg = dcfg.NAMED[teacher_model or model]()
        # (fast_teacher_init: the frozen teacher's random weights drawn on the GPU — the extra legs of the default run; a
        # ViT-L/14 state dict takes ~10 s on the host)
        teacher_clip = from_hf_state_dict(tcfg, synth.synth_clip_state_dict(tcfg, seed=7, device=dev if fast_teacher_init else None),
                                          device=dev)
        for p_ in teacher_clip.parameters():
            p_.requires_grad = False
        E = tcfg.projection_dim               # a wider teacher is bridged to the student inside the module (c5)
        teacher = PatchTextAggregation(embed_dim=E, num_heads=E // 64, clip_model=teacher_clip,
                                       tower_precision=tower_precision).to(dev)
    else:
        teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=cfg.projection_dim // 64,
                                       clip_model=student).to(dev)
    # the cross-modal block too from a seeded state (its constructor draws from the process's default RNG): the same frozen
    # teacher in every run and on every rank
    teacher.cross_modal_attention.load_state_dict(
        {k: v.to(dev) for k, v in synth.synth_cross_modal_state_dict(teacher.embed_dim, seed=31).items()})
    hp = argparse.Namespace(learning_rate=1e-6, warmup_steps=0, total_steps=10 ** 6, train_batch_size=batch_size,
                            eval_batch_size=batch_size)
    module = CLIPImageDistillation(hp, student, None, teacher=teacher, freeze_mode="north_star",
                                   process_group=group, student_precision=student_precision).to(dev)
    batch = {                                                     # resident in HBM before the timed region
        "pixel_values": synth.synth_pixel_values(batch_size, cfg.vision, seed=rank).to(dev),
        "input_ids": synth.synth_input_ids(batch_size, cfg.text, seed=100 + rank).to(dev),
    }
    if meta:
        batch["regions"] = synth.synth_regions(batch_size, regions, tcfg.vision, seed=2000 + rank).to(dev)
        # token-padding size of the batch, known on the host (the reference sizes it from the caption strings): the synthetic
        # captions are 77 tokens long = 75 word tokens; with it the teacher never synchronises the host with the stream
        batch["max_tokens"] = int((batch["input_ids"] == cfg.text.eos_token_id).int().argmax(dim=1).max()) - 1
    else:
        batch["teacher_image_emb"] = synth.synth_embeddings(batch_size, cfg.projection_dim, seed=1000 + rank).to(dev)
    return module, cfg, tcfg, batch


def meta_teacher_flops(cfg, tcfg, B, R, T) -> float:
    """R frozen region forwards + the teacher's own text forward + cross-attention (+ the bridge), per step."""
    E = tcfg.projection_dim
    return B * (R * (vision_fwd_flops(tcfg.vision) + 2.0 * tcfg.vision.hidden_size * E)
                + text_fwd_flops(tcfg.text, T) + 2.0 * T * tcfg.text.hidden_size * E
                + 8.0 * E * E * (T + R) + 8.0 * T * R * E
                + (4.0 * E * cfg.projection_dim if E != cfg.projection_dim else 0.0))


def bf16_traffic():
    """HBM-side bytes per bf16 GEMM launch of the c3 step from the committed counter passes (profiles/), or None."""
    path = os.path.join(REPO, "profiles", "gemm_bf16_traffic.json")
    return json.load(open(path))["hbm_bytes_per_launch"] if os.path.exists(path) else None


def extra_config(dev, timer, which: str, steps: int, warmup: int):
    """One of the OTHER BASELINE configs timed for a few steps AFTER the contract's timed region, so that the driver's record
    carries their figures too (fwd + bwd + clip + AdamW, eager launches):
      "c3"  as quoted: ViT-B/32 student in bf16 + meta-teacher on 8 region crops per image through frozen bf16 towers, 256 pairs;
      "c5"  one GPU's share: ViT-L/14 teacher towers (bf16) over the bf16 ViT-B/32 student through the declared bridge, 512 pairs;
      "c4"  one GPU's share: ViT-B/16 distill step, fp32, 128 pairs (teacher image embedding given, as in c2).
    `value` comes from `steps` steps WITHOUT per-launch events; the GEMM figure from two further steps with HIP events around
    every GEMM launch (they cost ~3 % of a bf16 step)."""
    from dclip_amd import optim
    R = 8
    if which == "c3":
        B, spec, dtype = 256, ("c3", "ViT-B/32", "ViT-B/32", 256, R, "bf16", "bf16"), "bf16"
        what = ("BASELINE config c3 as quoted: ViT-B/32 student bf16 (fp32 masters) + meta-teacher in the step (8 region crops/img "
                "through a frozen ViT-B/32 tower, bf16 GEMM inputs), bs=256")
    elif which == "c5":
        B, spec, dtype = 512, ("c5", "ViT-B/32", "ViT-L/14", 512, R, "bf16", "bf16"), "bf16"
        what = ("BASELINE config c5, one GPU's share (512 of the 4096 global pairs): ViT-B/32 student bf16 + meta-teacher on 8 region "
                "crops/img through frozen ViT-L/14 towers (bf16 GEMM inputs), 768->512 teacher bridge, bs=512")
    else:
        B, spec, dtype = 128, ("c2", "ViT-B/16", None, 128, R, "fp32", "fp32"), "f32"
        what = "BASELINE config c4, one GPU's share (128 of the 1024 global pairs): ViT-B/16 distill step, fp32, bs=128"
    module, cfg, tcfg, batch = build_workload(*spec, dev, None, 0, fast_teacher_init=True)
    # (as in the main loop: the frozen text tower beside the vision forward, except in the event-bracketed steps)
    module.overlap_frozen_text = not os.environ.get("DCLIP_NO_TEXT_OVERLAP") and os.environ.get("DCLIP_EAGER_TEXT_OVERLAP", "1") != "0"
    T = cfg.text.max_position_embeddings
    trainable = [p for p in module.parameters() if p.requires_grad]
    opt = optim.FusedAdamW(trainable, lr=1e-6, max_grad_norm=0.5)

    prefetch_ok = which != "c4" and os.environ.get("DCLIP_TEACHER_PREFETCH", "1") != "0"

    def one(prefetch=False):
        loss = module.training_step(batch)
        if prefetch:
            module.prefetch_teacher(batch)       # the next step's teacher beside this step's backward (see step() in main)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss.detach()

    for k_ in range(warmup + 1):            # (the first one in the event steps' one-stream configuration: see the main loop)
        with single_stream(module, k_ == 0):
            one(prefetch_ok and 0 < k_ < warmup)
    timer.enabled = False
    torch.cuda.synchronize()
    clock = ClockSampler(dev.index if dev.index is not None else 0)
    t0 = time.perf_counter()
    with clock:
        for k_ in range(steps):
            last = one(prefetch_ok and k_ + 1 < steps)      # exactly K teacher passes inside the K timed steps
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ev_steps = 2 if which != "c5" else 1
    with single_stream(module):
        # one unrecorded step first: after the pipelined steps the allocator hands the one-stream configuration fresh blocks
        # again, and kernels touching freshly mapped memory read slow (c5's single recorded step: 0.38 against 0.44)
        timer.enabled = False
        one()
        timer.records16.clear()
        timer.records.clear()
        timer.enabled = True
        for _ in range(ev_steps):
            one()
        torch.cuda.synchronize()
    timer.enabled = False
    ms = el * 1e3 / steps
    student_flops = step_flops_per_image(cfg, T) * B
    out = {"workload": what + ", fwd+bwd + clip-norm 0.5 + AdamW, eager launches; the GEMM roofline from further steps with HIP "
                              "events around every GEMM launch",
           "value": round(B * steps / el, 2), "unit": "images/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
           "dtype": dtype, "loss": float(last)}
    ck = clock.summary()
    if ck is not None:
        out["clock"] = ck
    if dtype == "bf16":
        f16, ms16, n16 = timer.summary(bf16=True)
        floor_ms = (student_flops + meta_teacher_flops(cfg, tcfg, B, R, T)) / 2500.0 / 1e9
        out["roofline_bf16"] = {"bound": "mfma", "kernel": "gemm_bf16_pp_kernel (v_mfma_f32_16x16x32_bf16)",
                                "achieved": round(f16 / (ms16 * 1e-3) / 1e12, 2) if ms16 else None, "peak": 2500.0,
                                "unit": "TFLOP/s", "frac": round(f16 / (ms16 * 1e-3) / 1e12 / 2500.0, 4) if ms16 else None,
                                "launches_per_step": n16 // ev_steps, "gemm_ms_per_step": round(ms16 / ev_steps, 3),
                                "traffic": bf16_traffic() if which == "c3" else None}
        out["roofline_step"] = {"bound": "mfma", "unit": "ms", "floor_ms": round(floor_ms, 3), "frac": round(floor_ms / ms, 4),
                                "peak_tflops": 2500.0}
    else:
        f32, ms32, n32 = timer.summary()
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                           "achieved": round(f32 / (ms32 * 1e-3) / 1e12, 2) if ms32 else None, "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(f32 / (ms32 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if ms32 else None,
                           "launches_per_step": n32 // ev_steps, "gemm_ms_per_step": round(ms32 / ev_steps, 3)}
        out["roofline_step"] = {"bound": "mfma", "achieved": round(student_flops / (ms * 1e-3) / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(student_flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                "flops_per_image": step_flops_per_image(cfg, T)}
    del module, opt, batch, trainable
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def self_launch(n: int) -> int:
    """Run `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same args>` as a child process,
    relay rank 0's JSON line (the only thing the ranks print on stdout) and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    # own session: if the ranks' watchdogs somehow do not fire, the whole launch (launcher + ranks = one process group, the
    # one started here) is ended by its group id after the watchdog time plus the import / rendezvous allowance
    limit = (WATCHDOG_S + 150) if WATCHDOG_S > 0 else None
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)
        out, _ = proc.communicate()
        print(f"bench.py: the {n}-rank launch did not finish within {limit} s and was killed", file=sys.stderr)
        return 124
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in out.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and lines:
        print(lines[-1], flush=True)
        return 0
    print(f"bench.py: the {n}-rank launch failed (exit code {proc.returncode})", file=sys.stderr)
    return proc.returncode or 1


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
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the untimed-by-the-contract extra measurements after the timed region (forward+backward "
                         "without the optimizer; HIP-graph replay)")
    ap.add_argument("--eager", action="store_true",
                    help="launch every step kernel by kernel — the default since round 3 (kept as a flag for scripts that pass it)")
    ap.add_argument("--hybrid-graph", action="store_true",
                    help="N=1, benched workload: forward+backward replayed from a captured HIP graph (bit-identical to the eager "
                         "step, dclip_amd/graph.py), every 5th step of the timed region launched eagerly with HIP events around "
                         "each GEMM launch.  The default of rounds 1-2; with the frozen text tower on its second stream in eager "
                         "steps too, the eager loop measures the same (62.22 vs 62.31 ms without events) or faster (64.5 vs 65.3 "
                         "with the sampled event steps) on the same box, so it is the default now")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+backward from a captured HIP graph (dclip_amd/graph.py; N=1 only).  Per-launch "
                         "GEMM events do not exist inside a graph: `roofline` is then taken from an eager pass of the "
                         "same step before the timed region")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5"],
                    help="c2 (default, the benched config): teacher image embedding given.  c3: the meta-teacher runs "
                         "inside the step on --regions crops per image (separate frozen teacher CLIP of --teacher-model). "
                         "c5: c3 with a ViT-L/14 teacher (768-wide, bridged to the 512-wide ViT-B/32 student by the "
                         "declared frozen projection) and bf16 teacher towers unless --tower-precision says otherwise")
    ap.add_argument("--regions", type=int, default=8)
    ap.add_argument("--teacher-model", default=None, choices=list(dcfg.NAMED))
    ap.add_argument("--student-precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: the student's vision tower multiplies in bf16 (forward, dgrad, wgrad; fp32 master weights) — "
                         "what BASELINE quotes c3 / c5 in.  The benched config c2 is fp32")
    ap.add_argument("--tower-precision", default=None, choices=["fp32", "bf16"],
                    help="c3 / c5: GEMM input precision of the FROZEN teacher towers (student is always fp32); default "
                         "fp32 for c3, bf16 for c5")
    args = ap.parse_args()
    if args.workload == "c5":
        args.teacher_model = args.teacher_model or "ViT-L/14"
        args.tower_precision = args.tower_precision or "bf16"
    args.tower_precision = args.tower_precision or "fp32"
    meta = args.workload in ("c3", "c5")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process has touched the GPU
        # yet (no HIP call, no torch.cuda.is_available()), and the ranks are CHILD processes (never an exec).
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("DCLIP_DIST_BACKEND", "nccl")     # "gloo" lets N ranks share one GPU for a rehearsal
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"{world} ranks need {world} GPUs (found {ndev})")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    from dclip_amd import dist as ddist
    group = ddist.init_from_env(backend, timeout_s=PG_TIMEOUT_S) if world > 1 else None
    # per-rank progress lines on stderr (host time since start; the host runs ahead of the GPU except where a collective or
    # a synchronize holds it): default on for gloo rehearsals, DCLIP_BENCH_STEP_LOG=1 elsewhere
    step_log = os.environ.get("DCLIP_BENCH_STEP_LOG", "1" if (world > 1 and backend == "gloo") else "0") == "1"
    t_start = time.perf_counter()

    def note(msg):
        if step_log:
            print(f"[rank {rank} +{time.perf_counter() - t_start:7.2f}s] {msg}", file=sys.stderr, flush=True)

    from dclip_amd import optim
    module, cfg, tcfg, batch = build_workload(args.workload, args.model, args.teacher_model, args.batch, args.regions,
                                              args.student_precision, args.tower_precision, dev, group, rank)
    hp = module.hparams
    trainable = [p for p in module.parameters() if p.requires_grad]
    opt = None if args.no_optimizer else optim.FusedAdamW(trainable, lr=hp.learning_rate, max_grad_norm=0.5)
    sync = ddist.GradSync(trainable, group, timing=True) if world > 1 else None
    if sync is not None:            # gradient all-reduce overlapped with the backward pass
        from dclip_amd import functional
        functional.set_grad_ready_hook(sync.on_grads_ready)
        functional.set_grad_alloc(sync.grad_buffer)       # wgrad GEMMs write straight into the all-reduce buckets
    B, T = args.batch, cfg.text.max_position_embeddings

    timer = GemmTimer()
    if not args.no_gemm_events:
        timer.install()

    def eager_fwd_bwd():
        for p_ in trainable:
            p_.grad = None
        module.training_step(batch).backward()

    graphed = None
    if args.graph:
        if world > 1:
            raise SystemExit("--graph captures a single-process step")
        from dclip_amd.graph import GraphedStep
        timer.enabled = True                     # eager pass: per-launch GEMM events for the roofline figure
        for _ in range(2):
            eager_fwd_bwd()
        torch.cuda.synchronize()
        timer.enabled = False
        eager_gemm = timer.summary()
        eager_gemm = (eager_gemm[0] / 2, eager_gemm[1] / 2, eager_gemm[2] // 2)      # per step
        for p_ in trainable:
            p_.grad = None
        graphed = GraphedStep(module, batch)

    # --hybrid-graph (N = 1, c2; the default of rounds 1-2): HIP-graph replay of forward + backward, with every `stride`-th step
    # of the timed region launched eagerly so that the GEMM launches of that step can be bracketed by HIP events.  Default:
    # every step launched eagerly (the host issues a step's ~430 launches in a fraction of its 62 ms).
    hybrid = None
    # the frozen text tower's forward on a second stream beside the vision forward: inside the replayed graph always; in
    # eagerly launched steps (N > 1, --eager, the meta-teacher workloads) too unless DCLIP_EAGER_TEXT_OVERLAP=0; never in
    # the event-sampled steps (single_stream)
    text_overlap = not os.environ.get("DCLIP_NO_TEXT_OVERLAP")
    eager_text_overlap = os.environ.get("DCLIP_EAGER_TEXT_OVERLAP", "1") != "0"
    module.overlap_frozen_text = text_overlap and eager_text_overlap
    exec_note = "eager launches" + ("; the frozen text tower's forward on a second stream beside the vision forward (not in the "
                                    "event-sampled steps)" if module.overlap_frozen_text else "")
    if meta:
        exec_note = ("eager launches; the frozen student text tower and the meta-teacher on second streams — step n+1's teacher is started after step n's forward and "
                     "runs beside its backward, optimizer and the next student forward (one teacher pass per step; a step next "
                     "to an event-sampled one runs its teacher in the step, beside the student's image forward)"
                     + (", the bf16 weight-gradient GEMMs beside the data-gradient chain" if args.student_precision == "bf16" else "")
                     + "; the steps sampled for per-launch GEMM events launch every kernel alone on one stream")
    if graphed is None and world == 1 and not meta and args.hybrid_graph and not args.eager and opt is not None:
        try:
            from dclip_amd.graph import GraphedStep
            for p_ in trainable:
                p_.grad = None
            # inside the replayed graph the frozen text tower's forward runs on a second stream beside the vision forward;
            # the eager (sampled, event-bracketed) steps launch every kernel alone on one stream
            module.overlap_frozen_text = text_overlap
            try:
                hybrid = GraphedStep(module, batch)
            finally:
                module.overlap_frozen_text = text_overlap and eager_text_overlap
            hybrid_grads = [(p_, p_.grad) for p_ in trainable if p_.grad is not None]     # the graph's static gradients
            exec_note = ("HIP-graph replay of forward+backward (frozen text forward on a second stream beside the vision forward); "
                         "sampled steps launched eagerly, one stream, per-launch GEMM events")
        except Exception as exc:            # a box where capture fails still gets its eager number
            hybrid = None
            exec_note = f"eager launches (HIP-graph capture failed: {type(exc).__name__}: {exc})"[:240]
            for p_ in trainable:
                p_.grad = None

    # meta-teacher workloads: the teacher of step n+1 is launched during step n (set per step by the loops below: never
    # ahead of a step whose GEMM launches are bracketed by events — that one runs its own teacher, alone on one stream)
    prefetch_ok = meta and os.environ.get("DCLIP_TEACHER_PREFETCH", "1") != "0"
    prefetch_next = {"on": False}

    def step(with_opt=True, eager_now=False):
        if hybrid is not None and not eager_now:
            for p_, g_ in hybrid_grads:          # an eager step in between left .grad at None
                p_.grad = g_
            loss = hybrid.step()
            if with_opt:
                opt.step()
            return loss
        if hybrid is not None:                   # the eager step of the hybrid schedule: fresh gradient tensors
            for p_ in trainable:
                p_.grad = None
        if graphed is not None:
            loss = graphed.step()                # inputs already sit in the captured buffers
            if opt is not None and with_opt:
                opt.step()                       # gradients stay allocated in the graph's pool: no zero_grad
            return loss
        loss = module.training_step(batch)      # N > 1: this rank's share of the global loss (dist.py)
        if prefetch_next["on"]:
            # the NEXT step's meta-teacher (the same synthetic batch) is started now, on its own stream: it runs beside this
            # step's backward + optimizer and the next step's student forward (CLIPImageDistillation.prefetch_teacher).
            # Every step still contains exactly one teacher pass and one student pass.
            module.prefetch_teacher(batch)
        loss.backward()
        if sync is not None:
            sync.finish()
        if opt is not None and with_opt:
            opt.step()
            opt.zero_grad(set_to_none=True)
        else:
            for p in trainable:
                p.grad = None
        return loss.detach()            # keep no reference to the autograd graph (a later HIP-graph capture needs none)

    def timed(n, fn):
        """n calls of fn bracketed by synchronize (+ barrier) on both sides; MAX over ranks, seconds."""
        torch.cuda.synchronize()
        if world > 1:
            note("barrier in")
            dist.barrier()
            note("barrier out")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        return el, out

    note("model + batch resident; warm-up starts")
    for i_ in range(args.warmup):
        prefetch_next["on"] = prefetch_ok and i_ + 1 < args.warmup       # (timed step 0 is an event-bracketed one)
        # the first warm-up step runs in the configuration of the event-bracketed steps (every kernel on one stream), so that
        # its one-time costs — scratch lanes growing, first touch of freshly mapped memory — stay outside the timed region:
        # without it c5 measured 977 ms / GEMMs 0.33 with one warm-up step against 709-728 ms / 0.43-0.44 with two or more
        with single_stream(module, i_ == 0 and graphed is None and not args.no_gemm_events):
            step()
        note(f"warm-up step {i_} launched")
    if sync is not None:
        sync.reset_stats()
    # HIP events around every GEMM launch cost ~1.8 % of the step (400 event records; measured 65.54 vs 64.38 ms on one
    # box), so they are recorded on every `stride`-th step of the timed region only: still measured live, inside the timed
    # region, on the launch stream — on a sample of the steps
    stride = 1 if args.steps < 8 else (4 if args.steps < 20 else 5)
    sampled = {"i": 0, "n": 0}

    def timed_step():
        on = graphed is None and not args.no_gemm_events and sampled["i"] % stride == 0
        timer.enabled = on
        sampled["n"] += int(on)
        sampled["i"] += 1
        next_on = graphed is None and not args.no_gemm_events and sampled["i"] % stride == 0
        prefetch_next["on"] = prefetch_ok and not on and not next_on and sampled["i"] < args.steps
        with single_stream(module, on):          # event-bracketed steps launch every kernel alone on one stream
            out_ = step(eager_now=on)
        note(f"timed step {sampled['i'] - 1} launched")
        return out_

    clock = ClockSampler(dev.index if dev.index is not None else 0)
    with clock:
        elapsed, last = timed(args.steps, timed_step)      # ---- THE timed region: exactly K steps
    timer.enabled = False
    n_sampled = max(1, sampled["n"])
    last_loss = float(last.detach())
    del last
    note(f"timed region done: {elapsed * 1e3 / args.steps:.2f} ms/step")
    comm = sync.stats() if sync is not None else None
    if sync is not None:
        # calibration leg (every rank): each gradient bucket all-reduced alone -> ms per bucket and the bus rate
        # 2(N-1)/N x bytes / time the collective reaches with nothing beside it
        per_bucket = sync.bucket_allreduce_ms()
        nbytes = sync.bucket_bytes()
        tot_ms, tot_b = sum(per_bucket), sum(nbytes)
        comm["allreduce_ms_per_bucket_alone"] = [round(x, 3) for x in per_bucket]
        comm["bucket_bytes"] = nbytes
        comm["allreduce_ms_per_step_alone"] = round(tot_ms, 3)
        comm["allreduce_busbw_GBps_alone"] = round(2.0 * (world - 1) / world * tot_b / (tot_ms * 1e-3) / 1e9, 2) if tot_ms else None
        note("bucket calibration done")

    # ---- extra legs, outside the contract's timed region ("optimizer excluded and reported separately")
    extra = {}
    if not args.no_extra_legs and graphed is None:
        n_x = max(3, min(args.steps, 10))
        if opt is not None:
            el, _ = timed(n_x, lambda: step(with_opt=False))
            extra["fwd_bwd_ms_per_step"] = round(el * 1e3 / n_x, 3)
            extra["fwd_bwd_images_per_s"] = round(world * B * n_x / el, 2)
            extra["optimizer_ms_per_step"] = round(elapsed * 1e3 / args.steps - el * 1e3 / n_x, 3)
        if hybrid is not None:               # the all-eager figure beside the default (graph) one
            el, _ = timed(n_x, lambda: step(eager_now=True))
            extra["eager_ms_per_step"] = round(el * 1e3 / n_x, 3)
            extra["eager_images_per_s"] = round(B * n_x / el, 2)
        if world == 1 and not meta and hybrid is None:     # (meta-teacher: host-sized token padding, not capturable)
            try:
                from dclip_amd.graph import GraphedStep
                for p_ in trainable:
                    p_.grad = None
                gs = GraphedStep(module, batch)

                def gstep():
                    loss = gs.step()
                    if opt is not None:
                        opt.step()
                    return loss

                for _ in range(2):
                    gstep()
                el, _ = timed(n_x, gstep)
                extra["graph_ms_per_step"] = round(el * 1e3 / n_x, 3)
                extra["graph_images_per_s"] = round(B * n_x / el, 2)
                del gs
            except Exception as exc:            # a graph variant is informative only; never fail the bench line on it
                extra["graph_error"] = f"{type(exc).__name__}: {exc}"[:200]

    if world > 1 and not args.no_extra_legs:
        # The same step WITHOUT the data-parallel parts, launched the same way (eagerly) on every rank at once: local
        # negatives only, no gradient all-reduce.  N-rank `value` / (N x this) is the scaling efficiency of like with like
        # (the N = 1 default line replays a HIP graph, ~3 % faster than eager launches).  After the timed region: the
        # replicas may drift apart here, nothing reads them afterwards.
        from dclip_amd import functional as _fn
        _fn.set_grad_ready_hook(None)
        _fn.set_grad_alloc(None)
        module.process_group = None
        sync_saved, sync = sync, None
        n_x = max(3, min(args.steps, 10))
        for _ in range(2):
            step()
        el, _ = timed(n_x, step)
        extra["n1_eager_ms_per_step"] = round(el * 1e3 / n_x, 3)
        extra["n1_eager_images_per_s"] = round(B * n_x / el, 2)
        sync = sync_saved
        note("single-GPU eager leg done")

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * B * args.steps / elapsed
        step_flops = step_flops_per_image(cfg, T) * B
        teacher_flops = 0.0
        if meta:
            teacher_flops = meta_teacher_flops(cfg, tcfg, B, args.regions, T)
        gflops, gms, glaunches = timer.summary() if not args.no_gemm_events else (0.0, 0.0, 0)
        n_ev = n_sampled                          # steps of the timed region that carried GEMM events
        if graphed is not None:                  # the eager per-step figures
            gflops, gms, glaunches = eager_gemm
            n_ev = 1
        achieved = gflops / (gms * 1e-3) / 1e12 if gms > 0 else None
        if os.environ.get("DCLIP_BENCH_SHAPES") and not args.no_gemm_events:
            for (M_, N_, K_, lay), (n, ms, tf) in sorted(timer.by_shape().items(), key=lambda kv: -kv[1][1]):
                print(f"gemm M={M_:6d} N={N_:5d} K={K_:6d} layout={lay} launches/step={n / n_ev:5.1f} "
                      f"ms/step={ms / n_ev:7.3f} {tf:6.1f} TF/s", file=sys.stderr)
        traffic, traffic_source = None, None
        pmc = os.path.join(REPO, "profiles", "gemm_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            traffic_source = "profiles/gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                             "committed; NOT re-measured in this run)"
        if args.workload == "c2":
            tag = "c2" if (cfg.name == "ViT-B/32" and B == 256) else \
                  ("c4, one GPU's share of the global batch (extra, not the benched config)" if cfg.name == "ViT-B/16" else
                   "c2's step at another model / batch (extra, not the benched config)")
            wl = (f"BASELINE config {tag}: {cfg.name} distill step, bs={B}/GPU, 224x224 + {T}-token synthetic pairs, "
                  f"contrastive+cosine loss, fp32, vision trainable / text frozen (north_star regime), ")
        else:
            wl = (f"BASELINE config {args.workload} (extra, not the benched config): {cfg.name} student "
                  f"{args.student_precision} + meta-teacher "
                  f"in the step ({args.regions} region crops/img through a frozen {tcfg.name} tower with "
                  f"{args.tower_precision} GEMM inputs, token-level text, cross-modal attention + aggregation"
                  + (f", {tcfg.projection_dim}->{cfg.projection_dim} teacher bridge" if module.teacher_bridge is not None
                     else "") + f"), bs={B}/GPU, ")
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
            "dtype": "bf16" if args.student_precision == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": wl + ("fwd+bwd only" if opt is None else "fwd+bwd + clip-norm 0.5 + AdamW")
                                   + (", fwd+bwd replayed from a HIP graph" if graphed is not None else ""),
                       "execution": exec_note,
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "loss": last_loss},
            "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                         "achieved": None if achieved is None else round(achieved, 2),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": None if achieved is None else round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "launches_per_step": glaunches // n_ev,
                         "gemm_ms_per_step": round(gms / n_ev, 3),
                         "gemm_flops_per_step": gflops / n_ev,
                         "event_sampled_steps": n_ev if graphed is None else 0},
            # whole step priced two ways: `frac` with the ALGORITHMIC flops of the reference's arithmetic (SURVEY §8d:
            # 3 F_vis + F_txt per image), `frac_executed` with the GEMM flops actually launched — lower because the last
            # vision layer is evaluated on the CLS rows only, which is exact ("dead-row elimination")
            "roofline_step": {"bound": "mfma", "achieved": round(step_flops / (ms_per_step * 1e-3) / 1e12, 2),
                              "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              "frac_executed": None if not gms else round(
                                  gflops / n_ev / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              "flops_per_image": step_flops_per_image(cfg, T)},
        }
        line.update(extra)
        ck = clock.summary()
        if ck is not None:
            # context for the rooflines: the same fractions against the peak at the clock the card actually ran at
            line["clock"] = ck
            if line["roofline"]["frac"] is not None:
                line["roofline"]["frac_at_measured_clock"] = round(line["roofline"]["frac"] * ck["nominal_mhz"] / ck["sclk_mhz"], 4)
        g16 = timer.summary(bf16=True) if not args.no_gemm_events else (0.0, 0.0, 0)
        if g16[2] and graphed is None:
            # bf16 GEMM family (frozen teacher towers, bf16 student): its own roofline against the dense bf16 MFMA peak
            tr16 = None
            pmc16 = os.path.join(REPO, "profiles", "gemm_bf16_traffic.json")
            if os.path.exists(pmc16) and args.workload == "c3":        # the committed counter passes are over the c3 step
                tr16 = json.load(open(pmc16))
            line["roofline_bf16"] = {"bound": "mfma", "kernel": "gemm_bf16_pp_kernel (v_mfma_f32_16x16x32_bf16) / gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16)",
                                     "traffic": None if tr16 is None else tr16["hbm_bytes_per_launch"],
                                     "traffic_source": None if tr16 is None else "profiles/gemm_bf16_traffic.json (committed; NOT re-measured in this run)",
                                     "achieved": round(g16[0] / (g16[1] * 1e-3) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                     "frac": round(g16[0] / (g16[1] * 1e-3) / 1e12 / 2500.0, 4),
                                     "launches_per_step": g16[2] // n_ev,
                                     "gemm_ms_per_step": round(g16[1] / n_ev, 3),
                                     "gemm_flops_per_step": g16[0] / n_ev}
            if ck is not None:
                line["roofline_bf16"]["frac_at_measured_clock"] = round(line["roofline_bf16"]["frac"] * ck["nominal_mhz"] / ck["sclk_mhz"], 4)
        if args.student_precision == "bf16" and args.workload == "c2":
            line["config"]["workload"] = line["config"]["workload"].replace(
                "fp32, vision", "STUDENT VISION TOWER IN bf16 (not the benched precision), vision")
        if meta:
            # mixed-precision floor: student flops at the fp32 matrix peak + frozen-teacher flops at the peak of the
            # precision they were multiplied in (bf16 dense 2500 TF/s, MI355X_MICROARCH.md)
            tpeak = 2500.0 if args.tower_precision == "bf16" else PEAK_F32_MFMA_TFLOPS
            speak = 2500.0 if args.student_precision == "bf16" else PEAK_F32_MFMA_TFLOPS
            t_min_ms = (step_flops / speak + teacher_flops / tpeak) / 1e9
            line["roofline_step"] = {"bound": "mfma", "unit": "ms", "floor_ms": round(t_min_ms, 3),
                                     "frac": round(t_min_ms / ms_per_step, 4),
                                     "student_flops_per_image": step_flops / B, "teacher_flops_per_image": teacher_flops / B,
                                     "teacher_peak_tflops": tpeak, "student_peak_tflops": speak}
        if comm is not None:
            line["comm"] = dict(comm, backend=backend, ranks=dist.get_world_size(group),
                                embedding_all_gather_bytes_per_rank=2 * B * cfg.projection_dim * 4)
        if world == 1 and not meta and not args.no_extra_legs and args.student_precision == "fp32" \
                and cfg.name == "ViT-B/32" and B == 256 and not args.no_gemm_events:
            # the default run also times the other BASELINE configs for a few steps each (outside the contract's timed region):
            # c3 as quoted, and one GPU's share of the 8-GPU configs c4 and c5
            line["extra_configs"] = {}
            for name, which, st_, wu_ in (("c3_bf16", "c3", 5, 2), ("c4_per_gpu", "c4", 3, 1), ("c5_per_gpu_bf16", "c5", 3, 1)):
                if os.environ.get("DCLIP_BENCH_EXTRA", "c3,c4,c5").find(which) < 0:
                    continue
                try:
                    line["extra_configs"][name] = extra_config(dev, timer, which, st_, wu_)
                except Exception as exc:          # informative only: never fail the bench line on it
                    line["extra_configs"][name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            secs = float(os.environ.get("DCLIP_BENCH_CPU_SECONDS", "8"))
            line["cpu_baseline"] = cpu_baseline(secs, "north_star")          # same regime as `value`
            line["cpu_baseline_c1"] = cpu_baseline(secs, "as_written")       # BASELINE config c1 as the reference runs it
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
