"""Two data-parallel ranks on the ONE GPU of the test box (gloo rendezvous, both ranks on cuda:0): the real kernels,
the global-negatives exchange and GradSync together reproduce the single-process step on the concatenated batch."""
import argparse
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dclip_amd import config as dcfg, synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev, group, student_precision="fp32"):
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    cfg = dcfg.tiny()
    student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), device=dev)
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=student).to(dev)
    hp = argparse.Namespace(learning_rate=1e-3, warmup_steps=0, total_steps=10, train_batch_size=4, eval_batch_size=4)
    return cfg, CLIPImageDistillation(hp, student, None, teacher=teacher, freeze_mode="north_star",
                                      process_group=group, student_precision=student_precision).to(dev)


def _batch(cfg, B):
    return {"pixel_values": synth.synth_pixel_values(B, cfg.vision, seed=0),
            "input_ids": synth.synth_input_ids(B, cfg.text, seed=3, ragged=True),
            "teacher_image_emb": synth.synth_embeddings(B, cfg.projection_dim, seed=1),
            "teacher_text_emb": synth.synth_embeddings(B, cfg.projection_dim, seed=5)}


def _worker(rank, world, port, B, out, overlap, precision="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dclip_amd import dist as ddist
    dev = torch.device("cuda:0")
    cfg, mod = _build(dev, dist.group.WORLD, precision)
    full = _batch(cfg, B)
    Bl = B // world
    shard = {k: v[rank * Bl:(rank + 1) * Bl].contiguous() for k, v in full.items()}
    from dclip_amd import functional
    trainable = [p for p in mod.parameters() if p.requires_grad]
    sync = ddist.GradSync(trainable, dist.group.WORLD, bucket_mb=0.05)
    if overlap:
        functional.set_grad_ready_hook(sync.on_grads_ready)      # all-reduce launched from inside the backward
        functional.set_grad_alloc(sync.grad_buffer)              # gradients written straight into the buckets
    share = mod.training_step(shard)
    share.backward()
    sync.finish()
    functional.set_grad_ready_hook(None)
    functional.set_grad_alloc(None)
    st = sync.stats()
    if overlap:      # every gradient of the hooked tower was produced in place: no packing copy
        assert st["grad_tensors_written_in_place_per_step"] >= 30 and st["grad_tensors_copied_per_step"] == 0, st
    tot = share.detach().clone()
    dist.all_reduce(tot)
    torch.cuda.synchronize()
    out[rank] = dict(loss=float(tot), g_proj=mod.student.visual_projection.weight.grad.cpu(),
                     g_qkv=mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.grad.cpu())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,precision", [(False, "fp32"), (True, "fp32"), (True, "bf16")])
def test_two_ranks_equal_single_process(overlap, precision):
    """precision "bf16": the bf16 student (BASELINE c5's quoted mode) under data parallelism — its split-K weight
    gradients, bias sums and LayerNorm gradients are written into the all-reduce buckets too (copied == 0, asserted in
    the worker) and the 2-rank step equals the single-process bf16 step on the concatenated batch."""
    B, world = 8, 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), B, out, overlap, precision), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    cfg, mod = _build(dev, None, precision)
    loss = mod.training_step(_batch(cfg, B))
    loss.backward()
    g_proj = mod.student.visual_projection.weight.grad.cpu()
    g_qkv = mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.grad.cpu()
    tol = 2e-5 if precision == "fp32" else 1e-4        # same bf16 roundings per row either way; fp32 sums in another order
    for r in range(world):
        assert abs(out[r]["loss"] - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
        assert float((out[r]["g_proj"] - g_proj).abs().max()) < tol * float(g_proj.abs().max())
        assert float((out[r]["g_qkv"] - g_qkv).abs().max()) < tol * float(g_qkv.abs().max())


def _rccl_worker(rank, world, port, B, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dclip_amd import dist as ddist, functional
    group = ddist.init_from_env("nccl")                    # RCCL over xGMI, one GPU per rank
    dev = torch.device("cuda", rank)
    cfg, mod = _build(dev, group)
    full = _batch(cfg, B)
    Bl = B // world
    shard = {k: v[rank * Bl:(rank + 1) * Bl].contiguous() for k, v in full.items()}
    trainable = [p for p in mod.parameters() if p.requires_grad]
    sync = ddist.GradSync(trainable, group, bucket_mb=0.05)
    functional.set_grad_ready_hook(sync.on_grads_ready)
    functional.set_grad_alloc(sync.grad_buffer)
    share = mod.training_step(shard)
    share.backward()
    sync.finish()
    functional.set_grad_ready_hook(None)
    functional.set_grad_alloc(None)
    tot = share.detach().clone()
    dist.all_reduce(tot, group=group)
    torch.cuda.synchronize()
    st = sync.stats()
    out[rank] = dict(loss=float(tot), g_proj=mod.student.visual_projection.weight.grad.cpu(),
                     g_qkv=mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.grad.cpu(),
                     buckets=st["grad_buckets_per_step"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs two GPUs (the driver's multi-GPU node)")
def test_two_ranks_over_rccl_equal_single_process():
    """backend "nccl" (= RCCL): embedding all-gather + LSE all-gather + bucketed gradient all-reduce launched from
    inside the backward, one GPU per rank; N-rank loss / gradients == single process on the concatenated batch."""
    B, world = 8, 2
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    cfg, mod = _build(dev, None)
    loss = mod.training_step(_batch(cfg, B))
    loss.backward()
    g_proj = mod.student.visual_projection.weight.grad.cpu()
    g_qkv = mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.grad.cpu()
    for r in range(world):
        assert abs(out[r]["loss"] - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
        assert float((out[r]["g_proj"] - g_proj).abs().max()) < 2e-5 * float(g_proj.abs().max())
        assert float((out[r]["g_qkv"] - g_qkv).abs().max()) < 2e-5 * float(g_qkv.abs().max())
        assert out[r]["buckets"] >= 2


def _rccl_one_rank_worker(rank, port, B, out):
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    from dclip_amd import dist as ddist
    group = dist.group.WORLD
    dev = torch.device("cuda:0")
    cfg, mod = _build(dev, group)
    batch = _batch(cfg, B)
    trainable = [p for p in mod.parameters() if p.requires_grad]
    sync = ddist.GradSync(trainable, group, bucket_mb=0.05, timing=True)
    losses = []
    for _ in range(2):                                   # twice: the persistent buckets are reused
        for p in trainable:
            p.grad = None
        share = mod.training_step(batch)
        with sync.hooks():
            share.backward()                             # all-reduces launched from the autograd thread, async
        sync.finish()
        losses.append(float(share.detach()))
    total = ddist.global_loss_value(mod.last_losses["loss_image"], mod.last_losses["loss_text"],
                                    mod.last_losses["loss_contrastive"], group)
    g_proj = mod.student.visual_projection.weight.grad.cpu()
    g_qkv = mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.grad.cpu()
    per_bucket = sync.bucket_allreduce_ms(repeats=2)      # (the calibration leg zero-fills the buckets: gradients read before)
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)             # what bench.py's timed() does with the elapsed time
    dist.barrier()
    torch.cuda.synchronize()
    st = sync.stats()
    out[0] = dict(losses=losses, total=float(total), maxed=float(t), buckets=len(per_bucket), stats=st,
                  g_proj=g_proj, g_qkv=g_qkv)
    dist.destroy_process_group()


def test_single_rank_rccl_group_runs_the_dp_protocol():
    """The test box has ONE GPU, so RCCL between ranks cannot run here — but every RCCL ENTRY POINT the N-rank step uses
    can: a one-rank `nccl` group takes the same calls (all_gather_into_tensor of the embeddings and of the LSE vectors,
    asynchronous bucket all-reduces launched from inside the backward on the autograd thread, work.wait() on the compute
    stream, barrier, the float64 MAX all-reduce of bench.py) through RCCL's streams and events.  With one rank every
    collective is the identity, so loss and gradients must equal the group-less step."""
    B = 8
    out = mp.Manager().dict()
    mp.spawn(_rccl_one_rank_worker, args=(_free_port(), B, out), nprocs=1, join=True)
    dev = torch.device("cuda:0")
    cfg, mod = _build(dev, None)
    loss = mod.training_step(_batch(cfg, B))
    loss.backward()
    o = out[0]
    assert o["losses"][0] == o["losses"][1] and abs(o["losses"][0] - float(loss.detach())) < 1e-6 * abs(float(loss.detach()))
    assert abs(o["total"] - float(loss.detach())) < 1e-6 * abs(float(loss.detach())) and o["maxed"] == 1.5
    assert torch.equal(o["g_proj"], mod.student.visual_projection.weight.grad.cpu())
    assert torch.equal(o["g_qkv"], mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.grad.cpu())
    assert o["buckets"] >= 2 and o["stats"]["grad_tensors_copied_per_step"] == 0
    assert o["stats"]["grad_allreduce_exposed_ms_per_step"] is not None
