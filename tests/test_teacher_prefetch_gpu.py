"""Cross-step pipelining of the frozen meta-teacher (CLIPImageDistillation.prefetch_teacher): starting batch n+1's
teacher during batch n's backward must change NOTHING in the arithmetic — the same targets, losses, gradients and
trained weights as the step that runs its teacher itself (reference step: training/CLIP_image_distillation.py:245-313,
teacher call :259-262; the teacher is frozen there too, its targets do not depend on the student's update)."""
import argparse

import pytest
import torch

from dclip_amd import config as dcfg, synth

pytestmark = pytest.mark.gpu


def _module(precision="fp32", seed=0, own_teacher_clip=True):
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=seed, gain=3.0), device=dev)
    E = cfg.projection_dim
    tclip = student
    if own_teacher_clip:          # the reference's arrangement: the teacher owns a frozen CLIP (training/CLIP_image_distillation.py:100-110)
        tclip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=seed + 7, gain=3.0), device=dev)
        for p in tclip.parameters():
            p.requires_grad_(False)
    teacher = PatchTextAggregation(embed_dim=E, num_heads=max(E // 64, 1), clip_model=tclip)
    teacher.load_state_dict({f"cross_modal_attention.{k}": v
                             for k, v in synth.synth_cross_modal_state_dict(E, seed=31).items()})
    hp = argparse.Namespace(learning_rate=1e-3, warmup_steps=0, total_steps=100, train_batch_size=4, eval_batch_size=4)
    mod = CLIPImageDistillation(hp, student, None, teacher=teacher.to(dev), freeze_mode="north_star",
                                student_precision=precision).to(dev)
    return mod, cfg


def _batches(cfg, n, B=4, R=3, device=None):
    out = []
    for k in range(n):
        b = {"pixel_values": synth.synth_pixel_values(B, cfg.vision, seed=10 + k),
             "input_ids": synth.synth_input_ids(B, cfg.text, seed=20 + k, ragged=True, min_len=5),
             "regions": synth.synth_regions(B, R, cfg.vision, seed=30 + k),
             "region_counts": torch.tensor([R, 1, 2, R][:B])}
        if device is not None:
            b = {k_: (v.to(device) if k_ != "region_counts" else v) for k_, v in b.items()}
        out.append(b)
    return out


def _grads(mod):
    return {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_prefetched_teacher_gives_the_same_step(precision):
    mod, cfg = _module(precision)
    a, b = _batches(cfg, 2, device=mod.device)
    # plain: b's teacher runs inside b's step
    mod.training_step(a).backward()
    mod.zero_grad(set_to_none=True)
    want = mod.training_step(b)
    want.backward()
    want_g = _grads(mod)
    mod.zero_grad(set_to_none=True)
    # pipelined: b's teacher is started between a's forward and a's backward
    loss_a = mod.training_step(a)
    assert mod.prefetch_teacher(b) is True
    assert mod._prefetched is not None
    loss_a.backward()
    mod.zero_grad(set_to_none=True)
    got = mod.training_step(b)
    assert mod._prefetched is None                      # consumed
    got.backward()
    got_g = _grads(mod)
    torch.cuda.synchronize()
    assert float(got.detach()) == float(want.detach())
    assert got_g.keys() == want_g.keys() and len(got_g) > 10
    for n in want_g:
        assert torch.equal(got_g[n], want_g[n]), n


def test_prefetch_of_another_batch_is_dropped_and_switches_hold(monkeypatch):
    mod, cfg = _module()
    a, b, c = _batches(cfg, 3, device=mod.device)
    want = float(mod.training_step(c).detach())
    mod.training_step(a)
    assert mod.prefetch_teacher(b)
    got = float(mod.training_step(c).detach())          # the loop skipped b: c's own teacher runs, b's target is released
    assert got == want and mod._prefetched is None
    # batches that bring their teacher target, CPU-side switches: nothing is started
    assert mod.prefetch_teacher({**a, "teacher_image_emb": torch.zeros(4, cfg.projection_dim)}) is False
    monkeypatch.setenv("DCLIP_TEACHER_PREFETCH", "0")
    assert mod.prefetch_teacher(b) is False
    monkeypatch.delenv("DCLIP_TEACHER_PREFETCH")
    monkeypatch.setenv("DCLIP_TEACHER_STREAM", "0")
    assert mod.prefetch_teacher(b) is False
    assert mod.prefetch_teacher((a["pixel_values"], ["x"] * 4, ["p"] * 4, [[]] * 4)) is False
    monkeypatch.delenv("DCLIP_TEACHER_STREAM")
    assert mod.prefetch_teacher(b) is True
    mod._drop_prefetched()
    # a teacher built on the student's own (trained) vision tower: its targets for the next batch depend on this update
    shared, _ = _module(own_teacher_clip=False)
    shared.training_step(a)
    assert shared.prefetch_teacher(b) is False and shared._prefetched is None


def test_trainer_with_and_without_the_pipelined_teacher_trains_the_same_weights(monkeypatch):
    from dclip_amd.lightning_lite import Trainer

    def run(prefetch_on):
        monkeypatch.setenv("DCLIP_TEACHER_PREFETCH", "1" if prefetch_on else "0")
        mod, cfg = _module(seed=3)
        batches = _batches(cfg, 5, device=mod.device)
        start = {n: p.detach().clone() for n, p in mod.named_parameters() if p.requires_grad}
        calls = []
        real = mod.prefetch_teacher
        mod.prefetch_teacher = lambda b_: calls.append(real(b_)) or calls[-1]
        mod.train_dataloader = lambda: batches
        mod.val_dataloader = lambda: None
        Trainer(max_epochs=2, accelerator="gpu", devices=1, gradient_clip_val=0.5, accumulate_grad_batches=2,
                enable_progress_bar=False, logger=False, enable_checkpointing=False).fit(mod)
        torch.cuda.synchronize()
        return {n: p.detach().clone() for n, p in mod.named_parameters() if p.requires_grad}, calls, start

    want, calls0, start = run(False)
    got, calls1, _ = run(True)
    assert calls0 == [False] * 8 and calls1 == [True] * 8          # 2 epochs x (5 batches - the last one)
    for n in want:
        assert torch.equal(got[n], want[n]), n
    assert sum(int(not torch.equal(want[n], start[n])) for n in want) > 10      # the run did train


def _dp_worker(rank, world, port, prefetch_on, out):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      DCLIP_TEACHER_PREFETCH="1" if prefetch_on else "0")
    import torch.distributed as dist
    from dclip_amd.lightning_lite import Trainer
    mod, cfg = _module(seed=5)
    batches = _batches(cfg, 2 * 4, device=mod.device)              # 4 batches per rank
    calls = []
    real = mod.prefetch_teacher
    mod.prefetch_teacher = lambda b_: calls.append(real(b_)) or calls[-1]
    tr = Trainer(max_epochs=1, gradient_clip_val=0.5, accumulate_grad_batches=2, devices=world, dist_backend="gloo", bucket_mb=0.05)
    tr.fit(mod, batches)
    torch.cuda.synchronize()
    out[(prefetch_on, rank)] = dict(params={n: p.detach().float().cpu().clone() for n, p in mod.named_parameters() if p.requires_grad},
                                    calls=list(calls), stats=tr.grad_sync.stats())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_the_pipelined_teacher_train_the_same_weights():
    """Data parallel + pipelined teacher: two gloo ranks on the one GPU, each starting ITS next batch's teacher while the
    all-reduce of the current gradients is launched from inside the backward — the trained weights equal those of the same
    two ranks with the teacher run inside the step, bit for bit, and the replicas agree."""
    import socket
    import torch.multiprocessing as mp
    res = {}
    for on in (False, True):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        out = mp.Manager().dict()
        mp.spawn(_dp_worker, args=(2, port, on, out), nprocs=2, join=True)
        res.update(dict(out))
    for r in (0, 1):
        assert res[(False, r)]["calls"] == [False] * 3 and res[(True, r)]["calls"] == [True] * 3       # 4 batches per rank
        assert res[(True, r)]["stats"]["grad_tensors_copied_per_step"] == 0
        for n, w in res[(False, r)]["params"].items():
            assert torch.equal(res[(True, r)]["params"][n], w), (r, n)
    for n, w in res[(True, 0)]["params"].items():
        assert torch.equal(res[(True, 1)]["params"][n], w), n
