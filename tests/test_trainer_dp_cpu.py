"""World-size-2 `gloo` test of data parallelism behind `lightning_lite.Trainer.fit` (CPU, host logic only).

A toy LightningLikeModule stands in for the student: a linear "vision tower" written as an autograd.Function that speaks
the same data-parallel protocol as functional.VisionTowerFn (asks `_GRAD_ALLOC` where to write its weight gradient, hands
it to `_GRAD_READY_HOOK`, returns None to autograd when the reducer took it), a plain linear "text tower" whose gradient
reaches the reducer through `.grad`, and the real loss glue (functional.* on tests/cpu_ops_shim.py, dist.*).

Property: N ranks running `fit()` with accumulate_grad_batches=2 for 3 optimizer steps end with the
parameters of ONE process running `fit()` on the concatenated batches; the all-reduce is launched from the hooks on the
boundary micro-batch only; rank 0 alone writes the checkpoint and its name carries the GLOBAL loss."""
import argparse
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

D_IN, P = 12, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_module(group=None):
    import tests.cpu_ops_shim as shim
    from dclip_amd import functional
    from dclip_amd.lightning_lite import LightningLikeModule
    functional.ops = shim

    class ToyTowerFn(torch.autograd.Function):
        """y = x W^T with the tower-side half of the GradSync protocol."""

        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x)
            ctx.w = w
            return x @ w.detach().t()

        @staticmethod
        def backward(ctx, dy):
            (x,) = ctx.saved_tensors
            alloc, hook = functional._GRAD_ALLOC, functional._GRAD_READY_HOOK
            buf = alloc(ctx.w, tuple(ctx.w.shape)) if alloc is not None else None
            dw = dy.t() @ x
            if buf is not None:
                buf.copy_(dw)
                dw = buf
            if hook is not None and hook([(ctx.w, dw)]):
                return None, None
            return None, dw

    class Toy(LightningLikeModule):
        def __init__(self):
            super().__init__()
            self.unused = torch.nn.Parameter(torch.ones(3))       # never receives a gradient: must not break the reducer
            g = torch.Generator().manual_seed(11)
            self.img = torch.nn.Parameter(torch.randn(P, D_IN, generator=g) * 0.3)
            self.txt = torch.nn.Linear(D_IN, P, bias=True)
            with torch.no_grad():
                self.txt.weight.copy_(torch.randn(P, D_IN, generator=g) * 0.3)
                self.txt.bias.zero_()
            self.process_group = group
            self.hparams = argparse.Namespace(learning_rate=0.05)
            self.hook_calls = 0

        def training_step(self, batch, batch_idx=0):
            from dclip_amd import dist as ddist
            i = ToyTowerFn.apply(batch["x"], self.img)
            t = self.txt(batch["y"])
            l_img = functional.cosine_distillation_loss(i, batch["ti"])
            l_txt = functional.cosine_distillation_loss(t, batch["tt"])
            l_con = functional.contrastive_loss(i, t, 0.05, self.process_group)
            self.last_losses = {"loss_image": l_img.detach(), "loss_text": l_txt.detach(), "loss_contrastive": l_con.detach()}
            if self.process_group is None:
                loss = l_img + l_txt + l_con
            else:
                loss = ddist.local_loss_for_backward(l_img, l_txt, l_con, dist.get_world_size(self.process_group))
            self.log("train_loss", loss.detach())
            return loss

        def configure_optimizers(self):
            # plain SGD on purpose: Adam and norm clipping are invariant to the SCALE of the gradient, and the scale
            # (1 / accumulate, mean over the global batch, SUM over ranks) is exactly what this test must pin
            return [torch.optim.SGD([p for p in self.parameters() if p.requires_grad], lr=0.05)], []

    return Toy()


def _batches(n, b):
    g = torch.Generator().manual_seed(5)
    return [{"x": torch.randn(b, D_IN, generator=g), "y": torch.randn(b, D_IN, generator=g),
             "ti": torch.randn(b, P, generator=g), "tt": torch.randn(b, P, generator=g)} for _ in range(n)]


def _none_grad_case(rank):
    """A parameter whose gradient exists on ONE rank only: after finish() every rank holds the same reduced .grad
    (otherwise the rank without a gradient skips the update and the replicas diverge)."""
    from dclip_amd import dist as ddist
    p = torch.nn.Parameter(torch.zeros(5))
    if rank == 0:
        p.grad = torch.full((5,), 3.0)
    sync = ddist.GradSync([p], dist.group.WORLD)
    sync.finish()
    return p.grad is not None and bool(torch.all(p.grad == 3.0))


def _worker(rank, world, port, ckpt_dir, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dclip_amd import dist as ddist
    from dclip_amd.lightning_lite import Trainer
    launches = {"hooked": 0, "late": 0}
    orig_launch = ddist.GradSync._launch

    def counting_launch(self, b):
        from dclip_amd import functional
        launches["hooked" if functional._GRAD_READY_HOOK is not None else "late"] += int(not self._launched[b])
        return orig_launch(self, b)

    ddist.GradSync._launch = counting_launch
    mod = _make_module()
    tr = Trainer(max_epochs=1, gradient_clip_val=None, accumulate_grad_batches=2, checkpoint_dir=ckpt_dir,
                 devices=world, dist_backend="gloo", bucket_mb=0.0001)          # tiny buckets: every tensor its own
    # 6 full rank groups (= 3 optimizer steps at accumulate 2) + one batch too many, dropped on every rank
    tr.fit(mod, _batches(6 * world + 1, 4))
    out[rank] = dict(img=mod.img.detach().clone(), tw=mod.txt.weight.detach().clone(), tb=mod.txt.bias.detach().clone(),
                     step=mod.global_step, saved=list(tr.saved), files=sorted(os.listdir(ckpt_dir)),
                     train_loss=mod.logged("train_loss"), launches=dict(launches),
                     unused_grad_is_view=_none_grad_case(rank))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_n_rank_fit_equals_single_process_fit_on_concatenated_batches(tmp_path, world):
    from dclip_amd.lightning_lite import Trainer
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path / "dp"), out), nprocs=world, join=True)

    per_gpu = _batches(6 * world + 1, 4)
    cat = [{k: torch.cat([per_gpu[world * j + r][k] for r in range(world)]) for k in per_gpu[0]} for j in range(6)]
    ref = _make_module()
    tr = Trainer(max_epochs=1, gradient_clip_val=None, accumulate_grad_batches=2, checkpoint_dir=str(tmp_path / "one"))
    tr.fit(ref, cat)
    assert ref.global_step == 3
    for r in range(world):
        o = out[r]
        assert o["step"] == 3
        for got, want in ((o["img"], ref.img), (o["tw"], ref.txt.weight), (o["tb"], ref.txt.bias)):
            assert torch.allclose(got, want.detach(), rtol=1e-5, atol=1e-6), float((got - want.detach()).abs().max())
        # the logged loss / checkpoint name is the single-process value of the last micro-batch
        assert abs(o["train_loss"] - ref.logged("train_loss")) < 1e-5 * abs(ref.logged("train_loss"))
        # buckets (reverse registration order, 26-element threshold): [txt.bias, txt.weight] [img] [unused].  Hooks are live
        # on the boundary micro-batch only: 3 optimizer steps => 3 hook-launched buckets (the toy tower's weight); the
        # text tower (through .grad) and the gradient-less parameter are launched from finish()
        assert o["launches"]["hooked"] == 3 and o["launches"]["late"] == 3 * 2, o["launches"]
        assert o["unused_grad_is_view"]
    for r in range(1, world):
        assert torch.equal(out[0]["img"], out[r]["img"])                  # replicas stay bit-identical
        assert out[r]["saved"] == []                                      # rank 0 alone writes
    assert len(out[0]["saved"]) == 1
    assert out[0]["files"] == sorted(os.listdir(tmp_path / "one"))        # same file name (global train_loss)


def test_devices_without_rendezvous_says_how_to_launch():
    from dclip_amd.lightning_lite import Trainer
    import pytest
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK") if k in os.environ}
    try:
        with pytest.raises(RuntimeError, match="torch.distributed.run"):
            Trainer(devices=4).fit(_make_module(), [])
    finally:
        os.environ.update(env)


def test_shard_batches_drops_the_incomplete_trailing_group():
    from dclip_amd.dist import shard_batches
    assert list(shard_batches(range(7), 0, 3)) == [0, 3] and list(shard_batches(range(7), 2, 3)) == [2, 5]
    assert list(shard_batches([], 0, 2)) == []


def test_json_pair_dataset_rank_shards_cover_the_single_process_order(tmp_path):
    import json
    from dclip_amd.train_contrastive_teacher import JsonPairDataset
    jf = tmp_path / "t.json"
    jf.write_text(json.dumps([{"image_path": f"i{i}.png", "captions": [f"c{i}"]} for i in range(23)]))
    one = [b[2] for b in JsonPairDataset(str(jf), 4, True)]
    shards = [[b[2] for b in JsonPairDataset(str(jf), 4, True, rank=k, world=3)] for k in range(3)]
    assert [len(s) for s in shards] == [2, 2, 2] and len(JsonPairDataset(str(jf), 4, True, rank=1, world=3)) == 2
    for j in range(2):                              # group j of the N-rank run = batches 3j .. 3j+2 of the single-process epoch
        for k in range(3):
            assert shards[k][j] == one[3 * j + k]


def _loader_worker(rank, world, port, jf, cdir, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dclip_amd import config as dcfg, synth
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    cfg = dcfg.tiny()
    student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7))
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=student)
    hp = argparse.Namespace(learning_rate=1e-3, warmup_steps=0, total_steps=10, train_batch_size=2, eval_batch_size=2,
                            train_file=jf, val_file=jf, cache_dir=cdir, train_cache_filename="train_precache.pkl",
                            val_cache_filename="train_precache.pkl")
    mod = CLIPImageDistillation(hp, student, None, teacher=teacher, process_group=dist.group.WORLD)
    tl, vl = mod.train_dataloader(), mod.val_dataloader()
    out[rank] = dict(sharded=(tl.rank_sharded, vl.rank_sharded), train=[p for b in tl for p in b[2]], val=[p for b in vl for p in b[2]])
    dist.barrier()
    dist.destroy_process_group()


def test_module_loaders_are_rank_sharded_under_a_process_group(tmp_path):
    """`train_dataloader()` / `val_dataloader()` of a module that has a process group: every rank LOADS only its share of
    the items (DistributedSampler) and says so, so that `Trainer.fit` does not deal the batches out a second time."""
    from tests.test_data_cpu import _materialise
    jf, cdir, _full = _materialise(tmp_path)
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_loader_worker, args=(world, _free_port(), jf, cdir, out), nprocs=world, join=True)
    assert out[0]["sharded"] == (True, True) and out[1]["sharded"] == (True, True)
    for split in ("train", "val"):
        a, b = out[0][split], out[1][split]
        assert a and b and len(a) == len(b) and not (set(a) & set(b))            # equal shares, nothing loaded twice
