"""Host logic of the cross-step teacher pipelining in `lightning_lite.Trainer.fit` (CPU): a module that offers
`prefetch_teacher` is handed the NEXT batch after `training_step(current)` and before the backward of the current one,
never after the last batch of an epoch, with and without gradient accumulation; modules without the method train as
before.  (The arithmetic — identical weights with and without it — is tests/test_teacher_prefetch_gpu.py.)"""
import torch

from dclip_amd.lightning_lite import LightningLikeModule, Trainer, _with_lookahead, _END


class _Recorder(LightningLikeModule):
    def __init__(self, with_prefetch=True):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(3))
        self.log_ = []
        if with_prefetch:
            self.prefetch_teacher = self._prefetch

    def _prefetch(self, batch):
        self.log_.append(("prefetch", int(batch["id"])))
        return True

    def training_step(self, batch, batch_idx=0):
        self.log_.append(("step", int(batch["id"])))
        loss = (self.w * batch["x"]).sum()
        loss.register_hook(lambda g, i=int(batch["id"]): self.log_.append(("backward", i)))
        return loss

    def configure_optimizers(self):
        return [torch.optim.SGD([self.w], lr=0.1)], []

    def train_dataloader(self):
        return [{"id": torch.tensor(i), "x": torch.full((3,), float(i + 1))} for i in range(4)]


def test_with_lookahead_yields_the_next_item():
    assert [(i, a, b if b is not _END else None) for i, a, b in _with_lookahead("abc")] == [(0, "a", "b"), (1, "b", "c"), (2, "c", None)]
    assert list(_with_lookahead([])) == []
    assert [(i, a, b is _END) for i, a, b in _with_lookahead(iter([7]))] == [(0, 7, True)]


def test_fit_prefetches_the_next_batch_between_forward_and_backward():
    for accum in (1, 2, 3):
        m = _Recorder()
        Trainer(max_epochs=2, accumulate_grad_batches=accum, gradient_clip_val=None).fit(m)
        per_epoch = []
        for i in range(4):
            per_epoch.append(("step", i))
            if i < 3:
                per_epoch.append(("prefetch", i + 1))
            per_epoch.append(("backward", i))
        assert m.log_ == per_epoch * 2, (accum, m.log_)


def test_modules_without_prefetch_train_as_before():
    a, b = _Recorder(True), _Recorder(False)
    Trainer(max_epochs=1, accumulate_grad_batches=2, gradient_clip_val=None).fit(a)
    Trainer(max_epochs=1, accumulate_grad_batches=2, gradient_clip_val=None).fit(b)
    assert torch.equal(a.w.detach(), b.w.detach())
    assert [e for e in a.log_ if e[0] != "prefetch"] == b.log_
