"""World-size-2 `gloo` tests of the data-parallel path (CPU): the global-negatives loss protocol and the gradient
all-reduce.  The kernels are replaced by tests/cpu_ops_shim.py — this checks the HOST logic; kernel parity is the
`-m gpu` suite's job.  Property F5: N-rank loss / gradients == single process on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dclip_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, P, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tests.cpu_ops_shim as shim
    from dclip_amd import functional, dist as ddist
    functional.ops = shim                                   # host logic under test; kernels are covered on the GPU
    g = torch.Generator().manual_seed(7)
    img = torch.randn(B, P, generator=g) * 1.3
    txt = torch.randn(B, P, generator=g) + 0.1 * img
    t_img, t_txt = torch.randn(B, P, generator=g), torch.randn(B, P, generator=g)
    Bl = B // world
    sl = slice(rank * Bl, (rank + 1) * Bl)
    i_l = img[sl].clone().requires_grad_(True)
    t_l = txt[sl].clone().requires_grad_(True)
    l_img = functional.cosine_distillation_loss(i_l, t_img[sl])
    l_txt = functional.cosine_distillation_loss(t_l, t_txt[sl])
    l_con = functional.contrastive_loss(i_l, t_l, 0.05, dist.group.WORLD)
    share = ddist.local_loss_for_backward(l_img, l_txt, l_con, world)
    share.backward()
    total = ddist.global_loss_value(l_img, l_txt, l_con, dist.group.WORLD)
    # a "parameter" shared by all ranks: its gradient must be SUM-reduced
    w = torch.nn.Parameter(torch.ones(P))
    w.grad = (i_l.grad.sum(0) + t_l.grad.sum(0)).clone()
    small = torch.nn.Parameter(torch.zeros(3))
    small.grad = torch.full((3,), float(rank + 1))
    sync = ddist.GradSync([w, small], dist.group.WORLD, bucket_mb=0.0001)      # force several buckets
    sync.reduce()
    out[rank] = dict(total=float(total), share=float(share.detach()), gi=i_l.grad.clone(), gt=t_l.grad.clone(),
                     w=w.grad.clone(), small=small.grad.clone())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 16), (2, 6)])
def test_global_negatives_loss_is_sharding_invariant(world, B):
    P = 64
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B, P, out), nprocs=world, join=True)
    g = torch.Generator().manual_seed(7)
    img = (torch.randn(B, P, generator=g) * 1.3).requires_grad_(True)
    txt0 = torch.randn(B, P, generator=g)
    txt = (txt0 + 0.1 * img.detach()).requires_grad_(True)
    t_img, t_txt = torch.randn(B, P, generator=g), torch.randn(B, P, generator=g)
    ref = O.cosine_distillation_loss(img, t_img) + O.cosine_distillation_loss(txt, t_txt) + O.contrastive_loss(img, txt)
    ref.backward()
    Bl = B // world
    assert abs(sum(out[r]["share"] for r in range(world)) - float(ref)) < 1e-5 * abs(float(ref))
    for r in range(world):
        assert abs(out[r]["total"] - float(ref)) < 1e-5 * abs(float(ref))
        sl = slice(r * Bl, (r + 1) * Bl)
        assert torch.allclose(out[r]["gi"], img.grad[sl], rtol=1e-4, atol=1e-7)
        assert torch.allclose(out[r]["gt"], txt.grad[sl], rtol=1e-4, atol=1e-7)
        assert torch.allclose(out[r]["w"], img.grad.sum(0) + txt.grad.sum(0), rtol=1e-4, atol=1e-6)
        assert torch.equal(out[r]["small"], torch.full((3,), float(sum(range(1, world + 1)))))


def test_single_process_path_matches_oracle():
    import tests.cpu_ops_shim as shim
    from dclip_amd import functional
    functional.ops = shim
    g = torch.Generator().manual_seed(3)
    img = torch.randn(9, 32, generator=g).requires_grad_(True)
    txt = torch.randn(9, 32, generator=g).requires_grad_(True)
    loss = functional.contrastive_loss(img, txt, 0.05, None)
    loss.backward()
    i2, t2 = img.detach().clone().requires_grad_(True), txt.detach().clone().requires_grad_(True)
    ref = O.contrastive_loss(i2, t2)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    assert torch.allclose(img.grad, i2.grad, rtol=1e-4, atol=1e-7) and torch.allclose(txt.grad, t2.grad, rtol=1e-4, atol=1e-7)


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dclip_amd import dist as ddist
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (1000, 7, 30000, 64, 25000, 3, 9000)]
    sync = ddist.GradSync(ps, dist.group.WORLD, bucket_mb=0.05)
    ok = True
    for step in range(4):                       # the buckets are persistent: several backward / finish rounds
        pairs = []
        for i, p in enumerate(reversed(ps)):    # deliveries in backward order; one gradient is NOT produced in place
            buf = sync.grad_buffer(p, tuple(p.shape))
            if i == 2:
                g = torch.full_like(p, float(rank + 1 + step))
            else:
                buf.fill_(float(rank + 1 + step))
                g = buf
            pairs.append((p, g))
        for k in range(0, len(pairs) - 1, 2):
            assert sync.on_grads_ready(pairs[k:k + 2]) is True
        ps[0].grad = pairs[-1][1].clone()       # the last parameter arrives through .grad (no hook), as another tower's would
        sync.finish()
        want = float(sum(r + 1 + step for r in range(world)))
        ok = ok and all(bool(torch.all(p.grad == want)) for p in ps)
        ok = ok and all(p.grad.data_ptr() == sync.grad_buffer(p, tuple(p.shape)).data_ptr() for p in ps)
    stats = sync.stats()
    try:
        sync.on_grads_ready([(ps[1], torch.ones(7))])
        sync.on_grads_ready([(ps[1], torch.ones(7))])
        ok = False
    except RuntimeError:                        # a second delivery in one backward is refused, not dropped
        pass
    out[rank] = dict(ok=ok, stats=stats)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gradsync_persistent_buckets_over_several_steps(world):
    out = mp.Manager().dict()
    mp.spawn(_bucket_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert out[r]["ok"], r
        st = out[r]["stats"]
        assert st["grad_buckets_per_step"] >= 3 and st["grad_tensors_copied_per_step"] == 2.0      # the two not in place
        assert st["grad_tensors_written_in_place_per_step"] == 5.0
