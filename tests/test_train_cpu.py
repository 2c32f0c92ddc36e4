"""CPU: host logic of the training path - losses, data transforms, and the flat-buffer gradient all-reduce on two gloo ranks
reproducing the single-process batch gradient (the N>1 path of BASELINE config 3)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_charbonnier_losses():
    from fcvsr_amd.train.loss import charbonnier_loss, charbonnier_loss_mmedit
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(2, 1, 8, 8, generator=g), torch.rand(2, 1, 8, 8, generator=g)
    d = x - y
    assert torch.allclose(charbonnier_loss(x, y), torch.sqrt(d * d + 1e-4).sum())             # opt/loss.py:20-31
    dm = d.reshape(2, -1).mean(1, keepdim=True)
    assert torch.allclose(charbonnier_loss(x, y, mean_res=True), torch.sqrt(dm * dm + 1e-4).sum())
    assert torch.allclose(charbonnier_loss_mmedit(x, y), torch.sqrt(d * d + 1e-12).mean())      # pixelwise_loss.py:41-51
    assert charbonnier_loss_mmedit(x, y, reduction="none").shape == x.shape
    with pytest.raises(ValueError):
        charbonnier_loss_mmedit(x, y, reduction="max")


def test_data_transforms_follow_the_reference_loader():
    from fcvsr_amd.train.step import augment, random_crop, to_tensor
    rs = np.random.RandomState(0)
    s = {"lr_imgs": rs.randint(0, 256, (7, 40, 48)).astype(np.uint8), "hr_imgs": rs.randint(0, 256, (1, 160, 192)).astype(np.uint8)}
    c = random_crop(s, 16, np.random.RandomState(1))
    assert c["lr_imgs"].shape == (7, 16, 16) and c["hr_imgs"].shape == (1, 64, 64)

    class R:                                      # hflip, vflip, rot90 all taken
        def random(self):
            return 0.0

    a = augment(c, R())
    assert np.array_equal(a["lr_imgs"], c["lr_imgs"][:, ::-1, ::-1].transpose(0, 2, 1))
    assert np.array_equal(a["hr_imgs"], c["hr_imgs"][:, ::-1, ::-1].transpose(0, 2, 1))
    t = to_tensor(a)
    assert t["lr_imgs"].shape == (1, 7, 16, 16) and float(t["lr_imgs"].max()) <= 1.0


def test_trainable_parameters_skip_the_never_called_conv_and_count_aliases_once():
    from fcvsr_amd.arch.CVSR_freq import GShiftNet_S
    from fcvsr_amd.train.step import trainable_parameters
    m = GShiftNet_S(n_features=32, ACNum=2, Freq_Inv=2, SCGroupN=1)
    named = trainable_parameters(m)
    names = [n for n, _ in named]
    assert not any(".DivEnh_block." in n and ".Conv." in n for n in names)
    assert len({id(p) for _, p in named}) == len(named)
    total = sum(p.numel() for _, p in named)
    unused = sum(p.numel() for n, p in m.named_parameters() if ".DivEnh_block." in n and ".Conv." in n)
    assert total + unused == sum(p.numel() for p in m.parameters())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Conv2d(7, 8, 3, padding=1), torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(8, 1, 3, padding=1))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcvsr_amd.harness.sharding import shard
    from fcvsr_amd.train.loss import charbonnier_loss
    from fcvsr_amd.train.step import FlatGradAllReduce
    net = _net()                                           # same seed on every rank = replicated weights
    g = torch.Generator().manual_seed(11)
    x, y = torch.rand(6, 7, 12, 10, generator=g), torch.rand(6, 1, 12, 10, generator=g)
    lo, hi = shard(6, rank, world)                         # clip data parallel: this rank's windows
    loss = charbonnier_loss(net(x[lo:hi]), y[lo:hi])
    loss.backward()
    ar = FlatGradAllReduce(list(net.parameters()), "sum")
    flat = ar()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.step()
    q.put((rank, flat.clone().numpy(), [p.detach().clone().numpy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_allreduce_reproduces_the_batch_gradient_and_update():
    from fcvsr_amd.train.loss import charbonnier_loss
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = _net()
    g = torch.Generator().manual_seed(11)
    x, y = torch.rand(6, 7, 12, 10, generator=g), torch.rand(6, 1, 12, 10, generator=g)
    charbonnier_loss(net(x), y).backward()                  # the SUM loss over the whole batch in one process
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    torch.optim.Adam(net.parameters(), lr=1e-3).step()
    for rank, flat, params in res:
        assert np.abs(flat - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        for a, p in zip(params, net.parameters()):
            assert np.abs(a - p.detach().numpy()).max() <= 1e-6
    assert np.array_equal(res[0][1], res[1][1])              # identical buffers on both ranks after the collective
