"""Data-parallel plumbing (dp.py) with world_size 2 on the CPU over gloo: the flat-bucket gradient exchange,
parameter broadcast, batch sharding and BatchNorm running-stat averaging.  The kernels are not involved
(they need the GPU); what is checked is that N ranks on N shards reproduce the 1-rank gradient of the whole
batch and end up with bitwise-identical replicas."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import PKG


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3),
                               torch.nn.Linear(3, 3))      # the last layer is deliberately left unused


def _loss(net, x, y):
    h = net[2](net[1](net[0](x)))
    return ((h - y) ** 2).mean()


def _worker(rank, world, port, out_dir):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    DP = importlib.import_module(PKG + ".dp")
    r, _, w = DP.init_from_env("gloo")
    assert (r, w) == (rank, world)
    net = _net()
    if rank != 0:                                   # replicas start different, the broadcast must fix that
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    DP.broadcast_parameters(net)
    g = torch.Generator().manual_seed(123)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = DP.shard_batch(8)
    assert (lo, hi) == (rank * 4, rank * 4 + 4)
    sync = DP.FlatGradSync(params=net.parameters())
    _loss(net, X[lo:hi], Y[lo:hi]).backward()
    sync.pack()                                     # the unused layer has no .grad: its slots must be zero
    sync.all_reduce()
    sync.unpack(scale=sync.average_scale())
    with torch.no_grad():
        for p in net.parameters():
            p -= 0.1 * p.grad
    bn = torch.nn.BatchNorm1d(4)
    bn.running_mean.fill_(float(rank))
    DP.average_bn_running_stats(bn)
    torch.save(dict(grads=[p.grad.clone() for p in net.parameters()], params=[p.detach().clone() for p in net.parameters()],
                    bn_mean=bn.running_mean.clone()), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    # replicas identical bit for bit after the exchange and the update
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)
    # and equal to the single-process gradient of the whole batch
    net = _net()
    g = torch.Generator().manual_seed(123)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    _loss(net, X, Y).backward()
    for p, got in zip(net.parameters(), r0["grads"]):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(got, ref, atol=1e-6), (got, ref)
    assert torch.allclose(r0["bn_mean"], torch.full((4,), 0.5))


def test_shard_batch_rejects_ragged():
    import importlib
    DP = importlib.import_module(PKG + ".dp")
    assert DP.shard_batch(64, rank=3, world=4) == (48, 64)
    with pytest.raises(ValueError):
        DP.shard_batch(10, rank=0, world=4)
    assert DP.world_size() == 1
    s = DP.FlatGradSync(flat_grad=torch.ones(5))
    assert s.all_reduce() is None and s.average_scale() == 1.0


# ---- the two-bucket exchange with the backward pass cut at the front end (dp.BucketedGradSync / dp.BackwardCut) --------
class ConvTC_Block(torch.nn.Module):
    """Toy stand-in with the attributes dp.late_parameters / dp.BackwardCut look for: `.cnn` (front end, late bucket) and
    a call to dp.cut_backward_here on the tensor between the front end and the rest."""

    def __init__(self, DP):
        super().__init__()
        self.DP = DP
        self.cnn = torch.nn.Sequential(torch.nn.Linear(6, 7), torch.nn.Tanh())
        self.tcn = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))

    def forward(self, x):
        return self.tcn(self.DP.cut_backward_here(self, self.cnn(x)))


class _FlatHolder:
    """What BucketedGradSync needs of FlatAdam: one flat gradient buffer, late parameters first, .grad views into it."""

    def __init__(self, params, late):
        late_ids = {id(p) for p in late}
        params = list(params)
        order = [p for p in params if id(p) in late_ids] + [p for p in params if id(p) not in late_ids]
        self.late_numel = sum(p.numel() for p in params if id(p) in late_ids)
        self.flat_grad = torch.zeros(sum(p.numel() for p in params))
        off = 0
        for p in order:
            p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)
            off += p.numel()


def _bucket_worker(rank, world, port, out_dir):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    DP = importlib.import_module(PKG + ".dp")
    DP.init_from_env("gloo")
    torch.manual_seed(0)
    net = ConvTC_Block(DP)
    late = DP.late_parameters(net)
    assert [id(p) for p in late] == [id(p) for p in net.cnn.parameters()]
    holder = _FlatHolder(net.parameters(), late)
    sync = DP.BucketedGradSync(holder, net)
    assert sync.cut is not None and sync.late_numel == holder.late_numel == 6 * 7 + 7
    g = torch.Generator().manual_seed(123)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = DP.shard_batch(8)
    stages = {}
    sync.cut.reset()
    loss = ((net(X[lo:hi]) - Y[lo:hi]) ** 2).mean()
    loss.backward()                                             # stops at the cut
    stages["late_after_phase1"] = holder.flat_grad[:holder.late_numel].abs().max().item()
    stages["main_after_phase1"] = holder.flat_grad[holder.late_numel:].abs().max().item()
    sync.reduce_main()
    sync.cut.finish()                                           # front end
    stages["late_after_phase2"] = holder.flat_grad[:holder.late_numel].abs().max().item()
    sync.reduce_late()
    sync.wait()
    torch.save(dict(flat=holder.flat_grad.clone(), stages=stages, scale=sync.average_scale(),
                    grads=[p.grad.clone() for p in net.parameters()]), os.path.join(out_dir, f"bucket{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_bucketed_exchange_with_backward_cut(tmp_path):
    import importlib
    DP = importlib.import_module(PKG + ".dp")
    port = _free_port()
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "bucket0.pt")
    r1 = torch.load(tmp_path / "bucket1.pt")
    assert torch.equal(r0["flat"], r1["flat"]) and r0["scale"] == 0.5
    for r in (r0, r1):      # phase 1 leaves the late bucket untouched and completes the main one
        assert r["stages"]["late_after_phase1"] == 0.0 and r["stages"]["main_after_phase1"] > 0.0
        assert r["stages"]["late_after_phase2"] > 0.0
    # mean over ranks of the exchanged sums == gradient of the whole batch in one process, without any cut
    torch.manual_seed(0)
    net = ConvTC_Block(DP)
    g = torch.Generator().manual_seed(123)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    ((net(X) - Y) ** 2).mean().backward()
    for p, got in zip(net.parameters(), r0["grads"]):
        assert torch.allclose(got * 0.5, p.grad, atol=1e-6)
