"""world_size-2 gloo test of the cross-rank negatives exchange, on CPU.

The product's collective plumbing (cm3p_amd.dist.AllGatherEmbeds / gather_pair) is device agnostic; the arithmetic around
it in this test is the oracle's.  Invariant: with DDP-style gradient AVERAGING, the parameter gradients on every rank
equal the gradients of the reference single-process loss on the concatenated N*b batch, and the mean of the per-rank
losses equals that loss (tolerance 1e-6, fp32).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _towers(params, xm, xb):
    """Stand-in towers: one linear map per modality, then L2 normalisation as the reference does."""
    from oracle import cm3p_oracle as O

    return O.l2_normalize(xm @ params["wm"].t()), O.l2_normalize(xb @ params["wb"].t())


def _make(world, b, seed=0):
    g = torch.Generator().manual_seed(seed)
    params = {"wm": torch.randn(16, 12, generator=g), "wb": torch.randn(16, 20, generator=g), "s": torch.tensor(1.3)}
    xm = torch.randn(world * b, 12, generator=g)
    xb = torch.randn(world * b, 20, generator=g)
    return params, xm, xb


def _worker(rank, world, port, b, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cm3p_amd.dist import start_gather

        params, xm, xb = _make(world, b)
        params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        sl = slice(rank * b, (rank + 1) * b)
        # the product's order of events (CM3PModel.forward): the beatmap gather is started as soon as its embeddings exist,
        # the metadata tower runs while it is in flight, both gathers are joined only where the logits need them
        be = _towers(params, xm[sl], xb[sl])[1]
        pending_b = start_gather(be)
        me = _towers(params, xm[sl], xb[sl])[0]
        pending_m = start_gather(me)
        b_all, m_all = pending_b.wait(), pending_m.wait()
        assert m_all.shape == (world * b, 16)
        scale = params["s"].exp()
        target = torch.arange(rank * b, (rank + 1) * b)
        loss = 0.5 * (F.cross_entropy(me @ b_all.t() * scale, target) + F.cross_entropy(be @ m_all.t() * scale, target))
        loss.backward()
        grads = {}
        for k, p in params.items():
            gavg = p.grad.clone()
            dist.all_reduce(gavg)  # what DDP does: sum, then divide by world size
            grads[k] = gavg / world
        lmean = loss.detach().clone()
        dist.all_reduce(lmean)
        out[rank] = (lmean / world, grads, m_all.detach(), b_all.detach())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,b", [(2, 3), (2, 1), (4, 2), (8, 1), (8, 4)])  # 2 and 4 processes; 8 = the node C3 / C5 run on
def test_gathered_loss_and_grads_equal_single_process_reference(world, b):
    from oracle import cm3p_oracle as O

    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, b, out), nprocs=world, join=True)

    params, xm, xb = _make(world, b)
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    me, be = _towers(params, xm, xb)
    ref_loss = O.cm3p_loss(me @ be.t() * params["s"].exp())  # ref:cm3p/modeling_cm3p.py:33-51,976-977 on the global batch
    ref_loss.backward()
    for rank in range(world):
        lmean, grads, m_all, b_all = out[rank]
        assert torch.allclose(m_all, me.detach(), atol=1e-7) and torch.allclose(b_all, be.detach(), atol=1e-7)  # rank order
        assert abs(lmean.item() - ref_loss.item()) <= 1e-6
        for k, p in params.items():
            assert torch.allclose(grads[k], p.grad, atol=1e-6, rtol=1e-5), (rank, k, (grads[k] - p.grad).abs().max())


def _report_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cm3p_amd.dist import choose_gemm_grid, replica_report

        g = torch.Generator().manual_seed(5)
        ps = [torch.nn.Parameter(torch.randn(7, 5, generator=g)), torch.nn.Parameter(torch.randn(11, generator=g)), torch.nn.Parameter(torch.zeros(3))]
        ps[0].grad, ps[1].grad = torch.randn(7, 5, generator=g), torch.randn(11, generator=g)  # (ps[2] has no gradient: skipped)
        same = replica_report(ps, torch.device("cpu"), float(2 ** 30 * (rank + 1)), float(2 ** 29))
        if rank == world - 1:
            ps[1].grad[3] += 1e-7  # one rank one ulp-scale off: the bit-level sum must notice
        diverged = replica_report(ps, torch.device("cpu"))
        # the grid vote: every rank is handed the same (max-over-ranks) times and must come to the same answer
        t = torch.tensor([100.0 + rank, 98.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[rank] = (same, diverged, choose_gemm_grid(0, float(t[0]), float(t[1]), "nccl"), choose_gemm_grid(0, float(t[0]), float(t[1]), "gloo"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_replica_report_and_grid_vote_at_world_size(world):
    """bench.py's N > 1 bookkeeping on CPU tensors over gloo (one card cannot host 8 ranks): the bit-level gradient checksum agrees on
    identical replicas and notices a single differing element on one rank; the per-rank memory list has one entry per rank in rank
    order; the GEMM-grid vote gives every rank the same decision."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_report_worker, args=(world, port, out), nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))
    for rank in range(world):
        same, diverged, vote_rccl, vote_gloo = out[rank]
        assert same["identical_on_all_ranks"] is True and same["checksum_min"] == same["checksum_max"] == same["gradient_checksum"]
        assert same["peak_memory_gb_per_rank"] == [float(r + 1) for r in range(world)]
        assert same["attention_workspace_gb_per_rank"] == [0.5] * world
        assert diverged["identical_on_all_ranks"] is False and diverged["checksum_min"] != diverged["checksum_max"]
        assert vote_rccl == out[0][2] and vote_gloo == out[0][3]
        assert vote_rccl["selected"] == 1024 and vote_rccl["grid"] == 1024  # 98 + (world - 1) against 100 + (world - 1): >= 1 % faster
        assert vote_gloo["selected"] == "one per CU" and vote_gloo["grid"] == 0  # (the surplus grid is an answer to RCCL's channels only)
