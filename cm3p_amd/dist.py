"""Cross-rank in-batch negatives: the one exchange step of the data-parallel contrastive step.

The reference never gathers: under DDP every rank contrasts only its own pairs.  With
`model.gather_negatives = True` each rank all-gathers the L2-normalised (b, P) beatmap and metadata embeddings of every
rank (one fused (b, 2, P) fp32 buffer; RCCL over xGMI on GPUs, gloo in the tests), scores its b rows against all
N*b columns in both directions, and takes the cross-entropy with targets r*b + i.  Parity definition: with DDP's
gradient averaging the parameter gradients equal those of the single-process loss on the concatenated N*b batch, and
the mean of the per-rank losses equals that loss.

The autograd node below is torch.distributed plumbing and device agnostic; the arithmetic around it (logits, loss) is
the HIP head of modeling_cm3p.py.  Its backward is the transpose collective: a sum reduce-scatter of the gradient
w.r.t. the gathered buffer, so gradients that other ranks hold for this rank's embeddings come home.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

Tensor = torch.Tensor


class AllGatherEmbeds(torch.autograd.Function):
    """(b, ...) -> (N*b, ...) concatenated in rank order; backward = reduce-scatter(sum)."""

    @staticmethod
    def forward(ctx, x: Tensor, group=None):
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        world = dist.get_world_size(ctx.group)
        g = g.contiguous()
        out = torch.empty((g.shape[0] // world, *g.shape[1:]), dtype=g.dtype, device=g.device)
        if dist.get_backend(ctx.group) == "gloo":
            # gloo has no reduce_scatter_tensor: all-reduce, then keep this rank's slice (CPU tests only)
            g = g.clone()
            dist.all_reduce(g, group=ctx.group)
            r = dist.get_rank(ctx.group)
            out.copy_(g[r * out.shape[0]:(r + 1) * out.shape[0]])
        else:
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None


def gather_pair(metadata_embeds: Tensor, beatmap_embeds: Tensor, group=None):
    """One collective for both modalities: (b, P), (b, P) -> (N*b, P), (N*b, P).

    The fused buffer is 2*b*P fp32 (128 KiB at b = 32): latency-bound, a few tens of microseconds on RCCL, and the
    logits need it immediately, so it is issued on the compute stream (a side stream would buy nothing here and would
    need cross-stream allocator bookkeeping)."""
    fused = torch.stack((metadata_embeds, beatmap_embeds), dim=1)  # (b, 2, P): layout only, no arithmetic
    allf = AllGatherEmbeds.apply(fused, group)
    return allf[:, 0].contiguous(), allf[:, 1].contiguous()


def gathered_contrastive(metadata_embeds: Tensor, beatmap_embeds: Tensor, logit_scale: Tensor, group=None):
    """-> (logits_per_metadata (b, N*b), logits_per_beatmap (b, N*b), loss) on this rank (HIP head)."""
    from .modeling_cm3p import _CrossEntropySumFn, _LogitsFn

    r = dist.get_rank(group)
    b = metadata_embeds.shape[0]
    m_all, b_all = gather_pair(metadata_embeds, beatmap_embeds, group)
    lpm = _LogitsFn.apply(metadata_embeds, b_all, logit_scale)  # this rank's metadata rows vs every beatmap
    lpb = _LogitsFn.apply(beatmap_embeds, m_all, logit_scale)   # this rank's beatmap rows vs every metadata
    n = lpm.shape[1]
    target = torch.arange(r * b, (r + 1) * b, device=lpm.device, dtype=torch.int64)
    specs = [(0, b, n, n, 1, None, target, 0.5), (1, b, n, n, 1, None, target, 0.5)]
    loss = _CrossEntropySumFn.apply(specs, lpm, lpb)
    return lpm, lpb, loss
