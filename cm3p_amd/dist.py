"""Cross-rank in-batch negatives: the one exchange step of the data-parallel contrastive step.

The reference never gathers: under DDP every rank contrasts only its own pairs.  With
`model.gather_negatives = True` each rank all-gathers the L2-normalised (b, P) beatmap and metadata embeddings of every
rank (RCCL over xGMI on GPUs, gloo in the tests), scores its b rows against all N*b columns in both directions, and
takes the cross-entropy with targets r*b + i.  Parity definition: with DDP's gradient averaging the parameter gradients
equal those of the single-process loss on the concatenated N*b batch, and the mean of the per-rank losses equals that
loss.

Overlap: a gather is STARTED as soon as its embeddings exist (`start_gather`, an asynchronous collective: on RCCL it runs
on the process group's own HIP stream, ordered after the producer kernels by an event, while the compute stream goes on)
and JOINED only where the logits need it (`PendingGather.wait`).  CM3PModel.forward starts the beatmap gather right after
the beatmap projection, so it travels under the whole metadata tower; the metadata gather, whose result is needed at once,
is the only exposed one (64 KiB per rank at b = 32).

The autograd node is torch.distributed plumbing and device agnostic; the arithmetic around it (logits, loss) is the HIP
head of modeling_cm3p.py.  Its backward is the transpose collective: a sum reduce-scatter of the gradient w.r.t. the
gathered buffer, so gradients that other ranks hold for this rank's embeddings come home.
"""
from __future__ import annotations

import warnings

import torch
import torch.distributed as dist

Tensor = torch.Tensor


class AllGatherEmbeds(torch.autograd.Function):
    """(b, ...) -> (N*b, ...) concatenated in rank order; backward = reduce-scatter(sum).

    With `holder` (a list) the forward only STARTS the collective and appends its work handle: the returned tensor must not be
    read before `holder[0].wait()` (PendingGather does that)."""

    @staticmethod
    def forward(ctx, x: Tensor, group=None, holder=None):
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        if holder is None:
            dist.all_gather_into_tensor(out, x, group=group)
        else:
            holder.append(dist.all_gather_into_tensor(out, x, group=group, async_op=True))
            holder.append(x)  # the send buffer stays alive until the join
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
        return out, None, None


class PendingGather:
    """An all-gather in flight: `.wait()` joins it into the calling stream and returns the (N*b, ...) tensor."""

    def __init__(self, x: Tensor, group=None):
        self._holder: list = []
        self._out = AllGatherEmbeds.apply(x, group, self._holder)

    def wait(self) -> Tensor:
        if self._holder:
            self._holder[0].wait()  # RCCL: the current stream waits for the collective's stream; gloo: blocks the host
            self._holder.clear()
        return self._out


def start_gather(x: Tensor, group=None) -> PendingGather:
    return PendingGather(x, group)


def gather_pair(metadata_embeds: Tensor, beatmap_embeds: Tensor, group=None):
    """Both modalities: (b, P), (b, P) -> (N*b, P), (N*b, P), two collectives started back to back and joined together."""
    pm, pb = start_gather(metadata_embeds, group), start_gather(beatmap_embeds, group)
    return pm.wait(), pb.wait()


def gathered_contrastive(metadata_embeds: Tensor, beatmap_embeds: Tensor, logit_scale: Tensor, group=None,
                         beatmap_pending: PendingGather | None = None):
    """-> (logits_per_metadata (b, N*b), logits_per_beatmap (b, N*b), loss) on this rank (HIP head).  `beatmap_pending`: the
    beatmap gather the caller started earlier (it has been travelling under the metadata tower)."""
    from .modeling_cm3p import _CrossEntropySumFn, _LogitsFn

    r = dist.get_rank(group)
    b = metadata_embeds.shape[0]
    pm = start_gather(metadata_embeds, group)
    pb = beatmap_pending if beatmap_pending is not None else start_gather(beatmap_embeds, group)
    b_all = pb.wait()
    lpm = _LogitsFn.apply(metadata_embeds, b_all, logit_scale)  # this rank's metadata rows vs every beatmap
    m_all = pm.wait()
    lpb = _LogitsFn.apply(beatmap_embeds, m_all, logit_scale)   # this rank's beatmap rows vs every metadata
    n = lpm.shape[1]
    target = torch.arange(r * b, (r + 1) * b, device=lpm.device, dtype=torch.int64)
    specs = [(0, b, n, n, 1, None, target, 0.5), (1, b, n, n, 1, None, target, 0.5)]
    loss = _CrossEntropySumFn.apply(specs, lpm, lpb)
    return lpm, lpb, loss


_warned_3d = False


def warn_variations_stay_local():
    """gather_negatives with (B, V, L) metadata variations: the variations are per-sample structured negatives and stay
    rank-local (decided and documented in DESIGN.md §6); say so once instead of silently ignoring the flag."""
    global _warned_3d
    if not _warned_3d:
        _warned_3d = True
        warnings.warn("CM3PModel.gather_negatives is set but metadata_ids is (B, V, L): metadata variations are rank-local "
                      "negatives; this batch is scored without cross-rank gathering.", RuntimeWarning, stacklevel=3)


# ---- what a multi-rank run says about itself (bench.py; device agnostic, so that world sizes one card cannot host are rehearsed over gloo
# on CPU tensors: tests/test_dist_gloo.py runs both at world 8)
def replica_report(parameters, device, peak_memory_bytes: float = 0.0, workspace_bytes: float = 0.0, group=None) -> dict:
    """After a DDP step every rank must hold the SAME averaged gradients, bit for bit: check-sum every rank's
    gradients (int32 view, 64-bit sum), compare the sums with one MIN and one MAX all-reduce, gather per-rank memory figures."""
    world = dist.get_world_size(group)
    acc = torch.zeros((), dtype=torch.int64, device=device)
    for p in parameters:
        if p.grad is not None:
            acc += p.grad.detach().contiguous().view(torch.int32).to(torch.int64).sum()
    lo, hi = acc.clone(), acc.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    mem = torch.tensor([peak_memory_bytes, workspace_bytes], device=device, dtype=torch.float64)
    mems = [torch.zeros_like(mem) for _ in range(world)]
    dist.all_gather(mems, mem, group=group)
    return {"gradient_checksum": int(acc.item()), "identical_on_all_ranks": bool(lo.item() == hi.item()),
            "checksum_min": int(lo.item()), "checksum_max": int(hi.item()),
            "peak_memory_gb_per_rank": [round(m[0].item() / 2 ** 30, 2) for m in mems],
            "attention_workspace_gb_per_rank": [round(m[1].item() / 2 ** 30, 2) for m in mems]}


def choose_gemm_grid(default_grid: int, ms_default: float, ms_surplus: float, backend: str) -> dict:
    """The grid of the ring-kernel GEMMs at N > 1, from two warm-up timings that are already the MAX over ranks (so every rank decides
    alike): 1024 workgroups when that is >= 1 % faster over RCCL (whose channel workgroups hold CUs), else what was set (0 = one per CU)."""
    use_surplus = default_grid == 0 and backend == "nccl" and ms_surplus < 0.99 * ms_default
    return {"candidates": {"one per CU" if default_grid == 0 else str(default_grid): ms_default, "1024": ms_surplus},
            "selected": 1024 if use_surplus else (default_grid or "one per CU"), "grid": 1024 if use_surplus else default_grid,
            "rule": "1024 workgroups if >= 1 % faster over RCCL, measured in the warm-up (2 steps each, max over ranks)"}
