"""Source-block sharding across the GPUs of one node.

Every (source, face, sample) unit is independent, and pass 2 for source l needs only
row l of the residual, so the path shards over contiguous source blocks -- the
reference's own batching axis (exp_bunny/test.py:66-67,161-167).  The mesh and the BVH
are replicated; transient rows stay on the GPU that owns the block.  The only exchange
step is ONE sum all-reduce of the 3V-double vertex gradient per optimisation step (RCCL
over xGMI under torch.distributed backend "nccl"; ~60 KB at V = 2.5k, i.e. latency
bound).  Each rank scales by 1/L_global inside the kernel
(smoothed_transient/transient_and_gradient.cpp:563), so the reduced gradient equals the
single-GPU one up to fp64 summation order.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_sources, rank, world_size):
    """Contiguous block [lo, hi) of sources owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_sources), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_reduce_gradient(gradient, group=None):
    """Sum the per-rank partial vertex gradients in place (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(gradient, op=dist.ReduceOp.SUM, group=group)
    return gradient


class ShardedRenderer:
    """Renders this rank's source block and all-reduces the vertex gradient.

    `renderer` is a device.TransientRenderer (or any object with the same
    render_transient / render_gradient methods: the CPU tests plug in a stand-in to cover
    the sharding and collective logic under gloo).
    """

    def __init__(self, renderer, n_sources, rank=None, world_size=None, group=None):
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.renderer = renderer
        self.n_sources = int(n_sources)
        self.rank, self.world_size, self.group = rank, world_size, group
        self.lo, self.hi = shard_bounds(n_sources, rank, world_size)

    def local(self, per_source):
        """Slice a [L_global, ...] tensor down to this rank's block (a view, no copy)."""
        return per_source[self.lo:self.hi]

    def render_transient(self, origin, normal, *args, **kw):
        return self.renderer.render_transient(origin, normal, *args, source_offset=self.lo,
                                              total_sources=self.n_sources, **kw)

    def render_gradient(self, origin, normal, *args, **kw):
        """origin/normal/data/weight are the LOCAL blocks. Returns (local transient rows,
        globally reduced gradient, pathlengths).  A caller-supplied `gradient=` buffer is accumulated into as
        the renderer does (v2 semantics) -- AFTER the reduction, so that what it already holds (e.g. a
        regulariser gradient present on every rank) is not multiplied by the world size."""
        into = kw.pop("gradient", None)
        transient, gradient, path = self.renderer.render_gradient(
            origin, normal, *args, source_offset=self.lo, total_sources=self.n_sources, **kw)
        all_reduce_gradient(gradient, self.group)
        if into is not None:
            into += gradient
            gradient = into
        return transient, gradient, path

    def gather_transient(self, local_rows):
        """Optional: assemble the full [L, T] transient on every rank (host asks for it rarely)."""
        if self.world_size == 1:
            return local_rows
        sizes = [shard_bounds(self.n_sources, r, self.world_size) for r in range(self.world_size)]
        maxrows = max(hi - lo for lo, hi in sizes)
        pad = torch.zeros((maxrows, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        pad[: local_rows.shape[0]] = local_rows
        out = [torch.empty_like(pad) for _ in range(self.world_size)]
        dist.all_gather(out, pad, group=self.group)
        return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)
