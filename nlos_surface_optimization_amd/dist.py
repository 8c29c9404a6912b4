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
import ctypes
import os

import torch
import torch.distributed as dist


class _NcclUniqueId(ctypes.Structure):          # ncclUniqueId: 128 opaque bytes, passed by value
    _fields_ = [("internal", ctypes.c_char * 128)]


def shard_bounds(n_sources, rank, world_size):
    """Contiguous block [lo, hi) of sources owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_sources), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


PARTITIONS = ("contiguous", "strided")


def shard_slice(n_sources, rank, world_size, partition="contiguous"):
    """The sources of `rank` as a slice of range(n_sources), and (source_offset, source_stride) for the renderer.
    contiguous: the block [lo, hi) of shard_bounds -- the reference's own batching (exp_bunny/test.py:66-67);
    strided: every world_size-th source from `rank` (l = rank mod N).  On a wall grid a contiguous block is one
    strip of the wall, and the strip under the object is the slowest (7 % spread over the 8 blocks of the benchmark);
    a strided shard is an even sample of the whole wall, so the ranks finish together.  RNG keys are made of the GLOBAL
    source index, so rows and reduced gradient do not depend on the partition."""
    if partition not in PARTITIONS:
        raise ValueError("partition must be one of %s" % (PARTITIONS,))
    if partition == "contiguous":
        lo, hi = shard_bounds(n_sources, rank, world_size)
        return slice(lo, hi), lo, 1
    return slice(int(rank), int(n_sources), int(world_size)), int(rank), int(world_size)


class RcclDirect:
    """The vertex-gradient all-reduce enqueued by this process itself on the stream the render kernels run on.

    torch.distributed's "nccl" backend (= RCCL) runs every collective on a stream of its own and hands over with two
    events (current stream -> collective stream -> current stream); on a 0.33 ms step of an 8-way split that hand-off is a
    visible share of the 58 KB all-reduce.  This class opens a communicator of its OWN on the librccl.so torch ships
    (ctypes: ncclGetUniqueId on rank 0, the 128-byte id broadcast through the existing torch process group,
    ncclCommInitRank) and calls ncclAllReduce(sum, float64, in place) directly on `torch.cuda.current_stream()`.
    STATUS: exercised with world size 1 on the GPU (tests/test_gpu_dist.py) -- no multi-GPU node has been available to
    this build, so like every N > 1 path here it is unmeasured on hardware; ShardedRenderer uses it only on request
    (all_reduce="rccl-direct").
    RULES OF USE: this is a SECOND communicator on the same GPUs as torch's process group.  Collectives of the two must
    not be in flight at the same time (a rank that enters a torch collective while another sits in this all-reduce can
    deadlock both): synchronise the render stream -- or at least finish every all_reduce_sum_ -- before calling into
    torch.distributed (ShardedRenderer.gather_transient does), and issue the calls in the same order on every rank.
    If ncclCommInitRank fails on one rank the others block in theirs: construct it right after the process group, where a
    launcher timeout catches that.  close() (or the context manager) destroys the communicator."""

    NCCL_FLOAT64, NCCL_SUM = 8, 0

    def __init__(self, rank, world_size, device, group=None, lib_path=None):
        if lib_path is None:
            lib_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._lib = ctypes.CDLL(lib_path)
        self._lib.ncclGetErrorString.restype = ctypes.c_char_p
        self._lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        self._lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
        self._lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        for fn in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy"):
            getattr(self._lib, fn).restype = ctypes.c_int
        self._lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        self._lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p]
        self.device = torch.device(device)
        uid = _NcclUniqueId()
        if rank == 0:
            self._check(self._lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        if world_size > 1:
            # the id travels through the process group that already exists (device tensor for "nccl", host for "gloo")
            on_gpu = dist.get_backend(group) == "nccl"
            buf = torch.frombuffer(bytearray(bytes(uid.internal) if rank == 0 else bytes(128)), dtype=torch.uint8).clone()
            if on_gpu:
                buf = buf.to(self.device)
            dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ctypes.memmove(ctypes.byref(uid), bytes(buf.cpu().numpy().tobytes()), 128)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self._lib.ncclCommInitRank(ctypes.byref(self._comm), int(world_size), uid, int(rank)), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self._lib.ncclGetErrorString(rc).decode()))

    def all_reduce_sum_(self, tensor):
        """In-place float64 sum over the ranks, enqueued on the current stream of the tensor's device."""
        assert tensor.is_cuda and tensor.dtype == torch.float64 and tensor.is_contiguous()
        stream = torch.cuda.current_stream(tensor.device).cuda_stream
        p = ctypes.c_void_p(tensor.data_ptr())
        with torch.cuda.device(tensor.device):
            self._check(self._lib.ncclAllReduce(p, p, tensor.numel(), self.NCCL_FLOAT64, self.NCCL_SUM, self._comm,
                                                ctypes.c_void_p(stream)), "ncclAllReduce")
        return tensor

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.ncclCommDestroy(self._comm)
            self._comm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def all_reduce_gradient(gradient, group=None):
    """Sum the per-rank partial vertex gradients in place (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(gradient, op=dist.ReduceOp.SUM, group=group)
    return gradient


class ShardedRenderer:
    """Renders this rank's sources and all-reduces the vertex gradient.

    `renderer` is a device.TransientRenderer (or any object with the same
    render_transient / render_gradient methods: the CPU tests plug in a stand-in to cover
    the sharding and collective logic under gloo).  `partition`: see shard_slice().
    """

    def __init__(self, renderer, n_sources, rank=None, world_size=None, group=None, partition="contiguous", all_reduce="torch"):
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.renderer = renderer
        self.n_sources = int(n_sources)
        self.rank, self.world_size, self.group = rank, world_size, group
        self.partition = partition
        if all_reduce not in ("torch", "rccl-direct"):
            raise ValueError("all_reduce must be 'torch' or 'rccl-direct'")
        # "rccl-direct": a communicator of this process's own, collectives on the render stream (see RcclDirect)
        self.direct = RcclDirect(rank, world_size, renderer.device, group) if all_reduce == "rccl-direct" else None
        self.slice, self.offset, self.stride = shard_slice(n_sources, rank, world_size, partition)
        self.n_local = len(range(*self.slice.indices(self.n_sources)))
        self._bounds = shard_bounds(n_sources, rank, world_size) if partition == "contiguous" else None

    # Block bounds exist for the contiguous partition only.  With the strided one a caller that still slices its rows as
    # [sr.lo:sr.hi] would hand the kernels rows of OTHER sources -- same shapes, a silently wrong gradient -- so asking
    # for them raises; local() is the one way to take a rank's rows.
    @property
    def lo(self):
        if self._bounds is None:
            raise AttributeError("ShardedRenderer.lo: the strided partition has no contiguous block; slice with local()")
        return self._bounds[0]

    @property
    def hi(self):
        if self._bounds is None:
            raise AttributeError("ShardedRenderer.hi: the strided partition has no contiguous block; slice with local()")
        return self._bounds[1]

    def close(self):
        """Destroys the rank's own RCCL communicator (all_reduce="rccl-direct"); a no-op otherwise."""
        if self.direct is not None:
            self.direct.close()
            self.direct = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _check_local(self, origin):
        if origin.shape[0] != self.n_local:
            raise ValueError("ShardedRenderer: %d local sources handed in, this rank owns %d of %d (%s partition): slice per-source "
                             "tensors with local()" % (origin.shape[0], self.n_local, self.n_sources, self.partition))

    def _keys(self):
        kw = dict(source_offset=self.offset, total_sources=self.n_sources)
        if self.stride != 1:
            kw["source_stride"] = self.stride
        return kw

    def local(self, per_source):
        """This rank's rows of a [L_global, ...] tensor: a view for the contiguous partition, a contiguous COPY for the
        strided one (the kernels read dense [L_local, ...] arrays; inputs are sliced once, outside the step loop)."""
        part = per_source[self.slice]
        return part if self.stride == 1 else part.contiguous()

    def render_transient(self, origin, normal, *args, **kw):
        self._check_local(origin)
        return self.renderer.render_transient(origin, normal, *args, **self._keys(), **kw)

    def render_gradient(self, origin, normal, *args, **kw):
        """origin/normal/data/weight are the LOCAL rows (self.local(...)). Returns (local transient rows,
        globally reduced gradient, pathlengths).  A caller-supplied `gradient=` buffer is accumulated into as
        the renderer does (v2 semantics) -- AFTER the reduction, so that what it already holds (e.g. a
        regulariser gradient present on every rank) is not multiplied by the world size."""
        self._check_local(origin)
        into = kw.pop("gradient", None)
        transient, gradient, path = self.renderer.render_gradient(origin, normal, *args, **self._keys(), **kw)
        if self.direct is not None:
            self.direct.all_reduce_sum_(gradient)
        else:
            all_reduce_gradient(gradient, self.group)
        if into is not None:
            into += gradient
            gradient = into
        return transient, gradient, path

    def gather_transient(self, local_rows):
        """Optional: assemble the full [L, T] transient on every rank (host asks for it rarely)."""
        if self.world_size == 1:
            return local_rows
        if self.direct is not None:     # nothing of the process's own communicator in flight while torch's collective runs
            torch.cuda.current_stream(local_rows.device).synchronize()
        slices = [shard_slice(self.n_sources, r, self.world_size, self.partition)[0] for r in range(self.world_size)]
        counts = [len(range(*sl.indices(self.n_sources))) for sl in slices]
        pad = torch.zeros((max(counts), local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        pad[: local_rows.shape[0]] = local_rows
        out = [torch.empty_like(pad) for _ in range(self.world_size)]
        dist.all_gather(out, pad, group=self.group)
        full = torch.empty((self.n_sources, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        for o, sl, n in zip(out, slices, counts):
            full[sl] = o[:n]
        return full
