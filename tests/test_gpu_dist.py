"""The exchange step of SURVEY.md section 8e on the GPU: torch.distributed backend "nccl" (= RCCL) initialises in
this image, sums the renderer's float64 vertex gradient in place on the stream the kernels ran on, and the
sharded renderer's block arithmetic reproduces the unsharded render.  A GPU box of this pool has one device,
so the process group has world size 1 here; the two-rank logic is covered on CPU (test_dist_cpu.py)."""
import socket

import numpy as np
import pytest

from conftest import grid_sources, rel_l2

pytestmark = pytest.mark.gpu

LB, UB, RES, T = 0.625, 1.625, 2.0 ** -9, 512


def test_rccl_group_and_sharded_blocks(bunny):
    import torch
    import torch.distributed as dist
    from nlos_surface_optimization_amd import device as nd
    from nlos_surface_optimization_amd.dist import ShardedRenderer, shard_bounds

    v_np, f_np = bunny
    dev = torch.device("cuda", 0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        origin_np, normal_np = grid_sources(3, 0.2)
        L = origin_np.shape[0]
        origin, normal = torch.from_numpy(origin_np).to(dev), torch.from_numpy(normal_np).to(dev)
        v, f = torch.from_numpy(v_np).to(dev), torch.from_numpy(f_np).to(dev)
        rs = np.random.RandomState(2)
        data = torch.from_numpy(rs.random_sample((L, T)) * 1e-3).to(dev)
        weight = torch.ones_like(data)
        r = nd.TransientRenderer(dev, seed=5)
        ns = 3 * f_np.shape[0]
        t_ref, g_ref, _ = r.render_gradient(origin, normal, v, f, ns, LB, UB, RES, data=data, weight=weight)
        t_ref, g_ref = t_ref.clone(), g_ref.clone()

        # the collective itself: in-place fp64 sum on the current stream, directly after the kernels
        sr = ShardedRenderer(r, L)
        assert (sr.rank, sr.world_size, sr.lo, sr.hi) == (0, 1, 0, L)
        t1, g1, _ = sr.render_gradient(origin, normal, v, f, ns, LB, UB, RES, data=data, weight=weight)
        g1 = g1.clone()
        dist.all_reduce(g1, op=dist.ReduceOp.SUM)
        dist.barrier()
        torch.cuda.synchronize()
        # fp64 atomics land in arrival order: repeat renders agree to rounding, not bit for bit
        assert rel_l2(t1.cpu().numpy(), t_ref.cpu().numpy()) < 1e-12
        # (gradient: to ~1e-8 -- the tap loop's (float)(-2 difference) amplifies the rows' last-bit noise at bins that sit
        # on fp32 rounding midpoints, tests/test_gpu_rows.py::test_render_step_is_hip_graph_capturable)
        assert rel_l2(g1.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6
        assert sr.gather_transient(t1) is t1

        # what two ranks would compute, done in turn on this device, then summed as the all-reduce would
        acc = torch.zeros_like(g_ref)
        rows = []
        for rank in range(2):
            part = ShardedRenderer(r, L, rank=rank, world_size=2)
            assert (part.lo, part.hi) == shard_bounds(L, rank, 2)
            t, g, _ = nd.TransientRenderer.render_gradient(
                r, part.local(origin), part.local(normal), v, f, ns, LB, UB, RES, data=part.local(data),
                weight=part.local(weight), source_offset=part.lo, total_sources=L)
            rows.append(t.clone())
            acc += g
        assert rel_l2(torch.cat(rows).cpu().numpy(), t_ref.cpu().numpy()) < 1e-12   # rows depend on their own source only
        assert rel_l2(acc.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6    # summation order, through the fp32 residual of the tap loop

        # the STRIDED partition (every N-th source, source_stride = N): the same rows, interleaved, and the same sum --
        # and a strided shard agrees with the oracle rendering the same shard (keys of the global source indices)
        import oracle
        acc = torch.zeros_like(g_ref)
        full = torch.zeros_like(t_ref)
        for rank in range(3):
            part = ShardedRenderer(r, L, rank=rank, world_size=3, partition="strided")
            assert (part.offset, part.stride) == (rank, 3) and part.local(origin).is_contiguous()
            t, g, _ = part.render_gradient(part.local(origin), part.local(normal), v, f, ns, LB, UB, RES,
                                           data=part.local(data), weight=part.local(weight))
            full[part.slice] = t
            acc += g
            if rank == 1:
                t_o, g_o, _ = oracle.render_gradient(origin_np[1::3], normal_np[1::3], v_np, f_np, ns, LB, UB, RES,
                                                     part.local(data).cpu().numpy(), part.local(weight).cpu().numpy(),
                                                     accel=1, seed=5, source_offset=1, source_stride=3, total_sources=L)
                assert rel_l2(t.cpu().numpy(), t_o) < 1e-12 and rel_l2(g.cpu().numpy(), g_o) < 1e-4
        assert rel_l2(full.cpu().numpy(), t_ref.cpu().numpy()) < 1e-12
        assert rel_l2(acc.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6

        # "rccl-direct": a communicator of the process's own on torch's librccl.so, ncclAllReduce enqueued on the render
        # stream itself (world size 1 here: the sum over one rank, in stream order behind the kernels that produce it)
        sd = ShardedRenderer(r, L, all_reduce="rccl-direct")
        try:
            t5, g5, _ = sd.render_gradient(origin, normal, v, f, ns, LB, UB, RES, data=data, weight=weight)
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):            # ... and on whatever stream is current
                t6, g6, _ = sd.render_gradient(origin, normal, v, f, ns, LB, UB, RES, data=data, weight=weight)
            side.synchronize()
            torch.cuda.synchronize()
            assert rel_l2(g5.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6 and rel_l2(g6.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6
            assert rel_l2(t5.cpu().numpy(), t_ref.cpu().numpy()) < 1e-12
        finally:
            sd.direct.close()
    finally:
        dist.destroy_process_group()
