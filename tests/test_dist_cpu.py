"""The N > 1 path on CPU: world_size-2 gloo process group, source-block sharding and the single
all-reduce of the vertex gradient.  The GPU renderer is replaced by a
stand-in that calls the CPU oracle, so the sharding/collective logic of
nlos_surface_optimization_amd.dist is what is under test here."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN


def test_shard_bounds_partition():
    from nlos_surface_optimization_amd.dist import shard_bounds
    for n, w in ((4096, 8), (10, 3), (7, 8), (0, 2), (5, 1)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and b >= a
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_shard_slice_partitions_the_sources():
    from nlos_surface_optimization_amd.dist import shard_slice
    for part in ("contiguous", "strided"):
        for n, w in ((4096, 8), (10, 3), (7, 8), (0, 2), (5, 1)):
            got = []
            for r in range(w):
                sl, off, stride = shard_slice(n, r, w, part)
                idx = list(range(n))[sl]
                assert idx == [off + i * stride for i in range(len(idx))]     # what the kernel's keys are made of
                got += idx
            assert sorted(got) == list(range(n))
    with pytest.raises(ValueError):
        shard_slice(4, 0, 2, "blocked")


class OracleStandIn:
    """Same render_* surface as device.TransientRenderer, computed by the oracle on CPU tensors."""

    def render_gradient(self, origin, normal, vertices, faces, num_sample, lb, ub, res, data=None, weight=None,
                        refine_scale=10, sigma_bin=1, source_offset=0, total_sources=0, source_stride=1, **kw):
        import oracle
        t, g, p = oracle.render_gradient(origin.numpy(), normal.numpy(), vertices.numpy(), faces.numpy(), num_sample,
                                         lb, ub, res, data.numpy(), weight.numpy(), refine=refine_scale,
                                         sigma_bin=sigma_bin, accel=1, threads=2, source_offset=source_offset,
                                         total_sources=total_sources, source_stride=source_stride)
        return torch.from_numpy(t), torch.from_numpy(g), torch.from_numpy(p)

    def render_transient(self, origin, normal, vertices, faces, num_sample, lb, ub, res, source_offset=0,
                         total_sources=0, source_stride=1, **kw):
        import oracle
        t, p = oracle.render_transient(origin.numpy(), normal.numpy(), vertices.numpy(), faces.numpy(), num_sample,
                                       lb, ub, res, accel=1, threads=2, source_offset=source_offset,
                                       total_sources=total_sources, source_stride=source_stride)
        return torch.from_numpy(t), torch.from_numpy(p)


def _worker(rank, world, port, out_dir, partition="contiguous"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nlos_surface_optimization_amd.dist import ShardedRenderer
        d = np.load(os.path.join(GOLDEN, "mannequin.npz"))
        v = torch.from_numpy(np.ascontiguousarray(d["v"], np.float32))
        f = torch.from_numpy(np.ascontiguousarray(d["f"], np.int32))
        g = np.linspace(-0.3, 0.3, 3)
        origin = torch.tensor([[x, y, 0] for y in g for x in g], dtype=torch.float32)[:7]   # ragged: 4 + 3
        normal = torch.tensor([[0, 0, 1.0]] * 7, dtype=torch.float32)
        lb, ub, res, ns = 0.0, 2.4576, 2.4e-3, 4000
        T = 1024
        rs = np.random.RandomState(0)
        data = torch.from_numpy(rs.random_sample((7, T)) * 1e-3)
        weight = torch.ones((7, T), dtype=torch.float64)
        sr = ShardedRenderer(OracleStandIn(), 7, partition=partition)
        if partition == "contiguous":
            assert (sr.lo, sr.hi, sr.offset, sr.stride) == ((0, 4, 0, 1) if rank == 0 else (4, 7, 4, 1))
        else:   # sources 0, 2, 4, 6 | 1, 3, 5
            assert (sr.offset, sr.stride, sr.local(origin).shape[0]) == ((0, 2, 4) if rank == 0 else (1, 2, 3))
            assert sr.local(origin).is_contiguous() and torch.equal(sr.local(origin), origin[rank::2])
            # (round-4 advice) no block bounds for a strided shard: a stale [sr.lo:sr.hi] must fail, not mis-slice
            for name in ("lo", "hi"):
                try:
                    getattr(sr, name)
                    raise SystemExit("ShardedRenderer.%s of a strided shard did not raise" % name)
                except AttributeError:
                    pass
        # rows of the wrong shard (shapes that do not match this rank's share) are refused before anything is rendered
        try:
            sr.render_transient(origin, normal, v, f, ns, lb, ub, res)
            raise SystemExit("ShardedRenderer accepted the global rows as local ones")
        except ValueError:
            pass
        t_loc, grad, _ = sr.render_gradient(sr.local(origin), sr.local(normal), v, f, ns, lb, ub, res,
                                            data=sr.local(data), weight=sr.local(weight))
        full = sr.gather_transient(t_loc)
        assert full.shape == (7, T)
        if rank == 0:
            ref_t, ref_g, _ = OracleStandIn().render_gradient(origin, normal, v, f, ns, lb, ub, res, data=data,
                                                             weight=weight, total_sources=7)
            np.save(os.path.join(out_dir, "ok.npy"), np.array([
                float((full - ref_t).abs().max()),
                float((grad - ref_g).norm() / ref_g.norm()),
                float(ref_g.norm())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition", ["contiguous", "strided"])
def test_two_rank_gloo_sharding_matches_single_process(tmp_path, partition):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), partition), nprocs=2, join=True)
    r = np.load(os.path.join(str(tmp_path), "ok.npy"))
    assert r[0] == 0.0            # transient rows identical (row l depends only on source l)
    assert r[1] < 1e-12 and r[2] > 0   # all-reduced gradient == single-process gradient (fp64 order)
