"""Rank program of tests/test_bench_launcher.py: bench.py's own rank body (bench.run_rank via bench.main) on a
gloo process group, with the CPU oracle standing in for the GPU renderer.  This is test infrastructure -- the
launcher, the WORLD_SIZE / rank checks, the all-reduce-of-ones membership proof, the parity gate plumbing, the
strong/weak source split and the JSON line are bench.py's; only the device and the renderer are swapped."""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402


class OracleStandIn:
    """render_* surface of device.TransientRenderer computed by the oracle on CPU tensors."""

    def __init__(self, seed=0, threads=2):
        self.seed, self.threads = seed, threads

    def render_gradient(self, origin, normal, vertices, faces, num_sample, lb, ub, res, data=None, weight=None,
                        refine_scale=10, sigma_bin=1, testing_flag=1, loss_flag=0, gradient=None, source_offset=0,
                        total_sources=0, seed=None, zero_gradient=False, source_stride=1, **kw):
        import oracle
        t, g, p = oracle.render_gradient(origin.numpy(), normal.numpy(), vertices.numpy(), faces.numpy(), num_sample,
                                         lb, ub, res, data.numpy(), weight.numpy(), refine=refine_scale,
                                         sigma_bin=sigma_bin, testing_flag=testing_flag, loss_flag=loss_flag, accel=1,
                                         threads=self.threads, source_offset=source_offset, total_sources=total_sources,
                                         source_stride=source_stride, seed=self.seed if seed is None else seed)
        g = torch.from_numpy(g)
        if gradient is not None:
            if zero_gradient:
                gradient.zero_()
            gradient += g
            g = gradient
        return torch.from_numpy(t), g, torch.from_numpy(p)

    def render_transient(self, origin, normal, vertices, faces, num_sample, lb, ub, res, source_offset=0,
                         total_sources=0, seed=None, source_stride=1, **kw):
        import oracle
        t, p = oracle.render_transient(origin.numpy(), normal.numpy(), vertices.numpy(), faces.numpy(), num_sample,
                                       lb, ub, res, accel=1, threads=self.threads, source_offset=source_offset,
                                       total_sources=total_sources, source_stride=source_stride,
                                       seed=self.seed if seed is None else seed)
        return torch.from_numpy(t), torch.from_numpy(p)


class CpuBackend:
    name = "gloo"

    def __init__(self, local_rank):
        self.device = torch.device("cpu")

    def make_renderer(self):
        return OracleStandIn(seed=0)

    def sync(self):
        pass


if __name__ == "__main__":
    sys.exit(bench.main(backend_factory=CpuBackend))
