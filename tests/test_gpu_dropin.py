"""The numpy drop-in (include/nlos_hip.h section 1, `renderer.renderStreamedGradient` on host arrays: what the reference's
main.py:114-115 / exp_bunny/rendering.py:252-269 call) after round 6's pipelining: uploads of `data` / `weight` ride behind
pass 1 on a copy stream, the rows come down behind pass 2.  Results must be those of the device-pointer path (up to the order of the fp64 atomics), an
array the caller mutated between two calls must be read again, and a second thread's call must not corrupt the first."""
import threading

import numpy as np
import pytest

from conftest import grid_sources, rel_l2

pytestmark = pytest.mark.gpu


def _inputs(bunny, n=8, T=512):
    v, f = bunny
    origin, normal = grid_sources(n, 0.25)
    L = origin.shape[0]
    rs = np.random.RandomState(5)
    data = np.ascontiguousarray(rs.uniform(0.0, 1e-3, (L, T)))
    weight = np.ascontiguousarray(rs.uniform(0.5, 1.5, (L, T)))
    return v, f, origin, normal, data, weight


def _call(v, f, origin, normal, data, weight, T=512):
    from nlos_surface_optimization_amd import renderer
    L = origin.shape[0]
    tr, path, grad = np.zeros((L, T)), np.zeros(T), np.zeros((v.shape[0], 3))
    renderer.renderStreamedGradient(origin, normal, v, f, 20000, 0.625, 1.625, 2.0 ** -9, tr, path, grad, data, weight,
                                    10, 1, 1, 0)
    return tr, path, grad


def test_dropin_equals_device_path(bunny):
    import torch
    from nlos_surface_optimization_amd import device as nd
    v, f, origin, normal, data, weight = _inputs(bunny)
    tr, path, grad = _call(v, f, origin, normal, data, weight)
    dev = torch.device("cuda", 0)
    r = nd.TransientRenderer(dev, seed=0)
    t = lambda x: torch.from_numpy(x).to(dev)
    g0 = torch.zeros((v.shape[0], 3), dtype=torch.float64, device=dev)
    td, gd, pd = r.render_gradient(t(origin), t(normal), t(v), t(f), 20000, 0.625, 1.625, 2.0 ** -9, data=t(data), weight=t(weight),
                                   refine_scale=10, sigma_bin=1, testing_flag=1, loss_flag=0, gradient=g0)
    # rows and gradient are accumulated with fp64 atomics (ds_add_f64 / global atomics: the order varies from launch to
    # launch, 1e-16 relative): the same values up to that, on either path
    assert rel_l2(td.cpu().numpy(), tr) <= 1e-13
    assert np.array_equal(pd.cpu().numpy(), path)
    assert rel_l2(grad, gd.cpu().numpy()) <= 1e-9


def test_mutated_data_and_weight_are_read_again(bunny, orc):
    """No stale device copy: the same numpy objects, changed in place between two calls, give the gradient of the NEW
    contents (checked against the oracle) -- and the old one again when changed back."""
    v, f, origin, normal, data, weight = _inputs(bunny, n=4)
    _, _, g_a = _call(v, f, origin, normal, data, weight)
    keep = data.copy()
    data *= 3.0
    data[1, 100:140] += 2e-3
    weight[2, :] = 0.25
    _, _, g_b = _call(v, f, origin, normal, data, weight)
    assert rel_l2(g_b, g_a) > 1e-3                                     # the inputs do matter
    _, g_o, _ = orc.render_gradient(origin, normal, v, f, 20000, 0.625, 1.625, 2.0 ** -9, data, weight, refine=10, sigma_bin=1,
                                    testing_flag=1, loss_flag=0, seed=0)
    assert rel_l2(g_b, g_o) <= 1e-4
    data[:] = keep
    weight[2, :] = _inputs(bunny, n=4)[5][2, :]
    _, _, g_c = _call(v, f, origin, normal, data, weight)
    assert rel_l2(g_c, g_a) <= 1e-9


def test_rows_are_complete_when_the_call_returns(bunny):
    """The rows come down on the copy stream behind pass 2: the call must not return before they have landed (fresh output
    arrays every call, every row compared with a forward-only render)."""
    from nlos_surface_optimization_amd import renderer
    v, f, origin, normal, data, weight = _inputs(bunny, n=8)
    L, T = origin.shape[0], 512
    ref, path = np.zeros((L, T)), np.zeros(T)
    renderer.renderStreamedTransient(origin, normal, v, f, 20000, 0.625, 1.625, 2.0 ** -9, ref, path, 1, 1)
    for _ in range(5):
        tr, _, _ = _call(v, f, origin, normal, data, weight)
        assert rel_l2(tr, ref) <= 1e-13 and np.count_nonzero(tr) == np.count_nonzero(ref)


def test_two_threads_share_the_default_context(bunny):
    """Section 1 serialises host calls on the default context (one mutex): two Python threads calling at once both get
    their own results."""
    v, f, origin, normal, data, weight = _inputs(bunny, n=4)
    data2 = np.ascontiguousarray(2.0 * data)
    want1 = _call(v, f, origin, normal, data, weight)[2]
    want2 = _call(v, f, origin, normal, data2, weight)[2]
    out = {}

    def work(key, d):
        for _ in range(4):
            out[key] = _call(v, f, origin, normal, d, weight)[2]
    th = [threading.Thread(target=work, args=(1, data)), threading.Thread(target=work, args=(2, data2))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert rel_l2(out[1], want1) <= 1e-9 and rel_l2(out[2], want2) <= 1e-9
