import sys, numpy as np, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nlos_surface_optimization_amd import device as nd, mesh_io
d = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/bunny_5k.npz"))
v0, f0 = np.ascontiguousarray(d["v"], np.float32), np.ascontiguousarray(d["f"], np.int32)
v, f = mesh_io.subdivide(v0, f0, 1)
g = np.linspace(-0.2, 0.2, 2)
o = np.array([[x, y, 0] for y in g for x in g], np.float32)
n = np.tile(np.array([0, 0, 1], np.float32), (4, 1))
dev = torch.device("cuda", 0)
r = nd.TransientRenderer(dev, seed=0)
tv, tf, to, tn = (torch.from_numpy(x).to(dev) for x in (v, f, o, n))
for nb, res, lb in ((512, 2.0**-9, 0.625), (1200, 0.0012, 0.0), (1024, 2.0**-10, 0.625), (1152, 1.0/1152, 0.625)):
    ub = float(np.float32(lb + nb * res))
    for spt in (2, 4):
        ns = spt * f.shape[0]
        t1, _ = r.render_transient(to, tn, tv, tf, ns, lb, ub, res)
        p = r.last_path(count=True)
        t2, _ = r.render_transient(to, tn, tv, tf, ns, lb, ub, res, force_bvh=True)
        print(nb, spt, t1.shape, p["backend"], p["rows_in_lds"], p.get("coarsened"), p.get("big_lds"), p.get("bvh_queries"),
              "maxdiff/max %.3e" % ((t1 - t2).abs().max().item() / t2.max().item()), "rows differing", int(((t1 - t2).abs().max(dim=1).values > 1e-12 * t2.max()).sum()))
