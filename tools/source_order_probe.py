import os, sys, time, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
os.environ["NLOS_FWD_ORDER"] = "0"   # the caller's order, as given
from nlos_surface_optimization_amd import device as nd
d = np.load(os.path.join(ROOT, "tests/golden/bunny_5k.npz"))
dev = torch.device("cuda", 0)
g = np.linspace(-0.25, 0.25, 64)
o_np = np.array([[x, y, 0] for y in g for x in g], np.float32)
v = torch.from_numpy(np.ascontiguousarray(d["v"], np.float32)).to(dev); f = torch.from_numpy(np.ascontiguousarray(d["f"], np.int32)).to(dev)
r = nd.TransientRenderer(dev, seed=0); r.enable_timing(True)
L = 4096
def run(perm, name):
    o = torch.from_numpy(np.ascontiguousarray(o_np[perm])).to(dev)
    n = torch.tensor(np.tile(np.array([0, 0, 1], np.float32), (L, 1)), device=dev)
    data, _ = r.render_transient(o, n, v, f, 20000, 0.625, 1.625, 2.0 ** -9, seed=1)
    w = torch.ones_like(data); grad = torch.zeros((v.shape[0], 3), dtype=torch.float64, device=dev)
    def step():
        r.render_gradient(o, n, v, f, 20000, 0.625, 1.625, 2.0 ** -9, data=data, weight=w, refine_scale=10, sigma_bin=1, testing_flag=1, loss_flag=0, gradient=grad, zero_gradient=True)
    for _ in range(10): step()
    torch.cuda.synchronize(); r.timing_reset(); t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize()
    print("%-28s %.4f ms/step  kernels %s" % (name, 1e3 * (time.perf_counter() - t0) / 40, [round(float(x), 4) for x in r.timing_mean_ms()[0]]))
idx = np.arange(L)
for rep in range(2):
    run(idx, "natural (row-major)")
    run((idx * 2731) % L, "stride 2731 mod 4096")
    run(np.random.RandomState(1).permutation(L), "random")
    # Morton / Z-order of the 64x64 grid
    x, y = idx % 64, idx // 64
    def spread(a):
        a = (a | (a << 4)) & 0x0F0F; a = (a | (a << 2)) & 0x3333; a = (a | (a << 1)) & 0x5555; return a
    mort = np.argsort(spread(x) | (spread(y) << 1))
    run(mort, "Morton order")
    run(idx[::-1].copy(), "reversed")
    run(mort[::-1].copy(), "Morton reversed")
    r2 = (x - 31.5) ** 2 + (y - 31.5) ** 2
    run(np.argsort(-r2, kind="stable"), "far from the centre first")
    run(np.argsort(r2, kind="stable"), "centre first")
    blk = np.argsort((y // 16) * 4096 + (x // 32) * 1024 + (y % 16) * 32 + (x % 32), kind="stable")
    run(blk, "32x16 blocks, row-major")
    run(np.argsort(x * 64 + y, kind="stable"), "column-major")
    # workgroup i runs on XCD i % 8: hand every XCD its own contiguous eighth of the Z-order curve
    k = np.arange(L)
    run(mort[(k % 8) * (L // 8) + k // 8], "Morton, one eighth of the curve per XCD")
    run(idx[(k % 8) * (L // 8) + k // 8], "row-major, one eighth per XCD")
    hil = None
