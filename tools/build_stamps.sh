#!/bin/bash
# diagnostic: per-phase cycles of k_build_bvh and k_forward_grid, work counters, and the shader clock the forward kernel runs at
# (NLOS_CLOCK_JSON=<file>: written as JSON; NLOS_STAMP_GRID=64 NLOS_STAMP_NS=20000: the metric workload).
# Builds a stamped copy of the library in /tmp, never the shipped one.
set -e
cd "$GRAFT_REPO_ROOT"
cp -r nlos_surface_optimization_amd /tmp/nlos_stamped && cp -r include /tmp/include
make -s -C /tmp/nlos_stamped/csrc clean >/dev/null; make -s -C /tmp/nlos_stamped/csrc -j4 EXTRA="-DNLOS_BUILD_STAMPS -DNLOS_FWD_STAMPS $NLOS_STAMP_EXTRA"
cd /tmp && ln -sf "$GRAFT_REPO_ROOT/tests" tests 2>/dev/null || true
python3 - <<'PY'
import sys, importlib.util, numpy as np, torch
sys.path.insert(0, "/tmp")
spec = importlib.util.spec_from_file_location("nlos_stamped", "/tmp/nlos_stamped/__init__.py", submodule_search_locations=["/tmp/nlos_stamped"])
m = importlib.util.module_from_spec(spec); sys.modules["nlos_stamped"] = m; spec.loader.exec_module(m)
from nlos_stamped import device as nd
import os
d = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/bunny_5k.npz"))
dev = torch.device("cuda", 0)
r = nd.TransientRenderer(dev)
vv, ff = d["v"], d["f"]
sub = int(os.environ.get("NLOS_STAMP_SUBDIV", "0"))
if sub:
    from nlos_stamped import mesh_io
    vv, ff = mesh_io.subdivide(vv, ff, sub)
v = torch.from_numpy(vv).to(dev); f = torch.from_numpy(ff).to(dev)
ns = int(os.environ.get("NLOS_STAMP_NS", str(4 * ff.shape[0])))
G = int(os.environ.get("NLOS_STAMP_GRID", "32"))
g = torch.linspace(-0.25, 0.25, G, device=dev); o = torch.stack([g.repeat(G), g.repeat_interleave(G), torch.zeros(G * G, device=dev)], 1).contiguous(); n = torch.tensor([[0, 0, 1.0]] * (G * G), device=dev)
for _ in range(3):
    r.render_transient(o, n, v, f, ns, 0.625, 1.625, 2.0 ** -9)
torch.cuda.synchronize()
PY
