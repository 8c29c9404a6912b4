#!/bin/bash
# tools/build_stamps.sh without the hipcc time on the GPU box: runs the stamped library prebuilt HERE by
#   AB_FILES="forward_grid nlos_api bvh_build" tools/ab_prebuild.sh "-DNLOS_BUILD_STAMPS -DNLOS_FWD_STAMPS"
# (build/ab/0/libnlos_hip.so) on the metric workload (NLOS_STAMP_GRID=64 NLOS_STAMP_NS=20000) or whatever the variables say.
set -e
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/nlos_stamped && cp -r nlos_surface_optimization_amd /tmp/nlos_stamped && cp build/ab/${1:-0}/libnlos_hip.so /tmp/nlos_stamped/libnlos_hip.so
cd /tmp && ln -sf "$GRAFT_REPO_ROOT/tests" tests 2>/dev/null || true
python3 - <<'PY'
import sys, importlib.util, numpy as np, torch, os
sys.path.insert(0, "/tmp")
spec = importlib.util.spec_from_file_location("nlos_stamped", "/tmp/nlos_stamped/__init__.py", submodule_search_locations=["/tmp/nlos_stamped"])
m = importlib.util.module_from_spec(spec); sys.modules["nlos_stamped"] = m; spec.loader.exec_module(m)
from nlos_stamped import device as nd
d = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/bunny_5k.npz"))
dev = torch.device("cuda", 0)
r = nd.TransientRenderer(dev)
vv, ff = d["v"], d["f"]
v = torch.from_numpy(vv).to(dev); f = torch.from_numpy(ff).to(dev)
ns = int(os.environ.get("NLOS_STAMP_NS", "20000"))
G = int(os.environ.get("NLOS_STAMP_GRID", "64"))
stride = int(os.environ.get("NLOS_STAMP_STRIDE", "1"))          # > 1: every stride-th source (one rank's strided shard)
g = torch.linspace(-0.25, 0.25, G, device=dev); o = torch.stack([g.repeat(G), g.repeat_interleave(G), torch.zeros(G * G, device=dev)], 1).contiguous(); n = torch.tensor([[0, 0, 1.0]] * (G * G), device=dev)
o = o[::stride].contiguous(); n = n[::stride].contiguous()
for _ in range(3):
    r.render_transient(o, n, v, f, ns, 0.625, 1.625, 2.0 ** -9, keep_visibility=True)
torch.cuda.synchronize()
PY
