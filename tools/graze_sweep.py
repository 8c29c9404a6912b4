#!/usr/bin/env python3
"""What does the grazing rule of the numeric contract (include/nlos_contract.h) change against the reference's
rule-free all-faces definition (Embree accepts every den != 0, SMO/transient_and_gradient.cpp:199-206)?

    python tools/graze_sweep.py [grid=10] > profiles/rNN_graze_sweep.json        (CPU only, the oracle; minutes)

For every BASELINE mesh / window (bunny 512 bins, mannequin +-0.35 m 1024 bins, bunny GGX 1024 bins), grid x grid
sources: the oracle's brute-force render with the rule at 2^-5 ... 2^-10 against the same render with the rule off.
Reports rows rel-L2, the worst single row, max-abs / max, gradient rel-L2 and the number of (source, face) pairs whose
accepted-sample count differs."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402


def sources(n, half):
    g = np.linspace(-half, half, n)
    o = np.array([[x, y, 0] for y in g for x in g], np.float32)
    return o, np.tile(np.array([0, 0, 1], np.float32), (o.shape[0], 1))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ratios = [float(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2.0 ** -5, 2.0 ** -6, 2.0 ** -7, 2.0 ** -8, 2.0 ** -9, 2.0 ** -10]
    G = os.path.join(ROOT, "tests", "golden")
    b = np.load(os.path.join(G, "bunny_5k.npz"))
    m = np.load(os.path.join(G, "mannequin.npz"))
    cases = {
        "bunny_5k 512 bins spt 5 (cfg 2/3, metric)": dict(v=b["v"], f=b["f"], half=0.25, lb=0.625, ub=1.625, res=2.0 ** -9, ns=20000, kw={}),
        "mannequin +-0.35 1024 bins spt 19 (cfg 4)": dict(v=m["v"], f=m["f"], half=0.35, lb=0.0, ub=1024 * 2.4e-3, res=2.4e-3, ns=20000, kw={}),
        "bunny_5k GGX alpha 0.3 1024 bins (cfg 5)": dict(v=b["v"], f=b["f"], half=0.25, lb=0.625, ub=1.625, res=2.0 ** -10, ns=20000, kw=dict(ggx_alpha=0.3)),
    }
    out = {"sources": n * n, "contract_ratio": orc.graze_ratio(), "cases": {}}
    for name, c in cases.items():
        o, nrm = sources(n, c["half"])
        rs = np.random.RandomState(3)

        def render(ratio):
            orc.set_graze_ratio(ratio)
            tr, _ = orc.render_transient(o, nrm, c["v"], c["f"], c["ns"], c["lb"], c["ub"], c["res"], seed=0, accel=0, **c["kw"])
            return tr

        t0 = time.time()
        tr_free = render(0.0)
        data = tr_free * (1.0 + 0.3 * rs.standard_normal(tr_free.shape))
        weight = 0.5 + rs.random_sample(tr_free.shape)

        def grad(ratio):
            orc.set_graze_ratio(ratio)
            _, g, _ = orc.render_gradient(o, nrm, c["v"], c["f"], c["ns"], c["lb"], c["ub"], c["res"], data, weight, refine=10,
                                          sigma_bin=1, testing_flag=1, loss_flag=0, seed=0, accel=0, **c["kw"])
            return g

        g_free = grad(0.0)
        rows = {}
        for r in ratios:
            tr = render(r)
            g = grad(r)
            rown = np.linalg.norm(tr - tr_free, axis=1) / np.maximum(np.linalg.norm(tr_free, axis=1), 1e-300)
            rows["2^%d" % int(round(np.log2(r)))] = {
                "rows_rel_l2": float(np.linalg.norm(tr - tr_free) / np.linalg.norm(tr_free)),
                "worst_row_rel_l2": float(rown.max()),
                "max_abs_over_max": float(np.abs(tr - tr_free).max() / tr_free.max()),
                "gradient_rel_l2": float(np.linalg.norm(g - g_free) / np.linalg.norm(g_free)),
                "rows_differing": int((rown > 0).sum()),
            }
            print(name, "ratio 2^%d" % int(round(np.log2(r))), rows["2^%d" % int(round(np.log2(r)))], file=sys.stderr, flush=True)
        out["cases"][name] = {"faces": int(c["f"].shape[0]), "by_ratio": rows, "seconds": round(time.time() - t0, 1)}
    orc.set_graze_ratio(-1.0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
