"""bench.py reports PMC counters (roofline.traffic / roofline.issue) only from a profile taken on THIS build's kernel
sources: profiles/pmc_summary.json carries `stamp.source_sha256_16` (tools/round_summary.py), and a summary of other
sources -- or an unstamped one, like every summary before round 4 -- is reported as stale, never as numbers."""
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from nlos_surface_optimization_amd import _lib  # noqa: E402


def test_source_stamp_is_a_hash_of_the_kernel_sources():
    s = _lib.source_stamp()
    assert re.fullmatch(r"[0-9a-f]{16}", s["source_sha256_16"])
    assert s == _lib.source_stamp()                                   # pure
    assert s["lib_sha256_16"] is None or re.fullmatch(r"[0-9a-f]{16}", s["lib_sha256_16"])


def test_counters_of_another_build_are_refused():
    here = _lib.source_stamp()
    good = {"kernel": "k_forward", "L": 4096, "F": 4902, "hbm_bytes_per_launch": 1.5e8, "issue": {"valu_busy": {"central": 0.9}},
            "stamp": dict(here)}
    t, issue, hs, ps = bench.counters_of_this_build(good, "k_forward", True, 4096, 4902)
    assert t == 1.5e8 and issue == good["issue"] and hs == here and ps["source_sha256_16"] == here["source_sha256_16"]
    for bad_stamp in ({"source_sha256_16": "0" * 16}, {}, None):
        bad = dict(good, stamp=bad_stamp)
        t, issue, _, _ = bench.counters_of_this_build(bad, "k_forward", True, 4096, 4902)
        assert t is None and set(issue) == {"stale"} and here["source_sha256_16"] in issue["stale"]
    # another workload, another kernel, a multi-GPU or side measurement: nothing to report, and no stale note either
    for args in (("k_gradient", True, 4096, 4902), ("k_forward", False, 4096, 4902), ("k_forward", True, 1024, 4902)):
        assert bench.counters_of_this_build(good, *args)[:2] == (None, None)
    assert bench.counters_of_this_build(None, "k_forward", True, 4096, 4902)[:2] == (None, None)
