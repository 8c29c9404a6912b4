"""The lean correctly rounded sqrt / reciprocal / division of csrc/nlos_device.h return the bits of hipcc's IEEE forms.

The grid trace (forward_grid.hip, NCM = 0) evaluates its three square roots and five divisions per ray as the bare
refinement sequences -- sqrt_cr / sqrt_cr0 / rcp_cr / div_by -- instead of the compiler's 17- / 12-instruction forms.
That keeps the numeric contract (IEEE fp32 as written) only if the results are THE SAME BITS on the
range the callers guarantee.  v_sqrt_f32 / v_rsq_f32 / v_rcp_f32 are deterministic, so tools/exact_math_check.hip
sweeps, on this chip, every bit pattern of the guarded range for the unary functions and 2^k random + structured operand
pairs for the division against the compiler's own sqrtf and `/` (profiles/r05_exact_math.json holds the 2^33-pair run).
"""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_lean_sqrt_rcp_div_are_bit_identical_to_ieee(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "exact_math_check")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result",
                    os.path.join(ROOT, "tools", "exact_math_check.hip"), "-o", exe], check=True, timeout=600)
    out = subprocess.run([exe, str(1 << 30)], check=True, capture_output=True, text=True, timeout=600).stdout
    res = json.loads(out.strip().splitlines()[-1])
    # unary functions: every float of the guarded range (2^-60 ... 2^61), and the sample map's T in [2^-24, 1) and T = 0
    for rng in ("guarded_range", "unit_interval_2^-24..1"):
        assert res[rng]["inputs"] > 2e8
        for fn in ("sqrt_cr", "sqrt_cr0", "rcp_cr"):
            assert res[rng][fn]["differing"] == 0, (rng, fn, res[rng][fn])
    assert res["sqrt_cr0(+-0)"]["differing"] == 0
    # the sweep can fail: outside the guarded range the bare sequences are NOT the IEEE results
    assert res["all_normal"]["sqrt_cr"]["differing"] > 0 and res["all_normal"]["rcp_cr"]["differing"] > 0
    # division: both operand regimes of the kernels (nlos_device.h: kLeanExp*, kLeanNum* / kLeanDen*)
    for reg in ("division_1", "division_2"):
        assert res[reg]["pairs"] == 1 << 30
        assert res[reg]["div_lean"]["differing"] == 0 and res[reg]["div_by(rcp_refined)"]["differing"] == 0, res[reg]
