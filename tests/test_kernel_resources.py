"""Registers, scratch and occupancy of the gfx950 kernels, as the compiler reports them at build time
(``-Rpass-analysis=kernel-resource-usage`` -> ``basq_amd/csrc/build/<unit>.resources.txt``, ``basq_amd._build.kernel_resources``).

No GPU needed.  Why a test: these numbers change silently.  In round 5 the squared-covariance block sums went from two waves per
SIMD to one -- two registers over 256 -- and took twice as long with every parity test green; forcing two waves on ALL of that
kernel's instantiations then made the wide ones spill 100-400 bytes per lane.  The hot kernels' occupancy is pinned here, and no
kernel may use scratch memory (one legacy kernel excepted)."""
import os
import re

import pytest

from basq_amd import _build


@pytest.fixture(scope="module")
def res():
    import torch

    have_reports = all(os.path.exists(_build.resources_path(s)) for s in _build.SOURCES)
    if torch.cuda.device_count() > 0 and not have_reports:
        # a GPU box: the reports under csrc/build/ do not travel with the tree, and compiling here would relink the
        # library over the one this very process may have mapped
        pytest.skip("compiler resource reports are checked where the library is built (the CPU box)")
    if not have_reports or _build.needs_build():
        _build.build(force=not have_reports, verbose=False)
    r = _build.kernel_resources()
    assert len(r) > 200, "resource reports missing: python -m basq_amd._build --force"
    return r


def _find(res, pattern):
    hits = {k: v for k, v in res.items() if re.search(pattern, k)}
    assert hits, f"no kernel matches {pattern}"
    return hits


# mangled-name pattern -> (minimum waves per SIMD, why it matters)
HOT = {
    r"15blocksum_kernelILi3ELi0ELi4ELi2ELb0E": (3, "block sums at d = 7..10 with per-candidate kernel weights (WSABI-L): three waves per SIMD"),
    r"15blocksum_kernelILi3ELi0ELi4ELi[12]ELb1E": (4, "headline block sums (RBF, d = 10, no kernel weights): 128 registers = FOUR waves per SIMD "
                                                     "(round 6; the general form needs 161)"),
    r"15blocksum_kernelILi9ELi1ELi2ELi2E": (3, "config 4's block sums (Matern-5/2, d = 32)"),
    r"15blocksum_kernelILi[1-9]ELi[012]ELi[24]ELi2E": (3, "every other dimension up to d = 34 (round 5: d = 3..6 and d = 15..18 sat at two waves)"),
    r"15blocksum_kernelILi(2ELi1|6ELi[12]|[78]ELi[012]|9ELi[02]|10ELi0)ELi[24]ELi2ELb1E": (4, "the instantiations without per-candidate kernel weights "
                                                                                              "that cross the 128-register line (round 6)"),
    r"18blocksum_sq_kernelILi3ELi0ELi4ELb[01]E": (2, "WSABI-M's squared covariance at config 5's shape, both variants"),
    r"28bidiag_reflectors_reg_kernelILi4ELi7E": (4, "one work-group of 16 waves on ONE compute unit: below 4 the launch fails"),
    r"28bidiag_reflectors_reg_kernelILi4ELi[24]E": (4, "the same for the smaller shapes"),
    r"25car_eliminate_ring_kernelILi7ELi16E": (4, "one work-group of 16 waves"),
    r"18skinny_gemm_kernelILi6ELi1ELi2ELb[01]E": (2, "the range finder's products at q = 99"),
    r"21bidiag_cluster_kernelILi8ELi4ELi8E": (2, "8 waves per work-group, one work-group per CU"),
    r"26car_eliminate_gring_kernelILi7ELi4ELi8E": (2, "8 waves per work-group"),
    r"19cholqr_fused_kernelILb[01]E": (2, "512 threads per work-group"),
}

SCRATCH_ALLOWED = {"_Z19chol_inv_lds_kernelPdiS_Pidi": 32}        # fallback Cholesky (q > 142 path of chol_inv): 8 spilled doubles


@pytest.mark.parametrize("pattern", sorted(HOT))
def test_hot_kernel_occupancy(res, pattern):
    want, why = HOT[pattern]
    for name, v in _find(res, pattern).items():
        assert v["occupancy"] >= want, f"{name}: {v['occupancy']} waves per SIMD (VGPRs {v['vgprs']} + AGPRs {v['agprs']}), want >= {want}: {why}"
        assert v["scratch"] <= SCRATCH_ALLOWED.get(name, 0), f"{name}: {v['scratch']} bytes of scratch per lane"


def test_no_kernel_spills(res):
    bad = {k: v["scratch"] for k, v in res.items() if v.get("scratch", 0) > SCRATCH_ALLOWED.get(k, 0)}
    assert not bad, f"kernels with scratch memory (spilled registers): {bad}"
    assert all(v.get("vgpr_spill", 0) == 0 for k, v in res.items() if k not in SCRATCH_ALLOWED)
