"""The -DSF_FAST_WEIGHTS=0 build (staticfusion_amd/csrc/libsf_hip_precise.so: IEEE division / square root in the
per-pixel IRLS weights instead of the 1-ulp hardware v_rcp_f32 / v_rsq_f32) as a TESTED build: the same bars as the
product against the oracle on both frame-kernel variants, and the measured distance between the two builds -- the cost of
the fast path in result terms (DESIGN.md section 6 quotes tools/parity_report.py for the full table).
Reference arithmetic under test: FrontEnd.cpp:619-637 (Cauchy weights), SegmentationBackground.cpp:133-174 (b-solve)."""
import os

import numpy as np
import pytest

from conftest import config2_params, driver_params, make_solver, trace_array
from staticfusion_amd.synth import pose_delta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["throughput", "latency"])
def precise(request):
    import staticfusion_amd as sf

    path = os.path.join(os.path.dirname(sf.LIB), "libsf_hip_precise.so")
    return sf.Api(path, "sf_").with_variant(request.param)  # OSError if the build is missing: no skip


def solve(api, mk, pr):
    s = make_solver(api, 240, 320, mk(api), pr)
    s.build_pyramid(True)
    s.run_solver(True)
    s.build_segm_image()
    return s


@pytest.mark.parametrize("sphere", [False, True])
def test_precise_build_against_the_oracle_and_the_fast_build(precise, ora, pair, sphere):
    import staticfusion_amd as sf

    mk = (lambda a: driver_params(a)) if sphere else (lambda a: config2_params(a, levels=3))
    fast = sf.load().with_variant(precise.default_variant)
    for seed in (1234, 1236):
        pr = pair(seed=seed, sphere=sphere, rows=240, cols=320)
        sp, sf_, so = solve(precise, mk, pr), solve(fast, mk, pr), solve(ora, mk, pr)
        a, b, c = sp.stats(), so.stats(), sf_.stats()
        assert (a.n_outer, a.n_irls, a.kmeans_iters, a.status, a.pixel_iters) == (b.n_outer, b.n_irls, b.kmeans_iters, b.status, b.pixel_iters)
        rot, trans = pose_delta(so.T(), sp.T())
        assert rot <= 1e-4 and trans <= 1e-4
        assert np.abs(trace_array(a, "twist_level") - trace_array(b, "twist_level")).max() < 2e-6
        assert np.abs(trace_array(a, "b_segm") - trace_array(b, "b_segm")).max() < 1e-4
        for L in range(sp.levels):
            assert np.array_equal(sp.labels(L), so.labels(L))
        assert np.array_equal(sp.b_image() > 0.5, so.b_image() > 0.5)
        # fast vs precise: what the hardware reciprocal / reciprocal square root cost in result terms
        rot, trans = pose_delta(sp.T(), sf_.T())
        assert rot <= 2e-6 and trans <= 2e-6, (rot, trans)
        assert (c.n_outer, c.n_irls) == (a.n_outer, a.n_irls)
        assert np.abs(trace_array(a, "b_segm") - trace_array(c, "b_segm")).max() < 1e-4
