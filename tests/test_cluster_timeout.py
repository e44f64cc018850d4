"""A rendezvous timeout of the cluster build must not poison the stream.

A workgroup of a cluster that is scheduled late -- another kernel held its CU -- makes the others give up waiting
(SF_STATUS_SYNC_TIMEOUT). They go on with stale words, the late one later finds granules of the wrong epoch: whatever
any of them holds at the end of the frame is garbage. sf_debug_stall_rank reproduces exactly that (one rank idles 300 ms,
the spin bound is lowered to a few milliseconds). Asserted:
  * the frame reports the status, whichever rank was late -- also rank 0, the one that writes the stream's results;
  * T_odometry, the twists, b, the K-means centres and the per-cluster residuals are bit for bit those of the last good frame;
  * the status is sticky: further frames do nothing but report it, until sf_clear_sync_timeout;
  * after that the stream solves again, bit-identical to a handle that never saw the failed frames.
"""
import numpy as np
import pytest

from conftest import driver_params, make_solver
from staticfusion_amd import STATUS_SYNC_TIMEOUT

pytestmark = pytest.mark.gpu


def _state(s):
    return [np.array(x).copy() for x in (s.T(), s.twist(), s.twist_old(), s.b(), s.kmeans_centres(), s.cluster_residuals())]


@pytest.mark.parametrize("late_rank", [0, 1, 5])
def test_timeout_keeps_the_last_good_state_and_is_sticky(hip_auto, pair, late_rank):
    api = hip_auto.with_variant("cluster")
    pr = pair(seed=7, sphere=True, rows=120, cols=160)
    pr2 = pair(seed=8, sphere=True, rows=120, cols=160)
    s = make_solver(api, 120, 160, driver_params(api), pr)
    assert s.variant()[2] >= 6  # workgroups per stream
    s.process_frame(0)
    assert s.stats().status & STATUS_SYNC_TIMEOUT == 0
    good = _state(s)

    s.debug_stall_rank(late_rank, stall_ms=300.0, spin_limit=2000)
    s.set_current(0, *pr2["new"])
    s.set_prediction(0, *pr2["old"])
    s.process_frame(1)
    assert s.stats().status & STATUS_SYNC_TIMEOUT
    for a, b in zip(good, _state(s)):
        assert np.array_equal(a, b, equal_nan=True)

    s.debug_stall_rank(-1)  # nobody is late any more: the stream still refuses to work
    s.process_frame(2)
    assert s.stats().status & STATUS_SYNC_TIMEOUT
    for a, b in zip(good, _state(s)):
        assert np.array_equal(a, b, equal_nan=True)

    s.clear_sync_timeout()
    s.process_frame(3)
    assert s.stats().status & STATUS_SYNC_TIMEOUT == 0
    ref = make_solver(api, 120, 160, driver_params(api), pr)
    ref.process_frame(0)
    ref.set_current(0, *pr2["new"])
    ref.set_prediction(0, *pr2["old"])
    ref.process_frame(3)
    for a, b in zip(_state(ref), _state(s)):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(ref.labels(0), s.labels(0)) and np.array_equal(ref.b_image(), s.b_image())
    assert (ref.stats().n_outer, ref.stats().n_irls) == (s.stats().n_outer, s.stats().n_irls)


def test_the_other_builds_have_nothing_to_clear(hip_auto, pair):
    s = make_solver(hip_auto.with_variant("throughput"), 60, 80, driver_params(hip_auto), pair(seed=3, rows=60, cols=80))
    s.clear_sync_timeout()  # no rendezvous: a no-op
    with pytest.raises(Exception):
        s.debug_stall_rank(0, 1.0, 100)
    s.process_frame(0)
    assert s.stats().status & STATUS_SYNC_TIMEOUT == 0
