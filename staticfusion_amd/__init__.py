"""staticfusion_amd — MI355X-native (gfx950, hand-written HIP) build of ONE hot path of
raluca-scona/staticfusion: the coupled odometry + static/dynamic segmentation solver
(`class StaticFusion`: runSolver / buildSegmImage / kMeans3DCoord and the pyramids they read).

The product is the C-ABI shared library `staticfusion_amd/csrc/libsf_hip.so` declared in
`include/sf.h`; this package only loads it (ctypes) and offers numpy plumbing for tests and the
bench.  There is NO CPU fallback: `load()` raises if the HIP library is missing.
"""
import os

from ._capi import Api, Solver, SurfelMap, SfParams, SfFrameStats, SfError  # noqa: F401
from ._capi import STATUS_EIG_SKIPPED, STATUS_EMPTY_LEVEL, STATUS_SYNC_TIMEOUT  # noqa: F401
from . import _capi as capi  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("SF_HIP_LIB", os.path.join(_HERE, "csrc", "libsf_hip.so"))  # override: A/B builds

_api = None


def load():
    """Bind libsf_hip.so. Fails loudly (OSError / AttributeError) when the extension is absent."""
    global _api
    if _api is None:
        if not os.path.exists(LIB):
            raise OSError(
                "HIP extension %s not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C staticfusion_amd/csrc`). There is no CPU fallback." % LIB
            )
        _api = Api(LIB, "sf_")
    return _api
