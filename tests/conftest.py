import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """A fresh checkout has no binaries (they are git-ignored): build the product libraries and the test oracle once.
    hipcc cross-compiles gfx950 without a GPU; on the GPU box the prebuilt files travel with the snapshot."""
    import subprocess

    csrc = os.path.join(ROOT, "staticfusion_amd", "csrc")
    if not (os.path.exists(os.path.join(csrc, "libsf_hip.so")) and os.path.exists(os.path.join(csrc, "libsf_io.so"))):
        subprocess.check_call(["make", "-C", csrc])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (oracle/liboracle.so), built on demand with gcc."""
    from oracle import binding

    binding.build()
    return binding.load()


# Every GPU test that takes the `hip` fixture runs once per build of the frame kernel: the
# driver's green therefore covers `sf_frame_kernel_nt256` (the build bench.py times) as well as `_nt1024`.
# SF_TEST_VARIANTS=throughput (comma separated) narrows the set for a quick run.
HIP_VARIANTS = [v for v in os.environ.get("SF_TEST_VARIANTS", "throughput,latency,cluster").split(",") if v]


@pytest.fixture(scope="session", params=HIP_VARIANTS)
def hip(request):
    """The product library, bound to one named build of the frame kernel (Solver(...) defaults to it through
    sf_create_ex). No fallback: a missing extension or GPU is a test ERROR, not a skip."""
    import staticfusion_amd as sf

    return sf.load().with_variant(request.param)


@pytest.fixture(scope="session")
def hip_auto():
    """The product library with sf_create's own choice of the build (by batch size)."""
    import staticfusion_amd as sf

    return sf.load()


def _pairs():
    from staticfusion_amd.synth import make_pair

    cache = {}

    def get(seed=11, sphere=False, rows=120, cols=160, xi=None):
        key = (seed, sphere, rows, cols, None if xi is None else tuple(xi))
        if key not in cache:
            kw = {} if xi is None else {"xi": xi}
            cache[key] = make_pair(seed=seed, sphere=sphere, out_rows=rows, out_cols=cols, **kw)
        return cache[key]

    return get


@pytest.fixture(scope="session")
def pair():
    return _pairs()


def make_solver(api, rows, cols, params, pair=None, batch=1, variant=None):
    import staticfusion_amd as sf

    try:
        s = sf.Solver(api, rows, cols, batch, params, variant=variant)
    except sf.SfError as e:
        if "SF_VARIANT_CLUSTER: batch too large" in str(e):  # every stream of a cluster handle needs several CUs at once
            pytest.skip("the cluster build serves at most CUs / 8 streams per XCD: batch %d is a throughput / latency case" % batch)
        raise
    if pair is not None:
        for b in range(batch):
            s.set_current(b, *pair["new"])
            s.set_prediction(b, *pair["old"])
    return s


def driver_params(api, kb=1.05, **over):
    p = api.default_params_struct()
    p.kb = kb
    for k, v in over.items():
        setattr(p, k, v)
    return p


def config2_params(api, levels=3, **over):
    p = api.ctor_params_struct()
    p.ctf_levels = levels
    p.segmentation_enabled = 0
    for k, v in over.items():
        setattr(p, k, v)
    return p


def trace_array(st, field, n=None):
    n = st.n_outer if n is None else n
    return np.array([list(getattr(st.outer[i], field)[:]) if hasattr(getattr(st.outer[i], field), "__len__")
                     else getattr(st.outer[i], field) for i in range(n)])
