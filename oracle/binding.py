"""ctypes binding of the CPU ORACLE (oracle/liboracle.so, symbol prefix ``sfo_``).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package (staticfusion_amd) never does.
"""
import os
import subprocess

from staticfusion_amd._capi import Api

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_ORACLE_LIB: another build of the same sources (tests/test_oracle_sanitizers.py points it at liboracle_asan.so)
LIB = os.environ.get("SF_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile the restatement with the committed Makefile (gcc only)."""
    if force or not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return LIB


_api = None


def load():
    global _api
    if _api is None:
        if not os.path.exists(LIB):
            build()
        _api = Api(LIB, "sfo_")
    return _api
