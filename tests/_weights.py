"""Test-side alias of upflow_pytorch_amd.synthetic (deterministic weights / frames shared by the golden
generator, the tests, bench.py and smoke()); the recipe itself lives in the package so that product entry
points never import from tests/."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upflow_pytorch_amd.synthetic import *  # noqa: F401,F403,E402
from upflow_pytorch_amd.synthetic import _EST, _CTX, _PYR, _SGU, _SGU_OUT  # noqa: F401,E402
