# Plugin entry for `model_selector` (run_path('src/model/stereodpnet/mainmodel.py')['STEREODPNET'](option)).
# The implementation lives in the dualpixelface_amd package (HIP kernels behind libdpf_hip.so).
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..'))
if _root not in sys.path:
    sys.path.insert(0, _root)

from dualpixelface_amd.plugin import STEREODPNET  # noqa: E402,F401
