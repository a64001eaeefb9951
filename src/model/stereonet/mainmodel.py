# Plugin entry for `model_selector` (run_path('src/model/stereonet/mainmodel.py')['STEREONET'](option)), SURVEY section 8f rank f4.
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..'))
if _root not in sys.path:
    sys.path.insert(0, _root)

from dualpixelface_amd.plugin import STEREONET  # noqa: E402,F401
