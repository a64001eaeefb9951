"""`FaceDPLoader(option, training)` at the path the reference's loader_selector resolves (dataloader/FaceDP/loader.py:79); the
implementation (host readers + device preprocessing kernels) lives in dualpixelface_amd/facedp.py."""
from dualpixelface_amd.facedp import FaceDPBatcher, FaceDPLoader  # noqa: F401
