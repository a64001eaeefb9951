"""dualpixelface_amd -- MI355X-native StereoDPNet hot path (HIP kernels behind a C ABI + the reference's plugin surface)."""
from .config import load_option, Option  # noqa: F401


def STEREODPNET(option):
    from .plugin import STEREODPNET as _cls
    return _cls(option)
