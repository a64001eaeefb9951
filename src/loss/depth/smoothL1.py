from dualpixelface_amd.losses import SMOOTHL1Loss  # noqa: F401
