from dualpixelface_amd.losses import COSINELoss  # noqa: F401
