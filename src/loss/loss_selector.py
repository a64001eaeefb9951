from dualpixelface_amd.losses import loss_selector  # noqa: F401  (same contract as the reference's loss_selector)
