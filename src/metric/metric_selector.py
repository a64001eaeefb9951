from dualpixelface_amd.selectors import metric_selector  # noqa: F401
