from temporal_latticenet_amd.compat_scatter import scatter_add, scatter_max, scatter_mean  # noqa: F401
