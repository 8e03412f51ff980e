from temporal_latticenet_amd.lovasz import LovaszSoftmax  # noqa: F401
