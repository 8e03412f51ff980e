from temporal_latticenet_amd.lattice_modules import *  # noqa: F401,F403
from temporal_latticenet_amd.lattice_modules import __all__  # noqa: F401
