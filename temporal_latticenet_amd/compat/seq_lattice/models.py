from temporal_latticenet_amd.lattice_modules import *  # noqa: F401,F403
from temporal_latticenet_amd.seq_modules import *  # noqa: F401,F403
from temporal_latticenet_amd.models import LNN_SEQ, summary  # noqa: F401
