from temporal_latticenet_amd.lattice import HashTable, Lattice, ModelParams  # noqa: F401
