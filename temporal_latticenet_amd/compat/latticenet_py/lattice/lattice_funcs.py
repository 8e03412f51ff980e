from temporal_latticenet_amd.lattice_modules import Im2RowIndicesLattice, Im2RowLattice  # noqa: F401

__all__ = ["Im2RowLattice", "Im2RowIndicesLattice"]
