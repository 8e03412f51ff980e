"""harness stand-in: the reference's cfgParser.py does `hjson.loads(text)`"""
from temporal_latticenet_amd.cfg import loads  # noqa: F401
