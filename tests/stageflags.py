"""The stage flags live in the package (carpedeam_amd/stageflags.py: bench.py and smoke() use them too); the tests keep this name."""
from carpedeam_amd.stageflags import *  # noqa: F401,F403
