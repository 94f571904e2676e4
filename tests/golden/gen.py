"""Deterministic weights / inputs for the golden vectors: the generator itself lives in ``objcavit_amd/synth.py`` (the
benchmark and ``__graft_entry__.smoke()`` load the same seeded weights and must not import from the test tree); the
tests and ``make_golden.py`` keep importing it under this name."""
from objcavit_amd.synth import *  # noqa: F401,F403
from objcavit_amd.synth import _rs  # noqa: F401
