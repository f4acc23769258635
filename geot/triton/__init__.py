"""`geot.triton` of the reference (geot/triton/__init__.py:1-5) - same launcher names, served by the HIP
engine; nothing here uses Triton.  See :mod:`geot_amd.comparators`."""
from geot_amd.comparators import *  # noqa: F401,F403
from geot_amd.comparators import __all__  # noqa: F401
