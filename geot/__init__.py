"""Alias so that existing call sites (`import geot; geot.index_scatter(...)`) pick up the MI355X
engine unchanged.  Everything lives in :mod:`geot_amd`."""
from geot_amd import *  # noqa: F401,F403
from geot_amd import __all__, __version__, hip  # noqa: F401
