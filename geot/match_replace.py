"""Alias of geot_amd.match_replace (the reference's geot.match_replace.pattern_transform entry point)."""
from geot_amd.match_replace import pattern_transform, rewrite_graph  # noqa: F401
