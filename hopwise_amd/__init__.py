"""hopwise_amd -- MI355X (gfx950) native engine for hopwise's KG-embedding scoring, negative sampling
and PGPR beam-search path.  See DESIGN.md for the path and its boundary."""

__version__ = "0.1.0"
