"""``scape.junction_handler`` import path of the reference (junction_handler.py:28-147)."""
from scape_amd.junction_handler import merge_pa, _merge_pa, merge_gene  # noqa: F401
