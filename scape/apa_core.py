"""Import path of the pickled result class: ``scape.apa_core.Parameters``
(what ``merge_pa`` imports, reference junction_handler.py:7)."""
from scape_amd.apa_core import Parameters, infer, infer_pa, _infer_pa, to_parameters  # noqa: F401
