"""``scape`` command surface for the MI355X build: only ``infer_pa`` is provided here
(reference console script ``scape = scape:main``, pyproject.toml:34-35; the other five
sub-commands of the reference are outside this build's scope)."""
import time as _time

_t0 = _time.perf_counter()
from .cli import cli, display_paper_info  # noqa: E402

_IMPORT_S = _time.perf_counter() - _t0        # seconds spent importing the package (numpy, click, the host modules)


def main():
    display_paper_info()
    cli(prog_name="scape")
