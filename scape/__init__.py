"""``scape`` command surface for the MI355X build: only ``infer_pa`` is provided here
(reference console script ``scape = scape:main``, pyproject.toml:34-35; the other five
sub-commands of the reference are outside this build's scope)."""
from .cli import cli, display_paper_info


def main():
    display_paper_info()
    cli(prog_name="scape")
