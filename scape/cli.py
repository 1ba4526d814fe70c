"""click group exposing ``infer_pa`` and its consumer ``merge_pa`` (reference cli.py:7-31 registers six
commands; the other four are outside this build's scope)."""
import click

from scape_amd.apa_core import infer_pa, infer_pa_all, prebin
from scape_amd.junction_handler import merge_pa


@click.group()
def cli():
    """SCAPE-APA `infer_pa` on AMD MI355X (HIP kernels behind the reference's CLI)."""


def display_paper_info():
    print()
    print("scape infer_pa (MI355X/HIP build). Method: SCAPE-APA, Cheng, Le, Zhou, Cheng,")
    print("bioRxiv 2024, https://doi.org/10.1101/2024.03.12.584547")
    print()


cli.add_command(infer_pa)
cli.add_command(infer_pa_all)
cli.add_command(merge_pa)
cli.add_command(prebin)
