"""`python -m fourierflow_amd.train CONFIG.yaml [overrides...] [--steps N ...]` -- shorthand for
`python -m fourierflow_amd train ...` (the counterpart of `fourierflow train`, reference commands/train.py:27-148);
all options are those of :func:`fourierflow_amd.cli.train`."""
import typer

from .cli import train


def main():
    typer.run(train)


if __name__ == "__main__":
    main()
