from .grid_2d_markov import Grid2DMarkovExperiment  # noqa: F401
