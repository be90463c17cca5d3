from .grid_2d_markov import Grid2DMarkovExperiment  # noqa: F401
from .structured_mesh import StructuredMeshExperiment  # noqa: F401
