from .grid_2d_markov import Grid2DMarkovExperiment  # noqa: F401
from .grid_2d_rollout import Grid2DRolloutExperiment  # noqa: F401
from .structured_mesh import StructuredMeshExperiment  # noqa: F401
