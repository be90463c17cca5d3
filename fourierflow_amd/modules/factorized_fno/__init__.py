from .grid_2d import FNOFactorized2DBlock, SpectralConv2d  # noqa: F401
from .mesh_3d import FNOFactorizedMesh3D  # noqa: F401
from .mesh_2d import FNOFactorizedMesh2D  # noqa: F401
