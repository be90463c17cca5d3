from .grid_2d import FNOZongyi2DBlock  # noqa: F401
from .grid_plus_2d import FNOPlus2DBlock  # noqa: F401
from .mesh_2d import FNOMesh2D  # noqa: F401
from .mesh_3d import FNOMesh3D  # noqa: F401
