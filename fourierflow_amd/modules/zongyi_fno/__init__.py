from .grid_plus_2d import FNOPlus2DBlock  # noqa: F401
