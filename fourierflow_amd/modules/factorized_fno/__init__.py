from .grid_2d import FNOFactorized2DBlock, SpectralConv2d  # noqa: F401
