"""fourierflow_amd -- MI355X-native (gfx950) F-FNO spectral-layer stack.

Python host over the C-ABI HIP library declared in include/ffno.h; mirrors the operator API of
alasdairtran/fourierflow's ``fourierflow.modules`` for the factorized-FNO hot path.
"""
__version__ = "0.1.0"
