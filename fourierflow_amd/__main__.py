"""`python -m fourierflow_amd {train,test,predict} ...` -- see fourierflow_amd/cli.py."""
from .cli import main

main()
