"""Re-export of the product package's synthetic generators (kept so that tests can say `from oracle import synth`)."""
from agatha_amd.workload import *  # noqa: F401,F403
from agatha_amd.workload import _ACGT  # noqa: F401
