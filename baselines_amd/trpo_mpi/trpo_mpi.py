"""baselines/trpo_mpi/trpo_mpi.py:69-80 is the same segment scan as ppo1's."""
from ..ppo1.pposgd_simple import add_vtarg_and_adv  # noqa: F401
