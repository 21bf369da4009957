"""MI355X-native PPO2 rollout-to-update path behind the `baselines.ppo2` surface.

Python here is host glue only: every array computation of the path runs in the
hand-written HIP kernels of ``libppo2hip.so`` (see ``include/ppo2hip.h``), reached
through ctypes.  torch supplies device memory, streams and ``torch.distributed``.
"""
__version__ = "0.1.0"
