"""Host-side pieces of baselines/a2c/utils.py that the PPO2 path touches.

``discount_with_dones`` keeps the reference's list-in/list-out signature but runs the
scan in the HIP kernel (one env column); the batched entry point the A2C runner needs is
``a2c_returns`` (time-major, all envs in one launch).
"""
import numpy as np


def ortho_init(scale=1.0):
    """Orthogonal initialiser (baselines/a2c/utils.py:25-40): draws from the global
    ``np.random`` stream, so seeding ``np.random`` reproduces the reference's weights."""
    def _ortho_init(shape, dtype=np.float32, partition_info=None):
        shape = tuple(shape)
        if len(shape) == 2:
            flat_shape = shape
        elif len(shape) == 4:   # NHWC conv kernel
            flat_shape = (int(np.prod(shape[:-1])), shape[-1])
        else:
            raise NotImplementedError
        a = np.random.normal(0.0, 1.0, flat_shape)
        u, _, v = np.linalg.svd(a, full_matrices=False)
        q = (u if u.shape == flat_shape else v).reshape(shape)
        return (scale * q[:shape[0], :shape[1]]).astype(np.float32)
    return _ortho_init


def discount_with_dones(rewards, dones, gamma):
    """baselines/a2c/utils.py:154-160 for one env: lists in, list of floats out."""
    import torch
    from .. import hip
    if len(rewards) == 0:
        return []
    r = torch.as_tensor(np.asarray(rewards, np.float32).reshape(-1, 1)).cuda()
    d = torch.as_tensor(np.asarray(dones).astype(np.uint8).reshape(-1, 1)).cuda()
    return hip.discount_with_dones(r, d, None, gamma).cpu().numpy().ravel().tolist()


def a2c_returns(rewards, dones, last_values, gamma):
    """Bootstrapped returns of the A2C runner (baselines/a2c/a2c.py:117-126) for all envs.
    rewards [T,E] float32, dones [T,E] (done AFTER each step), last_values [E]; device tensors."""
    from .. import hip
    return hip.discount_with_dones(rewards, dones, last_values, gamma)


def constant(p):
    return 1


def linear(p):
    return 1 - p
