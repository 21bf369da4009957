"""Parameter initialisation of the reference's layer builders.

Test infrastructure (see ``oracle/__init__.py``).
"""
import numpy as np


def ortho_init(shape, scale=1.0, rng=None):
    """Orthogonal init, ``baselines/a2c/utils.py:25-40``.

    Draws ``np.random.normal`` of the flattened shape from the global stream
    (or ``rng`` when given), takes the SVD factor with the matching shape and
    scales it; 4-D shapes are NHWC conv kernels flattened over their first
    three axes.
    """
    rng = rng or np.random
    shape = tuple(shape)
    if len(shape) == 2:
        flat_shape = shape
    elif len(shape) == 4:
        flat_shape = (int(np.prod(shape[:-1])), shape[-1])
    else:
        raise NotImplementedError
    a = rng.normal(0.0, 1.0, flat_shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == flat_shape else v
    q = q.reshape(shape)
    return (scale * q[:shape[0], :shape[1]]).astype(np.float32)
