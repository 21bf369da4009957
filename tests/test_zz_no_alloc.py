"""ppo2hip_adam_step_static — the entry point that allocates nothing — against ppo2hip_adam_step, through the
raw C ABI: same arithmetic, scratch in the library's own device array instead of a caller workspace.  Two calls in
a row on one GPU, the second with a smaller arena (the array keeps the first call's contents), bit for bit."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(L, check, torch, P, t, seed, max_norm, grad_scale, static):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    theta = torch.randn(P, generator=g, device="cuda")
    grads = torch.randn(P, generator=g, device="cuda") * 0.3
    m = torch.randn(P, generator=g, device="cuda") * 0.01
    v = torch.rand(P, generator=g, device="cuda") * 1e-3
    norm = torch.full((2,), -1.0, device="cuda")
    ptr = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if static:
        check(L.ppo2hip_adam_step_static(ptr(theta), ptr(grads), ptr(m), ptr(v), P, 3e-4, t, 0.9, 0.999, 1e-5, grad_scale,
                                         max_norm, ptr(norm), stream), "adam_step_static")
    else:
        ws = torch.zeros(int(L.ppo2hip_adam_scratch_bytes(P)), dtype=torch.uint8, device="cuda")
        check(L.ppo2hip_adam_step(ptr(theta), ptr(grads), ptr(m), ptr(v), P, 3e-4, t, 0.9, 0.999, 1e-5, grad_scale,
                                  max_norm, ptr(ws), stream), "adam_step")
        norm = ws[:8].view(torch.float32).clone()
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in (theta, m, v, norm)]


def test_adam_step_static_matches_adam_step_on_two_calls_in_a_row():
    import torch
    from baselines_amd import _lib
    L, check = _lib.lib, _lib.check
    # first call: a large arena (separate norm launch writes all 256 partial sums); second: a small one, for which
    # every block sums the norm itself and the stale partial sums must not be read; third: no clipping at all
    cases = [(1_684_000, 7, 11, 0.5, 1.0), (11_085, 8, 12, 0.5, 0.5), (777, 9, 13, 0.0, 1.0)]
    for P, t, seed, max_norm, scale in cases:
        got = _run(L, check, torch, P, t, seed, max_norm, scale, static=True)
        want = _run(L, check, torch, P, t, seed, max_norm, scale, static=False)
        for a, b, name in zip(got[:3], want[:3], ("theta", "m", "v")):
            assert np.array_equal(a, b), (P, name)
        assert got[3][1] == want[3][1], (P, "applied scale")
        if max_norm > 0:
            assert got[3][0] == want[3][0] and got[3][0] > 0, (P, "norm")
    # sizes beyond the stated maximum are refused with a message, before any launch
    one = ctypes.c_void_p(256)
    rc = L.ppo2hip_adam_step_static(one, one, one, one, (1 << 26) + 1, 3e-4, 1, 0.9, 0.999, 1e-5, 1.0, 0.5, None, None)
    assert rc == _lib.EINVAL and b"maximum" in L.ppo2hip_last_error()
