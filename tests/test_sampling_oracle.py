"""The counter-based generator of the sampling oracle against the published known-answer vectors
of Philox4x32-10 (Random123 distribution, kat_vectors: zero, all-ones and pi-digit inputs), and the
ranges / moments of the derived streams."""
import numpy as np

from oracle import sampling


def test_philox4x32_10_known_answers():
    kat = [
        ([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0),
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        got = sampling.philox4x32_10(np.array(ctr, np.uint32), key)
        assert [int(x) for x in got] == want


def test_streams():
    u = sampling.uniforms(50000, 5, 7, 3)
    assert u.dtype == np.float32 and 0.0 < u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.003
    z = sampling.normals(50000, 6, 7, 3)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.02 and abs((z.astype(np.float64) ** 4).mean() - 3) < 0.1
    assert not np.array_equal(z, sampling.normals(50000, 6, 7, 4))
    np.testing.assert_array_equal(z[:100], sampling.normals(100, 6, 7, 3))     # rows do not depend on B


def test_done_bootstrap_rule():
    r = np.array([[1.0, 1, 1, 1, 1], [-1.0, 1, 1, 1, 1], [1.0, 1, 1, 1, 1]], np.float32)
    v = np.full((3, 5), 2.0, np.float32)
    out = sampling.done_bootstrap(r, [True, True, False], v, [0.5] * 5)
    np.testing.assert_array_equal(out, [[2, 2, 2, 2, 2], [-1, 1, 2, 2, 2], [1, 1, 1, 1, 1]])
