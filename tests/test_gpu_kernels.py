"""Parity of the HIP kernels (through the C ABI) against the CPU oracle.  GPU only."""
import numpy as np
import pytest
import torch

from oracle import advs as oadvs
from oracle import scan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from baselines_amd import hip as h
    return h


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (32, 32, 2), (64, 64, 64), (33, 17, 5), (128, 64, 16), (257, 70, 19),
                                   (1000, 6, 64), (4096, 64, 17), (65, 64, 3000), (300, 512, 200)])
def test_gemm_f32(hip, M, N, K):
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    # asymmetric operands so a transposed or permuted fragment map cannot pass
    A = (rng.randn(M, K) + np.arange(K)[None, :] * 0.01).astype(np.float32)
    B = (rng.randn(K, N) + np.arange(N)[None, :] * 0.02).astype(np.float32)
    C = hip.gemm(dev(A), dev(B)).cpu().numpy()
    ref = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert np.all(np.abs(C - ref) <= 2e-6 * scale + 1e-30)   # fp32 fma chain, K up to 3000


@pytest.mark.parametrize("M,N,K,positive", [(2048, 512, 3136, False), (2048, 512, 3136, True), (4096, 64, 512, True),
                                            (1024, 192, 4096, True), (1111, 100, 1024, False)])
def test_gemm_f32_wide_shapes_on_the_bf16_pipe(hip, M, N, K, positive):
    """The shapes that take the bf16-piece products (128-row tiles, more than 32 columns, K a multiple of 16): float32
    results from six exact piece products per multiply-add.  Same-signed operands are the adversarial case: nothing
    cancels, so a biased cut (all three pieces truncated) shows as a drift of the whole sum — the mean signed error must
    stay at the float32 chain's level, not only the maximum."""
    rng = np.random.RandomState(M + N + K)
    draw = (lambda *s: rng.rand(*s)) if positive else (lambda *s: rng.randn(*s))
    A = draw(M, K).astype(np.float32)
    B = draw(K, N).astype(np.float32)
    C = hip.gemm(dev(A), dev(B)).cpu().numpy().astype(np.float64)
    again = hip.gemm(dev(A), dev(B)).cpu().numpy()
    assert np.array_equal(C.astype(np.float32), again)          # fixed summation order
    ref = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    rel = (C - ref) / scale
    assert np.abs(rel).max() <= 4e-6                             # float32 pipe on the same data: 4.0e-6 at K = 3136
    assert np.abs(rel).mean() <= 1e-6 and abs(rel.mean()) <= 3e-7


CASES = ["small", "mid", "cartpole_shape", "nodone", "alldone"]


@pytest.mark.parametrize("name", CASES)
def test_gae_golden_bit_exact(hip, golden, name):
    g = lambda k: golden["gae_%s_%s" % (name, k)]
    advs, rets, ori = hip.gae(dev(g("rewards")), dev(g("values")), dev(g("dones")), dev(g("last_values")),
                              dev(g("last_dones")), g("gamma"), float(g("lam")))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(hip.sf01(rets).cpu().numpy(), g("returns_flat"))
    o_advs, o_rets = scan.gae(g("rewards"), g("values"), g("dones"), g("last_values"), g("last_dones"),
                              g("gamma"), float(g("lam")))
    np.testing.assert_array_equal(advs.cpu().numpy(), o_advs)
    np.testing.assert_array_equal(ori.cpu().numpy(), o_rets - g("values"))


@pytest.mark.parametrize("T,E,R", [(1, 1, 1), (7, 3, 1), (256, 300, 1), (37, 129, 5), (128, 1024, 1), (9, 70, 8),
                                   (130, 45, 3), (64, 16, 1), (65, 17, 2), (40, 70000, 1), (33, 9000, 8)])
def test_gae_random_bit_exact(hip, T, E, R):
    rng = np.random.RandomState(T + E + R)
    rewards = rng.randn(T, E, R).astype(np.float32)
    values = rng.randn(T, E, R).astype(np.float32)
    dones = rng.rand(T, E) < 0.1
    last_values = rng.randn(E, R).astype(np.float32)
    last_dones = rng.rand(E) < 0.3
    gamma = np.linspace(0.99, 0.9, R)
    advs, rets, ori = hip.gae(dev(rewards), dev(values), dev(dones), dev(last_values), dev(last_dones), gamma, 0.95)
    o_advs, o_rets = scan.gae(rewards, values, dones, last_values, last_dones, gamma, 0.95)
    np.testing.assert_array_equal(advs.cpu().numpy(), o_advs)
    np.testing.assert_array_equal(rets.cpu().numpy(), o_rets)
    np.testing.assert_array_equal(ori.cpu().numpy(), o_rets - values)


def test_gae_empty_and_bad_shapes(hip):
    z = torch.zeros((0, 4, 1), device="cuda")
    advs, rets, _ = hip.gae(z, z, torch.zeros((0, 4), dtype=torch.bool, device="cuda"),
                            torch.zeros((4, 1), device="cuda"), torch.zeros(4, dtype=torch.bool, device="cuda"), 0.99, 0.95)
    assert advs.shape == (0, 4, 1)
    with pytest.raises(ValueError):
        hip.gae(torch.zeros((3, 4, 1), device="cuda"), torch.zeros((3, 4, 1), device="cuda"),
                torch.zeros((3, 5), dtype=torch.bool, device="cuda"), torch.zeros((4, 1), device="cuda"),
                torch.zeros(4, dtype=torch.bool, device="cuda"), 0.99, 0.95)
    from baselines_amd._lib import Ppo2HipError
    with pytest.raises(Ppo2HipError):   # R beyond the kernel's channel limit is refused, not truncated
        hip.gae(torch.zeros((3, 4, 9), device="cuda"), torch.zeros((3, 4, 9), device="cuda"),
                torch.zeros((3, 4), dtype=torch.bool, device="cuda"), torch.zeros((4, 9), device="cuda"),
                torch.zeros(4, dtype=torch.bool, device="cuda"), 0.99, 0.95)


def test_discount_with_dones(hip, golden):
    rew, dn, lv = golden["a2c_rewards"], golden["a2c_dones"], golden["a2c_last_values"]   # env-major [E,T]
    out = hip.discount_with_dones(dev(rew.T.copy()), dev(dn.T.copy()), dev(lv), 0.99).cpu().numpy().T
    np.testing.assert_array_equal(out, golden["a2c_returns"])
    # known-answer vector of the reference, no bootstrap
    kat = hip.discount_with_dones(dev(np.array([[1.], [2.], [3.], [4.]], np.float32)),
                                  dev(np.array([[0], [0], [1], [0]], np.uint8)), None, 0.9).cpu().numpy().ravel()
    np.testing.assert_array_equal(kat, np.asarray(golden["dwd_kat"], np.float32))
    rng = np.random.RandomState(3)
    T, E = 37, 1500
    rew = rng.randn(T, E).astype(np.float32)
    dn = rng.rand(T, E) < 0.2
    lv = rng.randn(E).astype(np.float32)
    out = hip.discount_with_dones(dev(rew), dev(dn), dev(lv), 0.97).cpu().numpy()
    np.testing.assert_array_equal(out.T, scan.a2c_returns(rew.T, dn.T, lv, 0.97))


@pytest.mark.parametrize("T,E,row,dtype", [(5, 3, (), np.float32), (128, 8, (4,), np.float32), (33, 65, (17,), np.float32),
                                           (16, 20, (5,), np.uint8), (7, 9, (), np.bool_), (4, 6, (84, 84, 4), np.uint8),
                                           (20, 33, (63,), np.float32)])
def test_sf01(hip, T, E, row, dtype):
    rng = np.random.RandomState(1)
    a = (rng.rand(T, E, *row) * 200).astype(dtype)
    np.testing.assert_array_equal(hip.sf01(dev(a)).cpu().numpy(), scan.sf01(a))


def _check_advs(out, ref, advs_ori):
    """Stated tolerance of the scalarisation: the result is a float32 sum of five float32
    products a_k * L divided by L.  The reference takes L from a float32 running mean over
    all N rows, the kernel from a float64 tree sum, so L can differ in its last bits and every
    product then rounds differently: agreement is to float32 rounding OF THE TERMS,
    |diff| <= 4e-7 * sum_k |a_k| (about six float32 roundings), not of a cancelled sum."""
    bound = 4e-7 * np.abs(advs_ori).astype(np.float64).sum(axis=1) + 1e-12
    assert np.all(np.abs(out - ref) <= bound), float(np.max(np.abs(out - ref) / bound))


@pytest.mark.parametrize("name", CASES)
def test_compute_advs_golden(hip, golden, name):
    g = lambda k: golden["gae_%s_%s" % (name, k)]
    values, returns = g("values_flat"), g("returns_flat")
    out = hip.compute_advs([200, 120, 0, 0, 0], dev(values), dev(returns - values)).cpu().numpy()
    _check_advs(out, g("advs_scalar"), returns - values)


def test_compute_advs_large(hip):
    rng = np.random.RandomState(5)
    N = 20000
    values = (rng.randn(N, 5) * 10).astype(np.float32)
    advs_ori = rng.randn(N, 5).astype(np.float32)
    out = hip.compute_advs([200, 120, 0, 0, 0], dev(values), dev(advs_ori)).cpu().numpy()
    _check_advs(out, oadvs.compute_advs([200, 120, 0, 0, 0], values, advs_ori), advs_ori)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_adv_stats(hip, dtype):
    rng = np.random.RandomState(9)
    T, E, B, nmb = 16, 40, 128, 5
    advs = (rng.randn(T, E) * 3 + 1).astype(dtype)
    perm = rng.permutation(T * E).astype(np.int32)
    st = hip.adv_stats(dev(advs), dev(perm), B, nmb, T, E).cpu().numpy()
    flat = scan.sf01(advs)
    for j in range(nmb):
        a = flat[perm[j * B:(j + 1) * B]].astype(np.float64)
        np.testing.assert_allclose(st[j], [a.sum(), (a * a).sum(), B], rtol=1e-12)


@pytest.mark.gpu
def test_compose_permutations(hip):
    rng = np.random.RandomState(3)
    n = 100003
    prev, sig = rng.permutation(n).astype(np.int32), rng.permutation(n).astype(np.int32)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    hip.compose(torch.from_numpy(prev).cuda(), torch.from_numpy(sig).cuda(), out)
    np.testing.assert_array_equal(out.cpu().numpy(), prev[sig])
    with pytest.raises(ValueError):
        hip.compose(torch.from_numpy(prev).cuda(), torch.from_numpy(sig[:10]).cuda(), out)


@pytest.mark.parametrize("shape,dtype", [((1,), np.float32), ((2048,), np.float32), ((129, 5), np.float64), ((40, 70, 3), np.float32)])
def test_math_util_discount(hip, shape, dtype):
    from baselines_amd.common import math_util
    x = np.random.RandomState(sum(shape)).randn(*shape).astype(dtype)
    got = math_util.discount(x, 0.97)
    assert got.dtype == np.float64 and got.shape == x.shape
    np.testing.assert_array_equal(got, scan.discount(x, 0.97))
    assert math_util.discount(np.zeros((0, 3), np.float32), 0.9).shape == (0, 3)
    with pytest.raises(TypeError):
        hip.discount(dev(x.astype(np.float64)), 0.9, out_dtype=torch.float32)


def test_math_util_discount_with_boundaries(hip, golden):
    from baselines_amd.common import math_util
    y = math_util.discount_with_boundaries(golden["dwb_x"], golden["dwb_starts"], 0.9)
    assert y.dtype == golden["dwb_x"].dtype
    np.testing.assert_allclose(y, golden["dwb_y"], rtol=0, atol=1e-6)     # fixture made with weak-scalar NumPy
    rng = np.random.RandomState(4)
    for dtype in (np.float32, np.float64):
        X = rng.randn(300, 6).astype(dtype)
        New = rng.rand(300) < 0.1
        np.testing.assert_array_equal(math_util.discount_with_boundaries(X, New, 0.95),
                                      scan.discount_with_boundaries_legacy(X, New, 0.95))


def test_add_vtarg_and_adv(hip, golden):
    from baselines_amd.ppo1.pposgd_simple import add_vtarg_and_adv
    from baselines_amd.trpo_mpi import trpo_mpi
    assert trpo_mpi.add_vtarg_and_adv is add_vtarg_and_adv
    for name in ["a", "b", "c"]:
        g = lambda k: golden["gae1_%s_%s" % (name, k)]
        nextv = g("last_value") * (1 - bool(g("last_done")))
        seg = dict(rew=g("rew"), vpred=g("vpred"), new=g("new"), nextvpred=nextv)
        add_vtarg_and_adv(seg, 0.99, 0.95)
        a64, r64 = scan.gae_single_env_f64(g("rew"), g("vpred"), g("new"), nextv, 0.99, 0.95)
        np.testing.assert_array_equal(seg["adv"], a64)
        np.testing.assert_array_equal(seg["tdlamret"], r64)
        np.testing.assert_allclose(seg["adv"], g("adv"), rtol=0, atol=1e-6)
        np.testing.assert_allclose(seg["tdlamret"], g("tdlamret"), rtol=0, atol=1e-6)
    rng = np.random.RandomState(8)
    T = 2048                                                     # ppo1's default horizon
    seg = dict(rew=rng.randn(T).astype(np.float32), vpred=rng.randn(T).astype(np.float32),
               new=(rng.rand(T) < 0.01).astype(np.int32), nextvpred=np.float32(0.3))
    add_vtarg_and_adv(seg, 0.99, 0.95)
    a64, r64 = scan.gae_single_env_f64(seg["rew"], seg["vpred"], seg["new"], seg["nextvpred"], 0.99, 0.95)
    np.testing.assert_array_equal(seg["adv"], a64)
    np.testing.assert_array_equal(seg["tdlamret"], r64)


@pytest.mark.parametrize("nrows,ncols", [(1, 5), (16, 5), (64, 5), (1000, 8), (37, 16)])
def test_row_mean(hip, nrows, ncols):
    rng = np.random.RandomState(nrows)
    rows = (rng.randn(nrows + 3, 16) * 10).astype(np.float32)
    got = hip.row_mean(dev(rows), nrows, ncols).cpu().numpy()
    np.testing.assert_allclose(got, rows[:nrows, :ncols].astype(np.float64).mean(0), rtol=2e-7, atol=1e-7)
    got = hip.row_mean(dev(rows), nrows, ncols, scale=0.5).cpu().numpy()
    np.testing.assert_allclose(got, 0.5 * rows[:nrows, :ncols].astype(np.float64).sum(0), rtol=2e-7, atol=1e-6)


@pytest.mark.parametrize("offa,offb", [(0, 0), (1, 0), (0, 3), (2, 1)])
def test_gemm_unaligned_operands_take_the_element_path(hip, offa, offb):
    """Operands whose first element is not 16-byte aligned cannot use the wide loads; the product must not
    depend on which path ran."""
    rng = np.random.RandomState(7)
    M, N, K = 200, 96, 64
    A = rng.randn(M, K).astype(np.float32)
    B = rng.randn(K, N).astype(np.float32)
    bufa = torch.zeros(M * K + 8, device="cuda")
    bufb = torch.zeros(K * N + 8, device="cuda")
    a = bufa[offa:offa + M * K].view(M, K)
    b = bufb[offb:offb + K * N].view(K, N)
    a.copy_(torch.from_numpy(A))
    b.copy_(torch.from_numpy(B))
    assert a.data_ptr() % 16 == 4 * offa and b.data_ptr() % 16 == 4 * offb
    C = hip.gemm(a, b).cpu().numpy()
    ref = hip.gemm(dev(A), dev(B)).cpu().numpy()          # aligned: wide loads
    np.testing.assert_array_equal(C, ref)                  # same fma chain per output element either way
    np.testing.assert_allclose(C, A.astype(np.float64) @ B.astype(np.float64), rtol=0, atol=2e-5)
