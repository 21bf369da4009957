"""CPU-side checks of the C-ABI library: every symbol the header declares is exported,
the host-only entry points (permutation generator, parameter layout) agree with NumPy
and with the oracle's variable tables, and bad arguments are refused with a message."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import tfgraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from baselines_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(lib):
    text = open(os.path.join(ROOT, "include", "ppo2hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(ppo2hip_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 25
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "declared in ppo2hip.h but not exported: " + name
    assert declared == set(lib.EXPORTED), (declared ^ set(lib.EXPORTED))
    assert lib.lib.ppo2hip_version() == 1


@pytest.mark.parametrize("seed", [0, 1, 1234])
@pytest.mark.parametrize("n", [1024, 4096])
def test_shuffle_matches_golden_and_numpy(lib, golden, seed, n):
    from baselines_amd.hip import LegacyShuffler
    exp = golden["perm_s%d_n%d" % (seed, n)]
    sh = LegacyShuffler(seed)
    for dtype in (np.int64, np.int32):
        sh.seed(seed)
        inds = np.arange(n, dtype=dtype)
        for e in range(4):
            sh.shuffle(inds)
            np.testing.assert_array_equal(inds, exp[e])     # bit-exact, cumulative across epochs


@pytest.mark.parametrize("n", [1, 2, 5, 255, 256, 257, 65536, 100000, 524288])
def test_shuffle_stream_is_numpys(lib, n):
    """Same permutation AND same generator state afterwards, so NumPy draws that follow
    (e.g. the env's) are unchanged."""
    from baselines_amd.hip import LegacyShuffler
    np.random.seed(n)
    np.random.rand(n % 7)                       # leave the stream mid-block
    sh = LegacyShuffler()
    sh.adopt_numpy()
    a = np.arange(n, dtype=np.int32)
    sh.shuffle(a)
    sh.release_numpy()
    after = np.random.randint(0, 2 ** 31 - 1, size=3)
    np.random.seed(n)
    np.random.rand(n % 7)
    b = np.arange(n)
    np.random.shuffle(b)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(after, np.random.randint(0, 2 ** 31 - 1, size=3))


def test_permutation_pipeline_all_epochs(lib, golden):
    from baselines_amd.hip import LegacyShuffler, PermPipeline
    sh = LegacyShuffler(1234)
    pp = PermPipeline(sh, 4096, 4, pinned=False)
    for _ in range(2):                          # a second job restarts from arange(n), as each update does
        sh.seed(1234)
        pp.begin(4)
        for e in range(4):
            np.testing.assert_array_equal(pp.wait(e).numpy(), golden["perm_s1234_n4096"][e])
        pp.wait_all()
    np.random.seed(1234)
    x = np.arange(4096)
    for _ in range(4):
        np.random.shuffle(x)
    key, pos = sh.get_state()
    st = np.random.get_state()
    assert pos == st[2] and np.array_equal(key, st[1])
    pp.begin(2)                                 # fewer epochs than the handle's maximum
    pp.wait_all()
    with pytest.raises(lib.Ppo2HipError):
        pp.begin(5)


@pytest.mark.parametrize("n,nepochs", [(4096, 4), (70001, 7), (2, 3), (1, 2)])
def test_permutation_pipeline_relative_mode(lib, n, nepochs):
    """Each epoch's own permutation of arange(n), chained by a gather, is the in-place shuffle
    sequence of baselines/ppo2/ppo2.py:110-113; the stream ends where NumPy's does."""
    from baselines_amd.hip import LegacyShuffler, PermPipeline
    sh = LegacyShuffler(77)
    pp = PermPipeline(sh, n, nepochs, pinned=False)
    pp.begin(nepochs, relative=True)
    np.random.seed(77)
    x = np.arange(n)
    run = None
    for e in range(nepochs):
        sig = pp.wait(e).numpy().copy()
        run = sig if run is None else run[sig]
        np.random.shuffle(x)
        np.testing.assert_array_equal(run, x)
    pp.wait_all()
    key, pos = sh.get_state()
    st = np.random.get_state()
    assert pos == st[2] and np.array_equal(key, st[1])


def test_permutation_pipeline_many_jobs_out_of_order_completion(lib):
    """Workers finish epochs in any order; wait(e) must see each one exactly when it is complete."""
    from baselines_amd.hip import LegacyShuffler, PermPipeline
    n, nepochs = 3001, 6
    sh = LegacyShuffler(5)
    pp = PermPipeline(sh, n, nepochs, pinned=False)
    np.random.seed(5)
    for job in range(40):
        pp.begin(nepochs, relative=(job % 2 == 0))
        x = np.arange(n)
        run = None
        for e in range(nepochs):
            got = pp.wait(e).numpy().copy()
            np.random.shuffle(x)
            if job % 2 == 0:
                run = got if run is None else run[got]
                got = run
            np.testing.assert_array_equal(got, x)
        pp.wait_all()


def test_shuffle_without_avx512_in_a_fresh_process():
    """The 256-bit draw path (taken on CPUs without AVX-512) against NumPy, whatever this machine has."""
    import os, subprocess, sys
    code = (
        "import numpy as np\n"
        "from baselines_amd.hip import LegacyShuffler, PermPipeline\n"
        "for n in (1, 2, 3, 100, 4097, 300001):\n"
        "    sh = LegacyShuffler(n); a = np.arange(n, dtype=np.int64); sh.shuffle(a)\n"
        "    np.random.seed(n); b = np.arange(n); np.random.shuffle(b)\n"
        "    assert np.array_equal(a, b), n\n"
        "sh = LegacyShuffler(9); pp = PermPipeline(sh, 50000, 3, pinned=False); pp.begin(3)\n"
        "np.random.seed(9); x = np.arange(50000)\n"
        "for e in range(3):\n"
        "    np.random.shuffle(x); assert np.array_equal(pp.wait(e).numpy(), x)\n"
        "pp.wait_all(); print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PPO2HIP_NO_AVX512="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_rng_argument_checks(lib):
    from baselines_amd.hip import LegacyShuffler
    sh = LegacyShuffler(0)
    with pytest.raises(lib.Ppo2HipError) as e:
        sh.set_state(np.zeros(624, np.uint32), 700)
    assert "pos" in str(e.value)
    with pytest.raises(TypeError):
        sh.shuffle(np.arange(10, dtype=np.float32))


def _spec(lib, **kw):
    from baselines_amd.hip import PolicySpec
    return PolicySpec(**kw)


def test_param_layout_matches_reference_variable_order(lib):
    # fork MlpPolicy: actor scope, then value scope (13 arrays, checkpoint order)
    s = _spec(lib, kind=lib.POLICY_MLP, dist=lib.DIST_GAUSSIAN, ac_dim=6, nvalue=5, ob_dim=17, hidden=(256, 128),
              order=lib.ORDER_FORK)
    ref = tfgraph.mlp_param_shapes(17, (256, 128), 6, 5, True, "fork")
    assert [v[0] for v in s.vars] == list(ref.keys()) and len(s.vars) == 13
    assert [v[2] for v in s.vars] == [tuple(x) for x in ref.values()]
    assert s.nparams == 76433
    # upstream MlpPolicy(64,64)
    s = _spec(lib, kind=lib.POLICY_MLP, dist=lib.DIST_GAUSSIAN, ac_dim=6, nvalue=1, ob_dim=17, hidden=(64, 64),
              order=lib.ORDER_UPSTREAM)
    ref = tfgraph.mlp_param_shapes(17, (64, 64), 6, 1, True, "upstream")
    assert [v[0] for v in s.vars] == list(ref.keys()) and s.nparams == 11085
    s = _spec(lib, kind=lib.POLICY_MLP, dist=lib.DIST_CATEGORICAL, ac_dim=2, nvalue=1, ob_dim=4, hidden=(64, 64),
              order=lib.ORDER_UPSTREAM)
    assert s.nparams == 9155 and "logstd" not in [v[0] for v in s.vars]
    # nature_cnn family
    for kind, lstm in ((lib.POLICY_CNN, None), (lib.POLICY_CNN_LSTM, "lstm"), (lib.POLICY_CNN_LNLSTM, "lnlstm")):
        s = _spec(lib, kind=kind, dist=lib.DIST_CATEGORICAL, ac_dim=6, nvalue=1, ob_hwc=(84, 84, 4), nlstm=256)
        ref = tfgraph.cnn_param_shapes(4, 6, lstm, 256)
        assert [v[0] for v in s.vars] == list(ref.keys())
        assert [int(np.prod(v[2])) for v in s.vars] == [int(np.prod(x)) for x in ref.values()]
        offs = [v[1] for v in s.vars]
        assert offs == sorted(offs) and offs[0] == 0


def test_bad_policy_descriptions_are_refused(lib):
    with pytest.raises(lib.Ppo2HipError):
        _spec(lib, kind=lib.POLICY_MLP, dist=lib.DIST_GAUSSIAN, ac_dim=0, nvalue=1, ob_dim=4, hidden=(64, 64))
    with pytest.raises(lib.Ppo2HipError):
        _spec(lib, kind=lib.POLICY_MLP, dist=lib.DIST_GAUSSIAN, ac_dim=2, nvalue=9, ob_dim=4, hidden=(64, 64))
    with pytest.raises(lib.Ppo2HipError):
        _spec(lib, kind=7, dist=lib.DIST_GAUSSIAN, ac_dim=2, nvalue=1, ob_dim=4, hidden=(64, 64))
    with pytest.raises(lib.Ppo2HipError):
        _spec(lib, kind=lib.POLICY_CNN, dist=lib.DIST_CATEGORICAL, ac_dim=4, nvalue=1, ob_hwc=(20, 20, 4))


def test_product_never_imports_the_oracle():
    """The oracle is the checker: nothing under baselines_amd/ may reference it."""
    for d, _, files in os.walk(os.path.join(ROOT, "baselines_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(d, f)
                assert "oracle/" not in text and "oracle." not in text, os.path.join(d, f)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 on its own, with no C++ or HIP type in it."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this machine")
    src = tmp_path / "hdr.c"
    src.write_text('#include "ppo2hip.h"\nint main(void) { return ppo2hip_version() == 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], check=True)


def test_import_fails_loudly_without_the_library():
    """No CPU fallback: if libppo2hip.so cannot be loaded, importing the binding raises (fresh process, the
    loader made to fail for that one file)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, sys\n"
            "real = ctypes.CDLL\n"
            "def cdll(path, *a, **k):\n"
            "    if str(path).endswith('libppo2hip.so'):\n"
            "        raise OSError('libppo2hip.so: cannot open shared object file')\n"
            "    return real(path, *a, **k)\n"
            "ctypes.CDLL = cdll\n"
            "try:\n"
            "    import baselines_amd._lib\n"
            "except OSError as e:\n"
            "    print('raised:', e); sys.exit(0)\n"
            "sys.exit(1)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "raised:" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_index_pipeline_under_host_sanitizers(tmp_path, sanitizer):
    """The worker ring of the index pipeline (token passing between threads, out-of-order epoch completion) built
    with ThreadSanitizer / AddressSanitizer on the HOST side only and driven through 40 jobs in both modes; the
    harness also checks every epoch against the one-shot shuffle."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "perm_san")
    subprocess.run([hipcc, "-O1", "-g", "-std=c++17", "-mavx2", "--offload-arch=gfx950", "-x", "hip", "-Xarch_host",
                    "-fsanitize=" + sanitizer, "-I", os.path.join(root, "include"), "-o", exe,
                    os.path.join(root, "tests", "host", "perm_sanitize.cpp"), os.path.join(root, "baselines_amd", "csrc", "perm.cpp"),
                    "-lpthread"], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatching epochs: 0" in r.stdout, r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr


def test_bad_arguments_are_refused_before_any_launch(lib):
    """Shape and pointer mistakes come back as PPO2HIP_EINVAL with a message; none of these calls reaches the GPU
    (they run on a machine without one)."""
    L = lib.lib
    null = ctypes.c_void_p(None)
    one = ctypes.c_void_p(16)            # never dereferenced: the checks below fail first
    g = (ctypes.c_double * 9)(*([0.99] * 9))
    calls = {
        "gae R too large": lambda: L.ppo2hip_gae_f32(one, one, one, one, one, g, 0.95, 4, 4, 9, one, one, one, null),
        "gae null": lambda: L.ppo2hip_gae_f32(null, one, one, one, one, g, 0.95, 4, 4, 1, one, one, one, null),
        "discount negative T": lambda: L.ppo2hip_discount(one, 0, null, 0.9, -1, 4, one, 1, null),
        "discount f64 -> f32": lambda: L.ppo2hip_discount(one, 1, null, 0.9, 4, 4, one, 0, null),
        "rms empty batch": lambda: L.ppo2hip_rms_update(one, 0, 0, 3, one, one, 4096, null),
        "moments empty batch": lambda: L.ppo2hip_column_moments(one, 0, 0, 3, one, one, one, 4096, null),
        "normalize negative clip": lambda: L.ppo2hip_normalize_clip(one, 0, 4, 3, one, one, 1e-8, -1.0, one, 0, null),
        "frame stack element size": lambda: L.ppo2hip_frame_stack(one, one, null, 2, 10, 1, 4, 2, null),
        "bootstrap R too large": lambda: L.ppo2hip_done_bootstrap(one, one, one, g, 4, 9, null),
        "row mean too wide": lambda: L.ppo2hip_row_mean(one, 4, 32, 17, 0.25, one, null),
        "compose negative n": lambda: L.ppo2hip_compose_i32(one, one, one, -1, null),
        "compose aliased": lambda: L.ppo2hip_compose_i32(one, one, one, 8, null),
        "sample null": lambda: L.ppo2hip_sample_actions(None, one, one, 4, 0, 0, 0, one, one, null, null),
        "act step null": lambda: L.ppo2hip_act_step(None, one, one, 4, null, null, 0, 0, 0, one, one, one, null, one, 1 << 20, null),
        "add null": lambda: L.ppo2hip_add_f64(null, one, 4, null),
    }
    for what, call in calls.items():
        rc = call()
        assert rc == lib.EINVAL, (what, rc)
        assert L.ppo2hip_last_error(), what
    # empty inputs are fine and launch nothing
    assert L.ppo2hip_gae_f32(null, null, null, null, null, g, 0.95, 0, 4, 1, null, null, null, null) == 0
    assert L.ppo2hip_discount(null, 0, null, 0.9, 0, 4, null, 1, null) == 0
    assert L.ppo2hip_compose_i32(null, null, null, 0, null) == 0
