"""
GPU parity tests: the HIP path (through the C ABI / the Python mirror of the
reference API) against the CPU oracle and the committed golden vectors.
Run on the MI355X box with `pytest -m gpu`.

Tolerances (north_star): posterior mean/variance and EI/PI/UCB within 1e-6
relative in fp64; arg-max indices exact.  Acquisition values additionally get
an absolute floor of 1e-12: deep in the tail EI = ydiff*cdf + sigma*pdf is the
difference of two nearly equal terms and 1+erf(.) has lost all digits in the
reference itself, so only values above the floor carry information.
"""
import ctypes

import numpy as np
import pytest

from conftest import load_golden, kern_from_golden, synth

pytestmark = pytest.mark.gpu

RT = 1e-6
ACQ_ATOL = 1e-12


@pytest.fixture(scope="module")
def ibo():
    import ibo_amd
    from ibo_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the product has no CPU fallback")
    err = ctypes.c_double()
    _lib.check(_lib.lib.ibo_selftest_mfma(0, ctypes.byref(err)))
    return ibo_amd


def close(a, b, rtol=RT, atol=1e-12):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def our_kernel(ktype, hyper):
    from ibo_amd.gaussianprocess import kernel as K
    ktype = int(ktype)
    if ktype == 0: return K.GaussianKernel_ard(np.array(hyper, float))
    if ktype == 1: return K.GaussianKernel_iso(np.array(hyper, float))
    if ktype == 2: return K.MaternKernel3(np.array(hyper, float))
    return K.MaternKernel5(np.array(hyper, float))


def our_prior(g, p):
    from ibo_amd.gaussianprocess.prior import RBFNMeanPrior
    if p + "pmeans" not in g.files:
        return None
    return RBFNMeanPrior(g[p + "pmeans"], g[p + "pbeta"], float(g[p + "ptheta"]), g[p + "plowerb"], g[p + "pwidth"])


def test_mfma_layout_selftest(ibo):
    from ibo_amd import _lib
    err = ctypes.c_double(-1)
    _lib.check(_lib.lib.ibo_selftest_mfma(0, ctypes.byref(err)))
    assert err.value == 0.0
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib.ibo_device_name(0, buf, 256))
    assert b"gfx950" in buf.value, buf.value


def test_g1_demo(ibo):
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import maximizeEI
    g = load_golden("g1_demo")
    GP = GaussianProcess(GaussianKernel_ard(g["hyper"]), noise=float(g["noise"]))
    GP.addData(g["X"], g["Y"])
    close(GP.R, g["R"], rtol=1e-12); close(GP.L, g["L"], rtol=1e-10)
    for q, ref in zip(g["probe"], g["post"]):
        mu, s2 = GP.posterior(q)
        close(mu, ref[0], atol=1e-10); close(s2, ref[1])
    opt, optx = maximizeEI(GP, g["bounds"].tolist(), xi=float(g["xi"]))
    close(opt, g["c_opt"]); close(optx, g["c_optx"], rtol=1e-9)


def test_g3_seeded_cases(ibo, oracle):
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.acquisition import maximizeEI, maximizePI, maximizeUCB, EI, PI, UCB, sweep
    from ibo_amd.utils.optimize import direct, cdirect
    g = load_golden("g3_cases")
    for name in g["names"]:
        p = str(name) + "/"
        bounds = g[p + "bounds"].tolist()
        GP = GaussianProcess(our_kernel(g[p + "ktype"], g[p + "hyper"]), g[p + "X"], g[p + "Y"],
                             prior=our_prior(g, p), noise=float(g[p + "noise"]))
        D = GP.X.shape[1]
        close(GP.R, g[p + "R"], rtol=1e-12); close(GP.L, g[p + "L"], rtol=1e-8)
        probe = g[p + "probe"]
        mu, s2 = GP.posteriors(probe)
        close(mu, g[p + "post"][:, 0], atol=1e-9); close(s2, g[p + "post"][:, 1])
        for q, ref in zip(probe[:3], g[p + "post"][:3]):       # single-point (GEMV) path
            m1, v1 = GP.posterior(q)
            close(m1, ref[0], atol=1e-9); close(v1, ref[1])
        # Python-class flavour (NR erf, clamp 1e-7)
        ei, pi, ucb = EI(GP), PI(GP), UCB(GP, D)
        close([ei.f(q) for q in probe[:4]], g[p + "ei_py"][:4], atol=ACQ_ATOL)
        close(ei.values(probe), g[p + "ei_py"], atol=ACQ_ATOL)
        close([pi.f(q) for q in probe[:4]], g[p + "pi_py"][:4], atol=ACQ_ATOL)
        close([ucb.f(q) for q in probe[:4]], g[p + "ucb_py"][:4])
        # native flavour per candidate (libm erf, clamp 1e-8) -- values from the real libego
        for acq, key, kw in (("ei", "ei_c", dict(xi=.01)), ("pi", "pi_c", dict(xi=.01)),
                             ("ucb", "ucb_c", dict(parm=float(g[p + "ucb_parm"])))):
            r = sweep(GP, probe, acq=acq, native=True, outputs=("acq",), **kw)
            close(r["acq"], g[p + key], atol=ACQ_ATOL)
            assert r["best_idx"] == int(np.argmax(g[p + key]))
        # DIRECT on the GPU objective vs the reference's maximize* (4 decimals / 0.01 is the
        # reference's own bar, ego/unittest_IBO.py:156-162; we hold 1e-6 / 1e-6)
        for fn, key in ((maximizeEI, "max_ei"), (maximizePI, "max_pi"), (maximizeUCB, "max_ucb")):
            opt, optx = fn(GP, bounds, maxiter=10)
            close(opt, g[p + key][0], atol=ACQ_ATOL); close(optx, g[p + key][1:], rtol=1e-9, atol=1e-12)
        opt, optx = maximizeEI(GP, bounds)
        close(opt, g[p + "max_ei50"][0], atol=ACQ_ATOL); close(optx, g[p + "max_ei50"][1:], rtol=1e-9, atol=1e-12)
        # Python objective through both DIRECT implementations
        f, x = cdirect(ei.negf, bounds, maxiter=10)
        close(f, g[p + "cdirect_ei"][0], atol=ACQ_ATOL); assert np.sum(np.abs(x - g[p + "cdirect_ei"][1:])) < .01
        f, x = direct(ei.negf, bounds, maxiter=10)
        close(f, g[p + "direct_ei"][0], atol=ACQ_ATOL); assert np.sum(np.abs(x - g[p + "direct_ei"][1:])) < .01


def test_batch_vs_sequential_identity(ibo):
    """ego/unittest_GP.py:109-156: R, mu, sigma2 are identical however the data arrive"""
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_iso
    from ibo_amd.utils.latinhypercube import lhcSample
    b = [[0., 10.]] * 4
    X = lhcSample(b, 25, seed=1)
    Y = [float(np.sum(np.sin(x))) for x in X]
    GP1 = GaussianProcess(GaussianKernel_iso([.1]), X, Y, noise=.2)
    GP2 = GaussianProcess(GaussianKernel_iso([.1]), noise=.2)
    for x, y in zip(X, Y):
        GP2.addData(x, y)
    GP4 = GaussianProcess(GaussianKernel_iso([.1]), X[:10], Y[:10], noise=.2)
    GP4.addData(X[10], Y[10]); GP4.addData(X[11:18], Y[11:18])
    for i in range(18, 25):
        GP4.addData(X[i], Y[i])
    assert np.all(GP1.R == GP2.R) and np.all(GP1.R == GP4.R)
    # the factor of a model grown point by point (block extension, as the reference grows it) differs from the
    # batch factor in the last bits; the reference's own test asks for 5-7 decimal places (assertAlmostEqual)
    for x in lhcSample(b, 25, seed=2):
        close(GP2.posterior(x), GP1.posterior(x), rtol=1e-11, atol=1e-13)
        close(GP4.posterior(x), GP1.posterior(x), rtol=1e-11, atol=1e-13)
    assert len(GP2.X) == 25


def test_add_data_block_extension_equals_refit(ibo, oracle):
    """GaussianProcess.addData on a fitted model extends L (and W = L^-1, the alpha vectors, every packed copy)
    on the device as ego/gaussianprocess/__init__.py:301-308 does; everything must agree with a refit from
    scratch at 1e-12 and with the oracle's (refitted) posterior at 1e-6 -- also across a 64-row padding boundary
    (where the extension hands over to a refit), for a batch, after a kernel change, and when it must fail"""
    from ibo_amd import _lib, NotPositiveDefinite
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.acquisition import sweep, maximizeEI
    for N0, D, kern, okern, adds in ((300, 4, K.GaussianKernel_ard([.3] * 4), ("ard", [.3] * 4), (1, 1, 3, 1, 16, 17)),
                                     (60, 3, K.MaternKernel5([.5, 1.0]), ("m5", [.5, 1.0]), (1, 1, 1, 1, 1, 1, 2)),     # crosses 64
                                     (1000, 8, K.MaternKernel3([.7, 1.0]), ("m3", [.7, 1.0]), (1, 5, 1))):
        tot = N0 + sum(adds)
        X, Y = synth(90 + D, tot, D)
        GP = GaussianProcess(kern, X[:N0], Y[:N0], noise=.1)
        n = N0
        for a in adds:
            GP.addData(X[n:n + a] if a > 1 else X[n], Y[n:n + a] if a > 1 else Y[n])
            n += a
            ref = GaussianProcess(kern, X[:n], Y[:n], noise=.1)
            np.testing.assert_array_equal(GP.R, ref.R)
            close(GP.L, ref.L, rtol=1e-12, atol=1e-13)
            Wa = np.empty((n, n)); Wb = np.empty((n, n))
            _lib.check(_lib.lib.ibo_gp_get_W(GP._handle(), _lib.dp(Wa))); _lib.check(_lib.lib.ibo_gp_get_W(ref._handle(), _lib.dp(Wb)))
            close(Wa, Wb, rtol=1e-10, atol=1e-11)
            probe = np.random.RandomState(n).rand(40, D)
            close(GP.posteriors(probe), ref.posteriors(probe), rtol=1e-11, atol=1e-12)
        # the large-batch kernel reads the packed copies (fragment-order W, augmented X, alpha): all extended
        cand = np.random.RandomState(7).rand(8300, D)
        ra = sweep(GP, cand, outputs=("mu", "s2", "acq")); rb = sweep(ref, cand, outputs=("mu", "s2", "acq"))
        close(ra["mu"], rb["mu"], rtol=1e-10, atol=1e-11); close(ra["s2"], rb["s2"], rtol=1e-10); assert ra["best_idx"] == rb["best_idx"]
        close(maximizeEI(GP, [[0., 1.]] * D, maxiter=8)[0], maximizeEI(ref, [[0., 1.]] * D, maxiter=8)[0], rtol=1e-9)
        ogp = oracle.GP(oracle.Kern(*okern), X[:n], Y[:n], noise=.1)
        close(GP.posteriors(probe), ogp.posteriors(probe), atol=1e-9)
    # a changed kernel is noticed (refit with the new one), not extended with the old factor
    X, Y = synth(3, 41, 2)
    GP = GaussianProcess(K.GaussianKernel_iso([.3]), X[:40], Y[:40], noise=.1)
    GP.kernel = K.GaussianKernel_iso([.5])
    GP.addData(X[40], Y[40])
    close(GP.posteriors(X[:5] + .01), GaussianProcess(K.GaussianKernel_iso([.5]), X, Y, noise=.1).posteriors(X[:5] + .01), rtol=1e-12)
    # a duplicate point without noise has no factor: numpy's LinAlgError, and the model stays as it was
    GP = GaussianProcess(K.GaussianKernel_iso([.3]), X[:20], Y[:20], noise=0.0)
    before = GP.posteriors(X[20:25])
    with pytest.raises(NotPositiveDefinite):
        GP.addData(X[3], Y[3])
    assert len(GP.X) == 20
    close(GP.posteriors(X[20:25]), before, rtol=1e-12)


@pytest.mark.parametrize("name", ["c1_n32_d2_ard", "c2r_n256_d4_ard", "c2r_n256_d4_iso", "c3r_n192_d8_m5",
                                  "c3r_n192_d8_m3", "c2_n1024_d4_ard"])
def test_g6_sweeps_vs_reference_vectors(ibo, name):
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.acquisition import sweep
    from ibo_amd import _lib
    g = load_golden("g6_sweeps")
    p = name + "/"
    N, D, M, seed = int(g[p + "N"]), int(g[p + "D"]), int(g[p + "M"]), int(g[p + "seed"])
    X, Y = synth(seed, N, D)
    cand = np.random.RandomState(100 + seed).rand(M, D)
    GP = GaussianProcess(our_kernel(g[p + "ktype"], g[p + "hyper"]), X, Y, noise=.1)
    for path in (2, 3, 1):     # MFMA tile kernel, its panel-split form (small batches), the GEMV kernel
        _lib.check(_lib.lib.ibo_set_option(b"sweep_path", path))
        try:
            sub = slice(0, M if path != 1 else min(M, 48))
            r = sweep(GP, cand[sub], acq='ei', xi=.01, native=False, outputs=("mu", "s2", "acq"))
            close(r["mu"], g[p + "mu"][sub], atol=1e-9); close(r["s2"], g[p + "s2"][sub])
            close(r["acq"], g[p + "ei_py"][sub], atol=ACQ_ATOL)
            assert r["best_idx"] == int(np.argmax(g[p + "ei_py"][sub]))
            r = sweep(GP, cand[sub], acq='pi', xi=.01, native=False, outputs=("acq",))
            close(r["acq"], g[p + "pi_py"][sub], atol=ACQ_ATOL)
            if p + "ei_c" in g.files:
                r = sweep(GP, cand[sub], acq='ei', xi=.01, native=True, outputs=("acq",))
                close(r["acq"], g[p + "ei_c"][sub], atol=ACQ_ATOL)
                assert r["best_idx"] == int(np.argmax(g[p + "ei_c"][sub]))
                r = sweep(GP, cand[sub], acq='ucb', parm=1.5, native=True, outputs=("acq",))
                close(r["acq"], g[p + "ucb_c"][sub])
                assert r["best_idx"] == int(np.argmax(g[p + "ucb_c"][sub]))
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"sweep_path", 0))


@pytest.mark.parametrize("N,D,kind,M", [(100, 3, "ard", 1000), (513, 5, "m5", 300), (1024, 4, "ard", 2048),
                                         (700, 16, "ard", 200), (65, 1, "iso", 130), (1, 2, "ard", 70),
                                         (64, 2, "m3", 64), (129, 8, "iso", 17), (1471, 4, "ard", 1),
                                         (2049, 7, "iso", 65)])
def test_sweep_vs_oracle_ragged(ibo, oracle, N, D, kind, M):
    """seeded inputs at awkward sizes (N not a multiple of 16/64/512, M not of 64, N=1, N on either side of a
    64-block boundary, N=2049: 33 blocks, the two-level Cholesky order of the fit path)"""
    from ibo_amd.gaussianprocess import GaussianProcess, kernel as K
    from ibo_amd.acquisition import sweep
    X, Y = synth(7 + N, N, D)
    cand = np.random.RandomState(N + M).rand(M, D)
    hyper = {"ard": [.3 + .05 * d for d in range(D)], "iso": [.4], "m5": [.5, 1.0], "m3": [.6, .9]}[kind]
    ok = oracle.Kern(kind, hyper)
    ogp = oracle.GP(ok, X, Y, noise=.1)
    cls = {"ard": K.GaussianKernel_ard, "iso": K.GaussianKernel_iso, "m5": K.MaternKernel5, "m3": K.MaternKernel3}[kind]
    GP = GaussianProcess(cls(hyper), X, Y, noise=.1)
    close(GP.R, ogp.R, rtol=1e-12); close(GP.L, ogp.L, rtol=1e-8, atol=1e-12)
    o_mu, o_s2 = ogp.posteriors(cand)
    r = sweep(GP, cand, acq='ei', xi=.01, native=False, outputs=("mu", "s2", "acq"))
    close(r["mu"], o_mu, atol=1e-9); close(r["s2"], o_s2)
    o_ei = oracle.acq_value(oracle.ACQ_EI, oracle.ERF_NR, o_mu, np.sqrt(o_s2), Y.max(), .01)
    close(r["acq"], o_ei, atol=ACQ_ATOL)
    assert r["best_idx"] == int(np.argmax(o_ei))
    sub = cand[:64]
    sw = oracle.sweep_native(ogp, sub, oracle.ACQ_EI, .01)
    r = sweep(GP, sub, acq='ei', xi=.01, native=True, outputs=("acq",))
    close(r["acq"], sw["acq"], atol=ACQ_ATOL)
    assert r["best_idx"] == sw["best_idx"]


def test_panel_split_kernel_in_dot_form_against_oracle(ibo, oracle):
    """the panel-split form of the first-generation MFMA kernel in its dot form -- by default what takes small batches on models beyond
    sweep2's LDS budget -- forced with sweep_path = 3 on a model of three 512-row panels, against the oracle"""
    from ibo_amd.gaussianprocess import GaussianProcess, kernel as K
    from ibo_amd.acquisition import sweep
    from ibo_amd import _lib
    N, D, M = 1100, 6, 333
    X, Y = synth(31, N, D)
    cand = np.random.RandomState(32).rand(M, D)
    hyper = [.35 + .03 * d for d in range(D)]
    ogp = oracle.GP(oracle.Kern("ard", hyper), X, Y, noise=.1)
    o_mu, o_s2 = ogp.posteriors(cand)
    GP = GaussianProcess(K.GaussianKernel_ard(hyper), X, Y, noise=.1)
    _lib.check(_lib.lib.ibo_set_option(b"sweep_path", 3))
    try:
        r = sweep(GP, cand, acq='ucb', parm=1.2, native=True, outputs=("mu", "s2", "acq"))
    finally:
        _lib.check(_lib.lib.ibo_set_option(b"sweep_path", 0))
    assert r["kernel"] == "sweep_mfma_kernel<split>"
    close(r["mu"], o_mu, atol=1e-9); close(r["s2"], o_s2)
    o = o_mu + 1.2 * np.sqrt(o_s2)
    close(r["acq"], o); assert r["best_idx"] == int(np.argmax(o))


def test_kstar_dot_form_matches_difference_form(ibo, oracle):
    """SE k* as exp(a_k + b_c + x~.c~) vs the difference form, incl. far-away candidates and an
    sf2 != 1 kernel; both against the oracle"""
    from ibo_amd.gaussianprocess import GaussianProcess, kernel as K
    from ibo_amd.acquisition import sweep
    from ibo_amd import _lib
    N, D, M = 300, 3, 500
    X, Y = synth(41, N, D)
    cand = np.random.RandomState(42).rand(M, D)
    cand[:50] = cand[:50] * 40 - 20                       # far outside the data (demo.py probes (-10,.5,-10))
    cand[50:60] = X[:10]                                  # exactly on observations
    for kern, okern in ((K.GaussianKernel_ard([.5, .5, .3]), oracle.Kern("ard", [.5, .5, .3])),
                        (K.SVGaussianKernel_iso([.4, 1.02]), oracle.Kern("sviso", [.4, 1.02])),
                        (K.MaternKernel5([.6, 1.0]), oracle.Kern("m5", [.6, 1.0])),
                        (K.MaternKernel3([.7, 1.0]), oracle.Kern("m3", [.7, 1.0]))):
        ogp = oracle.GP(okern, X, Y, noise=.1)
        o_mu, o_s2 = ogp.posteriors(cand)
        GP = GaussianProcess(kern, X, Y, noise=.1)
        res = []
        for form in (1, 0):
            _lib.check(_lib.lib.ibo_set_option(b"dot_form", form))
            try:
                r = sweep(GP, cand, acq='ei', xi=.01, native=False, outputs=("mu", "s2", "acq"))
            finally:
                _lib.check(_lib.lib.ibo_set_option(b"dot_form", -1))
            close(r["mu"], o_mu, atol=1e-9); close(r["s2"], o_s2)
            res.append(r)
        close(res[0]["mu"], res[1]["mu"], rtol=1e-9, atol=1e-11); close(res[0]["s2"], res[1]["s2"], rtol=1e-9)
        assert res[0]["best_idx"] == res[1]["best_idx"]


def test_sweep_exclusion_and_index_base(ibo):
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import sweep
    X, Y = synth(11, 64, 3)
    GP = GaussianProcess(GaussianKernel_ard([.3] * 3), X, Y)
    cand = np.random.RandomState(5).rand(1000, 3)
    r = sweep(GP, cand, outputs=("acq",))
    top = cand[r["best_idx"]]
    r2 = sweep(GP, cand, exclude=[top], exclude_radius=.5, outputs=("acq",), index_base=1000)
    d = np.linalg.norm(cand - top, axis=1)
    masked = np.where(d > .5, r["acq"], -np.inf)
    assert r2["best_idx"] - 1000 == int(np.argmax(masked))
    close(r2["acq"], r["acq"], rtol=0, atol=0)
    r3 = sweep(GP, cand, exclude=[top], exclude_radius=100.0)
    assert r3["best_idx"] == -1
    # ties go to the lowest index: duplicate the winner
    dup = np.vstack([cand, top])
    assert sweep(GP, dup)["best_idx"] == r["best_idx"]


def test_sweep_full_size_properties(ibo):
    """C2 size (N=1024, D=4, M=2^20): size-independent properties instead of a CPU oracle pass"""
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import sweep
    from ibo_amd import DeviceArray
    X, Y = synth(2, 1024, 4)
    GP = GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1)
    cand = np.random.RandomState(102).rand(1 << 20, 4)
    cand[777777] = X[5]                      # a candidate sitting on a training point
    dc = DeviceArray.from_host(cand)
    r = sweep(GP, dc, outputs=("mu", "s2", "acq"))
    assert r["best_idx"] == int(np.argmax(r["acq"]))                    # arg-max == argmax of the written values
    close(r["best_val"], r["acq"].max(), rtol=0, atol=0)
    assert np.all(r["s2"] >= 1e-8) and np.all(r["s2"] <= 1.1 + 1e-12)   # clamp / prior variance bound
    assert r["s2"][777777] < 1 / (1 + .1)                                # ego/unittest_GP.py:94-98
    # shard invariance: sweeping two halves and combining equals the full sweep
    h = 1 << 19
    a = sweep(GP, dc.view_rows(0, h)); b = sweep(GP, dc.view_rows(h, 2 * h), index_base=h)
    best = a if (a["best_val"] > b["best_val"] or (a["best_val"] == b["best_val"] and a["best_idx"] < b["best_idx"])) else b
    assert best["best_idx"] == r["best_idx"]
    # a random subset against the single-point posterior path
    idx = np.random.RandomState(0).choice(1 << 20, 8, replace=False)
    for i in idx:
        m, v = GP.posterior(cand[i])
        close(m, r["mu"][i], rtol=1e-9, atol=1e-11); close(max(v, 1e-7), max(r["s2"][i], 1e-7), rtol=1e-9)


def test_c2_full_size_winner_against_the_oracle(ibo, oracle):
    """C2 at its stated size: the winning candidate of the 2^20 sweep, the runner-up region and 256 random
    candidates are re-evaluated by the CPU oracle's reference-shaped posterior (two N^2 matvecs with inv(R),
    cpp/optimizeGP.cpp:57-215); values within 1e-6, and the oracle agrees that the winner beats them all"""
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import sweep
    from ibo_amd import DeviceArray
    X, Y = synth(2, 1024, 4)
    GP = GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1)
    M = 1 << 20
    cand = np.random.RandomState(102).rand(M, 4)
    r = sweep(GP, DeviceArray.from_host(cand), acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
    w = r["best_idx"]
    top = np.argsort(-r["acq"], kind="stable")[:32]                # the winner and its closest rivals
    assert top[0] == w
    idx = np.unique(np.r_[top, np.random.RandomState(7).choice(M, 256, replace=False)])
    ogp = oracle.GP(oracle.Kern("ard", [.3] * 4), X, Y, noise=.1)
    o = oracle.sweep_native(ogp, cand[idx], oracle.ACQ_EI, .01)
    close(r["mu"][idx], o["mu"], atol=1e-10); close(r["s2"][idx], o["s2"])
    close(r["acq"][idx], o["acq"], atol=ACQ_ATOL)
    assert idx[o["best_idx"]] == w                                  # the oracle picks the same winner
    close(r["best_val"], o["best_val"], atol=ACQ_ATOL)


def test_sweep2_kernel_against_oracle_and_the_other_kernels(ibo, oracle):
    """the large-batch kernel (sweep2.hip: 32-candidate tiles, exponent GEMM on the MFMA unit, table exp) on every
    kernel family, D = 1..16 (all five widths of the exponent GEMM), one / short-first / several row panels:
    per-candidate mu, s2, EI against the CPU oracle on a sample and against the independent GEMV kernel on all
    candidates; same arg-max as the first-generation tile kernel"""
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.acquisition import sweep
    cases = [(1024, 4, K.GaussianKernel_ard([.3] * 4), ("ard", [.3] * 4), 9000), (200, 3, K.GaussianKernel_iso([.4]), ("iso", [.4]), 8300),
             (64, 1, K.GaussianKernel_iso([.4]), ("iso", [.4]), 8300), (1000, 6, K.MaternKernel3([.6, 1.0]), ("m3", [.6, 1.0]), 8300),
             (2048, 8, K.MaternKernel5([.5, 1.0]), ("m5", [.5, 1.0]), 8300), (1500, 5, K.GaussianKernel_ard([.3] * 5), ("ard", [.3] * 5), 8300),
             (700, 10, K.GaussianKernel_ard([.5] * 10), ("ard", [.5] * 10), 8300), (600, 13, K.MaternKernel5([1.0, 1.0]), ("m5", [1.0, 1.0]), 8300),
             (1100, 16, K.GaussianKernel_ard([.9] * 16), ("ard", [.9] * 16), 8300),
             # more than 4096 observations: the alpha vectors' LDS window moves in the last panel (two panels short of a third)
             (4500, 6, K.MaternKernel5([.8, 1.0]), ("m5", [.8, 1.0]), 8200)]
    try:
        for N, D, kern, (okind, ohyp), M in cases:
            X, Y = synth(N + D, N, D)
            GP = GaussianProcess(kern, X, Y, noise=.1)
            cand = np.random.RandomState(N).rand(M, D)
            cand[77] = X[5]                                          # a candidate on top of an observation
            cand[M - 1] = cand[0]                                    # ragged last tile (M % 32 != 0) with a duplicate
            r = sweep(GP, cand, acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            assert r["kernel"] == "sweep2_kernel"
            _lib.check(_lib.lib.ibo_set_option(b"dot_form", 0))      # k* by differences: the first-generation tile kernel
            r1 = sweep(GP, cand, acq='ei', xi=.01, native=True)
            _lib.check(_lib.lib.ibo_set_option(b"dot_form", -1))
            assert r1["kernel"] == "sweep_mfma_kernel" and r1["best_idx"] == r["best_idx"]
            _lib.check(_lib.lib.ibo_set_option(b"sweep_path", 1))
            rg = sweep(GP, cand[:600], acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            _lib.check(_lib.lib.ibo_set_option(b"sweep_path", 0))
            assert rg["kernel"] == "sweep_gemv_kernel"
            close(r["mu"][:600], rg["mu"], rtol=1e-9, atol=1e-10); close(r["s2"][:600], rg["s2"], rtol=1e-9)
            close(r["acq"][:600], rg["acq"], atol=ACQ_ATOL)
            ogp = oracle.GP(oracle.Kern(okind, ohyp), X, Y, noise=.1)
            idx = np.r_[np.arange(0, M, M // 40), 77, M - 1, r["best_idx"]]
            o = oracle.sweep_native(ogp, cand[idx], oracle.ACQ_EI, .01)
            close(r["mu"][idx], o["mu"], atol=1e-9); close(r["s2"][idx], o["s2"]); close(r["acq"][idx], o["acq"], atol=ACQ_ATOL)
            assert r["best_idx"] == int(np.argmax(r["acq"])) and r["s2"][77] < 1 / 1.1
    finally:
        _lib.check(_lib.lib.ibo_set_option(b"dot_form", -1))
        _lib.check(_lib.lib.ibo_set_option(b"sweep_path", 0))


def test_small_batch_kernels_against_oracle_and_the_other_kernels(ibo, oracle):
    """batches of 1..4096 candidates (DIRECT's, posterior(x), posteriors of a few hundred points) run through small2.hip's three
    kernels: values against the CPU oracle, the panel-split kernel and the GEMV kernel; DIRECT takes the same samples
    whichever of them evaluates its batches"""
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.acquisition import sweep, gpuDirectGP
    opt = lambda k, v: _lib.check(_lib.lib.ibo_set_option(k, v))
    try:
        for N, D, kern, (okind, ohyp), M in ((1024, 4, K.GaussianKernel_ard([.3] * 4), ("ard", [.3] * 4), 57), (200, 3, K.GaussianKernel_iso([.4]), ("iso", [.4]), 17),
                                             (1000, 6, K.MaternKernel3([.6, 1.0]), ("m3", [.6, 1.0]), 600), (2048, 8, K.MaternKernel5([.5, 1.0]), ("m5", [.5, 1.0]), 31),
                                             (1500, 5, K.GaussianKernel_ard([.3] * 5), ("ard", [.3] * 5), 4096), (1100, 16, K.GaussianKernel_ard([.9] * 16), ("ard", [.9] * 16), 333),
                                             # down to a single candidate (posterior(x), DIRECT's first rectangles): also small2.hip since round 2
                                             (1024, 4, K.GaussianKernel_ard([.3] * 4), ("ard", [.3] * 4), 1), (2048, 8, K.MaternKernel5([.5, 1.0]), ("m5", [.5, 1.0]), 3),
                                             (200, 3, K.GaussianKernel_iso([.4]), ("iso", [.4]), 16), (64, 1, K.GaussianKernel_iso([.4]), ("iso", [.4]), 2),
                                             (4400, 5, K.GaussianKernel_ard([.6] * 5), ("ard", [.6] * 5), 21)):        # beyond 4096 observations
            X, Y = synth(N + D, N, D)
            GP = GaussianProcess(kern, X, Y, noise=.1)
            cand = np.random.RandomState(N).rand(M, D); cand[min(M - 1, 7)] = X[5]
            # (up to 512 observations and 128 candidates the wave-local kernel runs -- k* made by the wave that multiplies it -- beyond,
            # the separate k* / product kernels: the cases cover both)
            r = sweep(GP, cand, outputs=("mu", "s2", "acq"))
            assert r["kernel"] == "wk_small_kernel"
            opt(b"sweep_path", 3); r0 = sweep(GP, cand, outputs=("mu", "s2", "acq")); opt(b"sweep_path", 0)
            assert r0["kernel"] == "sweep_mfma_kernel<split>" and r0["best_idx"] == r["best_idx"]
            opt(b"sweep_path", 1); rg = sweep(GP, cand[:40], outputs=("mu", "s2", "acq")); opt(b"sweep_path", 0)
            for k in ("mu", "s2", "acq"):
                close(r[k], r0[k], rtol=1e-9, atol=1e-11); close(r[k][:40], rg[k], rtol=1e-9, atol=1e-11)
            ogp = oracle.GP(oracle.Kern(okind, ohyp), X, Y, noise=.1)
            idx = np.arange(0, M, max(1, M // 30))
            o = oracle.sweep_native(ogp, cand[idx], oracle.ACQ_EI, .01)
            close(r["mu"][idx], o["mu"], atol=1e-9); close(r["s2"][idx], o["s2"]); close(r["acq"][idx], o["acq"], atol=ACQ_ATOL)
        X, Y = synth(21, 300, 3)
        GP = GaussianProcess(K.GaussianKernel_ard([.25, .3, .35]), X, Y)
        runs = []
        for path in (0, 3, 1):                             # small2.hip's kernels, the panel-split kernel, the GEMV kernel
            opt(b"sweep_path", path)
            runs.append(gpuDirectGP(GP, [[0., 1.]] * 3, 30, 30, 10000, acqfunc='ei', xi=.01, return_samples=True))
        for v, x, ns in runs[1:]:
            assert ns == runs[0][2] and np.array_equal(x, runs[0][1]); close(v, runs[0][0], rtol=1e-9)
    finally:
        opt(b"sweep_path", 0)


def test_incremental_sweep_state_equals_full_sweeps(ibo):
    """sweep(incremental=True) on a fixed candidate array while the model grows by addData (the gallery's rounds):
    every round's per-candidate mu / s2 / EI and arg-max equal a full sweep of a freshly fitted model -- values at
    1e-6 (the bar), here 1e-9; the refresh kernel really runs; a refit, another array or a bigger jump start over"""
    from ibo_amd import DeviceArray
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.acquisition import sweep
    from ibo_amd.acquisition.gallery import fastUCBGallery
    for N0, D, kern in ((500, 4, K.GaussianKernel_ard([.3] * 4)), (1030, 8, K.MaternKernel5([.5, 1.0])), (120, 2, K.MaternKernel3([.4, 1.0]))):
        X, Y = synth(70 + D, N0 + 12, D)
        GP = GaussianProcess(kern, X[:N0], Y[:N0], noise=.1)
        dc = DeviceArray.from_host(np.random.RandomState(71).rand(9001, D))
        seen = []
        for rnd in range(7):
            excl = X[:2] if rnd % 2 else None
            r = sweep(GP, dc, acq='ei', xi=.4, native=False, exclude=excl, incremental=True, outputs=("mu", "s2", "acq"))
            seen.append(r["kernel"])
            ref = GaussianProcess(kern, GP.X, GP.Y, noise=.1)
            f = sweep(ref, dc, acq='ei', xi=.4, native=False, exclude=excl, outputs=("mu", "s2", "acq"))
            close(r["mu"], f["mu"], rtol=1e-9, atol=1e-10); close(r["s2"], f["s2"], rtol=1e-9); close(r["acq"], f["acq"], rtol=1e-9, atol=ACQ_ATOL)
            assert r["best_idx"] == f["best_idx"]
            n = len(GP.X)
            if rnd == 3:
                GP.addData(X[n:n + 2], Y[n:n + 2])              # two rows at once
            elif rnd == 4:
                GP._fit_device()                                 # a refit in between: the state must not survive it
            else:
                GP.addData(X[n], Y[n])
        assert seen[0] == "sweep2_kernel" and seen[1] == "sweep2_rank1_kernel" and seen[4] == "sweep2_rank1_kernel"
        assert seen[5] == "sweep2_kernel" and seen[6] == "sweep2_rank1_kernel"
        assert sweep(GP, dc, incremental=True)["kernel"] == "sweep2_rank1_kernel"      # the row added in the last round
        assert sweep(GP, dc, incremental=True)["kernel"] == "acq_finish_kernel"        # nothing new: acquisition only
        other = DeviceArray.from_host(np.random.RandomState(72).rand(9001, D))
        # (another array: a new state -- arg-max only, so in two parts where the model has the 512 padded rows for it)
        assert sweep(GP, other, incremental=True)["kernel"] == ("sweep2_kernel<part>" if N0 >= 500 else "sweep2_kernel")
    # the gallery is the caller: same picks with and without the kept state
    X, Y = synth(75, 600, 3)
    GP = GaussianProcess(K.GaussianKernel_ard([.25, .3, .35]), X, Y, noise=.1)
    cand = np.random.RandomState(76).rand(20000, 3)
    b = [[0., 1.]] * 3
    g_inc = np.array(fastUCBGallery(GP, b, 6, candidates=cand))
    g_full = np.array(fastUCBGallery(GP, b, 6, lhc_per_round=[cand] * 6))
    np.testing.assert_array_equal(g_inc, g_full)


def test_two_part_kept_state_finds_the_full_sweeps_maximum(ibo):
    """ibo_acq_sweep_incremental, arg-max only, EI / UCB: the state is formed over the first half of W's rows, the second half
    runs only for tiles whose bound can still win, and later rounds refresh only the tiles that can matter (gallery_prune = 1).  Every round of a gallery-like sequence returns the
    (value, index) of the same launches with every tile completed (gallery_prune = 2) BIT FOR BIT, the index of the one-kernel
    sweep (0) and of a freshly fitted model's full sweep, values at 1e-9; tiles really are left incomplete; a later call that
    wants per-candidate outputs completes them and equals the full sweep's outputs."""
    import ctypes
    from ibo_amd import DeviceArray, _lib
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.acquisition import sweep
    opt = lambda v: _lib.check(_lib.lib.ibo_set_option(b"gallery_prune", v))
    def state_info(GP):
        t, c = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib.ibo_sweep_state_info(GP._handle(), ctypes.byref(t), ctypes.byref(c)))
        return t.value, c.value
    try:
        for N0, D, kern, acq, kw in ((1000, 4, K.GaussianKernel_ard([.3] * 4), 'ei', dict(xi=.4, native=False)),
                                      (2040, 8, K.MaternKernel5([.5, 1.0]), 'ucb', dict()),
                                      (700, 2, K.MaternKernel3([.4, 1.0]), 'ei', dict(xi=.01))):
            X, Y = synth(170 + D, N0, D)
            cand = np.random.RandomState(171).rand(40000 + 7, D)
            runs = {}
            for mode in (1, 2, 0):
                opt(mode)
                GP = GaussianProcess(kern, X, Y, noise=.01)
                dc = DeviceArray.from_host(cand)
                out, shown = [], []
                for rnd in range(6):
                    excl = np.array(shown[:2]) if rnd >= 3 else None
                    r = sweep(GP, dc, acq=acq, exclude=excl, exclude_radius=.05, incremental=True, **kw)
                    tiles, done = state_info(GP)
                    out.append((r["best_val"], r["best_idx"], r["kernel"], tiles, done))
                    if mode == 1:        # against a freshly fitted model's full sweep
                        ref = GaussianProcess(kern, GP.X, GP.Y, noise=.01)
                        f = sweep(ref, dc, acq=acq, exclude=excl, exclude_radius=.05, **kw)
                        assert f["best_idx"] == r["best_idx"]; close(r["best_val"], f["best_val"], rtol=1e-9)
                    x = cand[r["best_idx"]]
                    shown.append(x)
                    # the gallery's hallucinated observation -- and once a real one, off the posterior mean: the stale tiles' means
                    # are then no bound any more, the drift margin says so, and nothing may be skipped on their account
                    GP.addData(x, GP.mu(x) + (0.3 if rnd == 2 else 0.0))
                if mode == 1:
                    full = sweep(GP, dc, acq=acq, incremental=True, outputs=("mu", "s2", "acq"), **kw)     # outputs: every tile completed first
                    assert state_info(GP)[0] == state_info(GP)[1]
                    ref = GaussianProcess(kern, GP.X, GP.Y, noise=.01)
                    f = sweep(ref, dc, acq=acq, outputs=("mu", "s2", "acq"), **kw)
                    close(full["mu"], f["mu"], rtol=1e-9, atol=1e-10); close(full["s2"], f["s2"], rtol=1e-9); close(full["acq"], f["acq"], rtol=1e-9, atol=ACQ_ATOL)
                    assert full["best_idx"] == f["best_idx"]
                runs[mode] = out
            for a, b, c in zip(runs[1], runs[2], runs[0]):
                assert a[0] == b[0] and a[1] == b[1], (a, b)                 # pruned == complete, bit for bit
                assert a[1] == c[1]; close(a[0], c[0], rtol=1e-9)
            assert runs[1][0][2] == "sweep2_kernel<part>" and runs[0][0][2] == "sweep2_kernel" and runs[1][1][2] == "sweep2_rank1_kernel"
            assert all(t == d for _, _, _, t, d in runs[2]) and all(t == d for _, _, _, t, d in runs[0])
            assert runs[1][0][4] < runs[1][0][3] // 2, runs[1][0]            # more than half of the tiles never ran their second part
            assert all(x[4] <= y[4] for x, y in zip(runs[1], runs[1][1:]) if y[2] != "sweep2_kernel<part>")  # completion only grows (until a refit starts a new state)
    finally:
        opt(1)


def test_two_part_kept_state_edge_cases(ibo):
    """the pruned state where its bookkeeping could slip: a candidate count that is no multiple of the tile, every candidate
    duplicated in another tile (the LOWER index must win the tie, so the tile of the first copy may not be left incomplete), most
    of the array inside exclusion balls, all of it inside them (nothing admissible), and PI on a pruned state (not monotone in the
    variance: every tile is completed first)."""
    import ctypes
    from ibo_amd import DeviceArray, _lib
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.acquisition import sweep
    opt = lambda v: _lib.check(_lib.lib.ibo_set_option(b"gallery_prune", v))
    def state_info(GP):
        t, c = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib.ibo_sweep_state_info(GP._handle(), ctypes.byref(t), ctypes.byref(c)))
        return t.value, c.value
    X, Y = synth(181, 900, 3)
    base = np.random.RandomState(182).rand(10000 + 13, 3)
    cand = np.vstack([base, base])                                  # every point twice, 10013 rows apart: never in the same tile
    kern = K.GaussianKernel_ard([.3] * 3)
    try:
        res = {}
        for mode in (1, 0):
            opt(mode)
            GP = GaussianProcess(kern, X, Y, noise=.01)
            dc = DeviceArray.from_host(cand)
            out = []
            r = sweep(GP, dc, acq='ei', xi=.1, native=False, incremental=True); out.append((r["best_idx"], r["best_val"]))
            assert r["best_idx"] < len(base)                         # first of the two equal maxima
            GP.addData(cand[r["best_idx"]], GP.mu(cand[r["best_idx"]]))
            far = cand[np.linalg.norm(cand - cand[r["best_idx"]], axis=1) > .9][:1]
            balls = np.vstack([cand[r["best_idx"]][None, :], far])
            r = sweep(GP, dc, acq='ei', xi=.1, native=False, exclude=balls, exclude_radius=.6, incremental=True); out.append((r["best_idx"], r["best_val"]))
            r = sweep(GP, dc, acq='ei', xi=.1, native=False, exclude=np.full((1, 3), .5), exclude_radius=5., incremental=True)
            out.append((r["best_idx"], r["best_val"]))               # nothing admissible
            if mode == 1:
                t, d = state_info(GP); assert d < t
            r = sweep(GP, dc, acq='pi', xi=.05, native=False, incremental=True); out.append((r["best_idx"], r["best_val"]))
            if mode == 1:
                t, d = state_info(GP); assert d == t                 # PI: everything completed
            f = sweep(GaussianProcess(kern, GP.X, GP.Y, noise=.01), dc, acq='pi', xi=.05, native=False)
            assert f["best_idx"] == r["best_idx"]; close(r["best_val"], f["best_val"], rtol=1e-9)
            res[mode] = out
        for a, b in zip(res[1], res[0]):
            assert a[0] == b[0]
            if np.isfinite(b[1]): close(a[1], b[1], rtol=1e-9)
            else: assert a[1] == b[1]
    finally:
        opt(1)


def test_incremental_state_cannot_alias_another_array(ibo):
    """the kept (q, alphaY.k*, alpha1.k*) state is keyed on the candidate array's GENERATION (ibo_dev_generation), not on its
    address: an array freed and reallocated at the same address, one overwritten in place through ibo_memcpy_h2d,
    another ndarray of the same shape, and ibo_gp_set_y all force the full sweep -- with the right values"""
    from ibo_amd import _lib
    from ibo_amd._lib import DeviceArray
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.acquisition import sweep
    X, Y = synth(81, 400, 3)
    kern = K.GaussianKernel_ard([.3, .35, .4])
    GP = GaussianProcess(kern, X[:390], Y[:390], noise=.1)
    A = np.random.RandomState(82).rand(9001, 3)
    B = np.random.RandomState(83).rand(9001, 3)

    def check(r, cand_host):
        f = sweep(GaussianProcess(kern, GP.X, GP.Y, noise=.1), cand_host, acq='ei', xi=.2, outputs=("mu", "s2", "acq"))
        close(r["mu"], f["mu"], rtol=1e-9, atol=1e-10); close(r["s2"], f["s2"], rtol=1e-9); close(r["acq"], f["acq"], rtol=1e-9, atol=ACQ_ATOL)
        assert r["best_idx"] == f["best_idx"]

    kw = dict(acq='ei', xi=.2, incremental=True, outputs=("mu", "s2", "acq"))
    dA = DeviceArray.from_host(A)
    genA, addrA = dA.generation(), dA.ptr.value
    assert genA > 0
    assert sweep(GP, dA, **kw)["kernel"] == "sweep2_kernel"
    GP.addData(X[390], Y[390])
    r = sweep(GP, dA, **kw); assert r["kernel"] == "sweep2_rank1_kernel"; check(r, A)
    # (1) free, allocate another array of the same shape: whatever address it gets, it is a different array
    dA.free()
    dB = DeviceArray.from_host(B)
    same_address = dB.ptr.value == addrA
    assert dB.generation() not in (0, genA)
    GP.addData(X[391], Y[391])
    r = sweep(GP, dB, **kw); assert r["kernel"] == "sweep2_kernel", (r["kernel"], same_address); check(r, B)
    GP.addData(X[392], Y[392])
    r = sweep(GP, dB, **kw); assert r["kernel"] == "sweep2_rank1_kernel"; check(r, B)
    # (2) overwritten in place: new generation, full sweep, values of the NEW contents
    g0 = dB.generation()
    dB.upload(A)
    assert dB.generation() != g0
    r = sweep(GP, dB, **kw); assert r["kernel"] == "sweep2_kernel"; check(r, A)
    r = sweep(GP, dB, **kw); assert r["kernel"] == "acq_finish_kernel"; check(r, A)
    # (3) ndarrays: uploaded to a temporary each time, never incremental, each with its own values
    for arr in (A, B, A):
        r = sweep(GP, arr, **kw); assert r["kernel"] == "sweep2_kernel"; check(r, arr)
    # (4) new targets through ibo_gp_set_y: the kept means belong to the old alpha vectors
    r = sweep(GP, dB, **kw); assert r["kernel"] == "acq_finish_kernel"    # (the temporaries never touched dB's state)
    Y2 = np.array(GP.Y) * 0.5 + 0.1
    _lib.check(_lib.lib.ibo_gp_set_y(GP._handle(), _lib.dp(_lib.f64(Y2))))
    r = sweep(GP, dB, **kw); assert r["kernel"] == "sweep2_kernel"
    f = sweep(GaussianProcess(kern, GP.X, Y2, noise=.1), A, acq='ei', xi=.2, outputs=("mu",))
    close(r["mu"], f["mu"], rtol=1e-9, atol=1e-10)
    # (5) a view into the same allocation at another offset is another array; memory of unknown origin has no generation
    v = dB.view_rows(1, 9001)
    assert sweep(GP, v, **kw)["kernel"] == "sweep2_kernel"
    gen = ctypes.c_uint64(7)
    _lib.check(_lib.lib.ibo_dev_generation(0, ctypes.c_void_p(A.ctypes.data), ctypes.byref(gen)))        # (a host address)
    assert gen.value == 0


def _longdouble_posterior(R, X, Y, kfun, cand, noise):
    """mu and 1 + noise - |L^-1 k*|^2 in 80-bit arithmetic (Cholesky and forward substitutions by hand): what the right
    answer is where the reference's two paths disagree with each other"""
    ld = np.longdouble
    N = len(Y)
    A = R.astype(ld)
    L = np.zeros((N, N), dtype=ld)
    for j in range(N):
        L[j, j] = np.sqrt(A[j, j] - np.dot(L[j, :j], L[j, :j]))
        if j + 1 < N:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j].dot(L[j, :j])) / L[j, j]

    def fsolve(b):
        z = np.zeros(N, dtype=ld)
        for i in range(N):
            z[i] = (b[i] - np.dot(L[i, :i], z[:i])) / L[i, i]
        return z
    zy = fsolve(Y.astype(ld))
    mu = np.zeros(len(cand)); s2 = np.zeros(len(cand))
    for c in range(len(cand)):
        z = fsolve(kfun(X, cand[c]).astype(ld))
        mu[c] = float(np.dot(z, zy)); s2[c] = float((ld(1) + ld(noise)) - np.dot(z, z))
    return mu, s2


def test_tolerance_where_conditioning_is_worst(ibo, oracle):
    """N in {1000, 2048} x noise in {1e-3, 1e-4} x D in {1, 2}, three tight clusters of near-duplicate points, SE-ARD and
    Matern-5/2 (the reference's default noise reaches 1e-4, ego/gaussianprocess/__init__.py:83).  cond(R) ~ N / noise
    up to 2e7 and sigma^2 = 1 + noise - q is a difference of two numbers that agree to four digits.
    Measured (tools/tolerance_probe.py, profiles/r03_tolerance_probe.txt): the device path stays within 3e-10 (sigma^2) and
    6e-9 (mu) of an 80-bit reference, and of the reference's PYTHON path (two triangular solves,
    ego/gaussianprocess/__init__.py:205-212); the reference's NATIVE path contracts with the explicit inv(R)
    (ego/acquisition/__init__.py:385-388, cpp/optimizeGP.cpp:141-170) and is itself 2.6e-6 .. 5.8e-6 away from both at
    noise 1e-4.  So the 1e-6 bar is asserted against the Python path and the 80-bit values everywhere, and against the native
    path wherever that path is itself within 1e-7 of the 80-bit values (all noise-1e-3 cases); elsewhere the device may be
    no farther from the native values than the native values are from the truth."""
    from ibo_amd.gaussianprocess import GaussianProcess, kernel as K
    from ibo_amd.acquisition import sweep
    worst = dict(py_mu=0., py_s2=0., py_ei=0., truth_mu=0., truth_s2=0., nat_s2_where_native_is_right=0., nat_ei_where_native_is_right=0., native_own_s2=0.)

    def rel(a, b, floor):
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
    for N in (1000, 2048):
        for noise in (1e-3, 1e-4):
            for D in (1, 2):
                for kind in ("ard", "m5"):
                    rs = np.random.RandomState(1000 * D + N + int(1e5 * noise))
                    c = rs.rand(3, D)
                    X = np.clip(np.vstack([c[i] + 0.02 * rs.randn(N // 4, D) for i in range(3)] + [rs.rand(N - 3 * (N // 4), D)]), 0, 1)
                    Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
                    hyp = np.full(D, .3) if kind == "ard" else np.r_[.5, 1.0]
                    okern = oracle.Kern(kind, hyp)
                    ogp = oracle.GP(okern, X, Y, noise=noise)
                    gp = GaussianProcess(K.GaussianKernel_ard(hyp) if kind == "ard" else K.MaternKernel5(hyp), X, Y, noise=noise)
                    M = 24
                    cand = np.vstack([rs.rand(M // 2, D), np.clip(X[rs.randint(0, N, M // 2)] + 1e-3 * rs.randn(M // 2, D), 0, 1)])
                    r_nat = sweep(gp, cand, acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
                    r_py = sweep(gp, cand, acq='ei', xi=.01, native=False, outputs=("mu", "s2", "acq"))
                    o_nat = oracle.sweep_native(ogp, cand, oracle.ACQ_EI, .01)
                    pm, ps = ogp.posteriors(cand)
                    o_ei = oracle.acq_value(oracle.ACQ_EI, oracle.ERF_NR, pm, np.sqrt(ps), np.max(Y), .01)
                    t_mu, t_s2 = _longdouble_posterior(ogp.R, X, Y, lambda Xm, q: np.array([okern.cov(x, q) for x in Xm]), cand, noise)
                    case = (N, noise, D, kind)
                    # the Python path and the 80-bit values: the bar, everywhere
                    e = rel(r_py["mu"], pm, 1e-9); worst["py_mu"] = max(worst["py_mu"], e); assert e < RT, case
                    e = rel(r_py["s2"], ps, 1e-300); worst["py_s2"] = max(worst["py_s2"], e); assert e < RT, case
                    live = np.abs(o_ei) > ACQ_ATOL
                    e = rel(r_py["acq"] * live, o_ei * live, ACQ_ATOL); worst["py_ei"] = max(worst["py_ei"], e); assert e < RT, case
                    e = rel(r_nat["mu"], t_mu, 1e-9); worst["truth_mu"] = max(worst["truth_mu"], e); assert e < 1e-7, case
                    e = rel(r_nat["s2"], np.clip(t_s2, 1e-8, 10), 1e-300); worst["truth_s2"] = max(worst["truth_s2"], e); assert e < 1e-8, case
                    # the native path: the bar where it is itself right, its own error elsewhere
                    own = rel(o_nat["s2"], np.clip(t_s2, 1e-8, 10), 1e-300)
                    worst["native_own_s2"] = max(worst["native_own_s2"], own)
                    e = rel(r_nat["s2"], o_nat["s2"], 1e-300)
                    live = np.abs(o_nat["acq"]) > ACQ_ATOL
                    ea = rel(r_nat["acq"] * live, o_nat["acq"] * live, ACQ_ATOL)
                    if own < 1e-7:
                        worst["nat_s2_where_native_is_right"] = max(worst["nat_s2_where_native_is_right"], e)
                        worst["nat_ei_where_native_is_right"] = max(worst["nat_ei_where_native_is_right"], ea)
                        assert e < RT and ea < RT, (case, e, ea)
                    else:
                        assert e < 1.05 * own + 1e-9, (case, e, own)
                    assert r_nat["best_idx"] == int(np.argmax(r_nat["acq"]))
    print("worst relative errors where conditioning is worst:", {k: "%.2e" % v for k, v in worst.items()})
    assert worst["native_own_s2"] > 1e-6        # the premise: the reference's two paths do part ways here


def test_sweep_index_base_beyond_32_bits(ibo):
    """global indices of a shard far into a huge candidate set: index_base > 2^31 (and > 2^32) is carried in
    64 bits through the kernel's partials, the final reduction and the ABI"""
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import sweep
    X, Y = synth(61, 96, 3)
    GP = GaussianProcess(GaussianKernel_ard([.3] * 3), X, Y, noise=.1)
    for M in (9, 700, 20000):                                        # GEMV, panel-split and tile kernels
        cand = np.random.RandomState(62).rand(M, 3)
        r0 = sweep(GP, cand)
        for base in ((1 << 31) + 5, (1 << 33) + 123456789, (1 << 52)):
            r = sweep(GP, cand, index_base=base)
            assert r["best_idx"] == base + r0["best_idx"] and r["best_val"] == r0["best_val"]


def test_g7_preference_gp(ibo, oracle):
    from ibo_amd.gaussianprocess import PrefGaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import sweep
    from ibo_amd.acquisition.gallery import fastUCBGallery
    g = load_golden("g7_prefs")
    for name in g["names"]:
        p = str(name) + "/"
        prefs = [(v, u, d) for v, u, d in zip(g[p + "pref_v"], g[p + "pref_u"], g[p + "pref_d"])]
        GP = PrefGaussianProcess(GaussianKernel_ard(g[p + "hyper"]))
        GP.addPreferences(prefs)
        close(GP.X, g[p + "X"], rtol=0, atol=0)
        # the MAP: convex S, our optimum must not be worse than the reference's
        ok = oracle.Kern("ard", g[p + "hyper"])
        ogp = oracle.pref_fit(ok, prefs, noise=.1, Y_map=g[p + "Y"])
        Lr = np.linalg.cholesky(ogp.R)
        S_ref = oracle.pref_S(g[p + "Y"], ogp.inds, Lr)
        S_our = oracle.pref_S(GP.Y, ogp.inds, Lr)
        assert S_our <= S_ref + 1e-6 * max(1.0, abs(S_ref))
        # everything downstream of the MAP at 1e-6: feed the reference's Y_map
        GP._set_map(g[p + "Y"], ogp.inds)
        close(GP.C, g[p + "C"], atol=1e-9); close(GP.R, g[p + "R"], rtol=1e-12); close(GP.L, g[p + "L"], atol=1e-9)
        mu, s2 = GP.posteriors(g[p + "probe"])
        close(mu, g[p + "post"][:, 0], atol=1e-9); close(s2, g[p + "post"][:, 1])
        r = sweep(GP, g[p + "probe"], acq='ei', xi=.01, native=True, outputs=("acq",))
        close(r["acq"], g[p + "ei_c"], atol=ACQ_ATOL)
        gal = fastUCBGallery(GP, g[p + "bounds"].tolist(), 4, lhc_per_round=list(g[p + "lhc"]))
        close(np.array(gal), g[p + "gallery"], atol=1e-9)


def test_pref_orderings(ibo):
    """ego/unittest_GP.py:398-441"""
    from ibo_amd.gaussianprocess import PrefGaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    x1, x2, x3, x4, x5, x6 = (np.array([v]) for v in (.2, .7, .4, .35, .9, .1))
    GP = PrefGaussianProcess(GaussianKernel_ard(np.array([.1])))
    GP.addPreferences([(x1, x2, 0)])
    assert GP.mu(x1) > GP.mu(x2)
    GP.addPreferences([(x3, x4, 0)])
    assert GP.mu(x1) > GP.mu(x2) and GP.mu(x3) > GP.mu(x4)
    GP.addPreferences([(x5, x6, 1)])
    assert GP.mu(x1) > GP.mu(x2) and GP.mu(x3) > GP.mu(x4) and GP.mu(x5) > GP.mu(x6)
    assert GP.mu(x5) - GP.mu(x6) > GP.mu(x1) - GP.mu(x2)
    assert GP.mu(x5) - GP.mu(x6) > GP.mu(x3) - GP.mu(x4)
    with pytest.raises(NotImplementedError):
        GP.addData(x1, 1.0)


def test_pref_device_steps_match_dense_algebra(ibo):
    """ibo_pref_*: one Newton step and the final factorisation of the preference GP against the same algebra in
    NumPy -- delta = -(R^-1 + sum rho (e_v - e_u)(e_v - e_u)^T)^-1 g, R^-1 delta, L = chol(R + (5 I + sum w ...)^-1);
    points that occur in several pairs (accumulated entries) and a size that is not a multiple of 64"""
    import ctypes
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import GaussianProcess, PrefGaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    n, D, P = 150, 3, 220
    rs = np.random.RandomState(77)
    X = rs.rand(n, D)
    GP = GaussianProcess(GaussianKernel_ard([.4] * D), X, rs.randn(n), noise=.05)
    h = GP._handle()
    _lib.check(_lib.lib.ibo_pref_begin(h))
    R = GP.R
    Rinv = np.linalg.inv(R)
    v = rs.randint(0, n, P); u = (v + 1 + rs.randint(0, n - 1, P)) % n
    rho = rs.rand(P) + .1
    g = rs.randn(n)
    lin, val = PrefGaussianProcess._pair_sum_entries(n, v, u, rho)
    assert len(lin) < 4 * P                                         # some entries really are sums
    H = Rinv.copy()
    for a, b, r in zip(v, u, rho):
        H[a, a] += r; H[b, b] += r; H[a, b] -= r; H[b, a] -= r
    delta = np.empty(n); rdelta = np.empty(n); out = np.empty(n); info = ctypes.c_int(0)
    i64 = ctypes.POINTER(ctypes.c_int64)
    _lib.check(_lib.lib.ibo_pref_newton_step(h, len(lin), lin.ctypes.data_as(i64), _lib.dp(val), _lib.dp(g), _lib.dp(delta),
                                             _lib.dp(rdelta), ctypes.byref(info)))
    ref = -np.linalg.solve(H, g)
    close(delta, ref, rtol=1e-8, atol=1e-10); close(rdelta, Rinv.dot(ref), rtol=1e-7, atol=1e-9)
    _lib.check(_lib.lib.ibo_pref_rinv_mul(h, _lib.dp(g), _lib.dp(out)))
    close(out, Rinv.dot(g), rtol=1e-8, atol=1e-10)
    C = 5 * np.eye(n)
    for a, b, r in zip(v, u, rho):
        C[a, a] += r; C[b, b] += r; C[a, b] -= r; C[b, a] -= r
    _lib.check(_lib.lib.ibo_pref_finish(h, len(lin), lin.ctypes.data_as(i64), _lib.dp(val), 5.0, ctypes.byref(info)))
    L = np.empty((n, n))
    _lib.check(_lib.lib.ibo_gp_get_L(h, _lib.dp(L)))
    close(np.tril(L), np.linalg.cholesky(R + np.linalg.inv(C)), rtol=1e-9, atol=1e-11)
    # a matrix entry out of range, a step without a begin since the last fit: refused
    bad = np.array([n * n], dtype=np.int64)
    assert _lib.lib.ibo_pref_finish(h, 1, bad.ctypes.data_as(i64), _lib.dp(np.ones(1)), 5.0, ctypes.byref(info)) == _lib.ERR_ARG
    GP._fit_device()
    assert _lib.lib.ibo_pref_rinv_mul(h, _lib.dp(g), _lib.dp(out)) == _lib.ERR_STATE


def test_g2_g8_marginal_likelihood(ibo):
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood, nlml_grid, nlml
    g = load_golden("g2_hyper")
    X, Y = g["X"], g["Y"]
    for name, cls, nh in (("ard", K.GaussianKernel_ard, 3), ("sviso", K.SVGaussianKernel_iso, 2),
                          ("m3", K.MaternKernel3, 2), ("m5", K.MaternKernel5, 2)):
        k = cls(g[name + "_hyper"])
        for h in range(nh):
            close(k.derivative(X, h), g["%s_d%d" % (name, h)], atol=1e-12)
        for tag, noise in (("n0", 0.0), ("n1", 1e-3)):
            v, d = marginalLikelihood(k, X, Y, nh, True, noise=noise)
            close(v, g["%s_%s_nlml" % (name, tag)]); close(d, g["%s_%s_grad" % (name, tag)], atol=1e-9)
    g8 = load_golden("g8_nlml")
    for N in (64, 256):
        Xs, Ys = synth(5, N, 16)
        vals, am = nlml_grid(K.GaussianKernel_ard, g8["n%d_theta" % N], Xs, Ys, noise=1e-3)
        close(vals, g8["n%d_nlml" % N]); assert am == int(np.argmin(g8["n%d_nlml" % N]))
    Xs, Ys = synth(5, 64, 16)
    v, d = marginalLikelihood(K.GaussianKernel_ard(g8["n64_grad_theta"]), Xs, Ys, 16, True, noise=1e-3)
    close(v, g8["n64_grad_nlml"]); close(d, g8["n64_grad"], atol=1e-9)
    # hyper-parameter learning loop (ego/unittest_GP.py:216-219): BFGS on log-theta with the device
    # value and gradient must reach the optimum the oracle's own BFGS finds
    from scipy import optimize
    from ibo_amd.gaussianprocess.trainhyper import dnlml
    x0 = np.log([2., 2., .1, 1.])
    ours = optimize.fmin_bfgs(nlml, x0, dnlml, args=(K.SVGaussianKernel_ard, X, Y), disp=False)
    import oracle.oracle as orc
    f_o = lambda lh: orc.marginal_likelihood(orc.Kern("svard", np.exp(lh)), X, Y, 4, False, 1e-3)
    g_o = lambda lh: orc.marginal_likelihood(orc.Kern("svard", np.exp(lh)), X, Y, 4, True, 1e-3)[1]
    ref = optimize.fmin_bfgs(f_o, x0, g_o, disp=False)
    close(f_o(ours), f_o(ref), rtol=1e-6)
    # the optima the reference's own test prints (ego/unittest_GP.py:216-219): [6.9714, 0.95405, -0.9769, 0.36469].
    # The first log length scale sits on a plateau (the data hardly vary along that axis): today's SciPy line
    # search stops at 8.9 on the CPU oracle too, where the reference's 2010 SciPy stopped at 6.97 -- the NLML
    # differs by 2e-6 between the two -- so that component is only required to be "large"; the other three are
    # pinned to the printed 2 decimals.
    assert ours[0] > 5.0
    np.testing.assert_allclose(ours[1:], [0.95405, -0.9769, 0.36469], atol=5e-3)
    # ... and the isotropic case (:250-252): [-0.0893, 0.29]
    ours_iso = optimize.fmin_bfgs(nlml, np.log([1.5, 1.1]), dnlml, args=(K.SVGaussianKernel_iso, X, Y), disp=False)
    np.testing.assert_allclose(ours_iso, [-0.0893, 0.29], atol=5e-3)
    # not-PD -> 100 through the nlml() wrapper (trainhyper.py:111-114).  A coincident pair plus a negative
    # "noise" makes K indefinite beyond any rounding doubt (2x2 minor [[.5, 1], [1, .5]])
    Xd = np.vstack([X, X[:1]])
    Yd = np.r_[Y, 1.0]
    from ibo_amd.gaussianprocess import trainhyper
    with pytest.raises(np.linalg.LinAlgError):
        marginalLikelihood(K.GaussianKernel_ard([2., 2., .1]), Xd, Yd, 3, True, noise=-0.5)
    with pytest.raises(np.linalg.LinAlgError):
        marginalLikelihood(K.GaussianKernel_ard([2., 2., .1]), Xd, Yd, 3, False, noise=-0.5)
    real = trainhyper.marginalLikelihood
    try:
        trainhyper.marginalLikelihood = lambda k, X_, Y_, n, computeGradient=True, **kw: real(k, X_, Y_, n, computeGradient, noise=-0.5)
        assert nlml(np.log([2., 2., .1]), K.GaussianKernel_ard, Xd, Yd) == 100
    finally:
        trainhyper.marginalLikelihood = real
    # the value/gradient memo is keyed on the data's CONTENT: an in-place edit that keeps every sum must not
    # return the stale pair
    Xa = np.array(X, dtype=float)
    lh = np.log([2., 2., .1])
    v1 = nlml(lh, K.GaussianKernel_ard, Xa, Y)
    Xa[0, 0], Xa[0, 1] = Xa[0, 1], Xa[0, 0]          # same row sum, same id(), different matrix
    v2 = nlml(lh, K.GaussianKernel_ard, Xa, Y)
    close(v2, marginalLikelihood(K.GaussianKernel_ard([2., 2., .1]), Xa, Y, 3, False), rtol=1e-12)
    assert v1 != v2


def test_legacy_acqmaxGP_symbol(ibo, oracle):
    """the .so under the reference's own ctypes signature (ego/acquisition/__init__.py:343-436)"""
    from ibo_amd import _lib
    g = load_golden("g3_cases")
    DP = ctypes.POINTER(ctypes.c_double)
    libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
    for name in ("s24_ard2d", "s0_shekel_iso3", "s0_branin_m3_n1e2", "s512_prior_ard"):
        p = name + "/"
        f64 = _lib.f64; dp = _lib.dp
        X = f64(g[p + "X"]); Y = f64(g[p + "Y"]); invR = f64(np.linalg.inv(g[p + "R"]))
        b = g[p + "bounds"]; lb = f64(b[:, 0]); ub = f64(b[:, 1]); hyp = f64(g[p + "hyper"])
        D = X.shape[1]
        if p + "pmeans" in g.files:
            nb = len(g[p + "pbeta"]); pm = f64(g[p + "pmeans"].reshape(-1)); pb = f64(g[p + "pbeta"])
            pl = f64(g[p + "plowerb"]); pw = f64(g[p + "pwidth"]); pt = float(g[p + "ptheta"])
        else:
            nb = 0; pm = pb = pl = pw = np.zeros(1); pt = 0.0
        for acq, key, parm in ((0, "max_ei", .01), (1, "max_pi", .01), (2, "max_ucb", float(g[p + "ucb_parm"]))):
            r = _lib.lib.acqmaxGP(D, dp(lb), dp(ub), dp(invR), dp(X), dp(Y), len(Y), acq, int(g[p + "ktype"]), dp(hyp),
                                  nb, dp(pm), dp(pb), pt, dp(pl), dp(pw), parm, float(g[p + "noise"]), 10, 30, 10000)
            assert bool(r)
            res = np.array([r[i] for i in range(D + 1)])
            libc.free(r)
            close(-res[0], g[p + key][0], atol=ACQ_ATOL); close(res[1:], g[p + key][1:], rtol=1e-9, atol=1e-12)
    assert not _lib.lib.acqmaxGP(D, dp(lb), dp(ub), dp(invR), dp(X), dp(Y), len(Y), 7, 0, dp(hyp), 0, dp(pm), dp(pb),
                                 0.0, dp(pl), dp(pw), .01, .1, 1, 1, 10)


def test_direct_sample_counts_match_oracle(ibo, oracle):
    """DIRECT on the GPU objective takes the same number of samples as the sequential CPU run"""
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import gpuDirectGP
    X, Y = synth(21, 48, 3)
    GP = GaussianProcess(GaussianKernel_ard([.25, .3, .35]), X, Y)
    ogp = oracle.GP(oracle.Kern("ard", [.25, .3, .35]), X, Y)
    for bounds in ([[0., 1.]] * 3, [[0., 1.], [.5, .5], [0., 1.]], [[.5, .5], [0., 1.], [0., 1.]]):
        o, ox, ons = oracle.acqmax_native(ogp, bounds, oracle.ACQ_EI, .01, maxiter=12)
        v, x, ns = gpuDirectGP(GP, bounds, 12, 30, 10000, acqfunc='ei', xi=.01, return_samples=True)
        assert ns == ons
        close(v, o, atol=ACQ_ATOL); close(x, ox, rtol=1e-9, atol=1e-12)
    # compat=False: dimension 0 fixed no longer stalls
    v, x, ns = gpuDirectGP(GP, [[.5, .5], [0., 1.], [0., 1.]], 12, 30, 10000, acqfunc='ei', xi=.01, compat=False,
                           return_samples=True)
    assert ns > 17 and x[0] == .5


def _server_info(GP):
    from ibo_amd import _lib
    import ctypes
    n = ctypes.c_int(); why = ctypes.c_char_p()
    _lib.check(_lib.lib.ibo_direct_server_info(GP._handle(), ctypes.byref(n), ctypes.byref(why)))
    return n.value, (why.value or b"").decode()


def test_resident_evaluation_server_of_direct_max(ibo, oracle):
    """ibo_set_option("direct_resident", 1): DIRECT's batches go to a kernel that stays on the chip for the call (csrc/small2.hip
    direct_server_kernel; the objective of cpp/direct.cpp:372-498 behind ego/acquisition/__init__.py:174-197) instead of three launches
    each.  Same (opt, optx) BIT FOR BIT as the launches -- small model (one hand-over), larger ones (two), Matern, a mean prior, PI and UCB,
    eight and sixteen dimensions -- and against the oracle's maximiser; the server really took the batches; a model it does not take
    (17 dimensions) goes by launches without a word.  BOUNDED WAITS: with the host silent after three batches for longer than the
    kernel's idle deadline (IBO_SRV_STALL_AFTER, what a dead caller looks like from the device) the kernel has left, the call finishes by
    launches, and the result is still the same."""
    import os
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.gaussianprocess.prior import RBFNMeanPrior
    from ibo_amd.acquisition import maximizeEI, maximizePI, maximizeUCB

    def both(f, GP, *a, **kw):
        out = []
        for mode in (1, 0):
            _lib.check(_lib.lib.ibo_set_option(b"direct_resident", mode))
            try:
                out.append(f(GP, *a, **kw))
                out.append(_server_info(GP))
            finally:
                _lib.check(_lib.lib.ibo_set_option(b"direct_resident", 0))
        (r1, i1, r0, i0) = out
        assert r1[0] == r0[0] and np.array_equal(np.asarray(r1[1]), np.asarray(r0[1])), (r1, r0)
        assert i0[0] == 0
        return r1, i1
    cases = [(60, 2, K.GaussianKernel_ard([.3, .3])), (500, 3, K.GaussianKernel_ard([.3] * 3)), (1024, 4, K.GaussianKernel_ard([.3] * 4)),
             (1500, 8, K.MaternKernel5([.5, 1.0])), (700, 6, K.MaternKernel3([.6, 1.0])), (900, 16, K.GaussianKernel_ard([.9] * 16))]
    for N, D, kern in cases:
        X, Y = synth(70 + N, N, D)
        GP = GaussianProcess(kern, X, Y, noise=.1)
        b = [[0., 1.]] * D
        r, info = both(maximizeEI, GP, b, maxiter=25)
        assert info[0] >= 20 and info[1] == "", (N, D, info)
    X, Y = synth(71, 300, 4)
    ogp = oracle.GP(oracle.Kern("ard", [.3] * 4), X, Y, noise=.1)
    GP = GaussianProcess(K.GaussianKernel_ard([.3] * 4), X, Y, noise=.1)
    r, info = both(maximizeEI, GP, [[0., 1.]] * 4, maxiter=20)
    o, ox, _ = oracle.acqmax_native(ogp, [[0., 1.]] * 4, oracle.ACQ_EI, .01, maxiter=20)
    close(r[0], o, atol=1e-12); close(r[1], ox, rtol=1e-9, atol=1e-12)
    both(maximizePI, GP, [[0., 1.]] * 4, maxiter=15); both(maximizeUCB, GP, [[0., 1.]] * 4, maxiter=15)
    # a mean prior (its basis functions read the candidates again in the finish)
    pr = RBFNMeanPrior(); pr.means = np.random.RandomState(5).rand(6, 4); pr.beta = np.random.RandomState(6).randn(6); pr.theta = 2.0
    pr.lowerb = np.zeros(4); pr.width = np.ones(4)
    GPp = GaussianProcess(K.GaussianKernel_ard([.3] * 4), X, Y, noise=.1, prior=pr)
    r, info = both(maximizeEI, GPp, [[0., 1.]] * 4, maxiter=15)
    assert info[0] > 0
    # outside the server's instantiations: by launches, silently
    X17, Y17 = synth(72, 200, 19)
    GP17 = GaussianProcess(K.GaussianKernel_ard([1.2] * 19), X17, Y17, noise=.1)
    r, info = both(maximizeEI, GP17, [[0., 1.]] * 19, maxiter=6)
    assert info[0] == 0 and info[1] != ""
    # the host falls silent in mid-call: the kernel leaves on its deadline, the call ends by launches with the same result
    _lib.check(_lib.lib.ibo_set_option(b"direct_idle_ms", 5))
    os.environ["IBO_SRV_STALL_AFTER"] = "3"
    try:
        r, info = both(maximizeEI, GP, [[0., 1.]] * 4, maxiter=20)
    finally:
        del os.environ["IBO_SRV_STALL_AFTER"]
        _lib.check(_lib.lib.ibo_set_option(b"direct_idle_ms", 20))
    assert info[0] == 3 and "deadline" in info[1], info


def test_rccl_argmax_world_of_one(ibo):
    """the RCCL exchange itself (csrc/comm.hip) on the one GPU this box has"""
    from ibo_amd.multigpu import RcclArgmax
    comm = RcclArgmax(1, 0, RcclArgmax.unique_id(), device=0)
    v, i, p, r = comm.argmax(1.25, 123456789012, [0.5, 0.25, 0.125])
    assert (v, i, r) == (1.25, 123456789012, 0) and p.tolist() == [0.5, 0.25, 0.125]
    v, i, p, r = comm.argmax(float('nan'), -1, [0.0])
    assert i == -1 and r == -1
    np.testing.assert_array_equal(comm.allreduce_sum([1.5, -2.0, 0.0]), [1.5, -2.0, 0.0])
    comm.barrier()
    # the sharded gallery and NLML grid run through the same exchange (one rank = the whole array)
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition.gallery import fastUCBGallery
    from ibo_amd.multigpu import sharded_gallery, sharded_nlml_grid
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid
    X, Y = synth(51, 80, 3)
    GP = GaussianProcess(GaussianKernel_ard([.3] * 3), X, Y)
    cand = np.random.RandomState(52).rand(3000, 3)
    b = [[0., 1.]] * 3
    g1 = np.array(fastUCBGallery(GP, b, 4, candidates=cand))
    g2 = np.array(sharded_gallery(GP, b, 4, cand, 0, comm))
    np.testing.assert_array_equal(g1, g2)
    # the one-call step (ibo_acq_sweep_exchange: the local arg-max goes from the sweep's output words into the all-reduce buffer on
    # the device) against sweep-then-exchange through host values: every field equal, with exclusion balls, on a kept state
    # (incremental), with an index base beyond 32 bits, and when every candidate is excluded (no admissible winner anywhere)
    from ibo_amd import DeviceArray
    from ibo_amd.multigpu import sharded_sweep
    dc = DeviceArray.from_host(cand)
    for kw in (dict(acq='ei', xi=.01, native=True), dict(acq='ei', xi=.4, native=False, exclude=cand[:3], exclude_radius=.3),
               dict(acq='ucb', native=True, incremental=True), dict(acq='ei', xi=.4, native=False, exclude=np.full((1, 3), .5), exclude_radius=5.)):
        fused = sharded_sweep(GP, dc, (1 << 40) + 7, comm, **kw)
        comm.device_exchange = False
        try:
            plain = sharded_sweep(GP, dc, (1 << 40) + 7, comm, **kw)
        finally:
            del comm.device_exchange
        assert fused["best_idx"] == plain["best_idx"] and fused["best_rank"] == plain["best_rank"], (kw, fused, plain)
        if plain["best_idx"] >= 0:
            assert fused["best_val"] == plain["best_val"] and fused["local"]["best_val"] == plain["local"]["best_val"]
            np.testing.assert_array_equal(fused["best_x"], plain["best_x"])
            np.testing.assert_array_equal(fused["best_x"], cand[fused["best_idx"] - (1 << 40) - 7])
            assert fused["local"]["best_idx"] == plain["local"]["best_idx"]
        else:
            assert fused["local"]["best_idx"] == -1
    thetas = np.random.RandomState(53).rand(5, 3) + .2
    v1, a1 = nlml_grid(GaussianKernel_ard, thetas, X, Y)
    v2, a2 = sharded_nlml_grid(GaussianKernel_ard, thetas, X, Y, comm)
    np.testing.assert_array_equal(v1, v2); assert a1 == a2
    comm.close()


def test_two_rank_rccl_bench_when_two_gpus_are_visible(ibo):
    """RCCL with more than one rank: bench.py's own launcher on two GPUs (the candidate-sharded sweep with its
    arg-max exchange, then the sharded C3 gallery and C5 grid of the `configs` block).  Needs two visible devices;
    the single-GPU boxes skip it (the slot protocol itself runs on two and three gloo ranks in the CPU suite)."""
    import json, os, subprocess, sys
    from ibo_amd import _lib
    from conftest import ROOT
    if _lib.device_count() < 2:
        pytest.skip("one GPU visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IBO_COMM_ID_FILE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j["n_gpus"] == 2 and j["rccl_nranks"] == 2 and j["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-extras"], capture_output=True, text=True, timeout=900, env=env)
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert j["value"] > 1.5 * j1["value"]                                # weak scaling: two shards, one exchange per step
    assert j["configs"]["c3_shard_sweep"]["value"] > 0 and j["configs"]["c5_nlml_grid"]["n_not_pd"] == 0


def test_errors_are_loud(ibo):
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd import NotPositiveDefinite
    with pytest.raises(ValueError):
        GaussianProcess(GaussianKernel_ard([1.]), X=np.zeros((2, 1)))
    GP = GaussianProcess(GaussianKernel_ard([1.]))
    assert GP.posterior(np.array([.3])) == (0.0, 1.0)
    X = np.array([[.1], [.1], [.2]])
    with pytest.raises(np.linalg.LinAlgError):
        GaussianProcess(GaussianKernel_ard([1.]), X, [1., 1., 2.], noise=-1.0)
    assert issubclass(NotPositiveDefinite, np.linalg.LinAlgError)
    # the round-6 entry points and switches refuse what they cannot do, with a message
    import ctypes
    from ibo_amd import _lib, IBOError
    GP = GaussianProcess(GaussianKernel_ard([1.]), np.array([[.1], [.4], [.8]]), [1., 2., 1.5])
    v = ctypes.c_double(); i = ctypes.c_int64(); r = ctypes.c_int(); x = np.zeros(1)
    rc = _lib.lib.ibo_acq_sweep_exchange(GP._handle(), None, 0, 1, None, 0, .01, 0, 1e-8, float('nan'), 0, None, .5, 0,
                                         ctypes.byref(v), ctypes.byref(i), ctypes.byref(v), ctypes.byref(i), _lib.dp(x), ctypes.byref(r))
    assert rc == 1 and _lib.lib.ibo_last_error()                  # IBO_ERR_ARG
    for key, bad in ((b"super_min_nb", 3), (b"direct_idle_ms", 0), (b"arena_mb", -1), (b"no_such_option", 1)):
        with pytest.raises(IBOError):
            _lib.check(_lib.lib.ibo_set_option(key, bad))
    ms = ctypes.c_double(-1.0)
    _lib.check(_lib.lib.ibo_gpu_time_ms(0, ctypes.byref(ms)))
    assert ms.value > 0.0                                   # (this process has fitted models: their device time is in it)
    assert _lib.lib.ibo_gpu_time_ms(0, None) == 1 and _lib.lib.ibo_direct_server_info(None, None, None) == 1


# --------------------------------------------------------------------------- full-size configurations
def _hartman6(x):
    A = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8], [17, 8, 0.05, 10, 0.1, 14]])
    P = np.array([[0.1312, 0.1696, 0.5569, 0.0124, 0.8283, 0.5886], [0.2329, 0.4135, 0.8307, 0.3736, 0.1004, 0.9991],
                  [0.2348, 0.1451, 0.3522, 0.2883, 0.3047, 0.6650], [0.4047, 0.8828, 0.8732, 0.5743, 0.1091, 0.0381]])
    C = np.array([1, 1.2, 3, 3.2])
    return float(np.sum(C * np.exp(-np.sum(A * (x - P) ** 2, axis=1))))


def test_c3_full_size_matern_gallery(ibo, oracle):
    """BASELINE config 3 on one GPU's shard: N=2048, D=8, Matern-5/2, 2^19 candidates, gallery"""
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import MaternKernel5
    from ibo_amd.acquisition import sweep
    from ibo_amd.acquisition.gallery import fastUCBGallery
    from ibo_amd import DeviceArray
    N, D, M = 2048, 8, 1 << 19
    X, Y = synth(3, N, D)
    GP = GaussianProcess(MaternKernel5([.5, 1.0]), X, Y, noise=.1)
    cand = np.random.RandomState(103).rand(M, D)
    dc = DeviceArray.from_host(cand)
    r = sweep(GP, dc, acq='ei', xi=.3, native=True, outputs=("mu", "s2", "acq"))
    assert r["best_idx"] == int(np.argmax(r["acq"]))
    # a slice of the sweep against the oracle (Python-path posterior, then libm EI)
    ogp = oracle.GP(oracle.Kern("m5", [.5, 1.0]), X, Y, noise=.1)
    idx = np.r_[np.arange(24), r["best_idx"]]
    o_mu, o_s2 = ogp.posteriors(cand[idx])
    close(r["mu"][idx], o_mu, atol=1e-9); close(np.maximum(r["s2"][idx], 1e-7), o_s2)
    o_ei = oracle.acq_value(oracle.ACQ_EI, oracle.ERF_LIBM, o_mu, np.sqrt(o_s2), Y.max(), .3)
    close(r["acq"][idx], o_ei, atol=ACQ_ATOL)
    # contiguous shards (what 8 GPUs would each sweep) combine to the same arg-max
    cuts = [0, 1 << 17, 3 << 17, M]
    parts = [sweep(GP, dc.view_rows(a, b), acq='ei', xi=.3, native=True, index_base=a) for a, b in zip(cuts, cuts[1:])]
    best = max(parts, key=lambda q: (q["best_val"], -q["best_idx"]))
    assert best["best_idx"] == r["best_idx"]
    gal = np.array(fastUCBGallery(GP, [[0., 1.]] * D, 4, candidates=dc))
    assert gal.shape == (4, D) and np.all(gal >= 0) and np.all(gal <= 1)
    assert min(np.linalg.norm(gal[i] - gal[j]) for i in range(4) for j in range(i)) > .5


def test_c4_preference_gp_128_pairs(ibo, oracle):
    """a larger preference GP (the oracle's BFGS with numerical gradients cannot go much further):
    the MAP must reach a lower S than the start, everything downstream of it is checked at 1e-6"""
    from ibo_amd.gaussianprocess import PrefGaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    P = 128
    hyp = [0.53, 0.57, 2.5, 0.34, 0.27, 0.35]
    pts = np.random.RandomState(4).rand(2 * P, 6)
    prefs = []
    for i in range(P):
        a, b = pts[2 * i], pts[2 * i + 1]
        prefs.append((a, b, 0) if _hartman6(a) > _hartman6(b) else (b, a, 0))
    GP = PrefGaussianProcess(GaussianKernel_ard(hyp), prefs)
    assert len(GP.X) == 2 * P and GP.C.shape == (2 * P, 2 * P)
    ogp = oracle.pref_fit(oracle.Kern("ard", hyp), prefs, noise=.1, Y_map=GP.Y)
    Lr = np.linalg.cholesky(ogp.R)
    start = np.array([.5 if i in set(v for v, _, _ in ogp.inds) else -.5 for i in range(2 * P)])
    assert oracle.pref_S(GP.Y, ogp.inds, Lr) < oracle.pref_S(start, ogp.inds, Lr)
    close(GP.C, ogp.C, atol=1e-9); close(GP.L, ogp.L, atol=1e-9)
    probe = np.random.RandomState(5).rand(40, 6)
    mu, s2 = GP.posteriors(probe)
    o_mu, o_s2 = ogp.posteriors(probe)
    close(mu, o_mu, atol=1e-9); close(s2, o_s2)
    assert sum(GP.mu(v) > GP.mu(u) for v, u, _ in prefs) >= 0.9 * P


def test_c4_preference_gp_512_pairs_full_size(ibo, oracle):
    """C4 at its stated size (512 pairs -> 1024 points, D=6): everything downstream of the MAP -- C, L = chol(R +
    C^-1), posterior mean/variance -- against the oracle fed with the same latent values; the MAP itself must
    lower S and respect the orderings (the reference's BFGS optimum is not pinned)"""
    from ibo_amd.gaussianprocess import PrefGaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    P = 512
    hyp = [0.53, 0.57, 2.5, 0.34, 0.27, 0.35]
    pts = np.random.RandomState(4).rand(2 * P, 6)
    prefs = []
    for i in range(P):
        a, b = pts[2 * i], pts[2 * i + 1]
        prefs.append((a, b, 0) if _hartman6(a) > _hartman6(b) else (b, a, 0))
    GP = PrefGaussianProcess(GaussianKernel_ard(hyp), prefs)
    assert len(GP.X) == 2 * P and GP.C.shape == (2 * P, 2 * P)
    ogp = oracle.pref_fit(oracle.Kern("ard", hyp), prefs, noise=.1, Y_map=GP.Y)
    np.testing.assert_array_equal(GP.X, ogp.X)                       # same point numbering
    close(GP.R, ogp.R, rtol=1e-12, atol=1e-14)
    close(GP.C, ogp.C, atol=1e-9); close(GP.L, ogp.L, atol=1e-9)
    Lr = np.linalg.cholesky(ogp.R)
    start = np.array([.5 if i in set(v for v, _, _ in ogp.inds) else -.5 for i in range(2 * P)])
    assert oracle.pref_S(GP.Y, ogp.inds, Lr) < oracle.pref_S(start, ogp.inds, Lr)
    probe = np.random.RandomState(5).rand(64, 6)
    mu, s2 = GP.posteriors(probe)
    o_mu, o_s2 = ogp.posteriors(probe)
    close(mu, o_mu, atol=1e-9); close(s2, o_s2)
    assert sum(GP.mu(v) > GP.mu(u) for v, u, _ in prefs[:128]) >= 0.9 * 128


def test_rbfn_prior_train_reproduces_the_reference(ibo):
    """RBFNMeanPrior.train (ego/gaussianprocess/prior.py:76-156) on the data the golden script trained the
    s512_prior_ard prior with (tests/golden/make_golden.py: 100 latin-hypercube points of Shekel5, k=10,
    seed=504): same centres (the k-means and its random stream are deterministic) and weights"""
    from ibo_amd.gaussianprocess.prior import RBFNMeanPrior, GPMeanPrior
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.utils.latinhypercube import lhcSample
    from ibo_amd.utils.testfunctions import Shekel5
    g = load_golden("g3_cases")
    p = "s512_prior_ard/"
    S5 = Shekel5()
    pX = lhcSample(S5.bounds, 100, seed=511)
    pY = [S5.f(x) for x in pX]
    prior = RBFNMeanPrior()
    prior.train(pX, pY, bounds=S5.bounds, k=10, seed=504)
    close(np.array(prior.means), g[p + "pmeans"], rtol=1e-12, atol=1e-14)
    close(prior.beta, g[p + "pbeta"], rtol=1e-7, atol=1e-9)
    close(prior.lowerb, g[p + "plowerb"]); close(prior.width, g[p + "pwidth"])
    close([prior.mu(q) for q in g[p + "probe"]], g[p + "prior_mu"], rtol=1e-7, atol=1e-10)
    # a prior assigned (or trained) AFTER the fit is honoured by the next evaluation, and taken away again
    X, Y = g[p + "X"], g[p + "Y"]
    GP = GaussianProcess(GaussianKernel_ard(g[p + "hyper"]), X, Y, noise=float(g[p + "noise"]))
    plain = GP.posteriors(g[p + "probe"])[0]
    GP.prior = our_prior(g, p)
    close(np.array([GP.posterior(q) for q in g[p + "probe"]]), g[p + "post"], atol=1e-9)
    GP.prior.beta = np.asarray(GP.prior.beta) * 2.0                  # edited in place
    assert not np.allclose(GP.posteriors(g[p + "probe"])[0], g[p + "post"][:, 0])
    GP.prior = None
    np.testing.assert_array_equal(GP.posteriors(g[p + "probe"])[0], plain)
    # priors the device cannot represent are refused, never silently dropped
    class Flat(GPMeanPrior):
        def mu(self, x):
            return 1.0
    GP.prior = Flat()
    with pytest.raises(NotImplementedError):
        GP.posterior(g[p + "probe"][0])
    GP.prior = RBFNMeanPrior()                                        # never trained
    with pytest.raises(ValueError):
        GP.posterior(g[p + "probe"][0])


def test_synthetic_test_function_and_learn_hyper(ibo):
    from ibo_amd.utils.testfunctions import Synthetic, Hartman3, learnHyper
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_iso
    b = [[0., 1.]] * 2
    tf = Synthetic(GaussianKernel_iso([.3]), b, 30, seed=3, xstar=np.array([.25, .75]), maximize=False)
    assert len(tf.GP.X) == 31 and np.allclose(tf.xstar, [.25, .75])
    grid = np.random.RandomState(0).rand(2000, 2)
    v = tf.values(grid)
    assert abs(tf.f(grid[5]) - v[5]) < 1e-12
    assert tf.f(tf.xstar) <= v.min() + 0.5                          # the planted point is (close to) the global minimum
    th = learnHyper(Hartman3(), GaussianKernel_iso, seed=1)
    assert th.shape == (1,) and 0.05 < th[0] < 2.0


def test_c5_nlml_full_size(ibo, oracle):
    """N=4096, D=16 ARD marginal likelihood: the GPU grid against NumPy's LAPACK on the oracle's K"""
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid
    N, D = 4096, 16
    X, Y = synth(5, N, D)
    thetas = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(512, D)))[:3]
    vals, am = nlml_grid(GaussianKernel_ard, thetas, X, Y, noise=1e-3)
    ref = oracle.marginal_likelihood(oracle.Kern("ard", thetas[1]), X, Y, D, compute_gradient=False, noise=1e-3)
    close(vals[1], ref, rtol=1e-9)
    assert np.all(np.isfinite(vals)) and am == int(np.argmin(vals))
    assert np.array_equal(nlml_grid(GaussianKernel_ard, thetas[1:2], X, Y, noise=1e-3)[0], vals[1:2])   # deterministic


def test_two_level_order_agrees_with_the_single_level_order(ibo):
    """A single matrix is factored in the pipelined single-level order (W riding along) below 104 block columns and in the two-level order
    (panels of four block columns, K = 256 updates, recursive-doubling inversion) from there on; ibo_set_option("fused2_min_nb") moves
    the switch.  Both against NumPy's Cholesky of GP.R at 1e-11 and against each other at 1e-12 (another order of the same sums), W L = I,
    ragged sizes (a last panel of one, two, three columns), a matrix that is not positive definite in both orders; and one size in the
    default two-level range (6720 rows: in-panel columns with more tiles than CUs take the rows-then-updates launches)."""
    from ibo_amd import _lib, NotPositiveDefinite
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard

    def fit(X, Y, N, noise=.05):
        GP = GaussianProcess(GaussianKernel_ard([.45] * X.shape[1]), X, Y, noise=noise)
        W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(GP._handle(), _lib.dp(W)))
        return GP, GP.L.copy(), W
    for N in (2113, 2200, 2250, 3000):                        # 34, 35, 36, 47 blocks
        X, Y = synth(N + 1, N, 5)
        _lib.check(_lib.lib.ibo_set_option(b"fused2_min_nb", 33))
        try:
            GP2, L2, W2 = fit(X, Y, N)
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"fused2_min_nb", 104))
        GP1, L1, W1 = fit(X, Y, N)                            # the default order at these sizes
        R = np.array(GP1.R)
        assert np.array_equal(R, R.T) and np.all(np.diag(R) == 1.05)
        Lr = np.linalg.cholesky(R)
        for L, W in ((L2, W2), (L1, W1)):
            assert np.abs(L - Lr).max() < 1e-11 and np.abs(np.triu(L, 1)).max() == 0.0
            assert np.abs(W.dot(L) - np.eye(N)).max() < 1e-10 and np.all(np.triu(W, 1) == 0.0)
        assert np.abs(L1 - L2).max() < 1e-12 and np.abs(W1 - W2).max() < 1e-9 * max(1.0, np.abs(W1).max())
    Xd = np.vstack([X[:2199], X[77:78]])                      # a duplicate point and no noise
    for min_nb in (33, 104):                                   # in both orders
        _lib.check(_lib.lib.ibo_set_option(b"fused2_min_nb", min_nb))
        try:
            with pytest.raises(NotPositiveDefinite):
                GaussianProcess(GaussianKernel_ard([.45] * 5), Xd, Y[:2200], noise=0.0)
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"fused2_min_nb", 104))
    N = 6720                                                   # 105 block columns: two-level by default
    X, Y = synth(N + 1, N, 6)
    GP, L, W = fit(X, Y, N, noise=.1)
    Lr = np.linalg.cholesky(np.array(GP.R))
    assert np.abs(L - Lr).max() < 1e-11 and np.abs(np.triu(L, 1)).max() == 0.0
    probe = np.random.RandomState(5).randint(0, N, 300)
    assert np.abs(W[probe].dot(L) - np.eye(N)[probe]).max() < 1e-10 and np.all(np.triu(W, 1) == 0.0)


def test_super_panel_order_has_the_bits_of_the_step_by_step_order(ibo):
    """From 64 block columns (4096 rows) a single-level factorisation runs in super-panels of 16 block columns (csrc/linalg.hip
    launch_cholesky_super: the pipelined launches keep their tiles inside a super-panel, the columns beyond take its sixteen steps as one deep
    update from packed operands, E riding along as the lower half of one tall matrix; linalg.cholesky of ego/gaussianprocess/__init__.py:299,
    the inverse of ego/acquisition/__init__.py:385-388).  ibo_set_option("super_min_nb") moves the switch: L and W of both orders are
    BIT-IDENTICAL -- two, three and four super-panels, a last one of 2, 15 and 16 block columns -- and agree with NumPy; the NLML gradient, which
    takes the same route, returns the same bits too; a matrix that is not positive definite is reported from the first and from a later
    super-panel; one size in the default range (4100 rows)."""
    from ibo_amd import _lib, NotPositiveDefinite
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood

    def fit(X, Y, N, nb):
        _lib.check(_lib.lib.ibo_set_option(b"super_min_nb", nb))
        try:
            GP = GaussianProcess(GaussianKernel_ard([.45] * X.shape[1]), X, Y, noise=.05)
            W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(GP._handle(), _lib.dp(W)))
            v, g = marginalLikelihood(GaussianKernel_ard([.45] * X.shape[1]), X, Y, X.shape[1], True, noise=.05)
            return np.array(GP.L), W, np.array(GP.R), v, np.asarray(g)
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"super_min_nb", 64))
    for N in (2113, 3000, 3072, 3905):                        # 34, 47, 48, 62 block columns
        X, Y = synth(N + 3, N, 5)
        L1, W1, R, v1, g1 = fit(X, Y, N, 1000)                # step by step
        L2, W2, _, v2, g2 = fit(X, Y, N, 32)                  # in super-panels
        assert np.array_equal(L1, L2) and np.array_equal(W1, W2), N
        assert v1 == v2 and np.array_equal(g1, g2), N
        Lr = np.linalg.cholesky(R)
        assert np.abs(L2 - Lr).max() < 1e-11 and np.abs(np.triu(L2, 1)).max() == 0.0
        probe = np.random.RandomState(N).randint(0, N, 200)
        assert np.abs(W2[probe].dot(L2) - np.eye(N)[probe]).max() < 1e-10 and np.all(np.triu(W2, 1) == 0.0)
    X, Y = synth(77, 2300, 4)
    for dup in (40, 1500):                                     # the first and the second super-panel
        Xd = X.copy(); Xd[dup + 5] = Xd[dup]
        _lib.check(_lib.lib.ibo_set_option(b"super_min_nb", 32))
        try:
            with pytest.raises(NotPositiveDefinite):
                GaussianProcess(GaussianKernel_ard([.4] * 4), Xd, Y, noise=0.0)
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"super_min_nb", 64))
    N = 4100                                                   # 65 block columns: super-panels by default
    X, Y = synth(78, N, 6)
    GP = GaussianProcess(GaussianKernel_ard([.5] * 6), X, Y, noise=.1)
    L = np.array(GP.L)
    assert np.abs(L - np.linalg.cholesky(np.array(GP.R))).max() < 1e-11


def test_nlml_gradient_against_the_oracle_every_family(ibo):
    """dnlml (csrc/assemble.hip nlml_grad_fast_kernel: the pairs' coordinates shared over a thread's 4 x 4 pairs, one short loop per
    derivative, lower tiles counted twice) against the oracle's value and gradient (ego/gaussianprocess/trainhyper.py:47-95): SE-ARD in
    2, 5 and 20 dimensions, SE-iso with signal variance, Matern-3/2, Matern-5/2, plain SE-iso"""
    import oracle.oracle as orc
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
    for N, theta in ((130, [.3, .6]), (700, [.3, .4, .5, .6, .7]), (500, list(np.linspace(.8, 1.6, 20)))):
        D = len(theta)
        X, Y = synth(N + D, N, D)
        v, g = marginalLikelihood(K.GaussianKernel_ard(theta), X, Y, D, True, noise=1e-2)
        ov, od = orc.marginal_likelihood(orc.Kern("ard", theta), X, Y, D, True, 1e-2)
        assert abs(v - ov) <= 1e-9 * abs(ov) and np.abs(np.asarray(g) - np.asarray(od)).max() <= 1e-8 * np.abs(od).max()
    X, Y = synth(405, 400, 4)
    for k, ok, nh in ((K.SVGaussianKernel_iso([.7, 1.3]), orc.Kern("sviso", [.7, 1.3]), 2), (K.MaternKernel3([.8, 1.1]), orc.Kern("m3", [.8, 1.1]), 2),
                      (K.MaternKernel5([.9, 1.2]), orc.Kern("m5", [.9, 1.2]), 2), (K.GaussianKernel_iso([.6]), orc.Kern("iso", [.6]), 1)):
        v, g = marginalLikelihood(k, X, Y, nh, True, noise=1e-2)
        ov, od = orc.marginal_likelihood(ok, X, Y, nh, True, 1e-2)
        assert abs(v - ov) <= 1e-9 * abs(ov) and np.abs(np.atleast_1d(g) - np.atleast_1d(od)).max() <= 1e-7 * np.abs(od).max(), (type(k).__name__, g, od)


def test_nlml_gradient_keeps_its_data_on_the_device_only_while_it_is_the_same_data(ibo):
    """ibo_nlml_grad (abi_nlml.hip) leaves X and Y on the device between calls and uploads them again when their CONTENT differs (a learning
    loop calls it dozens of times with one data set and another theta): the same data twice, then other targets, other points and another
    size in the same shapes' buffers, then the first data again and after ibo_trim -- every value and gradient against the oracle"""
    import oracle.oracle as orc
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
    D = 3
    XA, YA = synth(911, 300, D)
    XB, YB = synth(912, 300, D)
    XC, YC = synth(913, 290, D)
    def check(X, Y, theta):
        v, g = marginalLikelihood(K.GaussianKernel_ard(theta), X, Y, D, True, noise=1e-2)
        ov, od = orc.marginal_likelihood(orc.Kern("ard", theta), X, Y, D, True, 1e-2)
        assert abs(v - ov) <= 1e-9 * abs(ov) and np.abs(np.asarray(g) - np.asarray(od)).max() <= 1e-8 * np.abs(od).max()
        return v
    v1 = check(XA, YA, [.4, .5, .6])
    assert check(XA.copy(), YA.copy(), [.4, .5, .6]) == v1      # the same content at another address: nothing uploaded, the same bits
    check(XA, YA, [.5, .5, .7])
    check(XA, YB, [.4, .5, .6])                                  # other targets
    check(XB, YB, [.4, .5, .6])                                  # other points
    Y2 = YB.copy(); Y2[-1] += 1e-9
    check(XB, Y2, [.4, .5, .6])                                  # one target's last bits
    check(XC, YC, [.4, .5, .6])                                  # fewer rows (the buffers stay)
    check(XA[:290], YA[:290], [.4, .5, .6])                      # the same size as the call before, other content
    assert check(XA, YA, [.4, .5, .6]) == v1
    _lib.check(_lib.lib.ibo_trim(0))
    assert check(XA, YA, [.4, .5, .6]) == v1


def test_single_level_fit_at_every_launch_shape(ibo):
    """fits below 104 block columns: fused steps (up to three block columns), pipelined block columns with one step per pass over the trailing
    tiles (4 .. 11 block columns) and with two (from 12), W = L^-1 riding along: L against NumPy's Cholesky of GP.R, W L = I; the NLML
    gradient, which takes the same route (K^-1 = W^T W from the ride-along), against the oracle; a failed pivot in an early and a late
    block column is reported"""
    from ibo_amd import _lib, NotPositiveDefinite
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    for N in (60, 64, 129, 192, 193, 257, 700, 704, 768, 1024, 1500, 1985, 2048):
        X, Y = synth(N + 2, N, 4)
        GP = GaussianProcess(GaussianKernel_ard([.4] * 4), X, Y, noise=.05)
        W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(GP._handle(), _lib.dp(W)))
        L = GP.L
        assert np.abs(L - np.linalg.cholesky(np.array(GP.R))).max() < 1e-11 and np.abs(np.triu(L, 1)).max() == 0.0, N
        assert np.abs(W @ L - np.eye(N)).max() < 1e-10 and np.all(np.triu(W, 1) == 0.0), N
    import oracle.oracle as orc
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
    X, Y = synth(1603, 1600, 3)
    v, g = marginalLikelihood(GaussianKernel_ard([.3, .4, .5]), X, Y, 3, True, noise=1e-2)
    ov, od = orc.marginal_likelihood(orc.Kern("ard", [.3, .4, .5]), X, Y, 3, True, 1e-2)
    close(v, ov); close(g, od, atol=1e-8)
    for dup in (40, 1590):
        Xd = X.copy(); Xd[dup + 5] = Xd[dup]
        with pytest.raises(NotPositiveDefinite):
            GaussianProcess(GaussianKernel_ard([.3, .4, .5]), Xd, Y, noise=0.0)


def test_nlml_gradient_from_1792_rows_on_the_packed_operand_product(ibo):
    """from 1792 rows ibo_nlml_grad forms K^-1 = W^T W on chol_update3_kernel (csrc/update3.hip launch_syrk3: W^T packed into fragment order,
    128 x 128 tiles, long K ranges in pieces -- of 256 columns below 2560 rows, 512 below 4096, 1024 from there -- that are summed in a fixed
    order): value and gradient against the oracle (ego/gaussianprocess/trainhyper.py:47-95) at the first size of the route (1729 -> 1792), in
    the 256-piece range where the last tile row is 64 rows (1990 -> 2048 is whole; 2100 -> 2112 is not), in the 512-piece range where it is
    64 rows (2600 -> 2624) and where it is whole (2688), and just below the switch (1728, wtw_kernel); an evaluation repeated bit for bit"""
    import oracle.oracle as orc
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
    theta = [.5, .7, .9, 1.1]
    for N in (1728, 1729, 1990, 2100, 2600, 2688):
        X, Y = synth(N + 7, N, 4)
        v, g = marginalLikelihood(GaussianKernel_ard(theta), X, Y, 4, True, noise=1e-2)
        ov, od = orc.marginal_likelihood(orc.Kern("ard", theta), X, Y, 4, True, 1e-2)
        assert abs(v - ov) <= 1e-9 * abs(ov) and np.abs(np.asarray(g) - np.asarray(od)).max() <= 1e-8 * np.abs(od).max(), (N, g, od)
        v2, g2 = marginalLikelihood(GaussianKernel_ard(theta), X, Y, 4, True, noise=1e-2)
        assert v2 == v and np.array_equal(np.asarray(g2), np.asarray(g))


def test_nlml_grid_does_not_depend_on_what_shares_its_launches(ibo):
    """the batched NLML grid: the same values whatever the batch size, alone or together; a theta whose matrix is not positive definite
    leaves nothing behind"""
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid
    X, Y = synth(9, 1000, 5)
    thetas = np.exp(np.random.RandomState(3).uniform(np.log(.2), np.log(2), size=(7, 5)))
    vals, _ = nlml_grid(GaussianKernel_ard, thetas, X[:300], Y[:300], noise=.01)
    for b in (1, 2, 7):
        _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", b))
        try:
            assert np.array_equal(nlml_grid(GaussianKernel_ard, thetas, X[:300], Y[:300], noise=.01)[0], vals)
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", 0))
    assert np.array_equal(nlml_grid(GaussianKernel_ard, thetas[4:5], X[:300], Y[:300], noise=.01)[0], vals[4:5])
    # a theta whose matrix is not positive definite (length scales so long that K is numerically singular) yields 100
    # and leaves nothing behind in the matrix slot the next theta is factored in (nlml_batch = 1: the same slot),
    # nor in the workspace a later call finds
    bad = np.r_[[np.full(5, 3e3)], thetas[:2]]
    _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", 1))
    try:
        v3, _ = nlml_grid(GaussianKernel_ard, bad, X[:300], Y[:300], noise=1e-14)
        ref, _ = nlml_grid(GaussianKernel_ard, thetas[:2], X[:300], Y[:300], noise=1e-14)
    finally:
        _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", 0))
    assert not np.isfinite(v3[0]) or v3[0] == 100.0
    assert np.all(np.isfinite(ref)) and np.array_equal(v3[1:], ref)
    # the rows below a panel in one launch (chol_panel_rows8r_kernel, taken when rows x batch fill the chip: 24 matrices) or as the
    # per-column trsm / update launches it replaces (one matrix at a time): the same bits
    Xb, Yb = synth(12, 1100, 4)
    tb = np.exp(np.random.RandomState(4).uniform(np.log(.2), np.log(2), size=(24, 4)))
    va, _ = nlml_grid(GaussianKernel_ard, tb, Xb, Yb, noise=.01)
    _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", 1))
    try:
        vb, _ = nlml_grid(GaussianKernel_ard, tb, Xb, Yb, noise=.01)
    finally:
        _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", 0))
    assert np.all(np.isfinite(va)) and np.array_equal(va, vb)


def test_grid_covariance_on_the_mfma_unit_against_the_oracle_and_the_difference_form(ibo, oracle):
    """ibo_nlml_grid's covariance pass forms -z/2 = a_i + a_j + x~_i . x~_j as a product on the MFMA unit (csrc/assemble.hip
    cov_grid_mfma_kernel; ego/gaussianprocess/kernel.py:46-53,147-149, trainhyper.py:55) where every scaled point stays inside the dot
    form's guard, by coordinate differences elsewhere (ibo_set_option("dot_form", 0) forces those).  NLML values of both routes against
    the oracle at 1e-9 and against each other at 1e-10: ragged sizes (a last tile of one row; rows that are not a multiple of 16, 64 or
    128), 1 .. 32 dimensions (one to nine k4-steps of the exponent product), every covariance family; a theta-point with length scales so
    short that the guard trips sends the whole call to the difference form (same values as forced)."""
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid, nlml_values

    def both(kernels, X, Y, noise):
        v = nlml_values(kernels, X, Y, noise)
        _lib.check(_lib.lib.ibo_set_option(b"dot_form", 0))
        try:
            d = nlml_values(kernels, X, Y, noise)
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"dot_form", -1))
        return np.asarray(v), np.asarray(d)
    for N, D in ((65, 1), (193, 2), (500, 3), (1000, 6), (1345, 16), (777, 30), (300, 32)):
        X, Y = synth(40 + N, N, D)
        th = np.exp(np.random.RandomState(N).uniform(np.log(.3), np.log(2), size=(5, D))) * np.sqrt(D)
        v, d = both([K.GaussianKernel_ard(t) for t in th], X, Y, 1e-2)
        assert np.all(np.isfinite(v)) and np.abs(v - d).max() <= 1e-10 * np.abs(d).max(), (N, D, v, d)
        ref = oracle.marginal_likelihood(oracle.Kern("ard", th[2]), X, Y, D, compute_gradient=False, noise=1e-2)
        close(v[2], ref, rtol=1e-9); close(d[2], ref, rtol=1e-9)
    X, Y = synth(41, 600, 4)
    for kern, ok in ((K.MaternKernel3([.8, 1.1]), oracle.Kern("m3", [.8, 1.1])), (K.MaternKernel5([.9, 1.2]), oracle.Kern("m5", [.9, 1.2])),
                     (K.SVGaussianKernel_iso([.7, 1.3]), oracle.Kern("sviso", [.7, 1.3])), (K.GaussianKernel_iso([.6]), oracle.Kern("iso", [.6]))):
        v, d = both([kern], X, Y, 1e-2)
        ref = oracle.marginal_likelihood(ok, X, Y, len(kern.hyperparams), compute_gradient=False, noise=1e-2)
        close(v[0], ref, rtol=1e-9); close(d[0], ref, rtol=1e-9)
        assert abs(v[0] - d[0]) <= 1e-10 * abs(d[0]), (type(kern).__name__, v, d)
    # |x~|^2 beyond 1e5 for one theta-point of the call (length scale 1e-3 on coordinates up to 1): the guard keeps the difference form
    th = np.array([[.5, .6, .7, .8], [1e-3, .6, .7, .8]])
    v, d = both([K.GaussianKernel_ard(t) for t in th], X, Y, 1e-2)
    assert np.array_equal(v, d)


def test_left_looking_grid_equals_the_right_looking_one(ibo, oracle):
    """ibo_nlml_grid factors in the left-looking outer order (update3.hip: a panel's columns take every finished column's
    update in one long-K launch, from one packed copy of the factor, pad rows untouched, the y row's lonely last block column
    left out); same sums in the same order as the right-looking two-level order: identical values -- for sizes on, just
    below and just above multiples of 64 / 128 / 256, batches that are not multiples of 8, a theta that is not positive
    definite in the middle of a batch, batches large enough for the fused panel launches -- and the oracle's"""
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid
    rs = np.random.RandomState(31)
    for N, D, nth, noise in ((256, 3, 5, .01), (255, 3, 9, .01), (257, 2, 3, .01), (320, 4, 8, .01), (511, 5, 16, .01), (512, 5, 17, .01),
                             (700, 3, 4, .01), (1024, 4, 11, 1e-3), (1100, 6, 3, 1e-3), (2048, 8, 8, 1e-3),
                             # enough row blocks x matrices for the ONE-launch panels (chol_panel_fused_kernel: diagonal workgroups and row
                             # workgroups handing over through flags): an even and an odd number of row blocks below a panel
                             (1500, 5, 32, .01), (1599, 4, 32, .01), (2300, 3, 24, .01)):
        X, Y = synth(N + D, N, D)
        th = np.exp(rs.uniform(np.log(.2), np.log(2), size=(nth, D)))
        bad = nth >= 5 and N <= 1100                     # (beyond, the tiny noise below leaves no theta of the draw positive definite)
        if bad:
            th[2] = 3e3                                  # numerically singular with the tiny noise below: NaN in that slot
        nz = 1e-14 if bad else noise
        out = []
        for left in (1, 0, 1):
            _lib.check(_lib.lib.ibo_set_option(b"chol_left", left))
            try:
                out.append(nlml_grid(GaussianKernel_ard, th, X, Y, noise=nz)[0])
            finally:
                _lib.check(_lib.lib.ibo_set_option(b"chol_left", 1))
        assert np.array_equal(out[0], out[1], equal_nan=True) and np.array_equal(out[0], out[2], equal_nan=True), (N, D, nth)
        if bad:
            assert np.isnan(out[0][2]) and np.isfinite(out[0]).sum() >= nth // 2, out[0]
        else:
            assert np.all(np.isfinite(out[0])), (N, D, nth)
        if N <= 512 and nth < 5:
            for t in range(nth):
                close(out[0][t], oracle.nlml_c(oracle.Kern("ard", th[t]), X, Y, noise=nz), rtol=1e-9)
    # one matrix at a time and everything at once: the batch never changes a value
    X, Y = synth(77, 640, 4)
    th = np.exp(rs.uniform(np.log(.3), np.log(2), size=(10, 4)))
    ref = nlml_grid(GaussianKernel_ard, th, X, Y, noise=.01)[0]
    for b in (1, 3, 8):
        _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", b))
        try:
            assert np.array_equal(nlml_grid(GaussianKernel_ard, th, X, Y, noise=.01)[0], ref)
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", 0))


def test_fused_panel_launches_with_a_failed_matrix_in_the_batch(ibo):
    """chol_panel_fused_kernel's row workgroups wait on flags the diagonal workgroups raise: a matrix whose chain breaks down (NaN pivots) raises
    them all the same -- its slot is NaN, nobody waits for ever, and the other 31 values are those of the right-looking order and of a batch
    without the failed matrix"""
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid
    X, Y = synth(1505, 1500, 5)
    th = np.exp(np.random.RandomState(77).uniform(np.log(.05), np.log(.12), size=(32, 5)))      # nearly diagonal matrices: fine with next to no noise
    good = nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-14)[0]
    assert np.all(np.isfinite(good))
    th2 = th.copy(); th2[5] = 3e3                       # numerically singular
    out = []
    for left in (1, 0):
        _lib.check(_lib.lib.ibo_set_option(b"chol_left", left))
        try:
            out.append(nlml_grid(GaussianKernel_ard, th2, X, Y, noise=1e-14)[0])
        finally:
            _lib.check(_lib.lib.ibo_set_option(b"chol_left", 1))
    assert np.isnan(out[0][5]) and np.array_equal(out[0], out[1], equal_nan=True)
    keep = np.arange(32) != 5
    assert np.array_equal(np.asarray(out[0])[keep], np.asarray(good)[keep])


def test_randomised_parity_sweep(ibo, oracle):
    """a seeded slice of tools/fuzz_gpu.py: random (N, D, kernel family, noise, M) -- fit, posterior mean/variance and
    libego-flavoured EI of every candidate against the oracle, all within the 1e-6 bar; arg-max = first maximiser"""
    from ibo_amd.gaussianprocess import GaussianProcess, kernel as K
    from ibo_amd.acquisition import sweep
    rs = np.random.RandomState(20261003)
    worst = 0.0
    for case in range(24):
        N = int([1, 63, 64, 65, 129, 511][case] if case < 6 else rs.randint(2, 700))
        D = int(rs.randint(1, 17))
        kind = ["ard", "iso", "m3", "m5"][rs.randint(4)]
        M = int([1, 17, 64, 65, 1000, 8193][rs.randint(6)])
        noise = float([.1, .01, 1e-3][rs.randint(3)])
        X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
        th = np.exp(rs.uniform(np.log(.2), np.log(1.5), size=D))
        hyp = {"ard": th, "iso": th[:1], "m3": np.r_[th[0], 1.0], "m5": np.r_[th[0], 1.0]}[kind]
        cls = {"ard": K.GaussianKernel_ard, "iso": K.GaussianKernel_iso, "m3": K.MaternKernel3, "m5": K.MaternKernel5}[kind]
        ogp = oracle.GP(oracle.Kern(kind, hyp), X, Y, noise=noise)
        GP = GaussianProcess(cls(hyp), X, Y, noise=noise)
        cand = rs.rand(M, D)
        Mo = min(M, 200)
        r = sweep(GP, cand, acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
        o = oracle.sweep_native(ogp, cand[:Mo], oracle.ACQ_EI, .01)
        for key in ("mu", "s2", "acq"):
            err = np.abs(r[key][:Mo] - o[key]) / np.maximum(np.abs(o[key]), 1e-9)
            if key == "acq":
                err = err * (np.abs(o[key]) > ACQ_ATOL)
            worst = max(worst, float(err.max()))
        assert r["best_idx"] == int(np.argmax(r["acq"])), (N, D, kind, M)
    assert worst < RT, worst


def test_bayesian_optimisation_loop_end_to_end(ibo):
    """the loop the reference exists for (demo.py:59-98, repeated): fit -> maximizeEI -> evaluate -> addData.
    25 rounds on a smooth 2-D objective with a known maximum must get within 1e-2 of it, and every model refit /
    DIRECT call along the way goes through the GPU path"""
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import maximizeEI
    from ibo_amd.utils.latinhypercube import lhcSample

    def f(x):                                   # maximum 1.0 at (0.7, 0.3)
        return float(np.exp(-8 * ((x[0] - .7) ** 2 + (x[1] - .3) ** 2)) + .3 * np.exp(-20 * ((x[0] - .2) ** 2 + (x[1] - .8) ** 2)))
    bounds = [[0., 1.], [0., 1.]]
    X = lhcSample(bounds, 5, seed=3)
    GP = GaussianProcess(GaussianKernel_ard([.25, .25]), noise=1e-4)
    for x in X:
        GP.addData(x, f(x))
    for _ in range(25):
        opt, optx = maximizeEI(GP, bounds, xi=.01)
        GP.addData(optx, f(optx))
    best = int(np.argmax(GP.Y))
    assert GP.Y[best] > 1.0 - 1e-2 and np.linalg.norm(GP.X[best] - np.array([.7, .3])) < .05
    assert len(GP.Y) == 30


def test_recycled_device_buffers_do_not_leak_state(ibo, oracle):
    """buffers of destroyed models are handed to the next one (possibly larger than it asked for, full of the old
    model's data): results must only depend on the new model; ibo_amd.trim() empties the free list"""
    import gc
    import ibo_amd
    from ibo_amd.gaussianprocess import GaussianProcess, kernel as K
    probe = np.random.RandomState(77).rand(40, 3)
    for rnd in range(2):
        for N in (500, 300, 64, 700, 1, 300):
            X, Y = synth(60 + N, N, 3)
            GP = GaussianProcess(K.GaussianKernel_ard([.3, .4, .5]), X, Y, noise=.1)
            mu, s2 = GP.posteriors(probe)
            o_mu, o_s2 = oracle.GP(oracle.Kern("ard", [.3, .4, .5]), X, Y, noise=.1).posteriors(probe)
            close(mu, o_mu, atol=1e-9); close(s2, o_s2)
            del GP
            gc.collect()
        ibo_amd.trim()


def test_large_host_batches_are_pipelined_without_changing_results(ibo):
    """GP.posteriors on a big NumPy array goes through overlapped 2^17-point chunks (upload / sweep / download on
    three streams): same numbers as the single-shot path, ragged last chunk included"""
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    X, Y = synth(21, 300, 3)
    GP = GaussianProcess(GaussianKernel_ard([.3, .4, .5]), X, Y, noise=.05)
    Q = np.random.RandomState(22).rand((1 << 18) + (1 << 17) + 12345, 3)
    mu1, s21 = GP._posterior_arrays(Q, True)
    _lib.check(_lib.lib.ibo_set_option(b"host_pipeline", 0))
    try:
        mu0, s20 = GP._posterior_arrays(Q, True)
    finally:
        _lib.check(_lib.lib.ibo_set_option(b"host_pipeline", 1))
    assert np.array_equal(mu0, mu1) and np.array_equal(s20, s21)
    mu2 = GP._posterior_arrays(Q[: (1 << 18)], False)          # exactly two full chunks, mean only
    mu2 = mu2[0] if isinstance(mu2, tuple) else mu2
    assert np.array_equal(np.asarray(mu2).ravel(), mu0[: (1 << 18)])


def test_add_observation_point_augmented_variance(ibo, oracle):
    """PrefGaussianProcess.addObservationPoint (ego/gaussianprocess/__init__.py:214-223,502-519): the mean keeps
    using L = chol(R + C^-1), the variance switches to the factor of the augmented matrix"""
    from ibo_amd.gaussianprocess import PrefGaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    g = load_golden("g7_prefs")
    p = "p8/"
    prefs = [(v, u, d) for v, u, d in zip(g[p + "pref_v"], g[p + "pref_u"], g[p + "pref_d"])]
    GP = PrefGaussianProcess(GaussianKernel_ard(g[p + "hyper"]), prefs)
    probe = g[p + "probe"][:10]
    mu0, s20 = GP.posteriors(probe)
    newx = np.array([[.3, .4, .5, .6, .7, .8], [.9, .1, .2, .8, .3, .5]])
    GP.addObservationPoint(newx[0]); GP.addObservationPoint(newx[1])
    mu1, s21 = GP.posteriors(probe)
    # reference computation in NumPy from the model's own public matrices
    ok = oracle.Kern("ard", g[p + "hyper"])
    augX = np.vstack([GP.X, newx])
    n, m = len(augX), len(GP.X)
    augR = np.array([[ok.cov(a, b) for b in augX] for a in augX]); np.fill_diagonal(augR, 1 + GP.noise)
    invC = np.zeros((n, n)); invC[:m, :m] = np.linalg.inv(GP.C)
    augL = np.linalg.cholesky(augR + invC)
    r = np.array([[ok.cov(a, q) for q in probe] for a in augX])
    s2_ref = np.clip((1 + GP.noise) - np.sum(np.linalg.solve(augL, r) ** 2, axis=0), 10e-8, 10)
    close(mu1, mu0, rtol=1e-12, atol=1e-14)
    close(s21, s2_ref); close(GP.augL, augL, atol=1e-9); close(GP.augR, augR, rtol=1e-12)
    assert np.all(s21 <= s20 + 1e-12)                    # observing more can only shrink the variance
    assert len(GP.augX) == m + 2
    # the acquisition classes see the augmented variance, as the reference's EI.negf -> GP.posterior does
    from ibo_amd.acquisition import EI
    e = EI(GP, xi=.05)
    ref_ei = oracle.acq_value(oracle.ACQ_EI, oracle.ERF_NR, mu1[:3], np.sqrt(s2_ref[:3]), GP.Y.max(), .05)
    close([e.f(q) for q in probe[:3]], ref_ei, atol=ACQ_ATOL)


def test_sv_ard_kernel_and_ego_alias(ibo, oracle):
    """signal-variance ARD kernel through the device path, imported under the reference's package name"""
    import ibo_amd
    ibo_amd.install_as_ego()
    from ego.gaussianprocess import GaussianProcess
    from ego.gaussianprocess.kernel import SVGaussianKernel_ard
    from ego.acquisition import maximizeEI, EI
    from ego.utils.latinhypercube import lhcSample
    hyp = [.4, .5, .6, 1.03]
    b = [[0., 1.]] * 3
    X = np.array(lhcSample(b, 30, seed=3)); Y = np.sin(4 * X.sum(1))
    GP = GaussianProcess(SVGaussianKernel_ard(hyp), X, Y, noise=.1)
    ogp = oracle.GP(oracle.Kern("svard", hyp), X, Y, noise=.1)
    close(GP.R, ogp.R, rtol=1e-12)
    probe = np.array(lhcSample(b, 20, seed=4))
    mu, s2 = GP.posteriors(probe); o_mu, o_s2 = ogp.posteriors(probe)
    close(mu, o_mu, atol=1e-10); close(s2, o_s2)
    # maximizeEI follows libego: k* WITHOUT the magnitude (cpp/optimizeGP.cpp:303-310) on R built with it
    opt, optx = maximizeEI(GP, b, maxiter=10)
    o, ox, _ = oracle.acqmax_native(ogp, b, oracle.ACQ_EI, .01, maxiter=10)
    close(opt, o, atol=ACQ_ATOL); close(optx, ox, rtol=1e-9, atol=1e-12)
    close(EI(GP).f(probe[0]), oracle.acq_value(oracle.ACQ_EI, oracle.ERF_NR, o_mu[0], np.sqrt(o_s2[0]), Y.max(), .01)[0],
          atol=ACQ_ATOL)


def test_seventeen_to_thirty_two_dimensions(ibo, oracle):
    """D = 17..32 (rows of X padded to 32 coordinates, six to nine k4-steps in the exponent GEMM, the 3072-row alpha window,
    up to 33 gradient components): every sweep kernel, the fit, the block extension + refresh kernel, NLML with its
    gradient and DIRECT against the CPU oracle"""
    import oracle.oracle as orc
    from ibo_amd import _lib, DeviceArray
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
    from ibo_amd.acquisition import sweep, maximizeEI
    opt = lambda k, v: _lib.check(_lib.lib.ibo_set_option(k, v))
    cases = [(900, 17, K.GaussianKernel_ard([1.0 + .02 * d for d in range(17)]), ("ard", [1.0 + .02 * d for d in range(17)])),
             (1100, 20, K.MaternKernel5([1.4, 1.0]), ("m5", [1.4, 1.0])),
             (700, 23, K.GaussianKernel_iso([1.2]), ("iso", [1.2])),
             (1300, 27, K.MaternKernel3([1.6, 1.0]), ("m3", [1.6, 1.0])),
             (520, 32, K.GaussianKernel_ard([1.5] * 32), ("ard", [1.5] * 32)),
             (3300, 19, K.GaussianKernel_ard([1.1] * 19), ("ard", [1.1] * 19))]       # more rows than the 3072-row alpha window
    try:
        for N, D, kern, (okind, ohyp) in cases:
            X, Y = synth(N + D, N, D)
            GP = GaussianProcess(kern, X, Y, noise=.1)
            ogp = oracle.GP(oracle.Kern(okind, ohyp), X, Y, noise=.1)
            if N <= 1300:
                close(GP.R, ogp.R, rtol=1e-12); close(GP.L, ogp.L, rtol=1e-8, atol=1e-12)
            M = 8300
            cand = np.random.RandomState(N).rand(M, D); cand[77] = X[5]; cand[M - 1] = cand[0]
            r = sweep(GP, cand, acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            assert r["kernel"] == "sweep2_kernel"
            idx = np.r_[np.arange(0, M, M // 40), 77, M - 1, r["best_idx"]]
            o = oracle.sweep_native(ogp, cand[idx], oracle.ACQ_EI, .01)
            close(r["mu"][idx], o["mu"], atol=1e-9); close(r["s2"][idx], o["s2"]); close(r["acq"][idx], o["acq"], atol=ACQ_ATOL)
            assert r["best_idx"] == int(np.argmax(r["acq"]))
            # the first-generation tile kernel (difference form), the GEMV kernel, the small-batch kernels and the split form
            opt(b"dot_form", 0)
            r1 = sweep(GP, cand, acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            opt(b"dot_form", -1)
            assert r1["kernel"] == "sweep_mfma_kernel" and r1["best_idx"] == r["best_idx"]
            close(r1["mu"], r["mu"], rtol=1e-9, atol=1e-10); close(r1["s2"], r["s2"], rtol=1e-9)
            opt(b"sweep_path", 1); rg = sweep(GP, cand[:300], acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq")); opt(b"sweep_path", 0)
            assert rg["kernel"] == "sweep_gemv_kernel"
            rs = sweep(GP, cand[:300], acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            assert rs["kernel"] == "wk_small_kernel"
            opt(b"sweep_path", 3); rp = sweep(GP, cand[:300], acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq")); opt(b"sweep_path", 0)
            assert rp["kernel"] == "sweep_mfma_kernel<split>"
            for k in ("mu", "s2", "acq"):
                for other in (rg, rs, rp):
                    close(other[k], r[k][:300], rtol=1e-9, atol=1e-11)
            for form in (0, 1):                                   # the split kernel in both distance forms
                opt(b"dot_form", form); opt(b"sweep_path", 3)
                rf = sweep(GP, cand[:300], acq='ei', xi=.01, native=True, outputs=("mu", "s2"))
                opt(b"dot_form", -1); opt(b"sweep_path", 0)
                close(rf["mu"], r["mu"][:300], rtol=1e-9, atol=1e-10); close(rf["s2"], r["s2"][:300], rtol=1e-9)
            # one point (posterior(x): inline candidates), Python-flavoured acquisition
            mu1, s21 = GP.posterior(cand[3]); o1 = ogp.posteriors(cand[3:4])
            close(mu1, o1[0][0], atol=1e-9); close(s21, o1[1][0])
        # block extension + refresh kernel in 24 dimensions
        N0, D = 600, 24
        X, Y = synth(124, N0 + 4, D)
        kern = K.GaussianKernel_ard([1.3] * D)
        GP = GaussianProcess(kern, X[:N0], Y[:N0], noise=.1)
        dc = DeviceArray.from_host(np.random.RandomState(125).rand(9001, D))
        seen = []
        for rnd in range(4):
            r = sweep(GP, dc, acq='ei', xi=.4, native=False, incremental=True, outputs=("mu", "s2", "acq"))
            seen.append(r["kernel"])
            f = sweep(GaussianProcess(kern, GP.X, GP.Y, noise=.1), dc, acq='ei', xi=.4, native=False, outputs=("mu", "s2", "acq"))
            close(r["mu"], f["mu"], rtol=1e-9, atol=1e-10); close(r["s2"], f["s2"], rtol=1e-9); assert r["best_idx"] == f["best_idx"]
            GP.addData(X[len(GP.X)], Y[len(GP.X)])
        assert seen == ["sweep2_kernel"] + ["sweep2_rank1_kernel"] * 3
        ogp = oracle.GP(oracle.Kern("ard", [1.3] * D), np.asarray(GP.X), np.asarray(GP.Y), noise=.1)
        probe = np.random.RandomState(126).rand(30, D)
        close(GP.posteriors(probe), ogp.posteriors(probe), atol=1e-9)
        # NLML and its 21 / 33 partial derivatives
        for N, D, name, cls, hyp in ((300, 20, "svard", K.SVGaussianKernel_ard, [1.0 + .05 * d for d in range(20)] + [1.1]),
                                     (200, 32, "svard", K.SVGaussianKernel_ard, [1.5] * 32 + [.9]),
                                     (200, 32, "ard", K.GaussianKernel_ard, [1.2 + .01 * d for d in range(32)])):
            X, Y = synth(N + D, N, D)
            v, d = marginalLikelihood(cls(hyp), X, Y, len(hyp), True, noise=1e-3)
            ov, od = orc.marginal_likelihood(orc.Kern(name, hyp), X, Y, len(hyp), True, 1e-3)
            close(v, ov); close(d, od, atol=1e-9)
        # DIRECT over a 20-dimensional box: same optimum as the oracle's search on its own objective
        X, Y = synth(140, 150, 20)
        GP = GaussianProcess(K.GaussianKernel_iso([1.5]), X, Y, noise=.1)
        ogp = oracle.GP(oracle.Kern("iso", [1.5]), X, Y, noise=.1)
        opt_v, opt_x = maximizeEI(GP, [[0., 1.]] * 20, maxiter=6)
        o, ox, _ = oracle.acqmax_native(ogp, [[0., 1.]] * 20, oracle.ACQ_EI, .01, maxiter=6)
        close(opt_v, o, atol=ACQ_ATOL); close(opt_x, ox, rtol=1e-9, atol=1e-12)
        # a preference GP in 20 dimensions: C, L = chol(R + C^-1) and the posterior against the oracle fed with the same MAP
        from ibo_amd.gaussianprocess import PrefGaussianProcess
        hyp = [1.2 + .03 * d for d in range(20)]
        pts = np.random.RandomState(150).rand(48, 20)
        score = lambda x: float(np.sum(np.sin(2 * x)))
        prefs = [((a, b, 0) if score(a) > score(b) else (b, a, 0)) for a, b in zip(pts[0::2], pts[1::2])]
        PG = PrefGaussianProcess(K.GaussianKernel_ard(hyp), prefs)
        opg = oracle.pref_fit(oracle.Kern("ard", hyp), prefs, noise=.1, Y_map=PG.Y)
        np.testing.assert_array_equal(PG.X, opg.X)
        close(PG.R, opg.R, rtol=1e-12, atol=1e-14); close(PG.C, opg.C, atol=1e-9); close(PG.L, opg.L, atol=1e-9)
        probe = np.random.RandomState(151).rand(40, 20)
        close(PG.posteriors(probe), opg.posteriors(probe), atol=1e-9)
        with pytest.raises(Exception):
            GaussianProcess(K.GaussianKernel_iso([1.5]), np.random.rand(10, 65), np.random.rand(10), noise=.1).posterior(np.zeros(65))
    finally:
        opt(b"sweep_path", 0); opt(b"dot_form", -1)


def test_thirty_three_to_sixty_four_dimensions(ibo, oracle):
    """D = 33..64 (the reference's kernels are dimension-agnostic, ego/gaussianprocess/kernel.py:147-149): rows of X padded to 64
    coordinates, every batch through the difference-form kernels of sweep.hip (the exponent-GEMM kernels are instantiated up
    to 32 dimensions) -- the fit, large / small / single-point sweeps, the block extension, the likelihood grid, NLML with up to
    65 partial derivatives (four passes of 17) and DIRECT against the CPU oracle; D = 65 is refused"""
    import oracle.oracle as orc
    from ibo_amd import _lib, DeviceArray
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood, nlml_grid
    from ibo_amd.acquisition import sweep, maximizeEI
    opt = lambda k, v: _lib.check(_lib.lib.ibo_set_option(k, v))
    cases = [(700, 33, K.GaussianKernel_ard([2.0 + .01 * d for d in range(33)]), ("ard", [2.0 + .01 * d for d in range(33)])),
             (900, 40, K.MaternKernel5([2.2, 1.0]), ("m5", [2.2, 1.0])),
             (300, 57, K.GaussianKernel_iso([2.5]), ("iso", [2.5])),
             (1100, 64, K.GaussianKernel_ard([2.4] * 64), ("ard", [2.4] * 64)),
             (500, 64, K.MaternKernel3([2.6, 1.0]), ("m3", [2.6, 1.0]))]
    try:
        for N, D, kern, (okind, ohyp) in cases:
            X, Y = synth(N + D, N, D)
            GP = GaussianProcess(kern, X, Y, noise=.1)
            ogp = oracle.GP(oracle.Kern(okind, ohyp), X, Y, noise=.1)
            close(GP.R, ogp.R, rtol=1e-12); close(GP.L, ogp.L, rtol=1e-8, atol=1e-12)
            M = 8300
            cand = np.random.RandomState(N).rand(M, D); cand[77] = X[5]; cand[M - 1] = cand[0]
            r = sweep(GP, cand, acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            assert r["kernel"] == "sweep_mfma_kernel"
            idx = np.r_[np.arange(0, M, M // 40), 77, M - 1, r["best_idx"]]
            o = oracle.sweep_native(ogp, cand[idx], oracle.ACQ_EI, .01)
            close(r["mu"][idx], o["mu"], atol=1e-9); close(r["s2"][idx], o["s2"]); close(r["acq"][idx], o["acq"], atol=ACQ_ATOL)
            assert r["best_idx"] == int(np.argmax(r["acq"]))
            rs = sweep(GP, cand[:300], acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            assert rs["kernel"] == "sweep_mfma_kernel<split>"
            rg = sweep(GP, cand[:9], acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            assert rg["kernel"] == "sweep_gemv_kernel"
            opt(b"dot_form", 1)                                   # a forced dot form is ignored beyond 32 dimensions
            rf = sweep(GP, cand[:300], acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
            opt(b"dot_form", -1)
            for k in ("mu", "s2", "acq"):
                close(rs[k], r[k][:300], rtol=1e-9, atol=1e-11); close(rg[k], r[k][:9], rtol=1e-9, atol=1e-11); close(rf[k], rs[k], rtol=0, atol=0)
            mu1, s21 = GP.posterior(cand[3]); o1 = ogp.posteriors(cand[3:4])
            close(mu1, o1[0][0], atol=1e-9); close(s21, o1[1][0])
        # block extension in 48 dimensions (every round a full sweep: no kept state without the exponent-GEMM kernels)
        N0, D = 500, 48
        X, Y = synth(148, N0 + 3, D)
        kern = K.GaussianKernel_ard([2.3] * D)
        GP = GaussianProcess(kern, X[:N0], Y[:N0], noise=.1)
        dc = DeviceArray.from_host(np.random.RandomState(149).rand(9001, D))
        for rnd in range(3):
            r = sweep(GP, dc, acq='ei', xi=.4, native=False, incremental=True, outputs=("mu", "s2", "acq"))
            assert r["kernel"] == "sweep_mfma_kernel"
            f = sweep(GaussianProcess(kern, GP.X, GP.Y, noise=.1), dc, acq='ei', xi=.4, native=False, outputs=("mu", "s2", "acq"))
            close(r["mu"], f["mu"], rtol=1e-9, atol=1e-10); close(r["s2"], f["s2"], rtol=1e-9); assert r["best_idx"] == f["best_idx"]
            GP.addData(X[len(GP.X)], Y[len(GP.X)])
        ogp = oracle.GP(oracle.Kern("ard", [2.3] * D), np.asarray(GP.X), np.asarray(GP.Y), noise=.1)
        probe = np.random.RandomState(150).rand(30, D)
        close(GP.posteriors(probe), ogp.posteriors(probe), atol=1e-9)
        # NLML and its 41 / 65 / 64 partial derivatives; the grid
        for N, D, name, cls, hyp in ((300, 40, "svard", K.SVGaussianKernel_ard, [2.0 + .02 * d for d in range(40)] + [1.1]),
                                     (200, 64, "svard", K.SVGaussianKernel_ard, [2.5] * 64 + [.9]),
                                     (200, 64, "ard", K.GaussianKernel_ard, [2.2 + .01 * d for d in range(64)])):
            X, Y = synth(N + D, N, D)
            v, d = marginalLikelihood(cls(hyp), X, Y, len(hyp), True, noise=1e-3)
            ov, od = orc.marginal_likelihood(orc.Kern(name, hyp), X, Y, len(hyp), True, 1e-3)
            close(v, ov); close(d, od, atol=1e-9)
        X, Y = synth(164, 400, 50)
        th = np.exp(np.random.RandomState(165).uniform(np.log(1.5), np.log(4.), size=(9, 50)))
        vals, _ = nlml_grid(K.GaussianKernel_ard, th, X, Y, noise=1e-2)
        for t in (0, 4, 8):
            close(vals[t], orc.nlml_c(orc.Kern("ard", th[t]), X, Y, noise=1e-2), rtol=1e-9)
        # DIRECT over a 36-dimensional box
        X, Y = synth(166, 150, 36)
        GP = GaussianProcess(K.GaussianKernel_iso([2.5]), X, Y, noise=.1)
        ogp = oracle.GP(oracle.Kern("iso", [2.5]), X, Y, noise=.1)
        opt_v, opt_x = maximizeEI(GP, [[0., 1.]] * 36, maxiter=4)
        o, ox, _ = oracle.acqmax_native(ogp, [[0., 1.]] * 36, oracle.ACQ_EI, .01, maxiter=4)
        close(opt_v, o, atol=ACQ_ATOL); close(opt_x, ox, rtol=1e-9, atol=1e-12)
    finally:
        opt(b"sweep_path", 0); opt(b"dot_form", -1)
