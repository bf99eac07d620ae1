"""
CPU-side tests (`-m "not gpu"`): the C-ABI library loads and exports every symbol
include/ibo_abi.h declares, compute entry points fail loudly without a GPU, and the
host logic (latin hypercube, both DIRECT implementations, kernel scalars, shard
arithmetic) matches the golden vectors / the oracle.  No GPU compute is called.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ibo_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(ibo_[A-Za-z0-9_]+|acqmaxGP|direct|logCDFs)\s*\(", txt))
    return sorted(n for n in names if n not in ("ibo_gp", "ibo_comm"))


def test_library_exports_every_declared_symbol():
    from ibo_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(_lib.lib, s), "libibo_hip.so does not export %s" % s
        assert s in _lib.EXPORTED, "%s is declared in ibo_abi.h but not bound in ibo_amd/_lib.py" % s
    assert _lib.lib.ibo_abi_version() == 7


def test_the_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """the drop-in boundary is a C ABI (no C++ or torch types in the signatures): include/ibo_abi.h compiles as C99, and a C program
    compiled with gcc links against libibo_hip.so and calls the entry points that need no GPU (version, option switches, the host DIRECT
    with a C callback -- cpp/direct.h:76's signature -- and logCDFs)"""
    import subprocess
    from ibo_amd import _lib
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", os.path.join(inc, "ibo_abi.h")])
    src = tmp_path / "abi_c.c"
    src.write_text(r"""
#include <stdio.h>
#include <stdlib.h>
#include "ibo_abi.h"
static double bowl(int n, double *x) { double s = 0.0; for (int i = 0; i < n; i++) s += (x[i] - 0.3) * (x[i] - 0.3); return s; }
int main(void)
{
    double lb[2] = {0.0, 0.0}, ub[2] = {1.0, 1.0}, fmin = 0.0, xmin[2] = {0.0, 0.0};
    int64_t ns = 0;
    int pairs[4] = {0, 1, 1, 2};
    double y[3] = {0.5, 0.1, -0.2};
    if (ibo_abi_version() != IBO_ABI_VERSION) return 2;
    if (ibo_set_option("legacy_exact", 1) != IBO_OK || ibo_set_option("no such option", 1) == IBO_OK) return 3;
    if (ibo_direct_host(bowl, 2, lb, ub, 20, 10, 10000, 1, &fmin, xmin, &ns) != IBO_OK) return 4;
    const double *r = direct(bowl, 2, lb, ub, 20, 10, 10000);
    if (!r || r[0] != fmin || r[1] != xmin[0] || r[2] != xmin[1]) return 5;
    printf("%.17g %.17g %.17g %lld %.17g\n", fmin, xmin[0], xmin[1], (long long)ns, logCDFs(4, pairs, y));
    free((void *)r);
    return 0;
}
""")
    exe = tmp_path / "abi_c"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-l:libibo_hip.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr[-500:])
    fmin, x0, x1, ns, lc = out.stdout.split()
    assert abs(float(x0) - .3) < .02 and abs(float(x1) - .3) < .02 and float(fmin) < 1e-3 and int(ns) > 50
    # the same numbers through ctypes
    import ctypes
    lb = _lib.f64([0., 0.]); ub = _lib.f64([1., 1.]); xm = np.empty(2); fm = ctypes.c_double(); n2 = ctypes.c_int64()
    cb = _lib.OBJECTIVE(lambda n, x: sum((x[i] - 0.3) ** 2 for i in range(n)))
    _lib.check(_lib.lib.ibo_direct_host(cb, 2, _lib.dp(lb), _lib.dp(ub), 20, 10, 10000, 1, ctypes.byref(fm), _lib.dp(xm), ctypes.byref(n2)))
    assert n2.value == int(ns) and np.allclose([xm[0], xm[1]], [float(x0), float(x1)], rtol=0, atol=1e-15)


def test_no_gpu_means_loud_failure_not_fallback():
    from ibo_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    with pytest.raises(_lib.IBOError) as e:
        GaussianProcess(GaussianKernel_ard([.5, .5]), np.random.rand(5, 2), np.random.rand(5))
    assert e.value.code == _lib.ERR_NO_DEVICE
    with pytest.raises(_lib.IBOError):
        GaussianKernel_ard([.5, .5]).covMatrix(np.random.rand(4, 2))
    # the legacy symbol reports failure the way the reference does for bad input: NULL
    z = np.zeros(4)
    dp = _lib.dp
    assert not _lib.lib.acqmaxGP(1, dp(z), dp(z), dp(z), dp(z), dp(z), 1, 0, 0, dp(z), 0, dp(z), dp(z), 0.0, dp(z),
                                 dp(z), .01, .1, 1, 1, 10)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ibo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# the checker", ""), "%s mentions the oracle" % f


def test_lhc_bit_exact():
    from ibo_amd.utils.latinhypercube import lhcSample
    g = load_golden("g9_lhc")
    for seed, n in ((22, 5), (23, 5), (0, 10), (7, 12)):
        got = np.array(lhcSample(g["s%d_bounds" % seed].tolist(), n, seed=seed))
        assert np.array_equal(got, g["s%d" % seed])
    s = lhcSample([[0., 1.]], 100, seed=20)                  # ego/unittest_IBO.py:51-62
    assert sorted(int(v[0] * 100) for v in s) == list(range(100))


def shekel5(g):
    A, C = g["shekel_A"], g["shekel_C"]
    return lambda x: -sum(1. / (np.dot(x - a, x - a) + c) for a, c in zip(A, C))


def test_direct_known_answers(oracle):
    from ibo_amd.utils.optimize import direct, cdirect
    g = load_golden("g4_direct")
    f = shekel5(g)
    b = g["shekel_bounds"].tolist()
    fm, xm, ns = cdirect(f, b, maxiter=20, return_samples=True)
    assert np.array_equal(np.r_[fm, xm], g["shekel_cdirect"])
    assert ns == oracle.cdirect(f, b, maxiter=20)[2]
    assert abs(fm + 10.1532) < 1e-3 and np.all(np.abs(xm - 4.0) < 1e-3)
    fm, xm = direct(f, b, maxiter=20)
    assert np.array_equal(np.r_[fm, xm], g["shekel_direct"])

    def foo(x, a1, a2):
        return -np.sum(np.sin(np.array(x) * a1) + np.array(x) * a2)
    b3 = [[0., 5.]] * 3
    assert np.array_equal(np.r_[cdirect(foo, b3, args=[3.0, 0.0], maxiter=10)], np.r_[g["foo_c1"][0], g["foo_c1"][1:]])
    fm, xm = cdirect(foo, b3, args=[-2.0, 2.0], maxiter=10)
    assert np.array_equal(np.r_[fm, xm], g["foo_c2"])
    fm, xm = direct(foo, b3, args=[3.0, 0.0], maxiter=20)
    assert np.array_equal(np.r_[fm, xm], g["foo_d1"])
    with pytest.raises(ValueError):
        direct(f, b)
    (fm2, xm2), rep = direct(foo, b3, args=[3.0, 0.0], maxiter=3, debug=True)
    assert rep["samples"] > 0 and len(rep["rectangles"]) > 3
    foo3 = lambda x: float(np.sum((np.array(x) - .3) ** 2))
    for row, bb in zip(g["fixed_counts"], ([[0., 1.]] * 3, [[0., 1.], [.5, .5], [0., 1.]], [[.5, .5], [0., 1.], [0., 1.]])):
        fm, xm, ns = cdirect(foo3, bb, maxiter=50, maxsample=10000, return_samples=True)
        assert ns == int(row[0]) and np.array_equal(np.r_[fm, xm], row[1:])
    fm, xm, ns = cdirect(foo3, [[.5, .5], [0., 1.], [0., 1.]], maxiter=50, maxsample=10000, compat=False,
                         return_samples=True)
    assert ns > 1000 and abs(fm - .04) < 1e-9


def test_direct_grouped_selection_matches_the_all_pairs_reference(oracle):
    """the host DIRECT picks its potentially-optimal rectangles from per-size minima (O(R)) and removes divided
    rectangles in one pass; the oracle (and the compiled reference) test all pairs and erase one by one: same
    minimum, same point, same number of samples on assorted objectives, including plateaus and ties"""
    from ibo_amd.utils.optimize import cdirect
    rs = np.random.RandomState(17)
    cases = []
    for D in (1, 2, 3, 5, 8):
        c = rs.rand(D); w = rs.uniform(.5, 3, D); ph = rs.rand(D) * 6
        cases.append((lambda x, c=c, w=w, ph=ph: float(np.sum(w * (x - c) ** 2) + .3 * np.sum(np.sin(9 * x + ph))), D))
    cases.append((lambda x: float(np.round(np.sum((x - .4) ** 2), 1)), 3))          # plateaus: many equal values
    cases.append((lambda x: 0.0, 2))                                                 # everything ties
    cases.append((lambda x: float(-np.prod(np.cos(5 * x))), 4))
    for f, D in cases:
        b = [[0., 1.]] * D
        for maxiter in (7, 35):
            fm, xm, ns = cdirect(f, b, maxiter=maxiter, maxsample=20000, return_samples=True)
            o = oracle.cdirect(f, b, maxiter=maxiter, maxsample=20000)
            assert ns == o[2] and fm == o[0] and np.array_equal(xm, o[1]), (D, maxiter)
    if oracle.RefLib.available():
        f, D = cases[3]
        fm, xm, ns = cdirect(f, [[0., 1.]] * D, maxiter=30, maxsample=20000, return_samples=True)
        ref = oracle.RefLib().direct(f, [[0., 1.]] * D, maxiter=30, maxsample=20000)
        assert ns == ref[2] and fm == ref[0] and np.array_equal(xm, ref[1])


def test_direct_against_the_compiled_reference_on_random_objectives(oracle):
    """csrc/direct_host.cpp (per-class decisions, hull witnesses, one batch per iteration) against the reference's OWN compiled search
    (oracle/_ref/libego.so, cpp/direct.cpp:329-581) on 160 random objectives in 1..7 dimensions -- smooth, rippled, terraced (ties inside
    every size class), clipped at zero like an expected improvement (exact zeros over most of the box), shifted boxes, boxes with a fixed dimension -- at two budgets:
    the same minimum, the same point, the same number of samples, value for value"""
    if not oracle.RefLib.available():
        pytest.skip("oracle/_ref/libego.so not built")
    from ibo_amd.utils.optimize import cdirect
    ref = oracle.RefLib()
    rs = np.random.RandomState(4242)
    ncase = 0
    for case in range(160):
        D = 1 + case % 7
        c = rs.rand(D); w = rs.uniform(.5, 3, D); ph = rs.rand(D) * 6
        kind = case % 4
        if kind == 0: f = lambda x, c=c, w=w: float(np.sum(w * (x - c) ** 2))
        elif kind == 1: f = lambda x, c=c, w=w, ph=ph: float(np.sum(w * (x - c) ** 2) + .3 * np.sum(np.sin(9 * x + ph)))
        elif kind == 2: f = lambda x, c=c: float(np.floor(6 * np.sum(np.abs(x - c))) / 6)
        else: f = lambda x, c=c, D=D: -max(0.0, .04 * D - float(np.sum((x - c) ** 2)))
        lo = rs.uniform(-2, 0, D); b = [[float(l), float(l + rs.uniform(.5, 3))] for l in lo]
        if case % 7 == 3: b = [[0., 1.]] * D
        if D >= 2 and case % 11 == 5: b[case % D] = [b[case % D][0]] * 2            # a fixed dimension (dimension 0 among them: the stall)
        # (a fixed dimension maps to 0: the compiled reference does not survive an objective that returns NaN)
        fb = (lambda x, f=f, b=b: f((np.asarray(x) - np.array([q[0] for q in b])) / np.array([(q[1] - q[0]) or 1.0 for q in b])))
        for maxiter, maxsample in ((6, 20000), (28, 1500)):
            fm, xm, ns = cdirect(fb, b, maxiter=maxiter, maxsample=maxsample, return_samples=True)
            r = ref.direct(fb, b, maxiter=maxiter, maxsample=maxsample)
            assert ns == r[2] and fm == r[0] and np.array_equal(xm, r[1]), (case, D, kind, maxiter, ns, r[2])
            ncase += 1
    assert ncase == 320


def test_direct_one_batch_per_iteration_equals_the_sequential_search(oracle):
    """the schedule ibo_direct_max runs the GPU objective under: probes and GUESSED child centres of all potentially-optimal
    rectangles in one batch per iteration, the guesses verified bit for bit against the centres the division produces and
    re-sampled where they differ -- same minimum, point and sample count as the sequential search, and the objective is
    never asked for a point the sequential search does not visit (beyond the guesses that turn out wrong)"""
    from ibo_amd.utils.optimize import cdirect
    rs = np.random.RandomState(29)
    cases = []
    for D in (1, 2, 3, 4, 6, 8, 17):
        c = rs.rand(D); w = rs.uniform(.5, 3, D); ph = rs.rand(D) * 6
        cases.append((lambda x, c=c, w=w, ph=ph: float(np.sum(w * (x - c) ** 2) + .3 * np.sum(np.sin(9 * x + ph))), D, [[0., 1.]] * D))
    cases.append((lambda x: float(np.round(np.sum((x - .4) ** 2), 1)), 3, [[0., 1.]] * 3))
    cases.append((lambda x: 0.0, 2, [[0., 1.]] * 2))
    cases.append((lambda x: float(np.sum((x - .3) ** 2)), 3, [[0., 1.], [.5, .5], [0., 1.]]))       # a fixed dimension
    cases.append((lambda x: float(np.sum((x - .3) ** 2)), 3, [[.5, .5], [0., 1.], [0., 1.]]))       # the dimension-0 stall
    cases.append((lambda x: float(np.sum(np.cos(7 * x))), 5, [[-3., 11.]] * 5))
    for f, D, b in cases:
        for maxiter, maxsample in ((1, 20000), (9, 20000), (40, 20000), (40, 300)):
            seen_seq, seen_bat = set(), set()

            def fs(x, seen=seen_seq): seen.add(x.tobytes()); return f(x)
            def fb(x, seen=seen_bat): seen.add(x.tobytes()); return f(x)
            a = cdirect(fs, b, maxiter=maxiter, maxsample=maxsample, return_samples=True)
            c = cdirect(fb, b, maxiter=maxiter, maxsample=maxsample, return_samples=True, batched=True)
            assert a[0] == c[0] and np.array_equal(a[1], c[1]) and a[2] == c[2], (D, maxiter, maxsample)
            if maxsample > 300:
                # every point of the sequential search is sampled; the extra ones are guesses that did not survive the check
                assert seen_seq <= seen_bat and len(seen_bat - seen_seq) <= max(2, len(seen_seq) // 500), (len(seen_seq), len(seen_bat))
    # the guess really is wrong now and then (that is why it is verified): count over a long search in a stretched box
    seen_seq, seen_bat = set(), set()
    g = lambda x: float(np.sum((x - 1.2345) ** 2))
    a = cdirect(lambda x: (seen_seq.add(x.tobytes()), g(x))[1], [[-7.3, 11.9]] * 3, maxiter=120, maxsample=100000, return_samples=True)
    c = cdirect(lambda x: (seen_bat.add(x.tobytes()), g(x))[1], [[-7.3, 11.9]] * 3, maxiter=120, maxsample=100000, return_samples=True, batched=True)
    assert a[0] == c[0] and np.array_equal(a[1], c[1]) and a[2] == c[2]
    print("DIRECT: %d distinct points, %d guessed centres re-sampled" % (len(seen_seq), len(seen_bat - seen_seq)))


def test_direct_beyond_sixteen_dimensions_orders_ties_as_the_reference_library(oracle):
    """more than 16 longest sides: cpp/direct.cpp:194's std::sort is an introsort there, and with equal probe values the
    cut order is whatever it leaves.  The host DIRECT calls std::sort, the oracle restates libstdc++'s algorithm; both
    against the compiled reference on objectives made of ties (symmetric, plateaus, constants) in 17..32 dimensions"""
    from ibo_amd.utils.optimize import cdirect
    rs = np.random.RandomState(23)
    cases = []
    for D in (17, 20, 27, 32):
        cases.append((lambda x: float(np.sum((x - .5) ** 2)), D))                       # every dimension ties
        cases.append((lambda x: float(np.round(np.sum(np.abs(x - .3)), 0)), D))           # plateaus
        lv = rs.randint(0, 3, D).astype(float)
        cases.append((lambda x, lv=lv: float(np.sum(lv * np.round(3 * x))), D))           # a few distinct values, many repeats
        c = rs.rand(D)
        cases.append((lambda x, c=c: float(np.sum((x - c) ** 2)), D))                   # no ties
    cases.append((lambda x: 0.0, 19))
    have_ref = oracle.RefLib.available()
    for f, D in cases:
        b = [[0., 1.]] * D
        fm, xm, ns = cdirect(f, b, maxiter=4, maxsample=4000, return_samples=True)
        o = oracle.cdirect(f, b, maxiter=4, maxsample=4000)
        assert ns == o[2] and fm == o[0] and np.array_equal(xm, o[1]), D
        if have_ref:
            ref = oracle.RefLib().direct(f, b, maxiter=4, maxsample=4000)
            assert ns == ref[2] and fm == ref[0] and np.array_equal(xm, ref[1]), D


def test_legacy_direct_symbol_against_compiled_reference(oracle):
    """`direct` with the reference's exact signature (cpp/direct.h:76), same answers as _ref/libego.so"""
    from ibo_amd import _lib
    g = load_golden("g4_direct")
    f = shekel5(g)
    b = np.array(g["shekel_bounds"])
    lb, ub = _lib.f64(b[:, 0]), _lib.f64(b[:, 1])
    cnt = [0]

    def obj(n, x):
        cnt[0] += 1
        return float(f(np.array([x[i] for i in range(n)])))
    r = _lib.lib.direct(_lib.OBJECTIVE(obj), 4, _lib.dp(lb), _lib.dp(ub), 25, 30, 100000)
    res = np.array([r[i] for i in range(5)])
    libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]; libc.free(r)
    if oracle.RefLib.available():
        ref = oracle.RefLib().direct(f, b.tolist(), maxiter=25, maxsample=100000)
        assert cnt[0] == ref[2]
        assert np.array_equal(res, np.r_[ref[0], ref[1]])
    o = oracle.cdirect(f, b.tolist(), maxiter=25, maxsample=100000)
    assert cnt[0] == o[2] and np.array_equal(res, np.r_[o[0], o[1]])


def test_legacy_logCDFs_symbol_against_compiled_reference(oracle):
    """`logCDFs` (cpp/helpers.cpp:30-56): pair-stride sum of log(Phi(dx / sqrt 2) / sqrt 2), zero terms skipped"""
    import math
    from ibo_amd import _lib
    rs = np.random.RandomState(8)
    x = np.ascontiguousarray(rs.randn(12) * 3)
    x[3] = -60.0                                        # Phi underflows to exactly 0: that pair is skipped
    p = np.ascontiguousarray(np.r_[rs.randint(0, 12, size=38), 3, 0].astype(np.int32))   # last pair: x[3] - x[0]
    ip = p.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    got = _lib.lib.logCDFs(len(p), ip, _lib.dp(x))
    want = 0.0
    for i in range(0, len(p), 2):
        q = 0.5 * (1 + math.erf((x[p[i]] - x[p[i + 1]]) / math.sqrt(2)))
        if q / math.sqrt(2) != 0.0:
            want += math.log(q / math.sqrt(2))
    assert got == pytest.approx(want, rel=1e-14)
    assert _lib.lib.logCDFs(0, ip, _lib.dp(x)) == 0.0
    if oracle.RefLib.available():
        ref = oracle.RefLib().lib
        ref.logCDFs.restype = ctypes.c_double
        ref.logCDFs.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
        assert ref.logCDFs(len(p), ip, _lib.dp(x)) == got


def test_kernel_scalars_and_specs(oracle):
    from ibo_amd.gaussianprocess import kernel as K, CDF, PDF, erf
    rs = np.random.RandomState(3)
    a, b = rs.rand(5), rs.rand(5)
    for ours, kind, hyp in ((K.GaussianKernel_ard([.3, .4, .5, .6, .7]), "ard", [.3, .4, .5, .6, .7]),
                            (K.GaussianKernel_iso([.4]), "iso", [.4]),
                            (K.SVGaussianKernel_iso([.4, 1.3]), "sviso", [.4, 1.3]),
                            (K.SVGaussianKernel_ard([.3, .4, .5, .6, .7, .8]), "svard", [.3, .4, .5, .6, .7, .8]),
                            (K.MaternKernel3([.6, .9]), "m3", [.6, .9]), (K.MaternKernel5([.5, 1.2]), "m5", [.5, 1.2])):
        ok = oracle.Kern(kind, hyp)
        np.testing.assert_allclose(ours.cov(a, b), ok.cov(a, b), rtol=1e-14)
        kt, h, sf2, sf2n = ours._ibo_spec()
        assert kt == ok.ktype and abs(sf2 - ok.sf2_py) < 1e-15 and abs(sf2n - ok.sf2_native) < 1e-15
        assert np.array_equal(ours.hyperparams, np.array(hyp))
        with pytest.raises(ValueError):
            ours.hyperparams[0] = 5.0                     # read-only (kernel.py:32-40)
    assert K.GaussianKernel_ard([1e-9, 1e9])._theta.tolist() == [1e-4, 1e4]      # clip (kernel.py:141)
    for z in (-3.1, -.2, 0.0, .7, 2.5):
        assert erf(z) == oracle.lib().orc_erf_nr(z)
        assert abs(CDF(z) - 0.5 * (1 + oracle.lib().orc_erf_nr(z * 0.707106))) < 1e-16
    assert PDF(0.3) == np.exp(-(0.3 ** 2 / 2)) * 0.398942


def test_shard_bounds_and_slot_reduce():
    from ibo_amd.multigpu import shard_bounds, fill_slot, reduce_slots
    for M, W in ((4194304, 8), (10, 3), (7, 8), (1 << 20, 1)):
        cuts = [shard_bounds(M, W, r) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == M
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
        assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
    assert shard_bounds(4194304, 8, 3) == (3 * 524288, 4 * 524288)
    W = 4
    bufs = [fill_slot(W, 0, 1.5, 40, [1., 2.]), fill_slot(W, 1, 2.5, 300, [3., 4.]),
            fill_slot(W, 2, 2.5, 200, [5., 6.]), fill_slot(W, 3, float('nan'), -1, [0., 0.])]
    v, i, p, r = reduce_slots(sum(bufs), W, 2)
    assert (v, i, r) == (2.5, 200, 2) and p.tolist() == [5., 6.]       # tie -> lowest global index
    v, i, p, r = reduce_slots(sum(fill_slot(W, k, 0.0, -1, [0.]) for k in range(W)), W, 1)
    assert i == -1 and r == -1
    v, i, p, r = reduce_slots(fill_slot(1, 0, -3.0, 0, []), 1, 0)
    assert (v, i, r) == (-3.0, 0, 0)


def test_install_as_ego_aliases_every_module():
    import importlib
    import ibo_amd
    ibo_amd.install_as_ego()
    for name in ("ego", "ego.gaussianprocess", "ego.gaussianprocess.kernel", "ego.gaussianprocess.prior",
                 "ego.gaussianprocess.trainhyper", "ego.acquisition", "ego.acquisition.gallery", "ego.utils.optimize",
                 "ego.utils.latinhypercube"):
        assert importlib.import_module(name) is importlib.import_module(name.replace("ego", "ibo_amd", 1))
    from ego.acquisition import maximizeEI, maximizePI, maximizeUCB, EI, PI, UCB, cdirectGP      # noqa: F401
    from ego.acquisition.gallery import fastUCBGallery                                 # noqa: F401
    from ego.gaussianprocess import GaussianProcess, PrefGaussianProcess, CDF, PDF     # noqa: F401
    from ego.utils.optimize import direct, cdirect                                     # noqa: F401
    import inspect
    sig = inspect.signature(maximizeEI)
    assert [p for p in sig.parameters][:6] == ["model", "bounds", "useCDIRECT", "xi", "maxiter", "maxtime"]
    assert sig.parameters["maxsample"].default == 10000 and sig.parameters["xi"].default == 0.01
    assert inspect.signature(fastUCBGallery).parameters["samples"].default == 300
    # cdirectGP: the reference's positional order and defaults (ego/acquisition/__init__.py:307), `beta` included
    sig = inspect.signature(cdirectGP)
    assert [p for p in sig.parameters][:10] == ["model", "bounds", "maxiter", "maxtime", "maxsample", "acqfunc", "xi", "beta",
                                                "scale", "delta"]
    assert [sig.parameters[k].default for k in ("acqfunc", "xi", "beta", "scale", "delta")] == [None, -1, -1, -1, -1]
    sig = inspect.signature(GaussianProcess.__init__)
    assert sig.parameters["noise"].default == .1 and sig.parameters["gnoise"].default == 1e-4


def test_python_direct_matches_the_reference_restated(oracle):
    """ibo_amd.utils.optimize.direct (columns of boxes, size-grouped selection) against the oracle's statement-level
    restatement of ego/utils/optimize.py:58-280: same minimum, same location, same number of samples, on smooth,
    plateau, all-ties and zero-minimum objectives and under each termination rule"""
    from ibo_amd.utils.optimize import direct

    def bowl(D, seed):
        c = np.random.RandomState(seed).rand(D)
        return lambda x: float(np.sum((np.array(x) - c) ** 2) + 0.3 * np.sin(5 * np.sum(x)))
    cases = [(bowl(2, 1), [[-1., 2.]] * 2, dict(maxiter=12)), (bowl(3, 2), [[-1., 2.]] * 3, dict(maxiter=9)),
             (bowl(4, 3), [[0., 1.]] * 4, dict(maxsample=400)), (bowl(3, 5), [[-2., 1.]] * 3, dict(maxiter=30, maxsample=500)),
             (lambda x: 0.0, [[0., 1.]] * 2, dict(maxiter=8)),
             (lambda x: float(np.round(np.sum(np.abs(np.array(x))), 1)), [[-1., 1.]] * 2, dict(maxiter=8)),
             (lambda x: float(abs(x[0] - .5) + abs(x[1] - .5)), [[0., 1.]] * 2, dict(maxiter=10)),
             (lambda x, a: float((x[0] - a) ** 2), [[0., 3.]], dict(maxiter=15, args=[1.25]))]
    for f, b, kw in cases:
        (fm, xm), rep = direct(f, b, debug=True, **kw)
        o = oracle.direct_py(f, b, **kw)
        assert fm == o[0] and np.array_equal(xm, o[1]) and rep["samples"] == o[2], (kw, fm, o)
        assert len(rep["fmin evolution"]) >= 1 and all(hasattr(r, "lb") for r in rep["rectangles"])


def test_gallery_and_preference_host_helpers():
    from ibo_amd.acquisition.gallery import _best_observation_inside, _separated
    from ibo_amd.gaussianprocess import PrefGaussianProcess
    X = np.array([[.1, .1], [.9, .9], [2., .5], [.9, .9]])
    Y = np.array([1., 3., 9., 3.])
    b = [[0., 1.], [0., 1.]]
    np.testing.assert_array_equal(_best_observation_inside(X, Y, b), [.9, .9])       # best INSIDE the box, first of ties
    assert _best_observation_inside(X[2:3], Y[2:3], b) is None
    assert _separated([0., 0.], []) and _separated([0., 0.], [[.6, 0.]]) and not _separated([0., 0.], [[.5, 0.]])
    a, c, d = np.array([.1, .2]), np.array([.3, .4]), np.array([.5, .6])
    pts, pairs, fav = PrefGaussianProcess._index_preferences([(a, c, 0), (d, a, 1), (c, d, 0), (np.array([.1, .2]), d, 0)])
    np.testing.assert_array_equal(pts, [a, c, d])                                   # order of first appearance
    assert pairs == [(0, 1, 0), (2, 0, 1), (1, 2, 0), (0, 2, 0)] and fav == [True, True, True]
    pts, pairs, fav = PrefGaussianProcess._index_preferences([(a, c, 0), (a, np.array([0.0, -0.0]), 0), (a, np.array([0.0, 0.0]), 0)])
    assert len(pts) == 3 and fav == [True, False, False] and pairs[1][1] == pairs[2][1]


def test_test_function_zoo_known_minima():
    """analytic objectives of ego/utils/testfunctions.py: the documented minima at the documented places"""
    from ibo_amd.utils import testfunctions as tfs
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_iso, MaternKernel3
    for cls in (tfs.Shekel5, tfs.Shekel7, tfs.Shekel10, tfs.Hartman3, tfs.Hartman6, tfs.Branin):
        tf = cls(maximize=False)
        assert abs(tf.f(tf.argmin) - tf.minimum) < 2e-3, cls.__name__
        assert tf.f(tf.argmin) == -cls().f(tf.argmin)                               # maximize flips the sign
        rnd = np.random.RandomState(0).rand(500, len(tf.bounds)) * (np.array(tf.bounds)[:, 1] - np.array(tf.bounds)[:, 0]) \
            + np.array(tf.bounds)[:, 0]
        v = tf.values(rnd)
        assert v.shape == (500,) and np.all(v >= tf.f(tf.argmin) - 1e-9)
        assert abs(tf.f(rnd[3]) - v[3]) < 1e-15
    cb = tfs.Camelback(maximize=False)
    assert abs(cb.f([-0.0898, 0.7126]) + 1.0316) < 1e-3 and abs(cb.f([0.0898, -0.7126]) + 1.0316) < 1e-3
    assert abs(tfs.GoldsteinPrice(maximize=False).f([0., -1.]) - np.log(3.)) < 1e-12
    assert tfs.Sphere(3).f(np.zeros(3)) == 0.0 and tfs.SumSquares(3, maximize=False).f([1., 1., 1.]) == 6.0
    g = load_golden("g4_direct")
    x = np.array([3.7, 4.2, 4.1, 3.9])
    assert abs(tfs.Shekel5(maximize=False).f(x) - shekel5(g)(x)) < 1e-14             # the golden file's own Shekel data
    assert isinstance(tfs.Shekel5().createKernel(GaussianKernel_iso), GaussianKernel_iso)
    with pytest.raises(ValueError):
        tfs.Shekel5().createKernel(MaternKernel3)


def test_every_analytic_test_function_against_the_reference_values():
    """G10: f(x) of all nineteen analytic classes of ego/utils/testfunctions.py (21 + 2 instances) on seeded points,
    both signs, names and recorded minima, generated from the reference's own classes (make_golden.py g10)"""
    from ibo_amd.utils import testfunctions as tfs
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_iso, GaussianKernel_ard
    g = load_golden("g10_testfunctions")
    made = {"Michalewics2": (tfs.Michalewics, dict(d=2)), "Michalewics5": (tfs.Michalewics, dict(d=5)),
            "Michalewics10": (tfs.Michalewics, dict(d=10)), "Perm4": (tfs.Perm, dict(d=4)), "Perm3": (tfs.Perm, dict(d=3)),
            "Sphere4": (tfs.Sphere, dict(d=4)), "SumSquares4": (tfs.SumSquares, dict(d=4)), "SumSquares8": (tfs.SumSquares, dict(d=8)),
            "Zakharov2": (tfs.Zakharov, dict(d=2)), "Zakharov5": (tfs.Zakharov, dict(d=5)), "Levy2": (tfs.Levy, dict(d=2)),
            "Levy4": (tfs.Levy, dict(d=4))}
    assert len(g["names"]) == 23
    for name in g["names"]:
        name = str(name)
        cls, kw = made.get(name, (getattr(tfs, name, None), {}))
        assert cls is not None, name
        tf = cls(maximize=False, **kw)
        X = g[name + "_x"]
        np.testing.assert_allclose(tf.values(X), g[name + "_f"], rtol=1e-12, atol=1e-13, err_msg=name)
        np.testing.assert_allclose(cls(**kw).values(X), g[name + "_fmax"], rtol=1e-12, atol=1e-13, err_msg=name)
        assert abs(tf.f(X[5]) - g[name + "_f"][5]) <= 1e-12 * max(1.0, abs(g[name + "_f"][5]))
        np.testing.assert_array_equal(np.array(tf.bounds, dtype=float), g[name + "_bounds"])
        assert tf.name == str(g[name + "_name"]) and float(tf.minimum) == float(g[name + "_min"]), name
    # the kernels the reference tunes for the new classes (testfunctions.py:78-80, 321-326, 350-355, 409-414, 432)
    assert tfs.Poly4().createKernel(GaussianKernel_ard).hyperparams[0] == 1.628
    assert tfs.Levy(4).createKernel(GaussianKernel_iso).hyperparams[0] == 2.8
    assert tfs.Michalewics(10).createKernel(GaussianKernel_iso).hyperparams[0] == 1.36
    assert tfs.SumSquares(8).createKernel(GaussianKernel_iso).hyperparams[0] == 0.5
    assert tfs.Zakharov(3).createKernel(GaussianKernel_iso).hyperparams[0] == 0.5
    with pytest.raises(ValueError):
        tfs.Levy(3).createKernel(GaussianKernel_iso)
    with pytest.raises(ValueError):
        tfs.SumSquares(4).createKernel(GaussianKernel_ard)
    # the documented minima
    assert tfs.Perm(4, maximize=False).f(np.arange(1., 5.)) == 0.0 and tfs.Levy(3, maximize=False).f(np.ones(3)) < 1e-30
    assert abs(tfs.Michalewics(2, maximize=False).f([2.2029, 1.5708]) + 1.8013) < 1e-3
    assert abs(tfs.Schubert1(maximize=False).f([-0.195]) + 8.5178) < 2e-3
    found = tfs.checkMinimum([tfs.Zakharov(2, maximize=False)], samples=20, seed=3)
    assert found[0][0] == "Zakharov 2" and found[0][1] < 1e-8


def test_prefix_variance_bounds_the_reference_acquisitions(oracle):
    """What the two-part kept state (sweep2.hip: launch_sweep2_pruned) rests on, checked on the oracle's own arithmetic: with W = L^-1,
    q = |W k*|^2 summed over a PREFIX of W's rows is a lower bound of q, so 1 + noise - q_prefix bounds the variance from above; EI
    (libm and NR erf) and UCB evaluated there bound the candidate's value from above -- up to the rounding noise of the reference's EI
    where it is ~1e-16, which the slack of s2_part_limit covers; and the selection the device performs (the top of the bound ranking
    first, their best value as the threshold for the rest) never leaves the true maximiser incomplete."""
    orc = oracle
    rs = np.random.RandomState(41)
    for kind, hyper, D, noise in (("ard", [.3, .35, .4], 3, .01), ("m5", [.5, 1.0], 4, .001), ("m3", [.4, 1.0], 2, .1)):
        N, M = 192, 960
        X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .05 * rs.randn(N)
        kern = orc.Kern(kind, np.array(hyper, float))
        gp = orc.GP(kern, X, Y, noise=noise)
        Q = rs.rand(M, D)
        mu, s2 = gp.posteriors(Q)
        XQ = np.ascontiguousarray(np.vstack([X, Q]))                # the correlation matrix of all points holds K(X, Q) off the diagonal
        Rb = np.empty((N + M, N + M))
        orc.lib().orc_build_R(kern.ktype, N + M, D, XQ.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                              kern.oracle_hyper().ctypes.data_as(ctypes.POINTER(ctypes.c_double)), kern.sf2_py, noise,
                              Rb.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        Ks = Rb[:N, N:]
        V = np.linalg.solve(gp.L, Ks)                     # W k* per candidate (columns)
        q_rows = V * V
        q = q_rows.sum(0)
        s2_raw = 1.0 + noise - q
        ok = (s2_raw > 1e-7) & (s2_raw < 10)
        np.testing.assert_allclose(s2[ok], s2_raw[ok], rtol=1e-8, atol=1e-12)   # the oracle's variance IS 1 + noise - |W k*|^2
        h = N // 2
        q_a = q_rows[:h].sum(0)
        assert np.all(q_a <= q * (1 + 1e-15))
        s2_ub = np.clip(1.0 + noise - q_a, 1e-7, 10.0)
        ymax = float(Y.max())
        for acq, erf_mode, parm in ((orc.ACQ_EI, orc.ERF_LIBM, .01), (orc.ACQ_EI, orc.ERF_NR, .4), (orc.ACQ_UCB, orc.ERF_LIBM, 1.7)):
            exact = orc.acq_value(acq, erf_mode, mu, np.sqrt(np.clip(s2_raw, 1e-7, 10.0)), ymax, parm)
            bound = orc.acq_value(acq, erf_mode, mu, np.sqrt(s2_ub), ymax, parm)
            slack = 1e-9 * np.abs(exact) + 1e-13 * (1 + abs(ymax) + abs(parm))
            assert np.all(bound >= exact - slack), float(np.min(bound - exact))
            # the device's selection on 32-candidate tiles
            tb = bound.reshape(-1, 32).max(1); te = exact.reshape(-1, 32).max(1)
            cut = np.sort(tb)[::-1][max(0, len(tb) * 3 // 100)]
            done = tb >= cut
            best = te[done].max()
            lim = best - (1e-9 * abs(best) + 1e-13 * (1 + abs(ymax) + abs(parm)))
            done |= tb >= lim
            assert done[int(np.argmax(exact)) // 32]
            assert te[done].max() == exact.max()


def test_direct_host_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The library's host-only C++ (DIRECT's tree logic, csrc/direct_host.cpp) built with g++ -fsanitize=address,undefined and driven
    through 770 cases (tools/direct_host_check.cpp: 1..12 dimensions, both evaluation schedules -- which must agree in fmin, xmin and
    samples --, fixed dimensions, sample budgets of 1 and 7, constant / infinite objectives, an aborting evaluator).  CPU only: the
    GPU build cannot run under a sanitizer on this pool."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++")
    probe = tmp_path / "p.cpp"
    probe.write_text("int main(){return 0;}\n")
    if subprocess.run(["g++", "-fsanitize=address,undefined", str(probe), "-o", str(tmp_path / "p")], capture_output=True).returncode != 0:
        pytest.skip("g++ has no sanitizer run-time here")
    exe = tmp_path / "direct_host_check"
    # (IBO_DIRECT_SELFCHECK: every call of the potentially-optimal test -- per-class decisions, hull witnesses -- is compared with the full test
    # on every candidate rectangle, and a difference aborts)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-DIBO_DIRECT_SELFCHECK", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I", os.path.join(root, "ibo_amd", "csrc"), os.path.join(root, "tools", "direct_host_check.cpp"),
                    os.path.join(root, "ibo_amd", "csrc", "direct_host.cpp"), "-o", str(exe)], check=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failure(s)" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
