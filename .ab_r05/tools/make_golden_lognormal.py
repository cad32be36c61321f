#!/usr/bin/env python3
"""Golden fixtures for the LogNormal branch (run in the BUILD container only; imports the reference).

    python3 tools/make_golden_lognormal.py         # ~4 min
    python3 tools/make_golden_lognormal.py N300    # lognormal_N300.npz: the seed MAP solve at N = 300 only
    python3 tools/make_golden_lognormal.py N300_full   # lognormal_N300_full.npz: a WHOLE fit at N = 300 on the 1e6-
                                                       # visibility fixture's M, j (two processes: M and M(1+1e-15); ~3 min)
    python3 tools/make_golden_lognormal.py N300_1e7    # lognormal_N300_1e7.npz: BASELINE configs[2] itself (M, j of
                                                       # the 1e7-visibility Normal fixture)

Writes tests/golden/lognormal_N40.npz and lognormal_N80.npz:
  * one LogNormalMAPModel solve (statistical_models.py:1012-1160) on the seed power spectrum: inputs, s_MAP,
    Hessian at the MAP, posterior covariance diagonal, MinimizeNewton exit statistics;
  * one CriticalFilter.update_power_spectrum from that solve (filter.py:154-177);
  * whole FrankFitter(method='LogNormal') fits (radial_fitters.py:737-832): iteration count, final I and p, the
    first passes' diagnostics -- and the reference's OWN sensitivity: the same fit after M is perturbed by 1e-15
    relative (1 ulp-scale), because the Newton iteration is driven into round-off by design and the reference
    ignores its exit status (statistical_models.py:1142-1145).  Parity tests use that spread as their scale.
"""
import collections
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import scipy  # noqa: E402
import frank  # noqa: E402
import frank.statistical_models as sm  # noqa: E402
from frank.filter import CriticalFilter  # noqa: E402
from frank.geometry import FixedGeometry  # noqa: E402
from frank.radial_fitters import FrankFitter  # noqa: E402
from frank.statistical_models import GaussianModel, LogNormalMAPModel  # noqa: E402

from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
META = dict(reference_version=frank.__version__, numpy=np.__version__, scipy=scipy.__version__)
RMAX = 2.0
I_SCALE = 1e5

_newton_stats = []
_orig_newton = sm.MinimizeNewton


def _recording_newton(*a, **k):
    x, st = _orig_newton(*a, **k)
    _newton_stats.append(st)
    return x, st


sm.MinimizeNewton = _recording_newton


def fitter(N, alpha, ws, **kw):
    # the mock's longest baselines exceed Qmax for these small N; the bounds check is not the subject here
    return FrankFitter(RMAX, N, FixedGeometry(**MOCK_GEOMETRY), alpha=alpha, weights_smooth=ws, method="LogNormal",
                       I_scale=I_SCALE, store_iteration_diagnostics=True, verbose=False,
                       convergence_failure="ignore", check_qbounds=False, **kw)


def whole_fit(N, alpha, ws, mapping, perturb_seed=0):
    _newton_stats.clear()
    FF = fitter(N, alpha, ws)
    m = dict(mapping)
    if perturb_seed:
        rng = np.random.default_rng(perturb_seed)
        Mp = m["M"] * (1 + 1e-15 * rng.standard_normal(m["M"].shape))
        m["M"] = 0.5 * (Mp + Mp.T)
    t0 = time.perf_counter()
    sol = FF.fit_preprocessed(m)
    dt = time.perf_counter() - t0
    st = np.array(_newton_stats)
    hist = collections.Counter(st[:, 0].tolist())
    d = FF.iteration_diagnostics
    return dict(I=sol.I, s=sol._fit.MAP, p=sol.power_spectrum, niter=d["num_iterations"], t_fit=dt,
                diag_p=np.array(d["power_spectrum"][:6]), diag_s=np.array(d["MAP"][:6]),
                status_hist=np.array([hist.get(k, 0) for k in range(4)]),
                totals=np.array([len(st), st[:, 1].sum(), st[:, 2].sum(), st[:, 3].sum()]))


def case(name, N, n, fits):
    print("LogNormal N=%d, %d vis" % (N, n))
    u, v, V, w = mock_disc_visibilities(n, seed=5, noise_seed=6)
    FF = fitter(N, 1.05, 1e-4)
    mapping = FF.preprocess_visibilities(u, v, V, w)
    M, j = mapping["M"], mapping["j"]
    out = dict(N=N, n=n, seed=5, noise_seed=6, Rmax=RMAX, I_scale=I_SCALE, M=M, j=j, H0=mapping["null_likelihood"])
    D = FF._DHT
    # seeds of FrankFitter._fit (radial_fitters.py:744-763)
    fit = GaussianModel(D, M, j, np.ones(N), guess=np.ones(N))
    pI = np.max(D.transform(fit.MAP) ** 2) * (D.q / D.q[0]) ** -2
    fit = GaussianModel(D, M, j, pI)
    s0 = np.log(I_SCALE)
    s_guess = np.log(np.maximum(fit.MAP, 1e-3 * fit.MAP.max())) - s0
    p_seed = np.max(D.transform(s_guess) ** 2) * (D.q / D.q[0]) ** -4
    _newton_stats.clear()
    ln = LogNormalMAPModel(D, M, j, p_seed, guess=s_guess.copy(), s0=s0)
    filt = CriticalFilter(D, 1.3, 1e-35, 1e-2)
    out.update(seed_mu=fit.MAP, s_guess=s_guess, p_seed=p_seed, map_s=ln.MAP, map_Dinv=_hess(ln),
               map_cov_diag=np.diag(ln.covariance).copy(), map_stats=np.array(_newton_stats[-1]),
               map_p_updated=filt.update_power_spectrum(ln))
    print("   MAP solve exit (status, nstep, nfev, nhess) =", _newton_stats[-1])
    # the reference's OWN sensitivity of this solve to a 1e-15 relative perturbation of M: at N = 300 the faint outer
    # disc is loosely held by MinimizeNewton's tol = 1e-7 stop (1.6e-4 in s there), the profile to 1e-7 of its maximum
    rng = np.random.default_rng(1)
    Mp = M * (1 + 1e-15 * rng.standard_normal(M.shape))
    ln2 = LogNormalMAPModel(D, 0.5 * (Mp + Mp.T), j, p_seed, guess=s_guess.copy(), s0=s0)
    I1, I2 = np.exp(ln.MAP + s0), np.exp(ln2.MAP + s0)
    out.update(map_selfsens_s=np.max(np.abs(ln2.MAP - ln.MAP)), map_selfsens_I_relmax=np.max(np.abs(I2 - I1)) / I1.max(),
               map_selfsens_nstep=_newton_stats[-1][1])
    print("   self-sensitivity of the MAP solve: %.2e in s, %.2e of max I, nstep %d" % (
        out["map_selfsens_s"], out["map_selfsens_I_relmax"], out["map_selfsens_nstep"]))
    for tag, (alpha, ws) in fits.items():
        a = whole_fit(N, alpha, ws, mapping)
        b = whole_fit(N, alpha, ws, mapping, perturb_seed=1)
        out["alpha_" + tag], out["wsmooth_" + tag] = alpha, ws
        for k, val in a.items():
            out["%s_%s" % (k, tag)] = val
        out["selfsens_I_rel_" + tag] = np.max(np.abs(b["I"] / a["I"] - 1))
        out["selfsens_I_relmax_" + tag] = np.max(np.abs(b["I"] - a["I"])) / a["I"].max()
        out["selfsens_niter_" + tag] = b["niter"]
        print("   fit %s: alpha=%g ws=%g niter=%d (perturbed: %d)  %.1fs  exits=%s  self-sensitivity: %.2e elementwise, "
              "%.2e of max" % (tag, alpha, ws, a["niter"], b["niter"], a["t_fit"], a["status_hist"],
                               out["selfsens_I_rel_" + tag], out["selfsens_I_relmax_" + tag]))
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **out, **{"meta_" + k: v for k, v in META.items()})
    print("  wrote %s %.1f KB" % (name, os.path.getsize(path) / 1024))


def _n300_full_one(arg):
    perturb_seed, source = arg
    g = np.load(os.path.join(OUT, source))
    # a genuine mapping dict (hash, V, W, ... of the same basis and geometry) with the fixture's statistics put in
    u, v, V, w = mock_disc_visibilities(1000, seed=5, noise_seed=6)
    mapping = dict(fitter(300, 1.3, 1e-2).preprocess_visibilities(u, v, V, w))
    mapping.update(M=g["M"], j=g["j"], null_likelihood=float(g["H0"]))
    return whole_fit(300, 1.3, 1e-2, mapping, perturb_seed=perturb_seed)


def n300_full(source="fit_N300_1e6.npz", name="lognormal_N300_full.npz"):
    """BASELINE configs[2] at its basis size: the reference's whole method='LogNormal' fit (radial_fitters.py:737-832,
    statistical_models.py:1012-1160, minimizer.py:187-284) on the M, j of a Normal fixture (alpha = 1.3, w_smooth = 1e-2),
    and the same fit after a 1e-15 relative perturbation of M: the reference's own round-off spread, which the parity
    test uses as its scale.  source = fit_N300_1e7.npz is configs[2] itself (N = 300, 1e7 visibilities)."""
    import multiprocessing as mp
    print("LogNormal whole fit, N=300, M and j of %s, alpha=1.3 ws=1e-2 (two processes)" % source)
    g = np.load(os.path.join(OUT, source))
    import hashlib
    sha = hashlib.sha256(np.ascontiguousarray(g["M"]).tobytes() + np.ascontiguousarray(g["j"]).tobytes()).hexdigest()
    with mp.get_context("fork").Pool(2) as pool:
        a, b = pool.map(_n300_full_one, [(0, source), (1, source)])
    out = dict(N=300, Rmax=RMAX, I_scale=I_SCALE, alpha=1.3, wsmooth=1e-2, source=source, Mj_sha256=sha)
    for k, val in a.items():
        out[k] = val
    out.update(I_perturbed=b["I"], p_perturbed=b["p"], niter_perturbed=b["niter"], totals_perturbed=b["totals"],
               selfsens_I_rel=np.max(np.abs(b["I"] / a["I"] - 1)),
               selfsens_I_relmax=np.max(np.abs(b["I"] - a["I"])) / a["I"].max())
    print("   niter=%d (perturbed: %d)  %.0fs / %.0fs  exits=%s totals=%s  self-sensitivity: %.2e elementwise, %.2e of max"
          % (a["niter"], b["niter"], a["t_fit"], b["t_fit"], a["status_hist"], a["totals"], out["selfsens_I_rel"],
             out["selfsens_I_relmax"]))
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **out, **{"meta_" + k: v for k, v in META.items()})
    print("  wrote %s %.1f KB" % (path, os.path.getsize(path) / 1024))


def svd_seed():
    """method='LogNormal' when the Cholesky of a Normal SEED solve fails (radial_fitters.py:744-752 through
    statistical_models.py:747-755): eigenvalue 21 of a real N = 24 M flipped -- both seed solves go through the SVD
    pseudo-inverse, the LogNormal loop that follows is regular.  svd_seed_lognormal_N24.npz"""
    import warnings
    N, n, k = 24, 4000, 21
    u, v, V, w = mock_disc_visibilities(n, seed=61, noise_seed=62)
    ncalls = [0]
    orig = sm.scipy.linalg.svd

    def counting_svd(*a, **kw):
        ncalls[0] += 1
        return orig(*a, **kw)
    out = {}
    for tag, perturb in (("", 0), ("_perturbed", 1)):
        FF = fitter(N, 1.3, 1e-2, max_iter=40)
        m = dict(FF.preprocess_visibilities(u, v, V, w))
        lam, vec = np.linalg.eigh(m["M"])
        M2 = m["M"] - 2.0 * lam[k] * np.outer(vec[:, k], vec[:, k])
        M2 = 0.5 * (M2 + M2.T)
        if perturb:
            Mp = M2 * (1 + 1e-15 * np.random.default_rng(1).standard_normal(M2.shape))
            M2p = 0.5 * (Mp + Mp.T)
        m["M"] = M2p if perturb else M2
        ncalls[0] = 0
        sm.scipy.linalg.svd = counting_svd
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                sol = FF.fit_preprocessed(m)
        finally:
            sm.scipy.linalg.svd = orig
        d = FF.iteration_diagnostics
        print("   svd_seed%s: num_iterations=%d, svd calls=%d, min eig=%.3e" % (tag, d["num_iterations"], ncalls[0],
                                                                            np.linalg.eigvalsh(M2).min()))
        if not perturb:
            out.update(N=N, n=n, max_iter=40, alpha=1.3, wsmooth=1e-2, I_scale=I_SCALE, M=M2, j=m["j"],
                       H0=m["null_likelihood"], n_svd=ncalls[0], diag_p=np.array(d["power_spectrum"]),
                       diag_s=np.array(d["MAP"]))
        out["I" + tag], out["p" + tag], out["niter" + tag] = sol.I, sol.power_spectrum, d["num_iterations"]
    out["selfsens_I_relmax"] = np.max(np.abs(out["I_perturbed"] - out["I"])) / out["I"].max()
    print("   self-sensitivity %.2e of max" % out["selfsens_I_relmax"])
    path = os.path.join(OUT, "svd_seed_lognormal_N24.npz")
    np.savez_compressed(path, **out, **{"meta_" + k_: v_ for k_, v_ in META.items()})
    print("  wrote %s %.1f KB" % (path, os.path.getsize(path) / 1024))


def _hess(ln):
    """Hessian at the MAP: what LogNormalMAPModel._fit factorises (statistical_models.py:1147-1149)."""
    # cho_factor output holds the upper factor in the upper triangle (lower is untouched input garbage)
    U = np.triu(ln._Dchol[0])
    return U.T @ U


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "N300":
        # BASELINE configs[2]'s basis size: ONE LogNormalMAPModel seed solve + one update_power_spectrum (the blocked-LU
        # and 64-row-block solve geometry of the device kernel); no whole fit (hours in the reference)
        case("lognormal_N300.npz", 300, 200000, dict())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "N300_full":
        n300_full()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "svd_seed":
        svd_seed()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "N300_1e7":  # needs tests/golden/fit_N300_1e7.npz (tools/make_golden.py)
        n300_full("fit_N300_1e7.npz", "lognormal_N300_1e7.npz")
        return
    case("lognormal_N40.npz", 40, 5000, dict(a=(1.3, 1e-2)))
    case("lognormal_N80.npz", 80, 20000, dict(a=(1.05, 1e-4)))


if __name__ == "__main__":
    main()
