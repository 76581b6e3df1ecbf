"""ctypes binding of the CPU ORACLE (oracle/sls_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The shipped package never imports this module.

Arrays follow the reference's Eigen layout: X is (D, N) column-major, i.e. a
numpy array of shape (D, N) with order='F' (one data point per column).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsls_oracle.so")

KERNEL_SE, KERNEL_MATERN52 = 0, 1
ACQ_EI, ACQ_UCB = 0, 1
REG_GPR, REG_PREF = 0, 1


def build(force=False):
    src = os.path.join(_HERE, "sls_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsls_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.slso_kernel.restype = C.c_double
        _lib.slso_predict_mu.restype = C.c_double
        _lib.slso_predict_sigma.restype = C.c_double
        _lib.slso_acq_value_as_written.restype = C.c_double
        _lib.slso_logdet_from_chol.restype = C.c_double
        _lib.slso_log_lognormal.restype = C.c_double
        _lib.slso_log_lognormal_derivative.restype = C.c_double
        _lib.slso_norm_pdf.restype = C.c_double
        _lib.slso_norm_cdf.restype = C.c_double
        _lib.slso_btl.restype = C.c_double
        _lib.slso_gp_map_objective.restype = C.c_double
        _lib.slso_pref_objective.restype = C.c_double
        _lib.slso_regressor_create.restype = C.c_void_p
        for f in ("slso_log_lognormal", "slso_log_lognormal_derivative"):
            getattr(_lib, f).argtypes = [C.c_double] * 3
        _lib.slso_norm_pdf.argtypes = [C.c_double]
        _lib.slso_norm_cdf.argtypes = [C.c_double]
    return _lib


def _f(a):
    """Fortran-contiguous float64 copy."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


def fill_uniform(n, seed):
    out = np.empty(int(n), dtype=np.float64)
    lib().slso_fill_uniform(_p(out), C.c_long(int(n)), C.c_ulonglong(seed))
    return out


def fill_normal(n, seed):
    out = np.empty(int(n), dtype=np.float64)
    lib().slso_fill_normal(_p(out), C.c_long(int(n)), C.c_ulonglong(seed))
    return out


def kernel(ktype, xa, xb, theta):
    xa, xb, theta = _f(xa), _f(xb), _f(theta)
    return lib().slso_kernel(ktype, _p(xa), _p(xb), _p(theta), len(xa))


def kernel_theta_derivative(ktype, xa, xb, theta):
    xa, xb, theta = _f(xa), _f(xb), _f(theta)
    out = np.empty(len(xa) + 1)
    lib().slso_kernel_theta_derivative(ktype, _p(xa), _p(xb), _p(theta), len(xa), _p(out))
    return out


def kernel_first_arg_derivative(ktype, xa, xb, theta):
    xa, xb, theta = _f(xa), _f(xb), _f(theta)
    out = np.empty(len(xa))
    lib().slso_kernel_first_arg_derivative(ktype, _p(xa), _p(xb), _p(theta), len(xa), _p(out))
    return out


def calc_large_ky(ktype, X, theta, b):
    X, theta = _f(X), _f(theta)
    D, N = X.shape
    K = np.empty((N, N), order="F")
    lib().slso_calc_large_ky(ktype, _p(X), D, N, _p(theta), C.c_double(b), _p(K))
    return K


def calc_small_k(ktype, x, X, theta):
    x, X, theta = _f(x), _f(X), _f(theta)
    D, N = X.shape
    k = np.empty(N)
    lib().slso_calc_small_k(ktype, _p(x), _p(X), D, N, _p(theta), _p(k))
    return k


def cholesky(A):
    A = _f(A).copy(order="F")
    info = lib().slso_cholesky(_p(A), A.shape[0])
    return A, info


def chol_solve(L, B):
    L = _f(L)
    B = _f(B).copy(order="F")
    nrhs = 1 if B.ndim == 1 else B.shape[1]
    lib().slso_chol_solve(_p(L), L.shape[0], _p(B), nrhs)
    return B


def lu_inverse(A):
    A = _f(A)
    out = np.empty_like(A, order="F")
    info = lib().slso_lu_inverse(_p(A), A.shape[0], _p(out))
    return out, info


def spd_inverse_from_chol(L):
    L = _f(L)
    out = np.empty_like(L, order="F")
    lib().slso_spd_inverse_from_chol(_p(L), L.shape[0], _p(out))
    return out


def logdet_from_chol(L):
    L = _f(L)
    return lib().slso_logdet_from_chol(_p(L), L.shape[0])


def btl(f, scale):
    f = _f(f)
    return lib().slso_btl(_p(f), len(f), C.c_double(scale))


def btl_derivative(f, scale):
    f = _f(f)
    d = np.empty(len(f))
    lib().slso_btl_derivative(_p(f), len(f), C.c_double(scale), _p(d))
    return d


class Regressor:
    """Oracle regressor: GaussianProcessRegressor (fixed hyper-parameters) or the
    predictive state of PreferenceRegressor."""

    def __init__(self, X, y, theta, b, kernel=KERNEL_MATERN52, reg_type=REG_GPR):
        self.X, self.y, self.theta = _f(X), _f(y), _f(theta)
        self.D, self.N = self.X.shape
        self.b, self.kernel, self.reg_type = float(b), kernel, reg_type
        self.h = C.c_void_p(lib().slso_regressor_create(reg_type, kernel, _p(self.X), self.D, self.N, _p(self.y),
                                                        _p(self.theta), C.c_double(self.b)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().slso_regressor_free(self.h)
            self.h = None

    # as written (reference call structure)
    def predict_mu(self, x):
        x = _f(x)
        return lib().slso_predict_mu(self.h, _p(x))

    def predict_sigma(self, x):
        x = _f(x)
        return lib().slso_predict_sigma(self.h, _p(x))

    def predict_mu_derivative(self, x):
        x = _f(x)
        g = np.empty(self.D)
        lib().slso_predict_mu_derivative(self.h, _p(x), _p(g))
        return g

    def predict_sigma_derivative(self, x):
        x = _f(x)
        g = np.empty(self.D)
        lib().slso_predict_sigma_derivative(self.h, _p(x), _p(g))
        return g

    def predict_maximum_point_from_data(self):
        xb = np.empty(self.D)
        idx = lib().slso_predict_maximum_point_from_data(self.h, _p(xb))
        return idx, xb

    def best_index(self):
        """Hoisted PredictMaximumPointFromData (arg max_i y_i - b alpha_i), O(1): for sizes where the as-written N x PredictMu
        loop (O(N^3)) is out of reach."""
        return int(lib().slso_regressor_best_index(self.h))

    def mu_best(self):
        lib().slso_regressor_mu_best.restype = C.c_double
        return float(lib().slso_regressor_mu_best(self.h))

    def acq_value_as_written(self, x, acq=ACQ_EI, ucb_h=1.0):
        x = _f(x)
        return lib().slso_acq_value_as_written(self.h, _p(x), acq, C.c_double(ucb_h))

    def acq_derivative_as_written(self, x, acq=ACQ_EI, ucb_h=1.0):
        x = _f(x)
        g = np.empty(self.D)
        lib().slso_acq_derivative_as_written(self.h, _p(x), acq, C.c_double(ucb_h), _p(g))
        return g

    # hoisted, batched
    def predict_batch(self, Xs):
        Xs = _f(Xs)
        M = Xs.shape[1]
        mu, sg = np.empty(M), np.empty(M)
        lib().slso_predict_batch(self.h, _p(Xs), M, _p(mu), _p(sg))
        return mu, sg

    def predict_grad_batch(self, Xs):
        Xs = _f(Xs)
        M = Xs.shape[1]
        dm, ds = np.empty((self.D, M), order="F"), np.empty((self.D, M), order="F")
        lib().slso_predict_grad_batch(self.h, _p(Xs), M, _p(dm), _p(ds))
        return dm, ds

    def acq_eval_batch(self, Xs, acq=ACQ_EI, ucb_h=1.0, want_grad=True):
        Xs = _f(Xs)
        M = Xs.shape[1]
        val = np.empty(M)
        grad = np.empty((self.D, M), order="F") if want_grad else None
        lib().slso_acq_eval_batch(self.h, _p(Xs), M, acq, C.c_double(ucb_h), _p(val), _p(grad) if want_grad else None)
        return (val, grad) if want_grad else val

    def acq_maximize(self, starts, n_local, acq=ACQ_EI, ucb_h=1.0, n_threads=0, diag=False, ftol_rel=0.0, xtol_rel=0.0):
        """diag=True adds 'armijo_margin' / 'armijo_eval' (per start: the closest any Armijo test came to its threshold,
        relative, and the evaluation number) -- see slso_acq_maximize_diag."""
        starts = _f(starts)
        S = starts.shape[1]
        x_out, val = np.empty(self.D), C.c_double()
        x_stars, y_stars = np.empty((self.D, S), order="F"), np.empty(S)
        opts = None
        if ftol_rel or xtol_rel:
            opts = LbfgsOpts()
            lib().slso_lbfgs_default_opts(C.byref(opts))
            opts.ftol_rel, opts.xtol_rel = float(ftol_rel), float(xtol_rel)
            opts = C.byref(opts)
        if not diag:
            idx = lib().slso_acq_maximize(self.h, acq, C.c_double(ucb_h), _p(starts), S, int(n_local), opts, _p(x_out),
                                          C.byref(val), _p(x_stars), _p(y_stars), int(n_threads))
            return dict(index=idx, x=x_out, value=val.value, x_stars=x_stars, y_stars=y_stars)
        margin, ev = np.empty(S), np.empty(S, dtype=np.int32)
        idx = lib().slso_acq_maximize_diag(self.h, acq, C.c_double(ucb_h), _p(starts), S, int(n_local), opts, _p(x_out),
                                           C.byref(val), _p(x_stars), _p(y_stars), int(n_threads), _p(margin),
                                           ev.ctypes.data_as(C.POINTER(C.c_int)))
        return dict(index=idx, x=x_out, value=val.value, x_stars=x_stars, y_stars=y_stars, armijo_margin=margin, armijo_eval=ev)


def gp_map_objective(ktype, X, y, x, want_grad=True, as_written=False):
    X, y, x = _f(X), _f(y), _f(x)
    D, N = X.shape
    g = np.empty(D + 2) if want_grad else None
    v = lib().slso_gp_map_objective(ktype, _p(X), D, N, _p(y), _p(x), _p(g) if want_grad else None, int(as_written))
    return (v, g) if want_grad else v


class LbfgsOpts(C.Structure):   # slso_lbfgs_opts
    _fields_ = [("history", C.c_int), ("c1", C.c_double), ("shrink", C.c_double), ("gtol", C.c_double), ("max_backtracks", C.c_int),
                ("ftol_rel", C.c_double), ("xtol_rel", C.c_double)]


class PrefCfg(C.Structure):
    _fields_ = [("use_map_hyperparams", C.c_int), ("default_a", C.c_double), ("default_r", C.c_double),
                ("default_b", C.c_double), ("prior_var", C.c_double), ("btl_scale", C.c_double), ("noiseless", C.c_int)]


def pref_objective(ktype, X, prefs, x, use_map=False, a=0.5, r=0.5, b=0.005, prior_var=0.25, btl_scale=0.01,
                   noiseless=False, want_grad=True):
    X, x = _f(X), _f(x)
    D, M = X.shape
    flat = np.array([i for p in prefs for i in p], dtype=np.uint32)
    offs = np.zeros(len(prefs) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(p) for p in prefs])
    cfg = PrefCfg(int(use_map), a, r, b, prior_var, btl_scale, int(noiseless))
    g = np.empty(len(x)) if want_grad else None
    v = lib().slso_pref_objective(ktype, _p(X), D, M, flat.ctypes.data_as(C.POINTER(C.c_uint)),
                                  offs.ctypes.data_as(C.POINTER(C.c_int)), len(prefs), _p(x), C.byref(cfg),
                                  _p(g) if want_grad else None)
    return (v, g) if want_grad else v
