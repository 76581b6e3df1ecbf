"""sequential-line-search_amd -- MI355X-native GP regression + acquisition maximisation.

This module is a thin ctypes binding of the C ABI in include/sls_hip.h (libsls_hip.so, hand-written
gfx950 HIP kernels).  It exists for the tests and bench.py; the product host layer is the C++ classes
under include/sequential-line-search/ + host/.  There is NO CPU fallback here: if the shared library is
missing or no GPU is present every call raises.

Array convention = the reference's Eigen layout: X has shape (D, N), one data point per column.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLS_HIP_LIB") or os.path.join(_HERE, "libsls_hip.so")   # SLS_HIP_LIB: A/B runs of another build

KERNEL_SE, KERNEL_MATERN52 = 0, 1
ACQ_EI, ACQ_UCB = 0, 1
GP_K_Y, GP_K_Y_INV, GP_CHOL_L, GP_ALPHA, GP_MU_DATA = 0, 1, 2, 3, 4

_dp = C.POINTER(C.c_double)
_lib = None


class SlsError(RuntimeError):
    pass


class LbfgsOpts(C.Structure):
    """sls_lbfgs_opts (include/sls_hip.h).  struct_size is filled in here; the positional arguments are the option members."""
    _fields_ = [("struct_size", C.c_int), ("history", C.c_int), ("c1", C.c_double), ("shrink", C.c_double), ("gtol", C.c_double),
                ("max_backtracks", C.c_int), ("ftol_rel", C.c_double), ("xtol_rel", C.c_double)]

    def __init__(self, history=6, c1=1e-4, shrink=0.5, gtol=0.0, max_backtracks=20, ftol_rel=0.0, xtol_rel=0.0):
        super().__init__(C.sizeof(LbfgsOpts), history, c1, shrink, gtol, max_backtracks, ftol_rel, xtol_rel)


def lib():
    """Load libsls_hip.so (raises if it has not been built -- see __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlsError(f"{LIB_PATH} is missing: build it with `make -C sequential-line-search_amd/csrc` "
                           "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.sls_last_error.restype = C.c_char_p
    return _lib


EXPORTS = [
    "sls_last_error", "sls_version", "sls_ctx_create", "sls_ctx_destroy", "sls_ctx_set_stream", "sls_ctx_synchronize",
    "sls_ctx_set_candidate_chunk", "sls_gram", "sls_gram_cross", "sls_potrf", "sls_potrs", "sls_potri", "sls_gp_create",
    "sls_gp_destroy", "sls_gp_get_matrix", "sls_gp_get_summary", "sls_gp_predict", "sls_gp_predict_grad", "sls_acq_eval",
    "sls_lbfgs_default_opts", "sls_acq_maximize", "sls_acq_maximize_dev", "sls_gp_refit_dev", "sls_prof_enable",
    "sls_prof_reset", "sls_prof_get", "sls_nll_create", "sls_nll_destroy", "sls_nll_set_tolerances", "sls_nll_eval", "sls_gp_nll_grad", "sls_gp_nll_batch", "sls_multi_nll_create", "sls_multi_nll_destroy",
    "sls_multi_gp_nll_batch",
    "sls_pref_objective", "sls_pref_map_fit", "sls_gp_map_fit", "sls_gp_set_sigma_mode", "sls_acq_eval_pair", "sls_acq_maximize_pair", "sls_gp_append_point", "sls_acq_last_stats",
    "sls_multi_create", "sls_multi_destroy", "sls_multi_size", "sls_multi_exchange", "sls_multi_ctx", "sls_multi_gp_create", "sls_multi_gp_create_from",
    "sls_multi_gp_destroy", "sls_multi_gp_shard", "sls_multi_acq_maximize", "sls_multi_gp_predict", "sls_comm_unique_id", "sls_comm_create",
    "sls_comm_destroy", "sls_comm_allgather_best", "sls_device_trim_cache", "sls_tuning_reload", "sls_gp_generation",
]


def tuning_reload():
    """Re-read the SLS_* environment variables (the library parses them once per process)."""
    lib().sls_tuning_reload()


ERR_UNSUPPORTED = -5
SIGMA_EXPLICIT_INVERSE, SIGMA_CHOLESKY_SOLVE = 0, 1


class Unsupported(SlsError):
    """SLS_ERR_UNSUPPORTED: the problem is outside what the entry point runs on the device (use the general path)."""


def _ck(rc):
    if rc != 0:
        raise SlsError(f"libsls_hip error {rc}: {lib().sls_last_error().decode()}")


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        self._gps = []
        _ck(lib().sls_ctx_create(int(device), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            for ref in self._gps:          # handles must not outlive their context
                gp = ref()
                if gp is not None:
                    gp.close()
            self._gps = []
            lib().sls_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _ck(lib().sls_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        _ck(lib().sls_ctx_synchronize(self.h))

    def set_candidate_chunk(self, chunk):
        _ck(lib().sls_ctx_set_candidate_chunk(self.h, int(chunk)))

    def prof_enable(self, on=True):
        _ck(lib().sls_prof_enable(self.h, int(on)))

    def prof_reset(self):
        _ck(lib().sls_prof_reset(self.h))

    def prof_get(self, name):
        ms, n = C.c_double(), C.c_long()
        _ck(lib().sls_prof_get(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- free functions (src/regressor.cpp) ----
    def gram(self, X, theta, b, kernel):
        X, theta = _f(X), _f(theta)
        D, N = X.shape
        K = np.empty((N, N), order="F")
        _ck(lib().sls_gram(self.h, _p(X), D, N, _p(theta), C.c_double(b), int(kernel), _p(K)))
        return K

    def gram_cross(self, X, Xs, theta, kernel):
        X, Xs, theta = _f(X), _f(Xs), _f(theta)
        D, N = X.shape
        M = Xs.shape[1]
        Ks = np.empty((N, M), order="F")
        _ck(lib().sls_gram_cross(self.h, _p(X), D, N, _p(Xs), M, _p(theta), int(kernel), _p(Ks)))
        return Ks

    def potrf(self, A):
        A = _f(A).copy(order="F")
        _ck(lib().sls_potrf(self.h, _p(A), A.shape[0]))
        return A

    def potrs(self, L, B):
        L = _f(L)
        B = _f(B).copy(order="F")
        nrhs = 1 if B.ndim == 1 else B.shape[1]
        _ck(lib().sls_potrs(self.h, _p(L), L.shape[0], _p(B), nrhs))
        return B

    def potri(self, L):
        L = _f(L)
        out = np.empty_like(L, order="F")
        _ck(lib().sls_potri(self.h, _p(L), L.shape[0], _p(out)))
        return out


class GP:
    """Device-resident GP state (GaussianProcessRegressor with fixed hyper-parameters / PreferenceRegressor post-MAP)."""

    def __init__(self, ctx, X, y, theta, b, kernel=KERNEL_MATERN52):
        self.ctx = ctx
        X, y, theta = _f(X), _f(y), _f(theta)
        self.D, self.N = X.shape
        assert y.shape == (self.N,) and theta.shape == (self.D + 1,)
        self.h = C.c_void_p()
        _ck(lib().sls_gp_create(ctx.h, _p(X), self.D, self.N, _p(y), _p(theta), C.c_double(b), int(kernel), C.byref(self.h)))
        ctx._gps.append(weakref.ref(self))

    def close(self):
        if getattr(self, "h", None):
            lib().sls_gp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sigma_mode(self, mode):
        """SIGMA_EXPLICIT_INVERSE (GaussianProcessRegressor, default) or SIGMA_CHOLESKY_SOLVE (PreferenceRegressor)."""
        _ck(lib().sls_gp_set_sigma_mode(self.h, int(mode)))

    def generation(self):
        """A number that changes whenever the predictor behind the handle changes (fit, refit, appended point, sigma mode)."""
        g = C.c_long(0)
        _ck(lib().sls_gp_generation(self.h, C.byref(g)))
        return int(g.value)

    def append_point(self, x, y):
        x = _f(x)
        assert x.shape == (self.D,)
        _ck(lib().sls_gp_append_point(self.h, _p(x), C.c_double(y)))
        self.N += 1

    def matrix(self, what):
        out = np.empty((self.N, self.N), order="F") if what in (GP_K_Y, GP_K_Y_INV, GP_CHOL_L) else np.empty(self.N)
        _ck(lib().sls_gp_get_matrix(self.h, int(what), _p(out)))
        return out

    def summary(self):
        bi, mb, ld = C.c_int(), C.c_double(), C.c_double()
        _ck(lib().sls_gp_get_summary(self.h, C.byref(bi), C.byref(mb), C.byref(ld)))
        return dict(best_index=bi.value, mu_best=mb.value, logdet=ld.value)

    def predict(self, Xs):
        Xs = _f(Xs)
        M = Xs.shape[1]
        mu, sg = np.empty(M), np.empty(M)
        _ck(lib().sls_gp_predict(self.h, _p(Xs), M, _p(mu), _p(sg)))
        return mu, sg

    def predict_grad(self, Xs):
        Xs = _f(Xs)
        M = Xs.shape[1]
        dm, ds = np.empty((self.D, M), order="F"), np.empty((self.D, M), order="F")
        _ck(lib().sls_gp_predict_grad(self.h, _p(Xs), M, _p(dm), _p(ds)))
        return dm, ds

    def acq_eval(self, Xs, acq=ACQ_EI, ucb_h=1.0, want_grad=True):
        Xs = _f(Xs)
        M = Xs.shape[1]
        val = np.empty(M)
        grad = np.empty((self.D, M), order="F") if want_grad else None
        _ck(lib().sls_acq_eval(self.h, int(acq), C.c_double(ucb_h), _p(Xs), M, _p(val), _p(grad) if want_grad else None))
        return (val, grad) if want_grad else val

    def acq_maximize(self, starts, n_local, acq=ACQ_EI, ucb_h=1.0, offset=0, want_all=True, opts=None):
        starts = _f(starts)
        S = starts.shape[1]
        x, val, idx = np.empty(self.D), C.c_double(), C.c_long()
        xs = np.empty((self.D, S), order="F") if want_all else None
        ys = np.empty(S) if want_all else None
        _ck(lib().sls_acq_maximize(self.h, int(acq), C.c_double(ucb_h), _p(starts), S, int(n_local),
                                   C.byref(opts) if opts is not None else None, C.c_long(offset), _p(x), C.byref(val),
                                   C.byref(idx), _p(xs) if want_all else None, _p(ys) if want_all else None))
        return dict(index=idx.value, x=x, value=val.value, x_stars=xs, y_stars=ys)

    def acq_eval_pair(self, sigma_gp, Xs, acq=ACQ_EI, ucb_h=1.0, want_grad=True):
        """objective_for_multiple_points: mean (and mu+) from this handle, deviation from `sigma_gp`."""
        Xs = _f(Xs)
        M = Xs.shape[1]
        val = np.empty(M)
        grad = np.empty((self.D, M), order="F") if want_grad else None
        _ck(lib().sls_acq_eval_pair(self.h, sigma_gp.h, int(acq), C.c_double(ucb_h), _p(Xs), M, _p(val),
                                    _p(grad) if want_grad else None))
        return (val, grad) if want_grad else val

    def acq_maximize_pair(self, sigma_gp, starts, n_local, acq=ACQ_EI, ucb_h=1.0):
        starts = _f(starts)
        x, val, idx = np.empty(self.D), C.c_double(), C.c_long()
        _ck(lib().sls_acq_maximize_pair(self.h, sigma_gp.h, int(acq), C.c_double(ucb_h), _p(starts), starts.shape[1], int(n_local),
                                        None, _p(x), C.byref(val), C.byref(idx)))
        return dict(index=idx.value, x=x, value=val.value)

    def acq_maximize_dev(self, starts_dev_ptr, S, n_local, acq=ACQ_EI, ucb_h=1.0, offset=0, opts=None):
        x, val, idx = np.empty(self.D), C.c_double(), C.c_long()
        _ck(lib().sls_acq_maximize_dev(self.h, int(acq), C.c_double(ucb_h), C.c_void_p(starts_dev_ptr), int(S), int(n_local),
                                       C.byref(opts) if opts is not None else None, C.c_long(offset), _p(x), C.byref(val),
                                       C.byref(idx)))
        return dict(index=idx.value, x=x, value=val.value)

    def last_stats(self):
        """Evaluation counts of the last acq_maximize* call (sls_acq_last_stats)."""
        issued, cap, rounds, live = C.c_long(), C.c_long(), C.c_int(), C.c_int()
        _ck(lib().sls_acq_last_stats(self.h, C.byref(issued), C.byref(cap), C.byref(rounds), C.byref(live)))
        return dict(evals_issued=issued.value, evals_cap=cap.value, rounds=rounds.value, live_at_end=live.value)

    def refit_dev(self, X_dev_ptr, y_dev_ptr):
        _ck(lib().sls_gp_refit_dev(self.h, C.c_void_p(X_dev_ptr), C.c_void_p(y_dev_ptr)))


class PrefCfg(C.Structure):
    _fields_ = [("use_map_hyperparams", C.c_int), ("default_a", C.c_double), ("default_r", C.c_double),
                ("default_b", C.c_double), ("prior_var", C.c_double), ("btl_scale", C.c_double), ("noiseless", C.c_int)]


class Nll:
    """Device state for the MAP objectives on a fixed design matrix (sls_nll_* / sls_gp_nll_grad / sls_pref_objective)."""

    def __init__(self, ctx, X, kernel=KERNEL_MATERN52):
        self.ctx = ctx
        X = _f(X)
        self.D, self.N = X.shape
        self.h = C.c_void_p()
        _ck(lib().sls_nll_create(ctx.h, _p(X), self.D, self.N, int(kernel), C.byref(self.h)))
        ctx._gps.append(weakref.ref(self))

    def close(self):
        if getattr(self, "h", None):
            lib().sls_nll_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self, y, theta, b, want_grad=True):
        y, theta = _f(y), _f(theta)
        quad, logdet, gb = C.c_double(), C.c_double(), C.c_double()
        alpha = np.empty(self.N)
        gth = np.empty(self.D + 1)
        _ck(lib().sls_nll_eval(self.h, _p(y), _p(theta), C.c_double(b), C.byref(quad), C.byref(logdet), _p(alpha),
                               _p(gth) if want_grad else None, C.byref(gb) if want_grad else None))
        return dict(quad=quad.value, logdet=logdet.value, alpha=alpha, grad_theta=gth if want_grad else None,
                    grad_b=gb.value if want_grad else None)

    def gp_objective(self, y, x, want_grad=True):
        y, x = _f(y), _f(x)
        val = C.c_double()
        g = np.empty(self.D + 2) if want_grad else None
        _ck(lib().sls_gp_nll_grad(self.h, _p(y), _p(x), C.byref(val), _p(g) if want_grad else None))
        return (val.value, g) if want_grad else val.value

    def gp_objective_batch(self, y, xs):
        """Values of the GP MAP objective at the rows of xs (B x (D + 2): a, b, r_1..r_D), one device call (sls_gp_nll_batch)."""
        y = _f(y)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        assert xs.ndim == 2 and xs.shape[1] == self.D + 2
        out = np.empty(xs.shape[0])
        _ck(lib().sls_gp_nll_batch(self.h, _p(y), _p(xs), xs.shape[0], _p(out)))
        return out

    def pref_objective(self, prefs, x, use_map=False, a=0.5, r=0.5, b=0.005, prior_var=0.25, btl_scale=0.01,
                       noiseless=False, want_grad=True):
        x = _f(x)
        flat = np.array([i for p in prefs for i in p], dtype=np.uint32)
        offs = np.zeros(len(prefs) + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(p) for p in prefs])
        cfg = PrefCfg(int(use_map), a, r, b, prior_var, btl_scale, int(noiseless))
        val = C.c_double()
        g = np.empty(len(x)) if want_grad else None
        _ck(lib().sls_pref_objective(self.h, flat.ctypes.data_as(C.POINTER(C.c_uint)), offs.ctypes.data_as(C.POINTER(C.c_int)),
                                     len(prefs), _p(x), C.byref(cfg), C.byref(val), _p(g) if want_grad else None))
        return (val.value, g) if want_grad else val.value


    def set_tolerances(self, ftol_rel, xtol_rel):
        """NLopt's relative stopping tests for this handle's MAP fits (sls_nll_set_tolerances); 0 = off (the default)."""
        _ck(lib().sls_nll_set_tolerances(self.h, C.c_double(ftol_rel), C.c_double(xtol_rel)))

    def pref_map_fit(self, prefs, z0, lower, upper, max_evals, evals_per_launch=0, use_map=False, a=0.5, r=0.5, b=0.005,
                     prior_var=0.25, btl_scale=0.01, noiseless=False):
        """PreferenceRegressor::PerformMapEstimation on the device (sls_pref_map_fit): z = (y [, log a, log b, log r..]).
        Returns dict(z, value, evals); raises Unsupported outside the device-resident limits."""
        z0, lower, upper = _f(z0), _f(lower), _f(upper)
        flat = np.array([i for p in prefs for i in p], dtype=np.uint32)
        offs = np.zeros(len(prefs) + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(p) for p in prefs])
        cfg = PrefCfg(int(use_map), a, r, b, prior_var, btl_scale, int(noiseless))
        z = np.empty(len(z0))
        val, ev = C.c_double(), C.c_int()
        rc = lib().sls_pref_map_fit(self.h, flat.ctypes.data_as(C.POINTER(C.c_uint)), offs.ctypes.data_as(C.POINTER(C.c_int)),
                                    len(prefs), C.byref(cfg), _p(z0), _p(lower), _p(upper), int(max_evals), int(evals_per_launch),
                                    _p(z), C.byref(val), C.byref(ev))
        if rc == ERR_UNSUPPORTED:
            raise Unsupported(lib().sls_last_error().decode())
        _ck(rc)
        return dict(z=z, value=val.value, evals=ev.value)

    def gp_map_fit(self, y, z0, lower, upper, max_evals, evals_per_launch=0):
        """Local phase of GaussianProcessRegressor::PerformMapEstimation on the device (sls_gp_map_fit): z = log (a, b, r..)."""
        y, z0, lower, upper = _f(y), _f(z0), _f(lower), _f(upper)
        z = np.empty(len(z0))
        val, ev = C.c_double(), C.c_int()
        rc = lib().sls_gp_map_fit(self.h, _p(y), _p(z0), _p(lower), _p(upper), int(max_evals), int(evals_per_launch), _p(z),
                                  C.byref(val), C.byref(ev))
        if rc == ERR_UNSUPPORTED:
            raise Unsupported(lib().sls_last_error().decode())
        _ck(rc)
        return dict(z=z, value=val.value, evals=ev.value)


class Multi:
    """One process driving several GPUs (sls_multi_*): replicated fit, starts sharded, one ncclAllGather."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self.h = C.c_void_p()
        _ck(lib().sls_multi_create(arr, len(self.devices), C.byref(self.h)))
        lib().sls_multi_exchange.restype = C.c_char_p
        self._gps = []

    @property
    def exchange(self):
        return lib().sls_multi_exchange(self.h).decode()

    def close(self):
        if getattr(self, "h", None):
            for ref in self._gps:
                gp = ref()
                if gp is not None:
                    gp.close()
            self._gps = []
            lib().sls_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGP:
    def __init__(self, multi, X, y, theta, b, kernel=KERNEL_MATERN52):
        X, y, theta = _f(X), _f(y), _f(theta)
        self.D, self.N = X.shape
        self.h = C.c_void_p()
        _ck(lib().sls_multi_gp_create(multi.h, _p(X), self.D, self.N, _p(y), _p(theta), C.c_double(b), int(kernel), C.byref(self.h)))
        multi._gps.append(weakref.ref(self))

    def close(self):
        if getattr(self, "h", None):
            lib().sls_multi_gp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def acq_maximize(self, starts, n_local, acq=ACQ_EI, ucb_h=1.0, opts=None):
        starts = _f(starts)
        x, val, idx, issued = np.empty(self.D), C.c_double(), C.c_long(), C.c_long()
        _ck(lib().sls_multi_acq_maximize(self.h, int(acq), C.c_double(ucb_h), _p(starts), starts.shape[1], int(n_local),
                                         C.byref(opts) if opts is not None else None, _p(x), C.byref(val), C.byref(idx),
                                         C.byref(issued)))
        return dict(index=idx.value, x=x, value=val.value, evals_issued=issued.value)

    def predict(self, Xs):
        """PredictMu / PredictSigma at the M columns of Xs, the columns sharded over the devices (sls_multi_gp_predict)."""
        Xs = _f(Xs)
        M = Xs.shape[1]
        mu, sigma = np.empty(M), np.empty(M)
        _ck(lib().sls_multi_gp_predict(self.h, _p(Xs), M, _p(mu), _p(sigma)))
        return mu, sigma


class MultiNll:
    """GP MAP objective over the devices of a Multi: the points of a batch dealt round-robin (sls_multi_gp_nll_batch)."""

    def __init__(self, multi, X, kernel=KERNEL_MATERN52):
        X = _f(X)
        self.D, self.N = X.shape
        self.h = C.c_void_p()
        _ck(lib().sls_multi_nll_create(multi.h, _p(X), self.D, self.N, int(kernel), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().sls_multi_nll_destroy(self.h)
            self.h = None

    def gp_objective_batch(self, y, xs):
        y = _f(y)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        out = np.empty(xs.shape[0])
        _ck(lib().sls_multi_gp_nll_batch(self.h, _p(y), _p(xs), xs.shape[0], _p(out)))
        return out


class Comm:
    """One process per GPU: RCCL communicator inside the library (sls_comm_*).  `unique_id()` on rank 0, distribute the 128
    bytes, then Comm(ctx, id, rank, world) on every rank."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _ck(lib().sls_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, uid, rank, world):
        assert len(uid) == 128
        self.h = C.c_void_p()
        _ck(lib().sls_comm_create(ctx.h, uid, int(rank), int(world), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().sls_comm_destroy(self.h)
            self.h = None

    def allgather_best(self, value, index, x):
        x = _f(x)
        xo, vo, io = np.empty_like(x), C.c_double(), C.c_long()
        _ck(lib().sls_comm_allgather_best(self.h, C.c_double(value), C.c_long(index), _p(x), len(x), C.byref(vo), C.byref(io), _p(xo)))
        return vo.value, io.value, xo


def merge_rank_results(results):
    """Global argmax over per-rank (value, global_index, x) triples: highest value, ties -> lowest global index
    (Eigen maxCoeff 'first maximum', src/acquisition-function.cpp:146-153).  `results` is an iterable of
    (value, index, x) with x a length-D array."""
    best = None
    for v, i, x in results:
        if best is None or v > best[0] or (v == best[0] and i < best[1]):
            best = (float(v), int(i), np.asarray(x, dtype=np.float64))
    return best


def shard_range(n_starts, rank, world):
    """Contiguous slice [lo, hi) of the global start set owned by `rank` (the last ranks get the remainder-free part)."""
    base, rem = divmod(int(n_starts), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def exchange_best(value, index, x, device=None, group=None):
    """The single collective of a multi-GPU maximisation: all-gather (value, global index, x[D]) over the ranks of a
    torch.distributed group (RCCL on GPUs, gloo on CPU) and take the first maximum on every rank.
    Returns (value, index, x) -- identical on all ranks.  The global index travels as float64 (exact below 2**53)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    x = np.asarray(x, dtype=np.float64)
    mine = torch.from_numpy(np.concatenate([[float(value), float(index)], x]))
    if device is not None:
        mine = mine.to(device)
    gathered = torch.empty(world * mine.numel(), dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    g = gathered.cpu().numpy().reshape(world, mine.numel())
    return merge_rank_results((row[0], int(row[1]), row[2:]) for row in g)
