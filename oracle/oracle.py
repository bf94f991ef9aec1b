"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product path never
does.  PARITY UNPINNED (see wavenet_oracle.h): the reference cannot be built
in this image and ships no fixtures.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_dp = C.POINTER(C.c_double)


def build():
    """Compile liboracle.so with the Makefile next to this file."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    sig = {
        "orc_highpass_filter": (None, [_dp, C.c_int, _dp]),
        "orc_operator_by_rows": (None, [_dp, C.c_int, C.c_uint, _dp]),
        "orc_operator_by_indices": (None, [_dp, C.c_int, C.c_uint, _dp]),
        "orc_operator_construct": (C.c_int, [_dp, C.c_int, C.c_uint, _dp]),
        "orc_sparse_term": (C.c_double, [_dp, C.c_int]),
        "orc_sparse_term_deriv": (None, [_dp, C.c_int, _dp]),
        "orc_reg_term": (C.c_double, [_dp, C.c_int, C.c_int]),
        "orc_reg_term_deriv": (None, [_dp, C.c_int, C.c_int, _dp]),
        "orc_wavenet_new": (vp, []),
        "orc_wavenet_free": (None, [vp]),
        "orc_set_filter": (C.c_int, [vp, _dp, C.c_int]),
        "orc_set_lambda": (None, [vp, C.c_double]),
        "orc_set_alpha": (None, [vp, C.c_double]),
        "orc_set_inertia": (None, [vp, C.c_double]),
        "orc_set_inertia_time_scale": (None, [vp, C.c_double]),
        "orc_set_batch_size": (None, [vp, C.c_uint]),
        "orc_do_wavelet": (None, [vp, C.c_int]),
        "orc_clear": (None, [vp]),
        "orc_num_coeffs": (C.c_int, [vp]),
        "orc_lambda": (C.c_double, [vp]),
        "orc_alpha": (C.c_double, [vp]),
        "orc_batch_size": (C.c_uint, [vp]),
        "orc_get_filter": (None, [vp, _dp]),
        "orc_get_momentum": (None, [vp, _dp]),
        "orc_cost_log_size": (C.c_int, [vp]),
        "orc_get_cost_log": (None, [vp, _dp]),
        "orc_cost_log_pop_back": (None, [vp]),
        "orc_filter_log_size": (C.c_int, [vp]),
        "orc_get_filter_log": (None, [vp, _dp]),
        "orc_batch_queue_size": (C.c_int, [vp]),
        "orc_forward_1d": (None, [vp, _dp, C.c_int, _dp, _dp]),
        "orc_coeffs_from_activations_1d": (None, [_dp, _dp, C.c_int, _dp]),
        "orc_inverse_1d": (None, [vp, _dp, C.c_int, _dp]),
        "orc_backprop_1d": (None, [vp, _dp, C.c_int, _dp, _dp, _dp, _dp]),
        "orc_forward_2d": (None, [vp, _dp, C.c_int, C.c_int, _dp, _dp, _dp]),
        "orc_inverse_2d": (None, [vp, _dp, C.c_int, C.c_int, _dp]),
        "orc_backprop_2d": (None, [vp, _dp, C.c_int, C.c_int, _dp, _dp, _dp]),
        "orc_cost": (C.c_double, [vp, _dp, C.c_int]),
        "orc_train": (C.c_int, [vp, _dp, C.c_int, C.c_int]),
        "orc_flush_batch_queue": (None, [vp]),
        "orc_coach_run": (C.c_long, [vp, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_uint,
                                     C.c_uint, C.c_double]),
        "orb_forward_1d": (None, [_dp, C.c_int, _dp, C.c_int, _dp, _dp]),
        "orb_inverse_1d": (None, [_dp, C.c_int, _dp, C.c_int, _dp]),
        "orb_backprop_1d": (None, [_dp, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp]),
        "orb_forward_2d": (None, [_dp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
        "orb_inverse_2d": (None, [_dp, C.c_int, _dp, C.c_int, C.c_int, _dp]),
        "orb_backprop_2d": (None, [_dp, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp]),
        "orb_train_batch": (C.c_int, [_dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def _c(a):
    """Contiguous float64 copy (column-major data is passed flat)."""
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def _flat_colmajor(X):
    """R x C numpy matrix -> flat column-major vector."""
    X = np.asarray(X, dtype=np.float64)
    return np.ascontiguousarray(X.reshape(-1, order="F"))


def _mat_colmajor(v, R, C_):
    return np.asarray(v).reshape((R, C_), order="F").copy()


# --- operators ---------------------------------------------------------------

def highpass_filter(a):
    a = _c(a)
    b = np.empty_like(a)
    lib.orc_highpass_filter(_p(a), len(a), _p(b))
    return b


def operator(filt, size, how="construct"):
    """Dense 2^size x 2^(size+1) matrix for the taps `filt` (used as given)."""
    f = _c(filt)
    nr, nc = 1 << size, 2 << size
    M = np.empty(nr * nc)
    fn = {"rows": lib.orc_operator_by_rows, "indices": lib.orc_operator_by_indices,
          "construct": lib.orc_operator_construct}[how]
    fn(_p(f), len(f), size, _p(M))
    return M.reshape((nr, nc), order="F").copy()


def lowpass_operator(a, size):
    return operator(a, size)


def highpass_operator(a, size):
    return operator(highpass_filter(a), size)


# --- cost functions ----------------------------------------------------------

def sparse_term(c):
    c = _flat_colmajor(c)
    return lib.orc_sparse_term(_p(c), c.size)


def sparse_term_deriv(c):
    shape = np.shape(c)
    v = _flat_colmajor(c)
    g = np.empty_like(v)
    lib.orc_sparse_term_deriv(_p(v), v.size, _p(g))
    return g.reshape(shape, order="F").copy()


def reg_term(a, wavelet=True):
    a = _c(a)
    return lib.orc_reg_term(_p(a), len(a), int(wavelet))


def reg_term_deriv(a, wavelet=True):
    a = _c(a)
    g = np.empty_like(a)
    lib.orc_reg_term_deriv(_p(a), len(a), int(wavelet), _p(g))
    return g


# --- Wavenet -----------------------------------------------------------------

class Wavenet:
    """Dense restatement of wavenet::Wavenet (include/Wavenet/Wavenet.h)."""

    def __init__(self, lam=10.0, alpha=0.001):
        self._h = lib.orc_wavenet_new()
        self.set_lambda(lam)
        self.set_alpha(alpha)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.orc_wavenet_free(self._h)
            self._h = None

    def set_filter(self, a):
        a = _c(a)
        return bool(lib.orc_set_filter(self._h, _p(a), len(a)))

    def set_lambda(self, v): lib.orc_set_lambda(self._h, v)
    def set_alpha(self, v): lib.orc_set_alpha(self._h, v)
    def set_inertia(self, v): lib.orc_set_inertia(self._h, v)
    def set_inertia_time_scale(self, v): lib.orc_set_inertia_time_scale(self._h, v)
    def set_batch_size(self, v): lib.orc_set_batch_size(self._h, v)
    def do_wavelet(self, flag): lib.orc_do_wavelet(self._h, int(flag))
    def clear(self): lib.orc_clear(self._h)

    @property
    def N(self): return lib.orc_num_coeffs(self._h)
    def lam(self): return lib.orc_lambda(self._h)
    def alpha(self): return lib.orc_alpha(self._h)
    def batch_size(self): return lib.orc_batch_size(self._h)

    def filter(self):
        a = np.empty(self.N)
        lib.orc_get_filter(self._h, _p(a))
        return a

    def momentum(self):
        a = np.empty(self.N)
        lib.orc_get_momentum(self._h, _p(a))
        return a

    def cost_log(self):
        n = lib.orc_cost_log_size(self._h)
        a = np.empty(n)
        lib.orc_get_cost_log(self._h, _p(a))
        return a

    def filter_log(self):
        n = lib.orc_filter_log_size(self._h)
        a = np.empty(n * self.N)
        lib.orc_get_filter_log(self._h, _p(a))
        return a.reshape(n, self.N)

    def batch_queue_size(self): return lib.orc_batch_queue_size(self._h)

    # 1D
    def forward_1d(self, x):
        x = _c(x)
        n = len(x)
        LO = np.zeros(2 * n)
        HI = np.zeros(2 * n)
        lib.orc_forward_1d(self._h, _p(x), n, _p(LO), _p(HI))
        return LO, HI

    @staticmethod
    def coeffs_1d(LO, HI):
        n = len(LO) // 2
        y = np.empty(n)
        lib.orc_coeffs_from_activations_1d(_p(LO), _p(HI), n, _p(y))
        return y

    def inverse_1d(self, y):
        y = _c(y)
        x = np.empty_like(y)
        lib.orc_inverse_1d(self._h, _p(y), len(y), _p(x))
        return x

    def backprop_1d(self, delta, LO, HI):
        delta = _c(delta)
        n = len(delta)
        g = np.empty(self.N)
        din = np.empty(n)
        lib.orc_backprop_1d(self._h, _p(delta), n, _p(LO), _p(HI), _p(g), _p(din))
        return g, din

    # 2D
    def forward_2d(self, X):
        R, C_ = np.shape(X)
        x = _flat_colmajor(X)
        AR = np.zeros(R * 4 * C_)
        AC = np.zeros(C_ * 4 * R)
        Y = np.empty(R * C_)
        lib.orc_forward_2d(self._h, _p(x), R, C_, _p(AR), _p(AC), _p(Y))
        return AR, AC, _mat_colmajor(Y, R, C_)

    def inverse_2d(self, Y):
        R, C_ = np.shape(Y)
        y = _flat_colmajor(Y)
        x = np.empty_like(y)
        lib.orc_inverse_2d(self._h, _p(y), R, C_, _p(x))
        return _mat_colmajor(x, R, C_)

    def backprop_2d(self, Delta, AR, AC):
        R, C_ = np.shape(Delta)
        d = _flat_colmajor(Delta)
        g = np.empty(self.N)
        lib.orc_backprop_2d(self._h, _p(d), R, C_, _p(AR), _p(AC), _p(g))
        return g

    def cost(self, Y):
        y = _flat_colmajor(Y)
        return lib.orc_cost(self._h, _p(y), y.size)

    def train(self, X):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        if X.shape[0] == 1 and X.shape[1] > 1:
            X = X.T  # GeneratorBase.cxx:25-31 squeezes {1,n} to {n,1}
        R, C_ = X.shape
        x = _flat_colmajor(X)
        return bool(lib.orc_train(self._h, _p(x), R, C_))

    def flush(self): lib.orc_flush_batch_queue(self._h)

    def coach_run(self, init_filter, examples, num_events, num_epochs=1,
                  adaptive_lr=False, adaptive_batch=False, annealing=False, target_precision=-1.0):
        """examples: array (num_events, R, C) in natural numpy (row, col) indexing."""
        ex = np.asarray(examples, dtype=np.float64)
        E, R, C_ = ex.shape
        assert E >= num_events
        flat = np.ascontiguousarray(np.transpose(ex, (0, 2, 1))).reshape(-1)  # column-major per example
        a = _c(init_filter)
        flags = int(adaptive_lr) | (int(adaptive_batch) << 1) | (int(annealing) << 2)
        return lib.orc_coach_run(self._h, _p(a), len(a), _p(flat), R, C_, num_events, num_epochs,
                                 flags, float(target_precision))


# --- banded ------------------------------------------------------------------

def band_forward_1d(a, x):
    a, x = _c(a), _c(x)
    P = np.empty_like(x)
    y = np.empty_like(x)
    lib.orb_forward_1d(_p(a), len(a), _p(x), len(x), _p(P), _p(y))
    return P, y


def band_inverse_1d(a, y):
    a, y = _c(a), _c(y)
    x = np.empty_like(y)
    lib.orb_inverse_1d(_p(a), len(a), _p(y), len(y), _p(x))
    return x


def band_backprop_1d(a, delta, x, P):
    a, delta, x, P = _c(a), _c(delta), _c(x), _c(P)
    g = np.empty(len(a))
    din = np.empty_like(x)
    lib.orb_backprop_1d(_p(a), len(a), _p(delta), _p(x), _p(P), len(x), _p(g), _p(din))
    return g, din


def band_forward_2d(a, X):
    a = _c(a)
    R, C_ = np.shape(X)
    x = _flat_colmajor(X)
    outs = [np.empty(R * C_) for _ in range(4)]
    lib.orb_forward_2d(_p(a), len(a), _p(x), R, C_, *[_p(o) for o in outs])
    return tuple(_mat_colmajor(o, R, C_) for o in outs)  # Prow, Y1, Pcol, Y


def band_inverse_2d(a, Y):
    a = _c(a)
    R, C_ = np.shape(Y)
    y = _flat_colmajor(Y)
    x = np.empty_like(y)
    lib.orb_inverse_2d(_p(a), len(a), _p(y), R, C_, _p(x))
    return _mat_colmajor(x, R, C_)


def band_backprop_2d(a, Delta, X, Prow, Y1, Pcol):
    a = _c(a)
    R, C_ = np.shape(X)
    arrs = [_flat_colmajor(m) for m in (Delta, X, Prow, Y1, Pcol)]
    g = np.empty(len(a))
    lib.orb_backprop_2d(_p(a), len(a), *[_p(v) for v in arrs], R, C_, _p(g))
    return g


def band_train_batch(a, Xb):
    """Xb: (B, R, C) natural indexing.  Returns (gsum[N], ssum, n_nonfinite)."""
    a = _c(a)
    ex = np.asarray(Xb, dtype=np.float64)
    B, R, C_ = ex.shape
    flat = np.ascontiguousarray(np.transpose(ex, (0, 2, 1))).reshape(-1)
    g = np.empty(len(a))
    s = C.c_double(0)
    bad = lib.orb_train_batch(_p(a), len(a), _p(flat), B, R, C_, _p(g), C.byref(s))
    return g, s.value, bad
