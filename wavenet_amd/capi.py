"""ctypes binding of the C ABI declared in include/wavenet_hip.h.

Device buffers are passed as integer addresses (e.g. torch.Tensor.data_ptr()); torch is only
used by callers for device memory and streams, never for the arithmetic.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

WN_OK = 0
WN_ERR_ARG, WN_ERR_BAD_SHAPE, WN_ERR_BAD_FILTER, WN_ERR_UNSUPPORTED = 1, 2, 3, 4
WN_ERR_HIP, WN_ERR_COMM, WN_ERR_NONFINITE, WN_ERR_STATE = 5, 6, 7, 8
WN_MAX_TAPS = 32
WN_MAX_FUSED = 4096

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/wavenet_hip.h declares
SIGNATURES = {
    "wn_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "wn_destroy": (C.c_int, [_vp]),
    "wn_set_stream": (C.c_int, [_vp, _vp]),
    "wn_synchronize": (C.c_int, [_vp]),
    "wn_last_error": (C.c_char_p, [_vp]),
    "wn_status_string": (C.c_char_p, [C.c_int]),
    "wn_malloc": (C.c_int, [_vp, C.POINTER(_vp), C.c_size_t]),
    "wn_free": (C.c_int, [_vp, _vp]),
    "wn_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "wn_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "wn_set_filter": (C.c_int, [_vp, _dp, C.c_int]),
    "wn_get_filter": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "wn_set_momentum": (C.c_int, [_vp, _dp, C.c_int]),
    "wn_get_momentum": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "wn_lowpass_operator": (C.c_int, [_dp, C.c_int, C.c_uint, _dp]),
    "wn_highpass_operator": (C.c_int, [_dp, C.c_int, C.c_uint, _dp]),
    "wn_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "wn_inverse": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "wn_backprop": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "wn_sparse_term": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "wn_sparse_term_deriv": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "wn_reg_term": (C.c_int, [_dp, C.c_int, C.c_int, _dp]),
    "wn_reg_term_deriv": (C.c_int, [_dp, C.c_int, C.c_int, _dp]),
    "wn_sparse_cost_grid": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _vp]),
    "wn_generate_uniform": (C.c_int, [_vp, C.c_ulonglong, C.c_longlong, C.c_int, C.c_int, C.c_int, _vp]),
    "wn_generate_needle": (C.c_int, [_vp, C.c_ulonglong, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_double, _vp]),
    "wn_generate_calorimeter": (C.c_int, [_vp, C.c_ulonglong, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_int), _dp, C.c_int, C.c_int, _vp]),
    "wn_get_device": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "wn_generate_gaussian": (C.c_int, [_vp, C.c_ulonglong, C.c_longlong, C.c_int, C.c_int, C.c_int, _dp, _vp]),
    "wn_train_batch": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "wn_train_batch_static": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "wn_sums": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_int)]),
    "wn_get_sums": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "wn_set_sums_buffer": (C.c_int, [_vp, _vp]),
    "wn_profile_begin": (C.c_int, [_vp, C.c_int]),
    "wn_profile_collect": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "wn_apply_update": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "wn_train_step": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "wn_num_updates": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "wn_get_logs": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp]),
    "wn_get_updates": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), _dp, _dp, _dp]),
    "wn_clear_logs": (C.c_int, [_vp]),
    "wn_comm_unique_id": (C.c_int, [_vp]),
    "wn_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "wn_comm_destroy": (C.c_int, [_vp]),
    "wn_allreduce_sums": (C.c_int, [_vp]),
    "wn_comm_init_all": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "wn_allreduce_sums_all": (C.c_int, [C.POINTER(_vp), C.c_int]),
}


class WavenetError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[wn status {status}] {message}")
        self.status = status


def lib_path():
    """The library next to this package, unless WAVENET_HIP_LIBRARY names another build of it."""
    return os.environ.get("WAVENET_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libwavenet_hip.so")


_lib = None


def load_library():
    """Load libwavenet_hip.so and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64 under the same soname as /opt/rocm's; whichever is
        # mapped first serves the whole process, and a process that ends up with both cannot
        # see the GPU from the second.  Callers use torch for device memory, so let it go first.
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `make -C wavenet_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _addr(x):
    """Device address of a torch tensor, or an int/None passed through."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


def _hd(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


# --- host-only entry points ----------------------------------------------------------------------

def reg_term(a, wavelet=True):
    lib = load_library()
    a, pa = _hd(a)
    out = C.c_double(0)
    st = lib.wn_reg_term(pa, len(a), int(wavelet), C.byref(out))
    if st:
        raise WavenetError(st, lib.wn_status_string(st).decode())
    return out.value


def reg_term_deriv(a, wavelet=True):
    lib = load_library()
    a, pa = _hd(a)
    g = np.empty_like(a)
    st = lib.wn_reg_term_deriv(pa, len(a), int(wavelet), g.ctypes.data_as(_dp))
    if st:
        raise WavenetError(st, lib.wn_status_string(st).decode())
    return g


def dense_operator(a, size, highpass=False):
    """LowpassOperator(a, size) / HighpassOperator(a, size) as a 2^size x 2^(size+1) numpy matrix."""
    lib = load_library()
    a, pa = _hd(a)
    nr, nc = 1 << size, 2 << size
    M = np.empty(nr * nc)
    fn = lib.wn_highpass_operator if highpass else lib.wn_lowpass_operator
    st = fn(pa, len(a), size, M.ctypes.data_as(_dp))
    if st:
        raise WavenetError(st, lib.wn_status_string(st).decode())
    return M.reshape((nr, nc), order="F").copy()


class Context:
    """One context per (process, device): filter state, stream, scratch, communicator."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = _vp()
        st = self.lib.wn_create(int(device), C.byref(h))
        if st:
            raise WavenetError(st, self.lib.wn_last_error(None).decode())
        self._h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st:
            raise WavenetError(st, self.lib.wn_last_error(self._h).decode())

    def set_stream(self, stream):
        """stream: integer hipStream_t (torch.cuda.current_stream().cuda_stream) or None."""
        self._check(self.lib.wn_set_stream(self._h, stream))

    def synchronize(self):
        self._check(self.lib.wn_synchronize(self._h))

    # filter ----------------------------------------------------------------------------------
    def set_filter(self, a):
        a, pa = _hd(a)
        self._check(self.lib.wn_set_filter(self._h, pa, len(a)))

    def filter(self):
        n = C.c_int(0)
        self._check(self.lib.wn_get_filter(self._h, None, C.byref(n)))
        a = np.empty(n.value)
        self._check(self.lib.wn_get_filter(self._h, a.ctypes.data_as(_dp), C.byref(n)))
        return a

    def set_momentum(self, p):
        p, pp = _hd(p)
        self._check(self.lib.wn_set_momentum(self._h, pp, len(p)))

    def momentum(self):
        n = C.c_int(0)
        self._check(self.lib.wn_get_momentum(self._h, None, C.byref(n)))
        p = np.empty(n.value)
        self._check(self.lib.wn_get_momentum(self._h, p.ctypes.data_as(_dp), C.byref(n)))
        return p

    # transforms (device buffers: torch tensors or addresses) ----------------------------------
    def forward(self, X, B, R, Cc, Y=None, Prow=None, Y1=None, Pcol=None):
        self._check(self.lib.wn_forward(self._h, _addr(X), B, R, Cc, _addr(Y), _addr(Prow), _addr(Y1), _addr(Pcol)))

    def inverse(self, Y, B, R, Cc, X):
        self._check(self.lib.wn_inverse(self._h, _addr(Y), B, R, Cc, _addr(X)))

    def backprop(self, Delta, X, Prow, Y1, Pcol, B, R, Cc, grad, din=None):
        self._check(self.lib.wn_backprop(self._h, _addr(Delta), _addr(X), _addr(Prow), _addr(Y1), _addr(Pcol),
                                         B, R, Cc, _addr(grad), _addr(din)))

    def sparse_term(self, Y, B, n, S):
        self._check(self.lib.wn_sparse_term(self._h, _addr(Y), B, n, _addr(S)))

    def sparse_term_deriv(self, Y, B, n, Delta, S=None):
        self._check(self.lib.wn_sparse_term_deriv(self._h, _addr(Y), B, n, _addr(Delta), _addr(S)))

    def sparse_cost_grid(self, X, B, R, Cc, filters, S):
        """S[f * B + e] = sparse term of example e under filter f; filters: (F, N) numpy array."""
        f = np.ascontiguousarray(filters, dtype=np.float64)
        assert f.ndim == 2
        self._check(self.lib.wn_sparse_cost_grid(self._h, _addr(X), B, R, Cc, f.ctypes.data_as(_dp), f.shape[0],
                                                 f.shape[1], _addr(S)))

    # generators ---------------------------------------------------------------------------------
    def generate_uniform(self, X, B, R, Cc, seed=1, first=0):
        self._check(self.lib.wn_generate_uniform(self._h, seed, first, B, R, Cc, _addr(X)))

    def generate_needle(self, X, B, R, Cc, seed=1, first=0, threshold=0.995):
        self._check(self.lib.wn_generate_needle(self._h, seed, first, B, R, Cc, threshold, _addr(X)))

    def generate_calorimeter(self, X, B, R, Cc, seed=1, first=0, mean_deposits=None, jets=2, jet_deposits=30):
        """Same images as generators.calorimeter(...)[first:first + B], written in device layout."""
        from . import generators as G
        if mean_deposits is None:
            mean_deposits = max(1.0, 200.0 * (R * Cc) / 4096.0)
        ptab = np.ascontiguousarray(G._poisson_table(float(mean_deposits)), dtype=np.int32)
        etab = np.ascontiguousarray(G._EXP_TABLE, dtype=np.float64)
        self._check(self.lib.wn_generate_calorimeter(self._h, seed, first, B, R, Cc, ptab.ctypes.data_as(C.POINTER(C.c_int)),
                                                     etab.ctypes.data_as(_dp), jets, jet_deposits, _addr(X)))

    def generate_gaussian(self, X, B, R, Cc, seed=1, first=0):
        """wn_generate_gaussian: the sequence of generators.gaussian, written on the device."""
        from .generators import GAUSS_EXP_TABLE
        tab = np.ascontiguousarray(GAUSS_EXP_TABLE, dtype=np.float64)
        self._check(self.lib.wn_generate_gaussian(self._h, seed, first, B, R, Cc, tab.ctypes.data_as(_dp), _addr(X)))

    # training -----------------------------------------------------------------------------------
    def train_batch(self, X, B, R, Cc, grad=None, S=None):
        self._check(self.lib.wn_train_batch(self._h, _addr(X), B, R, Cc, _addr(grad), _addr(S)))

    def train_batch_static(self, X, B, R, Cc, sums):
        """wn_train_batch_static: no allocation; sums: device tensor/address of N + 2 doubles."""
        self._check(self.lib.wn_train_batch_static(self._h, _addr(X), B, R, Cc, _addr(sums)))

    def sums_ptr(self):
        p = _vp()
        n = C.c_int(0)
        self._check(self.lib.wn_sums(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def sums(self):
        n = C.c_int(0)
        self._check(self.lib.wn_get_sums(self._h, None, C.byref(n)))
        s = np.empty(n.value)
        self._check(self.lib.wn_get_sums(self._h, s.ctypes.data_as(_dp), C.byref(n)))
        return s

    def set_sums_buffer(self, buf):
        """buf: device tensor/address of >= WN_MAX_TAPS + 2 doubles, or None for the context's own."""
        self._check(self.lib.wn_set_sums_buffer(self._h, _addr(buf)))

    def profile_begin(self, max_records):
        self._check(self.lib.wn_profile_begin(self._h, int(max_records)))

    def profile_collect(self, capacity=4096):
        ms = (C.c_float * capacity)()
        n = C.c_int(0)
        self._check(self.lib.wn_profile_collect(self._h, ms, capacity, C.byref(n)))
        return np.array(ms[:n.value], dtype=np.float64)

    def apply_update(self, batch_total, lambda_sum, alpha, inertia_eff=0.0, wavelet=True):
        self._check(self.lib.wn_apply_update(self._h, float(batch_total), float(lambda_sum), float(alpha),
                                             float(inertia_eff), int(wavelet)))

    def train_step(self, X, B, R, Cc, batch_total, lambda_sum, alpha, inertia_eff=0.0, wavelet=True):
        """wn_train_step: train_batch + (all-reduce over the context's communicator) + apply_update, one submission."""
        self._check(self.lib.wn_train_step(self._h, _addr(X), B, R, Cc, float(batch_total), float(lambda_sum),
                                           float(alpha), float(inertia_eff), int(wavelet)))

    def num_updates(self):
        n = C.c_int(0)
        self._check(self.lib.wn_num_updates(self._h, C.byref(n)))
        return n.value

    def logs(self, first=0, count=None):
        if count is None:
            count = self.num_updates() - first
        N = len(self.filter())
        F = np.empty((count, N))
        Cs = np.empty(count)
        self._check(self.lib.wn_get_logs(self._h, first, count, F.ctypes.data_as(_dp), Cs.ctypes.data_as(_dp)))
        return F, Cs

    def updates(self, first=0, max_count=1024):
        """wn_get_updates: (steps, filters[count][N], costs[count], momentum[N], rejected) in one readback."""
        N = len(self.filter())
        F = np.empty((max_count, N))
        Cs = np.empty(max_count)
        P = np.empty(N)
        n = C.c_int(0)
        st = self.lib.wn_get_updates(self._h, first, max_count, C.byref(n), F.ctypes.data_as(_dp), Cs.ctypes.data_as(_dp),
                                     P.ctypes.data_as(_dp))
        if st not in (0, WN_ERR_NONFINITE):
            self._check(st)
        count = max(0, min(max_count, n.value - first))
        return n.value, F[:count], Cs[:count], P, st == WN_ERR_NONFINITE

    def clear_logs(self):
        self._check(self.lib.wn_clear_logs(self._h))

    # communicator ---------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        st = self.lib.wn_comm_unique_id(buf)
        if st:
            raise WavenetError(st, self.lib.wn_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, unique_id, nranks, rank):
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        self._check(self.lib.wn_comm_init(self._h, buf, nranks, rank))

    def allreduce_sums(self):
        self._check(self.lib.wn_allreduce_sums(self._h))
