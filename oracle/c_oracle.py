"""Oracle (test infrastructure): ctypes wrapper of the plain-C Kalman port (oracle/c/kf_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_kf.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_kalman_filter_f32.restype = C.c_int
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def precompute(a):
    """(G Q) G^T, (D R) D^T, G q0, D r0 in fp32 with the reference's association."""
    G, D, Q, R = (np.asarray(a[k], F32) for k in ("G", "D", "Q", "R"))
    GQG = np.matmul(np.matmul(G, Q, dtype=F32), G.T, dtype=F32)
    DRD = np.matmul(np.matmul(D, R, dtype=F32), D.T, dtype=F32)
    return GQG, DRD, np.matmul(G, a["q0"], dtype=F32), np.matmul(D, a["r0"], dtype=F32)


def kalman_filter(a, ys, init_means, init_covs=None, fields=("weights", "means", "covariances", "predicted_means",
                                                              "predicted_covariances", "loglik"), nthreads=0):
    """a: dict of model arrays (A,G,H,D,Q,R,q0,r0,P0); ys (B,T,m).  Returns dict of (B,1,T,...) arrays."""
    lib = load()
    ys = np.ascontiguousarray(ys, F32)
    B, T, m = ys.shape
    n = a["A"].shape[0]
    GQG, DRD, Gq0, Dr0 = (np.ascontiguousarray(v, F32) for v in precompute(a))
    A, H = np.ascontiguousarray(a["A"], F32), np.ascontiguousarray(a["H"], F32)
    m_in = np.ascontiguousarray(np.broadcast_to(np.asarray(init_means, F32).reshape(-1, n), (B, n)))
    P0 = a["P0"] if init_covs is None else init_covs
    P_in = np.ascontiguousarray(np.broadcast_to(np.asarray(P0, F32).reshape(-1, n, n), (B, n, n)))
    shapes = {"weights": (B, 1, T), "means": (B, 1, T, n), "covariances": (B, 1, T, n, n),
              "predicted_means": (B, 1, T, n), "predicted_covariances": (B, 1, T, n, n), "loglik": (B, 1, T)}
    out = {k: (np.empty(shapes[k], F32) if k in fields else None) for k in shapes}
    rc = lib.oracle_kalman_filter_f32(C.c_int(n), C.c_int(m), _p(A), _p(H), _p(GQG), _p(DRD), _p(Gq0), _p(Dr0),
                                      _p(ys), C.c_int64(B), C.c_int64(T), _p(m_in), _p(P_in), _p(out["weights"]),
                                      _p(out["means"]), _p(out["covariances"]), _p(out["predicted_means"]),
                                      _p(out["predicted_covariances"]), _p(out["loglik"]), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError("oracle_kalman_filter_f32 failed")
    return {k: v for k, v in out.items() if v is not None}


def max_threads():
    return load().oracle_max_threads()


def gsf_lorenz96(theta, H, Q, R, q0, r0, ys, init_means, P0, fields=("weights", "means", "covariances"), nthreads=0):
    """The EKF bank + weight update for Lorenz-96 dynamics / linear emission (oracle_gsf_lorenz96_f32).  ys (B, T, m),
    init_means (B, K, n).  Returns dict of (B, K, T, ...) arrays."""
    lib = load()
    lib.oracle_gsf_lorenz96_f32.restype = C.c_int
    ys = np.ascontiguousarray(ys, F32)
    B, T, m = ys.shape
    init_means = np.ascontiguousarray(init_means, F32)
    K, n = init_means.shape[1], init_means.shape[2]
    arrs = [np.ascontiguousarray(v, F32) for v in (theta, H, Q, R, q0, r0)]
    P0 = np.ascontiguousarray(P0, F32)
    shapes = {"weights": (B, K, T), "means": (B, K, T, n), "covariances": (B, K, T, n, n)}
    out = {k: (np.empty(shapes[k], F32) if k in fields else None) for k in shapes}
    rc = lib.oracle_gsf_lorenz96_f32(C.c_int(n), C.c_int(m), C.c_int(K), *[_p(a) for a in arrs], _p(ys), C.c_int64(B), C.c_int64(T),
                                     _p(init_means), _p(P0), _p(out["weights"]), _p(out["means"]), _p(out["covariances"]), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError("oracle_gsf_lorenz96_f32 failed")
    return {k: v for k, v in out.items() if v is not None}


def bpf_lorenz96(theta, q0, Qdiag, lp_diag, m0, P0diag, ys, N, key, ess_threshold=0.5, nthreads=0):
    """The bootstrap particle filter for Lorenz-96 dynamics, diagonal covariances, even-state emission
    (oracle_bpf_lorenz96_f32).  ys (B, T, m).  Returns {'mean': (B, T, n), 'resampled': (B, T)}."""
    lib = load()
    lib.oracle_bpf_lorenz96_f32.restype = C.c_int
    ys = np.ascontiguousarray(ys, F32)
    B, T, m = ys.shape
    n = len(m0)
    arrs = [np.ascontiguousarray(v, F32) for v in (theta, q0, np.sqrt(np.asarray(Qdiag, F32)), np.sqrt(np.asarray(lp_diag, F32)), m0,
                                                   np.sqrt(np.asarray(P0diag, F32)))]
    key = np.ascontiguousarray(key, np.uint32)
    mean = np.empty((B, T, n), F32)
    res = np.empty((B, T), F32)
    rc = lib.oracle_bpf_lorenz96_f32(C.c_int(n), C.c_int(m), C.c_int(N), *[_p(a) for a in arrs], _p(ys), C.c_int64(B), C.c_int64(T),
                                     _p(key), C.c_float(ess_threshold), _p(mean), _p(res), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError("oracle_bpf_lorenz96_f32 failed")
    return {"mean": mean, "resampled": res}
