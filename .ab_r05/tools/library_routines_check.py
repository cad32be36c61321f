"""The rocSOLVER / rocBLAS routines the library calls, checked against NumPy at the sizes where rocSOLVER 3.32's getri is wrong
(N = 127 mod 128) and a few others: getrf + getrs, potrf + potrs, trsm, gesvd.  (getri itself is no longer used.)
    python3 tools/library_routines_check.py"""
import ctypes

import numpy as np

hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
rb = ctypes.CDLL("/opt/rocm/lib/librocblas.so")
rs = ctypes.CDLL("/opt/rocm/lib/librocsolver.so")


def dmalloc(nbytes):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
    return p


def up(a):
    d = dmalloc(a.nbytes)
    hip.hipMemcpy(d, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes), 1)
    return d


def down(d, shape):
    x = np.empty(shape)
    hip.hipMemcpy(x.ctypes.data_as(ctypes.c_void_p), d, ctypes.c_size_t(x.nbytes), 2)
    return x


h = ctypes.c_void_p()
assert rb.rocblas_create_handle(ctypes.byref(h)) == 0
rng = np.random.default_rng(0)
one = ctypes.c_double(1.0)
for N in (127, 255, 300, 383, 511, 512, 639, 767, 1023):
    G = rng.normal(size=(N, N))
    A = G @ G.T + N * np.eye(N)          # SPD
    B = rng.normal(size=(N, 8))          # 8 right-hand sides (column-major N x 8 == row-major 8 x N of the transpose)
    Bcm = np.asfortranarray(B)
    # potrf (lower, column-major view of the symmetric A) + potrs
    dA, dB, dI = up(A), up(np.ascontiguousarray(Bcm.T)), dmalloc(4)
    r1 = rs.rocsolver_dpotrf(h, 122, N, dA, N, dI)   # rocblas_fill_lower = 122
    r2 = rs.rocsolver_dpotrs(h, 122, N, 8, dA, N, dB, N)
    hip.hipDeviceSynchronize()
    X = down(dB, (8, N)).T
    e_potrs = np.abs(A @ X - B).max() / np.abs(B).max()
    # getrf + getrs
    dA, dB, dP = up(np.ascontiguousarray(G.T)), up(np.ascontiguousarray(Bcm.T)), dmalloc(4 * N)
    rs.rocsolver_dgetrf(h, N, N, dA, N, dP, dI)
    rs.rocsolver_dgetrs(h, 111, N, 8, dA, N, dP, dB, N)
    hip.hipDeviceSynchronize()
    X = down(dB, (8, N)).T
    e_getrs = np.abs(G @ X - B).max() / np.abs(B).max()
    # trsm: L X = B with the Cholesky factor from NumPy (left, lower, no transpose, non-unit)
    Lc = np.linalg.cholesky(A)
    dL, dB = up(np.ascontiguousarray(Lc.T)), up(np.ascontiguousarray(Bcm.T))
    rb.rocblas_dtrsm(h, 141, 122, 111, 131, N, 8, ctypes.byref(one), dL, N, dB, N)  # side_left 141, lower 122, none 111, non_unit 131
    hip.hipDeviceSynchronize()
    X = down(dB, (8, N)).T
    e_trsm = np.abs(Lc @ X - B).max() / np.abs(B).max()
    # gesvd of a symmetric indefinite matrix: reconstruct
    S_ = G + G.T
    dA, dS, dU, dV, dE = up(S_), dmalloc(8 * N), dmalloc(8 * N * N), dmalloc(8 * N * N), dmalloc(8 * N)
    r = rs.rocsolver_dgesvd(h, 191, 191, N, N, dA, N, dS, dU, N, dV, N, dE, 201, dI)  # svect_all 191, outofplace 201
    hip.hipDeviceSynchronize()
    sv, U, Vt = down(dS, (N,)), down(dU, (N, N)).T, down(dV, (N, N)).T
    e_svd = np.abs((U * sv) @ Vt - S_).max() / np.abs(S_).max()
    print("N=%4d  potrf+potrs %.1e  getrf+getrs %.1e  trsm %.1e  gesvd %.1e" % (N, e_potrs, e_getrs, e_trsm, e_svd), flush=True)
