"""ctypes front-end for oracle/bwdtrans_ref.c plus an independent numpy restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference lines
its C counterpart follows; the numpy `einsum` forms are a second, independent statement of the
same maths used to cross-check the C loop nests.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIBS = {}

_c_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    want = [os.path.join(_BUILD, n) for n in ("liboracle.so", "liboracle_fast.so", "liboracle_avx512.so")]
    src = os.path.join(_HERE, "bwdtrans_ref.c")
    if force or not all(os.path.exists(w) and os.path.getmtime(w) >= os.path.getmtime(src)
                        for w in want):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return want


def host_has_avx512():
    """True when every CPU flag liboracle_avx512.so was compiled for is listed in /proc/cpuinfo."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    flags = set(line.split(":", 1)[1].split())
                    return {"avx512f", "avx512vl", "avx512dq", "fma"} <= flags
    except OSError:
        pass
    return False


def _lib(fast=False):
    """fast: False = parity build (-O2, no contraction), True = AVX2/FMA build, "avx512" = 8-wide build."""
    key = "avx512" if fast == "avx512" else "fast" if fast else "parity"
    if key == "avx512" and not host_has_avx512():
        raise RuntimeError("liboracle_avx512.so needs avx512f/vl/dq, which this host does not list")
    if key not in _LIBS:
        path = os.path.join(_BUILD, {"parity": "liboracle.so", "fast": "liboracle_fast.so",
                                     "avx512": "liboracle_avx512.so"}[key])
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        sz, u32, u64, dbl = ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint64, ctypes.c_double
        lib.oracle_max_threads.restype = ctypes.c_int
        lib.oracle_set_threads.argtypes = [ctypes.c_int]
        lib.set_threads_done = False
        lib.oracle_fill_sincos.argtypes = [_c_dp, sz, sz]
        lib.oracle_fill_basis.argtypes = [_c_dp, sz, sz]
        lib.oracle_fill_random.argtypes = [_c_dp, sz, u64, u64]
        lib.oracle_random_value.argtypes = [u64, u64]
        lib.oracle_random_value.restype = dbl
        lib.oracle_fill_l2norm.argtypes = [_c_dp, sz]
        lib.oracle_fill_vecadd.argtypes = [_c_dp, _c_dp, sz]
        lib.oracle_vector_add.argtypes = [_c_dp, _c_dp, sz, ctypes.c_int]
        lib.oracle_fill_matvec.argtypes = [_c_dp, _c_dp, sz, sz]
        lib.oracle_matvec.argtypes = [sz, sz, _c_dp, _c_dp, _c_dp]
        lib.oracle_sumsq.argtypes = [_c_dp, sz]
        lib.oracle_sumsq.restype = dbl
        lib.oracle_has_blocked.argtypes = [u32, u32, u32]
        lib.oracle_has_blocked.restype = ctypes.c_int
        for name in ("oracle_bwdtrans_hex_fused", "oracle_bwdtrans_hex_sweeps",
                     "oracle_bwdtrans_hex_vector", "oracle_bwdtrans_hex_blocked"):
            f = getattr(lib, name)
            f.argtypes = [u32, u32, u32, sz, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]
            f.restype = ctypes.c_int
        for name in ("oracle_bwdtrans_quad_fused", "oracle_bwdtrans_quad_sweeps"):
            f = getattr(lib, name)
            f.argtypes = [u32, u32, sz, _c_dp, _c_dp, _c_dp, _c_dp]
            f.restype = ctypes.c_int
        _LIBS[key] = lib
    return _LIBS[key]


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_dp)


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                tok = fh.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(int(tok[0]) / int(tok[1]))))
            else:
                quota = int(tok[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, quota // int(fh.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def set_threads(n, fast=False):
    _lib(fast).oracle_set_threads(int(n))


def max_threads(fast=False):
    return int(_lib(fast).oracle_max_threads())


# ---------------------------------------------------------------- initialisers -----------------

def fill_sincos(nelmt, nm_tot):
    """in[e][f] = sin(f+1), every element identical (benchmark05/benchmark05.cc:1206-1207)."""
    a = np.empty(nelmt * nm_tot, dtype=np.float64)
    _lib().oracle_fill_sincos(_p(a), nelmt, nm_tot)
    return a


def fill_basis(nm, nq):
    """basis[p*nq+i] = cos(p*nq+i) (benchmark05/benchmark05.cc:1216-1222)."""
    a = np.empty(nm * nq, dtype=np.float64)
    _lib().oracle_fill_basis(_p(a), nm, nq)
    return a


def fill_random(n, seed, first_idx=0):
    """Counter-based U[-1,1): value = f(seed, first_idx + i).  Not in the reference."""
    a = np.empty(n, dtype=np.float64)
    _lib().oracle_fill_random(_p(a), n, seed, first_idx)
    return a


def random_value_py(seed, idx):
    """Pure-Python statement of the generator (pins the C and the HIP implementations)."""
    m = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        return z ^ (z >> 31)

    h = mix((seed ^ mix(idx)) & m)
    return float(h >> 11) * (2.0 / 9007199254740992.0) - 1.0


def fill_l2norm(n):
    """x[i] = i%13 + (0.2 + 1e-5*(i%100191)) (benchmark01/benchmark01.cc:178)."""
    a = np.empty(n, dtype=np.float64)
    _lib().oracle_fill_l2norm(_p(a), n)
    return a


def fill_vecadd(n):
    """benchmark02 data (benchmark02/benchmark02.cc:84-85)."""
    x, y = np.empty(n, dtype=np.float64), np.empty(n, dtype=np.float64)
    _lib().oracle_fill_vecadd(_p(x), _p(y), n)
    return x, y


def vector_add(x, y, times=1):
    """x += y, `times` times, in place (benchmark02/benchmark02.cc:88-97 runs it 40 times)."""
    _lib().oracle_vector_add(_p(x), _p(y), x.size, times)
    return x


def fill_matvec(m, n):
    """benchmark03 data: A[i][j] = sin(i*N+j+1), x[j] = j (benchmark03/benchmark03.cc:160-167)."""
    a, x = np.empty(m * n, dtype=np.float64), np.empty(n, dtype=np.float64)
    _lib().oracle_fill_matvec(_p(a), _p(x), m, n)
    return a, x


def matvec(m, n, a, x):
    """y = A x, A row-major m x n (benchmark03/benchmark03.cc:80-104)."""
    y = np.empty(m, dtype=np.float64)
    _lib().oracle_matvec(m, n, _p(a), _p(x), _p(y))
    return y


def sumsq(x, fast=False):
    """sum x^2, pairwise (benchmark05/benchmark05.cc:1273-1276 semantics; sqrt at print :1397)."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    return float(_lib(fast).oracle_sumsq(_p(x), x.size))


# ---------------------------------------------------------------- BwdTrans ---------------------

def has_blocked(nq):
    """True when the register-blocked form (the timed CPU baseline) exists for isotropic order nq."""
    return bool(_lib().oracle_has_blocked(nq, nq, nq))


def bwdtrans_hex(nq, nelmt, b0, b1, b2, inp, form="sweeps", fast=False, out=None):
    """3D hex BwdTrans.  form='fused' -> benchmark05.cc:57-101, 'sweeps' -> :361-423,
    'vector' -> the same sweeps in a CPU-vectorisable loop order, 'blocked' -> the same sweeps
    register-blocked over 4-wide i-vectors (isotropic nq 2..10; the timed CPU baseline)."""
    nq0, nq1, nq2 = nq
    if out is None:     # a timing loop passes its own (already touched) buffer
        out = np.empty(nelmt * nq0 * nq1 * nq2, dtype=np.float64)
    assert out.size == nelmt * nq0 * nq1 * nq2
    f = getattr(_lib(fast), "oracle_bwdtrans_hex_" + form)
    rc = f(nq0, nq1, nq2, nelmt, _p(b0), _p(b1), _p(b2), _p(inp), _p(out))
    if rc != 0:
        raise ValueError(f"oracle_bwdtrans_hex_{form} rc={rc}")
    return out


def bwdtrans_quad(nq, nelmt, b0, b1, inp, form="sweeps", fast=False):
    """2D quad BwdTrans.  form='fused' -> benchmark04.cc:49-72, 'sweeps' -> :393-420."""
    nq0, nq1 = nq
    out = np.empty(nelmt * nq0 * nq1, dtype=np.float64)
    f = getattr(_lib(fast), "oracle_bwdtrans_quad_" + form)
    rc = f(nq0, nq1, nelmt, _p(b0), _p(b1), _p(inp), _p(out))
    if rc != 0:
        raise ValueError(f"oracle_bwdtrans_quad_{form} rc={rc}")
    return out


def bwdtrans_hex_numpy(nq, nelmt, b0, b1, b2, inp):
    """Independent statement: out[e,k,j,i] = sum_rqp in[e,r,q,p] B0[p,i] B1[q,j] B2[r,k]."""
    nq0, nq1, nq2 = nq
    nm0, nm1, nm2 = nq0 - 1, nq1 - 1, nq2 - 1
    u = inp.reshape(nelmt, nm2, nm1, nm0)
    out = np.einsum("erqp,pi,qj,rk->ekji", u, b0.reshape(nm0, nq0), b1.reshape(nm1, nq1),
                    b2.reshape(nm2, nq2), optimize=True)
    return np.ascontiguousarray(out).ravel()


def bwdtrans_quad_numpy(nq, nelmt, b0, b1, inp):
    """Independent statement: out[e,j,i] = sum_qp in[e,q,p] B0[p,i] B1[q,j]."""
    nq0, nq1 = nq
    nm0, nm1 = nq0 - 1, nq1 - 1
    u = inp.reshape(nelmt, nm1, nm0)
    out = np.einsum("eqp,pi,qj->eji", u, b0.reshape(nm0, nq0), b1.reshape(nm1, nq1),
                    optimize=True)
    return np.ascontiguousarray(out).ravel()


def rel_err(a, ref):
    """Norm-wise relative error max|a-ref| / max|ref| (outputs have cancellation)."""
    ref = np.asarray(ref)
    d = float(np.max(np.abs(np.asarray(a) - ref))) if ref.size else 0.0
    m = float(np.max(np.abs(ref))) if ref.size else 0.0
    return d / m if m > 0 else d
