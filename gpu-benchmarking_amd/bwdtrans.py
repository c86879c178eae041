"""Host-side mirror of the reference's kernel call sites, over torch device tensors.

torch is plumbing only (device memory, streams); all arithmetic happens in libsumfact.so.
Argument meaning follows the reference kernels (benchmark05/benchmark05.cc:291-297,
benchmark04/benchmark04.cc:353-358): extents nq (nm = nq-1), basis row-major nm x nq,
in[e][r][q][p], out[e][k][j][i].
"""
import ctypes

import torch

from . import capi

VARIANTS = {"auto": 0, "wave": 1, "thread": 2, "block-lds": 3, "block-glb": 4, "generic": 5,
            "mfma": 6, "mfma4": 7, "wave-rt": 8}


def _stream(stream, device=None):
    """hipStream_t of `stream`, or of the current stream of `device` (the tensor's device, not the thread's)."""
    if stream is None:
        stream = torch.cuda.current_stream(device)
    return ctypes.c_void_p(stream.cuda_stream)


def _check_sizes(what, basis, nq, wsp, wsp_need):
    """The C ABI takes raw pointers: a short basis or workspace would be read / written out of bounds."""
    for d, (b, q) in enumerate(zip(basis, nq)):
        if b.numel() != (q - 1) * q:
            raise ValueError(f"{what}: basis{d} has {b.numel()} values, nm*nq = {(q - 1) * q}")
    if wsp is not None and wsp.numel() < wsp_need:
        raise ValueError(f"{what}: wsp has {wsp.numel()} values, the variant needs {wsp_need}")


def _dev_f64(t, name, dtype=torch.float64):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError(f"{name} must be a contiguous {dtype} CUDA/HIP tensor")
    return ctypes.c_void_p(t.data_ptr())


def _dev_f32(t, name):
    return _dev_f64(t, name, torch.float32)


def _variant(v):
    return VARIANTS[v] if isinstance(v, str) else int(v)


def hex_wsp_doubles(nq, nelmt):
    """Workspace the block-glb / thread variants need (benchmark05/benchmark05.cc:1243-1244)."""
    nq0, nq1, nq2 = nq
    return nelmt * (nq0 * (nq1 - 1) * (nq2 - 1) + nq0 * nq1 * (nq2 - 1))


def quad_wsp_doubles(nq, nelmt):
    """Workspace of the 2D block-glb / thread variants (benchmark04/benchmark04.cc:894)."""
    nq0, nq1 = nq
    return nelmt * nq0 * (nq1 - 1)


def bwdtrans_hex(nq, basis0, basis1, basis2, inp, out=None, variant="auto", wsp=None, stream=None):
    """out[e][k][j][i] = sum_rqp in[e][r][q][p] B0[p][i] B1[q][j] B2[r][k] on inp's device."""
    nq0, nq1, nq2 = (int(x) for x in nq)
    nmt = (nq0 - 1) * (nq1 - 1) * (nq2 - 1)
    if nmt <= 0:
        raise capi.SumfactError(capi.SF_EINVAL, "bwdtrans_hex")
    nelmt = inp.numel() // nmt
    if nelmt * nmt != inp.numel():
        raise ValueError("in.numel() is not a multiple of nm0*nm1*nm2")
    if out is None:
        out = torch.empty(nelmt * nq0 * nq1 * nq2, dtype=inp.dtype, device=inp.device)
    elif out.numel() != nelmt * nq0 * nq1 * nq2:
        raise ValueError("out has the wrong size")
    v = _variant(variant)
    _check_sizes("bwdtrans_hex", (basis0, basis1, basis2), (nq0, nq1, nq2), wsp,
                 {2: nelmt * ((nq1 - 1) * (nq2 - 1) + (nq2 - 1)),
                  4: hex_wsp_doubles((nq0, nq1, nq2), nelmt)}.get(v, 0))
    if inp.dtype == torch.float32:      # T = float instantiation; AUTO strategy only
        with torch.cuda.device(inp.device):
            rc = capi.lib().sf_bwdtrans_hex_f32(
                nq0, nq1, nq2, nelmt, _dev_f32(basis0, "basis0"), _dev_f32(basis1, "basis1"),
                _dev_f32(basis2, "basis2"), _dev_f32(inp, "in"), _dev_f32(out, "out"),
                _stream(stream, inp.device))
        capi.check(rc, "sf_bwdtrans_hex_f32")
        return out
    v = _variant(variant)
    if wsp is None and v in (2, 4) and nelmt:
        wsp = torch.empty(hex_wsp_doubles((nq0, nq1, nq2), nelmt), dtype=torch.float64,
                          device=inp.device)
    with torch.cuda.device(inp.device):
        rc = capi.lib().sf_bwdtrans_hex_f64_variant(
            v, nq0, nq1, nq2, nelmt, _dev_f64(basis0, "basis0"), _dev_f64(basis1, "basis1"),
            _dev_f64(basis2, "basis2"), _dev_f64(inp, "in"),
            _dev_f64(wsp, "wsp") if wsp is not None else None, _dev_f64(out, "out"),
            _stream(stream, inp.device))
    capi.check(rc, "sf_bwdtrans_hex_f64")
    return out


def bwdtrans_quad(nq, basis0, basis1, inp, out=None, variant="auto", wsp=None, stream=None):
    """out[e][j][i] = sum_qp in[e][q][p] B0[p][i] B1[q][j] on inp's device."""
    nq0, nq1 = (int(x) for x in nq)
    nmt = (nq0 - 1) * (nq1 - 1)
    if nmt <= 0:
        raise capi.SumfactError(capi.SF_EINVAL, "bwdtrans_quad")
    nelmt = inp.numel() // nmt
    if nelmt * nmt != inp.numel():
        raise ValueError("in.numel() is not a multiple of nm0*nm1")
    if out is None:
        out = torch.empty(nelmt * nq0 * nq1, dtype=inp.dtype, device=inp.device)
    elif out.numel() != nelmt * nq0 * nq1:
        raise ValueError("out has the wrong size")
    v = _variant(variant)
    _check_sizes("bwdtrans_quad", (basis0, basis1), (nq0, nq1), wsp,
                 {2: nelmt * (nq1 - 1), 4: quad_wsp_doubles((nq0, nq1), nelmt)}.get(v, 0))
    if inp.dtype == torch.float32:
        with torch.cuda.device(inp.device):
            rc = capi.lib().sf_bwdtrans_quad_f32(
                nq0, nq1, nelmt, _dev_f32(basis0, "basis0"), _dev_f32(basis1, "basis1"),
                _dev_f32(inp, "in"), _dev_f32(out, "out"), _stream(stream, inp.device))
        capi.check(rc, "sf_bwdtrans_quad_f32")
        return out
    v = _variant(variant)
    if wsp is None and v in (2, 4) and nelmt:
        wsp = torch.empty(quad_wsp_doubles((nq0, nq1), nelmt), dtype=torch.float64,
                          device=inp.device)
    with torch.cuda.device(inp.device):
        rc = capi.lib().sf_bwdtrans_quad_f64_variant(
            v, nq0, nq1, nelmt, _dev_f64(basis0, "basis0"), _dev_f64(basis1, "basis1"),
            _dev_f64(inp, "in"), _dev_f64(wsp, "wsp") if wsp is not None else None,
            _dev_f64(out, "out"), _stream(stream, inp.device))
    capi.check(rc, "sf_bwdtrans_quad_f64")
    return out


def interleave64(src, nelmt, n, inverse=False, stream=None):
    """[e][n] -> [(e/64)][n][e%64] (padded to whole groups of 64 elements), or back."""
    padded = (nelmt + 63) // 64 * 64
    dst = torch.zeros((nelmt if inverse else padded) * n, dtype=torch.float64, device=src.device)
    with torch.cuda.device(src.device):
        capi.check(capi.lib().sf_interleave64_f64(_dev_f64(src, "src"), _dev_f64(dst, "dst"), nelmt,
                                                  n, 1 if inverse else 0, _stream(stream, src.device)),
                   "sf_interleave64_f64")
    return dst


def bwdtrans_hex_interleaved(nq, basis0, basis1, basis2, in_il, nelmt, stream=None):
    """Thread-per-element kernel on the wave-64 interleaved layout (the corrected `_Coa`,
    benchmark05/benchmark05.cc:104-201).  Returns out_il (interleaved)."""
    nq0, nq1, nq2 = (int(x) for x in nq)
    padded = (nelmt + 63) // 64 * 64
    out = torch.empty(padded * nq0 * nq1 * nq2, dtype=torch.float64, device=in_il.device)
    wsp = torch.empty(padded * ((nq1 - 1) * (nq2 - 1) + (nq2 - 1)), dtype=torch.float64,
                      device=in_il.device)
    with torch.cuda.device(in_il.device):
        rc = capi.lib().sf_bwdtrans_hex_f64_interleaved(
            nq0, nq1, nq2, nelmt, _dev_f64(basis0, "basis0"), _dev_f64(basis1, "basis1"),
            _dev_f64(basis2, "basis2"), _dev_f64(in_il, "in_il"), _dev_f64(wsp, "wsp"),
            _dev_f64(out, "out_il"), _stream(stream, in_il.device))
    capi.check(rc, "sf_bwdtrans_hex_f64_interleaved")
    return out


def sumsq(x, stream=None):
    """sum x^2 (blocking; deterministic) -- the reference's thrust::transform_reduce."""
    res = ctypes.c_double(0.0)
    with torch.cuda.device(x.device):
        if x.dtype == torch.float32:
            rc = capi.lib().sf_sumsq_f32(_dev_f32(x, "x"), x.numel(), ctypes.byref(res),
                                         _stream(stream, x.device))
        else:
            rc = capi.lib().sf_sumsq_f64(_dev_f64(x, "x"), x.numel(), ctypes.byref(res),
                                         _stream(stream, x.device))
    capi.check(rc, "sf_sumsq")
    return res.value


def _filled(n, device, call, what, dtype=torch.float64):
    x = torch.empty(n, dtype=dtype, device=device)
    with torch.cuda.device(x.device):
        capi.check(call(ctypes.c_void_p(x.data_ptr())), what)
    return x


def _sfx(dtype):
    if dtype not in (torch.float64, torch.float32):
        raise TypeError("dtype must be torch.float64 or torch.float32")
    return "f32" if dtype == torch.float32 else "f64"


def fill_sincos(nelmt, nm_tot, device="cuda", stream=None, dtype=torch.float64):
    """in[e][f] = sin((T)(f+1)) (benchmark05/benchmark05.cc:1206-1207), generated on the device."""
    st, fn = _stream(stream, device), getattr(capi.lib(), "sf_fill_sincos_" + _sfx(dtype))
    return _filled(nelmt * nm_tot, device, lambda p: fn(p, nelmt, nm_tot, st), "sf_fill_sincos",
                   dtype)


def fill_basis(nm, nq, device="cuda", stream=None, dtype=torch.float64):
    """basis[x] = cos((T)x) (benchmark05/benchmark05.cc:1220)."""
    st, fn = _stream(stream, device), getattr(capi.lib(), "sf_fill_basis_" + _sfx(dtype))
    return _filled(nm * nq, device, lambda p: fn(p, nm, nq, st), "sf_fill_basis", dtype)


def fill_random(n, seed, first_idx=0, device="cuda", stream=None, dtype=torch.float64):
    """Seeded per-value-distinct U[-1,1) data; bit-identical to oracle.fill_random (rounded to
    float for dtype=float32)."""
    st, fn = _stream(stream, device), getattr(capi.lib(), "sf_fill_random_" + _sfx(dtype))
    return _filled(n, device, lambda p: fn(p, n, seed, first_idx, st), "sf_fill_random", dtype)


def fill_l2norm(n, device="cuda", stream=None):
    """x[i] = i%13 + (0.2 + 1e-5*(i%100191)) (benchmark01/benchmark01.cc:178)."""
    st = _stream(stream, device)
    return _filled(n, device, lambda p: capi.lib().sf_fill_l2norm_f64(p, n, st),
                   "sf_fill_l2norm_f64")


def stream_copy(src, dst, stream=None):
    with torch.cuda.device(src.device):
        capi.check(capi.lib().sf_stream_copy_f64(_dev_f64(src, "src"), _dev_f64(dst, "dst"),
                                                 src.numel(), _stream(stream, src.device)),
                   "sf_stream_copy_f64")
    return dst


def fill_vecadd(n, device="cuda", stream=None):
    """benchmark02 data (benchmark02/benchmark02.cc:84-85) -> (x, y)."""
    x = torch.empty(n, dtype=torch.float64, device=device)
    y = torch.empty(n, dtype=torch.float64, device=device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().sf_fill_vecadd_f64(_dev_f64(x, "x"), _dev_f64(y, "y"), n,
                                                 _stream(stream, x.device)), "sf_fill_vecadd_f64")
    return x, y


def vector_add(x, y, stream=None):
    """x += y in place (benchmark02's operation)."""
    with torch.cuda.device(x.device):
        capi.check(capi.lib().sf_vector_add_f64(_dev_f64(x, "x"), _dev_f64(y, "y"), x.numel(),
                                                _stream(stream, x.device)), "sf_vector_add_f64")
    return x


def fill_matvec(m, n, device="cuda", stream=None):
    """benchmark03 data -> (A row-major m*n, x)."""
    a = torch.empty(m * n, dtype=torch.float64, device=device)
    x = torch.empty(n, dtype=torch.float64, device=device)
    with torch.cuda.device(a.device):
        capi.check(capi.lib().sf_fill_matvec_f64(_dev_f64(a, "A"), _dev_f64(x, "x"), m, n,
                                                 _stream(stream, a.device)), "sf_fill_matvec_f64")
    return a, x


def matvec(m, n, a, x, y=None, stream=None):
    """y = A x (benchmark03's operation)."""
    if y is None:
        y = torch.empty(m, dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        capi.check(capi.lib().sf_matvec_f64(m, n, _dev_f64(a, "A"), _dev_f64(x, "x"),
                                            _dev_f64(y, "y"), _stream(stream, a.device)), "sf_matvec_f64")
    return y


def device_info():
    cu, wave = ctypes.c_int(0), ctypes.c_int(0)
    name = ctypes.create_string_buffer(256)
    capi.check(capi.lib().sf_device_info(ctypes.byref(cu), ctypes.byref(wave), name, 256),
               "sf_device_info")
    return {"num_cu": cu.value, "wave_size": wave.value, "name": name.value.decode()}
