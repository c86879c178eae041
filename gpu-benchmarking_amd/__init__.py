"""gpu_benchmarking_amd -- MI355X-native BwdTrans sum-factorisation hot path.

The directory is named ``gpu-benchmarking_amd`` (not importable by name); load it with
``__graft_entry__.load_package()`` which registers it as ``gpu_benchmarking_amd``.

Layout
  csrc/   HIP kernels (gfx950) + the C ABI of lib/libsumfact.so (include/sumfact.h)
  host/   C++ drivers benchmark01/04/05 with the reference's run_test<T> signature, CLI and log grammar
  tools/  sf_tune (variant sweep), run.sh-style sweeps
  capi.py / bwdtrans.py / logfmt.py / shard.py : Python plumbing over the C ABI (tests, bench.py)

There is NO CPU fallback here: every compute entry point goes through libsumfact.so and raises if
it is missing.
"""
from . import capi, logfmt, shard  # noqa: F401
from .bwdtrans import (  # noqa: F401
    bwdtrans_hex, bwdtrans_quad, sumsq, fill_sincos, fill_basis, fill_random, fill_l2norm,
    stream_copy, device_info, interleave64, bwdtrans_hex_interleaved, fill_vecadd, vector_add, fill_matvec, matvec, hex_wsp_doubles, quad_wsp_doubles, VARIANTS,
)
