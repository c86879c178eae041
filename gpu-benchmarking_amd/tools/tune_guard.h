// tune_guard.h -- host-side bounds check shared by the tuning tools.
//
// Round 1 shipped a GPU memory access fault from tools/sf_tune_f32 (profiles/r01/tune_f32_chunk_size_2d.log:45):
// its buffers were sized for hex nq = 8 (343 / 512 scalars per element, a 256-float basis) and the quad nq = 20 / 24
// rows read 361 / 529 inputs, wrote 400 / 576 outputs per element and filled 380 / 552 basis entries.  The product
// kernels were never at fault; the harness was.  Every tool now (1) sizes its buffers from the maximum over the cases
// it instantiates and (2) declares what a case touches before launching it: a case that does not fit is reported
// and skipped, never launched.
#pragma once
#include <cstddef>
#include <cstdio>

namespace tune
{
struct Capacity
{
    size_t in_bytes = 0, out_bytes = 0, basis_bytes = 0;
};
inline Capacity &capacity()
{
    static Capacity c;
    return c;
}
// true when a case reading `in`, writing `out` and using a basis of `basis` bytes stays inside the allocations
inline bool fits(const char *label, size_t in, size_t out, size_t basis)
{
    const Capacity &c = capacity();
    if (in <= c.in_bytes && out <= c.out_bytes && basis <= c.basis_bytes)
        return true;
    std::printf("%-40s SKIPPED: touches in %zu / out %zu / basis %zu bytes, allocated %zu / %zu / %zu\n", label, in,
                out, basis, c.in_bytes, c.out_bytes, c.basis_bytes);
    std::fflush(stdout);
    return false;
}
constexpr size_t ipow(size_t b, int e)
{
    return e == 0 ? 1 : b * ipow(b, e - 1);
}
} // namespace tune
