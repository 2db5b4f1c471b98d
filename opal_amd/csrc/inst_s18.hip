// Plain-16-bit kernels with 2-byte profile entries (width 18): long queries whose 4-byte profile does not fit LDS.
// rows-per-lane in steps of 8 (one ds_read_b128 = 8 rows); the multi-chunk variants are in inst_s18multi.hip.
#include "dp_kernel.hpp"

template <int MODE, int FLAGS, int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<18, MODE, FLAGS, 8 * (R + 1)>...};
    return (rows >= 8 && rows <= 8 * (int)sizeof...(R) && rows % 8 == 0) ? table[rows / 8 - 1] : nullptr;
}

KernelFn opal_kernel_s18(int mode, int flags, int rows) {
    using Seq = std::make_integer_sequence<int, 5>;
    if (flags & FLAG_MULTI) return opal_kernel_s18_multi(mode, flags, rows);
    if (flags & FLAG_SKEW) return nullptr;
    switch (mode) {
    case MODE_NW: return pick<MODE_NW, 0>(rows, Seq{});
    case MODE_HW: return pick<MODE_HW, 0>(rows, Seq{});
    case MODE_OV: return pick<MODE_OV, 0>(rows, Seq{});
    case MODE_SW: return (flags & FLAG_SWEND) ? pick<MODE_SW, FLAG_SWEND>(rows, Seq{}) : pick<MODE_SW, 0>(rows, Seq{});
    }
    return nullptr;
}
