// Plain-16-bit kernels with 2-byte profile entries (width 18), multi-chunk variants: queries beyond one wave
// (more than 64 lanes x 40 rows) run as consecutive row chunks (dp_kernel.hpp, FLAG_MULTI); with 2-byte entries a
// chunk's profile takes half the LDS, so twice the waves share a CU. rows-per-lane 8, 16, 24, 32.
#include "dp_kernel.hpp"

template <int MODE, int FLAGS, int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<18, MODE, FLAGS, 8 * (R + 1)>...};
    return (rows >= 8 && rows <= 8 * (int)sizeof...(R) && rows % 8 == 0) ? table[rows / 8 - 1] : nullptr;
}

KernelFn opal_kernel_s18_multi(int mode, int flags, int rows) {
    using Seq = std::make_integer_sequence<int, 4>;  // (40 rows per lane do not fit 128 registers with the chunk boundary in flight)
    switch (mode) {
    case MODE_NW: return pick<MODE_NW, FLAG_MULTI>(rows, Seq{});
    // (HW at 24 rows and more: the compiler leaves a row block's profile values in scratch memory -- row chunks of HW
    // searches run at width 17, ladder.cpp)
    case MODE_HW: return pick<MODE_HW, FLAG_MULTI>(rows, std::make_integer_sequence<int, 2>{});
    case MODE_OV: return pick<MODE_OV, FLAG_MULTI>(rows, Seq{});
    case MODE_SW: return (flags & FLAG_SWEND) ? pick<MODE_SW, FLAG_SWEND | FLAG_MULTI>(rows, Seq{}) : pick<MODE_SW, FLAG_MULTI>(rows, Seq{});
    }
    return nullptr;
}
