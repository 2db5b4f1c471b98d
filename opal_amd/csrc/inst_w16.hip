// Width-16 kernels other than the SW score pass: SW end-location pass, NW, HW, OV.
// rows-per-lane in steps of 4 (pad rows make up the difference).
#include "dp_kernel.hpp"

template <int MODE, int FLAGS, int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<16, MODE, FLAGS, 4 * (R + 1)>...};
    return (rows >= 4 && rows <= 4 * (int)sizeof...(R) && rows % 4 == 0) ? table[rows / 4 - 1] : nullptr;
}

KernelFn opal_kernel_w16(int mode, int flags, int rows) {
    using Seq = std::make_integer_sequence<int, 10>;
    switch (mode) {
    case MODE_NW: return pick<MODE_NW, 0>(rows, Seq{});
    case MODE_HW: return pick<MODE_HW, 0>(rows, Seq{});
    case MODE_OV: return pick<MODE_OV, 0>(rows, Seq{});
    case MODE_SW: return (flags & FLAG_SWEND) ? pick<MODE_SW, FLAG_SWEND>(rows, Seq{}) : opal_kernel_sw16(rows);
    }
    return nullptr;
}

// Multi-chunk variants (queries longer than one wave can hold): every mode, rows in steps of 4.
KernelFn opal_kernel_w16_multi(int mode, int flags, int rows) {
    using Seq = std::make_integer_sequence<int, 10>;
    switch (mode) {
    case MODE_NW: return pick<MODE_NW, FLAG_MULTI>(rows, Seq{});
    case MODE_HW: return pick<MODE_HW, FLAG_MULTI>(rows, Seq{});
    case MODE_OV: return pick<MODE_OV, FLAG_MULTI>(rows, Seq{});
    case MODE_SW: return (flags & FLAG_SWEND) ? pick<MODE_SW, FLAG_SWEND | FLAG_MULTI>(rows, Seq{}) : pick<MODE_SW, FLAG_MULTI>(rows, Seq{});
    }
    return nullptr;
}
