// Scalar-16-bit kernels other than the SW score pass (SW end-location pass, NW, HW, OV) and the
// multi-chunk variants of all of them. rows-per-lane in steps of 4.
#include "dp_kernel.hpp"

template <int MODE, int FLAGS, int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<17, MODE, FLAGS, 4 * (R + 1)>...};
    return (rows >= 4 && rows <= 4 * (int)sizeof...(R) && rows % 4 == 0) ? table[rows / 4 - 1] : nullptr;
}

KernelFn opal_kernel_s16(int mode, int flags, int rows) {
    using Seq = std::make_integer_sequence<int, 10>;
    const bool multi = flags & FLAG_MULTI, end = flags & FLAG_SWEND, skew = (flags & FLAG_SKEW) && !multi;
    switch (mode) {
    case MODE_NW: return multi ? pick<MODE_NW, FLAG_MULTI>(rows, Seq{}) : skew ? opal_kernel_s16skew(mode, rows) : pick<MODE_NW, 0>(rows, Seq{});
    case MODE_HW: return multi ? pick<MODE_HW, FLAG_MULTI>(rows, Seq{}) : skew ? opal_kernel_s16skew(mode, rows) : pick<MODE_HW, 0>(rows, Seq{});
    case MODE_OV: return multi ? pick<MODE_OV, FLAG_MULTI>(rows, Seq{}) : skew ? opal_kernel_s16skew(mode, rows) : pick<MODE_OV, 0>(rows, Seq{});
    case MODE_SW:
        if (multi) return end ? pick<MODE_SW, FLAG_SWEND | FLAG_MULTI>(rows, Seq{}) : pick<MODE_SW, FLAG_MULTI>(rows, Seq{});
        return end ? opal_kernel_s16swend(rows) : opal_kernel_s16sw(rows);
    }
    return nullptr;
}
