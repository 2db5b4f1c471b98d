// Width-32 kernels (re-runs of saturated width-16 results, and inputs outside the 16-bit range).
// rows-per-lane in steps of 8.
#include "dp_kernel.hpp"

template <int MODE, int FLAGS, int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<32, MODE, FLAGS, 8 * (R + 1)>...};
    return (rows >= 8 && rows <= 8 * (int)sizeof...(R) && rows % 8 == 0) ? table[rows / 8 - 1] : nullptr;
}

KernelFn opal_kernel_w32(int mode, int flags, int rows) {
    using Seq = std::make_integer_sequence<int, 5>;
    switch (mode) {
    case MODE_NW: return pick<MODE_NW, 0>(rows, Seq{});
    case MODE_HW: return pick<MODE_HW, 0>(rows, Seq{});
    case MODE_OV: return pick<MODE_OV, 0>(rows, Seq{});
    case MODE_SW:
        if ((flags & FLAG_SWEND) && (flags & FLAG_WRAP)) return pick<MODE_SW, FLAG_SWEND | FLAG_WRAP>(rows, Seq{});
        if (flags & FLAG_SWEND) return pick<MODE_SW, FLAG_SWEND>(rows, Seq{});
        if (flags & FLAG_WRAP) return pick<MODE_SW, FLAG_WRAP>(rows, Seq{});
        return pick<MODE_SW, 0>(rows, Seq{});
    }
    return nullptr;
}

KernelFn opal_kernel_w32_multi(int mode, int flags, int rows) {
    using Seq = std::make_integer_sequence<int, 5>;
    switch (mode) {
    case MODE_NW: return pick<MODE_NW, FLAG_MULTI>(rows, Seq{});
    case MODE_HW: return pick<MODE_HW, FLAG_MULTI>(rows, Seq{});
    case MODE_OV: return pick<MODE_OV, FLAG_MULTI>(rows, Seq{});
    case MODE_SW:
        if ((flags & FLAG_SWEND) && (flags & FLAG_WRAP)) return pick<MODE_SW, FLAG_SWEND | FLAG_WRAP | FLAG_MULTI>(rows, Seq{});
        if (flags & FLAG_SWEND) return pick<MODE_SW, FLAG_SWEND | FLAG_MULTI>(rows, Seq{});
        if (flags & FLAG_WRAP) return pick<MODE_SW, FLAG_WRAP | FLAG_MULTI>(rows, Seq{});
        return pick<MODE_SW, FLAG_MULTI>(rows, Seq{});
    }
    return nullptr;
}
