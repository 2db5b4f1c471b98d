// Diagonal-skewed plain-16-bit NW/HW/OV kernels (FLAG_SKEW), one instantiation per rows-per-lane value 1..40: the
// bulk of a database runs on these, so the query is padded to the next multiple of the lane count only (a 300-row
// query on 8 lanes: 38 rows per lane instead of 40).
#include "dp_kernel.hpp"

template <int MODE, int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<17, MODE, FLAG_SKEW, R + 1>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

KernelFn opal_kernel_s16skew(int mode, int rows) {
    using Seq = std::make_integer_sequence<int, 40>;
    switch (mode) {
    case MODE_NW: return pick<MODE_NW>(rows, Seq{});
    case MODE_HW: return pick<MODE_HW>(rows, Seq{});
    case MODE_OV: return pick<MODE_OV>(rows, Seq{});
    }
    return nullptr;
}

// one-launch variants live in inst_uber_{nw,hw,ov}.hip
UberFn opal_uber_skew_nw(int rows);
UberFn opal_uber_skew_hw(int rows);
UberFn opal_uber_skew_ov(int rows);
UberFn opal_uber_skew(int mode, int rows) {
    switch (mode) {
    case MODE_NW: return opal_uber_skew_nw(rows);
    case MODE_HW: return opal_uber_skew_hw(rows);
    case MODE_OV: return opal_uber_skew_ov(rows);
    }
    return nullptr;
}
