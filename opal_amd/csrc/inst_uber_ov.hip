// One-launch diagonal-skewed OV kernels, bulk rows per lane 1..40.
#include "dp_kernel.hpp"

template <int... R>
static UberFn pick(int rows, std::integer_sequence<int, R...>) {
    static const UberFn table[] = {opal_dp_uber<2, FLAG_SKEW, R + 1>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

UberFn opal_uber_skew_ov(int rows) { return pick(rows, std::make_integer_sequence<int, 40>{}); }
