// One-launch Smith-Waterman kernels with end location, bulk rows per lane 1..40.
#include "dp_kernel.hpp"

template <int... R>
static UberFn pick(int rows, std::integer_sequence<int, R...>) {
    static const UberFn table[] = {opal_dp_uber<MODE_SW, FLAG_SWEND, R + 1>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

UberFn opal_uber_swend(int rows) { return pick(rows, std::make_integer_sequence<int, 40>{}); }
