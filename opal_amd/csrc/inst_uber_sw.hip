// One-launch ("uber") Smith-Waterman score kernels, one instantiation per bulk rows-per-lane value 1..40 (each holds
// the bodies for S0, ceil(S0/2) and ceil(S0/4) rows per lane).
#include "dp_kernel.hpp"

template <int... R>
static UberFn pick(int rows, std::integer_sequence<int, R...>) {
    static const UberFn table[] = {opal_dp_uber<MODE_SW, 0, R + 1>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

UberFn opal_uber_swend(int rows);
UberFn opal_uber_swzc(int rows);

UberFn opal_uber_sw(int flags, int rows) {
    if (flags == FLAG_ZC) return opal_uber_swzc(rows);
    if (flags & ~FLAG_SWEND) return nullptr;
    return (flags & FLAG_SWEND) ? opal_uber_swend(rows) : pick(rows, std::make_integer_sequence<int, 40>{});
}
