// Plain-16-bit local-alignment kernel WITH end location (single pass), one instantiation per
// rows-per-lane value, like the score-only table in inst_s16sw.hip.
#include "dp_kernel.hpp"

template <int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<17, MODE_SW, FLAG_SWEND, R + 1>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

KernelFn opal_kernel_s16swend(int rows) { return pick(rows, std::make_integer_sequence<int, 40>{}); }
