// Scalar-16-bit local-alignment score kernel (one sequence per lane, full-rate VALU ops), one
// instantiation per rows-per-lane value.
#include "dp_kernel.hpp"

template <int... R>
static KernelFn pick(int rows, std::integer_sequence<int, R...>) {
    static const KernelFn table[] = {opal_dp_kernel<17, MODE_SW, 0, R + 1>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

KernelFn opal_kernel_s16sw(int rows) { return pick(rows, std::make_integer_sequence<int, 40>{}); }
