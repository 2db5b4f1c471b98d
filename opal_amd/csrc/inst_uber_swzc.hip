// One-launch Smith-Waterman score kernels of a ZERO-COPY search (FLAG_ZC): the first search of a database reads the
// residues from the pinned staging buffer over PCIe and stores each one to the device copy, so the bytes cross the bus
// once and nothing waits for an upload (ladder.cpp, sliced launches). Own instantiations: the extra store would cost
// every other search 2 %. Bulk rows per lane 1..38 (the 39- and 40-row shapes have no register for the store's address).
#include "dp_kernel.hpp"

template <int... R>
static UberFn pick(int rows, std::integer_sequence<int, R...>) {
    static const UberFn table[] = {opal_dp_uber<MODE_SW, FLAG_ZC, R + 1>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

UberFn opal_uber_swzc(int rows) { return pick(rows, std::make_integer_sequence<int, OPAL_ZERO_COPY_MAX_ROWS>{}); }
