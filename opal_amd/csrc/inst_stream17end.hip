// Streaming SW kernels with end locations, width 17 (4-byte profile entries): rows per lane 1..36.
#include "stream_kernel.hpp"

template <int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {opal_sw_stream<17, R + 1, true>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

StreamFn opal_stream_s17end(int rows) { return pick(rows, std::make_integer_sequence<int, 36>{}); }
