// Streaming SW kernels with end locations, width 18 (2-byte profile entries): rows per lane 20..36.
#include "stream_kernel.hpp"

template <int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {opal_sw_stream<18, R + 20, true>...};
    return (rows >= 20 && rows < 20 + (int)sizeof...(R)) ? table[rows - 20] : nullptr;
}

StreamFn opal_stream_s18end(int rows) { return pick(rows, std::make_integer_sequence<int, 17>{}); }
