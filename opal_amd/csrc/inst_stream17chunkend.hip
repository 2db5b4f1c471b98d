// Streaming SW kernels with end locations for queries beyond one wave, width 17: one launch per row chunk
// (stream_kernel.hpp, CHUNK + SWEND). rows per lane 16..32.
#include "stream_kernel.hpp"

template <int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {opal_sw_stream<17, R + 16, true, true>...};
    return (rows >= 16 && rows < 16 + (int)sizeof...(R)) ? table[rows - 16] : nullptr;
}

StreamFn opal_stream_s17chunkend(int rows) { return pick(rows, std::make_integer_sequence<int, 17>{}); }
