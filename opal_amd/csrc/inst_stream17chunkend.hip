// Streaming SW kernels with end locations for queries beyond one wave, width 17: one launch per row chunk
// (stream_kernel.hpp, CHUNK + SWEND). rows per lane 16..32, without 31: that one shape keeps a lane value in scratch across the
// column loop (one register short); a chunk of 31 rows per lane runs on the 32-row kernel.
#include "stream_kernel.hpp"

template <int S>
static StreamFn chunk_end_kernel() {
    if constexpr (S == 31) return nullptr;
    else return opal_sw_stream<17, S, true, true>;
}
template <int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {chunk_end_kernel<R + 16>()...};
    return (rows >= 16 && rows < 16 + (int)sizeof...(R)) ? table[rows - 16] : nullptr;
}

StreamFn opal_stream_s17chunkend(int rows) { return pick(rows, std::make_integer_sequence<int, 17>{}); }
