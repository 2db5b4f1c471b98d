// Streaming SW score kernels for queries beyond one wave, width 17 (4-byte profile entries): one launch per row chunk,
// the chunk's bottom row through HBM (stream_kernel.hpp, CHUNK). rows per lane 16..40 without 38 (one register short of
// 128: the 39-row shape is scheduled differently and fits).
#include "stream_kernel.hpp"

constexpr int chunk_rows_of(int i) { return i + 16 < 38 ? i + 16 : i + 17; }

template <int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {opal_sw_stream<17, chunk_rows_of(R), false, true>...};
    if (rows < 16 || rows > 40 || rows == 38) return nullptr;
    return table[rows < 38 ? rows - 16 : rows - 17];
}

StreamFn opal_stream_s17chunk(int rows) { return pick(rows, std::make_integer_sequence<int, 24>{}); }
