// Streaming NW/HW/OV kernels for queries beyond one wave, width 17: one launch per row chunk, the chunk's bottom row through
// HBM as true values (stream_kernel.hpp, opal_gl_stream CHUNK). rows per lane 24..40.
#include "stream_kernel.hpp"

// (shapes that do not fit 128 registers are left out -- `make check-scratch`; the scheduler takes the next larger rows value
// or another geometry)
template <int MODE, int ROWS> struct GlFn { static constexpr StreamFn fn = opal_gl_stream<17, MODE, ROWS, true>; };
template <> struct GlFn<MODE_OV, 38> { static constexpr StreamFn fn = nullptr; };
template <> struct GlFn<MODE_OV, 39> { static constexpr StreamFn fn = nullptr; };
template <> struct GlFn<MODE_OV, 40> { static constexpr StreamFn fn = nullptr; };

template <int MODE, int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {GlFn<MODE, R + 24>::fn...};
    return (rows >= 24 && rows < 24 + (int)sizeof...(R)) ? table[rows - 24] : nullptr;
}

StreamFn opal_stream_gl17chunk(int mode, int rows) {
    using Seq = std::make_integer_sequence<int, 17>;
    if (mode == MODE_NW) return pick<MODE_NW>(rows, Seq{});
    if (mode == MODE_HW) return pick<MODE_HW>(rows, Seq{});
    if (mode == MODE_OV) return pick<MODE_OV>(rows, Seq{});
    return nullptr;
}
