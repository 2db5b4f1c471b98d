// Streaming NW/HW/OV kernels (diagonal-skewed), width 17 (4-byte profile entries): rows per lane 1..40.
#include "stream_kernel.hpp"

// (shapes that do not fit 128 registers are left out -- `make check-scratch`; the scheduler takes the next larger rows value,
// another geometry or the class kernels)
template <int MODE, int ROWS> struct GlFn { static constexpr StreamFn fn = opal_gl_stream<17, MODE, ROWS>; };
template <> struct GlFn<MODE_OV, 40> { static constexpr StreamFn fn = nullptr; };
template <> struct GlFn<MODE_OV, 39> { static constexpr StreamFn fn = nullptr; };

template <int MODE, int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {GlFn<MODE, R + 1>::fn...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

StreamFn opal_stream_gl17(int mode, int rows) {
    using Seq = std::make_integer_sequence<int, 40>;
    if (mode == MODE_NW) return pick<MODE_NW>(rows, Seq{});
    if (mode == MODE_HW) return pick<MODE_HW>(rows, Seq{});
    if (mode == MODE_OV) return pick<MODE_OV>(rows, Seq{});
    return nullptr;
}
