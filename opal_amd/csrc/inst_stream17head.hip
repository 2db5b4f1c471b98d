// Streaming SW kernels of width 17 that carry the class body for side items (stream_kernel.hpp, SIDE): the longest
// sequences of a long-tailed database run on 4G lanes as the first workgroups of the launch, the rest streams behind them.
// Instantiations of their own: inside the plain streaming kernels the extra body cost the C4 search 1.1 %.
#include "stream_kernel.hpp"

template <bool END, int... R>
static StreamFn pick(int rows, std::integer_sequence<int, R...>) {
    static const StreamFn table[] = {opal_sw_stream<17, R + 1, END, false, true>...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

StreamFn opal_stream_s17head(int rows, int want_end) {
    return want_end ? pick<true>(rows, std::make_integer_sequence<int, 36>{}) : pick<false>(rows, std::make_integer_sequence<int, STREAM_SW_SIDE_MAX_ROWS>{});
}
