// Role-swapped SW score kernels (swap_kernel.hpp): the target on the rows, 64 lanes x 8 / 16 / 24 / 40 residues per row chunk.
#include "swap_kernel.hpp"

SwapFn opal_swap_kernel(int rows, int want_end) {
    if (want_end) return rows == 8 ? (SwapFn)opal_swap_sw<8, true> : rows == 16 ? (SwapFn)opal_swap_sw<16, true> : nullptr;
    switch (rows) {
    case 8: return opal_swap_sw<8, false>;
    case 16: return opal_swap_sw<16, false>;
    case 24: return opal_swap_sw<24, false>;
    case 40: return opal_swap_sw<40, false>;
    }
    return nullptr;
}
size_t opal_swap_lds(int rows, int Q, int A) {
    switch (rows) {
    case 8: return opal_swap_lds_bytes<8>(Q, A);
    case 16: return opal_swap_lds_bytes<16>(Q, A);
    case 24: return opal_swap_lds_bytes<24>(Q, A);
    case 40: return opal_swap_lds_bytes<40>(Q, A);
    }
    return (size_t)1 << 30;
}

SwapFn opal_swap_gl_kernel(int mode, int rows) {
    if (rows == 8) return mode == MODE_NW ? (SwapFn)opal_swap_gl<MODE_NW, 8, 4> : mode == MODE_HW ? (SwapFn)opal_swap_gl<MODE_HW, 8, 4> : mode == MODE_OV ? (SwapFn)opal_swap_gl<MODE_OV, 8, 4> : nullptr;
    if (rows == 16) return mode == MODE_NW ? (SwapFn)opal_swap_gl<MODE_NW, 16, 2> : mode == MODE_HW ? (SwapFn)opal_swap_gl<MODE_HW, 16, 2> : mode == MODE_OV ? (SwapFn)opal_swap_gl<MODE_OV, 16, 2> : nullptr;
    return nullptr;
}
size_t opal_swap_gl_lds(int rows, int Q, int A) {
    return rows == 8 ? opal_swapgl_lds_bytes<8, 4>(Q, A) : rows == 16 ? opal_swapgl_lds_bytes<16, 2>(Q, A) : ((size_t)1 << 30);
}
