// One pair over several waves (pair_kernel.hpp): NW / HW / OV at 2, 4 and 8 rows per lane.
#include "pair_kernel.hpp"

template <int MODE>
static PairFn pick(int rows) {
    return rows == 2 ? opal_pair_kernel<MODE, 2> : rows == 4 ? opal_pair_kernel<MODE, 4> : rows == 8 ? opal_pair_kernel<MODE, 8> : nullptr;
}

PairFn opal_pair_fn(int mode, int rows) {
    switch (mode) {
    case MODE_NW: return pick<MODE_NW>(rows);
    case MODE_HW: return pick<MODE_HW>(rows);
    case MODE_OV: return pick<MODE_OV>(rows);
    }
    return nullptr;
}
