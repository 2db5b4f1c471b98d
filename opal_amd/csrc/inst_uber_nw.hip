// One-launch diagonal-skewed NW kernels, bulk rows per lane 1..40.
#include "dp_kernel.hpp"

// (30 bulk rows: the one shape for which the register allocator spills four values at the 128-register budget of the
// one-launch kernels -- `make check-scratch`; that shape runs its length classes as separate launches instead)
template <int R> struct UberNW { static constexpr UberFn fn = opal_dp_uber<0, FLAG_SKEW, R>; };
template <> struct UberNW<30> { static constexpr UberFn fn = nullptr; };

template <int... R>
static UberFn pick(int rows, std::integer_sequence<int, R...>) {
    static const UberFn table[] = {UberNW<R + 1>::fn...};
    return (rows >= 1 && rows <= (int)sizeof...(R)) ? table[rows - 1] : nullptr;
}

UberFn opal_uber_skew_nw(int rows) { return pick(rows, std::make_integer_sequence<int, 40>{}); }
