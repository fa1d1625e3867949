// grb_ops.hpp -- builtin operator arithmetic shared by all kernels (device + host callable).
//
// One switch-based evaluator: kernels specialised on a compile-time operator pass a constant and the
// switch folds away; generic kernels pass the runtime code (a wave-uniform scalar branch).
#pragma once
#include "grb_internal.hpp"

namespace grb {

#define GRB_HD __host__ __device__ __forceinline__

template <typename T> struct Widen { using type = T; };
template <> struct Widen<bool> { using type = int32_t; };
template <> struct Widen<int8_t> { using type = int32_t; };
template <> struct Widen<int16_t> { using type = int32_t; };
template <> struct Widen<uint8_t> { using type = uint32_t; };
template <> struct Widen<uint16_t> { using type = uint32_t; };

template <typename T> struct WrapOf { using type = T; };
template <> struct WrapOf<int8_t> { using type = uint8_t; };
template <> struct WrapOf<int16_t> { using type = uint16_t; };
template <> struct WrapOf<int32_t> { using type = uint32_t; };
template <> struct WrapOf<int64_t> { using type = uint64_t; };

// z = op(a, b).  For T = bool the host canonicalises PLUS/MAX->LOR, TIMES/MIN->LAND, MINUS->LXOR first.
template <typename T>
GRB_HD T apply_binop(int op, T a, T b)
{
    if constexpr (std::is_same<T, bool>::value) {
        switch (op) {
        case OP_FIRST: case OP_ANY: return a;
        case OP_SECOND: return b;
        case OP_PAIR: return true;
        case OP_PLUS: case OP_MAX: case OP_LOR: return a || b;
        case OP_TIMES: case OP_MIN: case OP_LAND: return a && b;
        case OP_MINUS: case OP_RMINUS: case OP_LXOR: return a != b;
        case OP_LXNOR: return a == b;
        default: return false;
        }
    } else {
        using U = typename WrapOf<T>::type;
        switch (op) {
        case OP_FIRST: case OP_ANY: return a;
        case OP_SECOND: return b;
        case OP_PAIR: return (T)1;
        case OP_PLUS: return (T)((U)a + (U)b);
        case OP_MINUS: return (T)((U)a - (U)b);
        case OP_RMINUS: return (T)((U)b - (U)a);
        case OP_TIMES: return (T)((U)a * (U)b);
        case OP_MIN:
            // (fmin / fmax: a NaN operand is ignored -- one v_min / v_max instruction instead of compares and branches)
            if constexpr (std::is_same<T, float>::value) return __builtin_fminf(a, b);
            else if constexpr (std::is_same<T, double>::value) return __builtin_fmin(a, b);
            else return a < b ? a : b;
        case OP_MAX:
            if constexpr (std::is_same<T, float>::value) return __builtin_fmaxf(a, b);
            else if constexpr (std::is_same<T, double>::value) return __builtin_fmax(a, b);
            else return a > b ? a : b;
        case OP_LOR: return (T)((a != (T)0) || (b != (T)0));
        case OP_LAND: return (T)((a != (T)0) && (b != (T)0));
        case OP_LXOR: return (T)((a != (T)0) != (b != (T)0));
        case OP_LXNOR: return (T)((a != (T)0) == (b != (T)0));
        default: return (T)0;
        }
    }
}

template <typename T>
GRB_HD T type_max()
{
    if constexpr (std::is_same<T, float>::value) return __builtin_huge_valf();
    else if constexpr (std::is_same<T, double>::value) return __builtin_huge_val();
    else if constexpr (std::is_signed<T>::value) return (T)((((unsigned long long)1) << (sizeof(T) * 8 - 1)) - 1);
    else return (T)~(T)0;
}
template <typename T>
GRB_HD T type_lowest()
{
    if constexpr (std::is_same<T, float>::value) return -__builtin_huge_valf();
    else if constexpr (std::is_same<T, double>::value) return -__builtin_huge_val();
    else if constexpr (std::is_signed<T>::value) return (T)(-(long long)((((unsigned long long)1) << (sizeof(T) * 8 - 1)) - 1) - 1);
    else return (T)0;
}

// identity of monoid `op` over the VALUE type T, returned in accumulator type W (W = T or Widen<T>)
template <typename T, typename W>
GRB_HD W monoid_identity(int op)
{
    if constexpr (std::is_same<T, bool>::value) {
        switch (op) {
        case OP_LAND: case OP_TIMES: case OP_MIN: case OP_LXNOR: return (W)1;
        default: return (W)0;
        }
    } else {
        switch (op) {
        case OP_TIMES: case OP_LAND: case OP_LXNOR: return (W)1;
        case OP_MIN: return (W)type_max<T>();
        case OP_MAX: return (W)type_lowest<T>();
        default: return (W)0;
        }
    }
}

// monoid terminal ("annihilator") test used for early exit: lor->true, land->false, any->anything
template <typename T>
GRB_HD bool monoid_is_terminal(int op, T v)
{
    switch (op) {
    case OP_ANY: return true;
    case OP_LOR: return v != (T)0;
    case OP_LAND: return v == (T)0;
    default: return false;
    }
}

inline int canonical_op(int type, int op)
{
    if (op >= OP_UNSUPPORTED) fail(GrB_NOT_IMPLEMENTED, "this builtin operator is not implemented by libgrb_mi355x (handle-only: import-time surface)");
    if (type != TC_BOOL) return op;
    switch (op) {
    case OP_PLUS: case OP_MAX: return OP_LOR;
    case OP_TIMES: case OP_MIN: return OP_LAND;
    case OP_MINUS: case OP_RMINUS: return OP_LXOR;
    default: return op;
    }
}

inline int flip_op(int op)  // op'(a,b) = op(b,a)
{
    switch (op) {
    case OP_FIRST: return OP_SECOND;
    case OP_SECOND: return OP_FIRST;
    case OP_MINUS: return OP_RMINUS;
    case OP_RMINUS: return OP_MINUS;
    default: return op;
    }
}

inline size_t type_size(int code)
{
    static const size_t s[TC_COUNT] = {1, 1, 2, 4, 8, 1, 2, 4, 8, 4, 8};
    return s[code];
}

// ---- GraphBLAS typecast (SuiteSparse rules: ->bool is x!=0, float->int saturates with NaN->0) -------
template <typename D, typename S>
GRB_HD D cast_value(S s)
{
    if constexpr (std::is_same<D, S>::value) return s;
    else if constexpr (std::is_same<D, bool>::value) return s != (S)0;
    else if constexpr (std::is_floating_point<S>::value && std::is_integral<D>::value) {
        if (s != s) return (D)0;
        const double x = (double)s;
        if (x <= (double)type_lowest<D>()) return type_lowest<D>();
        if (x >= (double)type_max<D>()) return type_max<D>();
        return (D)x;
    } else return (D)s;
}

// (experiment builds only -- scripts/build_variant_mxm.sh: -DGRB_DISPATCH_ONLY_INT64 compiles the INT64 instantiations alone, a tenth of
//  the compile time of grb_mxm.hip; every other type then fails with GrB_NOT_IMPLEMENTED.  Never defined for the shipped library.)
#ifdef GRB_DISPATCH_ONLY_INT64
#define GRB_DISPATCH_TYPE(code, T, ...)                                               \
    switch (code) {                                                                   \
    case ::grb::TC_INT64: { using T = int64_t; __VA_ARGS__; } break;                  \
    default: ::grb::fail(GrB_NOT_IMPLEMENTED, "experiment build: INT64 only");        \
    }
#else
#define GRB_DISPATCH_TYPE(code, T, ...)                                               \
    switch (code) {                                                                   \
    case ::grb::TC_BOOL: { using T = bool; __VA_ARGS__; } break;                      \
    case ::grb::TC_INT8: { using T = int8_t; __VA_ARGS__; } break;                    \
    case ::grb::TC_INT16: { using T = int16_t; __VA_ARGS__; } break;                  \
    case ::grb::TC_INT32: { using T = int32_t; __VA_ARGS__; } break;                  \
    case ::grb::TC_INT64: { using T = int64_t; __VA_ARGS__; } break;                  \
    case ::grb::TC_UINT8: { using T = uint8_t; __VA_ARGS__; } break;                  \
    case ::grb::TC_UINT16: { using T = uint16_t; __VA_ARGS__; } break;                \
    case ::grb::TC_UINT32: { using T = uint32_t; __VA_ARGS__; } break;                \
    case ::grb::TC_UINT64: { using T = uint64_t; __VA_ARGS__; } break;                \
    case ::grb::TC_FP32: { using T = float; __VA_ARGS__; } break;                     \
    case ::grb::TC_FP64: { using T = double; __VA_ARGS__; } break;                    \
    default: ::grb::fail(GrB_INVALID_OBJECT, "unknown type code");                    \
    }
#endif

// ---- accumulator helpers (device only) ----------------------------------------------------------------
template <typename T, typename W>
__device__ __forceinline__ T from_acc(W v)
{
    if constexpr (std::is_same<T, bool>::value) return v != (W)0;
    else return (T)v;
}

// slot = monoid(slot, v) on an LDS or global word.  plus / min / max / lor / land / lxor map to ONE hardware
// atomic (ds_* / global_atomic_*, no return value, so a lane can keep issuing); times, lxnor and any other
// operator fall back to a compare-and-swap loop.
template <typename W>
__device__ __forceinline__ void atomic_combine_cas(W *slot, W v, int monoid)
{
    if constexpr (sizeof(W) == 4) {
        unsigned int *p = (unsigned int *)slot;
        unsigned int old = *p, assumed;
        do {
            assumed = old;
            const W nw = apply_binop<W>(monoid, __builtin_bit_cast(W, assumed), v);
            old = atomicCAS(p, assumed, __builtin_bit_cast(unsigned int, nw));
        } while (old != assumed);
    } else {
        unsigned long long *p = (unsigned long long *)slot;
        unsigned long long old = *p, assumed;
        do {
            assumed = old;
            const W nw = apply_binop<W>(monoid, __builtin_bit_cast(W, assumed), v);
            old = atomicCAS(p, assumed, __builtin_bit_cast(unsigned long long, nw));
        } while (old != assumed);
    }
}

template <typename W>
__device__ __forceinline__ void atomic_combine(W *slot, W v, int monoid)
{
    if constexpr (std::is_same<W, float>::value || std::is_same<W, double>::value) {
        switch (monoid) {
        case OP_PLUS: atomicAdd(slot, v); return;
        case OP_MIN: if (v == v) atomicMin(slot, v); return;  // fmin semantics: a NaN operand is ignored
        case OP_MAX: if (v == v) atomicMax(slot, v); return;
        default: atomic_combine_cas<W>(slot, v, monoid); return;
        }
    } else if constexpr (std::is_same<W, int32_t>::value) {
        switch (monoid) {
        case OP_PLUS: atomicAdd((int *)slot, (int)v); return;
        case OP_MIN: atomicMin((int *)slot, (int)v); return;
        case OP_MAX: atomicMax((int *)slot, (int)v); return;
        case OP_LOR: if (v != 0) atomicOr((unsigned int *)slot, 1u); return;   // accumulators of lor/land/lxor hold 0/1
        case OP_LAND: if (v == 0) atomicAnd((unsigned int *)slot, 0u); return;
        case OP_LXOR: if (v != 0) atomicXor((unsigned int *)slot, 1u); return;
        default: atomic_combine_cas<W>(slot, v, monoid); return;
        }
    } else if constexpr (std::is_same<W, uint32_t>::value) {
        switch (monoid) {
        case OP_PLUS: atomicAdd((unsigned int *)slot, (unsigned int)v); return;
        case OP_MIN: atomicMin((unsigned int *)slot, (unsigned int)v); return;
        case OP_MAX: atomicMax((unsigned int *)slot, (unsigned int)v); return;
        default: atomic_combine_cas<W>(slot, v, monoid); return;
        }
    } else if constexpr (std::is_same<W, int64_t>::value) {
        switch (monoid) {
        case OP_PLUS: atomicAdd((unsigned long long *)slot, (unsigned long long)v); return;  // two's complement: same bits
        case OP_MIN: atomicMin((long long *)slot, (long long)v); return;
        case OP_MAX: atomicMax((long long *)slot, (long long)v); return;
        default: atomic_combine_cas<W>(slot, v, monoid); return;
        }
    } else if constexpr (std::is_same<W, uint64_t>::value) {
        switch (monoid) {
        case OP_PLUS: atomicAdd((unsigned long long *)slot, (unsigned long long)v); return;
        case OP_MIN: atomicMin((unsigned long long *)slot, (unsigned long long)v); return;
        case OP_MAX: atomicMax((unsigned long long *)slot, (unsigned long long)v); return;
        default: atomic_combine_cas<W>(slot, v, monoid); return;
        }
    } else {
        atomic_combine_cas<W>(slot, v, monoid);
    }
}

// ---- ordered-integer form of floating-point accumulators -------------------------------------------------------------
// gfx950 has no 32-bit floating-point min / max atomic in global memory (atomicMin(float *) is a compare-and-swap loop with a
// returned value); unsigned min / max are single fire-and-forget instructions.  The map below is strictly increasing from the
// floats (NaN excluded: a NaN product is not emitted) to the unsigned integers, so MIN / MAX accumulators of the class-strip
// kernel are kept in this form: initialised by k_long_init, combined by atomic_combine_ord, read back by acc_from_ord.
__device__ __forceinline__ uint32_t ord_of(float v)
{
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_to_f32(uint32_t e)
{
    return __builtin_bit_cast(float, (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}
__device__ __forceinline__ uint64_t ord_of(double v)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord_to_f64(uint64_t e)
{
    return __builtin_bit_cast(double, (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e);
}
template <typename W> GRB_HD constexpr bool acc_is_ordered(int monoid)
{
    return (std::is_same<W, float>::value || std::is_same<W, double>::value) && (monoid == OP_MIN || monoid == OP_MAX);
}
// the stored form of accumulator value v / the value of a stored accumulator (ord = acc_is_ordered for the call's monoid)
template <typename W> __device__ __forceinline__ W acc_to_stored(W v, bool ord)
{
    if constexpr (std::is_same<W, float>::value) return ord ? __builtin_bit_cast(float, ord_of(v)) : v;
    else if constexpr (std::is_same<W, double>::value) return ord ? __builtin_bit_cast(double, ord_of(v)) : v;
    else return v;
}
template <typename W> __device__ __forceinline__ W acc_from_stored(W s, bool ord)
{
    if constexpr (std::is_same<W, float>::value) return ord ? ord_to_f32(__builtin_bit_cast(uint32_t, s)) : s;
    else if constexpr (std::is_same<W, double>::value) return ord ? ord_to_f64(__builtin_bit_cast(uint64_t, s)) : s;
    else return s;
}
template <typename W> __device__ __forceinline__ void atomic_combine_ord(W *slot, W v, int monoid)
{
    if constexpr (std::is_same<W, float>::value) {
        if (monoid == OP_MIN || monoid == OP_MAX) {
            if (v != v) return;
            if (monoid == OP_MIN) atomicMin((unsigned int *)slot, ord_of(v));
            else atomicMax((unsigned int *)slot, ord_of(v));
            return;
        }
    } else if constexpr (std::is_same<W, double>::value) {
        if (monoid == OP_MIN || monoid == OP_MAX) {
            if (v != v) return;
            if (monoid == OP_MIN) atomicMin((unsigned long long *)slot, (unsigned long long)ord_of(v));
            else atomicMax((unsigned long long *)slot, (unsigned long long)ord_of(v));
            return;
        }
    }
    atomic_combine<W>(slot, v, monoid);
}

// ---- presence-bit helpers -------------------------------------------------------------------------
GRB_HD bool bit_test(const uint32_t *bits, int64_t i) { return (bits[i >> 5] >> (i & 31)) & 1u; }

}  // namespace grb
