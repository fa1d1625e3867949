// TEST INFRASTRUCTURE ONLY -- host stand-in for the two rocPRIM entry points grb_prim.hip uses,
// so that the CPU emulator build (tests/emu/build_emu.sh) links.  Semantics match rocPRIM: the
// radix sort is stable and orders ONLY by key bits [begin_bit, end_bit).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {

template <typename T>
struct plus {
    T operator()(const T &a, const T &b) const { return a + b; }
};

template <typename K, typename V>
hipError_t radix_sort_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, size_t n,
                            unsigned begin_bit, unsigned end_bit, hipStream_t)
{
    if (!tmp) {
        bytes = 16;
        return hipSuccess;
    }
    const K mask = (end_bit - begin_bit >= 8 * sizeof(K)) ? ~(K)0 : ((((K)1 << (end_bit - begin_bit)) - 1) << begin_bit);
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return (kin[a] & mask) < (kin[b] & mask); });
    std::vector<K> k2(n);
    std::vector<V> v2(n);
    for (size_t i = 0; i < n; i++) {
        k2[i] = kin[idx[i]];
        v2[i] = vin[idx[i]];
    }
    std::copy(k2.begin(), k2.end(), kout);
    std::copy(v2.begin(), v2.end(), vout);
    return hipSuccess;
}

template <typename T, typename Init, typename Op>
hipError_t exclusive_scan(void *tmp, size_t &bytes, const T *in, T *out, Init init, size_t n, Op op, hipStream_t)
{
    if (!tmp) {
        bytes = 16;
        return hipSuccess;
    }
    T acc = (T)init;
    for (size_t i = 0; i < n; i++) {
        const T v = in[i];
        out[i] = acc;
        acc = op(acc, v);
    }
    return hipSuccess;
}

}  // namespace rocprim
