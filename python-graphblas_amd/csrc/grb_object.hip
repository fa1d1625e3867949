// grb_object.hip -- Matrix / Vector lifetime, ingress (build, import) and egress (extractTuples,
// export) on the device, transpose cache, typecasts.
//
// Reference call sites this file serves (paths relative to /root/reference):
//   GrB_Matrix_new/free            graphblas/core/matrix.py:190-225      GrB_Vector_new/free  core/vector.py:159-191
//   GrB_Matrix_build_T             core/matrix.py:627-681 (uint64 indices :637-638; dup_op handling :657-681)
//   GrB_Matrix_import_T/export_T   core/matrix.py:1046-1068, 1601-1645
//   GrB_Matrix_extractTuples_T     core/matrix.py:561-578                GrB_Vector_extractTuples_T core/vector.py:~470
//   GrB_*_nvals / _wait / _error   core/matrix.py:493, 764-789; exceptions.py:171-189
#include <algorithm>

#include "grb_internal.hpp"
#include "grb_ops.hpp"

namespace grb {

static constexpr int BS = 256;
static inline dim3 grid_for(int64_t n, int per_block = BS)
{
    int64_t g = ceil_div(n, per_block);
    if (g < 1) g = 1;
    if (g > 0x7fffffff) fail(GrB_NOT_IMPLEMENTED, "problem too large for one launch");
    return dim3((unsigned)g);
}
#define LAUNCH(kernel, n, ...) hipLaunchKernelGGL(kernel, grid_for(n), dim3(BS), 0, ctx().stream, __VA_ARGS__)

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// grid-stride; one atomic per workgroup (thousands of wavefronts adding to one address serialise for tens of microseconds)
__global__ void k_popcount_sum(const uint64_t *bits, int64_t nwords, unsigned long long *out)
{
    __shared__ unsigned long long s_part[BS / 64];
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x)
        c += (unsigned long long)__popcll(bits[i]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s_part[w];
        if (t) atomicAdd(out, t);
    }
}

__global__ void k_word_popcounts(const uint64_t *bits, int64_t nwords, int64_t *cnt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) cnt[i] = __popcll(bits[i]);
}

template <typename T>
__global__ void k_vec_extract(const uint64_t *bits, const T *val, int64_t nwords, const int64_t *offs, uint64_t *I, T *X)
{
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint64_t b = bits[w];
    int64_t o = offs[w];
    while (b) {
        int t = __ffsll((unsigned long long)b) - 1;
        b &= b - 1;
        int64_t i = w * 64 + t;
        if (I) I[o] = (uint64_t)i;
        if (X) X[o] = val[i];
        o++;
    }
}

// keys for (row, col) tuples; err |= 1 on out-of-bounds
__global__ void k_make_keys(const uint64_t *I, const uint64_t *J, int64_t n, uint64_t nrows, uint64_t ncols, int cshift,
                            uint64_t *keys, uint32_t *perm, int *err)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t r = I[i], c = J ? J[i] : 0;
    if (r >= nrows || c >= ncols) { *err = 1; r = 0; c = 0; }
    keys[i] = (r << cshift) | c;
    perm[i] = (uint32_t)i;
}

__global__ void k_mark_heads(const uint64_t *keys, int64_t n, int64_t *flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one thread per sorted tuple; segment heads fold their duplicates in input order with dup_op
template <typename T>
__global__ void k_dedupe(const uint64_t *keys, const uint32_t *perm, const T *X, int64_t n, const int64_t *excl,
                         int dup_op, uint64_t *out_keys, T *out_vals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool head = (i == 0) || keys[i] != keys[i - 1];
    if (!head) return;
    T acc = X[perm[i]];
    for (int64_t j = i + 1; j < n && keys[j] == keys[i]; j++) acc = apply_binop<T>(dup_op, acc, X[perm[j]]);
    const int64_t o = excl[i];
    out_keys[o] = keys[i];
    out_vals[o] = acc;
}

__global__ void k_rowptr_from_keys(const uint64_t *keys, int64_t nuniq, int64_t nrows, int cshift, int64_t *rowptr)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    const uint64_t target = (uint64_t)r << cshift;  // first key of row r
    int64_t lo = 0, hi = nuniq;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    rowptr[r] = lo;
}

__global__ void k_cols_from_keys(const uint64_t *keys, int64_t n, int cshift, int32_t *col)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) col[i] = (int32_t)(keys[i] & ((1ull << cshift) - 1));
}

// row index of every stored entry: rows[p] = largest r with rowptr[r] <= p
__device__ __forceinline__ int64_t row_of_pos(const int64_t *rowptr, int64_t nrows, int64_t p)
{
    int64_t lo = 0, hi = nrows;  // invariant: rowptr[lo] <= p < rowptr[hi]
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ void k_expand_rows_u64(const int64_t *rowptr, int64_t nrows, int64_t nnz, uint64_t *rows)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) rows[p] = (uint64_t)row_of_pos(rowptr, nrows, p);
}

// transpose keys: (col << rshift) | row
__global__ void k_transpose_keys(const int64_t *rowptr, const int32_t *col, int64_t nrows, int64_t nnz, int rshift,
                                 uint64_t *keys, uint32_t *perm)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    keys[p] = ((uint64_t)(uint32_t)col[p] << rshift) | (uint64_t)row_of_pos(rowptr, nrows, p);
    perm[p] = (uint32_t)p;
}

template <typename T>
__global__ void k_gather(const T *src, const uint32_t *perm, int64_t n, T *dst)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

template <typename T>
__global__ void k_fill(T *dst, int64_t n, const T *one)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = one[0];
}

template <typename T>
__device__ __forceinline__ bool bits_equal(T a, T b)
{
    if constexpr (sizeof(T) == 1) return *(const uint8_t *)&a == *(const uint8_t *)&b;
    else if constexpr (sizeof(T) == 2) return *(const uint16_t *)&a == *(const uint16_t *)&b;
    else if constexpr (sizeof(T) == 4) return *(const uint32_t *)&a == *(const uint32_t *)&b;
    else return *(const uint64_t *)&a == *(const uint64_t *)&b;
}

template <typename T>
__global__ void k_iso_check(const T *val, int64_t n, int *not_iso)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && i > 0) {
        // bitwise comparison (NaN == NaN here; +0 != -0): iso means "one stored bit pattern"
        if (!bits_equal<T>(val[i], val[0])) *not_iso = 1;
    }
}

__global__ void k_i32_to_u64(const int32_t *src, int64_t n, uint64_t *dst)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint64_t)(uint32_t)src[i];
}
__global__ void k_i64_to_u64(const int64_t *src, int64_t n, uint64_t *dst)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint64_t)src[i];
}

template <typename D, typename S>
__global__ void k_cast(D *dst, const S *src, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = cast_value<D, S>(src[i]);
}

// scatter sorted unique (index, value) pairs into a dense-with-presence vector
template <typename T>
__global__ void k_vec_scatter(const uint64_t *idx, const T *vals, int64_t n, T *d_val, uint64_t *d_bits)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = idx[i];
    d_val[k] = vals[i];
    atomicOr((unsigned long long *)&d_bits[k >> 6], 1ull << (k & 63));
}

// out word = present & (value != 0); one wave produces one 64-bit word with a ballot
template <typename T>
__global__ void k_mask_bits(const uint64_t *bits, const T *val, int64_t n, uint64_t *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // blockDim is a multiple of 64
    bool p = false;
    if (i < n) {
        const uint64_t w = bits[i >> 6];  // (wave-uniform: an empty presence word -- most of a sparse frontier -- loads no values)
        if (w) p = ((w >> (i & 63)) & 1ull) && (val[i] != (T)0);
    }
    unsigned long long b = __ballot(p);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (int64_t)((n + 63) / 64)) out[i >> 6] = b;
}

// bits for "all n present"
__global__ void k_full_bits(uint64_t *bits, int64_t n)
{
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nwords = (n + 63) / 64;
    if (w >= nwords) return;
    uint64_t b = ~0ull;
    if (w == nwords - 1 && (n & 63)) b = (1ull << (n & 63)) - 1;
    bits[w] = b;
}

// ---------------------------------------------------------------------------------------------------
// typecast
// ---------------------------------------------------------------------------------------------------
void cast_array(int dst_type, void *dst, int src_type, const void *src, int64_t n)
{
    if (n <= 0) return;
    if (dst_type == src_type) {
        d2d(dst, src, (size_t)n * type_size(dst_type));
        return;
    }
    GRB_DISPATCH_TYPE(dst_type, D, {
        GRB_DISPATCH_TYPE(src_type, S, { LAUNCH((k_cast<D, S>), n, (D *)dst, (const S *)src, n); })
    })
}

// ---------------------------------------------------------------------------------------------------
// Vector
// ---------------------------------------------------------------------------------------------------
GB_Vector_opaque *vector_new(GrB_Type type, uint64_t n)
{
    auto *v = new GB_Vector_opaque();
    v->magic = MAGIC_VECTOR;
    v->type = type;
    v->n = n;
    v->d_val = nullptr;
    v->d_bits = nullptr;
    v->padded = false;
    v->nvals = 0;
    return v;
}

// storage of a vector: values + presence words, both optionally behind a front pad (see GB_Vector_opaque::padded)
void vector_alloc_pair(const GB_Vector_opaque *v, bool padded, bool zero_val, void **val, uint64_t **bits)
{
    const size_t vb = (size_t)v->n * v->type->size, bb = bits_words64(v->n) * 8;
    const size_t vpad = padded ? VEC_VAL_PAD : 0, bpad = padded ? VEC_BITS_PAD : 0;
    char *pv = (char *)dev_alloc(vb + vpad);
    char *pb = nullptr;
    try {
        pb = (char *)dev_alloc(bb + bpad);
    } catch (...) {
        dev_free(pv);
        throw;
    }
    if (zero_val) GRB_HIP(hipMemsetAsync(pv + vpad, 0, vb ? vb : 16, ctx().stream));
    GRB_HIP(hipMemsetAsync(pb + bpad, 0, bb ? bb : 16, ctx().stream));
    *val = pv + vpad;
    *bits = (uint64_t *)(pb + bpad);
}

void vector_free_pair(bool padded, void *val, uint64_t *bits)
{
    if (val) dev_free((char *)val - (padded ? VEC_VAL_PAD : 0));
    if (bits) dev_free((char *)bits - (padded ? VEC_BITS_PAD : 0));
}

void vector_release_storage(GB_Vector_opaque *v)
{
    vector_free_pair(v->padded, v->d_val, v->d_bits);
    v->d_val = nullptr;
    v->d_bits = nullptr;
    v->nvals = 0;
    if (v->order) {  // (a vector without entries is in every order)
        perm_release(v->order);
        v->order = nullptr;
    }
}

// ---- vertex orders (GB_Perm, grb_internal.hpp) ------------------------------------------------------------------------------
void perm_retain(GB_Perm *p)
{
    if (p) p->refs++;
}
void perm_release(GB_Perm *p)
{
    if (!p || --p->refs > 0) return;
    dev_free(p->d_rank);
    dev_free(p->d_inv);
    delete p;
}

// new[t] = old[src_of[t]] for values and presence bits (the presence word of 64 consecutive targets by one ballot)
template <typename T>
__global__ void k_vec_permute(int64_t n, const int32_t *src_of, const T *old_val, const uint64_t *old_bits, T *new_val, uint64_t *new_bits)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    if (t < n) {
        const int64_t s = src_of[t];
        has = (old_bits[s >> 6] >> (s & 63)) & 1ull;
        new_val[t] = old_val[s];
    }
    const unsigned long long b = __ballot(has);
    if ((threadIdx.x & 63) == 0 && t < ((n + 63) & ~(int64_t)63)) new_bits[t >> 6] = b;
}

// new[dst_of[s]] = old[s] for the PRESENT positions s only (new_bits zeroed by the caller): a thread per presence word of the old image
template <typename T>
__global__ void k_vec_permute_sparse(int64_t n, const int32_t *dst_of, const T *old_val, const uint64_t *old_bits, T *new_val, unsigned long long *new_bits)
{
    const int64_t wd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wd >= ((n + 63) >> 6)) return;
    for (uint64_t bits = old_bits[wd]; bits; bits &= bits - 1) {
        const int64_t s = (wd << 6) + (__ffsll((long long)bits) - 1);
        const int64_t t = dst_of[s];
        new_val[t] = old_val[s];
        atomicOr(&new_bits[t >> 6], 1ull << (t & 63));
    }
}

// A vector whose device pointers were handed out keeps them (converted into a temporary, copied back); any other vector swaps to a
// fresh pair.  Few entries (known count below n / 16, e.g. the frontier and level vectors of the first BFS levels): only the present
// entries move -- a pass over the presence words and a scatter -- instead of a gather over all n positions.
void vector_set_order(GB_Vector_opaque *v, GB_Perm *order)
{
    if (v->order == order) return;
    if (order && order->n != v->n) fail(GrB_PANIC, "vector order: size mismatch (internal error)");
    const bool has_entries = v->d_val && v->nvals != 0 && v->n > 0;
    if (has_entries && v->order && order) vector_set_order(v, nullptr);  // (between two orders: through the natural one)
    if (has_entries) {
        const int64_t n = (int64_t)v->n;
        const size_t vb = (size_t)n * v->type->size, bb = bits_words64(v->n) * 8;
        if (n >= (1 << 16) && v->nvals < 0) (void)vector_nvals(v);  // (one host read decides between the two forms)
        const bool sparse = v->nvals >= 0 && v->nvals * 16 < n;
        void *nval = nullptr;
        uint64_t *nbits = nullptr;
        vector_alloc_pair(v, v->padded, sparse, &nval, &nbits);  // (presence zeroed; the scatter form also zeroes the values: positions it does
                                                                  //  not write must not show the block's previous content through an export)
        try {
            if (sparse) {
                // forward maps: into an order, element i goes to position d_rank[i]; back, position p goes to element d_inv[p]
                const int32_t *dst_of = order ? order->d_rank : v->order->d_inv;
                const int64_t threads = (int64_t)bits_words64(v->n);
                switch (v->type->size) {
                case 1: LAUNCH((k_vec_permute_sparse<uint8_t>), threads, n, dst_of, (const uint8_t *)v->d_val, (const uint64_t *)v->d_bits, (uint8_t *)nval, (unsigned long long *)nbits); break;
                case 2: LAUNCH((k_vec_permute_sparse<uint16_t>), threads, n, dst_of, (const uint16_t *)v->d_val, (const uint64_t *)v->d_bits, (uint16_t *)nval, (unsigned long long *)nbits); break;
                case 4: LAUNCH((k_vec_permute_sparse<uint32_t>), threads, n, dst_of, (const uint32_t *)v->d_val, (const uint64_t *)v->d_bits, (uint32_t *)nval, (unsigned long long *)nbits); break;
                default: LAUNCH((k_vec_permute_sparse<uint64_t>), threads, n, dst_of, (const uint64_t *)v->d_val, (const uint64_t *)v->d_bits, (uint64_t *)nval, (unsigned long long *)nbits); break;
                }
            } else {
                // inverse maps: into an order, position p takes the element d_inv[p]; back, element i comes from position d_rank[i]
                const int32_t *src_of = order ? order->d_inv : v->order->d_rank;
                const int64_t threads = (int64_t)bits_words64(v->n) * 64;
                switch (v->type->size) {
                case 1: LAUNCH((k_vec_permute<uint8_t>), threads, n, src_of, (const uint8_t *)v->d_val, (const uint64_t *)v->d_bits, (uint8_t *)nval, nbits); break;
                case 2: LAUNCH((k_vec_permute<uint16_t>), threads, n, src_of, (const uint16_t *)v->d_val, (const uint64_t *)v->d_bits, (uint16_t *)nval, nbits); break;
                case 4: LAUNCH((k_vec_permute<uint32_t>), threads, n, src_of, (const uint32_t *)v->d_val, (const uint64_t *)v->d_bits, (uint32_t *)nval, nbits); break;
                default: LAUNCH((k_vec_permute<uint64_t>), threads, n, src_of, (const uint64_t *)v->d_val, (const uint64_t *)v->d_bits, (uint64_t *)nval, nbits); break;
                }
            }
        } catch (...) {
            vector_free_pair(v->padded, nval, nbits);
            throw;
        }
        if (v->exported) {
            d2d(v->d_val, nval, vb);
            d2d(v->d_bits, nbits, bb);
            vector_free_pair(v->padded, nval, nbits);
        } else {
            vector_free_pair(v->padded, v->d_val, v->d_bits);
            v->d_val = nval;
            v->d_bits = nbits;
        }
        ctx().reorder_count += 1;
    }
    perm_retain(order);
    perm_release(v->order);
    v->order = order;
}

GB_Perm *vectors_common_order(GB_Vector_opaque *const *vs, int count)
{
    GB_Perm *target = nullptr;
    bool pinned = false;
    for (int i = 0; i < count; i++) {
        GB_Vector_opaque *v = vs[i];
        if (!v) continue;
        pinned = pinned || v->pinned;
        if (!target && v->order && v->d_val && v->nvals != 0) target = v->order;
    }
    if (pinned) target = nullptr;
    if (target) perm_retain(target);  // (converting the last vector that holds it must not free it)
    for (int i = 0; i < count; i++)
        if (vs[i]) vector_set_order(vs[i], target);
    if (target) perm_release(target);  // (still referenced by the vectors, unless all of them were empty)
    return target;
}

uint64_t vector_position(GB_Vector_opaque *v, uint64_t i)
{
    if (!v->order) return i;
    int32_t p = 0;
    d2h(&p, v->order->d_rank + i, sizeof(p));
    return (uint64_t)p;
}

void vector_free(GB_Vector_opaque *v)
{
    if (!v) return;
    vector_release_storage(v);
    v->magic = MAGIC_FREED;
    delete v;
}

void vector_ensure_storage(GB_Vector_opaque *v)
{
    if (v->d_val) return;
    if (v->n > (1ull << 40)) fail(GrB_OUT_OF_MEMORY, "dense-with-presence vector of this size does not fit in HBM");
    v->padded = (int64_t)((size_t)v->n * v->type->size) >= ctx().vec_pad_min_bytes;
    vector_alloc_pair(v, v->padded, true, &v->d_val, &v->d_bits);
    v->nvals = 0;
}

int64_t vector_nvals(GB_Vector_opaque *v)
{
    if (!v->d_val) return 0;
    if (v->nvals >= 0) return v->nvals;
    DevBuf<unsigned long long> cnt(1, true);
    const int64_t nwords = (int64_t)bits_words64(v->n);
    hipLaunchKernelGGL(k_popcount_sum, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, BS), 128))), dim3(BS), 0, ctx().stream,
                       (const uint64_t *)v->d_bits, nwords, cnt.p);
    unsigned long long h = 0;
    d2h(&h, cnt.p, sizeof(h));
    v->nvals = (int64_t)h;
    return v->nvals;
}

GB_Vector_opaque *vector_cast_copy(GB_Vector_opaque *v, int type)
{
    GB_Vector_opaque *w = vector_new(type_of_code(type), v->n);
    if (v->d_val) {
        vector_ensure_storage(w);
        cast_array(type, w->d_val, v->type->code, v->d_val, (int64_t)v->n);
        d2d(w->d_bits, v->d_bits, bits_words64(v->n) * 8);
        w->nvals = v->nvals;
        perm_retain(v->order);  // (a copy is element-wise: it is in the order of its source)
        w->order = v->order;
    }
    return w;
}

void vector_mask_bits(GB_Vector_opaque *m, bool structure, uint64_t *out_bits)
{
    const size_t nw = bits_words64(m->n);
    if (!m->d_val) {
        GRB_HIP(hipMemsetAsync(out_bits, 0, nw * 8, ctx().stream));
        return;
    }
    if (structure) {
        d2d(out_bits, m->d_bits, nw * 8);
        return;
    }
    GRB_DISPATCH_TYPE(m->type->code, T, {
        LAUNCH((k_mask_bits<T>), (int64_t)nw * 64, m->d_bits, (const T *)m->d_val, (int64_t)m->n, out_bits);
    })
}

// sort tuples by key, fold duplicates.  Returns number of unique keys; outputs are fresh device arrays.
template <typename T>
static int64_t sort_dedupe(uint64_t *keys, uint32_t *perm, const T *X, int64_t n, int key_bits, int dup_op,
                           uint64_t **out_keys, T **out_vals, bool *had_dups)
{
    DevBuf<uint64_t> keys2(n);
    DevBuf<uint32_t> perm2(n);
    prim_sort_pairs_u64_u32(keys, keys2.p, perm, perm2.p, n, key_bits);
    DevBuf<int64_t> flags(n + 1, false);
    LAUNCH(k_mark_heads, n, keys2.p, n, flags.p);
    GRB_HIP(hipMemsetAsync(flags.p + n, 0, sizeof(int64_t), ctx().stream));
    prim_exclusive_sum_i64(flags.p, flags.p, n + 1);
    int64_t nuniq = 0;
    d2h(&nuniq, flags.p + n, sizeof(int64_t));
    *had_dups = nuniq < n;
    DevBuf<uint64_t> ok(nuniq);
    DevBuf<T> ov(nuniq);
    LAUNCH((k_dedupe<T>), n, keys2.p, perm2.p, X, n, flags.p, dup_op >= 0 ? dup_op : (int)OP_SECOND, ok.p, ov.p);
    *out_keys = ok.release();
    *out_vals = ov.release();
    return nuniq;
}

static inline int ceil_log2_u64(uint64_t x)
{
    int b = 0;
    while (b < 63 && (1ull << b) < x) b++;
    return b;
}

template <typename T>
static void vector_build_typed(GB_Vector_opaque *w, const uint64_t *I, const void *X_host, int x_type, int64_t n,
                               const GB_BinaryOp_opaque *dup)
{
    if (vector_nvals(w) > 0) fail(GrB_OUTPUT_NOT_EMPTY, "GrB_Vector_build: output already has entries");
    if (n == 0) return;
    if (!I || !X_host) fail(GrB_NULL_POINTER, "GrB_Vector_build: NULL index or value array");
    if (dup && (dup->type != w->type->code || op_is_comparison(dup->op))) fail(GrB_DOMAIN_MISMATCH, "dup operator type must match the vector type");
    if ((uint64_t)n > 0xffffffffull) fail(GrB_NOT_IMPLEMENTED, "more than 2^32 tuples in one build");
    vector_ensure_storage(w);
    DevBuf<uint64_t> dI(n);
    h2d(dI.p, I, sizeof(uint64_t) * n);
    DevBuf<char> dXraw((size_t)n * type_size(x_type));
    h2d(dXraw.p, X_host, (size_t)n * type_size(x_type));
    DevBuf<T> dX(n);
    cast_array(w->type->code, dX.p, x_type, dXraw.p, n);
    DevBuf<uint64_t> keys(n);
    DevBuf<uint32_t> perm(n);
    DevBuf<int> err(1, true);
    LAUNCH(k_make_keys, n, dI.p, (const uint64_t *)nullptr, n, w->n, (uint64_t)1, 0, keys.p, perm.p, err.p);
    int herr = 0;
    d2h(&herr, err.p, sizeof(int));
    if (herr) fail(GrB_INDEX_OUT_OF_BOUNDS, "GrB_Vector_build: index out of bounds");
    uint64_t *ok = nullptr;
    T *ov = nullptr;
    bool dups = false;
    const int dup_op = dup ? canonical_op(w->type->code, dup->op) : -1;
    int64_t nu = sort_dedupe<T>(keys.p, perm.p, dX.p, n, std::max(1, ceil_log2_u64(w->n)), dup_op, &ok, &ov, &dups);
    DevPtr<uint64_t> ok_own(ok);  // (released on every way out, a throwing launch included)
    DevPtr<T> ov_own(ov);
    if (dups && !dup) fail(GrB_INVALID_VALUE, "GrB_Vector_build: duplicate indices and no dup operator");
    LAUNCH((k_vec_scatter<T>), nu, ok, ov, nu, (T *)w->d_val, w->d_bits);
    w->nvals = nu;
}

template <typename T>
static void vector_extract_typed(GB_Vector_opaque *v, uint64_t *I, void *X, int x_type, uint64_t *nvals_io)
{
    if (!nvals_io) fail(GrB_NULL_POINTER, "nvals is NULL");
    const int64_t nv = vector_nvals(v);
    if ((uint64_t)nv > *nvals_io && (I || X)) fail(GrB_INSUFFICIENT_SPACE, "extractTuples: output arrays too small");
    *nvals_io = (uint64_t)nv;
    if (nv == 0 || (!I && !X)) return;
    const int64_t nwords = (int64_t)bits_words64(v->n);
    DevBuf<int64_t> offs(nwords + 1);
    LAUNCH(k_word_popcounts, nwords, v->d_bits, nwords, offs.p);
    prim_exclusive_sum_i64(offs.p, offs.p, nwords);
    DevBuf<uint64_t> dI(nv);
    DevBuf<T> dX(nv);
    LAUNCH((k_vec_extract<T>), nwords, v->d_bits, (const T *)v->d_val, nwords, offs.p, dI.p, dX.p);
    if (I) d2h(I, dI.p, sizeof(uint64_t) * nv);
    if (X) {
        if (x_type == v->type->code) d2h(X, dX.p, sizeof(T) * nv);
        else {
            DevBuf<char> dXc((size_t)nv * type_size(x_type));
            cast_array(x_type, dXc.p, v->type->code, dX.p, nv);
            d2h(X, dXc.p, (size_t)nv * type_size(x_type));
        }
    }
}

// out bit i = present bit i && val[i]   (raw device images; used for the bit-packed BOOL operand of the pull SpMV)
void pack_bool_values(const uint64_t *present, const bool *val, int64_t n, uint64_t *out)
{
    const int64_t nw = (int64_t)bits_words64((uint64_t)n);
    LAUNCH((k_mask_bits<bool>), nw * 64, present, val, n, out);
}

// out word w, bits (2k, 2k+1) = (present, present && value) of element 16 w + k: presence and value of a BOOL operand in one
// gather (pull SpMV with a u that is not full)
__global__ void k_pack_pv(const uint64_t *present, const bool *val, int64_t n, uint32_t *out)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ((n + 15) >> 4)) return;
    const uint32_t p16 = (uint32_t)(present[w >> 2] >> ((w & 3) * 16)) & 0xffffu;
    uint32_t o = 0;
    if (p16 == 0u) {  // (nothing present in these 16: the values are not read)
        out[w] = 0u;
        return;
    }
    if ((w << 4) + 16 <= n && (((uintptr_t)val) & 15) == 0) {
        // the 16 one-byte values in one load (a byte load per element was 14 us per call at scale 24: a twentieth of the BFS level step)
        const uint4 v = *(const uint4 *)(val + (w << 4));
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t b = (vw[k >> 2] >> ((k & 3) * 8)) & 0xffu;
            if ((p16 >> k) & 1u) o |= (1u | (b ? 2u : 0u)) << (2 * k);
        }
        out[w] = o;
        return;
    }
    for (int k = 0; k < 16; k++) {
        const int64_t i = (w << 4) + k;
        if (i < n && ((p16 >> k) & 1u)) o |= (1u | (val[i] ? 2u : 0u)) << (2 * k);
    }
    out[w] = o;
}
void pack_bool_pv(const uint64_t *present, const bool *val, int64_t n, uint32_t *out)
{
    LAUNCH(k_pack_pv, (n + 15) >> 4, present, val, n, out);
}

// indices of the present entries, ascending, as a fresh device array (caller frees); returns the count
int64_t vector_index_list(GB_Vector_opaque *v, uint64_t **d_idx)
{
    *d_idx = nullptr;
    const int64_t nv = vector_nvals(v);
    if (nv == 0) return 0;
    const int64_t nwords = (int64_t)bits_words64(v->n);
    DevBuf<int64_t> offs(nwords + 1);
    LAUNCH(k_word_popcounts, nwords, v->d_bits, nwords, offs.p);
    prim_exclusive_sum_i64(offs.p, offs.p, nwords);
    uint64_t *idx = (uint64_t *)dev_alloc(sizeof(uint64_t) * (size_t)nv);
    LAUNCH((k_vec_extract<uint8_t>), nwords, v->d_bits, (const uint8_t *)nullptr, nwords, offs.p, idx, (uint8_t *)nullptr);
    *d_idx = idx;
    return nv;
}

// ---------------------------------------------------------------------------------------------------
// Matrix
// ---------------------------------------------------------------------------------------------------
GB_Matrix_opaque *matrix_new(GrB_Type type, uint64_t nrows, uint64_t ncols)
{
    auto *A = new GB_Matrix_opaque();
    A->magic = MAGIC_MATRIX;
    A->type = type;
    A->nrows = nrows;
    A->ncols = ncols;
    A->nvals = 0;
    A->d_ptr = nullptr;
    A->d_col = nullptr;
    A->d_val = nullptr;
    A->iso = false;
    A->owns = true;
    A->tr = nullptr;
    A->d_tile_row = nullptr;
    A->n_tiles = 0;
    A->tile_items = 0;
    A->d_col_hot = nullptr;
    A->d_hot_cols = nullptr;
    A->hot_k = 0;
    A->hot_state = 0;
    A->hot_cols_dropped = false;
    A->short_part = nullptr;
    A->d_long_bits = nullptr;
    A->d_long_rows = nullptr;
    A->d_chunk_slot = nullptr;
    A->d_chunk_start = nullptr;
    A->d_chunk_len = nullptr;
    A->d_long_prefix = nullptr;
    A->d_lcol = nullptr;
    A->d_lval = nullptr;
    A->d_it_start = nullptr;
    A->d_it_len = nullptr;
    A->d_it_slot = nullptr;
    A->d_item_begin = nullptr;
    A->long_nnz = 0;
    A->n_items = 0;
    for (auto &x : A->item_begin) x = 0;
    A->n_long = A->n_chunks = 0;
    A->split_state = 0;
    A->split_hot = false;
    return A;
}

void matrix_invalidate_caches(GB_Matrix_opaque *A)
{
    if (A->tr) {
        matrix_free(A->tr);
        A->tr = nullptr;
    }
    if (A->ord) {
        matrix_free(A->ord);
        A->ord = nullptr;
    }
    perm_release(A->perm);  // (vectors in this order keep it alive until they are converted)
    A->perm = nullptr;
    A->col_order_only = false;  // (a shard set-up describes the content that is going away)
    A->ord_state = 0;
    dev_free(A->d_cold_bounds); dev_free(A->d_ct_order);
    A->d_cold_bounds = nullptr; A->d_ct_order = nullptr; A->ct_ntiles = 0;
    if (A->hot_identity) A->d_col_hot = nullptr;  // (an alias of d_col)
    dev_free(A->d_tile_row);
    A->d_tile_row = nullptr;
    A->n_tiles = 0;
    A->tile_items = 0;
    dev_free(A->d_col_hot);
    dev_free(A->d_hot_cols);
    A->d_col_hot = nullptr;
    A->d_hot_cols = nullptr;
    A->hot_k = 0;
    A->hot_state = 0;
    A->hot_cols_dropped = false;
    if (A->short_part) matrix_free(A->short_part);
    A->short_part = nullptr;
    dev_free(A->d_long_bits);
    dev_free(A->d_long_rows);
    dev_free(A->d_chunk_slot);
    dev_free(A->d_chunk_start);
    dev_free(A->d_chunk_len);
    dev_free(A->d_long_prefix);
    dev_free(A->d_lcol);
    dev_free(A->d_lval);
    dev_free(A->d_it_start);
    dev_free(A->d_it_len);
    dev_free(A->d_it_slot);
    dev_free(A->d_item_begin);
    dev_free(A->d_sstart);
    dev_free(A->d_sslot);
    dev_free(A->d_sslot16); dev_free(A->d_sslot_base);
    A->d_sslot16 = nullptr; A->d_sslot_base = nullptr;
    dev_free(A->d_hrec);
    dev_free(A->d_vdict); dev_free(A->d_vd_table); dev_free(A->d_vd_codes);
    A->d_vdict = nullptr; A->d_vd_table = nullptr; A->d_vd_codes = nullptr;
    A->vdict_n = 0;
    dev_free(A->d_ct_col); dev_free(A->d_ct_val); dev_free(A->d_ct_loc); dev_free(A->d_ct_tiles);
    A->d_ct_col = nullptr; A->d_ct_val = nullptr; A->d_ct_loc = nullptr; A->d_ct_tiles = nullptr; A->ct_units = 0;
    A->d_sstart = nullptr;
    A->d_sslot = nullptr;
    A->d_hrec = nullptr;
    A->strip_nseg = 0;
    A->hub_ncls = 0;
    dev_free(A->d_wg_tab); dev_free(A->d_strip_cb);
    A->d_wg_tab = nullptr; A->d_strip_cb = nullptr; A->wg_tab_g = 0;
    A->pull_calls = 0;
    dev_free(A->d_tg_off); dev_free(A->d_tg_col); dev_free(A->d_tg_val); dev_free(A->d_tg_tag); dev_free(A->d_tg_nonempty);
    A->d_tg_off = nullptr; A->d_tg_col = nullptr; A->d_tg_val = nullptr; A->d_tg_tag = nullptr; A->d_tg_nonempty = nullptr;
    dev_free(A->d_probe); A->d_probe = nullptr; A->probe_k = 0;
    dev_free(A->d_rt_col); dev_free(A->d_rt_tag); dev_free(A->d_rt_val); dev_free(A->d_rt_tiles); dev_free(A->d_rt_order); dev_free(A->d_rt_counter);
    A->d_rt_col = nullptr; A->d_rt_tag = nullptr; A->d_rt_val = nullptr; A->d_rt_tiles = nullptr; A->d_rt_order = nullptr; A->d_rt_counter = nullptr;
    A->rt_state = 0; A->rt_units = 0; A->rt_ntiles = 0;
    A->tg_state = 0;
    A->d_lcol = nullptr;
    A->d_lval = nullptr;
    A->d_it_start = nullptr;
    A->d_it_len = nullptr;
    A->d_it_slot = nullptr;
    A->d_item_begin = nullptr;
    A->long_nnz = 0;
    A->n_items = 0;
    A->d_long_bits = nullptr;
    A->d_long_rows = nullptr;
    A->d_chunk_slot = nullptr;
    A->d_chunk_start = nullptr;
    A->d_chunk_len = nullptr;
    A->d_long_prefix = nullptr;
    A->n_long = A->n_chunks = 0;
    A->split_state = 0;
}

void matrix_release_storage(GB_Matrix_opaque *A)
{
    matrix_invalidate_caches(A);
    if (A->owns) {
        dev_free(A->d_ptr);
        dev_free(A->d_col);
        dev_free(A->d_val);
    }
    A->d_ptr = nullptr;
    A->d_col = nullptr;
    A->d_val = nullptr;
    A->nvals = 0;
    A->iso = false;
    A->owns = true;
}

void matrix_free(GB_Matrix_opaque *A)
{
    if (!A) return;
    matrix_release_storage(A);
    A->magic = MAGIC_FREED;
    delete A;
}

const int64_t *matrix_rowptr(GB_Matrix_opaque *A)
{
    if (!A->d_ptr) {
        if (A->nrows > (1ull << 40)) fail(GrB_OUT_OF_MEMORY, "row-pointer array of this size does not fit in HBM");
        A->d_ptr = (int64_t *)dev_alloc_zero(sizeof(int64_t) * (A->nrows + 1));
        A->owns = true;
    }
    return A->d_ptr;
}

static void check_index_width(uint64_t nrows, uint64_t ncols)
{
    if (ncols > 0x7fffffffull) fail(GrB_NOT_IMPLEMENTED, "matrices with entries need ncols < 2^31 (int32 column indices)");
    if (nrows > 0xffffffffull) fail(GrB_NOT_IMPLEMENTED, "matrices with entries need nrows <= 2^32");
}

// keep ONE value if all stored bit patterns are equal
template <typename T>
static void matrix_detect_iso(GB_Matrix_opaque *A)
{
    if (A->nvals <= 1 || A->iso || !A->owns) return;
    DevBuf<int> flag(1, true);
    LAUNCH((k_iso_check<T>), A->nvals, (const T *)A->d_val, A->nvals, flag.p);
    int h = 0;
    d2h(&h, flag.p, sizeof(int));
    if (!h) {
        T *one = (T *)dev_alloc(sizeof(T));
        d2d(one, A->d_val, sizeof(T));
        dev_free(A->d_val);
        A->d_val = one;
        A->iso = true;
    }
}

// Install sorted unique keys (row<<cshift|col) + values as the CSR of A.  Takes the values out of `vals` once A owns them.
template <typename T>
static void matrix_install_from_keys(GB_Matrix_opaque *A, const uint64_t *keys, DevPtr<T> &vals, int64_t nuniq, int cshift)
{
    matrix_release_storage(A);
    A->owns = true;  // (what is attached below is released with A, whatever throws in between)
    A->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (A->nrows + 1));
    LAUNCH(k_rowptr_from_keys, (int64_t)A->nrows + 1, keys, nuniq, (int64_t)A->nrows, cshift, A->d_ptr);
    A->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(nuniq ? nuniq : 1));
    LAUNCH(k_cols_from_keys, nuniq, keys, nuniq, cshift, A->d_col);
    A->d_val = vals.release();
    A->nvals = nuniq;
    A->iso = false;
    A->owns = true;
    matrix_detect_iso<T>(A);
}

// COO on the device -> CSR.  dI/dJ are device uint64 arrays, dX device values already of A's type.
template <typename T>
static void matrix_build_device(GB_Matrix_opaque *A, const uint64_t *dI, const uint64_t *dJ, const T *dX, int64_t n,
                                const GB_BinaryOp_opaque *dup, const char *who)
{
    check_index_width(A->nrows, A->ncols);
    if ((uint64_t)n > 0xffffffffull) fail(GrB_NOT_IMPLEMENTED, "more than 2^32 tuples in one build");
    const int cshift = std::max(1, ceil_log2_u64(A->ncols));
    const int key_bits = std::min(64, cshift + std::max(1, ceil_log2_u64(A->nrows)));
    DevBuf<uint64_t> keys(n);
    DevBuf<uint32_t> perm(n);
    DevBuf<int> err(1, true);
    LAUNCH(k_make_keys, n, dI, dJ, n, A->nrows, A->ncols, cshift, keys.p, perm.p, err.p);
    int herr = 0;
    d2h(&herr, err.p, sizeof(int));
    if (herr) fail(GrB_INDEX_OUT_OF_BOUNDS, std::string(who) + ": index out of bounds");
    uint64_t *ok = nullptr;
    T *ov = nullptr;
    bool dups = false;
    const int dup_op = dup ? canonical_op(A->type->code, dup->op) : -1;
    int64_t nu = sort_dedupe<T>(keys.p, perm.p, dX, n, key_bits, dup_op, &ok, &ov, &dups);
    DevPtr<uint64_t> ok_own(ok);
    DevPtr<T> ov_own(ov);
    if (dups && !dup) fail(GrB_INVALID_VALUE, std::string(who) + ": duplicate indices and no dup operator");
    matrix_install_from_keys<T>(A, ok, ov_own, nu, cshift);
}

template <typename T>
static void matrix_build_typed(GB_Matrix_opaque *C, const uint64_t *I, const uint64_t *J, const void *X_host, int x_type,
                               int64_t n, const GB_BinaryOp_opaque *dup)
{
    if (C->nvals > 0) fail(GrB_OUTPUT_NOT_EMPTY, "GrB_Matrix_build: output already has entries");
    if (n == 0) return;
    if (!I || !J || !X_host) fail(GrB_NULL_POINTER, "GrB_Matrix_build: NULL index or value array");
    if (dup && (dup->type != C->type->code || op_is_comparison(dup->op))) fail(GrB_DOMAIN_MISMATCH, "dup operator type must match the matrix type");
    DevBuf<uint64_t> dI(n), dJ(n);
    h2d(dI.p, I, sizeof(uint64_t) * n);
    h2d(dJ.p, J, sizeof(uint64_t) * n);
    DevBuf<char> dXraw((size_t)n * type_size(x_type));
    h2d(dXraw.p, X_host, (size_t)n * type_size(x_type));
    DevBuf<T> dX(n);
    cast_array(C->type->code, dX.p, x_type, dXraw.p, n);
    matrix_build_device<T>(C, dI.p, dJ.p, dX.p, n, dup, "GrB_Matrix_build");
}

// full-length device copy of A's values (expands iso)
template <typename T>
static T *matrix_values_expanded(GB_Matrix_opaque *A)
{
    T *out = (T *)dev_alloc(sizeof(T) * (size_t)(A->nvals ? A->nvals : 1));
    if (A->iso) LAUNCH((k_fill<T>), A->nvals, out, A->nvals, (const T *)A->d_val);
    else d2d(out, A->d_val, sizeof(T) * (size_t)A->nvals);
    return out;
}

template <typename T>
static void matrix_extract_typed(GB_Matrix_opaque *A, uint64_t *I, uint64_t *J, void *X, int x_type, uint64_t *nvals_io)
{
    if (!nvals_io) fail(GrB_NULL_POINTER, "nvals is NULL");
    const int64_t nv = A->nvals;
    if ((uint64_t)nv > *nvals_io && (I || J || X)) fail(GrB_INSUFFICIENT_SPACE, "extractTuples: output arrays too small");
    *nvals_io = (uint64_t)nv;
    if (nv == 0) return;
    if (I) {
        DevBuf<uint64_t> rows(nv);
        LAUNCH(k_expand_rows_u64, nv, A->d_ptr, (int64_t)A->nrows, nv, rows.p);
        d2h(I, rows.p, sizeof(uint64_t) * nv);
    }
    if (J) {
        DevBuf<uint64_t> cols(nv);
        LAUNCH(k_i32_to_u64, nv, A->d_col, nv, cols.p);
        d2h(J, cols.p, sizeof(uint64_t) * nv);
    }
    if (X) {
        DevPtr<T> vals_own(matrix_values_expanded<T>(A));
        T *vals = vals_own.p;
        if (x_type == A->type->code) d2h(X, vals, sizeof(T) * nv);
        else {
            DevBuf<char> c((size_t)nv * type_size(x_type));
            cast_array(x_type, c.p, A->type->code, vals, nv);
            d2h(X, c.p, (size_t)nv * type_size(x_type));
        }
    }
}

// ---- transpose ---------------------------------------------------------------------------------------
template <typename T>
static GB_Matrix_opaque *matrix_transpose_new(GB_Matrix_opaque *A)
{
    GB_Matrix_opaque *B = matrix_new(A->type, A->ncols, A->nrows);
    if (A->nvals == 0) return B;
    check_index_width(B->nrows, B->ncols);
    const int64_t nnz = A->nvals;
    const int rshift = std::max(1, ceil_log2_u64(A->nrows));  // B's "column" (= A's row) bits
    const int key_bits = std::min(64, rshift + std::max(1, ceil_log2_u64(A->ncols)));
    DevBuf<uint64_t> keys(nnz), keys2(nnz);
    DevBuf<uint32_t> perm(nnz), perm2(nnz);
    LAUNCH(k_transpose_keys, nnz, A->d_ptr, A->d_col, (int64_t)A->nrows, nnz, rshift, keys.p, perm.p);
    prim_sort_pairs_u64_u32(keys.p, keys2.p, perm.p, perm2.p, nnz, key_bits);
    B->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (B->nrows + 1));
    LAUNCH(k_rowptr_from_keys, (int64_t)B->nrows + 1, keys2.p, nnz, (int64_t)B->nrows, rshift, B->d_ptr);
    B->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnz);
    LAUNCH(k_cols_from_keys, nnz, keys2.p, nnz, rshift, B->d_col);
    if (A->iso) {
        B->d_val = dev_alloc(sizeof(T));
        d2d(B->d_val, A->d_val, sizeof(T));
        B->iso = true;
    } else {
        B->d_val = dev_alloc(sizeof(T) * (size_t)nnz);
        LAUNCH((k_gather<T>), nnz, (const T *)A->d_val, perm2.p, nnz, (T *)B->d_val);
    }
    B->nvals = nnz;
    return B;
}

GB_Matrix_opaque *matrix_transpose_cached(GB_Matrix_opaque *A)
{
    if (!A->tr) {
        GRB_DISPATCH_TYPE(A->type->code, T, { A->tr = matrix_transpose_new<T>(A); })
        A->tr->tr_of = A;
        A->tr->ranked = A->ranked;  // (labels ranked for rows and columns alike: GrX_Matrix_hint_ranked)
    }
    return A->tr;
}

GB_Matrix_opaque *matrix_cast_copy(GB_Matrix_opaque *A, int type)
{
    GB_Matrix_opaque *B = matrix_new(type_of_code(type), A->nrows, A->ncols);
    if (A->nvals == 0) return B;
    B->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (A->nrows + 1));
    d2d(B->d_ptr, A->d_ptr, sizeof(int64_t) * (A->nrows + 1));
    B->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)A->nvals);
    d2d(B->d_col, A->d_col, sizeof(int32_t) * (size_t)A->nvals);
    const int64_t nv = A->iso ? 1 : A->nvals;
    B->d_val = dev_alloc(type_size(type) * (size_t)nv);
    cast_array(type, B->d_val, A->type->code, A->d_val, nv);
    B->iso = A->iso;
    B->nvals = A->nvals;
    return B;
}

static GB_Matrix_opaque *matrix_dup(GB_Matrix_opaque *A) { return matrix_cast_copy(A, A->type->code); }

// import host CSR/CSC (uint64 pointers and indices) into the (empty) matrix A.  Entries inside a row need not be sorted.
// iso: Ax holds ONE value that every entry takes (the GxB import / pack forms; GrB_Matrix_import has no such flag).
template <typename T>
static void matrix_import_into(GB_Matrix_opaque *A, const uint64_t *Ap, const uint64_t *Ai, const void *Ax, int x_type, uint64_t Ap_len,
                               uint64_t Ai_len, uint64_t Ax_len, GrB_Format format, bool iso, const char *who)
{
    if (format != GrB_CSR_FORMAT && format != GrB_CSC_FORMAT) fail(GrB_NOT_IMPLEMENTED, std::string(who) + ": only CSR and CSC formats");
    const bool csr = (format == GrB_CSR_FORMAT);
    const uint64_t nvec = csr ? A->nrows : A->ncols;
    if (!Ap) fail(GrB_NULL_POINTER, std::string(who) + ": Ap is NULL");
    if (Ap_len < nvec + 1) fail(GrB_INVALID_VALUE, std::string(who) + ": Ap_len too small");
    const uint64_t nnz = Ap[nvec];
    if (Ai_len < nnz || Ax_len < (iso ? (nnz ? 1 : 0) : nnz)) fail(GrB_INVALID_VALUE, std::string(who) + ": Ai_len/Ax_len smaller than Ap[n]");
    if (nnz == 0) return;
    if (!Ai || !Ax) fail(GrB_NULL_POINTER, std::string(who) + ": NULL array");
    // expand the pointer array into major indices on the device
    DevBuf<int64_t> dp(nvec + 1);
    DevBuf<uint64_t> dmaj(nnz), dmin(nnz);
    h2d(dp.p, Ap, sizeof(uint64_t) * (nvec + 1));  // same bits (values < 2^63)
    LAUNCH(k_expand_rows_u64, (int64_t)nnz, dp.p, (int64_t)nvec, (int64_t)nnz, dmaj.p);
    h2d(dmin.p, Ai, sizeof(uint64_t) * nnz);
    const uint64_t nx = iso ? 1 : nnz;
    DevBuf<char> raw((size_t)nx * type_size(x_type));
    h2d(raw.p, Ax, (size_t)nx * type_size(x_type));
    DevBuf<T> dX(nnz);
    if (iso) {
        DevBuf<T> one(1);
        cast_array(A->type->code, one.p, x_type, raw.p, 1);
        LAUNCH((k_fill<T>), (int64_t)nnz, dX.p, (int64_t)nnz, (const T *)one.p);
    } else {
        cast_array(A->type->code, dX.p, x_type, raw.p, (int64_t)nnz);
    }
    matrix_build_device<T>(A, csr ? dmaj.p : dmin.p, csr ? dmin.p : dmaj.p, dX.p, (int64_t)nnz, nullptr, who);
}

template <typename T>
static GB_Matrix_opaque *matrix_import_typed(GrB_Type type, uint64_t nrows, uint64_t ncols, const uint64_t *Ap,
                                             const uint64_t *Ai, const void *Ax, int x_type, uint64_t Ap_len,
                                             uint64_t Ai_len, uint64_t Ax_len, GrB_Format format)
{
    if (format != GrB_CSR_FORMAT && format != GrB_CSC_FORMAT) fail(GrB_NOT_IMPLEMENTED, "GrB_Matrix_import: only CSR and CSC formats");
    GB_Matrix_opaque *A = matrix_new(type, nrows, ncols);
    try {
        matrix_import_into<T>(A, Ap, Ai, Ax, x_type, Ap_len, Ai_len, Ax_len, format, false, "GrB_Matrix_import");
    } catch (...) {
        matrix_free(A);
        throw;
    }
    return A;
}

template <typename T>
static void matrix_export_typed(GB_Matrix_opaque *A, uint64_t *Ap, uint64_t *Ai, void *Ax, int x_type, uint64_t *Ap_len,
                                uint64_t *Ai_len, uint64_t *Ax_len, GrB_Format format)
{
    if (format != GrB_CSR_FORMAT && format != GrB_CSC_FORMAT) fail(GrB_NOT_IMPLEMENTED, "GrB_Matrix_export: only CSR and CSC formats");
    GB_Matrix_opaque *S = (format == GrB_CSR_FORMAT) ? A : matrix_transpose_cached(A);
    if (!Ap || !Ai || !Ax || !Ap_len || !Ai_len || !Ax_len) fail(GrB_NULL_POINTER, "GrB_Matrix_export: NULL argument");
    const uint64_t nvec = S->nrows, nnz = (uint64_t)S->nvals;
    if (*Ap_len < nvec + 1 || *Ai_len < nnz || *Ax_len < nnz) fail(GrB_INSUFFICIENT_SPACE, "GrB_Matrix_export: arrays too small");
    *Ap_len = nvec + 1;
    *Ai_len = nnz;
    *Ax_len = nnz;
    DevBuf<uint64_t> p(nvec + 1);
    LAUNCH(k_i64_to_u64, (int64_t)nvec + 1, matrix_rowptr(S), (int64_t)nvec + 1, p.p);
    d2h(Ap, p.p, sizeof(uint64_t) * (nvec + 1));
    if (nnz) {
        uint64_t n_io = nnz;
        matrix_extract_typed<T>(S, nullptr, Ai, Ax, x_type, &n_io);
    }
}

}  // namespace grb

using namespace grb;

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" GrB_Info GrB_Vector_new(GrB_Vector *v, GrB_Type type, GrB_Index n)
{
    GRB_TRY
    require_init();
    if (!v || !type) fail(GrB_NULL_POINTER, "GrB_Vector_new: NULL argument");
    *v = nullptr;
    if (n > GrB_INDEX_MAX + 1) fail(GrB_INVALID_VALUE, "GrB_Vector_new: size exceeds GrB_INDEX_MAX+1");
    *v = vector_new(type, n);
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Vector_dup(GrB_Vector *w, const GrB_Vector u)
{
    GRB_TRY
    require_init();
    if (!w) fail(GrB_NULL_POINTER, "GrB_Vector_dup: NULL output");
    check_vector_any(u, "u");
    *w = vector_cast_copy(u, u->type->code);
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_Vector_dup_as(GrB_Vector *w, const GrB_Type type, const GrB_Vector u)
{
    GRB_TRY
    require_init();
    if (!w) fail(GrB_NULL_POINTER, "GrX_Vector_dup_as: NULL output");
    if (!type) fail(GrB_NULL_POINTER, "GrX_Vector_dup_as: NULL type");
    check_vector_any(u, "u");
    *w = vector_cast_copy(u, type->code);
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Vector_free(GrB_Vector *v)
{
    if (!v || !*v) return GrB_SUCCESS;
    if ((*v)->magic == MAGIC_VECTOR) vector_free(*v);
    *v = nullptr;
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrB_Vector_clear(GrB_Vector v)
{
    GRB_TRY
    check_vector_any(v, "v");
    vector_release_storage(v);
    GRB_CATCH(errp(v))
}

extern "C" GrB_Info GrB_Vector_size(GrB_Index *n, const GrB_Vector v)
{
    GRB_TRY
    if (!n) fail(GrB_NULL_POINTER, "n is NULL");
    check_vector_any(v, "v");
    *n = v->n;
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Vector_nvals(GrB_Index *nvals, const GrB_Vector v)
{
    GRB_TRY
    if (!nvals) fail(GrB_NULL_POINTER, "nvals is NULL");
    check_vector_any(v, "v");
    *nvals = (GrB_Index)vector_nvals(v);
    GRB_CATCH(errp(v))
}

extern "C" GrB_Info GrB_Vector_wait(GrB_Vector v, GrB_WaitMode)
{
    GRB_TRY
    check_vector_any(v, "v");
    sync_stream();
    GRB_CATCH(errp(v))
}

extern "C" GrB_Info GrB_Vector_error(const char **error, const GrB_Vector v)
{
    if (!error) return GrB_NULL_POINTER;
    if (!v || v->magic != MAGIC_VECTOR) { *error = ""; return v ? GrB_INVALID_OBJECT : GrB_NULL_POINTER; }
    *error = v->err.c_str();
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrB_Matrix_new(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols)
{
    GRB_TRY
    require_init();
    if (!A || !type) fail(GrB_NULL_POINTER, "GrB_Matrix_new: NULL argument");
    *A = nullptr;
    if (nrows > GrB_INDEX_MAX + 1 || ncols > GrB_INDEX_MAX + 1) fail(GrB_INVALID_VALUE, "GrB_Matrix_new: dimension exceeds GrB_INDEX_MAX+1");
    *A = matrix_new(type, nrows, ncols);
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Matrix_dup(GrB_Matrix *C, const GrB_Matrix A)
{
    GRB_TRY
    require_init();
    if (!C) fail(GrB_NULL_POINTER, "GrB_Matrix_dup: NULL output");
    check_matrix(A, "A");
    *C = matrix_dup(A);
    GRB_CATCH(nullptr)
}

// C = a copy of A with its values cast to `type` on the device (the reference's dup(dtype): core/matrix.py:469-497 builds the
// new object and assigns `rv << self`, i.e. an identity apply with a typecast)
extern "C" GrB_Info GrX_Matrix_dup_as(GrB_Matrix *C, const GrB_Type type, const GrB_Matrix A)
{
    GRB_TRY
    require_init();
    if (!C) fail(GrB_NULL_POINTER, "GrX_Matrix_dup_as: NULL output");
    if (!type) fail(GrB_NULL_POINTER, "GrX_Matrix_dup_as: NULL type");
    check_matrix(A, "A");
    *C = matrix_cast_copy(A, type->code);
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Matrix_free(GrB_Matrix *A)
{
    if (!A || !*A) return GrB_SUCCESS;
    if ((*A)->magic == MAGIC_MATRIX) matrix_free(*A);
    *A = nullptr;
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrB_Matrix_clear(GrB_Matrix A)
{
    GRB_TRY
    check_matrix(A, "A");
    matrix_release_storage(A);
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrB_Matrix_nrows(GrB_Index *nrows, const GrB_Matrix A)
{
    GRB_TRY
    if (!nrows) fail(GrB_NULL_POINTER, "nrows is NULL");
    check_matrix(A, "A");
    *nrows = A->nrows;
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Matrix_ncols(GrB_Index *ncols, const GrB_Matrix A)
{
    GRB_TRY
    if (!ncols) fail(GrB_NULL_POINTER, "ncols is NULL");
    check_matrix(A, "A");
    *ncols = A->ncols;
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Matrix_nvals(GrB_Index *nvals, const GrB_Matrix A)
{
    GRB_TRY
    if (!nvals) fail(GrB_NULL_POINTER, "nvals is NULL");
    check_matrix(A, "A");
    *nvals = (GrB_Index)A->nvals;
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrB_Matrix_wait(GrB_Matrix A, GrB_WaitMode)
{
    GRB_TRY
    check_matrix(A, "A");
    sync_stream();
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrB_Matrix_error(const char **error, const GrB_Matrix A)
{
    if (!error) return GrB_NULL_POINTER;
    if (!A || A->magic != MAGIC_MATRIX) { *error = ""; return A ? GrB_INVALID_OBJECT : GrB_NULL_POINTER; }
    *error = A->err.c_str();
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrB_Matrix_exportSize(GrB_Index *Ap_len, GrB_Index *Ai_len, GrB_Index *Ax_len, GrB_Format format,
                                          const GrB_Matrix A)
{
    GRB_TRY
    if (!Ap_len || !Ai_len || !Ax_len) fail(GrB_NULL_POINTER, "GrB_Matrix_exportSize: NULL argument");
    check_matrix(A, "A");
    if (format == GrB_CSR_FORMAT) *Ap_len = A->nrows + 1;
    else if (format == GrB_CSC_FORMAT) *Ap_len = A->ncols + 1;
    else fail(GrB_NOT_IMPLEMENTED, "GrB_Matrix_exportSize: only CSR and CSC formats");
    *Ai_len = (GrB_Index)A->nvals;
    *Ax_len = (GrB_Index)A->nvals;
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrB_Matrix_exportHint(GrB_Format *format, const GrB_Matrix A)
{
    if (!format) return GrB_NULL_POINTER;
    if (!A) return GrB_NULL_POINTER;
    *format = GrB_CSR_FORMAT;
    return GrB_SUCCESS;
}

// ---- device-side comparison (reference Matrix.isequal / isclose, core/matrix.py:373-467: same shape, same pattern, values equal /
//      within max(rel_tol * max(|a|, |b|), abs_tol) after a cast to a common type) -- nothing travels to the host but the verdict ----------------
template <typename T>
__global__ void k_mat_compare(const int64_t *Ap, const int64_t *Bp, int64_t m1, const int32_t *Aj, const int32_t *Bj, const T *Ax, int a_iso,
                              const T *Bx, int b_iso, int64_t nnz, double rel_tol, double abs_tol, int *differ)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < m1) bad = Ap[i] != Bp[i];
    if (i < nnz) {
        bad = bad || Aj[i] != Bj[i];
        const T x = Ax[a_iso ? 0 : i], y = Bx[b_iso ? 0 : i];
        if (rel_tol == 0.0 && abs_tol == 0.0) bad = bad || !(x == y);
        else {
            const double dx = (double)x, dy = (double)y;
            // the reference's _isclose (core/operator/binary.py:329): x == y or |x - y| <= max(rel_tol * max(|x|, |y|), abs_tol)
            const double d = dx > dy ? dx - dy : dy - dx, ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
            const double rt = rel_tol * (ax > ay ? ax : ay);
            bad = bad || !(dx == dy || d <= (rt > abs_tol ? rt : abs_tol));
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) *differ = 1;
}

extern "C" GrB_Info GrX_Matrix_isclose(bool *result, const GrB_Matrix A, const GrB_Matrix B, double rel_tol, double abs_tol)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    check_matrix(B, "B");
    if (!result) fail(GrB_NULL_POINTER, "result is NULL");
    *result = false;
    if (A->nrows != B->nrows || A->ncols != B->ncols || A->nvals != B->nvals) return GrB_SUCCESS;
    if (A->nvals == 0) {
        *result = true;
        return GrB_SUCCESS;
    }
    // values in a common type: the wider of the two (FP64 when they differ in kind)
    int ct = A->type->code;
    if (B->type->code != ct) ct = (A->type->code >= TC_FP32 || B->type->code >= TC_FP32) ? TC_FP64 : (A->type->size >= B->type->size ? A->type->code : B->type->code);
    struct Owned {  // (typecast copies are released on every way out, also when a launch or the read-back throws)
        GB_Matrix_opaque *p = nullptr;
        ~Owned() { if (p) matrix_free(p); }
    } Ac, Bc;
    if (ct != A->type->code) Ac.p = matrix_cast_copy(A, ct);
    if (ct != B->type->code) Bc.p = matrix_cast_copy(B, ct);
    const GB_Matrix_opaque *X = Ac.p ? Ac.p : A, *Y = Bc.p ? Bc.p : B;
    DevBuf<int> differ(1, true);
    const int64_t threads = std::max<int64_t>((int64_t)A->nrows + 1, A->nvals);
    GRB_DISPATCH_TYPE(ct, T, {
        LAUNCH((k_mat_compare<T>), threads, (const int64_t *)X->d_ptr, (const int64_t *)Y->d_ptr, (int64_t)A->nrows + 1, (const int32_t *)X->d_col,
               (const int32_t *)Y->d_col, (const T *)X->d_val, X->iso ? 1 : 0, (const T *)Y->d_val, Y->iso ? 1 : 0, A->nvals, rel_tol, abs_tol, differ.p);
    })
    int h = 0;
    d2h(&h, differ.p, sizeof(int));
    *result = h == 0;
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrB_transpose(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A,
                                  const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_matrix(C, "C");
    check_matrix(A, "A");
    if (Mask) check_matrix(Mask, "Mask");
    const bool t0 = desc && desc->t0;
    const bool replace = desc && desc->replace, comp = desc && desc->comp, structure = desc && desc->structure;
    GB_Matrix_opaque *S = t0 ? A : matrix_transpose_cached(A);  // transposing a transpose is a copy
    if (C->nrows != S->nrows || C->ncols != S->ncols) fail(GrB_DIMENSION_MISMATCH, "GrB_transpose: output shape mismatch");
    if (Mask && (Mask->nrows != C->nrows || Mask->ncols != C->ncols)) fail(GrB_DIMENSION_MISMATCH, "GrB_transpose: mask shape does not match the output");
    if (accum && (accum->type != C->type->code || op_is_comparison(accum->op))) fail(GrB_DOMAIN_MISMATCH, "GrB_transpose: accum operator type must equal the output type");
    if (!Mask && comp) {  // complement of "no mask": nothing may be written
        if (replace) matrix_release_storage(C);
        return GrB_SUCCESS;
    }
    // T = the (transposed) copy in C's type -- a fresh object, so C may alias A or the mask -- then the write rule
    // (reference core/base.py:401-411: C(mask, accum, replace) << A.T)
    GB_Matrix_opaque *copy = matrix_cast_copy(S, C->type->code);
    if (copy->iso && copy->nvals && (Mask || accum)) {  // (the write rule reads one value per entry)
        GRB_DISPATCH_TYPE(C->type->code, TC_, {
            void *full = matrix_values_expanded<TC_>(copy);
            dev_free(copy->d_val);
            copy->d_val = full;
        })
        copy->iso = false;
    }
    try {
        if (C == A) matrix_invalidate_caches(C);
        matrix_apply_write_rule(C, Mask, accum, copy, replace, comp, structure);
    } catch (...) {
        matrix_free(copy);
        throw;
    }
    matrix_free(copy);
    GRB_CATCH(errp(C))
}

#define DEF_TYPED_ABI(NAME, ctype)                                                                                     \
    extern "C" GrB_Info GrB_Matrix_build_##NAME(GrB_Matrix C, const GrB_Index *I, const GrB_Index *J, const ctype *X,  \
                                                GrB_Index nvals, const GrB_BinaryOp dup)                               \
    {                                                                                                                  \
        GRB_TRY                                                                                                        \
        require_init();                                                                                                \
        check_matrix(C, "C");                                                                                          \
        GRB_DISPATCH_TYPE(C->type->code, T, { matrix_build_typed<T>(C, I, J, X, TC_##NAME, (int64_t)nvals, dup); })    \
        GRB_CATCH(errp(C))                                                   \
    }                                                                                                                  \
    extern "C" GrB_Info GrB_Matrix_extractTuples_##NAME(GrB_Index *I, GrB_Index *J, ctype *X, GrB_Index *nvals,        \
                                                        const GrB_Matrix A)                                            \
    {                                                                                                                  \
        GRB_TRY                                                                                                        \
        require_init();                                                                                                \
        check_matrix(A, "A");                                                                                          \
        GRB_DISPATCH_TYPE(A->type->code, T, { matrix_extract_typed<T>(A, I, J, X, TC_##NAME, nvals); })                \
        GRB_CATCH(errp(A))                                                   \
    }                                                                                                                  \
    extern "C" GrB_Info GrB_Matrix_import_##NAME(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols,       \
                                                 const GrB_Index *Ap, const GrB_Index *Ai, const ctype *Ax,            \
                                                 GrB_Index Ap_len, GrB_Index Ai_len, GrB_Index Ax_len,                 \
                                                 GrB_Format format)                                                    \
    {                                                                                                                  \
        GRB_TRY                                                                                                        \
        require_init();                                                                                                \
        if (!A || !type) fail(GrB_NULL_POINTER, "GrB_Matrix_import: NULL argument");                                   \
        *A = nullptr;                                                                                                  \
        GRB_DISPATCH_TYPE(type->code, T, {                                                                             \
            *A = matrix_import_typed<T>(type, nrows, ncols, Ap, Ai, Ax, TC_##NAME, Ap_len, Ai_len, Ax_len, format);    \
        })                                                                                                             \
        GRB_CATCH(nullptr)                                                                                             \
    }                                                                                                                  \
    extern "C" GrB_Info GrB_Matrix_export_##NAME(GrB_Index *Ap, GrB_Index *Ai, ctype *Ax, GrB_Index *Ap_len,           \
                                                 GrB_Index *Ai_len, GrB_Index *Ax_len, GrB_Format format,              \
                                                 const GrB_Matrix A)                                                   \
    {                                                                                                                  \
        GRB_TRY                                                                                                        \
        require_init();                                                                                                \
        check_matrix(A, "A");                                                                                          \
        GRB_DISPATCH_TYPE(A->type->code, T, {                                                                          \
            matrix_export_typed<T>(A, Ap, Ai, Ax, TC_##NAME, Ap_len, Ai_len, Ax_len, format);                          \
        })                                                                                                             \
        GRB_CATCH(errp(A))                                                   \
    }                                                                                                                  \
    extern "C" GrB_Info GrB_Vector_build_##NAME(GrB_Vector w, const GrB_Index *I, const ctype *X, GrB_Index nvals,     \
                                                const GrB_BinaryOp dup)                                                \
    {                                                                                                                  \
        GRB_TRY                                                                                                        \
        require_init();                                                                                                \
        check_vector(w, "w");                                                                                          \
        GRB_DISPATCH_TYPE(w->type->code, T, { vector_build_typed<T>(w, I, X, TC_##NAME, (int64_t)nvals, dup); })       \
        GRB_CATCH(errp(w))                                                   \
    }                                                                                                                  \
    extern "C" GrB_Info GrB_Vector_extractTuples_##NAME(GrB_Index *I, ctype *X, GrB_Index *nvals, const GrB_Vector v)  \
    {                                                                                                                  \
        GRB_TRY                                                                                                        \
        require_init();                                                                                                \
        check_vector(v, "v");                                                                                          \
        GRB_DISPATCH_TYPE(v->type->code, T, { vector_extract_typed<T>(v, I, X, TC_##NAME, nvals); })                   \
        GRB_CATCH(errp(v))                                                   \
    }
GRB_FOR_EACH_TYPE(DEF_TYPED_ABI)
#undef DEF_TYPED_ABI

// ---- GrX device-resident import / export -------------------------------------------------------------------
extern "C" GrB_Info GrX_Matrix_import_CSR_device(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols,
                                                 const int64_t *d_Ap, const int32_t *d_Aj, const void *d_Ax,
                                                 GrB_Index nvals, int iso, int copy)
{
    GRB_TRY
    require_init();
    if (!A || !type) fail(GrB_NULL_POINTER, "GrX_Matrix_import_CSR_device: NULL argument");
    *A = nullptr;
    GB_Matrix_opaque *M = matrix_new(type, nrows, ncols);
    if (nvals == 0) { *A = M; return GrB_SUCCESS; }
    try {
        if (!d_Ap || !d_Aj || !d_Ax) fail(GrB_NULL_POINTER, "GrX_Matrix_import_CSR_device: NULL device pointer");
        check_index_width(nrows, ncols);
        const size_t nv = iso ? 1 : (size_t)nvals;
        if (copy) {
            M->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (nrows + 1));
            d2d(M->d_ptr, d_Ap, sizeof(int64_t) * (nrows + 1));
            M->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nvals);
            d2d(M->d_col, d_Aj, sizeof(int32_t) * (size_t)nvals);
            M->d_val = dev_alloc(type->size * nv);
            d2d(M->d_val, d_Ax, type->size * nv);
            M->owns = true;
        } else {
            M->d_ptr = const_cast<int64_t *>(d_Ap);
            M->d_col = const_cast<int32_t *>(d_Aj);
            M->d_val = const_cast<void *>(d_Ax);
            M->owns = false;
        }
        M->iso = iso != 0;
        M->nvals = (int64_t)nvals;
    } catch (...) {
        matrix_free(M);
        throw;
    }
    *A = M;
    GRB_CATCH(nullptr)
}

// ---- the reference's zero-copy ingress names (SuiteSparse GxB layer; reference graphblas/core/ss/matrix.py:1279-1349) ------------
// python-graphblas's ``Matrix.ss.import_csr`` / ``ss.pack_csr`` bind GxB_Matrix_import_CSR / GxB_Matrix_pack_CSR: host arrays the
// caller allocated with the library's allocator (GxB_init; default malloc) whose OWNERSHIP passes to the library -- SuiteSparse
// keeps them as the matrix, this library keeps its matrices in HBM: ONE host-to-device copy, then the host arrays are released with
// the registered deallocator and the caller's pointers are set to NULL, exactly what the reference's wrapper expects
// (it ``unclaim_buffer``s the numpy arrays after the call: nobody else would free them).  Sizes are in BYTES.  jumbled = the
// column indices inside a row are not sorted (sorted here either way).  The device-resident form without any copy is
// GrX_Matrix_import_CSR_device.
static void gxb_take_csr(GB_Matrix_opaque *M, GrB_Index **Ap, GrB_Index **Aj, void **Ax, GrB_Index Ap_size, GrB_Index Aj_size,
                         GrB_Index Ax_size, bool iso, const char *who)
{
    if (!Ap || !Aj || !Ax) fail(GrB_NULL_POINTER, std::string(who) + ": NULL argument");
    if (!*Ap) fail(GrB_NULL_POINTER, std::string(who) + ": *Ap is NULL");
    if (Ap_size < (M->nrows + 1) * sizeof(GrB_Index)) fail(GrB_INVALID_VALUE, std::string(who) + ": Ap_size too small");
    const uint64_t nnz = (*Ap)[M->nrows];
    if (nnz) check_index_width(M->nrows, M->ncols);
    GRB_DISPATCH_TYPE(M->type->code, T, {
        matrix_import_into<T>(M, *Ap, *Aj, *Ax, M->type->code, Ap_size / sizeof(GrB_Index), Aj_size / sizeof(GrB_Index),
                              Ax_size / M->type->size, GrB_CSR_FORMAT, iso, who);
    })
    sync_stream();  // (the copies have left the host arrays)
    void (*release)(void *) = ctx().host_free ? ctx().host_free : free;
    release(*Ap);
    if (*Aj) release(*Aj);
    if (*Ax) release(*Ax);
    *Ap = nullptr;
    *Aj = nullptr;
    *Ax = nullptr;
}

extern "C" GrB_Info GxB_Matrix_import_CSR(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols, GrB_Index **Ap,
                                          GrB_Index **Aj, void **Ax, GrB_Index Ap_size, GrB_Index Aj_size, GrB_Index Ax_size,
                                          bool iso, bool jumbled, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    (void)jumbled;
    (void)desc;  // (secure_import: the indices are range-checked on every import)
    if (!A || !type) fail(GrB_NULL_POINTER, "GxB_Matrix_import_CSR: NULL argument");
    *A = nullptr;
    GB_Matrix_opaque *M = matrix_new(type, nrows, ncols);
    try {
        gxb_take_csr(M, Ap, Aj, Ax, Ap_size, Aj_size, Ax_size, iso, "GxB_Matrix_import_CSR");
    } catch (...) {
        matrix_free(M);
        throw;
    }
    *A = M;
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GxB_Matrix_pack_CSR(GrB_Matrix A, GrB_Index **Ap, GrB_Index **Aj, void **Ax, GrB_Index Ap_size,
                                        GrB_Index Aj_size, GrB_Index Ax_size, bool iso, bool jumbled, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    (void)jumbled;
    (void)desc;
    check_matrix(A, "A");
    // pack replaces the content of an existing matrix; type and shape stay.  The arrays are validated and copied into a fresh object
    // first: a call that fails (a NULL cell, a size that is too small, an index out of range) leaves A as it was and the caller's
    // arrays with the caller (SuiteSparse validates before it frees the old content).
    GB_Matrix_opaque *M = matrix_new(A->type, A->nrows, A->ncols);
    try {
        gxb_take_csr(M, Ap, Aj, Ax, Ap_size, Aj_size, Ax_size, iso, "GxB_Matrix_pack_CSR");
    } catch (...) {
        matrix_free(M);
        throw;
    }
    matrix_release_storage(A);
    A->d_ptr = M->d_ptr; A->d_col = M->d_col; A->d_val = M->d_val;
    A->nvals = M->nvals; A->iso = M->iso; A->owns = M->owns;
    M->d_ptr = nullptr; M->d_col = nullptr; M->d_val = nullptr; M->nvals = 0;
    matrix_free(M);
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrX_Matrix_export_CSR_device(const int64_t **d_Ap, const int32_t **d_Aj, const void **d_Ax,
                                                 GrB_Index *nvals, int *iso, const GrB_Matrix A)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    if (d_Ap) *d_Ap = matrix_rowptr(A);
    if (d_Aj) *d_Aj = A->d_col;
    if (d_Ax) *d_Ax = A->d_val;
    if (nvals) *nvals = (GrB_Index)A->nvals;
    if (iso) *iso = A->iso ? 1 : 0;
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrX_Matrix_cache_transpose(GrB_Matrix A)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    (void)matrix_transpose_cached(A);
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrX_Vector_import_dense_device(GrB_Vector *v, GrB_Type type, GrB_Index n, const void *d_val,
                                                   const uint32_t *d_present)
{
    GRB_TRY
    require_init();
    if (!v || !type) fail(GrB_NULL_POINTER, "GrX_Vector_import_dense_device: NULL argument");
    *v = nullptr;
    GB_Vector_opaque *w = vector_new(type, n);
    try {
        if (n) {
            if (!d_val) fail(GrB_NULL_POINTER, "GrX_Vector_import_dense_device: NULL device pointer");
            vector_ensure_storage(w);
            d2d(w->d_val, d_val, (size_t)n * type->size);
            if (d_present) {
                d2d(w->d_bits, d_present, (size_t)((n + 31) / 32) * 4);
                w->nvals = -1;
                // clear any bits >= n the caller may have left in the last word
                if (n & 63) {
                    uint64_t last = 0;
                    d2h(&last, w->d_bits + (bits_words64(n) - 1), 8);
                    last &= (1ull << (n & 63)) - 1;
                    h2d(w->d_bits + (bits_words64(n) - 1), &last, 8);
                }
            } else {
                LAUNCH(k_full_bits, (int64_t)bits_words64(n), w->d_bits, (int64_t)n);
                w->nvals = (int64_t)n;
            }
        }
    } catch (...) {
        vector_free(w);
        throw;
    }
    *v = w;
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_Vector_export_dense_device(const void **d_val, const uint32_t **d_present, const GrB_Vector v)
{
    GRB_TRY
    require_init();
    check_vector(v, "v");
    vector_ensure_storage(v);
    v->exported = true;  // (from now on conversions between vertex orders keep these pointers)
    // An export pins: the caller aliases the image for as long as it likes, and a product with an ordered matrix would otherwise permute
    // the very buffers it reads.  GrX_Vector_pin_natural(v, 0) hands the vector back.
    v->pinned = true;
    if (d_val) *d_val = v->d_val;
    if (d_present) *d_present = (const uint32_t *)v->d_bits;
    GRB_CATCH(errp(v))
}

extern "C" GrB_Info GrX_Vector_pin_natural(GrB_Vector v, int pinned)
{
    GRB_TRY
    require_init();
    check_vector(v, "v");  // (natural from now on)
    v->pinned = pinned != 0;
    GRB_CATCH(errp(v))
}

extern "C" GrB_Info GrX_Vector_modified(GrB_Vector v)
{
    GRB_TRY
    check_vector(v, "v");
    v->nvals = -1;  // the caller wrote the HBM image obtained from GrX_Vector_export_dense_device
    GRB_CATCH(errp(v))
}

// ---------------------------------------------------------------------------------------------------
// resize (reference: Matrix.resize core/matrix.py:512-523, Vector.resize core/vector.py:455-463): growing adds empty
// positions, shrinking drops the entries beyond the new bounds
// ---------------------------------------------------------------------------------------------------
namespace grb {
__global__ void k_trim_bits(uint64_t *bits, int64_t n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && (n & 63)) bits[n >> 6] &= (1ull << (n & 63)) - 1ull;
}
// keep[p] = entry p survives (its row < new_rows and its column < new_cols); row of p by binary search in the row pointers
__global__ void k_resize_keep(const int64_t *ptr, const int32_t *col, int64_t old_rows, int64_t nnz, int64_t new_rows, int64_t new_cols,
                              int64_t *keep)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > nnz) return;
    if (p == nnz) { keep[p] = 0; return; }
    int64_t lo = 0, hi = old_rows;  // last row with ptr[row] <= p
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (ptr[mid] <= p) lo = mid;
        else hi = mid;
    }
    keep[p] = (lo < new_rows && (int64_t)col[p] < new_cols) ? 1 : 0;
}
__global__ void k_resize_rowptr(const int64_t *old_ptr, const int64_t *pos, int64_t old_rows, int64_t new_rows, int64_t *new_ptr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= new_rows) new_ptr[i] = pos[old_ptr[i < old_rows ? i : old_rows]];
}
template <typename T>
__global__ void k_resize_scatter(const int64_t *pos, int64_t nnz, const int32_t *col, const T *val, int iso, int32_t *ncol, T *nval)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz && pos[p + 1] != pos[p]) {
        ncol[pos[p]] = col[p];
        if (!iso) nval[pos[p]] = val[p];
    }
}
}  // namespace grb

extern "C" GrB_Info GrB_Vector_resize(GrB_Vector v, GrB_Index new_size)
{
    GRB_TRY
    require_init();
    check_vector(v, "v");
    if (new_size > GrB_INDEX_MAX + 1) fail(GrB_INVALID_VALUE, "GrB_Vector_resize: size exceeds GrB_INDEX_MAX+1");
    if (new_size == v->n) return GrB_SUCCESS;
    if (!v->d_val) {
        v->n = new_size;
        return GrB_SUCCESS;
    }
    if (new_size > (1ull << 40)) fail(GrB_OUT_OF_MEMORY, "dense-with-presence vector of this size does not fit in HBM");
    const uint64_t old_n = v->n;
    void *old_val = v->d_val;
    uint64_t *old_bits = v->d_bits;
    const bool old_padded = v->padded;
    v->n = new_size;
    const bool padded = (int64_t)((size_t)new_size * v->type->size) >= ctx().vec_pad_min_bytes;
    void *nv = nullptr;
    uint64_t *nb = nullptr;
    try {
        vector_alloc_pair(v, padded, true, &nv, &nb);
    } catch (...) {
        v->n = old_n;
        throw;
    }
    const uint64_t keep = std::min<uint64_t>(old_n, new_size);
    d2d(nv, old_val, (size_t)keep * v->type->size);
    d2d(nb, old_bits, bits_words64(keep) * 8);
    if (new_size < old_n) {
        LAUNCH(k_trim_bits, 1, nb, (int64_t)new_size);
        v->nvals = -1;
    }
    vector_free_pair(old_padded, old_val, old_bits);
    v->d_val = nv;
    v->d_bits = nb;
    v->padded = padded;
    if (ctx().blocking) sync_stream();
    GRB_CATCH(errp(v))
}

extern "C" GrB_Info GrB_Matrix_resize(GrB_Matrix A, GrB_Index new_nrows, GrB_Index new_ncols)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    if (new_nrows > GrB_INDEX_MAX + 1 || new_ncols > GrB_INDEX_MAX + 1) fail(GrB_INVALID_VALUE, "GrB_Matrix_resize: dimension exceeds GrB_INDEX_MAX+1");
    if (new_nrows == A->nrows && new_ncols == A->ncols) return GrB_SUCCESS;
    if (A->nvals == 0 || !A->d_ptr) {
        matrix_release_storage(A);
        A->nrows = new_nrows;
        A->ncols = new_ncols;
        return GrB_SUCCESS;
    }
    check_index_width(new_nrows, std::min<uint64_t>(new_ncols, A->ncols));  // (surviving columns are below both bounds)
    const int64_t nnz = A->nvals, old_rows = (int64_t)A->nrows;
    DevBuf<int64_t> pos(nnz + 1);
    LAUNCH(k_resize_keep, nnz + 1, (const int64_t *)A->d_ptr, (const int32_t *)A->d_col, old_rows, nnz, (int64_t)new_nrows,
           (int64_t)std::min<uint64_t>(new_ncols, 0x7fffffffull), pos.p);
    prim_exclusive_sum_i64(pos.p, pos.p, nnz + 1);
    int64_t kept = 0;
    d2h(&kept, pos.p + nnz, sizeof(int64_t));
    GB_Matrix_opaque *N = matrix_new(A->type, new_nrows, new_ncols);
    try {
        if (kept > 0) {
            if (new_ncols > 0x7fffffffull) fail(GrB_NOT_IMPLEMENTED, "matrices with entries need ncols < 2^31 (int32 column indices)");
            N->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(new_nrows + 1));
            LAUNCH(k_resize_rowptr, (int64_t)new_nrows + 1, (const int64_t *)A->d_ptr, (const int64_t *)pos.p, old_rows, (int64_t)new_nrows,
                   N->d_ptr);
            N->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)kept);
            N->d_val = dev_alloc(A->type->size * (size_t)(A->iso ? 1 : kept));
            if (A->iso) d2d(N->d_val, A->d_val, A->type->size);
            GRB_DISPATCH_TYPE(A->type->code, T, {
                LAUNCH((k_resize_scatter<T>), nnz, (const int64_t *)pos.p, nnz, (const int32_t *)A->d_col, (const T *)A->d_val, A->iso ? 1 : 0,
                       N->d_col, (T *)N->d_val);
            })
            N->nvals = kept;
            N->iso = A->iso;
            N->owns = true;
        }
        sync_stream();  // `pos` is released at the end of this scope
        matrix_release_storage(A);  // (also drops the cached transpose, tile table, hot table and split)
        A->nrows = new_nrows;
        A->ncols = new_ncols;
        A->d_ptr = N->d_ptr; A->d_col = N->d_col; A->d_val = N->d_val;
        A->nvals = N->nvals; A->iso = N->iso; A->owns = true;
        N->d_ptr = nullptr; N->d_col = nullptr; N->d_val = nullptr; N->nvals = 0;
    } catch (...) {
        matrix_free(N);
        throw;
    }
    matrix_free(N);
    GRB_CATCH(errp(A))
}

namespace grb {
void preload_object() { hipFuncAttributes at; (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_popcount_sum)); (void)hipGetLastError(); }
}  // namespace grb
