// grb_vecops.hip -- the vector operations that close the BFS / SSSP loops around the mxv / vxm path on the device
// (reference notebooks/Example B.1 -- Level BFS.ipynb, docs/getting_started/primer.rst:236-246):
//
//   q[s] << True                      GrB_Vector_setElement_<T>      (reference core/vector.py:1880-1910)
//   v[i].new()                        GrB_Vector_extractElement_<T>  (core/vector.py:1840-1866)
//   del v[i]                          GrB_Vector_removeElement       (core/vector.py:1916-1930)
//   v[:](mask=q.V) << d               GrB_Vector_assign_<T> over GrB_ALL (scalar assign, core/vector.py:1979-2035)
//   succ << q.reduce(monoid.lor)      GrB_Vector_reduce_<T>          (core/vector.py:1635-1684)
//
// Vectors are dense-with-presence in HBM (values + bit-packed presence), so these are element-wise kernels over the
// presence words: one wavefront per 64 elements, the new presence word from one __ballot.
#include <vector>

#include "grb_internal.hpp"
#include "grb_ops.hpp"

namespace grb {

template <typename T>
__global__ void k_set_element(T *val, uint64_t *bits, int64_t i, T x)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        val[i] = x;
        bits[i >> 6] |= 1ull << (i & 63);
    }
}
__global__ void k_remove_element(uint64_t *bits, int64_t i)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) bits[i >> 6] &= ~(1ull << (i & 63));
}

// w<m, replace> = accum(w, s) for every index: one wavefront per presence word
template <typename T>
__global__ void k_assign_all(int64_t n, T *val, uint64_t *bits, const uint64_t *m_bits, int has_mask, int m_comp, int accum,
                             int replace, T s)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t g = i >> 6;
    const int64_t nwords = (n + 63) >> 6;
    bool new_has = false;
    if (i < n) {
        const bool old_has = (bits[g] >> lane) & 1ull;
        bool mact = true;
        if (has_mask) {
            mact = (m_bits[g] >> lane) & 1ull;
            if (m_comp) mact = !mact;
        }
        if (!mact) {
            new_has = replace ? false : old_has;
        } else {
            val[i] = (accum >= 0 && old_has) ? apply_binop<T>(accum, val[i], s) : s;
            new_has = true;
        }
    }
    const unsigned long long b = __ballot(new_has);
    if (lane == 0 && g < nwords) bits[g] = b;
}

// val[i] = ident wherever bit i is clear (the partial product of a rank before the monoid all-reduce, sharded.allreduce_monoid)
template <typename T>
__global__ void k_fill_absent(T *val, const uint64_t *bits, int64_t n, T ident)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !((bits[i >> 6] >> (i & 63)) & 1ull)) val[i] = ident;
}

template <typename W>
__global__ void k_fill_one(W *p, W v)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) p[0] = v;
}

// out[0] = monoid-fold of the present values (out starts at the identity)
template <typename T>
__global__ void k_reduce(const T *val, const uint64_t *bits, int64_t n, int monoid, typename Widen<T>::type *out)
{
    using W = typename Widen<T>::type;
    W acc = monoid_identity<T, W>(monoid);
    bool any = false;
    // a thread takes whole presence words: empty words (most of a BFS frontier) cost one 8-byte load and no value loads
    const int64_t nwords = (n + 63) >> 6;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nwords; g += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long b = bits[g];
        while (b) {
            const int t = __ffsll(b) - 1;
            b &= b - 1;
            const W x = (W)val[(g << 6) + t];
            acc = any ? apply_binop<W>(monoid, acc, x) : x;
            any = true;
        }
    }
    int has = any ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) {
        const W o = __shfl_down(acc, off);
        const int oh = __shfl_down(has, off);
        if (oh) {
            acc = has ? apply_binop<W>(monoid, acc, o) : o;
            has = 1;
        }
    }
    // one atomic per workgroup
    __shared__ W s_acc[4];
    __shared__ int s_has[4];
    if ((threadIdx.x & 63) == 0) {
        s_acc[threadIdx.x >> 6] = acc;
        s_has[threadIdx.x >> 6] = has;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        W t = monoid_identity<T, W>(monoid);
        int th = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
            if (s_has[w]) {
                t = th ? apply_binop<W>(monoid, t, s_acc[w]) : s_acc[w];
                th = 1;
            }
        }
        if (th) {
            if (monoid == OP_ANY) out[0] = t;
            else atomic_combine<W>(out, t, monoid);
        }
    }
}

// t = u (op) v on the union (eWiseAdd: a value present on one side only passes through) or the intersection (eWiseMult)
template <typename T>
__global__ void k_ewise(int64_t n, const T *u_val, const uint64_t *u_bits, const T *v_val, const uint64_t *v_bits, int op, int is_add,
                        T *t_val, uint64_t *t_bits)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t g = i >> 6;
    bool has = false;
    if (i < n) {
        const bool hu = u_bits && ((u_bits[g] >> lane) & 1ull), hv = v_bits && ((v_bits[g] >> lane) & 1ull);
        if (hu && hv) {
            t_val[i] = apply_binop<T>(op, u_val[i], v_val[i]);
            has = true;
        } else if (is_add && (hu || hv)) {
            t_val[i] = hu ? u_val[i] : v_val[i];
            has = true;
        }
    }
    const unsigned long long b = __ballot(has);
    if (lane == 0 && g < ((n + 63) >> 6)) t_bits[g] = b;
}

struct VDesc {
    bool replace = false, comp = false, structure = false;
};
static VDesc vflags(const GB_Descriptor_opaque *d)
{
    VDesc f;
    if (d) { f.replace = d->replace; f.comp = d->comp; f.structure = d->structure; }
    return f;
}

template <typename T>
static void set_element(GB_Vector_opaque *w, T x, uint64_t i)
{
    if (i >= w->n) fail(GrB_INVALID_INDEX, "setElement: index " + std::to_string(i) + " is outside a vector of size " + std::to_string(w->n));
    vector_ensure_storage(w);
    i = vector_position(w, i);  // (a vector kept in a matrix's vertex order: where element i lives)
    GRB_DISPATCH_TYPE(w->type->code, TW, {
        hipLaunchKernelGGL((k_set_element<TW>), dim3(1), dim3(64), 0, ctx().stream, (TW *)w->d_val, w->d_bits, (int64_t)i,
                           cast_value<TW, T>(x));
    })
    w->nvals = -1;
    if (ctx().blocking) sync_stream();
}

template <typename T>
static GrB_Info extract_element(T *x, GB_Vector_opaque *u, uint64_t i)
{
    if (!x) fail(GrB_NULL_POINTER, "extractElement: output pointer is NULL");
    if (i >= u->n) fail(GrB_INVALID_INDEX, "extractElement: index " + std::to_string(i) + " is outside a vector of size " + std::to_string(u->n));
    if (!u->d_val) return GrB_NO_VALUE;
    i = vector_position(u, i);
    uint64_t word = 0;
    d2h(&word, u->d_bits + (i >> 6), sizeof(word));
    if (!((word >> (i & 63)) & 1ull)) return GrB_NO_VALUE;
    GRB_DISPATCH_TYPE(u->type->code, TU, {
        TU v;
        d2h(&v, (const TU *)u->d_val + i, sizeof(TU));
        *x = cast_value<T, TU>(v);
    })
    return GrB_SUCCESS;
}

template <typename T>
static void assign_all(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, T x, const GB_Descriptor_opaque *desc)
{
    const VDesc f = vflags(desc);
    if (mask && mask->n != w->n) fail(GrB_DIMENSION_MISMATCH, "assign: mask size does not match the output size");
    if (accum && (accum->type != w->type->code || op_is_comparison(accum->op))) fail(GrB_DOMAIN_MISMATCH, "assign: accum operator type must equal the output type");
    if (!mask && f.comp) {  // complement of "no mask": nothing may be written
        if (f.replace) vector_release_storage(w);
        return;
    }
    if (w->n == 0) return;
    {   // element-wise: any vertex order will do, as long as it is the same for w and the mask
        GB_Vector_opaque *vs[2] = {w, mask};
        (void)vectors_common_order(vs, 2);
    }
    vector_ensure_storage(w);
    DevBuf<uint64_t> mbits(mask ? bits_words64(w->n) : 1);
    if (mask) vector_mask_bits(mask, f.structure, mbits.p);  // (a snapshot: the mask may be w itself)
    GRB_DISPATCH_TYPE(w->type->code, TW, {
        const int64_t threads = (int64_t)bits_words64(w->n) * 64;
        hipLaunchKernelGGL((k_assign_all<TW>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, ctx().stream, (int64_t)w->n,
                           (TW *)w->d_val, w->d_bits, (const uint64_t *)mbits.p, mask ? 1 : 0, f.comp ? 1 : 0,
                           accum ? canonical_op(w->type->code, accum->op) : -1, f.replace ? 1 : 0, cast_value<TW, T>(x));
    })
    w->nvals = !mask ? (int64_t)w->n : -1;
    if (ctx().blocking) sync_stream();  // (mbits goes back to the stream-ordered block cache: no wait needed)
}

template <typename T>
static void reduce_to(T *val, const GB_BinaryOp_opaque *accum, const GB_Monoid_opaque *monoid, GB_Vector_opaque *u)
{
    if (!val) fail(GrB_NULL_POINTER, "reduce: output pointer is NULL");
    if (!monoid) fail(GrB_NULL_POINTER, "reduce: monoid is NULL");
    const int mt = monoid->type;
    const int op = canonical_op(mt, monoid->op);
    T t{};
    GRB_DISPATCH_TYPE(mt, TM, {
        using W = typename Widen<TM>::type;
        W h = monoid_identity<TM, W>(op);
        if (u->d_val && u->n > 0) {
            // values in the monoid's type
            DevBuf<char> cast_buf(0);
            const void *src = u->d_val;
            if (u->type->code != mt) {
                dev_free(cast_buf.p);
                cast_buf.p = (char *)dev_alloc(sizeof(TM) * (size_t)u->n);
                cast_array(mt, cast_buf.p, u->type->code, u->d_val, (int64_t)u->n);
                src = cast_buf.p;
            }
            DevBuf<W> out(1);
            hipLaunchKernelGGL((k_fill_one<W>), dim3(1), dim3(64), 0, ctx().stream, out.p, h);
            const int64_t blocks = std::min<int64_t>(ceil_div((int64_t)bits_words64(u->n), 256), (int64_t)ctx().num_cus * 4);
            hipLaunchKernelGGL((k_reduce<TM>), dim3((unsigned)blocks), dim3(256), 0, ctx().stream, (const TM *)src,
                               (const uint64_t *)u->d_bits, (int64_t)u->n, op, out.p);
            d2h(&h, out.p, sizeof(W));
        }
        const TM r = std::is_same<TM, bool>::value ? (TM)(h != (W)0) : (TM)h;
        t = cast_value<T, TM>(r);
    })
    if (accum) {
        // val = accum(val, t) in the accumulator's type (host side: one scalar)
        GRB_DISPATCH_TYPE(accum->type, TA, {
            *val = cast_value<T, TA>(apply_binop<TA>(canonical_op(accum->type, accum->op), cast_value<TA, T>(*val), cast_value<TA, T>(t)));
        })
    } else {
        *val = t;
    }
}

template <typename T>
__device__ __forceinline__ bool apply_cmp(int op, T a, T b)
{
    switch (op) {
    case OP_EQ: return a == b;
    case OP_NE: return a != b;
    case OP_GT: return a > b;
    case OP_LT: return a < b;
    case OP_GE: return a >= b;
    default: return a <= b;
    }
}
// comparison operators: t (BOOL) = u cmp v on the intersection; eWiseAdd passes single entries through cast to BOOL
template <typename T>
__global__ void k_ewise_cmp(int64_t n, const T *u_val, const uint64_t *u_bits, const T *v_val, const uint64_t *v_bits, int op, int is_add,
                            bool *t_val, uint64_t *t_bits)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t g = i >> 6;
    bool has = false;
    if (i < n) {
        const bool hu = u_bits && ((u_bits[g] >> lane) & 1ull), hv = v_bits && ((v_bits[g] >> lane) & 1ull);
        if (hu && hv) {
            t_val[i] = apply_cmp<T>(op, u_val[i], v_val[i]);
            has = true;
        } else if (is_add && (hu || hv)) {
            t_val[i] = (hu ? u_val[i] : v_val[i]) != (T)0;
            has = true;
        }
    }
    const unsigned long long b = __ballot(has);
    if (lane == 0 && g < ((n + 63) >> 6)) t_bits[g] = b;
}

// w<mask, replace> = accum(w, u (op) v), element-wise over the union (is_add) or the intersection of the patterns
// (reference core/vector.py:960-1150 -> GrB_Vector_eWiseAdd_* / eWiseMult_*)
static void ewise_core(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, int op_in, int ot,
                       GB_Vector_opaque *u, GB_Vector_opaque *v, const GB_Descriptor_opaque *desc, bool is_add)
{
    const VDesc f = vflags(desc);
    if (u->n != v->n) fail(GrB_DIMENSION_MISMATCH, "eWise: input sizes " + std::to_string(u->n) + " and " + std::to_string(v->n) + " differ");
    if (w->n != u->n) fail(GrB_DIMENSION_MISMATCH, "eWise: output size does not match the inputs");
    if (mask && mask->n != w->n) fail(GrB_DIMENSION_MISMATCH, "eWise: mask size does not match the output size");
    if (accum && (accum->type != w->type->code || op_is_comparison(accum->op))) fail(GrB_DOMAIN_MISMATCH, "eWise: accum operator type must equal the output type");
    if (!mask && f.comp) {
        if (f.replace) vector_release_storage(w);
        return;
    }
    if (w->n == 0) return;
    {   // element-wise: any vertex order will do, as long as all four vectors are in it
        GB_Vector_opaque *vs[4] = {w, u, v, mask};
        (void)vectors_common_order(vs, 4);
    }
    const int64_t n = (int64_t)w->n;
    const bool cmp = op_is_comparison(op_in);
    const int op = cmp ? op_in : canonical_op(ot, op_in);
    const int tt = cmp ? (int)TC_BOOL : ot;  // type of the element-wise result
    const size_t ob = type_size(ot);
    // operands in the operator's type
    DevBuf<char> u_cast(0), v_cast(0);
    const void *uv = u->d_val, *vv = v->d_val;
    if (u->d_val && u->type->code != ot) {
        dev_free(u_cast.p);
        u_cast.p = (char *)dev_alloc(ob * (size_t)n);
        cast_array(ot, u_cast.p, u->type->code, u->d_val, n);
        uv = u_cast.p;
    }
    if (v->d_val && v->type->code != ot) {
        dev_free(v_cast.p);
        v_cast.p = (char *)dev_alloc(ob * (size_t)n);
        cast_array(ot, v_cast.p, v->type->code, v->d_val, n);
        vv = v_cast.p;
    }
    DevBuf<char> t_val(std::max(ob, type_size(tt)) * (size_t)n);
    DevBuf<uint64_t> t_bits(bits_words64((uint64_t)n));
    const int64_t threads = (int64_t)bits_words64((uint64_t)n) * 64;
    GRB_DISPATCH_TYPE(ot, T, {
        if (cmp)
            hipLaunchKernelGGL((k_ewise_cmp<T>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, ctx().stream, n, (const T *)uv,
                               (const uint64_t *)(u->d_val ? u->d_bits : nullptr), (const T *)vv,
                               (const uint64_t *)(v->d_val ? v->d_bits : nullptr), op, is_add ? 1 : 0, (bool *)t_val.p, t_bits.p);
        else
            hipLaunchKernelGGL((k_ewise<T>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, ctx().stream, n, (const T *)uv,
                               (const uint64_t *)(u->d_val ? u->d_bits : nullptr), (const T *)vv,
                               (const uint64_t *)(v->d_val ? v->d_bits : nullptr), op, is_add ? 1 : 0, (T *)t_val.p, t_bits.p);
    })
    // the write rule in w's type (w may be u or v: t is a separate buffer, the rule is element-wise)
    DevBuf<uint64_t> mbits(mask ? bits_words64(w->n) : 1);
    if (mask) vector_mask_bits(mask, f.structure, mbits.p);
    vector_ensure_storage(w);
    DevBuf<char> tc(0);
    const void *tw = t_val.p;
    if (w->type->code != tt) {
        dev_free(tc.p);
        tc.p = (char *)dev_alloc(w->type->size * (size_t)n);
        cast_array(w->type->code, tc.p, tt, t_val.p, n);
        tw = tc.p;
    }
    vector_write_rule(w, tw, t_bits.p, mask ? mbits.p : nullptr, f.comp, accum ? canonical_op(w->type->code, accum->op) : -1, f.replace);
    w->nvals = -1;
    if (ctx().blocking) sync_stream();
}

// ---- assign / extract with an index list (reference core/vector.py:1906-2035 -> GrB_Vector_assign, GrB_Vector_assign_<T>,
//      GrB_Vector_extract; C API 2.0 sections 4.3.7 / 4.3.6) --------------------------------------------------------------
// Z = w; Z(I) = accum ? accum(w(I), u) : u   (no accumulator: positions of I without an entry in u lose theirs), then
// w<mask, replace> = Z over the WHOLE of w (GrB_assign, not subassign: the mask has w's size).  Duplicate indices in I are
// undefined behaviour in the specification; here the last writer wins per element.
template <typename T>
__global__ void k_assign_scatter(T *z_val, unsigned long long *z_bits, const uint64_t *I, int64_t ni, int64_t n, const T *u_val,
                                 const uint64_t *u_bits, T scalar, int accum, int *oob)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ni) return;
    const uint64_t i = I[k];
    if (i >= (uint64_t)n) {
        *oob = 1;
        return;
    }
    const bool up = u_bits ? ((u_bits[k >> 6] >> (k & 63)) & 1ull) : true;
    const T uv = u_val ? u_val[k] : scalar;
    const unsigned long long bit = 1ull << (i & 63);
    if (accum >= 0) {
        if (up) {
            const bool zp = (z_bits[i >> 6] & bit) != 0;
            z_val[i] = zp ? apply_binop<T>(accum, z_val[i], uv) : uv;
            if (!zp) atomicOr(&z_bits[i >> 6], bit);
        }
    } else if (up) {
        z_val[i] = uv;
        atomicOr(&z_bits[i >> 6], bit);
    } else {
        atomicAnd(&z_bits[i >> 6], ~bit);
    }
}
// t(k) = u(I[k])
template <typename T>
__global__ void k_extract_gather(T *t_val, uint64_t *t_bits, const uint64_t *I, int64_t ni, int64_t n, const T *u_val,
                                 const uint64_t *u_bits, int *oob)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    if (k < ni) {
        const uint64_t i = I[k];
        if (i >= (uint64_t)n) *oob = 1;
        else if (u_bits && ((u_bits[i >> 6] >> (i & 63)) & 1ull)) {
            has = true;
            t_val[k] = u_val[i];
        }
    }
    const unsigned long long b = __ballot(has);
    if ((threadIdx.x & 63) == 0 && (k >> 6) < ((ni + 63) >> 6)) t_bits[k >> 6] = b;
}

static void check_oob(int *d_flag, const char *what)
{
    int h = 0;
    d2h(&h, d_flag, sizeof(int));
    if (h) fail(GrB_INDEX_OUT_OF_BOUNDS, std::string(what) + ": an index is outside the vector");
}

// u == nullptr: scalar assign of `scalar_w` (already in w's type, as raw bytes)
static void assign_indexed(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, GB_Vector_opaque *u,
                           const void *scalar_w, const uint64_t *indices, uint64_t ni, const GB_Descriptor_opaque *desc)
{
    const VDesc f = vflags(desc);
    if (mask && mask->n != w->n) fail(GrB_DIMENSION_MISMATCH, "assign: mask size does not match the output size");
    if (u && u->n != ni) fail(GrB_DIMENSION_MISMATCH, "assign: the input has " + std::to_string(u->n) + " elements, the index list " + std::to_string(ni));
    if (accum && (accum->type != w->type->code || op_is_comparison(accum->op))) fail(GrB_DOMAIN_MISMATCH, "assign: accum operator type must equal the output type");
    if (!indices) fail(GrB_NULL_POINTER, "assign: index list is NULL");
    if (!mask && f.comp) {
        if (f.replace) vector_release_storage(w);
        return;
    }
    if (w->n == 0) return;
    vector_ensure_storage(w);
    const int wt = w->type->code;
    DevBuf<uint64_t> d_idx(ni);
    if (ni) h2d(d_idx.p, indices, sizeof(uint64_t) * (size_t)ni);
    // u in w's type
    DevBuf<char> u_cast(0);
    const void *uval = nullptr;
    const uint64_t *ubits = nullptr;
    if (u) {
        vector_ensure_storage(u);
        uval = u->d_val;
        ubits = u->d_bits;
        if (u->type->code != wt) {
            dev_free(u_cast.p);
            u_cast.p = (char *)dev_alloc(w->type->size * (size_t)std::max<uint64_t>(u->n, 1));
            cast_array(wt, u_cast.p, u->type->code, u->d_val, (int64_t)u->n);
            uval = u_cast.p;
        }
    }
    // Z: w itself when nothing masks the write, else a copy
    void *z_val = w->d_val;
    uint64_t *z_bits = w->d_bits;
    DevBuf<char> zv(0);
    DevBuf<uint64_t> zb(0);
    const bool in_place = !mask && !(u == w);
    DevBuf<char> u_snap(0);
    DevBuf<uint64_t> ub_snap(0);
    if (u == w) {  // (aliased input: read a snapshot)
        dev_free(u_snap.p); dev_free(ub_snap.p);
        u_snap.p = (char *)dev_alloc(w->type->size * (size_t)w->n);
        ub_snap.p = (uint64_t *)dev_alloc(bits_words64(w->n) * 8);
        d2d(u_snap.p, w->d_val, w->type->size * (size_t)w->n);
        d2d(ub_snap.p, w->d_bits, bits_words64(w->n) * 8);
        uval = u_snap.p;
        ubits = ub_snap.p;
    }
    if (!in_place && mask) {
        dev_free(zv.p); dev_free(zb.p);
        zv.p = (char *)dev_alloc(w->type->size * (size_t)w->n);
        zb.p = (uint64_t *)dev_alloc(bits_words64(w->n) * 8);
        d2d(zv.p, w->d_val, w->type->size * (size_t)w->n);
        d2d(zb.p, w->d_bits, bits_words64(w->n) * 8);
        z_val = zv.p;
        z_bits = zb.p;
    }
    DevBuf<int> oob(1, true);
    if (ni) {
        GRB_DISPATCH_TYPE(wt, TW, {
            TW sc{};
            if (scalar_w) memcpy(&sc, scalar_w, sizeof(TW));
            hipLaunchKernelGGL((k_assign_scatter<TW>), dim3((unsigned)ceil_div((int64_t)ni, 256)), dim3(256), 0, ctx().stream, (TW *)z_val,
                               (unsigned long long *)z_bits, (const uint64_t *)d_idx.p, (int64_t)ni, (int64_t)w->n, (const TW *)uval, ubits,
                               sc, accum ? canonical_op(wt, accum->op) : -1, oob.p);
        })
    }
    check_oob(oob.p, "assign");
    if (mask) {
        DevBuf<uint64_t> mbits(bits_words64(w->n));
        vector_mask_bits(mask, f.structure, mbits.p);
        vector_write_rule(w, z_val, z_bits, mbits.p, f.comp, -1, f.replace);
    }
    w->nvals = -1;
    sync_stream();  // (the index list was a host array borrowed for the call; temporaries go back to the block cache)
}

static void extract_indexed(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, GB_Vector_opaque *u,
                            const uint64_t *indices, uint64_t ni, const GB_Descriptor_opaque *desc)
{
    const VDesc f = vflags(desc);
    if (w->n != ni) fail(GrB_DIMENSION_MISMATCH, "extract: the output has " + std::to_string(w->n) + " elements, the index list " + std::to_string(ni));
    if (mask && mask->n != w->n) fail(GrB_DIMENSION_MISMATCH, "extract: mask size does not match the output size");
    if (accum && (accum->type != w->type->code || op_is_comparison(accum->op))) fail(GrB_DOMAIN_MISMATCH, "extract: accum operator type must equal the output type");
    if (!indices) fail(GrB_NULL_POINTER, "extract: index list is NULL");
    if (!mask && f.comp) {
        if (f.replace) vector_release_storage(w);
        return;
    }
    if (ni == 0) return;
    DevBuf<uint64_t> d_idx(ni);
    h2d(d_idx.p, indices, sizeof(uint64_t) * (size_t)ni);
    const int ut = u->type->code;
    DevBuf<char> t_val(u->type->size * (size_t)ni);
    DevBuf<uint64_t> t_bits(bits_words64(ni), true);
    DevBuf<int> oob(1, true);
    GRB_DISPATCH_TYPE(ut, TU, {
        hipLaunchKernelGGL((k_extract_gather<TU>), dim3((unsigned)ceil_div((int64_t)bits_words64(ni) * 64, 256)), dim3(256), 0, ctx().stream,
                           (TU *)t_val.p, t_bits.p, (const uint64_t *)d_idx.p, (int64_t)ni, (int64_t)u->n, (const TU *)u->d_val,
                           (const uint64_t *)(u->d_val ? u->d_bits : nullptr), oob.p);
    })
    check_oob(oob.p, "extract");
    vector_ensure_storage(w);
    DevBuf<char> tc(0);
    const void *tw = t_val.p;
    if (w->type->code != ut) {
        dev_free(tc.p);
        tc.p = (char *)dev_alloc(w->type->size * (size_t)ni);
        cast_array(w->type->code, tc.p, ut, t_val.p, (int64_t)ni);
        tw = tc.p;
    }
    DevBuf<uint64_t> mbits(mask ? bits_words64(w->n) : 1);
    if (mask) vector_mask_bits(mask, f.structure, mbits.p);
    vector_write_rule(w, tw, t_bits.p, mask ? mbits.p : nullptr, f.comp, accum ? canonical_op(w->type->code, accum->op) : -1, f.replace);
    w->nvals = -1;
    sync_stream();
}

}  // namespace grb

using namespace grb;

extern "C" const uint64_t *GrB_ALL;

extern "C" GrB_Info GrB_Vector_assign(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u,
                                      const GrB_Index *indices, GrB_Index nindices, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_vector(w, "w");
    if (mask) check_vector(mask, "mask");
    check_vector(u, "u");
    if (indices == GrB_ALL) {
        // every index, in order: the identity list (a plain copy under the write rule)
        if (u->n != w->n) fail(GrB_DIMENSION_MISMATCH, "assign: input and output sizes differ");
        std::vector<uint64_t> all(w->n);
        for (uint64_t i = 0; i < w->n; i++) all[i] = i;
        assign_indexed(w, mask, accum, u, nullptr, all.data(), w->n, desc);
    } else {
        assign_indexed(w, mask, accum, u, nullptr, indices, nindices, desc);
    }
    GRB_CATCH(errp(w))
}

extern "C" GrB_Info GrB_Vector_extract(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u,
                                       const GrB_Index *indices, GrB_Index nindices, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_vector(w, "w");
    if (mask) check_vector(mask, "mask");
    check_vector(u, "u");
    if (indices == GrB_ALL) {
        std::vector<uint64_t> all(u->n);
        for (uint64_t i = 0; i < u->n; i++) all[i] = i;
        extract_indexed(w, mask, accum, u, all.data(), u->n, desc);
    } else {
        extract_indexed(w, mask, accum, u, indices, nindices, desc);
    }
    GRB_CATCH(errp(w))
}

// Write the monoid's identity into the positions of v that hold no entry (ANY, which has none: the smallest value of the type, so
// that a rank without the entry never wins a MAX) -- the values image can then go through an all-reduce as it is.  The presence words
// are untouched: v still holds the same entries.
extern "C" GrB_Info GrX_Vector_fill_absent(GrB_Vector v, const GrB_Monoid monoid)
{
    GRB_TRY
    require_init();
    check_vector(v, "v");
    if (!monoid) fail(GrB_NULL_POINTER, "monoid is NULL");
    if (monoid->type != v->type->code) fail(GrB_DOMAIN_MISMATCH, "GrX_Vector_fill_absent: the monoid's type must be the vector's");
    if (v->n == 0) return GrB_SUCCESS;
    vector_ensure_storage(v);
    const int op = canonical_op(monoid->type, monoid->op);
    GRB_DISPATCH_TYPE(v->type->code, T, {
        const T ident = op == OP_ANY ? (std::is_same<T, bool>::value ? (T)0 : type_lowest<T>()) : (T)monoid_identity<T, T>(op);
        hipLaunchKernelGGL((k_fill_absent<T>), dim3((unsigned)ceil_div((int64_t)v->n, 256)), dim3(256), 0, ctx().stream, (T *)v->d_val,
                           (const uint64_t *)v->d_bits, (int64_t)v->n, ident);
    })
    if (ctx().blocking) sync_stream();
    GRB_CATCH(errp(v))
}

extern "C" GrB_Info GrB_Vector_removeElement(GrB_Vector w, GrB_Index i)
{
    GRB_TRY
    require_init();
    check_vector_any(w, "w");
    if (i >= w->n) fail(GrB_INVALID_INDEX, "removeElement: index " + std::to_string(i) + " is outside a vector of size " + std::to_string(w->n));
    if (w->d_val) {
        i = vector_position(w, i);
        hipLaunchKernelGGL(k_remove_element, dim3(1), dim3(64), 0, ctx().stream, w->d_bits, (int64_t)i);
        w->nvals = -1;
        if (ctx().blocking) sync_stream();
    }
    GRB_CATCH(errp(w))
}

#define GRB_VECOPS_TYPED(NAME, ctype)                                                                                                   \
    extern "C" GrB_Info GrB_Vector_setElement_##NAME(GrB_Vector w, ctype x, GrB_Index i)                                                \
    {                                                                                                                                   \
        GRB_TRY                                                                                                                         \
        require_init();                                                                                                                 \
        check_vector_any(w, "w");                                                                                                       \
        set_element<ctype>(w, x, i);                                                                                                    \
        GRB_CATCH(errp(w))                                                                                                              \
    }                                                                                                                                   \
    extern "C" GrB_Info GrB_Vector_extractElement_##NAME(ctype *x, const GrB_Vector u, GrB_Index i)                                     \
    {                                                                                                                                   \
        GRB_TRY                                                                                                                         \
        require_init();                                                                                                                 \
        check_vector_any(u, "u");                                                                                                       \
        return extract_element<ctype>(x, u, i);                                                                                         \
        GRB_CATCH(errp(u))                                                                                                              \
    }                                                                                                                                   \
    extern "C" GrB_Info GrB_Vector_assign_##NAME(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, ctype x,                \
                                                 const GrB_Index *indices, GrB_Index nindices, const GrB_Descriptor desc)               \
    {                                                                                                                                   \
        GRB_TRY                                                                                                                         \
        require_init();                                                                                                                 \
        check_vector_any(w, "w");                                                                                                       \
        if (mask) check_vector_any(mask, "mask");                                                                                       \
        if (indices != GrB_ALL) {                                                                                                       \
            check_vector(w, "w");                                                                                                       \
            if (mask) check_vector(mask, "mask");                                                                                       \
            GRB_DISPATCH_TYPE(w->type->code, TW_, {                                                                                     \
                const TW_ xs = cast_value<TW_, ctype>(x);                                                                               \
                assign_indexed(w, mask, accum, nullptr, &xs, indices, nindices, desc);                                                  \
            })                                                                                                                          \
        } else {                                                                                                                        \
            assign_all<ctype>(w, mask, accum, x, desc);                                                                                 \
        }                                                                                                                               \
        GRB_CATCH(errp(w))                                                                                                              \
    }                                                                                                                                   \
    extern "C" GrB_Info GrB_Vector_reduce_##NAME(ctype *val, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Vector u,     \
                                                 const GrB_Descriptor desc)                                                             \
    {                                                                                                                                   \
        GRB_TRY                                                                                                                         \
        require_init();                                                                                                                 \
        /* (a reduction does not care where the elements are -- except a floating-point PLUS / TIMES, whose rounding depends on the  */ \
        /*  order of the fold: those see the natural order, so the result does not depend on what an earlier product left behind)     */ \
        check_vector_any(u, "u");                                                                                                       \
        if (monoid && u->order && (monoid->type == TC_FP32 || monoid->type == TC_FP64)) {                                               \
            const int mop = canonical_op(monoid->type, monoid->op);                                                                     \
            if (mop == OP_PLUS || mop == OP_TIMES) check_vector(u, "u");                                                                \
        }                                                                                                                               \
        (void)desc;                                                                                                                     \
        reduce_to<ctype>(val, accum, monoid, u);                                                                                        \
        GRB_CATCH(errp(u))                                                                                                              \
    }
GRB_FOR_EACH_TYPE(GRB_VECOPS_TYPED)
#undef GRB_VECOPS_TYPED

#define GRB_EWISE(FUNC, HANDLE, IS_ADD)                                                                                        \
    extern "C" GrB_Info FUNC(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const HANDLE op, const GrB_Vector u,   \
                             const GrB_Vector v, const GrB_Descriptor desc)                                                      \
    {                                                                                                                          \
        GRB_TRY                                                                                                                \
        require_init();                                                                                                        \
        check_vector_any(w, "w");                                                                                              \
        if (mask) check_vector_any(mask, "mask");                                                                              \
        check_vector_any(u, "u");                                                                                              \
        check_vector_any(v, "v");                                                                                              \
        if (!op) fail(GrB_NULL_POINTER, "eWise: operator is NULL");                                                            \
        ewise_core(w, mask, accum, op->op, op->type, u, v, desc, IS_ADD);                                                      \
        GRB_CATCH(errp(w))                                                                                                     \
    }
GRB_EWISE(GrB_Vector_eWiseAdd_BinaryOp, GrB_BinaryOp, true)
GRB_EWISE(GrB_Vector_eWiseAdd_Monoid, GrB_Monoid, true)
GRB_EWISE(GrB_Vector_eWiseMult_BinaryOp, GrB_BinaryOp, false)
GRB_EWISE(GrB_Vector_eWiseMult_Monoid, GrB_Monoid, false)
#undef GRB_EWISE

namespace grb {
void preload_vecops() { hipFuncAttributes at; (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_remove_element)); (void)hipGetLastError(); }
}  // namespace grb
