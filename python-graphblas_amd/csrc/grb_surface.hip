// grb_surface.hip -- the import-time surface beyond the hot path (include/grb_mi355x.h, last section).
//
// An unmodified python-graphblas resolves, while it is imported, handles far outside mxm / mxv / vxm (graphblas/core/mask.py:1-5:
// select.valuene, unary.one; core/operator/base.py:803-893: every GrB_* / GxB_* operator name it finds) and binds one C entry
// point per operation.  This file makes those names exist:
//   * builtin unary, index-unary and the remaining binary operators as DATA symbols.  The kernels implement none of them: every
//     entry point that consumes an operator goes through canonical_op(), which rejects their codes with GrB_NOT_IMPLEMENTED;
//   * GrB_Scalar for real (a host-side value + presence: the reference passes scalars by GrB_Scalar handle in C API 2.0 calls);
//   * the entry points of the operations this library does not accelerate, with their C API 2.0 signatures, returning
//     GrB_NOT_IMPLEMENTED and leaving a message for GrB_*_error.
#include <cstring>

#include "grb_internal.hpp"
#include "grb_ops.hpp"

using namespace grb;

struct GB_UnaryOp_opaque {
    int op;
    int type;
    const char *name;
};
struct GB_IndexUnaryOp_opaque {
    int op;
    int type;
    const char *name;
};
constexpr uint64_t MAGIC_SCALAR = 0x4752425343414c52ULL;  // "GRBSCALR"
struct GB_Scalar_opaque {
    uint64_t magic;
    GrB_Type type;
    bool has;
    unsigned char value[8];
    std::string err;
};

// ---- operator handles (data symbols) ----------------------------------------------------------------------------------
#define DEF_UNOP(SYM, T)                                                    \
    static GB_UnaryOp_opaque uop_obj_##SYM = {OP_UNSUPPORTED, TC_##T, #SYM}; \
    extern "C" GrB_UnaryOp SYM = &uop_obj_##SYM;
#define DEF_IDXOP(SYM, T)                                                        \
    static GB_IndexUnaryOp_opaque iop_obj_##SYM = {OP_UNSUPPORTED, TC_##T, #SYM}; \
    extern "C" GrB_IndexUnaryOp SYM = &iop_obj_##SYM;
#define DEF_BINOP_X(SYM, OP, T)                                        \
    static GB_BinaryOp_opaque bopx_obj_##SYM = {OP, TC_##T, #SYM};     \
    extern "C" GrB_BinaryOp SYM = &bopx_obj_##SYM;
#define DEF_SURFACE_OPS(T)                       \
    DEF_UNOP(GrB_IDENTITY_##T, T)                \
    DEF_UNOP(GrB_AINV_##T, T)                    \
    DEF_UNOP(GrB_MINV_##T, T)                    \
    DEF_UNOP(GrB_ABS_##T, T)                     \
    DEF_UNOP(GxB_ONE_##T, T)                     \
    DEF_UNOP(GxB_LNOT_##T, T)                    \
    DEF_IDXOP(GrB_VALUEEQ_##T, T)                \
    DEF_IDXOP(GrB_VALUENE_##T, T)                \
    DEF_IDXOP(GrB_VALUEGT_##T, T)                \
    DEF_IDXOP(GrB_VALUEGE_##T, T)                \
    DEF_IDXOP(GrB_VALUELT_##T, T)                \
    DEF_IDXOP(GrB_VALUELE_##T, T)                \
    DEF_BINOP_X(GrB_DIV_##T, OP_UNSUPPORTED, T)  \
    DEF_BINOP_X(GxB_RDIV_##T, OP_UNSUPPORTED, T) \
    DEF_BINOP_X(GxB_RMINUS_##T, OP_RMINUS, T)    \
    DEF_BINOP_X(GxB_ISEQ_##T, OP_UNSUPPORTED, T) \
    DEF_BINOP_X(GxB_ISNE_##T, OP_UNSUPPORTED, T) \
    DEF_BINOP_X(GxB_ISGT_##T, OP_UNSUPPORTED, T) \
    DEF_BINOP_X(GxB_ISLT_##T, OP_UNSUPPORTED, T) \
    DEF_BINOP_X(GxB_ISGE_##T, OP_UNSUPPORTED, T) \
    DEF_BINOP_X(GxB_ISLE_##T, OP_UNSUPPORTED, T) \
    DEF_BINOP_X(GxB_POW_##T, OP_UNSUPPORTED, T)
GRB_FOR_EACH_TNAME(DEF_SURFACE_OPS)
#undef DEF_SURFACE_OPS
DEF_UNOP(GrB_LNOT, BOOL)
DEF_UNOP(GrB_BNOT_INT8, INT8)
DEF_UNOP(GrB_BNOT_INT16, INT16)
DEF_UNOP(GrB_BNOT_INT32, INT32)
DEF_UNOP(GrB_BNOT_INT64, INT64)
DEF_UNOP(GrB_BNOT_UINT8, UINT8)
DEF_UNOP(GrB_BNOT_UINT16, UINT16)
DEF_UNOP(GrB_BNOT_UINT32, UINT32)
DEF_UNOP(GrB_BNOT_UINT64, UINT64)
DEF_IDXOP(GrB_ROWINDEX_INT32, INT32)
DEF_IDXOP(GrB_ROWINDEX_INT64, INT64)
DEF_IDXOP(GrB_COLINDEX_INT32, INT32)
DEF_IDXOP(GrB_COLINDEX_INT64, INT64)
DEF_IDXOP(GrB_DIAGINDEX_INT32, INT32)
DEF_IDXOP(GrB_DIAGINDEX_INT64, INT64)
DEF_IDXOP(GrB_TRIL, INT64)
DEF_IDXOP(GrB_TRIU, INT64)
DEF_IDXOP(GrB_DIAG, INT64)
DEF_IDXOP(GrB_OFFDIAG, INT64)
DEF_IDXOP(GrB_COLLE, INT64)
DEF_IDXOP(GrB_COLGT, INT64)
DEF_IDXOP(GrB_ROWLE, INT64)
DEF_IDXOP(GrB_ROWGT, INT64)

// ---- GrB_Scalar -----------------------------------------------------------------------------------------------------
static void check_scalar(const GB_Scalar_opaque *s, const char *what)
{
    if (!s) fail(GrB_NULL_POINTER, std::string(what) + " is NULL");
    if (s->magic != MAGIC_SCALAR) fail(GrB_INVALID_OBJECT, std::string(what) + " is not a valid GrB_Scalar");
}
static std::string *errp(GB_Scalar_opaque *s) { return (s && s->magic == MAGIC_SCALAR) ? &s->err : nullptr; }

extern "C" GrB_Info GrB_Scalar_new(GrB_Scalar *s, GrB_Type type)
{
    if (!s || !type) return GrB_NULL_POINTER;
    auto *o = new GB_Scalar_opaque();
    o->magic = MAGIC_SCALAR;
    o->type = type;
    o->has = false;
    memset(o->value, 0, sizeof(o->value));
    *s = o;
    return GrB_SUCCESS;
}
extern "C" GrB_Info GrB_Scalar_dup(GrB_Scalar *s, const GrB_Scalar t)
{
    GRB_TRY
    if (!s) fail(GrB_NULL_POINTER, "output pointer is NULL");
    check_scalar(t, "t");
    auto *o = new GB_Scalar_opaque(*t);
    o->err.clear();
    *s = o;
    GRB_CATCH(errp(t))
}
extern "C" GrB_Info GrB_Scalar_free(GrB_Scalar *s)
{
    if (!s) return GrB_NULL_POINTER;
    if (*s && (*s)->magic == MAGIC_SCALAR) {
        (*s)->magic = MAGIC_FREED;
        delete *s;
    }
    *s = nullptr;
    return GrB_SUCCESS;
}
extern "C" GrB_Info GrB_Scalar_clear(GrB_Scalar s)
{
    GRB_TRY
    check_scalar(s, "s");
    s->has = false;
    GRB_CATCH(errp(s))
}
extern "C" GrB_Info GrB_Scalar_nvals(GrB_Index *nvals, const GrB_Scalar s)
{
    GRB_TRY
    if (!nvals) fail(GrB_NULL_POINTER, "nvals is NULL");
    check_scalar(s, "s");
    *nvals = s->has ? 1 : 0;
    GRB_CATCH(errp(s))
}
extern "C" GrB_Info GrB_Scalar_wait(GrB_Scalar s, GrB_WaitMode)
{
    GRB_TRY
    check_scalar(s, "s");
    GRB_CATCH(errp(s))
}
extern "C" GrB_Info GrB_Scalar_error(const char **error, const GrB_Scalar s)
{
    if (!error) return GrB_NULL_POINTER;
    *error = (s && s->magic == MAGIC_SCALAR) ? s->err.c_str() : "";
    return GrB_SUCCESS;
}
template <typename X> static void scalar_set(GB_Scalar_opaque *s, X x)
{
    GRB_DISPATCH_TYPE(s->type->code, TS, {
        const TS v = cast_value<TS, X>(x);
        memcpy(s->value, &v, sizeof(TS));
    })
    s->has = true;
}
template <typename X> static GrB_Info scalar_get(X *x, const GB_Scalar_opaque *s)
{
    if (!x) fail(GrB_NULL_POINTER, "output pointer is NULL");
    if (!s->has) return GrB_NO_VALUE;
    GRB_DISPATCH_TYPE(s->type->code, TS, {
        TS v;
        memcpy(&v, s->value, sizeof(TS));
        *x = cast_value<X, TS>(v);
    })
    return GrB_SUCCESS;
}

// a scalar handed to a vector entry point: through the typed form of the same entry point
extern "C" GrB_Info GrB_Vector_setElement_Scalar(GrB_Vector w, const GrB_Scalar s, GrB_Index i)
{
    GRB_TRY
    check_vector(w, "w");
    check_scalar(s, "s");
    if (!s->has) return GrB_Vector_removeElement(w, i);
    GrB_Info rc;
    const int tc = s->type->code;
    if (tc == TC_INT64 || tc == TC_UINT64) {
        int64_t q = 0;
        memcpy(&q, s->value, 8);
        rc = tc == TC_INT64 ? GrB_Vector_setElement_INT64(w, q, i) : GrB_Vector_setElement_UINT64(w, (uint64_t)q, i);
    } else {
        double d = 0;
        scalar_get<double>(&d, s);
        rc = GrB_Vector_setElement_FP64(w, d, i);
    }
    return rc;
    GRB_CATCH(errp(w))
}
extern "C" GrB_Info GrB_Vector_extractElement_Scalar(GrB_Scalar s, const GrB_Vector u, GrB_Index i)
{
    GRB_TRY
    check_scalar(s, "s");
    check_vector(u, "u");
    // (the typed entry points cast: read through the widest exact type of u's kind, store in the scalar's own type)
    GrB_Info rc;
    const int tc = u->type->code;
    if (tc == TC_FP32 || tc == TC_FP64) {
        double v = 0;
        rc = GrB_Vector_extractElement_FP64(&v, u, i);
        if (rc == GrB_SUCCESS) scalar_set<double>(s, v);
    } else if (tc == TC_UINT64) {
        uint64_t v = 0;
        rc = GrB_Vector_extractElement_UINT64(&v, u, i);
        if (rc == GrB_SUCCESS) scalar_set<uint64_t>(s, v);
    } else {
        int64_t v = 0;
        rc = GrB_Vector_extractElement_INT64(&v, u, i);
        if (rc == GrB_SUCCESS) scalar_set<int64_t>(s, v);
    }
    if (rc == GrB_NO_VALUE) {
        s->has = false;
        rc = GrB_SUCCESS;  // (C API 2.0: an absent element empties the scalar)
    }
    return rc;
    GRB_CATCH(errp(s))
}
extern "C" GrB_Info GrB_Vector_assign_Scalar(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Scalar s,
                                             const GrB_Index *I, GrB_Index ni, const GrB_Descriptor desc)
{
    GRB_TRY
    check_vector(w, "w");
    check_scalar(s, "s");
    if (!s->has) fail(GrB_NOT_IMPLEMENTED, "assign of an EMPTY scalar (deleting the selected entries) is outside this library's path");
    double d = 0;
    int64_t q = 0;
    scalar_get<double>(&d, s);
    scalar_get<int64_t>(&q, s);
    const int tc = s->type->code;
    if (tc == TC_INT64) return GrB_Vector_assign_INT64(w, mask, accum, q, I, ni, desc);
    if (tc == TC_UINT64) {
        uint64_t uq = 0;
        scalar_get<uint64_t>(&uq, s);
        return GrB_Vector_assign_UINT64(w, mask, accum, uq, I, ni, desc);
    }
    return GrB_Vector_assign_FP64(w, mask, accum, d, I, ni, desc);
    GRB_CATCH(errp(w))
}

// ---- entry points of the operations outside the path: GrB_NOT_IMPLEMENTED, with a message on the output object -------------
static GrB_Info not_impl(GB_Matrix_opaque *C, const char *fn)
{
    if (C && C->magic == MAGIC_MATRIX) C->err = std::string(fn) + ": this operation is outside libgrb_mi355x's accelerated path";
    return GrB_NOT_IMPLEMENTED;
}
static GrB_Info not_impl(GB_Vector_opaque *w, const char *fn)
{
    if (w && w->magic == MAGIC_VECTOR) w->err = std::string(fn) + ": this operation is outside libgrb_mi355x's accelerated path";
    return GrB_NOT_IMPLEMENTED;
}
#define NI_V(FN, ...) extern "C" GrB_Info FN(GrB_Vector w, __VA_ARGS__) { return not_impl(w, #FN); }
#define NI_M(FN, ...) extern "C" GrB_Info FN(GrB_Matrix C, __VA_ARGS__) { return not_impl(C, #FN); }

NI_V(GrB_Vector_apply, const GrB_Vector, const GrB_BinaryOp, const GrB_UnaryOp, const GrB_Vector, const GrB_Descriptor)
NI_M(GrB_Matrix_apply, const GrB_Matrix, const GrB_BinaryOp, const GrB_UnaryOp, const GrB_Matrix, const GrB_Descriptor)
NI_V(GrB_Vector_eWiseAdd_Semiring, const GrB_Vector, const GrB_BinaryOp, const GrB_Semiring, const GrB_Vector, const GrB_Vector, const GrB_Descriptor)
NI_V(GrB_Vector_eWiseMult_Semiring, const GrB_Vector, const GrB_BinaryOp, const GrB_Semiring, const GrB_Vector, const GrB_Vector, const GrB_Descriptor)
#define NI_MAT_BINARY(FN, HANDLE) NI_M(FN, const GrB_Matrix, const GrB_BinaryOp, const HANDLE, const GrB_Matrix, const GrB_Matrix, const GrB_Descriptor)
NI_MAT_BINARY(GrB_Matrix_eWiseAdd_BinaryOp, GrB_BinaryOp)
NI_MAT_BINARY(GrB_Matrix_eWiseAdd_Monoid, GrB_Monoid)
NI_MAT_BINARY(GrB_Matrix_eWiseAdd_Semiring, GrB_Semiring)
NI_MAT_BINARY(GrB_Matrix_eWiseMult_BinaryOp, GrB_BinaryOp)
NI_MAT_BINARY(GrB_Matrix_eWiseMult_Monoid, GrB_Monoid)
NI_MAT_BINARY(GrB_Matrix_eWiseMult_Semiring, GrB_Semiring)
NI_MAT_BINARY(GrB_Matrix_kronecker_BinaryOp, GrB_BinaryOp)
NI_MAT_BINARY(GrB_Matrix_kronecker_Monoid, GrB_Monoid)
NI_MAT_BINARY(GrB_Matrix_kronecker_Semiring, GrB_Semiring)
NI_M(GrB_Matrix_assign, const GrB_Matrix, const GrB_BinaryOp, const GrB_Matrix, const GrB_Index *, GrB_Index, const GrB_Index *, GrB_Index, const GrB_Descriptor)
NI_M(GrB_Row_assign, const GrB_Vector, const GrB_BinaryOp, const GrB_Vector, GrB_Index, const GrB_Index *, GrB_Index, const GrB_Descriptor)
NI_M(GrB_Col_assign, const GrB_Vector, const GrB_BinaryOp, const GrB_Vector, const GrB_Index *, GrB_Index, GrB_Index, const GrB_Descriptor)
NI_M(GrB_Matrix_extract, const GrB_Matrix, const GrB_BinaryOp, const GrB_Matrix, const GrB_Index *, GrB_Index, const GrB_Index *, GrB_Index, const GrB_Descriptor)
NI_V(GrB_Col_extract, const GrB_Vector, const GrB_BinaryOp, const GrB_Matrix, const GrB_Index *, GrB_Index, GrB_Index, const GrB_Descriptor)
NI_V(GrB_Matrix_reduce_BinaryOp, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, const GrB_Descriptor)
NI_M(GrB_Matrix_removeElement, GrB_Index, GrB_Index)
NI_M(GrB_Matrix_assign_Scalar, const GrB_Matrix, const GrB_BinaryOp, const GrB_Scalar, const GrB_Index *, GrB_Index, const GrB_Index *, GrB_Index, const GrB_Descriptor)
NI_M(GrB_Matrix_setElement_Scalar, const GrB_Scalar, GrB_Index, GrB_Index)
NI_V(GrB_Vector_select_Scalar, const GrB_Vector, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Vector, const GrB_Scalar, const GrB_Descriptor)
NI_M(GrB_Matrix_select_Scalar, const GrB_Matrix, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Matrix, const GrB_Scalar, const GrB_Descriptor)
NI_V(GrB_Vector_apply_BinaryOp1st_Scalar, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Scalar, const GrB_Vector, const GrB_Descriptor)
NI_V(GrB_Vector_apply_BinaryOp2nd_Scalar, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Vector, const GrB_Scalar, const GrB_Descriptor)
NI_V(GrB_Vector_apply_IndexOp_Scalar, const GrB_Vector, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Vector, const GrB_Scalar, const GrB_Descriptor)
NI_M(GrB_Matrix_apply_BinaryOp1st_Scalar, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Scalar, const GrB_Matrix, const GrB_Descriptor)
NI_M(GrB_Matrix_apply_BinaryOp2nd_Scalar, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, const GrB_Scalar, const GrB_Descriptor)
NI_M(GrB_Matrix_apply_IndexOp_Scalar, const GrB_Matrix, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Matrix, const GrB_Scalar, const GrB_Descriptor)

extern "C" GrB_Info GrB_Matrix_reduce_Monoid_Scalar(GrB_Scalar, const GrB_BinaryOp, const GrB_Monoid, const GrB_Matrix, const GrB_Descriptor) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_Vector_reduce_Monoid_Scalar(GrB_Scalar, const GrB_BinaryOp, const GrB_Monoid, const GrB_Vector, const GrB_Descriptor) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_Matrix_extractElement_Scalar(GrB_Scalar, const GrB_Matrix, GrB_Index, GrB_Index) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_Matrix_diag(GrB_Matrix *, const GrB_Vector, int64_t) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_Type_new(GrB_Type *, size_t) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_UnaryOp_new(GrB_UnaryOp *, void *, GrB_Type, GrB_Type) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_BinaryOp_new(GrB_BinaryOp *, void *, GrB_Type, GrB_Type, GrB_Type) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_IndexUnaryOp_new(GrB_IndexUnaryOp *, void *, GrB_Type, GrB_Type, GrB_Type) { return GrB_NOT_IMPLEMENTED; }
extern "C" GrB_Info GrB_Semiring_new(GrB_Semiring *, GrB_Monoid, GrB_BinaryOp) { return GrB_NOT_IMPLEMENTED; }

#define DEF_SURFACE_TYPED(NAME, ctype)                                                                                               \
    extern "C" GrB_Info GrB_Scalar_setElement_##NAME(GrB_Scalar s, ctype x)                                                          \
    {                                                                                                                                \
        GRB_TRY                                                                                                                      \
        check_scalar(s, "s");                                                                                                        \
        scalar_set<ctype>(s, x);                                                                                                     \
        GRB_CATCH(errp(s))                                                                                                           \
    }                                                                                                                                \
    extern "C" GrB_Info GrB_Scalar_extractElement_##NAME(ctype *x, const GrB_Scalar s)                                               \
    {                                                                                                                                \
        GRB_TRY                                                                                                                      \
        check_scalar(s, "s");                                                                                                        \
        return scalar_get<ctype>(x, s);                                                                                              \
        GRB_CATCH(errp(s))                                                                                                           \
    }                                                                                                                                \
    extern "C" GrB_Info GrB_Monoid_new_##NAME(GrB_Monoid *, GrB_BinaryOp, ctype) { return GrB_NOT_IMPLEMENTED; }                     \
    NI_M(GrB_Matrix_setElement_##NAME, ctype, GrB_Index, GrB_Index)                                                                  \
    extern "C" GrB_Info GrB_Matrix_extractElement_##NAME(ctype *, const GrB_Matrix, GrB_Index, GrB_Index) { return GrB_NOT_IMPLEMENTED; } \
    extern "C" GrB_Info GrB_Matrix_reduce_##NAME(ctype *, const GrB_BinaryOp, const GrB_Monoid, const GrB_Matrix, const GrB_Descriptor) { return GrB_NOT_IMPLEMENTED; } \
    NI_M(GrB_Matrix_assign_##NAME, const GrB_Matrix, const GrB_BinaryOp, ctype, const GrB_Index *, GrB_Index, const GrB_Index *, GrB_Index, const GrB_Descriptor) \
    NI_V(GrB_Vector_select_##NAME, const GrB_Vector, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Vector, ctype, const GrB_Descriptor) \
    NI_M(GrB_Matrix_select_##NAME, const GrB_Matrix, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Matrix, ctype, const GrB_Descriptor) \
    NI_V(GrB_Vector_apply_BinaryOp1st_##NAME, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, ctype, const GrB_Vector, const GrB_Descriptor) \
    NI_V(GrB_Vector_apply_BinaryOp2nd_##NAME, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Vector, ctype, const GrB_Descriptor) \
    NI_V(GrB_Vector_apply_IndexOp_##NAME, const GrB_Vector, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Vector, ctype, const GrB_Descriptor) \
    NI_M(GrB_Matrix_apply_BinaryOp1st_##NAME, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, ctype, const GrB_Matrix, const GrB_Descriptor) \
    NI_M(GrB_Matrix_apply_BinaryOp2nd_##NAME, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, ctype, const GrB_Descriptor) \
    NI_M(GrB_Matrix_apply_IndexOp_##NAME, const GrB_Matrix, const GrB_BinaryOp, const GrB_IndexUnaryOp, const GrB_Matrix, ctype, const GrB_Descriptor)
GRB_FOR_EACH_TYPE(DEF_SURFACE_TYPED)
#undef DEF_SURFACE_TYPED
