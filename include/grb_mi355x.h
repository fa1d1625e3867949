/* grb_mi355x.h -- C-ABI of libgrb_mi355x.so: the GraphBLAS C API 2.0 entry points that
 * python-graphblas dispatches to on its mxm / mxv / vxm hot path, implemented with hand-written
 * gfx950 (MI355X / CDNA4) HIP kernels.  Plain C: opaque handles, plain pointers and sizes.
 *
 * What each group replaces in the reference (paths relative to /root/reference):
 *   - every function below is reached through graphblas/core/base.py:23-54 `call(cfunc_name, args)`
 *     (symbol lookup graphblas/core/utils.py:11-23) on the cffi `lib` bound at
 *     graphblas/__init__.py:143-199; return codes are mapped by graphblas/exceptions.py:123-189.
 *   - GrB_mxv  <- graphblas/core/matrix.py:2252-2259  ("GrB_mxv", args marshalled base.py:496-501)
 *   - GrB_mxm  <- graphblas/core/matrix.py:2318-2328  ("GrB_mxm")
 *   - GrB_vxm  <- graphblas/core/vector.py:1367-1375  ("GrB_vxm")
 *   - GrB_Matrix_new/free/nvals/... <- graphblas/core/matrix.py:190-225, 493; Vector twins core/vector.py:159-191
 *   - GrB_Matrix_build_T <- core/matrix.py:627-681; GrB_Matrix_import_T/export_T <- :992-1068, :1601-1645
 *   - GrB_Matrix_extractTuples_T <- core/matrix.py:525-594
 *   - GrB_DESC_* globals <- core/descriptor.py:51-84; GrB_<TYPE> globals <- core/dtypes.py:329-420
 *   - semiring / monoid / binaryop globals are discovered by regex over dir(lib):
 *     core/operator/base.py:803-893, core/operator/semiring.py:185-219
 *   - GrB_*_error <- exceptions.py:171-189
 *
 * GrX_* functions are this library's own extensions (device-resident import/export, stream and
 * timing hooks); the reference's own zero-copy import (core/ss/matrix.py:1279-1349) is offered under its GxB names below, over host arrays.
 */
#ifndef GRB_MI355X_H
#define GRB_MI355X_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRB_VERSION 2
#define GRB_SUBVERSION 0

typedef uint64_t GrB_Index;
#define GrB_INDEX_MAX ((GrB_Index)(1ULL << 60) - 1)

/* GraphBLAS C API 2.0 return codes (spec values) */
typedef enum {
    GrB_SUCCESS = 0,
    GrB_NO_VALUE = 1,
    GrB_UNINITIALIZED_OBJECT = -1,
    GrB_NULL_POINTER = -2,
    GrB_INVALID_VALUE = -3,
    GrB_INVALID_INDEX = -4,
    GrB_DOMAIN_MISMATCH = -5,
    GrB_DIMENSION_MISMATCH = -6,
    GrB_OUTPUT_NOT_EMPTY = -7,
    GrB_NOT_IMPLEMENTED = -8,
    GrB_PANIC = -101,
    GrB_OUT_OF_MEMORY = -102,
    GrB_INSUFFICIENT_SPACE = -103,
    GrB_INVALID_OBJECT = -104,
    GrB_INDEX_OUT_OF_BOUNDS = -105,
    GrB_EMPTY_OBJECT = -106
} GrB_Info;

typedef enum { GrB_NONBLOCKING = 0, GrB_BLOCKING = 1 } GrB_Mode;
typedef enum { GrB_COMPLETE = 0, GrB_MATERIALIZE = 1 } GrB_WaitMode;
typedef enum { GrB_OUTP = 0, GrB_MASK = 1, GrB_INP0 = 2, GrB_INP1 = 3 } GrB_Desc_Field;
typedef enum { GrB_DEFAULT = 0, GrB_REPLACE = 1, GrB_COMP = 2, GrB_TRAN = 3, GrB_STRUCTURE = 4 } GrB_Desc_Value;
typedef enum { GrB_CSR_FORMAT = 0, GrB_CSC_FORMAT = 1, GrB_COO_FORMAT = 2 } GrB_Format;

typedef struct GB_Type_opaque *GrB_Type;
typedef struct GB_BinaryOp_opaque *GrB_BinaryOp;
typedef struct GB_Monoid_opaque *GrB_Monoid;
typedef struct GB_Semiring_opaque *GrB_Semiring;
typedef struct GB_Descriptor_opaque *GrB_Descriptor;
typedef struct GB_Vector_opaque *GrB_Vector;
typedef struct GB_Matrix_opaque *GrB_Matrix;
typedef struct GB_UnaryOp_opaque *GrB_UnaryOp;           /* handles only: see "import-time surface" below */
typedef struct GB_IndexUnaryOp_opaque *GrB_IndexUnaryOp;
typedef struct GB_Scalar_opaque *GrB_Scalar;

/* GrB_ALL: the reference only compares this pointer (core/expr.py:14) */
extern const uint64_t *GrB_ALL;

/* ---- context ------------------------------------------------------------------------------ */
GrB_Info GrB_init(GrB_Mode mode);      /* requires a gfx950 device: GrB_PANIC otherwise (no CPU fallback) */
GrB_Info GrB_finalize(void);
GrB_Info GrB_getVersion(unsigned int *version, unsigned int *subversion);

/* ---- the hot path ------------------------------------------------------------------------- */
GrB_Info GrB_mxm(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_mxv(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Matrix A, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GrB_vxm(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Vector u, const GrB_Matrix A, const GrB_Descriptor desc);

/* ---- Matrix ------------------------------------------------------------------------------- */
GrB_Info GrB_Matrix_new(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols);
GrB_Info GrB_Matrix_dup(GrB_Matrix *C, const GrB_Matrix A);
GrB_Info GrB_Matrix_free(GrB_Matrix *A);
GrB_Info GrB_Matrix_clear(GrB_Matrix A);
GrB_Info GrB_Matrix_resize(GrB_Matrix A, GrB_Index nrows, GrB_Index ncols); /* core/matrix.py:512-523 */
/* row-wise (desc T0: column-wise) reduction with a monoid: core/matrix.py:2636-2710; the pull SpMV over (monoid, FIRST) */
GrB_Info GrB_Matrix_reduce_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Monoid monoid,
                                  const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_nrows(GrB_Index *nrows, const GrB_Matrix A);
GrB_Info GrB_Matrix_ncols(GrB_Index *ncols, const GrB_Matrix A);
GrB_Info GrB_Matrix_nvals(GrB_Index *nvals, const GrB_Matrix A);
GrB_Info GrB_Matrix_wait(GrB_Matrix A, GrB_WaitMode mode);
GrB_Info GrB_Matrix_error(const char **error, const GrB_Matrix A);
GrB_Info GrB_Matrix_exportSize(GrB_Index *Ap_len, GrB_Index *Ai_len, GrB_Index *Ax_len, GrB_Format format,
                               const GrB_Matrix A);
GrB_Info GrB_Matrix_exportHint(GrB_Format *format, const GrB_Matrix A);
GrB_Info GrB_transpose(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A,
                       const GrB_Descriptor desc); /* C<Mask, replace> = accum(C, A') (or A with T0), through the matrix write rule of GrB_mxm */

/* ---- Vector ------------------------------------------------------------------------------- */
GrB_Info GrB_Vector_new(GrB_Vector *v, GrB_Type type, GrB_Index n);
GrB_Info GrB_Vector_dup(GrB_Vector *w, const GrB_Vector u);
GrB_Info GrB_Vector_free(GrB_Vector *v);
GrB_Info GrB_Vector_clear(GrB_Vector v);
GrB_Info GrB_Vector_resize(GrB_Vector v, GrB_Index size);                    /* core/vector.py:455-463 */
/* w<mask>(I) = accum(w(I), u) / w<mask> = accum(w, u(I)): index lists of any length, or GrB_ALL (reference core/vector.py:1906-2035,
 * core/expr.py:404-560; C API 2.0 GrB_assign -- the mask has w's size -- and GrB_extract).  The typed scalar form
 * GrB_Vector_assign_<T> below takes index lists too. */
GrB_Info GrB_Vector_assign(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index *indices,
                           GrB_Index nindices, const GrB_Descriptor desc);
GrB_Info GrB_Vector_extract(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index *indices,
                            GrB_Index nindices, const GrB_Descriptor desc);
GrB_Info GrB_Vector_removeElement(GrB_Vector w, GrB_Index i);                /* core/vector.py:1916-1930 */
/* element-wise union / intersection (core/vector.py:960-1150): op on the entries both have; eWiseAdd passes single entries through */
GrB_Info GrB_Vector_eWiseAdd_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                      const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseAdd_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Monoid op,
                                    const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseMult_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                       const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseMult_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Monoid op,
                                     const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_size(GrB_Index *n, const GrB_Vector v);
GrB_Info GrB_Vector_nvals(GrB_Index *nvals, const GrB_Vector v);
GrB_Info GrB_Vector_wait(GrB_Vector v, GrB_WaitMode mode);
GrB_Info GrB_Vector_error(const char **error, const GrB_Vector v);

/* ---- typed ingress / egress (C type per GraphBLAS type; GrB_BOOL is `bool`, 1 byte) --------- */
#define GRB_FOR_EACH_TYPE(X) \
    X(BOOL, bool)            \
    X(INT8, int8_t)          \
    X(INT16, int16_t)        \
    X(INT32, int32_t)        \
    X(INT64, int64_t)        \
    X(UINT8, uint8_t)        \
    X(UINT16, uint16_t)      \
    X(UINT32, uint32_t)      \
    X(UINT64, uint64_t)      \
    X(FP32, float)           \
    X(FP64, double)

#define GRB_DECL_TYPED(NAME, ctype)                                                                                    \
    extern GrB_Type GrB_##NAME;                                                                                        \
    GrB_Info GrB_Matrix_build_##NAME(GrB_Matrix C, const GrB_Index *I, const GrB_Index *J, const ctype *X,            \
                                     GrB_Index nvals, const GrB_BinaryOp dup);                                         \
    GrB_Info GrB_Matrix_extractTuples_##NAME(GrB_Index *I, GrB_Index *J, ctype *X, GrB_Index *nvals,                   \
                                             const GrB_Matrix A);                                                      \
    GrB_Info GrB_Matrix_import_##NAME(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols,                  \
                                      const GrB_Index *Ap, const GrB_Index *Ai, const ctype *Ax, GrB_Index Ap_len,     \
                                      GrB_Index Ai_len, GrB_Index Ax_len, GrB_Format format);                          \
    GrB_Info GrB_Matrix_export_##NAME(GrB_Index *Ap, GrB_Index *Ai, ctype *Ax, GrB_Index *Ap_len, GrB_Index *Ai_len,   \
                                      GrB_Index *Ax_len, GrB_Format format, const GrB_Matrix A);                       \
    GrB_Info GrB_Vector_build_##NAME(GrB_Vector w, const GrB_Index *I, const ctype *X, GrB_Index nvals,                \
                                     const GrB_BinaryOp dup);                                                          \
    GrB_Info GrB_Vector_extractTuples_##NAME(GrB_Index *I, ctype *X, GrB_Index *nvals, const GrB_Vector v);               \
    /* the vector operations around the path (BFS / SSSP loops: core/vector.py:1635-1684, 1840-1930, 1979-2035) */     \
    GrB_Info GrB_Vector_setElement_##NAME(GrB_Vector w, ctype x, GrB_Index i);                                         \
    GrB_Info GrB_Vector_extractElement_##NAME(ctype *x, const GrB_Vector u, GrB_Index i);                              \
    GrB_Info GrB_Vector_assign_##NAME(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, ctype x,          \
                                      const GrB_Index *indices, GrB_Index nindices, const GrB_Descriptor desc);       \
    GrB_Info GrB_Vector_reduce_##NAME(ctype *val, const GrB_BinaryOp accum, const GrB_Monoid monoid,                   \
                                      const GrB_Vector u, const GrB_Descriptor desc);
GRB_FOR_EACH_TYPE(GRB_DECL_TYPED)
#undef GRB_DECL_TYPED

/* ---- descriptors (core/descriptor.py:51-84) ------------------------------------------------- */
GrB_Info GrB_Descriptor_new(GrB_Descriptor *desc);
GrB_Info GrB_Descriptor_set(GrB_Descriptor desc, GrB_Desc_Field field, GrB_Desc_Value value);
GrB_Info GrB_Descriptor_free(GrB_Descriptor *desc);
extern GrB_Descriptor
    GrB_DESC_T1, GrB_DESC_T0, GrB_DESC_T0T1, GrB_DESC_C, GrB_DESC_CT1, GrB_DESC_CT0, GrB_DESC_CT0T1,
    GrB_DESC_S, GrB_DESC_ST1, GrB_DESC_ST0, GrB_DESC_ST0T1, GrB_DESC_SC, GrB_DESC_SCT1, GrB_DESC_SCT0,
    GrB_DESC_SCT0T1, GrB_DESC_R, GrB_DESC_RT1, GrB_DESC_RT0, GrB_DESC_RT0T1, GrB_DESC_RC, GrB_DESC_RCT1,
    GrB_DESC_RCT0, GrB_DESC_RCT0T1, GrB_DESC_RS, GrB_DESC_RST1, GrB_DESC_RST0, GrB_DESC_RST0T1, GrB_DESC_RSC,
    GrB_DESC_RSCT1, GrB_DESC_RSCT0, GrB_DESC_RSCT0T1;

/* ---- builtin operators ------------------------------------------------------------------------
 * Names follow the GraphBLAS C API / SuiteSparse GxB conventions matched by the reference's regexes
 * (core/operator/semiring.py:185-219, binary.py, monoid.py). */
#define GRB_FOR_EACH_NUMERIC(X) X(INT8) X(INT16) X(INT32) X(INT64) X(UINT8) X(UINT16) X(UINT32) X(UINT64) X(FP32) X(FP64)
#define GRB_FOR_EACH_TNAME(X) X(BOOL) GRB_FOR_EACH_NUMERIC(X)

#define GRB_DECL_BINOPS(T)                                                                                             \
    extern GrB_BinaryOp GrB_FIRST_##T, GrB_SECOND_##T, GrB_ONEB_##T, GxB_PAIR_##T, GrB_PLUS_##T, GrB_MINUS_##T,        \
        GrB_TIMES_##T, GrB_MIN_##T, GrB_MAX_##T, GxB_ANY_##T, GxB_LOR_##T, GxB_LAND_##T, GxB_LXOR_##T;                 \
    /* comparisons (T x T -> BOOL): operands of GrB_Vector_eWiseAdd / eWiseMult only (isequal, core/vector.py:340-379) */  \
    extern GrB_BinaryOp GrB_EQ_##T, GrB_NE_##T, GrB_GT_##T, GrB_LT_##T, GrB_GE_##T, GrB_LE_##T;                        \
    extern GrB_Monoid GxB_ANY_##T##_MONOID;                                                                            \
    extern GrB_Semiring GxB_ANY_PAIR_##T, GxB_ANY_FIRST_##T, GxB_ANY_SECOND_##T;
GRB_FOR_EACH_TNAME(GRB_DECL_BINOPS)
#undef GRB_DECL_BINOPS
extern GrB_BinaryOp GrB_LOR, GrB_LAND, GrB_LXOR, GrB_LXNOR;
extern GrB_Monoid GrB_LOR_MONOID_BOOL, GrB_LAND_MONOID_BOOL, GrB_LXOR_MONOID_BOOL, GrB_LXNOR_MONOID_BOOL;

#define GRB_DECL_NUMERIC_OPS(T)                                                                                        \
    extern GrB_Monoid GrB_PLUS_MONOID_##T, GrB_TIMES_MONOID_##T, GrB_MIN_MONOID_##T, GrB_MAX_MONOID_##T;               \
    extern GrB_Semiring GrB_PLUS_TIMES_SEMIRING_##T, GrB_PLUS_MIN_SEMIRING_##T, GrB_MIN_PLUS_SEMIRING_##T,             \
        GrB_MIN_TIMES_SEMIRING_##T, GrB_MIN_FIRST_SEMIRING_##T, GrB_MIN_SECOND_SEMIRING_##T,                           \
        GrB_MIN_MAX_SEMIRING_##T, GrB_MAX_PLUS_SEMIRING_##T, GrB_MAX_TIMES_SEMIRING_##T,                               \
        GrB_MAX_FIRST_SEMIRING_##T, GrB_MAX_SECOND_SEMIRING_##T, GrB_MAX_MIN_SEMIRING_##T;                             \
    extern GrB_Semiring GxB_PLUS_PLUS_##T, GxB_PLUS_PAIR_##T, GxB_PLUS_FIRST_##T, GxB_PLUS_SECOND_##T,                 \
        GxB_PLUS_MAX_##T, GxB_MIN_MIN_##T, GxB_MAX_MAX_##T, GxB_MIN_PAIR_##T, GxB_MAX_PAIR_##T, GxB_TIMES_TIMES_##T,   \
        GxB_TIMES_PLUS_##T;
GRB_FOR_EACH_NUMERIC(GRB_DECL_NUMERIC_OPS)
#undef GRB_DECL_NUMERIC_OPS

extern GrB_Semiring GrB_LOR_LAND_SEMIRING_BOOL, GrB_LAND_LOR_SEMIRING_BOOL, GrB_LXOR_LAND_SEMIRING_BOOL,
    GrB_LXNOR_LOR_SEMIRING_BOOL;
extern GrB_Semiring GxB_LOR_LOR_BOOL, GxB_LAND_LAND_BOOL, GxB_LOR_FIRST_BOOL, GxB_LOR_SECOND_BOOL, GxB_LOR_PAIR_BOOL,
    GxB_LAND_FIRST_BOOL, GxB_LAND_SECOND_BOOL, GxB_LOR_LXOR_BOOL, GxB_LAND_LXOR_BOOL, GxB_LXOR_LOR_BOOL,
    GxB_LXOR_LXOR_BOOL, GxB_LXOR_FIRST_BOOL, GxB_LXOR_SECOND_BOOL, GxB_LXOR_PAIR_BOOL;

/* =================================================================================================
 * GrX_* : extensions of this library (no reference counterpart).
 * ================================================================================================= */

/* Adopt (copy=0: take a reference; the caller keeps the allocation alive and unmodified until the
 * matrix is freed) or copy (copy=1) a CSR that already lives in HBM.  d_Ap: int64[nrows+1],
 * d_Aj: int32[nvals] sorted within each row, d_Ax: nvals values of `type` (or ONE value when iso!=0). */
GrB_Info GrX_Matrix_import_CSR_device(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols,
                                      const int64_t *d_Ap, const int32_t *d_Aj, const void *d_Ax, GrB_Index nvals,
                                      int iso, int copy);
/* ---- the reference's zero-copy ingress, under the names it binds (SuiteSparse GxB layer) -------------------------------------
 * graphblas/core/ss/matrix.py:1279-1349 (`Matrix.ss.import_csr` / `ss.pack_csr` -> `GxB_Matrix_{import,pack}_CSR`, arguments
 * marshalled at :1316-1333) and graphblas/__init__.py:170-173 (GxB_init through suitesparse_graphblas.initialize).
 * Ap / Aj / Ax are HOST arrays from the allocator given to GxB_init (default malloc); sizes in BYTES; iso: *Ax holds one value;
 * jumbled: columns inside a row unsorted (sorted here either way).  On success the library OWNS the arrays: they are copied to HBM
 * once (the matrices of this library live there), released with the registered deallocator, and *Ap = *Aj = *Ax = NULL -- the
 * protocol the reference's wrapper relies on (it unclaims its numpy buffers after the call).  On failure the arrays stay the
 * caller's.  pack replaces the content of an existing matrix of the same type and shape.  The form without any copy, for a CSR
 * that is already in HBM, is GrX_Matrix_import_CSR_device above. */
GrB_Info GxB_init(GrB_Mode mode, void *(*user_malloc)(size_t), void *(*user_calloc)(size_t, size_t),
                  void *(*user_realloc)(void *, size_t), void (*user_free)(void *));
GrB_Info GxB_Matrix_import_CSR(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols, GrB_Index **Ap, GrB_Index **Aj,
                               void **Ax, GrB_Index Ap_size, GrB_Index Aj_size, GrB_Index Ax_size, bool iso, bool jumbled,
                               const GrB_Descriptor desc);
GrB_Info GxB_Matrix_pack_CSR(GrB_Matrix A, GrB_Index **Ap, GrB_Index **Aj, void **Ax, GrB_Index Ap_size, GrB_Index Aj_size,
                             GrB_Index Ax_size, bool iso, bool jumbled, const GrB_Descriptor desc);
/* (round 5) A performance hint, never a semantic one: the caller's labels are POPULARITY RANKS -- column 0 is the most referred-to column,
 * heavy rows come first -- e.g. because the application relabelled its graph once (rows and columns by the same ranking) before sharding it
 * by rows.  The library then builds the popularity-ordered layouts (hub level, hot strips whose LDS heads are lines of the operand, no
 * per-call operand image, empty-tail skip, sorted row tiles) in the caller's own index order: no permutation, no vector is ever converted,
 * square and non-square matrices alike -- what GrB_mxv does by itself for a large square matrix, for the row blocks of a sharded run. */
GrB_Info GrX_Matrix_hint_ranked(GrB_Matrix A, int ranked);
/* Round 6: a row block of a sharded graph takes a column order derived from the GLOBAL column reference counts (identical on every rank:
 * the host layer all-reduces the ranks' histograms once) -- the library ranks the columns itself and builds the block's ordered layouts in
 * that order, whatever the application's labels are.  `like` != NULL: share the order object of another set-up block of the same width.
 * Where the reference would carry this: graphblas/core/ss/context.py:71-85 (`ngpus` / per-object settings), SURVEY.md section 5. */
GrB_Info GrX_Matrix_shard_setup(GrB_Matrix A, const uint32_t *col_counts, int on_device, const GrB_Matrix like);
/* Borrow the device CSR of A (valid until A is modified or freed). */
GrB_Info GrX_Matrix_export_CSR_device(const int64_t **d_Ap, const int32_t **d_Aj, const void **d_Ax, GrB_Index *nvals,
                                      int *iso, const GrB_Matrix A);
/* Dense-with-presence vector image in HBM: d_val = n values, d_present = bit-packed presence
 * (bit i of 32-bit word i>>5), NULL meaning "all n entries present".  Always copies. */
GrB_Info GrX_Vector_import_dense_device(GrB_Vector *v, GrB_Type type, GrB_Index n, const void *d_val,
                                        const uint32_t *d_present);
/* Borrow the device image of v (the pointers stay valid until v is resized, cleared or freed). *d_present has ceil(n/64)*2 words.
 * The image is in natural index order when the call returns AND STAYS SO: an export pins the vector (round 5; before, a later product
 * with a large square matrix could leave v in that matrix's vertex order inside the very buffers the caller still aliased).  Products
 * that involve a pinned vector run on the natural-order layouts of their matrix.  A caller that is done with the pointers -- or that
 * re-exports after every library call anyway -- hands the vector back with GrX_Vector_pin_natural(v, 0). */
GrB_Info GrX_Vector_export_dense_device(const void **d_val, const uint32_t **d_present, const GrB_Vector v);
/* pinned != 0: v is never left in another than the natural index order (its image is aliased outside the library for longer than
 * one call: RCCL buffers, torch views); products that involve it run on the natural-order layouts.  pinned == 0: the caller holds no
 * pointer into v any more (or will not use one across a library call): v may live in a matrix's vertex order again. */
GrB_Info GrX_Vector_pin_natural(GrB_Vector v, int pinned);
/* Tell the library that the caller wrote into the image returned by GrX_Vector_export_dense_device
 * (e.g. an RCCL all-gather landed there): the cached entry count is dropped. */
GrB_Info GrX_Vector_modified(GrB_Vector v);
/* Build and cache the transpose now (otherwise built lazily on the first T0/vxm use). */
GrB_Info GrX_Matrix_cache_transpose(GrB_Matrix A);
/* The hipStream_t the library launches on (NULL: the null stream). */
GrB_Info GrX_get_stream(void **hip_stream);
/* Launch on this hipStream_t (default: the null stream, which orders with torch's default stream). */
GrB_Info GrX_set_stream(void *hip_stream);
GrB_Info GrX_synchronize(void);
/* Return the device memory the library caches (its size-class block cache and the stream-ordered HIP pool behind it) to the
 * driver: for a process that shares the GPU with another allocator (torch) between two phases with different footprints. */
GrB_Info GrX_trim_memory(void);
/* HIP-event stopwatch on the library's stream. */
GrB_Info GrX_timer_start(void);
GrB_Info GrX_timer_stop(float *elapsed_ms);
/* Statistics of the most recent mxv/vxm/mxm call (for bench/roofline bookkeeping). */
typedef struct {
    int64_t kernel_launches;  /* HIP kernels launched by the call */
    int64_t tiles;            /* merge-path tiles (mxv/vxm) or row bins (mxm) */
    int64_t flops;            /* mxm: sum_k nnz(A(:,k)) nnz(B(k,:)) from the symbolic pass; mxv: nnz(A) */
    int64_t out_nvals;        /* mxm: nnz(T); mxv: -1 (not counted) */
    int32_t method;           /* 1 pull SpMV, 2 push (SpMSpV), 3 hash SpGEMM, 4 mask-driven SpGEMM, 5 mxv by row length (PAIR, full operand), 6 empty operand: write rule only, 7 SpGEMM with the complemented mask fused into the product */
    int32_t fused_epilogue;   /* 1 if mask/accum/replace were applied inside the product kernel (2: by the short-row kernel with the LDS head) */
    int64_t hot_k;            /* mxv/vxm: entries of the hot-column table used by the call (0 = none) */
    int64_t long_entries;     /* mxv/vxm over a split matrix: entries held by the long rows (0 = no split) */
    int64_t long_segments;    /* ... as class strips: (class, sub-range, row) segments = atomics of an unmasked call */
    int32_t long_kernel;      /* ... long-row kernel that ran: 0 chunks, 1 class items, 2 mixed class strips, 4 hot / cold strips; -1 = no split */
    int32_t reorders;         /* vectors this call converted between vertex orders (0 in the steady state of a loop) */
    int64_t ordered;          /* 1: the product ran on the matrix's popularity-ordered layouts with operands kept in that order */
    int64_t value_dict;       /* distinct values of the matrix when its hot-strip records carry one-byte value codes (0: full values) */
    int64_t fill_absent;      /* 1: a sparse operand was run as a full one with the multiply's absorbing value under its absent entries */
    int64_t long_probe;       /* BOOL product under a terminal monoid (LOR.LAND, ANY.PAIR): entries of every admitted long row tested bottom-up before the item kernels (0 = not probed) */
    int64_t long_tails;       /* 1: the cold entries of the long rows below the hub level ran with the short rows (sorted row tiles / tagged row groups), merged with the strips' accumulators (option "cold_in_rows") */
    int64_t pinned_natural;   /* 1: the product would have taken the matrix's popularity-ordered layouts but one of its vectors is pinned to the natural order (an exported device view, GrX_Vector_pin_natural): it ran the natural-order layouts */
} GrX_Stats;
GrB_Info GrX_last_stats(GrX_Stats *stats);
/* T = A (+.x) B in row batches of A whose products fit `budget_bytes` of device memory; every batch runs the full two-pass
 * SpGEMM, then only its entry count and the (wrapping, integer) sum of its values are kept: the product of a matrix whose result
 * does not fit one GPU (R-MAT scale 22: 900 GB), e.g. as the single-GPU denominator of a row-sharded multi-GPU run.  Operands must
 * have the semiring's type.  Replaces nothing in the reference: GrB_mxm (graphblas/core/matrix.py:2264-2331) materialises C. */
GrB_Info GrX_mxm_streamed(const GrB_Semiring semiring, const GrB_Matrix A, const GrB_Matrix B, uint64_t budget_bytes,
                          uint64_t *nvals, uint64_t *checksum, uint64_t *flops, uint64_t *batches);
/* Same shape, same pattern and values equal (rel_tol = abs_tol = 0) or close (|a - b| <= max(rel_tol max(|a|, |b|), abs_tol), the
 * reference's _isclose, core/operator/binary.py:329), compared on the
 * device in a common type (reference Matrix.isequal / isclose, core/matrix.py:373-467, do it with eWiseMult + reduce). */
GrB_Info GrX_Matrix_isclose(bool *result, const GrB_Matrix A, const GrB_Matrix B, double rel_tol, double abs_tol);
/* A copy with the values cast to `type`, made on the device (reference dup(dtype=...), core/matrix.py:469-497 and
 * core/vector.py:392-420: a new object of the target type, then `rv << self`). */
GrB_Info GrX_Matrix_dup_as(GrB_Matrix *C, const GrB_Type type, const GrB_Matrix A);
GrB_Info GrX_Vector_dup_as(GrB_Vector *w, const GrB_Type type, const GrB_Vector u);
/* Write the monoid's identity into the VALUE image of v wherever v holds no entry (for ANY, which has no identity: the smallest value of
 * the type); the presence words stay.  The image can then be all-reduced as it is: the monoid all-reduce of the row-sharded vxm
 * (sharded.allreduce_monoid; reference call graphblas/core/vector.py:1309-1378 run on N ranks). */
GrB_Info GrX_Vector_fill_absent(GrB_Vector v, const GrB_Monoid monoid);
/* Device bytes of the SpMV layouts cached with A so far (hot-coded columns, short part, long-row strips / items). */
GrB_Info GrX_Matrix_cache_bytes(const GrB_Matrix A, uint64_t *bytes);
/* Tuning / diagnostics knobs (also read from the environment at GrB_init as GRB_<NAME upper-case>):
 *   "debug_flags"   path selectors that keep results right: 128 no long/short row split, 256 no LDS bitmap in the symbolic
 *                   SpGEMM pass, 2048 no (presence, value) packing for BOOL, 65536 no row-length path for (monoid, PAIR) over a
 *                   full operand; any other bit is GrB_INVALID_VALUE.  The kernel ablation switches of the benchmark scripts
 *                   (some make results WRONG on purpose) exist only in a -DGRB_ABLATE build (`make -C csrc ablate`, and the
 *                   emulator build of the CPU test tier), never in the shipped library.
 *   "pull_ipt"      merge items per thread of the pull SpMV (0 = default)
 *   "hot_min_cols"  matrices with at least this many columns get a hot-column table for the pull SpMV
 *   "hot_k"         entries of that table (0 = sized to ~2 MiB of x values)
 *   "rows_tile"     (round 5) 1 (default): the short rows of a matrix in its popularity order are kept a second time as SORTED ROW TILES
 *                   (tiles of up to "rtile_rows" = 8192 | 16384 rows and about "rtile_entries" = 32768 entries, the entries of a tile sorted
 *                   by column code) and run by k_mxv_rtile / k_mxv_rtile_bool where the call allows it: a compiled semiring over a full
 *                   operand (or lor.land / any.pair over presence / value pairs); 0: the tagged row groups everywhere; 2: the tiles on the
 *                   natural-order layouts of hot-coded matrices too (measured slower there)
 *   "lazy_tagged"   (round 6) 1 (default): a matrix whose short rows have sorted row tiles lays the entries of its tagged row groups out only when a call
 *                   arrives that the tiles do not take (another semiring, a sparse operand that cannot be filled) -- from the tiles; 0: at layout build
 *   "ctile_pack"    (round 6) 0 (default; 0 .. 2; measured: no gain): 1: the cold tiles of an ordered matrix keep an entry's column (as an offset in its
 *                   column range, every range below 2^19 codes) and its row slot in one 32-bit word -- tiles of 8192 rows, 8 instead of 10 bytes per
 *                   entry; 2: and one-byte dictionary codes for the values (5 bytes per entry); 0: three streams
 *   "strip_slot16"  (round 6) 1 (default): the class strips keep a lane's accumulator slot as a 16-bit offset from the smallest slot of its chunk of
 *                   64 lanes (one 32-bit base per chunk) when every chunk's slots span fewer than 65535 rows: half the slot stream; 0: 32-bit slots
 *   "rtile_pack"    (round 6) 1 (default): the sorted row tiles of a dictionary-coded matrix with at most 2^24 columns keep an entry's column code and
 *                   value code in one 32-bit word (6 instead of 7 bytes per entry, no separate value stream); 0: separate streams
 *   "cold_in_rows"  (round 6) 0 (default; measured slower when on): > 0: on an ordered matrix the long rows with fewer entries than this (and than
 *                   "hub_min_len") keep only their LDS-resident entries in the hot strips; their other, COLD entries are stored with the short rows
 *                   (sorted row tiles, tagged row groups) and the short-row kernel merges the strips' accumulator into the row's result.
 *                   0: every cold entry of a long row in the cold tiles (k_mxv_ctile), whose XCD-pinned column ranges fetch an operand line once
 *                   per XCD instead of once per gather
 *   "stream_nt_min_nnz" (round 5) 48 Mi: matrices with at least this many entries read the streams of their row tiles and cold tiles non-temporal
 *                   (the streams no longer displace the operand's lines from L2); smaller ones -- whose layouts the 256 MB infinity cache keeps
 *                   from call to call -- keep plain loads
 *   "bool_probe"    (round 5) 8 (default; 0 .. 16): BOOL products under a terminal monoid (lor.land over an iso matrix, any.pair) with an operand
 *                   that is not full test the first "bool_probe" entries of every admitted long row before the item kernels run -- the bottom-up
 *                   step of a direction-optimising BFS: a row whose partner is found there is decided; 0: never
 *   "rows_head"     1 (default): the short rows of an ordered BOOL matrix multiplied with a BOOL operand that is not full run in persistent
 *                   workgroups that keep the presence / value pairs of the hottest columns in LDS; 0: never.
 *                   "rows_head_min_groups" (16384): ... for matrices with at least this many groups of 64 rows
 *   "push_small"    1 (default): a pushed frontier whose rows hold at most 64 work items of 1024 entries runs the three push passes in ONE
 *                   workgroup (frontier kernel, one host read, one kernel); 0: always the pass-per-kernel form
 *   "fill_absent"   1 (default): floating-point min_plus / max_plus products with a sparse operand on an ordered matrix whose values are all
 *                   finite run the full-operand kernels on an image with +-inf under the absent entries (exact: an operand that holds an
 *                   infinity takes the general path); 0: never
 *   "hub_min_len"   rows of an ordered matrix from this many entries (default 1024) are dealt to 64 column classes instead of 16; 0: no hub level
 *   "value_dict"    1 (default): a matrix of a 4-byte type with at most 256 distinct finite values keeps one-byte value codes in the
 *                   lane records of its hot strips (half the bytes of the stream that bounds them; exact: codes stand for bit patterns); 0: never
 *   "order_mode"    1 (default): square matrices with at least "order_min_nnz" (24 Mi; 48 Mi before round 5) entries get their pull layouts in a vertex
 *                   order of their own -- vertices by falling column count -- and the vectors they are multiplied with are KEPT in
 *                   that order between calls (converted on first use, converted back by every entry point that is not element-wise:
 *                   build / extractTuples / export / indexed assign and extract / a product with another matrix); results are the
 *                   ones of the natural order, element for element.  0: never.  GrX_Stats.ordered / .reorders report it per call.
 *   "split_min_nnz" matrices with at least this many entries get the long/short row split of the pull SpMV
 *   "split_min_len" a row is "long" from this many entries (default 0 = 64 for the class strips, 256 for the item kernel)
 *   "lazy_layout"   1 (default): the cached SpMV layouts of a matrix with at least "lazy_min_nnz" (4 Mi) entries are built at its
 *                   second pull product; the first one runs on the CSR arrays as they are.  0: built at the first product.
 *   "drop_hot_cols" 1 (default): the hot-coded copy of a matrix's whole column array is released once the long / short split has
 *                   been built from it (the split's parts carry their own re-coded columns)
 *   "long_kernel"   layout / kernel of the long rows: 5 (default) by type and size -- items for BOOL matrices, hot strips + cold tiles
 *                   from "lean_min_nnz" (48 Mi) entries, mixed class strips below; 4 hot strips + cold tiles for every matrix (entries
 *                   whose column is LDS-resident in its class as 16-bit lane records, k_mxv_hstrip; the others as tagged tiles of
 *                   ~2 MiB column ranges, k_mxv_ctile; operands the lean hot kernel does not take: k_mxv_strip); 3 mixed class
 *                   strips, items for BOOL matrices; 2 mixed class strips (k_mxv_strip); 1 class-partitioned items (k_mxv_long_grp);
 *                   0 chunks straight from the CSR arrays (k_mxv_long)
 *   "long_classes"  column classes of the class strips: 8, 16 (default), 32 or 64 distinct LDS heads across the chip
 *   "short_kernel"  short rows of a split matrix: 6 (default) by size -- 5 from "lean_min_nnz" entries, 1 below; 5 tagged row groups (the
 *                   row of every entry stored with it, k_mxv_rows_tag),
 *                   1 one wavefront per 64 rows with row marks and a segmented fold (k_mxv_rows), 0 merge-path tiles.  (2 / 3 / 4 were
 *                   three alternatives measured slower in rounds 1-2 -- DESIGN.md section 4.1.3 -- and are GrB_INVALID_VALUE now.)
 *   "long_sub"      sub-ranges per class of the cold columns of the long rows (0 = sized from the operand image),
 *   "long_sub_min_len"  for rows from this many entries (0 = 512 per sub-range)
 *   "mxm_mask_mode" mask-driven SpGEMM for non-complemented masks: 1 (default) when the product costs clearly more than the mask,
 *                   0 never, 2 always;  complemented masks: fused into the product (the forbidden positions never enter it)
 *                   unless 0 (full product, then the write rule)
 *   "mat_write_kernel"  the write rule C<M,replace> = accum(C, T) of matrix results: 1 (default) a wavefront per row, long rows
 *                   cut into column pieces; 0 a thread per row (rounds 1-2: 1.6 s on a 9.7 G-entry T)
 *   "mxm_heavy_kernel"  SpGEMM rows beyond the LDS hash tables: 1 (default) (row, column window) work units (k_spgemm_unit),
 *                   0 the 1024-thread row kernels of round 1
 *   "mxm_window_groups"  (round 5) the (row, window) units of a product walk GROUPS of this many 16 Ki-column windows (1 / 2 / 4 / 8;
 *                   0 = default: 1 up to 64 windows per row, 2 beyond -- measured: wider groups lose more occupancy to their bitmaps
 *                   than they save units); a group that holds more entries than the densest compact class is walked window by
 *                   window as before
 *   "mxm_xcd_map"   (round 5) 1 (default): the SpGEMM unit kernels give every XCD a contiguous eighth of the unit order, so that the windows
 *                   of a row -- which read the same rows of B -- share one L2; 0: workgroups take the units round-robin
 *   "mxm_checksum_pass"  GrX_mxm_streamed: 0 (default) the checksum of the product is folded into the numeric kernels' stores, 1 a pass
 *                   of its own re-reads every batch's values (round 4)
 *   "mxm_sym_windows"  consecutive column windows (groups of windows) of a row one symbolic unit walks (rows of up to 128 entries of A; default 8,
 *                   1 = one window per unit as in round 2, at most 64)
 *   "mxm_unit_min_flops" / "mxm_unit_min_per_window"  rows with more products than this (1024) and than this many per column window
 *                   (16), at most 4096, are walked as units
 *   "mxm_unit_small" / "mxm_unit_mid" / "mxm_unit_dense"  entry counts of a unit up to which one wavefront with 512 accumulators /
 *                   four wavefronts with 1024 / with 4096 accumulators take it (512, 1024, 4096); denser units get an
 *                   accumulator per column of the window
 *   "mxm_bitmap_pool_mb"  device memory (MiB, default 16384; never more than a quarter of the free memory) for the bitmaps of
 *                   units the symbolic pass keeps for the numeric pass; "mxm_bitmap_pool_cap": the same as a count of bitmaps
 *   "push_mode"     mxv/vxm direction: 0 always pull, 1 (default) push when u has fewer than n/64 entries and the
 *                   matrix indexed like u is at hand, 2 always push when possible */
GrB_Info GrX_option_set(const char *name, int64_t value);
const char *GrX_version_string(void);

/* ================================================================================================================
 * Import-time surface beyond the hot path: what an unmodified python-graphblas resolves on `lib`
 * while it is IMPORTED -- handle types, builtin unary / index-unary / remaining binary operators as data symbols
 * (graphblas/core/mask.py:1-5 `select.valuene`, `unary.one`; core/operator/base.py:803-893 regex discovery) -- and the entry
 * points of the operations this library does not accelerate.  Those entry points exist with their C API 2.0 signatures and
 * return GrB_NOT_IMPLEMENTED; operator handles the kernels do not implement are rejected with GrB_NOT_IMPLEMENTED wherever
 * a call consumes them.  GrB_Scalar is real (a host-side value + presence).  scripts/lib_surface_report.py diffs the names
 * the reference touches against `nm -D` of the built library.
 * ================================================================================================================ */
#define GRB_DECL_UNARY(T)                                                                                              \
    extern GrB_UnaryOp GrB_IDENTITY_##T, GrB_AINV_##T, GrB_MINV_##T, GrB_ABS_##T, GxB_ONE_##T, GxB_LNOT_##T;           \
    extern GrB_IndexUnaryOp GrB_VALUEEQ_##T, GrB_VALUENE_##T, GrB_VALUEGT_##T, GrB_VALUEGE_##T, GrB_VALUELT_##T,       \
        GrB_VALUELE_##T;                                                                                               \
    extern GrB_BinaryOp GrB_DIV_##T, GxB_RDIV_##T, GxB_RMINUS_##T, GxB_ISEQ_##T, GxB_ISNE_##T, GxB_ISGT_##T,           \
        GxB_ISLT_##T, GxB_ISGE_##T, GxB_ISLE_##T, GxB_POW_##T;
GRB_FOR_EACH_TNAME(GRB_DECL_UNARY)
#undef GRB_DECL_UNARY
extern GrB_UnaryOp GrB_LNOT, GrB_BNOT_INT8, GrB_BNOT_INT16, GrB_BNOT_INT32, GrB_BNOT_INT64, GrB_BNOT_UINT8, GrB_BNOT_UINT16,
    GrB_BNOT_UINT32, GrB_BNOT_UINT64;
extern GrB_IndexUnaryOp GrB_ROWINDEX_INT32, GrB_ROWINDEX_INT64, GrB_COLINDEX_INT32, GrB_COLINDEX_INT64, GrB_DIAGINDEX_INT32,
    GrB_DIAGINDEX_INT64, GrB_TRIL, GrB_TRIU, GrB_DIAG, GrB_OFFDIAG, GrB_COLLE, GrB_COLGT, GrB_ROWLE, GrB_ROWGT;

GrB_Info GrB_Scalar_new(GrB_Scalar *s, GrB_Type type);
GrB_Info GrB_Scalar_dup(GrB_Scalar *s, const GrB_Scalar t);
GrB_Info GrB_Scalar_free(GrB_Scalar *s);
GrB_Info GrB_Scalar_clear(GrB_Scalar s);
GrB_Info GrB_Scalar_nvals(GrB_Index *nvals, const GrB_Scalar s);
GrB_Info GrB_Scalar_wait(GrB_Scalar s, GrB_WaitMode mode);
GrB_Info GrB_Scalar_error(const char **error, const GrB_Scalar s);

GrB_Info GrB_Vector_apply(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_UnaryOp op, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_apply(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_UnaryOp op, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseAdd_Semiring(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseMult_Semiring(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
#define GRB_DECL_MAT_BINARY(FN, HANDLE) \
    GrB_Info FN(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const HANDLE op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GRB_DECL_MAT_BINARY(GrB_Matrix_eWiseAdd_BinaryOp, GrB_BinaryOp)
GRB_DECL_MAT_BINARY(GrB_Matrix_eWiseAdd_Monoid, GrB_Monoid)
GRB_DECL_MAT_BINARY(GrB_Matrix_eWiseAdd_Semiring, GrB_Semiring)
GRB_DECL_MAT_BINARY(GrB_Matrix_eWiseMult_BinaryOp, GrB_BinaryOp)
GRB_DECL_MAT_BINARY(GrB_Matrix_eWiseMult_Monoid, GrB_Monoid)
GRB_DECL_MAT_BINARY(GrB_Matrix_eWiseMult_Semiring, GrB_Semiring)
GRB_DECL_MAT_BINARY(GrB_Matrix_kronecker_BinaryOp, GrB_BinaryOp)
GRB_DECL_MAT_BINARY(GrB_Matrix_kronecker_Monoid, GrB_Monoid)
GRB_DECL_MAT_BINARY(GrB_Matrix_kronecker_Semiring, GrB_Semiring)
#undef GRB_DECL_MAT_BINARY
GrB_Info GrB_Matrix_assign(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Row_assign(GrB_Matrix C, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, GrB_Index i, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Col_assign(GrB_Matrix C, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index *I, GrB_Index ni, GrB_Index j, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_extract(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Col_extract(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index *I, GrB_Index ni, GrB_Index j, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_reduce_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_reduce_Monoid_Scalar(GrB_Scalar s, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Vector_reduce_Monoid_Scalar(GrB_Scalar s, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_removeElement(GrB_Matrix C, GrB_Index i, GrB_Index j);
GrB_Info GrB_Matrix_diag(GrB_Matrix *C, const GrB_Vector v, int64_t k);
GrB_Info GrB_Vector_assign_Scalar(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Scalar s, const GrB_Index *I, GrB_Index ni, const GrB_Descriptor desc);
GrB_Info GrB_Vector_setElement_Scalar(GrB_Vector w, const GrB_Scalar s, GrB_Index i);
GrB_Info GrB_Vector_extractElement_Scalar(GrB_Scalar s, const GrB_Vector u, GrB_Index i);
GrB_Info GrB_Matrix_assign_Scalar(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Scalar s, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_setElement_Scalar(GrB_Matrix C, const GrB_Scalar s, GrB_Index i, GrB_Index j);
GrB_Info GrB_Matrix_extractElement_Scalar(GrB_Scalar s, const GrB_Matrix A, GrB_Index i, GrB_Index j);
GrB_Info GrB_Vector_select_Scalar(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Vector u, const GrB_Scalar y, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_select_Scalar(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Matrix A, const GrB_Scalar y, const GrB_Descriptor desc);
GrB_Info GrB_Vector_apply_BinaryOp1st_Scalar(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Scalar x, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GrB_Vector_apply_BinaryOp2nd_Scalar(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, const GrB_Scalar y, const GrB_Descriptor desc);
GrB_Info GrB_Vector_apply_IndexOp_Scalar(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Vector u, const GrB_Scalar y, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_apply_BinaryOp1st_Scalar(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Scalar x, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_apply_BinaryOp2nd_Scalar(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Scalar y, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_apply_IndexOp_Scalar(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Matrix A, const GrB_Scalar y, const GrB_Descriptor desc);
/* user-defined types and operators need a JIT (numba -> C callbacks in the reference): not on this library's path */
GrB_Info GrB_Type_new(GrB_Type *type, size_t size);
GrB_Info GrB_UnaryOp_new(GrB_UnaryOp *op, void *fn, GrB_Type ztype, GrB_Type xtype);
GrB_Info GrB_BinaryOp_new(GrB_BinaryOp *op, void *fn, GrB_Type ztype, GrB_Type xtype, GrB_Type ytype);
GrB_Info GrB_IndexUnaryOp_new(GrB_IndexUnaryOp *op, void *fn, GrB_Type ztype, GrB_Type xtype, GrB_Type ytype);
GrB_Info GrB_Semiring_new(GrB_Semiring *semiring, GrB_Monoid add, GrB_BinaryOp mult);
#define GRB_DECL_SURFACE_TYPED(NAME, ctype)                                                                                                  \
    GrB_Info GrB_Scalar_setElement_##NAME(GrB_Scalar s, ctype x);                                                                            \
    GrB_Info GrB_Scalar_extractElement_##NAME(ctype *x, const GrB_Scalar s);                                                                 \
    GrB_Info GrB_Monoid_new_##NAME(GrB_Monoid *monoid, GrB_BinaryOp op, ctype identity);                                                     \
    GrB_Info GrB_Matrix_setElement_##NAME(GrB_Matrix C, ctype x, GrB_Index i, GrB_Index j);                                                  \
    GrB_Info GrB_Matrix_extractElement_##NAME(ctype *x, const GrB_Matrix A, GrB_Index i, GrB_Index j);                                       \
    GrB_Info GrB_Matrix_reduce_##NAME(ctype *val, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A, const GrB_Descriptor desc); \
    GrB_Info GrB_Matrix_assign_##NAME(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, ctype x, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc); \
    GrB_Info GrB_Vector_select_##NAME(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Vector u, ctype y, const GrB_Descriptor desc); \
    GrB_Info GrB_Matrix_select_##NAME(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Matrix A, ctype y, const GrB_Descriptor desc); \
    GrB_Info GrB_Vector_apply_BinaryOp1st_##NAME(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, ctype x, const GrB_Vector u, const GrB_Descriptor desc); \
    GrB_Info GrB_Vector_apply_BinaryOp2nd_##NAME(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, ctype y, const GrB_Descriptor desc); \
    GrB_Info GrB_Vector_apply_IndexOp_##NAME(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Vector u, ctype y, const GrB_Descriptor desc); \
    GrB_Info GrB_Matrix_apply_BinaryOp1st_##NAME(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, ctype x, const GrB_Matrix A, const GrB_Descriptor desc); \
    GrB_Info GrB_Matrix_apply_BinaryOp2nd_##NAME(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, ctype y, const GrB_Descriptor desc); \
    GrB_Info GrB_Matrix_apply_IndexOp_##NAME(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_IndexUnaryOp op, const GrB_Matrix A, ctype y, const GrB_Descriptor desc);
GRB_FOR_EACH_TYPE(GRB_DECL_SURFACE_TYPED)
#undef GRB_DECL_SURFACE_TYPED

#ifdef __cplusplus
}
#endif
#endif /* GRB_MI355X_H */
