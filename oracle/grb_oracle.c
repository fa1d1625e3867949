/* ============================================================================
 * TEST INFRASTRUCTURE ONLY.  CPU oracle for the GraphBLAS mxv / vxm / mxm hot
 * path.  Nothing in the product (python-graphblas_amd/, libgrb_mi355x.so) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * What it restates.  The reference (python-graphblas) has no arithmetic of its
 * own on this path: graphblas/core/matrix.py:2254 ("GrB_mxv"), :2321
 * ("GrB_mxm"), graphblas/core/vector.py:1370 ("GrB_vxm") name a C function and
 * graphblas/core/base.py:496-503 dispatches it into the third-party
 * SuiteSparse:GraphBLAS (PyPI suitesparse-graphblas >=7.4.0.0,
 * pyproject.toml:64,75), which is NOT under /root/reference and not installed
 * here.  This file therefore restates the published GraphBLAS C API 2.0
 * semantics of those three calls (semiring product, accumulator, mask,
 * replace) and is PINNED against every known-answer literal the reference's
 * own tests/docs hold for the path (tests/golden/reference_literals.json,
 * checked by tests/test_oracle_golden.py).  lor_land / any_pair have no
 * value-level literal in the reference: for those the
 * oracle is cross-checked against an independent dense brute-force evaluator
 * (oracle/dense_eval.py) -- "parity unpinned by the reference" for those two.
 *
 * Build: see oracle/Makefile  (gcc -O3 -fopenmp -shared -fPIC).
 * ==========================================================================*/
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

enum { ORA_BOOL = 0, ORA_INT8, ORA_INT16, ORA_INT32, ORA_INT64,
       ORA_UINT8, ORA_UINT16, ORA_UINT32, ORA_UINT64, ORA_FP32, ORA_FP64 };

/* binary ops / monoids share one code space */
enum { ORA_FIRST = 0, ORA_SECOND, ORA_PAIR, ORA_PLUS, ORA_MINUS, ORA_TIMES,
       ORA_MIN, ORA_MAX, ORA_LOR, ORA_LAND, ORA_LXOR, ORA_LXNOR, ORA_ANY };

static int ora_cmp_i64(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

#define ORA_CAT2(a, b) a##_##b
#define ORA_CAT(a, b) ORA_CAT2(a, b)
#define FN(name) ORA_CAT(name, SUF)

#define ORA_IS_BOOL
#define T uint8_t
#define UT uint8_t
#define SUF bool
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#undef ORA_IS_BOOL

#define ORA_IS_INT
#define T int8_t
#define UT uint8_t
#define SUF i8
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T int16_t
#define UT uint16_t
#define SUF i16
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T int32_t
#define UT uint32_t
#define SUF i32
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T int64_t
#define UT uint64_t
#define SUF i64
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T uint8_t
#define UT uint8_t
#define SUF u8
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T uint16_t
#define UT uint16_t
#define SUF u16
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T uint32_t
#define UT uint32_t
#define SUF u32
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T uint64_t
#define UT uint64_t
#define SUF u64
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#undef ORA_IS_INT

#define ORA_IS_FP
#define T float
#define UT float
#define SUF f32
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#define T double
#define UT double
#define SUF f64
#include "grb_oracle_typed.inc"
#undef T
#undef UT
#undef SUF
#undef ORA_IS_FP

#define DISPATCH(type, CALL)                                  \
    switch (type) {                                           \
    case ORA_BOOL:   CALL(bool, uint8_t);  break;             \
    case ORA_INT8:   CALL(i8, int8_t);     break;             \
    case ORA_INT16:  CALL(i16, int16_t);   break;             \
    case ORA_INT32:  CALL(i32, int32_t);   break;             \
    case ORA_INT64:  CALL(i64, int64_t);   break;             \
    case ORA_UINT8:  CALL(u8, uint8_t);    break;             \
    case ORA_UINT16: CALL(u16, uint16_t);  break;             \
    case ORA_UINT32: CALL(u32, uint32_t);  break;             \
    case ORA_UINT64: CALL(u64, uint64_t);  break;             \
    case ORA_FP32:   CALL(f32, float);     break;             \
    case ORA_FP64:   CALL(f64, double);    break;             \
    default: return -2;                                       \
    }

/* ---- exported entry points (ctypes; see oracle/grb_oracle.py) ------------ */

int grbo_mxv(int type, int monoid, int mult, int64_t nrows,
             const int64_t *Ap, const int64_t *Aj, const void *Ax, int A_iso,
             const uint8_t *u_has, const void *u_val, const uint8_t *row_active,
             uint8_t *t_has, void *t_val)
{
#define CALL(s, ty) mxv_##s(monoid, mult, nrows, Ap, Aj, (const ty *)Ax, A_iso, u_has, (const ty *)u_val, row_active, t_has, (ty *)t_val)
    DISPATCH(type, CALL)
#undef CALL
    return 0;
}

int grbo_vec_write(int type, int64_t n, uint8_t *w_has, void *w_val,
                   const uint8_t *t_has, const void *t_val,
                   const uint8_t *mask_true, int mask_comp, int accum, int replace)
{
#define CALL(s, ty) vec_write_##s(n, w_has, (ty *)w_val, t_has, (const ty *)t_val, mask_true, mask_comp, accum, replace)
    DISPATCH(type, CALL)
#undef CALL
    return 0;
}

int grbo_mxm(int type, int monoid, int mult, int64_t nrows, int64_t ncols,
             const int64_t *Ap, const int64_t *Aj, const void *Ax, int A_iso,
             const int64_t *Bp, const int64_t *Bj, const void *Bx, int B_iso,
             const int64_t *Fp, const int64_t *Fj, int F_comp,
             int64_t **Tp, int64_t **Tj, void **Tx)
{
    int rc = 0;
#define CALL(s, ty) rc = mxm_##s(monoid, mult, nrows, ncols, Ap, Aj, (const ty *)Ax, A_iso, Bp, Bj, (const ty *)Bx, B_iso, Fp, Fj, F_comp, Tp, Tj, (ty **)Tx)
    DISPATCH(type, CALL)
#undef CALL
    return rc;
}

int grbo_mat_write(int type, int64_t nrows,
                   const int64_t *Cp, const int64_t *Cj, const void *Cx,
                   const int64_t *Tp, const int64_t *Tj, const void *Tx,
                   const int64_t *Mp, const int64_t *Mj, int has_mask, int mask_comp,
                   int accum, int replace,
                   int64_t **Np, int64_t **Nj, void **Nx)
{
    int rc = 0;
#define CALL(s, ty) rc = mat_write_##s(nrows, Cp, Cj, (const ty *)Cx, Tp, Tj, (const ty *)Tx, Mp, Mj, has_mask, mask_comp, accum, replace, Np, Nj, (ty **)Nx)
    DISPATCH(type, CALL)
#undef CALL
    return rc;
}

void grbo_free(void *p) { free(p); }

/* CSR transpose (counting sort; keeps columns sorted).  esize = bytes per value (0 = pattern only). */
int grbo_transpose(int64_t nrows, int64_t ncols, const int64_t *Ap, const int64_t *Aj,
                   const void *Ax, int64_t esize, int64_t *Bp, int64_t *Bj, void *Bx)
{
    const int64_t nnz = Ap[nrows];
    memset(Bp, 0, sizeof(int64_t) * (size_t)(ncols + 1));
    for (int64_t p = 0; p < nnz; p++) Bp[Aj[p] + 1]++;
    for (int64_t j = 0; j < ncols; j++) Bp[j + 1] += Bp[j];
    int64_t *next = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncols + 1));
    if (!next) return -1;
    memcpy(next, Bp, sizeof(int64_t) * (size_t)(ncols + 1));
    for (int64_t i = 0; i < nrows; i++)
        for (int64_t p = Ap[i]; p < Ap[i + 1]; p++) {
            const int64_t q = next[Aj[p]]++;
            Bj[q] = i;
            if (esize) memcpy((char *)Bx + q * esize, (const char *)Ax + p * esize, (size_t)esize);
        }
    free(next);
    return 0;
}

#ifdef _OPENMP
#include <omp.h>
int grbo_num_threads(void) { return omp_get_max_threads(); }
void grbo_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int grbo_num_threads(void) { return 1; }
void grbo_set_num_threads(int n) { (void)n; }
#endif
