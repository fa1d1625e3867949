// grb_globals.hip -- process-lifetime builtin handles: types, binary ops, monoids, semirings,
// descriptors.  The reference discovers these by name with regexes over dir(lib)
// (graphblas/core/operator/base.py:803-893, core/operator/semiring.py:185-219) and never frees them
// (core/operator/base.py:540-547, core/descriptor.py:42-48).
#include "grb_internal.hpp"

using namespace grb;

static const uint64_t grb_all_sentinel = 0;
extern "C" const uint64_t *GrB_ALL = &grb_all_sentinel;

// ---- types -------------------------------------------------------------------------------------------
#define DEF_TYPE(NAME, ctype)                                                              \
    static GB_Type_opaque type_obj_##NAME = {TC_##NAME, sizeof(ctype), "GrB_" #NAME};      \
    extern "C" GrB_Type GrB_##NAME = &type_obj_##NAME;
GRB_FOR_EACH_TYPE(DEF_TYPE)
#undef DEF_TYPE

namespace grb {
GrB_Type type_of_code(int code)
{
    static GrB_Type table[TC_COUNT] = {GrB_BOOL,  GrB_INT8,   GrB_INT16,  GrB_INT32, GrB_INT64, GrB_UINT8,
                                       GrB_UINT16, GrB_UINT32, GrB_UINT64, GrB_FP32,  GrB_FP64};
    return table[code];
}
}  // namespace grb

// ---- binary ops, ANY monoid, ANY_* semirings for every type ----------------------------------------------
#define DEF_BINOP(SYM, OP, T)                                          \
    static GB_BinaryOp_opaque bop_obj_##SYM = {OP, TC_##T, #SYM};      \
    extern "C" GrB_BinaryOp SYM = &bop_obj_##SYM;
#define DEF_MONOID(SYM, OP, T)                                         \
    static GB_Monoid_opaque mon_obj_##SYM = {OP, TC_##T, #SYM};        \
    extern "C" GrB_Monoid SYM = &mon_obj_##SYM;
#define DEF_SEMIRING(SYM, MON, MUL, T)                                 \
    static GB_Semiring_opaque sr_obj_##SYM = {MON, MUL, TC_##T, #SYM}; \
    extern "C" GrB_Semiring SYM = &sr_obj_##SYM;

#define DEF_ALL_TYPES(T)                           \
    DEF_BINOP(GrB_FIRST_##T, OP_FIRST, T)          \
    DEF_BINOP(GrB_SECOND_##T, OP_SECOND, T)        \
    DEF_BINOP(GrB_ONEB_##T, OP_PAIR, T)            \
    DEF_BINOP(GxB_PAIR_##T, OP_PAIR, T)            \
    DEF_BINOP(GrB_PLUS_##T, OP_PLUS, T)            \
    DEF_BINOP(GrB_MINUS_##T, OP_MINUS, T)          \
    DEF_BINOP(GrB_TIMES_##T, OP_TIMES, T)          \
    DEF_BINOP(GrB_MIN_##T, OP_MIN, T)              \
    DEF_BINOP(GrB_MAX_##T, OP_MAX, T)              \
    DEF_BINOP(GxB_ANY_##T, OP_ANY, T)              \
    DEF_BINOP(GxB_LOR_##T, OP_LOR, T)              \
    DEF_BINOP(GxB_LAND_##T, OP_LAND, T)            \
    DEF_BINOP(GxB_LXOR_##T, OP_LXOR, T)            \
    DEF_BINOP(GrB_EQ_##T, OP_EQ, T)                \
    DEF_BINOP(GrB_NE_##T, OP_NE, T)                \
    DEF_BINOP(GrB_GT_##T, OP_GT, T)                \
    DEF_BINOP(GrB_LT_##T, OP_LT, T)                \
    DEF_BINOP(GrB_GE_##T, OP_GE, T)                \
    DEF_BINOP(GrB_LE_##T, OP_LE, T)                \
    DEF_MONOID(GxB_ANY_##T##_MONOID, OP_ANY, T)    \
    DEF_SEMIRING(GxB_ANY_PAIR_##T, OP_ANY, OP_PAIR, T)   \
    DEF_SEMIRING(GxB_ANY_FIRST_##T, OP_ANY, OP_FIRST, T) \
    DEF_SEMIRING(GxB_ANY_SECOND_##T, OP_ANY, OP_SECOND, T)
GRB_FOR_EACH_TNAME(DEF_ALL_TYPES)
#undef DEF_ALL_TYPES

DEF_BINOP(GrB_LOR, OP_LOR, BOOL)
DEF_BINOP(GrB_LAND, OP_LAND, BOOL)
DEF_BINOP(GrB_LXOR, OP_LXOR, BOOL)
DEF_BINOP(GrB_LXNOR, OP_LXNOR, BOOL)
DEF_MONOID(GrB_LOR_MONOID_BOOL, OP_LOR, BOOL)
DEF_MONOID(GrB_LAND_MONOID_BOOL, OP_LAND, BOOL)
DEF_MONOID(GrB_LXOR_MONOID_BOOL, OP_LXOR, BOOL)
DEF_MONOID(GrB_LXNOR_MONOID_BOOL, OP_LXNOR, BOOL)

#define DEF_NUMERIC(T)                                                      \
    DEF_MONOID(GrB_PLUS_MONOID_##T, OP_PLUS, T)                             \
    DEF_MONOID(GrB_TIMES_MONOID_##T, OP_TIMES, T)                           \
    DEF_MONOID(GrB_MIN_MONOID_##T, OP_MIN, T)                               \
    DEF_MONOID(GrB_MAX_MONOID_##T, OP_MAX, T)                               \
    DEF_SEMIRING(GrB_PLUS_TIMES_SEMIRING_##T, OP_PLUS, OP_TIMES, T)         \
    DEF_SEMIRING(GrB_PLUS_MIN_SEMIRING_##T, OP_PLUS, OP_MIN, T)             \
    DEF_SEMIRING(GrB_MIN_PLUS_SEMIRING_##T, OP_MIN, OP_PLUS, T)             \
    DEF_SEMIRING(GrB_MIN_TIMES_SEMIRING_##T, OP_MIN, OP_TIMES, T)           \
    DEF_SEMIRING(GrB_MIN_FIRST_SEMIRING_##T, OP_MIN, OP_FIRST, T)           \
    DEF_SEMIRING(GrB_MIN_SECOND_SEMIRING_##T, OP_MIN, OP_SECOND, T)         \
    DEF_SEMIRING(GrB_MIN_MAX_SEMIRING_##T, OP_MIN, OP_MAX, T)               \
    DEF_SEMIRING(GrB_MAX_PLUS_SEMIRING_##T, OP_MAX, OP_PLUS, T)             \
    DEF_SEMIRING(GrB_MAX_TIMES_SEMIRING_##T, OP_MAX, OP_TIMES, T)           \
    DEF_SEMIRING(GrB_MAX_FIRST_SEMIRING_##T, OP_MAX, OP_FIRST, T)           \
    DEF_SEMIRING(GrB_MAX_SECOND_SEMIRING_##T, OP_MAX, OP_SECOND, T)         \
    DEF_SEMIRING(GrB_MAX_MIN_SEMIRING_##T, OP_MAX, OP_MIN, T)               \
    DEF_SEMIRING(GxB_PLUS_PLUS_##T, OP_PLUS, OP_PLUS, T)                    \
    DEF_SEMIRING(GxB_PLUS_PAIR_##T, OP_PLUS, OP_PAIR, T)                    \
    DEF_SEMIRING(GxB_PLUS_FIRST_##T, OP_PLUS, OP_FIRST, T)                  \
    DEF_SEMIRING(GxB_PLUS_SECOND_##T, OP_PLUS, OP_SECOND, T)                \
    DEF_SEMIRING(GxB_PLUS_MAX_##T, OP_PLUS, OP_MAX, T)                      \
    DEF_SEMIRING(GxB_MIN_MIN_##T, OP_MIN, OP_MIN, T)                        \
    DEF_SEMIRING(GxB_MAX_MAX_##T, OP_MAX, OP_MAX, T)                        \
    DEF_SEMIRING(GxB_MIN_PAIR_##T, OP_MIN, OP_PAIR, T)                      \
    DEF_SEMIRING(GxB_MAX_PAIR_##T, OP_MAX, OP_PAIR, T)                      \
    DEF_SEMIRING(GxB_TIMES_TIMES_##T, OP_TIMES, OP_TIMES, T)                \
    DEF_SEMIRING(GxB_TIMES_PLUS_##T, OP_TIMES, OP_PLUS, T)
GRB_FOR_EACH_NUMERIC(DEF_NUMERIC)
#undef DEF_NUMERIC

DEF_SEMIRING(GrB_LOR_LAND_SEMIRING_BOOL, OP_LOR, OP_LAND, BOOL)
DEF_SEMIRING(GrB_LAND_LOR_SEMIRING_BOOL, OP_LAND, OP_LOR, BOOL)
DEF_SEMIRING(GrB_LXOR_LAND_SEMIRING_BOOL, OP_LXOR, OP_LAND, BOOL)
DEF_SEMIRING(GrB_LXNOR_LOR_SEMIRING_BOOL, OP_LXNOR, OP_LOR, BOOL)
DEF_SEMIRING(GxB_LOR_LOR_BOOL, OP_LOR, OP_LOR, BOOL)
DEF_SEMIRING(GxB_LAND_LAND_BOOL, OP_LAND, OP_LAND, BOOL)
DEF_SEMIRING(GxB_LOR_FIRST_BOOL, OP_LOR, OP_FIRST, BOOL)
DEF_SEMIRING(GxB_LOR_SECOND_BOOL, OP_LOR, OP_SECOND, BOOL)
DEF_SEMIRING(GxB_LOR_PAIR_BOOL, OP_LOR, OP_PAIR, BOOL)
DEF_SEMIRING(GxB_LAND_FIRST_BOOL, OP_LAND, OP_FIRST, BOOL)
DEF_SEMIRING(GxB_LAND_SECOND_BOOL, OP_LAND, OP_SECOND, BOOL)
DEF_SEMIRING(GxB_LOR_LXOR_BOOL, OP_LOR, OP_LXOR, BOOL)
DEF_SEMIRING(GxB_LAND_LXOR_BOOL, OP_LAND, OP_LXOR, BOOL)
DEF_SEMIRING(GxB_LXOR_LOR_BOOL, OP_LXOR, OP_LOR, BOOL)
DEF_SEMIRING(GxB_LXOR_LXOR_BOOL, OP_LXOR, OP_LXOR, BOOL)
DEF_SEMIRING(GxB_LXOR_FIRST_BOOL, OP_LXOR, OP_FIRST, BOOL)
DEF_SEMIRING(GxB_LXOR_SECOND_BOOL, OP_LXOR, OP_SECOND, BOOL)
DEF_SEMIRING(GxB_LXOR_PAIR_BOOL, OP_LXOR, OP_PAIR, BOOL)

// ---- descriptors: GrB_DESC_[R][S][C][T0][T1] ---------------------------------------------------------------
#define DEF_DESC(SYM, r, s, c, t0, t1)                                     \
    static GB_Descriptor_opaque desc_obj_##SYM = {r, c, s, t0, t1, true};  \
    extern "C" GrB_Descriptor GrB_DESC_##SYM = &desc_obj_##SYM;
DEF_DESC(T1, 0, 0, 0, 0, 1)
DEF_DESC(T0, 0, 0, 0, 1, 0)
DEF_DESC(T0T1, 0, 0, 0, 1, 1)
DEF_DESC(C, 0, 0, 1, 0, 0)
DEF_DESC(CT1, 0, 0, 1, 0, 1)
DEF_DESC(CT0, 0, 0, 1, 1, 0)
DEF_DESC(CT0T1, 0, 0, 1, 1, 1)
DEF_DESC(S, 0, 1, 0, 0, 0)
DEF_DESC(ST1, 0, 1, 0, 0, 1)
DEF_DESC(ST0, 0, 1, 0, 1, 0)
DEF_DESC(ST0T1, 0, 1, 0, 1, 1)
DEF_DESC(SC, 0, 1, 1, 0, 0)
DEF_DESC(SCT1, 0, 1, 1, 0, 1)
DEF_DESC(SCT0, 0, 1, 1, 1, 0)
DEF_DESC(SCT0T1, 0, 1, 1, 1, 1)
DEF_DESC(R, 1, 0, 0, 0, 0)
DEF_DESC(RT1, 1, 0, 0, 0, 1)
DEF_DESC(RT0, 1, 0, 0, 1, 0)
DEF_DESC(RT0T1, 1, 0, 0, 1, 1)
DEF_DESC(RC, 1, 0, 1, 0, 0)
DEF_DESC(RCT1, 1, 0, 1, 0, 1)
DEF_DESC(RCT0, 1, 0, 1, 1, 0)
DEF_DESC(RCT0T1, 1, 0, 1, 1, 1)
DEF_DESC(RS, 1, 1, 0, 0, 0)
DEF_DESC(RST1, 1, 1, 0, 0, 1)
DEF_DESC(RST0, 1, 1, 0, 1, 0)
DEF_DESC(RST0T1, 1, 1, 0, 1, 1)
DEF_DESC(RSC, 1, 1, 1, 0, 0)
DEF_DESC(RSCT1, 1, 1, 1, 0, 1)
DEF_DESC(RSCT0, 1, 1, 1, 1, 0)
DEF_DESC(RSCT0T1, 1, 1, 1, 1, 1)

extern "C" GrB_Info GrB_Descriptor_new(GrB_Descriptor *desc)
{
    if (!desc) return GrB_NULL_POINTER;
    *desc = new GB_Descriptor_opaque{false, false, false, false, false, false};
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrB_Descriptor_set(GrB_Descriptor d, GrB_Desc_Field field, GrB_Desc_Value value)
{
    if (!d) return GrB_NULL_POINTER;
    if (d->builtin) return GrB_INVALID_VALUE;
    switch (field) {
    case GrB_OUTP:
        if (value != GrB_DEFAULT && value != GrB_REPLACE) return GrB_INVALID_VALUE;
        d->replace = (value == GrB_REPLACE);
        return GrB_SUCCESS;
    case GrB_MASK:
        if (value == GrB_DEFAULT) { d->comp = d->structure = false; }
        else if (value == GrB_COMP) d->comp = true;
        else if (value == GrB_STRUCTURE) d->structure = true;
        else if ((int)value == (GrB_COMP | GrB_STRUCTURE)) d->comp = d->structure = true;
        else return GrB_INVALID_VALUE;
        return GrB_SUCCESS;
    case GrB_INP0:
        if (value != GrB_DEFAULT && value != GrB_TRAN) return GrB_INVALID_VALUE;
        d->t0 = (value == GrB_TRAN);
        return GrB_SUCCESS;
    case GrB_INP1:
        if (value != GrB_DEFAULT && value != GrB_TRAN) return GrB_INVALID_VALUE;
        d->t1 = (value == GrB_TRAN);
        return GrB_SUCCESS;
    default:
        return GrB_INVALID_VALUE;
    }
}

extern "C" GrB_Info GrB_Descriptor_free(GrB_Descriptor *desc)
{
    if (!desc) return GrB_NULL_POINTER;
    if (*desc && !(*desc)->builtin) delete *desc;
    *desc = nullptr;
    return GrB_SUCCESS;
}
