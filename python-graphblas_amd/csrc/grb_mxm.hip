// placeholder until the SpGEMM lands
#include "grb_internal.hpp"
using namespace grb;
extern "C" GrB_Info GrB_mxm(GrB_Matrix C, const GrB_Matrix, const GrB_BinaryOp, const GrB_Semiring, const GrB_Matrix,
                            const GrB_Matrix, const GrB_Descriptor)
{
    GRB_TRY
    fail(GrB_NOT_IMPLEMENTED, "GrB_mxm: not built yet");
    GRB_CATCH(errp(C))
}
