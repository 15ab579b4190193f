// a10: L1 models (placeholder translation unit until the solver kernels land).
#include "psk_internal.h"
extern "C" int psk_logreg_l1_fit(psk_ctx *ctx, const uint8_t *, const int32_t *, int, int, const int32_t *,
                                 const double *, const int32_t *, int, double, int, double *, double *, int32_t *)
{
    return psk_fail(ctx, PSK_ESTATE, "psk_logreg_l1_fit: not built yet");
}
extern "C" int psk_lasso_fit(psk_ctx *ctx, const uint8_t *, const double *, int, int, const int32_t *, const double *,
                             const int32_t *, int, double, int, double *, double *, int32_t *)
{
    return psk_fail(ctx, PSK_ESTATE, "psk_lasso_fit: not built yet");
}
