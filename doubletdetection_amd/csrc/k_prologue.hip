// fit() prologue on the device (dd.py:165-176): per-gene float32 variances and column restriction.
#include "ddx_internal.h"

namespace ddx {

int stage_gene_variances(ddx_ctx* ctx, float* var_out) {
    (void)var_out;
    return set_err(ctx, DDX_E_UNSUPPORTED, "device HVG prologue not built yet");
}

int stage_select_columns(ddx_ctx* ctx, const int64_t* cols, int32_t n_cols) {
    (void)cols; (void)n_cols;
    return set_err(ctx, DDX_E_UNSUPPORTED, "device HVG prologue not built yet");
}

}  // namespace ddx
