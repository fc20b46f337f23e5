// placeholder: filled in by the pipeline milestone
#include "ddk_internal.h"
namespace ddk {
int model_finalize(ddk_ctx* ctx) { (void)ctx; return DDK_OK; }
void model_destroy(ddk_ctx* ctx) { (void)ctx; }
}
using namespace ddk;
extern "C" {
int ddk_complex_create(ddk_ctx* ctx, const ddk_complex_desc*, int32_t, ddk_complex**) { return fail(ctx, DDK_ERR_STATE, "not implemented"); }
void ddk_complex_destroy(ddk_ctx*, ddk_complex*) {}
int ddk_score_forward(ddk_ctx* ctx, ddk_complex*, int32_t, const float*, float, float, float, float*, float*, float*, void*) { return fail(ctx, DDK_ERR_STATE, "not implemented"); }
int ddk_se3_update(ddk_ctx* ctx, ddk_complex*, int32_t, const float*, const float*, const float*, const float*, float*, void*) { return fail(ctx, DDK_ERR_STATE, "not implemented"); }
int ddk_sample(ddk_ctx* ctx, ddk_complex*, int32_t, int32_t, const float*, const float*, const float*, const float*, float*, void*) { return fail(ctx, DDK_ERR_STATE, "not implemented"); }
int ddk_last_graph_stats(ddk_ctx* ctx, ddk_complex*, int64_t*, void*) { return fail(ctx, DDK_ERR_STATE, "not implemented"); }
int ddk_last_node_features(ddk_ctx* ctx, ddk_complex*, int32_t, float*, float*, void*) { return fail(ctx, DDK_ERR_STATE, "not implemented"); }
}
