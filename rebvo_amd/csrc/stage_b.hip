// stage_b.hip — tracker: EstimateQuantile, build_field, TryVelRot, Minimizer_RV (placeholder, see below)
#include "ctx.h"
namespace edgehip {
int quantile_enqueue(edgehip_ctx *, int, double, double, double, int) { set_error("not implemented"); return EDGEHIP_ERR_STATE; }
int build_field_enqueue(edgehip_ctx *, int, int, float) { set_error("not implemented"); return EDGEHIP_ERR_STATE; }
int tvr_prepare_enqueue(edgehip_ctx *, int) { set_error("not implemented"); return EDGEHIP_ERR_STATE; }
int minimizer_enqueue(edgehip_ctx *, int, int) { set_error("not implemented"); return EDGEHIP_ERR_STATE; }
}
using namespace edgehip;
extern "C" {
int edgehip_quantile(edgehip_ctx *c, int slot, double a, double b, double p, int n) { return quantile_enqueue(c, slot, a, b, p, n); }
int edgehip_build_field(edgehip_ctx *c, int slot, int r, float m) { return build_field_enqueue(c, slot, r, m); }
int edgehip_try_velrot(edgehip_ctx *, int, int, const double *, int, int, double, const double *, uint32_t, double, int, int, double *) { return EDGEHIP_ERR_STATE; }
int edgehip_download_resid(edgehip_ctx *, int, double *) { return EDGEHIP_ERR_STATE; }
int edgehip_minimizer_rv(edgehip_ctx *c, int a, int b) { return minimizer_enqueue(c, a, b); }
}
