// stage_c.hip — matching + mapping (placeholder)
#include "ctx.h"
namespace edgehip {
int forward_match_enqueue(edgehip_ctx *, int, int) { return EDGEHIP_ERR_STATE; }
int rotate_enqueue(edgehip_ctx *, int, const double *) { return EDGEHIP_ERR_STATE; }
int directed_enqueue(edgehip_ctx *, int, int) { return EDGEHIP_ERR_STATE; }
int regekf_enqueue(edgehip_ctx *, int, int, int) { return EDGEHIP_ERR_STATE; }
int rescale_enqueue(edgehip_ctx *, int) { return EDGEHIP_ERR_STATE; }
int pose_enqueue(edgehip_ctx *, int, const double *) { return EDGEHIP_ERR_STATE; }
}
using namespace edgehip;
extern "C" {
int edgehip_forward_match(edgehip_ctx *c, int a, int b) { return forward_match_enqueue(c, a, b); }
int edgehip_rotate_keylines(edgehip_ctx *c, int s, const double *R) { return rotate_enqueue(c, s, R); }
int edgehip_directed_matching(edgehip_ctx *c, int a, int b) { return directed_enqueue(c, a, b); }
int edgehip_regularize_ekf(edgehip_ctx *c, int s, int r, int e) { return regekf_enqueue(c, s, r, e); }
int edgehip_rescale(edgehip_ctx *c, int s) { return rescale_enqueue(c, s); }
int edgehip_process_frame(edgehip_ctx *, const double *) { return EDGEHIP_ERR_STATE; }
int edgehip_next_slot(edgehip_ctx *c) { return c ? (c->frame_slot + 1) % c->plan.nslots : -1; }
int edgehip_cur_slot(edgehip_ctx *c) { return c ? c->frame_slot : -1; }
int edgehip_read_nav(edgehip_ctx *, edgehip_nav *) { return EDGEHIP_ERR_STATE; }
int edgehip_reset(edgehip_ctx *) { return EDGEHIP_ERR_STATE; }
}
