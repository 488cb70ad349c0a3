// net_keypoint.cpp — see rebvo/net_keypoint.h (reference: src/CommLib/net_keypoint.cpp:29-108).
#include "rebvo/net_keypoint.h"

#include <algorithm>
#include <cmath>

namespace rebvo {

namespace {
// util::clamp_uchar / clamp_ushort (include/UtilLib/util.h:52-66): the argument arrives as float
inline uint8_t clamp_uchar(float f) { return f < 0 ? 0 : f > 255.0 ? 255 : (uint8_t)f; }
inline uint16_t clamp_ushort(float f) {
    if (f < 0) return 0;
    if (f > 65535.0) return 65535;
    return (uint16_t)f;
}
}  // namespace

int copy_net_keyline(KeyLine *from, int kn, const KeyLine *from_pair, net_keyline *to, int kl_size, double k_prof) {
    int j = 0;
    for (int i = 0; i < kn; i++) {
        KeyLine &kl = from[i];
        if (j >= kl_size) break;
        to[j].qx = (uint16_t)std::round(kl.c_p.x);
        to[j].qy = (uint16_t)std::round(kl.c_p.y);
        to[j].rho = std::max(clamp_ushort((float)(NET_RHO_SCALING * kl.rho / k_prof)), (uint16_t)1);
        to[j].s_rho = std::max(clamp_ushort((float)(NET_RHO_SCALING * kl.s_rho / k_prof)), (uint16_t)1);
        if (from_pair) {
            if (kl.stereo_m_id >= 0 && std::fabs(std::round(-kl.c_p.x + from_pair[kl.stereo_m_id].c_p.x)) < 127 &&
                std::fabs(std::round(-kl.c_p.y + from_pair[kl.stereo_m_id].c_p.y)) < 127) {
                to[j].extra.flow.x = clamp_uchar((float)std::round((-kl.c_p.x + from_pair[kl.stereo_m_id].c_p.x) + 127.0));
                to[j].extra.flow.y = clamp_uchar((float)std::round((-kl.c_p.y + from_pair[kl.stereo_m_id].c_p.y) + 127.0));
            } else {
                to[j].extra.flow.x = 127;
                to[j].extra.flow.y = 127;
            }
        } else {
            to[j].extra.flow.x = clamp_uchar((float)std::round((kl.p_m.x - kl.p_m_0.x) * 10 + 127.0));
            to[j].extra.flow.y = clamp_uchar((float)std::round((kl.p_m.y - kl.p_m_0.y) * 10 + 127.0));
        }
        to[j].n_kl = -1;
        to[j].m_num = clamp_uchar((float)kl.m_num);
        kl.net_id = j;
        j++;
    }
    return j;
}

int copy_net_keyline_nextid(const KeyLine *from, int kn, net_keyline *to, int kl_size) {
    int j = 0;
    for (int i = 0; i < kn; i++) {
        const KeyLine &kl = from[i];
        if (j >= kl_size) break;   // as in the reference: tested against the PREVIOUS KeyLine's net index
        j = kl.net_id;
        if (j < 0 || j >= kl_size) continue;   // not packed (the reference would index out of bounds here)
        if (kl.n_id >= 0) to[j].n_kl = from[kl.n_id].net_id;
    }
    return j;
}

}  // namespace rebvo

extern "C" {
int rebvo_copy_net_keyline(void *keylines, int kn, const void *keylines_pair, void *out, int kl_size, double k_prof) {
    return rebvo::copy_net_keyline((rebvo::KeyLine *)keylines, kn, (const rebvo::KeyLine *)keylines_pair, (rebvo::net_keyline *)out, kl_size, k_prof);
}
int rebvo_copy_net_keyline_nextid(const void *keylines, int kn, void *out, int kl_size) {
    return rebvo::copy_net_keyline_nextid((const rebvo::KeyLine *)keylines, kn, (rebvo::net_keyline *)out, kl_size);
}
}
