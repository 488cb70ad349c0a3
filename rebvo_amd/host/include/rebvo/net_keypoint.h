// net_keypoint.h — the wire format of an edge map for the reference's visualizer (include/CommLib/net_keypoint.h:38-78)
// and the packer that fills it (src/CommLib/net_keypoint.cpp:29-108).  Packing only: the UDP / TCP transport
// (src/CommLib/udp_port.cpp, edgemap_com.cpp) is device I/O this repository does not rebuild.
#ifndef REBVO_AMD_HOST_NET_KEYPOINT_H
#define REBVO_AMD_HOST_NET_KEYPOINT_H

#include <cstdint>

#include "rebvo/rebvo.h"

namespace rebvo {

constexpr double NET_RHO_SCALING = 10000.0;
constexpr double NET_POS_SCALING = 64.0;

#pragma pack(push, 1)
struct net_keyline {   // 15 bytes
    uint16_t qx, qy;       // rounded image position
    uint16_t rho, s_rho;   // inverse depth and its deviation, scaled by NET_RHO_SCALING / k_prof, at least 1
    int32_t n_kl;          // net index of the next KeyLine on the edge (-1: none)
    uint8_t m_num;
    union {
        struct { uint8_t x, y; } flow;               // matched displacement * 10 + 127 (or stereo disparity + 127)
        struct { unsigned a : 10; unsigned m : 6; } gradient;
    } extra;
};
#pragma pack(pop)
static_assert(sizeof(net_keyline) == 15, "net_keyline wire layout");

// Pack up to kl_size KeyLines of `from` (and, with a stereo pair map, the disparity of the stereo matches); sets every
// packed KeyLine's net_id.  Returns the number packed.
int copy_net_keyline(KeyLine *from, int kn, const KeyLine *from_pair, net_keyline *to, int kl_size, double k_prof);
// Second pass: n_kl = net index of each KeyLine's edge successor.
int copy_net_keyline_nextid(const KeyLine *from, int kn, net_keyline *to, int kl_size);

inline int copy_net_keyline(edge_tracker &from, edge_tracker *from_pair, net_keyline *to, int kl_size, double k_prof) {
    return copy_net_keyline(from.begin(), from.KNum(), from_pair ? from_pair->begin() : nullptr, to, kl_size, k_prof);
}
inline int copy_net_keyline_nextid(edge_tracker &from, net_keyline *to, int kl_size) {
    return copy_net_keyline_nextid(from.begin(), from.KNum(), to, kl_size);
}

}  // namespace rebvo

extern "C" {
/* flat view for non-C++ callers (tests): KeyLine arrays in the reference's 168-byte layout, 15-byte records out */
int rebvo_copy_net_keyline(void *keylines, int kn, const void *keylines_pair, void *out, int kl_size, double k_prof);
int rebvo_copy_net_keyline_nextid(const void *keylines, int kn, void *out, int kl_size);
}
#endif
