// wave_reduce.h — sums of several per-lane fp64 values across the 64 lanes of a wave, shared by the tracker kernels
// (stage_b.hip) and ExtRotVel's normal equations (stage_c.hip).
#pragma once
#include "ctx.h"

namespace edgehip {

// ---------------------------------------------------------------------------------------------------
// Sum 28 per-lane values across the 64 lanes of a wave in 29 pair exchanges instead of 28 * 6: at every butterfly step a
// lane keeps one half of its values and hands the other half to its partner, so the number of live values halves
// (28 -> 14 -> 7 -> 4 -> 2 -> 1) while the partial sums double in coverage.  The pairs and their order are those of the plain
// xor butterfly (lane ^ 32, 16, 8, 4, 2, 1), so the sums are the butterfly's bit for bit.  On return v[0] of lane l is the
// full sum of value
//     idx = b1 + 2*b2 + 4*b3 + 7*b4 + 14*b5      (b_k = bit k of l; lanes with b1+2*b2+4*b3 == 7 hold padding)
// and lanes l, l^1 hold the same value.  Fixed order => bit-reproducible from run to run.
// The exchanges themselves: v_permlane32_swap / v_permlane16_swap (gfx950) for the two big steps, DPP moves for lane ^ 8, 4, 2, 1
// (no ds_bpermute: a shuffle through the LDS crossbar costs an LDS instruction each way and was most of the reduction's time when
// every step used it).  The DPP moves carry bound_ctrl: every lane has a source, so the flag changes no value — it tells the compiler
// that the destination's old content is dead, which saves the `v_mov_b32 v, 0` it otherwise puts in front of every one of them.
// ---------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void halve_step(double *v, int lane, int off) {
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double send = hi ? v[i] : v[i + N];
        const double keep = hi ? v[i + N] : v[i];
        v[i] = keep + __shfl_xor(send, off, 64);
    }
}
// The two big steps (partner = lane ^ 32, lane ^ 16) with gfx950's v_permlane32_swap / v_permlane16_swap: swapping the upper
// half (odd rows) of v[i] with the lower half (even rows) of v[i + N] leaves, in every lane, exactly the two operands the
// halving step adds — own and partner's v[i] below, own and partner's v[i + N] above — so a pair costs two swaps (one per
// dword) and the add instead of four selects, two ds_bpermute and the add.  Same pairs, same sums, bit for bit.
template <int N, bool ROWS16>
__device__ __forceinline__ void halve_swap(double *v) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        const unsigned alo = (unsigned)__double2loint(v[i]), ahi = (unsigned)__double2hiint(v[i]);
        const unsigned blo = (unsigned)__double2loint(v[i + N]), bhi = (unsigned)__double2hiint(v[i + N]);
        const auto rl = ROWS16 ? __builtin_amdgcn_permlane16_swap(alo, blo, false, false) : __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto rh = ROWS16 ? __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false) : __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        v[i] = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
}
// partner = lane ^ 8 is a rotation by 8 inside a row of 16, lane ^ 2 / lane ^ 1 are quad permutations: DPP moves
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N, int CTRL>
__device__ __forceinline__ void halve_dpp(double *v, int lane, int off) {
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double send = hi ? v[i] : v[i + N];
        const double keep = hi ? v[i + N] : v[i];
        v[i] = keep + dpp_mov_f64<CTRL>(send);
    }
}
// partner = lane ^ 4: the lower four lanes of every group of eight read four lanes up (row_shl:4), the upper four read four lanes down
// (row_shr:4) — two DPP moves with complementary bank masks, and because each move names its own source register the "send" select of the
// halving step disappears: lower lanes receive the partner's v[i], upper lanes the partner's v[i + N].  Same pairs, same sums as the
// ds_bpermute form it replaces (halve_step), without the LDS round trip in the reduction's tail.
#ifndef EDGEHIP_REDUCE_DPP4
#define EDGEHIP_REDUCE_DPP4 1
#endif
__device__ __forceinline__ int dpp_xor4(int from_lower_rule, int from_upper_rule) {
    const int r = __builtin_amdgcn_update_dpp(0, from_lower_rule, 0x104, 0xf, 0xf, true);   // row_shl:4: lane l <- l + 4 (the upper banks are overwritten next)
    return __builtin_amdgcn_update_dpp(r, from_upper_rule, 0x114, 0xf, 0xa, false);          // row_shr:4 into banks 1 and 3: lane l <- l - 4
}
template <int N>
__device__ __forceinline__ void halve_dpp4(double *v, int lane) {
#if EDGEHIP_REDUCE_DPP4
    const bool hi = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double keep = hi ? v[i + N] : v[i];
        const int lo = dpp_xor4(__double2loint(v[i]), __double2loint(v[i + N]));
        const int hw = dpp_xor4(__double2hiint(v[i]), __double2hiint(v[i + N]));
        v[i] = keep + __hiloint2double(hw, lo);
    }
#else
    halve_step<N>(v, lane, 4);
#endif
}
__device__ __forceinline__ int wave_reduce28(double (&s)[kNumSums], int lane) {
    double v[32];
#pragma unroll
    for (int i = 0; i < kNumSums; i++) v[i] = s[i];
    halve_swap<14, false>(v);      // 28 -> 14   (lane ^ 32)
    halve_swap<7, true>(v);        // 14 -> 7    (lane ^ 16)
    v[7] = 0.0;
    halve_dpp<4, 0x128>(v, lane, 8);   // 8 -> 4   row_ror:8
    halve_dpp4<2>(v, lane);            // 4 -> 2   row_shl:4 / row_shr:4 by bank
    halve_dpp<1, 0x4E>(v, lane, 2);    // 2 -> 1   quad_perm:[2,3,0,1]
    v[0] += dpp_mov_f64<0xB1>(v[0]);   //          quad_perm:[1,0,3,2]
    s[0] = v[0];
    const int b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
    const int low = b1 + 2 * b2 + 4 * b3;
    return low == 7 ? 31 : low + 7 * b4 + 14 * b5;
}

// ---- the same halving reduction for 28 fp32 values (the float tracker, Minimizer_RV<float>): one 32-bit swap / DPP move per value ----
template <int N, bool ROWS16>
__device__ __forceinline__ void halve_swap_f32(float *v) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        const unsigned a = __float_as_uint(v[i]), b = __float_as_uint(v[i + N]);
        const auto r = ROWS16 ? __builtin_amdgcn_permlane16_swap(a, b, false, false) : __builtin_amdgcn_permlane32_swap(a, b, false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int N, int CTRL>
__device__ __forceinline__ void halve_dpp_f32(float *v, int lane, int off) {
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float send = hi ? v[i] : v[i + N];
        const float keep = hi ? v[i + N] : v[i];
        v[i] = keep + dpp_mov_f32<CTRL>(send);
    }
}
__device__ __forceinline__ int wave_reduce28_f32(float (&s)[kNumSums], int lane) {
    float v[32];
#pragma unroll
    for (int i = 0; i < kNumSums; i++) v[i] = s[i];
    halve_swap_f32<14, false>(v);
    halve_swap_f32<7, true>(v);
    v[7] = 0.f;
    halve_dpp_f32<4, 0x128>(v, lane, 8);
    {   // lane ^ 4: row_shl:4 / row_shr:4 by bank (halve_dpp4)
        const bool hi = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float keep = hi ? v[i + 2] : v[i];
#if EDGEHIP_REDUCE_DPP4
            v[i] = keep + __int_as_float(dpp_xor4(__float_as_int(v[i]), __float_as_int(v[i + 2])));
#else
            const float send = hi ? v[i] : v[i + 2];
            v[i] = keep + __shfl_xor(send, 4, 64);
#endif
        }
    }
    halve_dpp_f32<1, 0x4E>(v, lane, 2);
    v[0] += dpp_mov_f32<0xB1>(v[0]);
    s[0] = v[0];
    const int b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
    const int low = b1 + 2 * b2 + 4 * b3;
    return low == 7 ? 31 : low + 7 * b4 + 14 * b5;
}

// The same for up to 16 values (the 3-DoF tracker's 10): 16 -> 8 -> 4 -> 2 -> 1 over lane ^ 32, 16, 8, 4, then the two
// remaining butterfly steps (lane ^ 2, lane ^ 1) on the single value.  On return v[0] of lane l is the full sum of value
//     idx = 8*b5 + 4*b4 + 2*b3 + b2      (b_k = bit k of l), the same in all four lanes of a quad.
// Pairs and order are those of the plain xor butterfly (32, 16, 8, 4, 2, 1): bit-identical sums.
__device__ __forceinline__ int wave_reduce16(double (&v)[16], int lane) {
    halve_swap<8, false>(v);           // 16 -> 8   (lane ^ 32)
    halve_swap<4, true>(v);            // 8 -> 4    (lane ^ 16)
    halve_dpp<2, 0x128>(v, lane, 8);   // 4 -> 2    row_ror:8
    halve_dpp4<1>(v, lane);            // 2 -> 1    row_shl:4 / row_shr:4 by bank
    v[0] += dpp_mov_f64<0x4E>(v[0]);   // lane ^ 2
    v[0] += dpp_mov_f64<0xB1>(v[0]);   // lane ^ 1
    return 8 * ((lane >> 5) & 1) + 4 * ((lane >> 4) & 1) + 2 * ((lane >> 3) & 1) + ((lane >> 2) & 1);
}

}  // namespace edgehip
