// agt_pyramid_body.h -- device body of cv::pyrDown (u8, 5x5 [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101).
// Included by agt_pyramid.hip (stand-alone kernel) and agt_step.hip (fused per-frame launch).
// Semantics: OpenCV modules/imgproc/src/pyramids.cpp pyrDown_, restated in oracle/cv_lk.c cvo_pyr_down_u8.
// HBM-bound integer work: each source byte is read once with aligned dword loads (all of a
// thread's loads issued before any is consumed), the (2*TH+3) x (2*TW+8) source tile is staged
// in LDS, filtered horizontally into a u16 LDS plane, then vertically, and written as packed dwords.
#pragma once
#include "agt_device.h"
#include "agt_kernels.h"

namespace agt_pyr {

constexpr int TW = 128;             // output tile width  (pixels)
constexpr int TH = 16;              // output tile height
constexpr int SW = 2 * TW + 8;      // staged source bytes per row: x in [2*ox0-4, 2*ox0+2*TW+4)
constexpr int SH = 2 * TH + 3;      // staged source rows:          y in [2*oy0-2, 2*oy0+2*TH+1)
constexpr int NT = 256;

constexpr int PYR_LDS_BYTES = SH * SW + SH * TW * 2;     // staged source tile + u16 horizontal plane

// one 128x16 output tile; (bx, by, bz) = tile x, tile y, image index.  lds: PYR_LDS_BYTES, 16-B aligned
__device__ __forceinline__ void pyr_down_body(const AgtPyrArgs& A, int bx, int by, int bz, uint8_t* lds)
{
    uint8_t* s_src = lds;
    uint16_t* s_h = reinterpret_cast<uint16_t*>(lds + SH * SW);
    const uint8_t* __restrict__ src = A.src;
    uint8_t* __restrict__ dst = A.dst;
    const int sw = A.sw, sh = A.sh, dw = A.dw, dh = A.dh;
    const long spitch = A.spitch, dpitch = A.dpitch;

    const int tid = threadIdx.x;
    const int ox0 = bx * TW, oy0 = by * TH;
    const uint8_t* img = src + (long)bz * A.sbatch;
    uint8_t* out = dst + (long)bz * A.dbatch;
    const int sx0 = 2 * ox0 - 4, sy0 = 2 * oy0 - 2;

    // ---- stage the source tile: all of a thread's loads are issued before any is consumed
    // (aligned dwords; per-byte reflect only at the image edge)
    constexpr int NLD = (SH * (SW / 4) + NT - 1) / NT;
    uint32_t regs[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int i = tid + k * NT;
        uint32_t v = 0;
        if (i < SH * (SW / 4)) {
            const int r = i / (SW / 4), c4 = i - r * (SW / 4);
            const int gy = agt_reflect101(sy0 + r, sh);
            const int gx = sx0 + 4 * c4;
            const uint8_t* row = img + (long)gy * spitch;
            if (gx >= 0 && gx + 3 < sw) {
                v = *reinterpret_cast<const uint32_t*>(row + gx);
            } else {
                v = (uint32_t)row[agt_reflect101(gx, sw)] | ((uint32_t)row[agt_reflect101(gx + 1, sw)] << 8) |
                    ((uint32_t)row[agt_reflect101(gx + 2, sw)] << 16) | ((uint32_t)row[agt_reflect101(gx + 3, sw)] << 24);
            }
        }
        regs[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int i = tid + k * NT;
        if (i < SH * (SW / 4)) *reinterpret_cast<uint32_t*>(&s_src[4 * i]) = regs[k];
    }
    __syncthreads();

    // ---- horizontal [1 4 6 4 1]: two adjacent outputs per thread from one 8-byte window
    for (int i = tid; i < SH * (TW / 2); i += NT) {
        const int r = i / (TW / 2), p = i - r * (TW / 2);
        // outputs ox = 2p, 2p+1 -> centres at staged byte 4p+4 and 4p+6; window bytes [4p+2, 4p+9)
        const uint8_t* s = &s_src[r * SW + 4 * p];
        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(s);       // bytes 0..3
        const uint32_t w1 = *reinterpret_cast<const uint32_t*>(s + 4);   // bytes 4..7
        const uint32_t w2 = *reinterpret_cast<const uint32_t*>(s + 8);   // bytes 8..11
        const int b2 = (w0 >> 16) & 0xff, b3 = w0 >> 24;
        const int b4 = w1 & 0xff, b5 = (w1 >> 8) & 0xff, b6 = (w1 >> 16) & 0xff, b7 = w1 >> 24;
        const int b8 = w2 & 0xff;
        const int h0 = b4 * 6 + (b3 + b5) * 4 + b2 + b6;
        const int h1 = b6 * 6 + (b5 + b7) * 4 + b4 + b8;
        *reinterpret_cast<uint32_t*>(&s_h[r * TW + 2 * p]) = (uint32_t)h0 | ((uint32_t)h1 << 16);
    }
    __syncthreads();

    // ---- vertical [1 4 6 4 1] + (v + 128) >> 8, four outputs per thread, dword stores
    for (int i = tid; i < TH * (TW / 4); i += NT) {
        const int oy = i / (TW / 4), q = i - oy * (TW / 4);
        const int gy = oy0 + oy, gx = ox0 + 4 * q;
        if (gy >= dh || gx >= dw) continue;
        const uint16_t* h = &s_h[(2 * oy) * TW + 4 * q];
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = h[2 * TW + k] * 6 + (h[TW + k] + h[3 * TW + k]) * 4 + h[k] + h[4 * TW + k];
            packed |= (uint32_t)((v + 128) >> 8) << (8 * k);
        }
        uint8_t* o = out + (long)gy * dpitch + gx;
        if (gx + 3 < dw) {
            *reinterpret_cast<uint32_t*>(o) = packed;
        } else {
            for (int k = 0; k < 4 && gx + k < dw; k++) o[k] = (uint8_t)(packed >> (8 * k));
        }
    }
}

}  // namespace agt_pyr
