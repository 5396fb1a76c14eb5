// agt_preproc.hip -- frame pre-processing of the reference's per-frame loop, on the device
// (SURVEY.md section 8f rank 1):
//   PoseDetector.undistort_frame  /root/reference/aprilgroup_tracking/aprilgroup_pose_estimation/detect_pose.py:147-183
//       cv.getOptimalNewCameraMatrix (host, agt_get_optimal_new_camera_matrix)
//       cv.undistort = initUndistortRectifyMap(CV_16SC2) + remap(INTER_LINEAR, BORDER_CONSTANT) + ROI crop
//   cv.cvtColor(frame, COLOR_BGR2GRAY)   detect_pose.py:602
// Semantics: OpenCV undistort.dispatch.cpp / imgwarp.cpp / color_rgb, restated in oracle/cv_imgproc.c;
// maps and images are bit-identical to that oracle.
//
// Data flow: the undistortion map depends on the camera only, so it is built ONCE per camera
// (6 B/pixel: int16 x, int16 y, 5+5 fraction bits) and stays L2/Infinity-Cache resident across
// frames and streams.  Per frame one fused kernel gathers the four BGR taps, interpolates in
// 15-bit fixed point per channel (exactly cv.undistort's rounding), converts to gray
// (14-bit fixed point, exactly cv.cvtColor's) and writes only the ROI: 3 B/px in, 1 B/px out,
// no intermediate BGR image in HBM.
#include "agt_device.h"
#include "agt_kernels.h"

namespace {

constexpr int INTER_BITS = 5, INTER_TAB = 32;

struct MapParams {
    double ir[9];                 // inverse of the new camera matrix
    double fx, fy, u0, v0;        // original camera
    double k[12];                 // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double tilt[9];               // matTilt (tilted sensor, coefficients 13 / 14); read when has_tilt
    int has_tilt, rsv_;
    int w, h;
    short2* map1;
    unsigned short* map2;
};

// cv::initUndistortRectifyMap, R = I, m1type = CV_16SC2 (FP64 exactly as the oracle: no contraction)
__global__ __launch_bounds__(256) void undistort_map_kernel(const MapParams P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= P.w) return;
    const double* ir = P.ir;
    const double _x = j * ir[0] + (i * ir[1] + ir[2]), _y = j * ir[3] + (i * ir[4] + ir[5]), _w = j * ir[6] + (i * ir[7] + ir[8]);
    const double ww = 1. / _w, x = _x * ww, y = _y * ww;
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
    const double k1 = P.k[0], k2 = P.k[1], p1 = P.k[2], p2 = P.k[3], k3 = P.k[4], k4 = P.k[5], k5 = P.k[6], k6 = P.k[7];
    const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
    const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + P.k[8] * r2 + P.k[9] * r2 * r2);
    const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + P.k[10] * r2 + P.k[11] * r2 * r2);
    double u = P.fx * xd + P.u0, v = P.fy * yd + P.v0;
    if (P.has_tilt) {
        // vecTilt = matTilt (xd, yd, 1); u = fx * invProj * vecTilt(0) + u0 (initUndistortRectifyMap's own order of operations)
        double t[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { double a = 0.0; a += P.tilt[r * 3] * xd; a += P.tilt[r * 3 + 1] * yd; a += P.tilt[r * 3 + 2] * 1.0; t[r] = a; }
        const double ip = t[2] != 0.0 ? 1. / t[2] : 1.0;
        u = P.fx * ip * t[0] + P.u0; v = P.fy * ip * t[1] + P.v0;
    }
    const double us = u * INTER_TAB, vs = v * INTER_TAB;
    const int iu = us >= 2147483647. ? 2147483647 : us <= -2147483648. ? (-2147483647 - 1) : __double2int_rn(us);
    const int iv = vs >= 2147483647. ? 2147483647 : vs <= -2147483648. ? (-2147483647 - 1) : __double2int_rn(vs);
    const long o = (long)i * P.w + j;
    P.map1[o] = make_short2((short)(iu >> INTER_BITS), (short)(iv >> INTER_BITS));
    P.map2[o] = (unsigned short)((iv & (INTER_TAB - 1)) * INTER_TAB + (iu & (INTER_TAB - 1)));
}

struct RemapParams {
    const uint8_t* src; long spitch, sbatch; int sw, sh;
    const short2* map1; const unsigned short* map2; int mw;      // maps cover mw x (any) pixels
    int rx, ry, rw, rh;                                          // output window in map coordinates
    uint8_t* dst; long dpitch, dbatch;
    int undistort;                                               // 0: taps come straight from (x, y)
    int B;                                                       // images (gray path: flattened work order)
    int xshift;                                                  // log2 of the XCD count the gray path's band order is laid out for
};

struct __attribute__((packed)) Tap6 { uint32_t lo; uint16_t hi; };

// every product below has operands under 2^24 (pixels 8 bit, weights <= 2^15): full-rate 24-bit multiplies
__device__ __forceinline__ int gray14(int b, int g, int r) { return (__mul24(b, 1868) + __mul24(g, 9617) + __mul24(r, 4899) + (1 << 13)) >> 14; }
__device__ __forceinline__ int blend4(int a, int b, int c, int d, int w00, int w01, int w10, int w11)
{
    return (__mul24(a, w00) + __mul24(b, w01) + __mul24(c, w10) + __mul24(d, w11) + (1 << 14)) >> 15;
}
__device__ __forceinline__ void tap_weights(int q, int& w00, int& w01, int& w10, int& w11)
{
    const int fx = q & 31, fy = (q >> 5) & 31;
    w00 = __mul24(32 - fx, 32 - fy) << 5; w01 = __mul24(fx, 32 - fy) << 5; w10 = __mul24(32 - fx, fy) << 5; w11 = __mul24(fx, fy) << 5;
}

// one BGR pixel of cv.remap(INTER_LINEAR, BORDER_CONSTANT = 0) at map entry (sx, sy, fraction q)
__device__ __forceinline__ void remap_bgr(const uint8_t* __restrict__ img, long spitch, int sw, int sh,
                                          int sx, int sy, int q, int& b, int& g, int& r)
{
    int w00, w01, w10, w11;
    tap_weights(q, w00, w01, w10, w11);
    if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
        // the two BGR taps of a row are 6 contiguous bytes at an arbitrary byte offset: one unaligned
        // dword + one unaligned ushort (gfx950 global loads need no alignment) instead of six byte loads
        const Tap6 s = *reinterpret_cast<const Tap6*>(img + (long)sy * spitch + sx * 3);
        const Tap6 t = *reinterpret_cast<const Tap6*>(img + (long)(sy + 1) * spitch + sx * 3);
        const int s0 = s.lo & 0xff, s1 = (s.lo >> 8) & 0xff, s2 = (s.lo >> 16) & 0xff, s3 = s.lo >> 24, s4 = s.hi & 0xff, s5 = s.hi >> 8;
        const int t0 = t.lo & 0xff, t1 = (t.lo >> 8) & 0xff, t2 = (t.lo >> 16) & 0xff, t3 = t.lo >> 24, t4 = t.hi & 0xff, t5 = t.hi >> 8;
        b = blend4(s0, s3, t0, t3, w00, w01, w10, w11);
        g = blend4(s1, s4, t1, t4, w00, w01, w10, w11);
        r = blend4(s2, s5, t2, t5, w00, w01, w10, w11);
        return;
    }
    b = g = r = 0;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return;
    const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw, y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
    int acc[3] = { 1 << 14, 1 << 14, 1 << 14 };
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (x0 && y0) acc[c] += img[(long)sy * spitch + sx * 3 + c] * w00;
        if (x1 && y0) acc[c] += img[(long)sy * spitch + (sx + 1) * 3 + c] * w01;
        if (x0 && y1) acc[c] += img[(long)(sy + 1) * spitch + sx * 3 + c] * w10;
        if (x1 && y1) acc[c] += img[(long)(sy + 1) * spitch + (sx + 1) * 3 + c] * w11;
    }
    b = acc[0] >> 15; g = acc[1] >> 15; r = acc[2] >> 15;
}

// GRAY: undistort (optional) + cvtColor + crop, four output pixels per thread (one dword store).
// !GRAY: cv.undistort to BGR, one pixel per thread.
template <bool GRAY>
__global__ __launch_bounds__(256) void preprocess_kernel(const RemapParams P)
{
    if (GRAY) {
        // A workgroup = 4 rows x 64 groups of four output pixels.  Grid (X * 2 * nseg, images, bands / X), X = 2^xshift XCDs: gridDim.x
        // is a multiple of X and workgroups are dealt round-robin to the X XCDs in x-fastest order, so
        // blockIdx.x % X IS the XCD: XCD k owns the bands k * gridDim.z .. (k + 1) * gridDim.z - 1 (8 rows each),
        // and walks band by band, image by image.  Inside a band the map rows (6 B/px, shared by all images) stay
        // in that XCD's L2 while the images go by, and the source rows that consecutive output rows share
        // (sy and sy + 1) are fetched into one L2 instead of two.  No divisions: the split is done by the grid.
        const int nseg = (int)gridDim.x >> (P.xshift + 1);
        const int xcd = blockIdx.x & ((1 << P.xshift) - 1), j = (int)blockIdx.x >> P.xshift;
        const int strip = j >= nseg ? 1 : 0, seg = j - strip * nseg;
        const int band = xcd * (int)gridDim.z + (int)blockIdx.z;
        const int bz = blockIdx.y;
        const int y = band * 8 + strip * 4 + ((int)threadIdx.x >> 6);          // row inside the ROI
        const int x0 = (seg * 64 + ((int)threadIdx.x & 63)) * 4;
        if (y >= P.rh || x0 >= P.rw) return;
        const uint8_t* img = P.src + (long)bz * P.sbatch;
        uint8_t* out = P.dst + (long)bz * P.dbatch;
        uint32_t packed = 0;
        // map entries of the thread's four pixels: 16 + 8 contiguous bytes (unaligned vector loads)
        struct __attribute__((packed)) M1x4 { short2 m[4]; };
        struct __attribute__((packed)) M2x4 { unsigned short q[4]; };
        M1x4 mm; M2x4 qq;
        const long o0 = (long)(P.ry + y) * P.mw + (P.rx + x0);
        if (P.undistort) {
            if (x0 + 3 < P.rw) { mm = *reinterpret_cast<const M1x4*>(P.map1 + o0); qq = *reinterpret_cast<const M2x4*>(P.map2 + o0); }
            else for (int k = 0; k < 4; k++) { const bool in = x0 + k < P.rw; mm.m[k] = in ? P.map1[o0 + k] : make_short2(0, 0); qq.q[k] = in ? P.map2[o0 + k] : 0; }
        }
        if (P.undistort && x0 + 3 < P.rw) {
            // Interior path: all four map entries inside the source.  Each pixel's two taps per row are 6 contiguous
            // bytes at an arbitrary byte offset: one unaligned 8-byte load per pixel per row (gfx950 global loads need
            // no alignment), all eight requested before the first is used.  (A variant that served "regular" groups
            // -- sx advancing by exactly one -- with two 16-byte loads made the kernel slower: lens + zoom maps mix
            // regular and irregular groups inside every wave, so both paths ran.)
            bool interior = true;
            // (sx <= sw - 4 keeps the 12 aligned bytes of a tap inside its row; an unaligned frame base also needs
            // the bytes just before the tap to exist)
            const bool base4 = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)P.spitch) & 3) == 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                interior = interior && (unsigned)mm.m[k].x < (unsigned)(P.sw - 3) && (unsigned)mm.m[k].y < (unsigned)(P.sh - 1) &&
                           (base4 || mm.m[k].x > 1 || mm.m[k].y > 0);
            if (interior) {
                // The 6 tap bytes of a row start at an arbitrary byte offset o.  Misaligned vector loads go through the
                // texture addresser lane by lane; instead the three ALIGNED dwords that cover [o, o + 6) are loaded
                // (dwordx3, 4-byte aligned) and funnel-shifted by o & 3 bytes (v_alignbyte_b32).
                struct Row12 { uint32_t d0, d1, d2; };
                Row12 s[4], t[4];
                unsigned sh[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned o = (unsigned)(__mul24((int)mm.m[k].y, (int)P.spitch) + __mul24((int)mm.m[k].x, 3));   // < 2^31: one frame
                    const uintptr_t a = reinterpret_cast<uintptr_t>(img) + o;
                    sh[k] = (unsigned)(a & 3);
                    const uint8_t* al = reinterpret_cast<const uint8_t*>(a & ~(uintptr_t)3);
                    s[k] = *reinterpret_cast<const Row12*>(al);
                    t[k] = *reinterpret_cast<const Row12*>(al + P.spitch);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int w00, w01, w10, w11;
                    tap_weights(qq.q[k], w00, w01, w10, w11);
                    const uint32_t slo = __builtin_amdgcn_alignbyte(s[k].d1, s[k].d0, sh[k]), shi = __builtin_amdgcn_alignbyte(s[k].d2, s[k].d1, sh[k]);
                    const uint32_t tlo = __builtin_amdgcn_alignbyte(t[k].d1, t[k].d0, sh[k]), thi = __builtin_amdgcn_alignbyte(t[k].d2, t[k].d1, sh[k]);
                    const int b = blend4(slo & 0xff, slo >> 24, tlo & 0xff, tlo >> 24, w00, w01, w10, w11);
                    const int g = blend4((slo >> 8) & 0xff, shi & 0xff, (tlo >> 8) & 0xff, thi & 0xff, w00, w01, w10, w11);
                    const int r = blend4((slo >> 16) & 0xff, (shi >> 8) & 0xff, (tlo >> 16) & 0xff, (thi >> 8) & 0xff, w00, w01, w10, w11);
                    packed |= (uint32_t)gray14(b, g, r) << (8 * k);
                }
                uint8_t* o = out + (long)y * P.dpitch + x0;
                if ((((uintptr_t)o) & 3) == 0) *reinterpret_cast<uint32_t*>(o) = packed;
                else for (int k = 0; k < 4; k++) o[k] = (uint8_t)(packed >> (8 * k));
                return;
            }
        }
        if (!P.undistort && x0 + 3 < P.rw) {
            // gray + crop only: the four BGR pixels are 12 contiguous bytes
            struct __attribute__((packed)) Px4 { uint32_t d[3]; };
            const Px4 v = *reinterpret_cast<const Px4*>(img + (long)(P.ry + y) * P.spitch + (P.rx + x0) * 3);
            auto byte = [](const Px4& q, int i) -> int { return (q.d[i >> 2] >> (8 * (i & 3))) & 0xff; };
#pragma unroll
            for (int k = 0; k < 4; k++) packed |= (uint32_t)gray14(byte(v, 3 * k), byte(v, 3 * k + 1), byte(v, 3 * k + 2)) << (8 * k);
            uint8_t* o = out + (long)y * P.dpitch + x0;
            if ((((uintptr_t)o) & 3) == 0) *reinterpret_cast<uint32_t*>(o) = packed;
            else for (int k = 0; k < 4; k++) o[k] = (uint8_t)(packed >> (8 * k));
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = x0 + k;
            if (x < P.rw) {
                int b, g, r;
                if (P.undistort) {
                    remap_bgr(img, P.spitch, P.sw, P.sh, mm.m[k].x, mm.m[k].y, qq.q[k], b, g, r);
                } else {
                    const uint8_t* S = img + (long)(P.ry + y) * P.spitch + (P.rx + x) * 3;
                    b = S[0]; g = S[1]; r = S[2];
                }
                packed |= (uint32_t)gray14(b, g, r) << (8 * k);
            }
        }
        uint8_t* o = out + (long)y * P.dpitch + x0;
        if (x0 + 3 < P.rw && (((uintptr_t)o) & 3) == 0) *reinterpret_cast<uint32_t*>(o) = packed;
        else for (int k = 0; k < 4 && x0 + k < P.rw; k++) o[k] = (uint8_t)(packed >> (8 * k));
    } else {
        const uint8_t* img = P.src + (long)blockIdx.z * P.sbatch;
        uint8_t* out = P.dst + (long)blockIdx.z * P.dbatch;
        const int y = blockIdx.y;
        const int x = blockIdx.x * blockDim.x + threadIdx.x;
        if (x >= P.rw) return;
        const long o = (long)(P.ry + y) * P.mw + (P.rx + x);
        const short2 m = P.map1[o];
        int b, g, r;
        remap_bgr(img, P.spitch, P.sw, P.sh, m.x, m.y, P.map2[o], b, g, r);
        uint8_t* d = out + (long)y * P.dpitch + x * 3;
        d[0] = (uint8_t)b; d[1] = (uint8_t)g; d[2] = (uint8_t)r;
    }
}

}  // namespace

hipError_t agt_launch_undistort_map(hipStream_t stream, const double* K, const AgtCameraHost& cam, const AgtTiltHost& tilt, const double* ir,
                                    int w, int h, short2* map1, unsigned short* map2)
{
    MapParams P;
    for (int i = 0; i < 9; i++) P.ir[i] = ir[i];
    P.fx = K[0]; P.fy = K[4]; P.u0 = K[2]; P.v0 = K[5];
    for (int i = 0; i < 12; i++) P.k[i] = cam.k[i];
    for (int i = 0; i < 9; i++) P.tilt[i] = tilt.m[i];
    P.has_tilt = tilt.on; P.rsv_ = 0;
    P.w = w; P.h = h; P.map1 = map1; P.map2 = map2;
    hipLaunchKernelGGL(undistort_map_kernel, dim3((w + 255) / 256, h), dim3(256), 0, stream, P);
    return hipGetLastError();
}

hipError_t agt_launch_preprocess(hipStream_t stream, const uint8_t* src, long spitch, long sbatch, int sw, int sh,
                                 const short2* map1, const unsigned short* map2, int mw,
                                 int rx, int ry, int rw, int rh, uint8_t* dst, long dpitch, long dbatch,
                                 int undistort, int gray, int B)
{
    RemapParams P;
    P.src = src; P.spitch = spitch; P.sbatch = sbatch; P.sw = sw; P.sh = sh;
    P.map1 = map1; P.map2 = map2; P.mw = mw; P.rx = rx; P.ry = ry; P.rw = rw; P.rh = rh;
    P.dst = dst; P.dpitch = dpitch; P.dbatch = dbatch; P.undistort = undistort;
    P.B = B;
    P.xshift = agt_chip_current().xshift;
    const int nseg = (rw + 255) / 256, bands = (rh + 7) / 8, X = 1 << P.xshift;
    if (gray) hipLaunchKernelGGL(preprocess_kernel<true>, dim3(2 * X * nseg, B, (bands + X - 1) / X), dim3(256), 0, stream, P);
    else hipLaunchKernelGGL(preprocess_kernel<false>, dim3((rw + 255) / 256, rh, B), dim3(256), 0, stream, P);
    return hipGetLastError();
}
