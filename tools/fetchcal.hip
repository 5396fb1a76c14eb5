// fetchcal.hip -- calibration of rocprofv3's FETCH_SIZE (and the TCC request counters) for the access widths the LK kernels use.
// MI355X_MICROARCH.md (HBM section) calibrates FETCH_SIZE for 16-B-per-lane streaming reads only (reports 1/2 of the bytes) and says
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The one-wave LK kernel stages
// its tiles with buffer_load_dword (agt_lk_rs_body.h: 24 rows x 7 dwords of the previous image, 40 rows x 11 dwords of the next), so
// its "3.3 x algorithmic" figure of rounds 1-4 was an upper bound.  This program reads KNOWN byte counts from a 1 GiB buffer (four
// times the Infinity Cache) in those patterns, one kernel name per pattern, and prints the bytes each launch must move -- exact
// bytes, bytes in whole 64-B sectors, bytes in whole 128-B lines.  Run it directly behind `rocprofv3 --pmc <counter> --kernel-trace
// --output-format csv --` (one counter set per pass); tools/fetchcal_summary.py puts counters and byte counts side by side.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/fetchcal tools/fetchcal.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <set>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int v4u __attribute__((__vector_size__(16)));
typedef unsigned int v2u __attribute__((__vector_size__(8)));

constexpr long BUF = 1L << 30;          // 1 GiB
constexpr int NT = 256;

// coalesced streams: lane i of the grid reads element i (16 / 8 / 4 bytes), every byte of `bytes` exactly once
__global__ __launch_bounds__(NT) void stream_b128(const uint8_t* p, long bytes, uint32_t* out)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p), 0, (int)bytes, 0x00020000);
    const long n = bytes / 16;
    uint32_t acc = 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i * 16), 0, 0);
        acc += t[0] ^ t[1] ^ t[2] ^ t[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(NT) void stream_b64(const uint8_t* p, long bytes, uint32_t* out)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p), 0, (int)bytes, 0x00020000);
    const long n = bytes / 8;
    uint32_t acc = 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const v2u t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(i * 8), 0, 0);
        acc += t[0] ^ t[1];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(NT) void stream_b32(const uint8_t* p, long bytes, uint32_t* out)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p), 0, (int)bytes, 0x00020000);
    const long n = bytes / 4;
    uint32_t acc = 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT)
        acc += __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(i * 4), 0, 0);
    if (acc == 0x12345678u) out[0] = acc;
}

// LK tile pattern: one wave per tile; the tile is ROWS rows of NDW aligned dwords at byte offset org[w] (row pitch `pitch`); a load
// instruction covers 64 / NDW rows (lane = (row, dword)), as agt_lk_rs_body.h stages a corner's tiles.  Tiles never overlap.
template <int ROWS, int NDW>
__device__ __forceinline__ void tile_body(const uint8_t* p, const int* org, int ntiles, int pitch, uint32_t* out)
{
    const int w = (int)(((long)blockIdx.x * NT + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (w >= ntiles) return;
    constexpr int R = 64 / NDW, K = (ROWS + R - 1) / R;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p), 0, (int)BUF, 0x00020000);
    const int row = lane / NDW, dw = lane - row * NDW;
    const int so = __builtin_amdgcn_readfirstlane(org[w]);
    const int vo = row * pitch + 4 * dw;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < K; k++)
        if (row < R && k * R + row < ROWS) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so + k * R * pitch, 0);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(NT) void tile_j40x11(const uint8_t* p, const int* org, int ntiles, int pitch, uint32_t* out) { tile_body<40, 11>(p, org, ntiles, pitch, out); }
__global__ __launch_bounds__(NT) void tile_i24x7(const uint8_t* p, const int* org, int ntiles, int pitch, uint32_t* out) { tile_body<24, 7>(p, org, ntiles, pitch, out); }
// the same tiles fetched as whole 128-B lines: every row by 16-B loads of the lines it touches (lane = (row, 16-B piece of 2 lines))
template <int ROWS>
__device__ __forceinline__ void tile_lines_body(const uint8_t* p, const int* org, int ntiles, int pitch, uint32_t* out)
{
    const int w = (int)(((long)blockIdx.x * NT + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (w >= ntiles) return;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p), 0, (int)BUF, 0x00020000);
    const int row = lane >> 2, piece = lane & 3;          // 16 rows per instruction, 64 B per row (the tile's row lies within 64 aligned bytes: see main)
    const int so = __builtin_amdgcn_readfirstlane(org[w] & ~63);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < (ROWS + 15) / 16; k++)
        if (k * 16 + row < ROWS) {
            const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, row * pitch + 16 * piece, so + k * 16 * pitch, 0);
            acc += t[0] ^ t[1] ^ t[2] ^ t[3];
        }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(NT) void tile_j40_lines64(const uint8_t* p, const int* org, int ntiles, int pitch, uint32_t* out) { tile_lines_body<40>(p, org, ntiles, pitch, out); }

static void footprint(const std::vector<int>& org, int rows, int ndw, int pitch, long& exact, long& s32, long& s64, long& l128)
{
    std::set<long> a, b, c;
    exact = 0;
    for (int o : org)
        for (int r = 0; r < rows; r++) {
            const long lo = (long)o + (long)r * pitch, hi = lo + 4 * ndw - 1;
            exact += 4 * ndw;
            for (long x = lo >> 5; x <= hi >> 5; x++) a.insert(x);
            for (long x = lo >> 6; x <= hi >> 6; x++) b.insert(x);
            for (long x = lo >> 7; x <= hi >> 7; x++) c.insert(x);
        }
    s32 = (long)a.size() * 32; s64 = (long)b.size() * 64; l128 = (long)c.size() * 128;
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    uint8_t* buf; uint32_t* out; int* d_org;
    CHECK(hipMalloc(&buf, BUF)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, BUF)); CHECK(hipMemset(out, 0, 64));
    // tiles: images of 1280 x 720 (pitch 1280) laid back to back over the buffer; 3,072 x 8 tiles per launch at pseudo-random,
    // non-overlapping places: one tile per 64-row x 128-column cell, dword-aligned x inside the cell (the row then spans 1 or 2 lines)
    const int pitch = 1280, cells_x = 1280 / 128, cell_rows = 64;
    const long ncell_rows = BUF / ((long)pitch * cell_rows);
    std::vector<int> org, org64;
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    const int ntiles = 3072 * 8;
    for (int t = 0; t < ntiles; t++) {
        const long cy = (long)(((unsigned long)t * 2654435761ul) % (unsigned long)ncell_rows), cx = t % cells_x;
        const int x = (int)(rnd() % 20) * 4, y = (int)(rnd() % 20);
        org.push_back((int)((cy * cell_rows + y) * pitch + cx * 128 + x));
        // variant whose 44-byte row always lies inside ONE aligned 64-byte sector (for the whole-sector fetch kernel)
        org64.push_back((int)((cy * cell_rows + y) * pitch + cx * 128 + (int)(rnd() % 5) * 4 + 64 * (int)(rnd() & 1)));
    }
    {   // the hash above may map two tiles into one cell: keep the first of each cell
        std::set<long> seen; std::vector<int> o2, o3;
        for (int t = 0; t < ntiles; t++) { const long cell = (long)(org[t] / (pitch * cell_rows)) * cells_x + (org[t] % pitch) / 128; if (seen.insert(cell).second) { o2.push_back(org[t]); o3.push_back(org64[t]); } }
        org.swap(o2); org64.swap(o3);
    }
    const int nt = (int)org.size();
    CHECK(hipMalloc(&d_org, sizeof(int) * nt * 2));
    CHECK(hipMemcpy(d_org, org.data(), sizeof(int) * nt, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_org + nt, org64.data(), sizeof(int) * nt, hipMemcpyHostToDevice));
    long ex, s32, s64, l128;
    printf("{\n \"buffer_bytes\": %ld, \"reps\": %d, \"tiles\": %d,\n", BUF, reps, nt);
    const long sbytes = 512L << 20;
    printf(" \"stream_b128\": {\"bytes\": %ld},\n \"stream_b64\": {\"bytes\": %ld},\n \"stream_b32\": {\"bytes\": %ld},\n", sbytes, sbytes, sbytes);
    footprint(org, 40, 11, pitch, ex, s32, s64, l128);
    printf(" \"tile_j40x11\": {\"bytes\": %ld, \"bytes_32B_sectors\": %ld, \"bytes_64B_sectors\": %ld, \"bytes_128B_lines\": %ld},\n", ex, s32, s64, l128);
    footprint(org, 24, 7, pitch, ex, s32, s64, l128);
    printf(" \"tile_i24x7\": {\"bytes\": %ld, \"bytes_32B_sectors\": %ld, \"bytes_64B_sectors\": %ld, \"bytes_128B_lines\": %ld},\n", ex, s32, s64, l128);
    {
        std::vector<int> o64; for (int o : org64) o64.push_back(o & ~63);
        footprint(o64, 40, 16, pitch, ex, s32, s64, l128);
        printf(" \"tile_j40_lines64\": {\"bytes\": %ld, \"bytes_32B_sectors\": %ld, \"bytes_64B_sectors\": %ld, \"bytes_128B_lines\": %ld}\n}\n", ex, s32, s64, l128);
    }
    const int grid = 256 * 8;
    for (int r = 0; r < reps; r++) {
        // rotate the streamed half so that consecutive launches never re-read what the previous one left in the Infinity Cache
        const uint8_t* sp = buf + (r & 1) * sbytes;
        hipLaunchKernelGGL(tile_j40x11, dim3((nt * 64 + NT - 1) / NT), dim3(NT), 0, 0, buf, d_org, nt, pitch, out);
        hipLaunchKernelGGL(stream_b128, dim3(grid), dim3(NT), 0, 0, sp, sbytes, out);
        hipLaunchKernelGGL(tile_i24x7, dim3((nt * 64 + NT - 1) / NT), dim3(NT), 0, 0, buf, d_org, nt, pitch, out);
        hipLaunchKernelGGL(stream_b64, dim3(grid), dim3(NT), 0, 0, buf + ((r + 1) & 1) * sbytes, sbytes, out);
        hipLaunchKernelGGL(tile_j40_lines64, dim3((nt * 64 + NT - 1) / NT), dim3(NT), 0, 0, buf, d_org + nt, nt, pitch, out);
        hipLaunchKernelGGL(stream_b32, dim3(grid), dim3(NT), 0, 0, sp, sbytes, out);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
