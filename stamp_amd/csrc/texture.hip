// texture.hip -- the tile background filter of the reference, `_has_enough_texture`
// (src/stamp/preprocessing/tiling.py:280-291): tile.convert("L") -> cv2.Canny(gray, 40, 100) -> edges.mean()/255 >= cutoff.
// Upstream it runs per tile on one DataLoader worker (OpenCV on the host, SURVEY.md row H4: the likely host bottleneck of
// the feed); here one workgroup owns one tile, the grey image and the edge map live in LDS, nothing but the u8 tile is read
// and one float per tile is written.
//   grey  : Pillow's ITU-R 601 fixed point, L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16           (pinned: Pillow is here)
//   Canny : OpenCV's algorithm for aperture 3, L2gradient = false (third-party, opencv-python 4.13 in the reference's lock
//           file, NOT installed here -> parity unpinned): Sobel 3x3 with replicated borders, magnitude |dx| + |dy|,
//           non-maximum suppression with the fixed-point tan(22.5 deg) sector test (TG22 = 13573 / 2^15) and zero magnitude
//           outside the image, double threshold (m > high: edge; low < m <= high: candidate), 8-connected hysteresis.
//   The hysteresis fixed point (candidates connected to an edge) does not depend on visiting order, so the stack walk of
//   the CPU code becomes an in-LDS relaxation that repeats until no pixel changes.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace amds {

constexpr int TEX_MAX_S = 224;

__device__ __forceinline__ int tex_clamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// |dx| + |dy| and the signed derivatives at (y, x), replicated borders
__device__ __forceinline__ int tex_sobel(const uint8_t* g, int S, int y, int x, int& dx, int& dy) {
    const int ym = tex_clamp(y - 1, S - 1), yp = tex_clamp(y + 1, S - 1), xm = tex_clamp(x - 1, S - 1), xp = tex_clamp(x + 1, S - 1);
    const int a = g[ym * S + xm], b = g[ym * S + x], c = g[ym * S + xp];
    const int d = g[y * S + xm], f = g[y * S + xp];
    const int p = g[yp * S + xm], q = g[yp * S + x], r = g[yp * S + xp];
    dx = (c + 2 * f + r) - (a + 2 * d + p);
    dy = (p + 2 * q + r) - (a + 2 * b + c);
    return abs(dx) + abs(dy);
}

__device__ __forceinline__ int tex_mag(const uint8_t* g, int S, int y, int x) {
    if (y < 0 || y >= S || x < 0 || x >= S) return 0;          // the magnitude plane is zero outside the image
    int dx, dy;
    return tex_sobel(g, S, y, x, dx, dy);
}

__global__ void __launch_bounds__(256) tile_canny_kernel(const uint8_t* __restrict__ tiles, float* __restrict__ frac, uint8_t* __restrict__ edges_out,
                                                         uint8_t* __restrict__ gray_out, int S, int low, int high) {
    extern __shared__ uint8_t tex_smem[];
    uint8_t* g = tex_smem;                 // [S][S] grey
    uint8_t* m = tex_smem + S * S;         // [S][S] 0 = candidate, 1 = not an edge, 2 = edge
    __shared__ int s_changed, s_count;
    const int tid = threadIdx.x, b = blockIdx.x, n = S * S;
    const uint8_t* t = tiles + (size_t)b * n * 3;
    for (int i = tid; i < n; i += 256) {
        const int R = t[3 * i], G = t[3 * i + 1], B = t[3 * i + 2];
        g[i] = (uint8_t)((19595 * R + 38470 * G + 7471 * B + 0x8000) >> 16);
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (gray_out) for (int i = tid; i < n; i += 256) gray_out[(size_t)b * n + i] = g[i];
    for (int i = tid; i < n; i += 256) {
        const int y = i / S, x = i - y * S;
        int dx, dy;
        const int mag = tex_sobel(g, S, y, x, dx, dy);
        uint8_t state = 1;
        if (mag > low) {
            const int ax = abs(dx), ay = abs(dy) << 15;
            const int tg22x = ax * 13573;
            bool keep;
            if (ay < tg22x) {
                keep = mag > tex_mag(g, S, y, x - 1) && mag >= tex_mag(g, S, y, x + 1);
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    keep = mag > tex_mag(g, S, y - 1, x) && mag >= tex_mag(g, S, y + 1, x);
                } else {
                    const int s = ((dx ^ dy) < 0) ? -1 : 1;
                    keep = mag > tex_mag(g, S, y - 1, x - s) && mag > tex_mag(g, S, y + 1, x + s);
                }
            }
            if (keep) state = mag > high ? 2 : 0;
        }
        m[i] = state;
    }
    __syncthreads();
    // hysteresis: grow the edge set into 8-connected candidates until nothing changes
    for (;;) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        bool any = false;
        for (int i = tid; i < n; i += 256) {
            if (m[i] != 0) continue;
            const int y = i / S, x = i - y * S;
            bool hit = false;
#pragma unroll
            for (int dyy = -1; dyy <= 1; ++dyy)
#pragma unroll
                for (int dxx = -1; dxx <= 1; ++dxx) {
                    const int yy = y + dyy, xx = x + dxx;
                    if (yy >= 0 && yy < S && xx >= 0 && xx < S && m[yy * S + xx] == 2) hit = true;
                }
            if (hit) { m[i] = 2; any = true; }
        }
        if (any) s_changed = 1;
        __syncthreads();
        const int ch = s_changed;
        __syncthreads();
        if (!ch) break;
    }
    int cnt = 0;
    for (int i = tid; i < n; i += 256) {
        const bool e = m[i] == 2;
        cnt += e;
        if (edges_out) edges_out[(size_t)b * n + i] = e ? 255 : 0;
    }
    cnt = (int)wave_sum((float)cnt);
    if ((tid & 63) == 0) atomicAdd(&s_count, cnt);
    __syncthreads();
    if (tid == 0) frac[b] = (float)s_count / (float)n;       // == edges.mean() / 255
}

// The same filter, four pixels per thread and step (S % 4 == 0): the byte-at-a-time kernel above spent 1.25 ms on 256 tiles -- 2.9 % of the slide pipeline's GPU
// time -- in ~25 one-byte LDS reads per pixel, three one-byte global loads per pixel and a hysteresis sweep that visited every pixel of every round.  Here a thread
// owns a QUAD (4 consecutive pixels of a row = one dword of each LDS plane): 12-byte global loads, the 5 x 8 grey neighbourhood of a quad from 3 dwords per row,
// the neighbours' magnitudes only for quads that hold a pixel above the low threshold, and a hysteresis sweep that skips a quad without candidates after ONE
// dword read.  Same arithmetic per pixel, same fixed point: bit-identical outputs (tests/test_gpu_texture.py holds both kernels to the oracle).
__device__ __forceinline__ int tex_byte(uint32_t a, uint32_t b, uint32_t c, int idx) {      // byte idx (0..11) of the 12 bytes a | b | c
    const uint32_t w = idx < 4 ? a : (idx < 8 ? b : c);
    return (int)((w >> ((idx & 3) * 8)) & 0xffu);
}

__global__ void __launch_bounds__(256) tile_canny_quad_kernel(const uint8_t* __restrict__ tiles, float* __restrict__ frac, uint8_t* __restrict__ edges_out,
                                                              uint8_t* __restrict__ gray_out, int S, int low, int high) {
    extern __shared__ uint8_t tex_smem[];
    uint32_t* g32 = reinterpret_cast<uint32_t*>(tex_smem);                 // [S][S / 4] grey, 4 pixels per dword
    uint32_t* m32 = reinterpret_cast<uint32_t*>(tex_smem + S * S);         // [S][S / 4] state bytes: 0 = candidate, 1 = not an edge, 2 = edge
    __shared__ int s_changed, s_count;
    const int tid = threadIdx.x, b = blockIdx.x, n = S * S, Q = S / 4, nq = n / 4;
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tiles + (size_t)b * n * 3);
    for (int q = tid; q < nq; q += 256) {
        const uint32_t a = t32[3 * q], bb = t32[3 * q + 1], c = t32[3 * q + 2];
        uint32_t out = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int R = tex_byte(a, bb, c, 3 * e), G = tex_byte(a, bb, c, 3 * e + 1), B = tex_byte(a, bb, c, 3 * e + 2);
            out |= (uint32_t)((19595 * R + 38470 * G + 7471 * B + 0x8000) >> 16) << (8 * e);
        }
        g32[q] = out;
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (gray_out) {
        uint32_t* go = reinterpret_cast<uint32_t*>(gray_out + (size_t)b * n);
        for (int q = tid; q < nq; q += 256) go[q] = g32[q];
    }
    // grey of row r (clamped by the caller), columns x0 - 2 .. x0 + 5, replicated at the left / right border
    auto load_row = [&](int r, int qx, int (&w)[8]) {
        const uint32_t mid = g32[r * Q + qx];
        const uint32_t lft = qx > 0 ? g32[r * Q + qx - 1] : 0u, rgt = qx + 1 < Q ? g32[r * Q + qx + 1] : 0u;
        const int x0 = 4 * qx;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int col = x0 - 2 + k;
            col = col < 0 ? 0 : (col > S - 1 ? S - 1 : col);
            w[k] = tex_byte(lft, mid, rgt, col - x0 + 4);
        }
    };
    for (int q = tid; q < nq; q += 256) {
        const int y = q / Q, qx = q - y * Q, x0 = 4 * qx;
        int w[5][8];                                  // rows y - 2 .. y + 2 (replicated), columns x0 - 2 .. x0 + 5
        load_row(tex_clamp(y - 1, S - 1), qx, w[1]);
        load_row(y, qx, w[2]);
        load_row(tex_clamp(y + 1, S - 1), qx, w[3]);
        // Sobel at (row index rr of w, column index k of w): needs w[rr - 1 .. rr + 1][k - 1 .. k + 1]
        auto sob = [&](int rr, int k, int& dx, int& dy) {
            const int a = w[rr - 1][k - 1], bq = w[rr - 1][k], c = w[rr - 1][k + 1], d = w[rr][k - 1], f = w[rr][k + 1], p = w[rr + 1][k - 1], qq = w[rr + 1][k], r = w[rr + 1][k + 1];
            dx = (c + 2 * f + r) - (a + 2 * d + p);
            dy = (p + 2 * qq + r) - (a + 2 * bq + c);
            return abs(dx) + abs(dy);
        };
        int dxs[4], dys[4], mags[4];
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mags[e] = sob(2, 2 + e, dxs[e], dys[e]);
            any = any || mags[e] > low;
        }
        uint32_t st = 0x01010101u;
        if (any) {
            // the neighbours' magnitudes: rows y - 1, y + 1 need rows y - 2, y + 2 (their own replicated borders); a neighbour OUTSIDE the image has magnitude 0
            load_row(tex_clamp(y - 2, S - 1), qx, w[0]);
            load_row(tex_clamp(y + 2, S - 1), qx, w[4]);
            // w's rows are the CLAMPED rows y-2..y+2; the Sobel of the pixel in row y-1 must see rows clamp(y-2), y-1, clamp(y): w[0], w[1], w[2] -- except
            // at the top border, where row y-1 itself does not exist (magnitude 0), and likewise at the bottom.  Inside the image clamp(y-1) = y-1.
            int mg[3][6];                             // magnitude at rows y - 1 .. y + 1, columns x0 - 1 .. x0 + 4
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int yy = y - 1 + rr, xx = x0 - 1 + k;
                    int ddx, ddy;
                    // the replicated COLUMN border of a neighbour at xx uses columns clamp(xx - 1), xx, clamp(xx + 1): w's column k + 1 is xx and its neighbours in w are
                    // already the clamped ones except when xx itself is a border column whose outer neighbour lies two past x0's window -- covered because w holds
                    // columns x0 - 2 .. x0 + 5 with clamping applied per column
                    mg[rr][k] = (yy < 0 || yy >= S || xx < 0 || xx >= S) ? 0 : (rr == 1 && k >= 1 && k <= 4 ? mags[k - 1] : sob(rr + 1, k + 1, ddx, ddy));
                }
            st = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mag = mags[e], dx = dxs[e], dy = dys[e], k = e + 1;
                uint32_t state = 1;
                if (mag > low) {
                    const int ax = abs(dx), ay = abs(dy) << 15;
                    const int tg22x = ax * 13573;
                    bool keep;
                    if (ay < tg22x) {
                        keep = mag > mg[1][k - 1] && mag >= mg[1][k + 1];
                    } else {
                        const int tg67x = tg22x + (ax << 16);
                        if (ay > tg67x) {
                            keep = mag > mg[0][k] && mag >= mg[2][k];
                        } else {
                            const int sgn = ((dx ^ dy) < 0) ? -1 : 1;
                            keep = mag > mg[0][k - sgn] && mag > mg[2][k + sgn];
                        }
                    }
                    if (keep) state = mag > high ? 2 : 0;
                }
                st |= state << (8 * e);
            }
        }
        m32[q] = st;
    }
    __syncthreads();
    // hysteresis: grow the edge set into 8-connected candidates until nothing changes.  Only quads that hold a candidate are ever visited again: their indices are
    // collected once (the grey plane is dead by now: the list lives in its LDS), and a round walks the list, not the image.  (A thread per row sweeping left / right
    // was tried: fewer rounds, but 112 dependent LDS reads per thread and round -- 0.9 ms against 0.62.)  The fixed point does not depend on the visiting order.
    __shared__ int s_nlist;
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    uint16_t* list = reinterpret_cast<uint16_t*>(tex_smem);                 // <= S * S / 4 = 12 544 entries of 2 bytes in the grey plane's 50 KB
    for (int q = tid; q < nq; q += 256) {
        const uint32_t v = m32[q];
        if ((((v - 0x01010101u) & ~v) & 0x80808080u) != 0) list[atomicAdd(&s_nlist, 1)] = (uint16_t)q;
    }
    __syncthreads();
    const int nlist = s_nlist;
    for (;;) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        bool anyc = false;
        for (int k = tid; k < nlist; k += 256) {
            const int q = list[k];
            uint32_t v = m32[q];
            if ((((v - 0x01010101u) & ~v) & 0x80808080u) == 0) continue;          // every candidate of this quad has been decided
            const int y = q / Q, qx = q - y * Q;
            uint32_t nb[3][3];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int yy = y - 1 + rr;
                const bool in = yy >= 0 && yy < S;
                nb[rr][0] = in && qx > 0 ? m32[yy * Q + qx - 1] : 0x01010101u;
                nb[rr][1] = in ? (rr == 1 ? v : m32[yy * Q + qx]) : 0x01010101u;
                nb[rr][2] = in && qx + 1 < Q ? m32[yy * Q + qx + 1] : 0x01010101u;
            }
            bool changed = false;
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)                                    // forward, then backward inside the quad
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const int e = rep == 0 ? ee : 3 - ee;
                    if (((v >> (8 * e)) & 0xffu) != 0) continue;
                    bool hit = false;
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int dxx = -1; dxx <= 1; ++dxx) hit = hit || tex_byte(nb[rr][0], nb[rr][1], nb[rr][2], 4 + e + dxx) == 2;
                    if (hit) {
                        v |= 2u << (8 * e);
                        nb[1][1] = v;
                        changed = true;
                    }
                }
            if (changed) { m32[q] = v; anyc = true; }
        }
        if (anyc) s_changed = 1;
        __syncthreads();
        const int ch = s_changed;
        __syncthreads();
        if (!ch) break;
    }
    int cnt = 0;
    uint32_t* eo = edges_out ? reinterpret_cast<uint32_t*>(edges_out + (size_t)b * n) : nullptr;
    for (int q = tid; q < nq; q += 256) {
        const uint32_t v = m32[q];
        uint32_t o = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ed = ((v >> (8 * e)) & 0xffu) == 2;
            cnt += ed;
            o |= ed ? (0xffu << (8 * e)) : 0u;
        }
        if (eo) eo[q] = o;
    }
    cnt = (int)wave_sum((float)cnt);
    if ((tid & 63) == 0) atomicAdd(&s_count, cnt);
    __syncthreads();
    if (tid == 0) frac[b] = (float)s_count / (float)n;       // == edges.mean() / 255
}

// ---------------------------------------------------------------------------------------------
// Keep-mask compaction on the device (no host round trip between the texture filter and the tile encoder): rows whose score passes the
// cutoff are appended, IN ORDER, to a destination buffer whose fill level lives in device memory.
//   slot[i] = score ? (score[i] >= cutoff ? *count + (number of kept rows before i) : -1) : *count + i;   *count += kept
// One workgroup scans (n <= 4096 rows per call: one batch of decoded tiles), a second launch moves the rows.  Deterministic: positions
// are a prefix sum, not atomics.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) compact_scan_kernel(const float* __restrict__ score, float cutoff, int* __restrict__ count, int* __restrict__ slot,
                                                           int n, int capacity) {
    __shared__ int s_part[256];
    const int tid = threadIdx.x;
    const int per = (n + 255) / 256, i0 = tid * per, i1 = min(n, i0 + per);
    int c = 0;
    for (int i = i0; i < i1; ++i) c += (score == nullptr || score[i] >= cutoff) ? 1 : 0;
    s_part[tid] = c;
    __syncthreads();
    int before = 0;
    for (int j = 0; j < tid; ++j) before += s_part[j];          // 256 terms: not worth a tree
    const int base = *count;
    int pos = base + before;
    for (int i = i0; i < i1; ++i) {
        const bool keep = score == nullptr || score[i] >= cutoff;
        slot[i] = keep ? (pos < capacity ? pos : -2) : -1;      // -2: would not fit (the host sizes the buffer so that it cannot happen)
        pos += keep ? 1 : 0;
    }
    __syncthreads();
    if (tid == 255) *count = min(pos, capacity);
}
__global__ void __launch_bounds__(256) compact_move_kernel(const uint8_t* __restrict__ src, long row_bytes, const int* __restrict__ slot,
                                                           uint8_t* __restrict__ dst) {
    const int i = blockIdx.y, s = slot[i];
    if (s < 0) return;
    const u32x4* a = reinterpret_cast<const u32x4*>(src + (long)i * row_bytes);
    u32x4* b = reinterpret_cast<u32x4*>(dst + (long)s * row_bytes);
    const long nv = row_bytes >> 4;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) b[v] = a[v];
}

// rows [m, *count) of src -> rows [0, *count - m) of dst, then *count -= m: what is left of an accumulation buffer after its first m rows
// went to the encoder moves to the front of the other buffer, the fill level staying on the device
__global__ void __launch_bounds__(256) compact_shift_move_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long row_bytes, int m,
                                                                 const int* __restrict__ count) {
    const int r = m + blockIdx.y;
    if (r >= *count) return;
    const u32x4* a = reinterpret_cast<const u32x4*>(src + (long)r * row_bytes);
    u32x4* b = reinterpret_cast<u32x4*>(dst + (long)(r - m) * row_bytes);
    const long nv = row_bytes >> 4;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) b[v] = a[v];
}
__global__ void compact_shift_count_kernel(int* __restrict__ count, int m) { *count = max(*count - m, 0); }

}  // namespace amds

using namespace amds;

extern "C" int amds_compact_shift_u8(const uint8_t* src, uint8_t* dst, long row_bytes, int m, int max_rows, int* count_dev, void* stream) {
    AMDS_REQUIRE(src && dst && count_dev && row_bytes > 0 && row_bytes % 16 == 0 && m >= 0 && max_rows >= 0 && max_rows <= 65535,
                 "amds_compact_shift_u8: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (max_rows > 0) {
        const int bx = (int)std::min<long>(64, (row_bytes / 16 + 255) / 256);
        hipLaunchKernelGGL(compact_shift_move_kernel, dim3(bx, max_rows), dim3(256), 0, st, src, dst, row_bytes, m, count_dev);
        AMDS_LAUNCH_CHECK("compact_shift_move_kernel");
    }
    hipLaunchKernelGGL(compact_shift_count_kernel, dim3(1), dim3(1), 0, st, count_dev, m);
    AMDS_LAUNCH_CHECK("compact_shift_count_kernel");
    return AMDS_OK;
}

extern "C" int amds_compact_rows_u8(const uint8_t* src, long row_bytes, const float* score, float cutoff, uint8_t* dst, int capacity_rows,
                                    int* count_dev, int* slot_out, int n, void* stream) {
    AMDS_REQUIRE(n >= 0 && n <= 4096 && row_bytes > 0 && row_bytes % 16 == 0 && capacity_rows > 0, "amds_compact_rows_u8: bad sizes (n <= 4096, row_bytes %% 16 == 0)");
    if (n == 0) return AMDS_OK;
    AMDS_REQUIRE(src && dst && count_dev && slot_out, "amds_compact_rows_u8: null pointer");
    AMDS_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "amds_compact_rows_u8: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(256), 0, st, score, cutoff, count_dev, slot_out, n, capacity_rows);
    AMDS_LAUNCH_CHECK("compact_scan_kernel");
    const int bx = (int)std::min<long>(64, (row_bytes / 16 + 255) / 256);
    hipLaunchKernelGGL(compact_move_kernel, dim3(bx, n), dim3(256), 0, st, src, row_bytes, slot_out, dst);
    AMDS_LAUNCH_CHECK("compact_move_kernel");
    return AMDS_OK;
}

extern "C" int amds_tile_edge_fraction_u8(const uint8_t* tiles, float* frac, uint8_t* edges, uint8_t* gray, int B, int S, int low, int high,
                                          void* stream) {
    AMDS_REQUIRE(B >= 0 && S >= 3 && S <= TEX_MAX_S, "amds_tile_edge_fraction_u8: tile size %d unsupported (3..%d)", S, TEX_MAX_S);
    AMDS_REQUIRE(low >= 0 && high >= low, "amds_tile_edge_fraction_u8: thresholds low=%d high=%d", low, high);
    if (B == 0) return AMDS_OK;
    AMDS_REQUIRE(tiles && frac, "amds_tile_edge_fraction_u8: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)2 * S * S;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_canny_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TEX_MAX_S * TEX_MAX_S));
        attr_set = true;
    }
    ProfScope prof(PROF_OTHER, (double)B * S * S * 3, st);
    // four pixels per thread and step when the rows split into dwords (S = 224 does); AMDS_CANNY_QUAD=0: the byte-at-a-time kernel (A/B, and any other S)
    static const bool quad_on = !(getenv("AMDS_CANNY_QUAD") && atoi(getenv("AMDS_CANNY_QUAD")) == 0);
    const bool quad = quad_on && S % 4 == 0 && S >= 8 && ((uintptr_t)tiles & 3) == 0 && (edges == nullptr || ((uintptr_t)edges & 3) == 0) && (gray == nullptr || ((uintptr_t)gray & 3) == 0);
    if (quad) {
        static bool attr_q = false;
        if (!attr_q) {
            AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_canny_quad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TEX_MAX_S * TEX_MAX_S));
            attr_q = true;
        }
        hipLaunchKernelGGL(tile_canny_quad_kernel, dim3(B), dim3(256), lds, st, tiles, frac, edges, gray, S, low, high);
    } else {
        hipLaunchKernelGGL(tile_canny_kernel, dim3(B), dim3(256), lds, st, tiles, frac, edges, gray, S, low, high);
    }
    AMDS_LAUNCH_CHECK("tile_canny_kernel");
    return AMDS_OK;
}
