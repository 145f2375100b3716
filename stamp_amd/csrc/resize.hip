// resize.hip -- supertile -> tiles: the resize + crop step of the reference's WSI reader ("next" row N3),
// reference src/stamp/preprocessing/tiling.py:326-343 (`slide.read_region(...).resize((k*tile_px, k*tile_px)).convert("RGB")`) and
// :225-246 (`supertile.crop(...)` into k x k tiles, row-major).  `read_region` returns RGBA; PIL's `Image.resize` on an RGBA image
// (default filter: bicubic) works on the PREMULTIPLIED image ("RGBa") and converts back afterwards -- Pillow 12.1.1 (reference
// uv.lock), restated here bit for bit and pinned against the installed Pillow by tests/test_oracle_tiling.py:
//   premultiply    c' = MULDIV255(c, a) = (t = c*a + 128, ((t >> 8) + t) >> 8)                       (Convert.c rgbA2rgba)
//   resample       two passes, horizontal then vertical, 8-bit intermediate image; per output sample the normalised filter taps
//                  as 32-bit fixed point with 22 fraction bits, acc = 2^21 + sum(pixel * coef), out = clip8(acc >> 22)   (Resample.c)
//   un-premultiply a == 0 or a == 255: copy; else clip8(255 * c' / a)                                 (Convert.c rgba2rgbA)
//   convert("RGB") drops alpha                                                                         (regions past the slide edge: black)
// The tap tables (bounds + coefficients per output coordinate; the supertile is square, one table serves both passes) are
// computed by the host in double precision exactly as Pillow's precompute_coeffs does (stamp_amd/tiling.py).
// HBM-bound byte work: per supertile 4 S^2 bytes in, 4 S O intermediate, 3 O^2 out (S = 1024, O = 224 k).
#include "common.h"
#include <stdlib.h>

namespace amds {

constexpr int RS_PREC = 22;

__device__ __forceinline__ int muldiv255(int c, int a) { const int t = c * a + 128; return ((t >> 8) + t) >> 8; }
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass: rgba [n][S][S] (uchar4) -> mid [n][S][O] (uchar4, premultiplied), one thread per output sample
__global__ void __launch_bounds__(256) resize_h_kernel(const uchar4* __restrict__ src, uchar4* __restrict__ mid, int S, int O,
                                                       const int2* __restrict__ bounds, const int* __restrict__ coef, int ksize) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (xx >= O) return;
    const uchar4* row = src + ((long)blockIdx.z * S + y) * S;
    const int2 b = bounds[xx];
    const int* k = coef + (long)xx * ksize;
    int a0 = 1 << (RS_PREC - 1), a1 = a0, a2 = a0, a3 = a0;
    for (int x = 0; x < b.y; ++x) {
        const uchar4 p = row[b.x + x];
        const int w = k[x], al = p.w;
        int r = p.x, g = p.y, bl = p.z;
        if (al != 255) { r = muldiv255(r, al); g = muldiv255(g, al); bl = muldiv255(bl, al); }
        a0 += r * w; a1 += g * w; a2 += bl * w; a3 += al * w;
    }
    mid[((long)blockIdx.z * S + y) * O + xx] = make_uchar4((unsigned char)clip8(a0 >> RS_PREC), (unsigned char)clip8(a1 >> RS_PREC),
                                                           (unsigned char)clip8(a2 >> RS_PREC), (unsigned char)clip8(a3 >> RS_PREC));
}

// The same pass with the source row staged in LDS once (coalesced, premultiplied on the way in): every output sample of a row reads ~2 x the down-scaling factor taps
// and neighbouring samples share most of them -- read through the caches by one thread per output sample the 268 MB of a 64-supertile batch became 1.3 GB of
// 4-byte loads (0.62 TB/s of input).  Same integers, same order of the tap sum.
__global__ void __launch_bounds__(256) resize_h_lds_kernel(const uchar4* __restrict__ src, uchar4* __restrict__ mid, int S, int O,
                                                           const int2* __restrict__ bounds, const int* __restrict__ coef, int ksize) {
    extern __shared__ uchar4 rs_row[];
    const int y = blockIdx.y;
    const uchar4* row = src + ((long)blockIdx.z * S + y) * S;
    for (int x = threadIdx.x; x < S; x += 256) {
        uchar4 p = row[x];
        const int al = p.w;
        if (al != 255) { p.x = (unsigned char)muldiv255(p.x, al); p.y = (unsigned char)muldiv255(p.y, al); p.z = (unsigned char)muldiv255(p.z, al); }
        rs_row[x] = p;
    }
    __syncthreads();
    for (int xx = threadIdx.x; xx < O; xx += 256) {
        const int2 b = bounds[xx];
        const int* k = coef + (long)xx * ksize;
        int a0 = 1 << (RS_PREC - 1), a1 = a0, a2 = a0, a3 = a0;
        for (int x = 0; x < b.y; ++x) {
            const uchar4 p = rs_row[b.x + x];
            const int w = k[x];
            a0 += p.x * w; a1 += p.y * w; a2 += p.z * w; a3 += p.w * w;
        }
        mid[((long)blockIdx.z * S + y) * O + xx] = make_uchar4((unsigned char)clip8(a0 >> RS_PREC), (unsigned char)clip8(a1 >> RS_PREC),
                                                               (unsigned char)clip8(a2 >> RS_PREC), (unsigned char)clip8(a3 >> RS_PREC));
    }
}

// vertical pass + un-premultiply + drop alpha + crop into k x k tiles of t x t: tiles [n*k*k][t][t][3]
__global__ void __launch_bounds__(256) resize_v_kernel(const uchar4* __restrict__ mid, unsigned char* __restrict__ tiles, int S, int O, int kt, int t,
                                                       const int2* __restrict__ bounds, const int* __restrict__ coef, int ksize) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= O) return;
    const uchar4* img = mid + (long)blockIdx.z * S * O;
    const int2 b = bounds[yy];
    const int* k = coef + (long)yy * ksize;
    int a0 = 1 << (RS_PREC - 1), a1 = a0, a2 = a0, a3 = a0;
    for (int y = 0; y < b.y; ++y) {
        const uchar4 p = img[(long)(b.x + y) * O + xx];
        const int w = k[y];
        a0 += p.x * w; a1 += p.y * w; a2 += p.z * w; a3 += p.w * w;
    }
    int r = clip8(a0 >> RS_PREC), g = clip8(a1 >> RS_PREC), bl = clip8(a2 >> RS_PREC);
    const int al = clip8(a3 >> RS_PREC);
    if (al != 255 && al != 0) { r = clip8(255 * r / al); g = clip8(255 * g / al); bl = clip8(255 * bl / al); }
    const int ty = yy / t, tx = xx / t;
    unsigned char* o = tiles + ((((long)blockIdx.z * kt + ty) * kt + tx) * t + (yy - ty * t)) * (long)t * 3 + (long)(xx - tx * t) * 3;
    o[0] = (unsigned char)r; o[1] = (unsigned char)g; o[2] = (unsigned char)bl;
}

// ---- RGB tile: Resize(O, bicubic) then CenterCrop(t), the transform some of the reference's extractors put in front of their model
// (gigapath.py:21-28 Resize(256, BICUBIC) + CenterCrop(224)).  torchvision's Resize on a PIL image is PIL's own resize:
// the same two-pass 8-bit resample as above without the alpha handling; the crop picks columns / rows [c0, c0 + t) of the O x O image, so only
// those are computed.  tiles [B][S][S][3] -> mid [B][S][t][3] -> out [B][t][t][3].
__global__ void __launch_bounds__(256) tile_resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ mid, int S, int t, int c0,
                                                            const int2* __restrict__ bounds, const int* __restrict__ coef, int ksize) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (xo >= t) return;
    const unsigned char* row = src + ((long)blockIdx.z * S + y) * S * 3;
    const int2 b = bounds[c0 + xo];
    const int* k = coef + (long)(c0 + xo) * ksize;
    int a0 = 1 << (RS_PREC - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < b.y; ++x) {
        const unsigned char* p = row + (b.x + x) * 3;
        const int w = k[x];
        a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
    }
    unsigned char* o = mid + (((long)blockIdx.z * S + y) * t + xo) * 3;
    o[0] = (unsigned char)clip8(a0 >> RS_PREC); o[1] = (unsigned char)clip8(a1 >> RS_PREC); o[2] = (unsigned char)clip8(a2 >> RS_PREC);
}

__global__ void __launch_bounds__(256) tile_resize_v_kernel(const unsigned char* __restrict__ mid, unsigned char* __restrict__ out, int S, int t, int c0,
                                                            const int2* __restrict__ bounds, const int* __restrict__ coef, int ksize) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y;
    if (xo >= t) return;
    const unsigned char* img = mid + (long)blockIdx.z * S * t * 3;
    const int2 b = bounds[c0 + yo];
    const int* k = coef + (long)(c0 + yo) * ksize;
    int a0 = 1 << (RS_PREC - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < b.y; ++y) {
        const unsigned char* p = img + ((long)(b.x + y) * t + xo) * 3;
        const int w = k[y];
        a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
    }
    unsigned char* o = out + (((long)blockIdx.z * t + yo) * t + xo) * 3;
    o[0] = (unsigned char)clip8(a0 >> RS_PREC); o[1] = (unsigned char)clip8(a1 >> RS_PREC); o[2] = (unsigned char)clip8(a2 >> RS_PREC);
}

}  // namespace amds

using namespace amds;

extern "C" size_t amds_tile_resize_crop_workspace_bytes(int n, int S, int t) { return (size_t)n * S * t * 3; }

extern "C" int amds_tile_resize_crop_u8(const uint8_t* tiles, uint8_t* out, int n, int S, int O, int t, const int* bounds, const int* coef, int ksize, void* ws,
                                        size_t ws_bytes, void* stream) {
    if (n == 0) return AMDS_OK;
    AMDS_REQUIRE(tiles && out && bounds && coef && ws, "amds_tile_resize_crop_u8: null pointer");
    AMDS_REQUIRE(n > 0 && n <= 65535 && S > 0 && S <= 65535 && O >= t && t > 0 && t <= 65535 && ksize > 0, "amds_tile_resize_crop_u8: bad shape (crop %d of %d)", t, O);
    if (ws_bytes < amds_tile_resize_crop_workspace_bytes(n, S, t)) { set_error("amds_tile_resize_crop_u8: workspace too small"); return AMDS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int q = (O - t) / 2;                                      // torchvision center_crop: int(round((size - crop) / 2.0)); Python rounds half to even
    const int c0e = ((O - t) % 2 == 0 || q % 2 == 0) ? q : q + 1;
    hipLaunchKernelGGL(tile_resize_h_kernel, dim3(cdiv(t, 256), S, n), dim3(256), 0, st, tiles, (unsigned char*)ws, S, t, c0e, (const int2*)bounds, coef, ksize);
    AMDS_LAUNCH_CHECK("tile_resize_h_kernel");
    hipLaunchKernelGGL(tile_resize_v_kernel, dim3(cdiv(t, 256), t, n), dim3(256), 0, st, (const unsigned char*)ws, out, S, t, c0e, (const int2*)bounds, coef, ksize);
    AMDS_LAUNCH_CHECK("tile_resize_v_kernel");
    return AMDS_OK;
}

extern "C" size_t amds_supertiles_to_tiles_workspace_bytes(int n, int S, int k, int t) { return (size_t)n * S * k * t * 4; }

extern "C" int amds_supertiles_to_tiles_u8(const uint8_t* rgba, uint8_t* tiles, int n, int S, int k, int t, const int* bounds, const int* coef,
                                           int ksize, void* ws, size_t ws_bytes, void* stream) {
    if (n == 0) return AMDS_OK;
    AMDS_REQUIRE(rgba && tiles && bounds && coef && ws, "amds_supertiles_to_tiles_u8: null pointer");
    AMDS_REQUIRE(n >= 0 && n <= 65535 && S > 0 && S <= 65535 && k > 0 && t > 0 && (long)k * t <= 65535 && ksize > 0, "amds_supertiles_to_tiles_u8: bad shape");
    if (ws_bytes < amds_supertiles_to_tiles_workspace_bytes(n, S, k, t)) { set_error("amds_supertiles_to_tiles_u8: workspace too small"); return AMDS_ERR_WORKSPACE; }
    AMDS_REQUIRE(((uintptr_t)rgba & 3) == 0 && ((uintptr_t)ws & 3) == 0, "amds_supertiles_to_tiles_u8: 4-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const int O = k * t;
    ProfScope prof(PROF_OTHER, (double)n * (4.0 * S * S + 8.0 * S * O + 3.0 * O * O), st);
    static const bool lds_on = !(getenv("AMDS_RESIZE_LDS") && atoi(getenv("AMDS_RESIZE_LDS")) == 0);       // 0: one thread per output sample through the caches (A/B)
    if (lds_on && (size_t)S * 4 <= 48 * 1024)
        hipLaunchKernelGGL(resize_h_lds_kernel, dim3(1, S, n), dim3(256), (size_t)S * 4, st, (const uchar4*)rgba, (uchar4*)ws, S, O, (const int2*)bounds, coef, ksize);
    else
        hipLaunchKernelGGL(resize_h_kernel, dim3(cdiv(O, 256), S, n), dim3(256), 0, st, (const uchar4*)rgba, (uchar4*)ws, S, O, (const int2*)bounds, coef, ksize);
    AMDS_LAUNCH_CHECK("resize_h_kernel");
    hipLaunchKernelGGL(resize_v_kernel, dim3(cdiv(O, 256), O, n), dim3(256), 0, st, (const uchar4*)ws, tiles, S, O, k, t, (const int2*)bounds, coef, ksize);
    AMDS_LAUNCH_CHECK("resize_v_kernel");
    return AMDS_OK;
}
