// ds_read_b64_tr_b16 (gfx950): what does lane l get when every lane hands in its own 8-byte-aligned LDS address?
// LDS holds u16 element e at byte 2 e.  Case A: lane l reads address 8 l (64 consecutive 4-element vectors).  Prints, per lane, the 4 element ids it got.
//   hipcc --offload-arch=gfx950 -O2 -o tr16_probe tools/ubench/tr16_probe.hip && ./tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr;
    if (mode == 0) addr = 8 * l;                                     // lane l: elements 4l .. 4l+3
    else if (mode == 1) addr = (l & 15) * 64 + (l >> 4) * 8;         // 16 rows of 32 elements (64 B); lane group g reads column chunk g
    else addr = ((l & 15) >> 2) * 512 + (l & 3) * 8 + (l >> 4) * 32; // rows of 256 elements: lane (r = (l&15)>>2, c4 = l&3), group g -> +16 elements
    s16x4 v;
    const unsigned a = (unsigned)(size_t)(lds) + addr;               // LDS byte address
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("%s", (l & 3) == 3 ? "\n" : " |"); }
    }
    return 0;
}
