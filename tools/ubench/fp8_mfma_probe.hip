// Which lane holds which (row, k) of the A / B operands of v_mfma_f32_16x16x128_f8f6f4, and what do the scale arguments do?
// hipcc --offload-arch=gfx950 -O2 tools/ubench/fp8_mfma_probe.hip -o /tmp/fp8_probe && /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// e4m3fn (OCP) decode
static float e4m3_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 0) r = std::ldexp((float)m, -9);
    else if (e == 15 && m == 7) r = NAN;
    else r = std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}

template <int SA, int SB>
__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int hyp) {
    const int lane = threadIdx.x;
    const int r = lane & 15, g = lane >> 4;
    v8i a, b;
    const uint8_t* ap;
    const uint8_t* bp;
    // hypothesis 0: lane (r, g) holds k = 32 g .. 32 g + 31 (contiguous);  hypothesis 1: two 16-byte halves k = 16 g .. +15 and 64 + 16 g .. +15
    uint8_t ta[32], tb[32];
    for (int i = 0; i < 32; ++i) {
        const int k = hyp == 0 ? 32 * g + i : (i < 16 ? 16 * g + i : 64 + 16 * g + (i - 16));
        ta[i] = A[r * 128 + k];
        tb[i] = B[r * 128 + k];      // B given as [n][k]
    }
    memcpy(&a, ta, 32);
    memcpy(&b, tb, 32);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, SA, 0, SB);
    for (int i = 0; i < 4; ++i) C[lane * 4 + i] = c[i];
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    srand(1);
    for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x3c; }
    for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x3c; }
    std::vector<double> ref(16 * 16);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 128; ++k) s += (double)e4m3_to_f32(A[i * 128 + k]) * e4m3_to_f32(B[j * 128 + k]);
            ref[i * 16 + j] = s;
        }
    uint8_t *dA, *dB;
    float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    std::vector<float> C(256);
    auto report = [&](const char* name) {
        hipDeviceSynchronize();
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        // C layouts: L0: lane -> col = lane & 15, row = 4 (lane >> 4) + reg (A rows x B rows);  L1: transposed
        double e0 = 0, e1 = 0, nrm = 0, ratio = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i) {
                const int col = lane & 15, row = 4 * (lane >> 4) + i;
                const double v = C[lane * 4 + i];
                e0 += (v - ref[row * 16 + col]) * (v - ref[row * 16 + col]);
                e1 += (v - ref[col * 16 + row]) * (v - ref[col * 16 + row]);
                nrm += ref[row * 16 + col] * ref[row * 16 + col];
            }
        ratio = C[0] / ref[0];
        printf("%-44s rel err: C[row = A row][col = B row] %.3e | transposed %.3e | C[0]/ref %.6g\n", name, std::sqrt(e0 / nrm), std::sqrt(e1 / nrm), ratio);
    };
    for (int hyp = 0; hyp < 2; ++hyp) {
        probe<0, 0><<<1, 64>>>(dA, dB, dC, hyp);
        report(hyp == 0 ? "hyp 0 (k = 32g..32g+31), scales 0, 0" : "hyp 1 (two 16-byte halves), scales 0, 0");
    }
    probe<127, 127><<<1, 64>>>(dA, dB, dC, 0);
    report("hyp 0, scales 127, 127");
    probe<128, 127><<<1, 64>>>(dA, dB, dC, 0);
    report("hyp 0, scales 128, 127");
    return 0;
}
