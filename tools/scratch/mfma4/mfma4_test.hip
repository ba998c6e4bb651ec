// Prints the operand layout of v_mfma_f32_4x4x1_16B_f32 (and its A-broadcast form) as observed on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float a = 1.f + l, b = 1000.f + l;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    f32x4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 3, 0);
    for (int r = 0; r < 4; ++r) { out[l * 8 + r] = d0[r]; out[l * 8 + 4 + r] = d1[r]; }
}
int main() {
    float* d; hipMalloc(&d, 64 * 8 * 4);
    k<<<1, 64>>>(d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        if (!(l < 8 || l == 13 || l == 63)) continue;
        printf("lane %2d:", l);
        for (int v = 0; v < 2; ++v) {
            printf(v ? "  | cbsz4 abid3:" : " plain:");
            for (int r = 0; r < 4; ++r) {
                // decode product = (1 + la) * (1000 + lb)
                const float p = h[l * 8 + 4 * v + r];
                int la = -1, lb = -1;
                for (int x = 0; x < 64 && la < 0; ++x) for (int y = 0; y < 64; ++y) if ((1.f + x) * (1000.f + y) == p) { la = x; lb = y; break; }
                printf(" r%d=A%d*B%d", r, la, lb);
            }
        }
        printf("\n");
    }
    return 0;
}
