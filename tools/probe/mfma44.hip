// Probe of v_mfma_f32_4x4x1_16b_f32's operand layout: A = lane id as float * 1, B = 1 for one lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    // test 1: a = 100 + l, b = (l == sel) ? 1 : 0 for a few sel; record D for all lanes / regs
    for (int sel = 0; sel < 64; ++sel) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(100.f + l, l == sel ? 1.f : 0.f, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(sel * 64 + l) * 4 + r] = c[r];
    }
}
int main() {
    float* d; hipMalloc(&d, 64 * 64 * 4 * sizeof(float));
    probe<<<1, 64>>>(d);
    static float h[64 * 64 * 4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // for each sel (B lane), list which (lane, reg) are non-zero and their values (= A lane + 100)
    for (int sel : {0, 1, 4, 5, 17, 63}) {
        printf("B lane %d ->", sel);
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                if (h[(sel * 64 + l) * 4 + r] != 0.f) printf(" D[lane %d][reg %d]=A(lane %d)", l, r, (int)h[(sel * 64 + l) * 4 + r] - 100);
        printf("\n");
    }
    return 0;
}
