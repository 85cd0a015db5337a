// GPU box probe: which elements does a lane's E8M0 scale byte of v_mfma_scale_f32_32x32x64_f8f6f4 apply to?
// Hypothesis (MX block scaling, block = 32 along K): lane l of the B operand holds column n = l & 31, contraction block l >> 5 (32 bytes),
// and ITS scale byte multiplies exactly those 32 elements. A = B = all ones (e4m3 0x38), scale A = 2^0, scale B per lane = 2^(e(l) - 127).
// Then D[m][n] = 32 * 2^(e(n) - 127) + 32 * 2^(e(n + 32) - 127) for every m.
//   hipcc --offload-arch=gfx950 -O2 probe_mfma_scale.hip -o probe_mfma_scale && ./probe_mfma_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(float* out, const unsigned* sb) {
    const int l = threadIdx.x;
    unsigned scale_b = sb[l];
    float r0, r5, r15;
    asm volatile(
        "v_mov_b32 v16, 0x38383838\n v_mov_b32 v17, 0x38383838\n v_mov_b32 v18, 0x38383838\n v_mov_b32 v19, 0x38383838\n"
        "v_mov_b32 v20, 0x38383838\n v_mov_b32 v21, 0x38383838\n v_mov_b32 v22, 0x38383838\n v_mov_b32 v23, 0x38383838\n"
        "v_mov_b32 v24, 0x7f7f7f7f\n s_nop 4\n"
        "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[16:23], v[16:23], 0, v24, %3 op_sel_hi:[0,0,0]\n"
        "s_nop 15\n s_nop 15\n"
        "v_mov_b32 %0, v0\n v_mov_b32 %1, v5\n v_mov_b32 %2, v15\n"
        : "=v"(r0), "=v"(r5), "=v"(r15) : "v"(scale_b)
        : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24");
    out[l * 3] = r0; out[l * 3 + 1] = r5; out[l * 3 + 2] = r15;
}
int main() {
    unsigned h[64]; for (int l = 0; l < 64; ++l) h[l] = 120 + (l % 7) + 8 * (l / 32) + ((l * 2654435761u) & 0xffffff00u);   // junk in the upper bytes: only byte 0 is used
    unsigned* d; float* o; hipMalloc(&d, 256); hipMalloc(&o, 64 * 12);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d);
    float r[192]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int n = l & 31;
        const float want = 32.f * ldexpf(1.f, (int)(h[n] & 255) - 127) + 32.f * ldexpf(1.f, (int)(h[n + 32] & 255) - 127);
        if (r[3 * l] != want || r[3 * l + 1] != want || r[3 * l + 2] != want) { ++bad; if (bad < 6) printf("lane %d: got %g %g %g want %g\n", l, r[3*l], r[3*l+1], r[3*l+2], want); }
    }
    printf("per-lane B scale (column l & 31, k-block l >> 5): %s (%d lanes differ)\n", bad ? "NO" : "CONFIRMED", bad);
    return 0;
}
