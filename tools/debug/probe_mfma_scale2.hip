// GPU box probe 2: WHICH lane's E8M0 byte scales a given (lane, register) element of the B operand of v_mfma_scale_f32_32x32x64_f8f6f4?
// A = all ones; B = 0 except byte 0 of register R of lane L = 1.0; scale B: lanes 0-31 carry 2^-3, lanes 32-63 carry 2^0.
// D[.][L & 31] = the scale that was applied to that element: 0.125 -> the lower lane's byte, 1 -> the upper lane's.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int L, int R) {
    const int l = threadIdx.x;
    unsigned b[8];
    for (int r = 0; r < 8; ++r) b[r] = (l == L && r == R) ? 0x38u : 0u;
    unsigned scale_b = l < 32 ? 124u : 127u;
    float r0;
    asm volatile(
        "v_mov_b32 v16, 0x38383838\n v_mov_b32 v17, 0x38383838\n v_mov_b32 v18, 0x38383838\n v_mov_b32 v19, 0x38383838\n"
        "v_mov_b32 v20, 0x38383838\n v_mov_b32 v21, 0x38383838\n v_mov_b32 v22, 0x38383838\n v_mov_b32 v23, 0x38383838\n"
        "v_mov_b32 v24, 0x7f7f7f7f\n"
        "v_mov_b32 v32, %1\n v_mov_b32 v33, %2\n v_mov_b32 v34, %3\n v_mov_b32 v35, %4\n v_mov_b32 v36, %5\n v_mov_b32 v37, %6\n v_mov_b32 v38, %7\n v_mov_b32 v39, %8\n"
        "s_nop 4\n"
        "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[16:23], v[32:39], 0, v24, %9 op_sel_hi:[0,0,0]\n"
        "s_nop 15\n s_nop 15\n"
        "v_mov_b32 %0, v0\n"
        : "=v"(r0) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(scale_b)
        : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24",
          "v32","v33","v34","v35","v36","v37","v38","v39");
    out[l] = r0;
}
int main() {
    float* o; (void)hipMalloc(&o, 256);
    for (int L : {0, 5, 32, 37})
        for (int R = 0; R < 8; ++R) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, L, R);
            float r[64]; (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
            printf("element (lane %2d, reg %d): D[.][%d] = %g  -> scaled by the byte of the %s lane\n", L, R, L & 31, r[L & 31], r[L & 31] == 0.125f ? "LOWER" : (r[L & 31] == 1.f ? "UPPER" : "??"));
        }
    return 0;
}
