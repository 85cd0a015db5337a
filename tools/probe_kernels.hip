// probe_kernels.hip — hardware-semantics probes used while bringing the kernel up (not product code).
//   probe_tr16: what ds_read_b64_tr_b16 returns for lane-linear addresses over LDS filled with s[i] = i.
//   probe_mfma: checks the assumed A/B/C lane layouts of v_mfma_f32_32x32x16_bf16 against a host matmul.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe_tr16_kernel(const int* byte_addr, short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (unsigned short)i;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) s16x4*)((unsigned char*)s + byte_addr[threadIdx.x]);
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}

// A[32][16], B[16][32] row-major float inputs (small ints), C[32][32] out.
__global__ void probe_mfma_kernel(const float* A, const float* B, float* C) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[(l & 31) * 16 + (l >> 5) * 8 + e];      // A[i=l&31][k=8*(l>>5)+e]
        b[e] = (__bf16)B[((l >> 5) * 8 + e) * 32 + (l & 31)];    // B[k=8*(l>>5)+e][j=l&31]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        C[row * 32 + (l & 31)] = c[r];
    }
}

extern "C" int probe_tr16(const int* d_addr, short* d_out, void* stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_addr, d_out);
    return (int)hipGetLastError();
}
extern "C" int probe_mfma(const float* A, const float* B, float* C, void* stream) {
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C);
    return (int)hipGetLastError();
}
