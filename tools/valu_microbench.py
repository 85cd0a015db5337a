#!/usr/bin/env python
"""VALU issue-cost microbenchmark for gfx950, ONE wave per SIMD (the occupancy of the hand-scheduled x64 kernels).

    python tools/valu_microbench.py build      # here: writes + compiles build_variants/valu_microbench (hipcc, gfx950)
    build_variants/valu_microbench             # on the GPU box: prints one line per case

Every case is a loop of 64 independent VALU instructions (sources v0..v31, destinations v64..v127) run by 256 workgroups of
4 waves with 100 KiB of LDS each (one workgroup per CU), optionally with one MFMA in front of every `per_mfma` instructions.
Reported: shader cycles per instruction (s_memtime) and the effective clock (cycles / wall time). What it answers: the real
issue cost of the softmax instructions (v_exp_f32 vs v_exp_f16, packed fp32 / fp16 forms, the fp8 / f16 converts, v_dot2 row
sums) alone and under MFMAs — the numbers the fp8 softmax redesign of round 3 is priced with (HISTORY.md section 4.2).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(ROOT, "build_variants")

MFMA_F8 = "v_mfma_scale_f32_32x32x64_f8f6f4 a[{a}:{b}], v[32:39], v[40:47], a[{a}:{b}], v48, v48 op_sel_hi:[0,0,0]"
MFMA_BF16 = "v_mfma_f32_32x32x16_bf16 a[{a}:{b}], v[32:35], v[40:43], a[{a}:{b}]"


def case(name, templates, mfma=None, per_mfma=0):
    """templates: list of format strings with {d} (dst), {d2} (even-aligned 64-bit dst), {s0} {s1} {s2} (sources), {p0} {p1} (64-bit sources)."""
    lines = []
    n = 64
    m = 0
    for i in range(n):
        if mfma and per_mfma and i % per_mfma == 0:
            a = 16 * (m % 8)
            lines.append(mfma.format(a=a, b=a + 15))
            m += 1
        t = templates[i % len(templates)]
        d = 64 + (i % 60)
        d2 = 64 + 2 * (i % 30)
        s0, s1, s2 = i % 28, (i + 7) % 28, (i + 13) % 28
        p0, p1 = 2 * (i % 14), 2 * ((i + 5) % 14)
        lines.append(t.format(d=f"v{d}", d2=f"v[{d2}:{d2 + 1}]", s0=f"v{s0}", s1=f"v{s1}", s2=f"v{s2}",
                              p0=f"v[{p0}:{p0 + 1}]", p1=f"v[{p1}:{p1 + 1}]", acc=f"v{96 + (i % 8)}"))
    return name, lines, n


FP8_NOW = ["v_fma_f32 {d}, {s0}, s20, {s1}", "v_fma_f32 {d}, {s1}, s20, {s2}", "v_exp_f32 {d}, {s0}", "v_exp_f32 {d}, {s1}",
           "v_add_f32 {acc}, {acc}, {s0}", "v_add_f32 {acc}, {acc}, {s1}", "v_cvt_pk_fp8_f32 {d}, {s0}, {s1}"]
FP8_F16 = ["v_fma_f32 {d}, {s0}, s20, {s1}", "v_fma_f32 {d}, {s1}, s20, {s2}", "v_cvt_pk_f16_f32 {d}, {s0}, {s1}",
           "v_exp_f16 {d}, {s0}", "v_exp_f16_sdwa {d}, {s1} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1",
           "v_dot2_f32_f16 {acc}, {s0}, v49, {acc}", "v_cvt_scalef32_pk_fp8_f16 {d}, {s0}, v50"]
FP8_F16B = ["v_fma_f32 {d}, {s0}, s20, {s1}", "v_fma_f32 {d}, {s1}, s20, {s2}", "v_cvt_pk_f16_f32 {d}, {s0}, {s1}",
            "v_exp_f16 {d}, {s0}", "v_exp_f16 {d}, {s1}", "v_pack_b32_f16 {d}, {s0}, {s1}",
            "v_dot2_f32_f16 {acc}, {s0}, v49, {acc}", "v_cvt_scalef32_pk_fp8_f16 {d}, {s0}, v50"]

FP8_LIN = ["v_fma_f32 {d}, {s0}, s20, {s1}", "v_fma_f32 {d}, {s1}, s20, {s2}", "v_cvt_pk_u8_f32 {acc}, {s0}, 0, {acc}",
           "v_cvt_pk_u8_f32 {acc}, {s1}, 1, {acc}"]
FP8_PK = ["v_pk_fma_f32 {d2}, {p0}, s[22:23], {p1} op_sel_hi:[1,0,1]", "v_exp_f32 {d}, {s0}", "v_exp_f32 {d}, {s1}",
          "v_pk_add_f32 v[96:97], v[96:97], {p0}", "v_cvt_pk_fp8_f32 {d}, {s0}, {s1}"]
FP8_PK_V = ["v_pk_fma_f32 {d2}, {p0}, v[52:53], {p1}", "v_exp_f32 {d}, {s0}", "v_exp_f32 {d}, {s1}",
            "v_pk_add_f32 v[96:97], v[96:97], {p0}", "v_cvt_pk_fp8_f32 {d}, {s0}, {s1}"]
# dependent forms: the consumer reads what the producer just wrote (register 64 / 65 carried around the loop)
DEP_FMA_EXP = ["v_fma_f32 v64, {s0}, s20, {s1}", "v_exp_f32 v65, v64"]
DEP_PKFMA_EXP = ["v_pk_fma_f32 v[64:65], {p0}, s[22:23], {p1} op_sel_hi:[1,0,1]", "v_exp_f32 v66, v64", "v_exp_f32 v67, v65"]
DEP_FMA2_EXP = ["v_fma_f32 v64, {s0}, s20, {s1}", "v_fma_f32 v65, {s1}, s20, {s2}", "v_exp_f32 v66, v64", "v_exp_f32 v67, v65"]
DEP_EXP_PKADD = ["v_exp_f32 v64, {s0}", "v_exp_f32 v65, {s1}", "v_pk_add_f32 v[96:97], v[96:97], v[64:65]"]
DEP_EXP_ADD2 = ["v_exp_f32 v64, {s0}", "v_exp_f32 v65, {s1}", "v_add_f32 v96, v96, v64", "v_add_f32 v97, v97, v65"]

CASES = [
    case("v_fma_f32", ["v_fma_f32 {d}, {s0}, {s1}, {s2}"]),
    case("v_fma_f32 (sgpr)", ["v_fma_f32 {d}, {s0}, s20, {s2}"]),
    case("v_add_f32", ["v_add_f32 {d}, {s0}, {s1}"]),
    case("v_exp_f32", ["v_exp_f32 {d}, {s0}"]),
    case("v_exp_f16", ["v_exp_f16 {d}, {s0}"]),
    case("v_exp_f16_sdwa hi", ["v_exp_f16_sdwa {d}, {s0} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"]),
    case("v_log_f32", ["v_log_f32 {d}, {s0}"]),
    case("v_rcp_f32", ["v_rcp_f32 {d}, {s0}"]),
    case("v_pk_fma_f32", ["v_pk_fma_f32 {d2}, {p0}, {p1}, {p0}"]),
    case("v_pk_add_f32", ["v_pk_add_f32 {d2}, {p0}, {p1}"]),
    case("v_pk_mul_f32", ["v_pk_mul_f32 {d2}, {p0}, {p1}"]),
    case("v_pk_fma_f16", ["v_pk_fma_f16 {d}, {s0}, {s1}, {s2}"]),
    case("v_pk_add_f16", ["v_pk_add_f16 {d}, {s0}, {s1}"]),
    case("v_pk_max_f16", ["v_pk_max_f16 {d}, {s0}, {s1}"]),
    case("v_dot2_f32_f16", ["v_dot2_f32_f16 {d}, {s0}, {s1}, {s2}"]),
    case("v_dot2c_f32_f16", ["v_dot2c_f32_f16 {d}, {s0}, {s1}"]),
    case("v_dot2_f32_bf16", ["v_dot2_f32_bf16 {d}, {s0}, {s1}, {s2}"]),
    case("v_cvt_pk_fp8_f32", ["v_cvt_pk_fp8_f32 {d}, {s0}, {s1}"]),
    case("v_cvt_scalef32_pk_fp8_f16", ["v_cvt_scalef32_pk_fp8_f16 {d}, {s0}, {s1}"]),
    case("v_cvt_scalef32_pk_fp8_f32", ["v_cvt_scalef32_pk_fp8_f32 {d}, {s0}, {s1}, {s2}"]),
    case("v_cvt_pk_f16_f32", ["v_cvt_pk_f16_f32 {d}, {s0}, {s1}"]),
    case("v_cvt_pk_bf16_f32", ["v_cvt_pk_bf16_f32 {d}, {s0}, {s1}"]),
    case("v_cvt_f16_f32", ["v_cvt_f16_f32 {d}, {s0}"]),
    case("v_pack_b32_f16", ["v_pack_b32_f16 {d}, {s0}, {s1}"]),
    case("v_max3_f32", ["v_max3_f32 {d}, {s0}, {s1}, {s2}"]),
    case("v_maximum3_f32", ["v_maximum3_f32 {d}, {s0}, {s1}, {s2}"]),
    case("v_pk_maximum3_f16", ["v_pk_maximum3_f16 {d}, {s0}, {s1}, {s2}"]),
    case("v_cvt_i32_f32", ["v_cvt_i32_f32 {d}, {s0}"]),
    case("v_ldexp_f32", ["v_ldexp_f32 {d}, {s0}, {s1}"]),
    case("v_perm_b32", ["v_perm_b32 {d}, {s0}, {s1}, {s2}"]),
    case("v_mov_b32", ["v_mov_b32 {d}, {s0}"]),
    case("exp_f32 : fma 1:1", ["v_exp_f32 {d}, {s0}", "v_fma_f32 {d}, {s0}, {s1}, {s2}"]),
    case("exp_f32 : fma 1:2", ["v_exp_f32 {d}, {s0}", "v_fma_f32 {d}, {s0}, {s1}, {s2}", "v_add_f32 {d}, {s0}, {s1}"]),
    case("exp_f32 : fma 1:3", ["v_exp_f32 {d}, {s0}", "v_fma_f32 {d}, {s0}, {s1}, {s2}", "v_add_f32 {d}, {s0}, {s1}", "v_mul_f32 {d}, {s0}, {s1}"]),
    case("exp_f16 : fma 1:1", ["v_exp_f16 {d}, {s0}", "v_fma_f32 {d}, {s0}, {s1}, {s2}"]),
    case("exp_f16 : fma 1:3", ["v_exp_f16 {d}, {s0}", "v_fma_f32 {d}, {s0}, {s1}, {s2}", "v_add_f32 {d}, {s0}, {s1}", "v_mul_f32 {d}, {s0}, {s1}"]),
    case("fp8 softmax group, as built (7 per 2 scores)", FP8_NOW),
    case("fp8 softmax group, f16 + sdwa (7 per 2 scores)", FP8_F16),
    case("fp8 softmax group, f16 + pack (8 per 2 scores)", FP8_F16B),
    case("fp8 softmax group, pk fma/add sgpr c (5 per 2 scores)", FP8_PK),
    case("fp8 softmax group, pk fma/add vgpr c (5 per 2 scores)", FP8_PK_V),
    case("chain v_fma_f32 -> v_fma_f32", ["v_fma_f32 v64, v64, {s0}, {s1}"]),
    case("chain v_pk_fma_f32 -> v_pk_fma_f32", ["v_pk_fma_f32 v[64:65], v[64:65], {p0}, {p1}"]),
    case("chain v_add_f32", ["v_add_f32 v64, v64, {s0}"]),
    case("chain v_pk_add_f32", ["v_pk_add_f32 v[64:65], v[64:65], {p0}"]),
    case("chain v_exp_f32", ["v_exp_f32 v64, v64"]),
    case("dep fma -> exp (2 per score)", DEP_FMA_EXP),
    case("dep fma, fma -> exp, exp (4 per 2)", DEP_FMA2_EXP),
    case("dep pk_fma -> exp, exp (3 per 2)", DEP_PKFMA_EXP),
    case("dep exp, exp -> add, add (4 per 2)", DEP_EXP_ADD2),
    case("dep exp, exp -> pk_add (3 per 2)", DEP_EXP_PKADD),
    case("pk group sgpr c + 1 fp8 MFMA per 25", FP8_PK, MFMA_F8, 25),
    case("pk group vgpr c + 1 fp8 MFMA per 25", FP8_PK_V, MFMA_F8, 25),
    case("pk group sgpr c + 1 fp8 MFMA per 15", FP8_PK, MFMA_F8, 15),
    case("as-built group + 1 fp8 MFMA per 14", FP8_NOW, MFMA_F8, 14),
    case("v_pk_fma_f32 + 1 fp8 MFMA per 16", ["v_pk_fma_f32 {d2}, {p0}, {p1}, {p0}"], MFMA_F8, 16),
    case("v_fma_f32 + 1 fp8 MFMA per 16", ["v_fma_f32 {d}, {s0}, {s1}, {s2}"], MFMA_F8, 16),
    case("v_pk_fma_f32 + 1 bf16 MFMA per 8", ["v_pk_fma_f32 {d2}, {p0}, {p1}, {p0}"], MFMA_BF16, 8),
    case("v_fma_f32 + 1 bf16 MFMA per 8", ["v_fma_f32 {d}, {s0}, {s1}, {s2}"], MFMA_BF16, 8),
    case("v_dot2_f32_bf16 + 1 bf16 MFMA per 8", ["v_dot2_f32_bf16 {d}, {s0}, {s1}, {s2}"], MFMA_BF16, 8),
    case("v_dot2c_f32_bf16 (VOP2) + 1 bf16 MFMA per 8", ["v_dot2c_f32_bf16 {d}, {s0}, {s1}"], MFMA_BF16, 8),
    case("v_dot2c_f32_f16 (VOP2) + 1 bf16 MFMA per 8", ["v_dot2c_f32_f16 {d}, {s0}, {s1}"], MFMA_BF16, 8),
    case("v_add_f32 + 1 bf16 MFMA per 8", ["v_add_f32 {d}, {s0}, {s1}"], MFMA_BF16, 8),
    case("v_cvt_pk_bf16_f32 + 1 bf16 MFMA per 8", ["v_cvt_pk_bf16_f32 {d}, {s0}, {s1}"], MFMA_BF16, 8),
    case("v_exp_f32 + 1 bf16 MFMA per 8", ["v_exp_f32 {d}, {s0}"], MFMA_BF16, 8),
    case("v_max3_f32 + 1 bf16 MFMA per 8", ["v_max3_f32 {d}, {s0}, {s1}, {s2}"], MFMA_BF16, 8),
    case("v_pk_add_f16 + 1 bf16 MFMA per 8", ["v_pk_add_f16 {d}, {s0}, {s1}"], MFMA_BF16, 8),
    case("v_perm_b32 + 1 bf16 MFMA per 8", ["v_perm_b32 {d}, {s0}, {s1}, {s2}"], MFMA_BF16, 8),
    case("v_cvt_pk_fp8_f32 + 1 fp8 MFMA per 16", ["v_cvt_pk_fp8_f32 {d}, {s0}, {s1}"], MFMA_F8, 16),
    case("fp8 MFMA only (8 per 64 slots)", ["s_nop 0"], MFMA_F8, 8),
    case("fp8 group as built + 1 fp8 MFMA per 28", FP8_NOW, MFMA_F8, 28),
    case("fp8 group f16+sdwa + 1 fp8 MFMA per 28", FP8_F16, MFMA_F8, 28),
    case("fp8 group as built + 1 fp8 MFMA per 21", FP8_NOW, MFMA_F8, 21),
    case("fp8 group f16+sdwa + 1 fp8 MFMA per 21", FP8_F16, MFMA_F8, 21),
    case("bf16 MFMA only (8 per 64 slots)", ["s_nop 0"], MFMA_BF16, 8),
    case("fp8 group as built + 1 bf16 MFMA per 7", FP8_NOW, MFMA_BF16, 7),
    # round 3: the log-linear e4m3 encoding of P (gen_fwd_x64_fp8.py "lin"): one FMA + one byte convert per score, no v_exp_f32
    case("v_cvt_pk_u8_f32", ["v_cvt_pk_u8_f32 {d}, {s0}, 1, {s1}"]),
    case("v_cvt_pk_u8_f32 chained x4", ["v_cvt_pk_u8_f32 {acc}, {s0}, 0, {acc}", "v_cvt_pk_u8_f32 {acc}, {s1}, 1, {acc}",
                                        "v_cvt_pk_u8_f32 {acc}, {s2}, 2, {acc}", "v_cvt_pk_u8_f32 {acc}, {s0}, 3, {acc}"]),
    case("v_cvt_pk_u8_f32 + 1 fp8 MFMA per 16", ["v_cvt_pk_u8_f32 {d}, {s0}, 1, {s1}"], MFMA_F8, 16),
    case("lin group (fma, fma, cvt_u8, cvt_u8)", FP8_LIN),
    case("lin group + 1 fp8 MFMA per 16", FP8_LIN, MFMA_F8, 16),
    case("lin group + 1 fp8 MFMA per 12", FP8_LIN, MFMA_F8, 12),
    case("lin group + 1 fp8 MFMA per 10", FP8_LIN, MFMA_F8, 10),
    case("lin group + 1 fp8 MFMA per 8", FP8_LIN, MFMA_F8, 8),
    case("fp8 group as built + 1 fp8 MFMA per 16", FP8_NOW, MFMA_F8, 16),
]


def source():
    ks, calls = [], []
    for idx, (name, lines, n) in enumerate(CASES):
        body = "\\n".join(lines)
        ks.append(f'''
__global__ void __launch_bounds__(256, 1) k{idx}(unsigned long long* out, int iters) {{
    extern __shared__ unsigned char smem[];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\\ns_waitcnt lgkmcnt(0)" : "=s"(t0));
    asm volatile(
        "s_mov_b32 s21, %0\\n"
        "s_mov_b32 s20, 0x3f000000\\ns_mov_b32 s22, 0x3f000000\\ns_mov_b32 s23, 0x3f000000\\nv_mov_b32 v52, 0x3f000000\\nv_mov_b32 v53, 0x3f000000\\n"
        "v_mov_b32 v48, 0x7f7f7f7f\\nv_mov_b32 v49, 0x3c003c00\\nv_mov_b32 v50, 1.0\\n"
        INIT
        "1:\\n"
        "{body}\\n"
        "s_sub_u32 s21, s21, 1\\n"
        "s_cmp_lg_u32 s21, 0\\n"
        "s_cbranch_scc1 1b\\n"
        "s_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15\\n"
        : : "s"(iters) : CLOB);
    asm volatile("s_memtime %0\\ns_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (iters < 0) smem[threadIdx.x] = 1;
}}''')
        calls.append(f'    run("{name}", k{idx}, {n});')
    init = "".join(f'"v_mov_b32 v{r}, 0x3c003c{r:02x}\\n"' for r in range(0, 32)) + "".join(
        f'"v_mov_b32 v{r}, 0\\n"' for r in range(64, 128)) + "".join(f'"v_accvgpr_write_b32 a{r}, 0\\n"' for r in range(128))
    clob = ", ".join(f'"v{r}"' for r in range(128)) + ", " + ", ".join(f'"a{r}"' for r in range(128)) + ', "s20", "s21", "s22", "s23", "scc", "vcc", "memory"'
    return f'''// GENERATED by tools/valu_microbench.py — VALU issue-cost microbenchmark (gfx950, one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define INIT {init}
#define CLOB {clob}
{"".join(ks)}

template <typename K>
static void run(const char* name, K kern, int n_instr) {{
    const int grid = 256, iters = 20000, lds = 100 * 1024;
    unsigned long long* d;
    hipMalloc(&d, grid * sizeof(unsigned long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, d, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto x : h) cyc += double(x);
    cyc /= grid;
    // s_memtime counts at a fixed 100 MHz on gfx9: convert through wall time instead. cycles/instr needs the shader clock, which
    // the fixed counter cannot give; report ns per instruction and instructions per microsecond per wave, and the ratio to v_fma_f32.
    const double ns_per = double(ms) * 1e6 / (double(iters) * n_instr);
    static double base = 0;
    if (base == 0) base = ns_per;
    printf("%-58s %8.3f ms  %6.3f ns/instr  %6.2f ticks/instr  %6.1f ticks/64  clk %.2f GHz\\n", name, ms, ns_per, cyc / (double(iters) * n_instr), cyc / double(iters), cyc / (double(ms) * 1e6));
    hipFree(d);
}}

static void probe();
int main(int argc, char** argv) {{
    if (argc > 1 && argv[1][0] == 'p') {{ probe(); return 0; }}
    const char* only = argc > 1 ? argv[1] : nullptr;
#define run(name, k, n) if (!only || strstr(name, only)) run(name, k, n)
{chr(10).join(calls)}
#undef run
    return 0;
}}
''' + PROBE


PROBE = r"""
__global__ void probe_k(const float* x, unsigned* y, int n) {
    int i = threadIdx.x;
    if (i < n) {
        unsigned r = 0xAABBCCDDu;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(x[i]));
        y[i] = r;
    }
}
static void probe() {
    const float xs[] = {0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 119.5f, 120.5f, 254.4f, 254.5f, 255.5f, 300.f, 1e9f, -0.4f, -0.6f, -5.f,
                        -INFINITY, INFINITY, NAN, 103.54f, 7.999f};
    const int n = sizeof(xs) / sizeof(xs[0]);
    float* dx; unsigned* dy;
    hipMalloc(&dx, sizeof(xs)); hipMalloc(&dy, n * 4);
    hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_k, dim3(1), dim3(64), 0, 0, dx, dy, n);
    unsigned ys[64];
    hipMemcpy(ys, dy, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) byte1 -> 0x%08x (%u)\n", xs[i], ys[i], (ys[i] >> 8) & 255);
}
"""


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    src = os.path.join(OUT_DIR, "valu_microbench.hip")
    with open(src, "w") as f:
        f.write(source())
    exe = os.path.join(OUT_DIR, "valu_microbench")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, src], check=True)
    print("built", exe)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        print(__doc__)
