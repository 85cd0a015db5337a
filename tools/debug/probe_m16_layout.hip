// probe_m16_layout.hip — the index math of the 16x16x32 body (gen_fwd_x64_m16.py), checked end to end on ONE tile before any assembly:
//   S^T = K Q^T with v_mfma_f32_16x16x32_bf16 (A = K rows from the XOR-swizzled K image by ds_read_b128, B = Q fragments),
//   P = bf16(S) (no softmax here: the data path is what is probed), O^T = V^T P^T with A = V^T fragments by ds_read_b64_tr_b16
//   from the V image under the NEW 32-byte-granule swizzle (chunk ^= (row & 7) << 1), B = P straight from the S^T accumulators,
//   the transposing row reduction (permlane16_swap / permlane32_swap) and the pair-swap of the O^T store.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/debug/probe_m16_layout.hip -o /tmp/probe_m16 && /tmp/probe_m16
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int D = 128, BN = 64, QR = 64, ROW = 256;   // one wave: 64 query rows x one 64-key tile

__device__ inline unsigned short f2bf(float x) {      // RNE
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<unsigned short>(u >> 16);
}

__global__ void __launch_bounds__(64) probe(const unsigned short* q, const unsigned short* k, const unsigned short* v, float* s_out,
                                            float* o_out, float* rowmax_out, unsigned short* o_bf16) {
    __shared__ __attribute__((aligned(16))) unsigned char kimg[BN * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char vimg[BN * ROW];
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    // LDS images: K chunk c of row r at r*256 + ((c ^ (r & 15)) << 4); V chunk c of row r at r*256 + ((c ^ ((r & 7) << 1)) << 4)
    for (int idx = lane; idx < BN * 16; idx += 64) {
        const int r = idx >> 4, c = idx & 15;
        const uint4 kk = *reinterpret_cast<const uint4*>(k + r * D + c * 8);
        const uint4 vv = *reinterpret_cast<const uint4*>(v + r * D + c * 8);
        *reinterpret_cast<uint4*>(kimg + r * ROW + ((c ^ (r & 15)) << 4)) = kk;
        *reinterpret_cast<uint4*>(vimg + r * ROW + ((c ^ ((r & 7) << 1)) << 4)) = vv;
    }
    __syncthreads();
    // Q fragments (B operand): lane (j, g): query 16 qb + j, d = 32 ks + 8 g .. + 7
    bf16x8 qf[4][4];
    for (int qb = 0; qb < 4; ++qb)
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = *reinterpret_cast<const bf16x8*>(q + (16 * qb + j) * D + 32 * ks + 8 * g);
    // S^T(qb, kb) = sum_ks K(kb, ks) Q(qb, ks): lane (j, g) gets keys 16 kb + 4 g + r of query 16 qb + j
    f32x4 s[4][4];
    for (int qb = 0; qb < 4; ++qb)
        for (int kb = 0; kb < 4; ++kb) s[qb][kb] = f32x4{0, 0, 0, 0};
    for (int ks = 0; ks < 4; ++ks)
        for (int kb = 0; kb < 4; ++kb) {
            // K fragment (A operand): lane (i = j, g): key 16 kb + i, chunk 4 ks + g, swizzled by the row (row & 15 = i)
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kimg + (16 * kb + j) * ROW + (((4 * ks + g) ^ j) << 4));
            for (int qb = 0; qb < 4; ++qb) s[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], s[qb][kb], 0, 0, 0);
        }
    for (int qb = 0; qb < 4; ++qb)
        for (int kb = 0; kb < 4; ++kb)
            for (int r = 0; r < 4; ++r) s_out[(16 * qb + j) * BN + 16 * kb + 4 * g + r] = s[qb][kb][r];
    // transposing row max: X[qb] = in-lane max of the lane's 16 keys of query (qb, j); result lane (j, g) = full max of query 16 g + j
    float x[4];
    for (int qb = 0; qb < 4; ++qb) {
        float m = -INFINITY;
        for (int kb = 0; kb < 4; ++kb)
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[qb][kb][r]);
        x[qb] = m;
    }
    {
        auto sw16 = [](float& a, float& b) {          // v_permlane16_swap a, b: a's odd 16-lane rows <-> b's even rows
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
        };
        auto sw32 = [](float& a, float& b) {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
        };
        sw16(x[0], x[1]); float A = fmaxf(x[0], x[1]);
        sw16(x[2], x[3]); float B = fmaxf(x[2], x[3]);
        sw32(A, B);
        rowmax_out[lane] = fmaxf(A, B);               // query row 16 g + j = lane
    }
    // P = bf16(S), compacted as the body does: B operand (qb, kk) = {S(qb, 2 kk)[0..3], S(qb, 2 kk + 1)[0..3]} as 8 bf16:
    // k-slot e < 4 = key 32 kk + 4 g + e, e >= 4 = key 32 kk + 16 + 4 g + (e - 4)
    f32x4 o[4][8];
    for (int qb = 0; qb < 4; ++qb)
        for (int db = 0; db < 8; ++db) o[qb][db] = f32x4{0, 0, 0, 0};
    for (int kk = 0; kk < 2; ++kk)
        for (int db = 0; db < 8; ++db) {
            // V^T fragment (A operand): lane (i = j -> d = 16 db + j, g): k-slots as above. Two transpose reads: the 16 lanes of group g
            // read the 4 x 16 block rows 32 kk + 4 g + (t >> 2) (+ 16), cols 16 db + 4 (t & 3) .. + 3; lane t receives column t.
            const int t = j, rrow = 4 * g + (t >> 2);
            const int chunk = 2 * db + ((t & 3) >> 1), within = ((t & 3) & 1) * 8;
            const unsigned char* a0 = vimg + (32 * kk + rrow) * ROW + ((chunk ^ ((rrow & 7) << 1)) << 4) + within;
            const unsigned char* a1 = a0 + 16 * ROW;      // + 16 keys: the same (row & 7), the same swizzle
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
            union { bf16x8 v; short h[8]; } vf;
            for (int e = 0; e < 4; ++e) { vf.h[e] = lo[e]; vf.h[4 + e] = hi[e]; }
            for (int qb = 0; qb < 4; ++qb) {
                union { bf16x8 v; unsigned short h[8]; } pf;
                for (int e = 0; e < 4; ++e) { pf.h[e] = f2bf(s[qb][2 * kk][e]); pf.h[4 + e] = f2bf(s[qb][2 * kk + 1][e]); }
                o[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[qb][db], 0, 0, 0);
            }
        }
    // O^T(qb, db): lane (j, g): query 16 qb + j, d = 16 db + 4 g + r
    for (int qb = 0; qb < 4; ++qb)
        for (int db = 0; db < 8; ++db)
            for (int r = 0; r < 4; ++r) o_out[(16 * qb + j) * D + 16 * db + 4 * g + r] = o[qb][db][r];
    // the store path of the epilogue: per q-block, pairs of d-blocks (a = 2 p, b = 2 p + 1); bf16 pack (2 regs of 2 bf16 each), then
    // v_permlane16_swap per register between X = block a and Y = block b: even groups keep block a (own + the odd partner's 8 bytes:
    // 16 contiguous bytes at d = 16 a + 4 g), odd groups block b (the even partner's + own: 16 bytes at d = 16 b + 4 (g - 1))
    for (int qb = 0; qb < 4; ++qb)
        for (int p = 0; p < 4; ++p) {
            unsigned X[2], Y[2];
            for (int h = 0; h < 2; ++h) {
                X[h] = f2bf(o[qb][2 * p][2 * h]) | (static_cast<unsigned>(f2bf(o[qb][2 * p][2 * h + 1])) << 16);
                Y[h] = f2bf(o[qb][2 * p + 1][2 * h]) | (static_cast<unsigned>(f2bf(o[qb][2 * p + 1][2 * h + 1])) << 16);
            }
            for (int h = 0; h < 2; ++h) {
                auto r = __builtin_amdgcn_permlane16_swap(X[h], Y[h], false, false);
                X[h] = r[0]; Y[h] = r[1];
            }
            const int dcol = (g & 1) ? 16 * (2 * p + 1) + 4 * (g - 1) : 16 * (2 * p) + 4 * g;
            uint4 w = {X[0], X[1], Y[0], Y[1]};
            *reinterpret_cast<uint4*>(o_bf16 + (16 * qb + j) * D + dcol) = w;
        }
}

static float bf2f(unsigned short h) { unsigned u = static_cast<unsigned>(h) << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf_h(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return static_cast<unsigned short>(u >> 16); }

int main() {
    std::vector<unsigned short> q(QR * D), k(BN * D), v(BN * D);
    srand(1);
    auto rnd = []() { return f2bf_h((rand() % 2001 - 1000) / 500.0f); };
    for (auto& x : q) x = rnd();
    for (auto& x : k) x = rnd();
    for (auto& x : v) x = rnd();
    unsigned short *dq, *dk, *dv, *dob;
    float *ds, *dout, *dm;
    hipMalloc(&dq, q.size() * 2); hipMalloc(&dk, k.size() * 2); hipMalloc(&dv, v.size() * 2); hipMalloc(&dob, QR * D * 2);
    hipMalloc(&ds, QR * BN * 4); hipMalloc(&dout, QR * D * 4); hipMalloc(&dm, 64 * 4);
    hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dk, k.data(), k.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dv, v.data(), v.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dob, 0xff, QR * D * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dq, dk, dv, ds, dout, dm, dob);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> s(QR * BN), o(QR * D), m(64);
    std::vector<unsigned short> ob(QR * D);
    hipMemcpy(s.data(), ds, s.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), dm, m.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ob.data(), dob, ob.size() * 2, hipMemcpyDeviceToHost);
    double es = 0, eo = 0, em = 0, eb = 0;
    std::vector<float> sref(QR * BN);
    for (int i = 0; i < QR; ++i)
        for (int n = 0; n < BN; ++n) {
            float acc = 0;
            for (int d = 0; d < D; ++d) acc += bf2f(q[i * D + d]) * bf2f(k[n * D + d]);
            sref[i * BN + n] = acc;
            es = fmax(es, fabs(acc - s[i * BN + n]));
        }
    for (int i = 0; i < QR; ++i) {
        float mx = -INFINITY;
        for (int n = 0; n < BN; ++n) mx = fmaxf(mx, s[i * BN + n]);
        em = fmax(em, fabs(mx - m[i]));
        for (int d = 0; d < D; ++d) {
            float acc = 0;
            for (int n = 0; n < BN; ++n) acc += bf2f(f2bf_h(s[i * BN + n])) * bf2f(v[n * D + d]);
            eo = fmax(eo, fabs(acc - o[i * D + d]) / (1.0 + fabs(acc)));
            eb = fmax(eb, fabs(bf2f(f2bf_h(o[i * D + d])) - bf2f(ob[i * D + d])));
        }
    }
    printf("max |S - ref| = %.3g   max rel |O - ref| = %.3g   max |rowmax - ref| = %.3g   max |O_bf16 store - bf16(O)| = %.3g\n", es, eo, em, eb);
    const bool ok = es < 1e-2 && eo < 1e-3 && em == 0.0 && eb == 0.0;
    printf(ok ? "PROBE OK\n" : "PROBE FAILED\n");
    return ok ? 0 : 1;
}
