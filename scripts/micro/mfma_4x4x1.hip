// Microbenchmark / layout probe for v_mfma_f32_4x4x1_16b_f32 on gfx950 (the row-per-lane scan kernel's instruction).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_4x4x1 mfma_4x4x1.hip && ./mfma_4x4x1
// Prints (1) which (A lane, B lane) pair feeds every (lane, vgpr) of D, (2) cycles per instruction for 1 / 2 / 4 independent
// accumulator chains next to v_mfma_f32_16x16x4_f32, (3) whether a 128-step chain is bit-equal to an fmaf chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(float *out) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 + l), (float)(1000 + l), acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) out[l * 4 + i] = acc[i];
}

// A-matrix broadcast (CBSZ = 4: the A block named by ABID feeds all 16 blocks)
template <int ABID>
__global__ void k_layout_bc(float *out) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 + l), (float)(1000 + l), acc, 4, ABID, 0);
    for (int i = 0; i < 4; i++) out[l * 4 + i] = acc[i];
}

template <int C0>
__device__ __forceinline__ void bc16(f32x4 &acc, const float qa, const float *row) {
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 0], acc, 4, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 1], acc, 4, 1, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 2], acc, 4, 2, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 3], acc, 4, 3, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 4], acc, 4, 4, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 5], acc, 4, 5, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 6], acc, 4, 6, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 7], acc, 4, 7, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 8], acc, 4, 8, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 9], acc, 4, 9, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 10], acc, 4, 10, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 11], acc, 4, 11, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 12], acc, 4, 12, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 13], acc, 4, 13, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 14], acc, 4, 14, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(qa, row[C0 + 15], acc, 4, 15, 0);
}

// queries as the (broadcast) A operand: lane (b, i) register c holds q_i[16c + b]; rows as B, one row per lane:
// D[lane l][vgpr i] = q_i . row_l, chain in column order
__global__ void k_exact_bc(const float *rows /*[64][128]*/, const float *qs /*[4][128]*/, float *out_mfma, float *out_ref) {
    const int l = threadIdx.x, b = l >> 2, i = l & 3;
    float qa[8], row[128];
    for (int c = 0; c < 8; c++) qa[c] = qs[i * 128 + 16 * c + b];
    for (int k = 0; k < 128; k++) row[k] = rows[l * 128 + k];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bc16<0>(acc, qa[0], row);
    bc16<16>(acc, qa[1], row);
    bc16<32>(acc, qa[2], row);
    bc16<48>(acc, qa[3], row);
    bc16<64>(acc, qa[4], row);
    bc16<80>(acc, qa[5], row);
    bc16<96>(acc, qa[6], row);
    bc16<112>(acc, qa[7], row);
    for (int q = 0; q < 4; q++) {
        out_mfma[l * 4 + q] = acc[q];
        float r = 0.f;
        for (int k = 0; k < 128; k++) r = __fmaf_rn(rows[l * 128 + k], qs[q * 128 + k], r);
        out_ref[l * 4 + q] = r;
    }
}

template <int CH, int KIND>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters, long long *clk) {
    const int l = threadIdx.x & 63;
    f32x4 acc[CH];
    for (int c = 0; c < CH; c++) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (KIND == 0)
                    acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
                else if (KIND == 2)
                    acc[c] = (u & 1) ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 5, 0)
                                     : __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 10, 0);
                else
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < CH; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// 64 rows (one per lane) x 4 queries, d = 128: chain on the MFMA vs fmaf chain
__global__ void k_exact(const float *rows /*[64][128]*/, const float *qs /*[4][128]*/, float *out_mfma /*[64][4] lane,vgpr*/,
                        float *out_ref /*[64 rows][4 q]*/) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 128; k++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(rows[l * 128 + k], qs[(l & 3) * 128 + k], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) out_mfma[l * 4 + i] = acc[i];
    for (int q = 0; q < 4; q++) {
        float r = 0.f;
        for (int k = 0; k < 128; k++) r = __fmaf_rn(rows[l * 128 + k], qs[q * 128 + k], r);
        out_ref[l * 4 + q] = r;
    }
}

template <int CH, int KIND>
static void rate(const char *name, float *d_out, long long *d_clk) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k_rate<CH, KIND>), dim3(blocks), dim3(256), 0, 0, d_out, 10, d_clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<CH, KIND>), dim3(blocks), dim3(256), 0, 0, d_out, iters, d_clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = (double)iters * 16 * CH;  // per wave
    const double flop_per = KIND == 1 ? 2048.0 : 512.0;
    const double tflops = n_inst * flop_per * blocks * 4 / (ms * 1e-3) / 1e12;
    // one wave per SIMD: ns per instruction per wave -> cycles at 2.4 GHz
    printf("%-28s chains=%d  %.3f ms  %.1f TFLOP/s  %.2f cycles/inst @2.4GHz\n", name, CH, ms, tflops, ms * 1e-3 / n_inst * 2.4e9);
}

int main() {
    float *d_out;
    long long *d_clk;
    hipMalloc(&d_out, 256 * 256 * 4 * 4);
    hipMalloc(&d_clk, 256 * 8);
    // (1) layout
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d_out, 256 * 4, hipMemcpyDeviceToHost);
    printf("layout of v_mfma_f32_4x4x1_16b_f32: D[lane][vgpr] = A(lane x) * B(lane y)\n");
    bool as_expected = true;
    for (int l = 0; l < 64; l++) {
        for (int i = 0; i < 4; i++) {
            int fx = -1, fy = -1;
            for (int x = 0; x < 64 && fx < 0; x++)
                for (int y = 0; y < 64; y++)
                    if ((float)(100 + x) * (float)(1000 + y) == h[l * 4 + i]) { fx = x; fy = y; break; }
            if (l < 8 || l >= 60) printf("  lane %2d vgpr %d: A lane %2d  B lane %2d\n", l, i, fx, fy);
            // expectation: block b = l / 4, j = l % 4: D[i][j] = A(row i of block b = lane 4b+i) * B(col j of block b = lane 4b+j)
            if (fx != (l / 4) * 4 + i || fy != l) as_expected = false;
        }
    }
    printf("layout matches 'lane 4b+j, vgpr i = A(lane 4b+i) * B(lane 4b+j)': %s\n", as_expected ? "YES" : "NO");
    // (2) rate
    rate<1, 0>("4x4x1_16b", d_out, d_clk);
    rate<2, 0>("4x4x1_16b", d_out, d_clk);
    rate<4, 0>("4x4x1_16b", d_out, d_clk);
    rate<8, 0>("4x4x1_16b", d_out, d_clk);
    rate<1, 2>("4x4x1_16b cbsz=4", d_out, d_clk);
    rate<2, 2>("4x4x1_16b cbsz=4", d_out, d_clk);
    rate<4, 2>("4x4x1_16b cbsz=4", d_out, d_clk);
    rate<1, 1>("16x16x4", d_out, d_clk);
    rate<2, 1>("16x16x4", d_out, d_clk);
    rate<4, 1>("16x16x4", d_out, d_clk);
    // (3) exactness
    std::vector<float> rows(64 * 128), qs(4 * 128);
    srand(7);
    for (auto &v : rows) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto &v : qs) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *d_rows, *d_qs, *d_m, *d_r;
    hipMalloc(&d_rows, rows.size() * 4);
    hipMalloc(&d_qs, qs.size() * 4);
    hipMalloc(&d_m, 256 * 4);
    hipMalloc(&d_r, 256 * 4);
    hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_qs, qs.data(), qs.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, d_rows, d_qs, d_m, d_r);
    std::vector<float> hm(256), hr(256);
    hipMemcpy(hm.data(), d_m, 256 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), d_r, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            // D[lane l][vgpr i] = row (4b+i) . query (l%4)
            const int row = (l / 4) * 4 + i, q = l & 3;
            uint32_t a, b;
            memcpy(&a, &hm[l * 4 + i], 4);
            memcpy(&b, &hr[row * 4 + q], 4);
            bad += a != b;
        }
    printf("128-step 4x4x1 chain vs fmaf chain: %d of 256 results differ in bits\n", bad);
    // (4) A broadcast: CBSZ = 4, ABID = n -> D[lane l][vgpr i] = A(lane 4n+i) * B(lane l)
    {
        bool ok = true;
        for (int abid : {0, 5, 15}) {
            if (abid == 0) hipLaunchKernelGGL(k_layout_bc<0>, dim3(1), dim3(64), 0, 0, d_out);
            if (abid == 5) hipLaunchKernelGGL(k_layout_bc<5>, dim3(1), dim3(64), 0, 0, d_out);
            if (abid == 15) hipLaunchKernelGGL(k_layout_bc<15>, dim3(1), dim3(64), 0, 0, d_out);
            hipMemcpy(h.data(), d_out, 256 * 4, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 4; i++) {
                    const float want = (float)(100 + 4 * abid + i) * (float)(1000 + l);
                    if (h[l * 4 + i] != want) {
                        if (ok) printf("  cbsz=4 abid=%d lane %d vgpr %d: got %.0f want %.0f\n", abid, l, i, h[l * 4 + i], want);
                        ok = false;
                    }
                }
        }
        printf("CBSZ=4/ABID=n gives D[lane l][vgpr i] = A(lane 4n+i) * B(lane l): %s\n", ok ? "YES" : "NO");
        hipLaunchKernelGGL(k_exact_bc, dim3(1), dim3(64), 0, 0, d_rows, d_qs, d_m, d_r);
        hipMemcpy(hm.data(), d_m, 256 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), d_r, 256 * 4, hipMemcpyDeviceToHost);
        int bad2 = 0;
        for (int x = 0; x < 256; x++) bad2 += memcmp(&hm[x], &hr[x], 4) != 0;
        printf("128-step broadcast-A chain (queries in 8 registers, rows as B) vs fmaf chain: %d of 256 results differ in bits\n", bad2);
    }
    return 0;
}
