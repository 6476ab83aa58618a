// What rounding does hipcc's float -> __bf16 conversion (v_cvt_pk_bf16_f32 on gfx950) apply?  The prefilter's error bound
// (qk_scan_rl.hip, qk_dense_pf.hip) assumes round-to-nearest: relative error <= 2^-9 per operand.  Converts 2^24 floats on the device
// (random bit patterns over the whole exponent range, ties included) and compares with a host round-to-nearest-even reference.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/cvt_bf16 scripts/micro/cvt_bf16.hip && scripts/micro/cvt_bf16
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__global__ void k_cvt(const float *x, uint16_t *y, int n) {
    int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 < n) {
        bf16x2 v = {(__bf16)x[i], (__bf16)x[i + 1]};
        uint32_t b = __builtin_bit_cast(uint32_t, v);
        y[i] = (uint16_t)(b & 0xFFFFu);
        y[i + 1] = (uint16_t)(b >> 16);
    }
}

static uint16_t rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16) | ((u & 0xFFFFu) ? 0x40 : 0);  // inf / NaN (quiet)
    const uint32_t lsb = (u >> 16) & 1u;
    u += 0x7FFFu + lsb;
    return (uint16_t)(u >> 16);
}

int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t u = (uint32_t)(s >> 16);
        if ((i & 15) == 0) u = (u & 0xFFFF0000u) | 0x8000u;  // exact ties
        if ((u & 0x7F800000u) == 0x7F800000u) u &= 0xBFFFFFFFu;  // no inf / NaN
        memcpy(&h[i], &u, 4);
    }
    float *dx; uint16_t *dy;
    hipMalloc((void **)&dx, n * 4); hipMalloc((void **)&dy, n * 2);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dy, n);
    std::vector<uint16_t> r(n);
    hipMemcpy(r.data(), dy, n * 2, hipMemcpyDeviceToHost);
    long diff = 0, worse = 0, denorm_diff = 0;
    double max_rel = 0;
    for (int i = 0; i < n; i++) {
        const uint16_t want = rne(h[i]);
        uint32_t ub = (uint32_t)r[i] << 16;
        float back; memcpy(&back, &ub, 4);
        uint32_t hb; memcpy(&hb, &h[i], 4);
        const bool denorm = (hb & 0x7F800000u) == 0;
        if (r[i] != want) { diff++; if (denorm) denorm_diff++; }
        if (!denorm && h[i] != 0.0f) {
            const double rel = fabs((double)back - (double)h[i]) / fabs((double)h[i]);
            if (rel > max_rel) max_rel = rel;
            if (rel > 0.001953125 * 1.0000001) worse++;  // 2^-9
        }
    }
    printf("{\"values\": %d, \"differ_from_round_to_nearest_even\": %ld, \"of_which_denormal_inputs\": %ld, \"max_relative_error_normal_inputs\": %.9g, "
           "\"two_pow_minus_9\": %.9g, \"normal_inputs_beyond_2^-9\": %ld}\n", n, diff, denorm_diff, max_rel, 0.001953125, worse);
    return (worse == 0) ? 0 : 1;
}
