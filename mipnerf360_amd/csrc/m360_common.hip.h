// Shared device helpers for libm360 (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "../../include/m360.h"

namespace m360 {

constexpr int kWave = 64;
constexpr float kEpsG = 1e-6f;       // g(): intern/parameterization.py:18
constexpr int kIpeDirs = 21;         // intern/encoding.py:9-30
constexpr int kIpeCh = 2 * kIpeDirs; // 42

// host-side error plumbing (m360_capi.hip)
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
// deterministic column sums of a row-major [R, C] matrix (m360_linear.hip): scratch holds slices * C floats
int launch_colsum(const float *in, long R, int C, int ld, float *scratch, int slices, float *out, hipStream_t st);

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// ---- scalar math matching torch's fp32 semantics -------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// torch.nn.Softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
// torch.nan_to_num default: nan -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX
__device__ __forceinline__ float nan_to_numf_(float x) {
    if (isnan(x)) return 0.0f;
    if (isinf(x)) return x > 0 ? FLT_MAX : -FLT_MAX;
    return x;
}
// torch.maximum / torch.minimum: NaN if either operand is NaN (fmaxf / fminf return the other operand)
__device__ __forceinline__ float nan_maxf_(float a, float b) { return (a != a || b != b) ? a + b : fmaxf(a, b); }
__device__ __forceinline__ float nan_minf_(float a, float b) { return (a != a || b != b) ? a + b : fminf(a, b); }
// ReLU that lets NaN through, as torch.relu does (v_max_f32 returns the non-NaN operand: relu(NaN) would be 0 and a ray
// with NaN features would come out of the MLP finite).  Signed-integer max on the bit pattern: everything with the sign
// bit set becomes +0, everything else - +NaN included, the only NaN the matrix pipes and the fp32 adders produce - stays.
// One v_max_i32, the same cost as v_max_f32.  (Fixture G18.)
__device__ __forceinline__ float relu_nanf_(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// NaN -> the positive quiet NaN 0x7FC00000, everything else unchanged.  The fp32 MFMA and the fp32 adders hand a NaN
// operand on with its sign and payload (measured: tools/diag/nan_bits.py), so a feature row that enters the MLP with its
// NaNs canonicalised stays +NaN through every layer - which is what relu_nanf_ relies on.
__device__ __forceinline__ float canon_nanf_(float v) { return v != v ? __builtin_bit_cast(float, 0x7FC00000) : v; }
// bf16x3 mode: an fp32 value as two bf16 terms, hi = bf16(v) (round to nearest even), lo = bf16(v - hi): 16 significant bits.
// A product x w is then formed as xh wh + xl wh + xh wl on the bf16 MFMA with fp32 accumulation (the xl wl term, 2^-16 of
// the product, is dropped).
// A non-finite hi carries the whole value: lo = 0 (Inf - Inf would make the second term NaN and turn an Inf activation into NaN
// one layer earlier than the fp32 path does).
__device__ __forceinline__ __bf16 bf16_lo_(float v, __bf16 hi) {
    const float h = (float)hi;
    return (__bf16)(__builtin_isfinite(h) ? v - h : 0.0f);
}
__device__ __forceinline__ void split_bf16_(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;
    lo = bf16_lo_(v, hi);
}
// Three bf16 terms hi + mid + lo: all 24 significant bits of an fp32 value (v - hi and (v - hi) - mid are exact in fp32).  The
// first layers of the bf16 / bf16x3 modes multiply such triples ("x6" rows, m360_linear.hip): six products
// xl wh + xm wm + xh wl + xm wh + xh wm + xh wh on the bf16 MFMA with fp32 accumulation give the fp32 product up to 2^-24 terms.
__device__ __forceinline__ void split3_bf16_(float v, __bf16 &hi, __bf16 &mid, __bf16 &lo) {
    hi = (__bf16)v;
    const float h = (float)hi;
    const float r = __builtin_isfinite(h) ? v - h : 0.0f;
    mid = (__bf16)r;
    lo = (__bf16)(r - (float)mid);
}
// value returned by the reference's g() on its `calls`-th application to the same tensor
__device__ __forceinline__ float g_calls(float x, int calls) {
    for (int i = 0; i < calls; ++i) x = x + kEpsG;
    return 1.0f / x;
}
// torch.linspace(start, end, steps)[i] (symmetric two-sided evaluation, fp32 step)
__device__ __forceinline__ float linspacef_(float start, float end, int steps, int i) {
    if (steps <= 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// ---- counter-based random numbers (round 5): randomized=True without materialised uniform tensors ------------------------
// The reference draws `torch.rand(batch, num_samples + 1)` twice per forward (intern/ray.py:104 stratified jitter, :31 randomized
// inverse CDF).  The kernels draw the same KIND of numbers themselves: Philox4x32-10 (Salmon et al., SC'11 - the generator behind
// torch's own device RNG), key = the torch generator's seed, counter = (generator offset / 4 as 64 bits, element index, stream id), so
// a value is a pure function of (seed, offset, stream, element): the prologue can draw a ray's jitter once for t and again for the
// norm's partial sums, a test can dump exactly the uniforms a launch used (m360_philox_uniform) and replay them through the CPU restatement of the path,
// and the same seed gives the same bits.  The host advances the generator's offset by 4 per call, so no later torch kernel (whose
// counters start at its own offset / 4) ever reuses a counter.  Stream ids: 0 = t_rand (jitter), 1 = u_rand (inverse CDF).
struct rng_t {
    unsigned long long seed, offset;
    int on;  // draw in the kernel (the matching t_rand / u_rand pointer is NULL and m360_hyper_t.randomized asks for it)
};
__device__ __forceinline__ unsigned philox4x32_10_x(unsigned long long seed, unsigned long long offset, unsigned stream, unsigned long long idx) {
    unsigned c0 = (unsigned)offset, c1 = (unsigned)(offset >> 32), c2 = (unsigned)idx, c3 = (stream << 28) | (unsigned)((idx >> 32) & 0x0FFFFFFFu);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}
// uniform in [0, 1) with 24 bits, like torch.rand's float32
__device__ __forceinline__ float philox_uniform(const rng_t &g, unsigned stream, unsigned long long idx) {
    return (float)(philox4x32_10_x(g.seed, g.offset, stream, idx) >> 8) * 5.9604644775390625e-08f;
}

// ---- wave-level reductions / scans (64 lanes) ---------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
// inclusive wave scan in fp64 (torch's CPU cumsum accumulates fp32 inputs in double and rounds every prefix once, which also
// keeps the CDF monotone), on the DPP network - no LDS traffic, no ds_bpermute chains: four row_shr steps scan each row of
// 16 lanes, row_bcast:15 adds lane 15 of rows 0 / 2 to rows 1 / 3, row_bcast:31 adds lane 31 to rows 2 and 3.  A double
// travels as its two 32-bit halves under the same DPP control; lanes without a source receive 0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_d_(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_incl_scan_d(double v) {
    v += dpp_shift_d_<0x111, 0xF>(v);  // row_shr:1
    v += dpp_shift_d_<0x112, 0xF>(v);  // row_shr:2
    v += dpp_shift_d_<0x114, 0xF>(v);  // row_shr:4
    v += dpp_shift_d_<0x118, 0xF>(v);  // row_shr:8
    v += dpp_shift_d_<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_shift_d_<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// xor-butterfly over 8 consecutive lanes (DPP: no LDS traffic); every lane of the group gets the sum.  Shared by the fused
// epilogue of the last hidden layer (m360_linear_persist.hip.h) and the finishers' tail-row path (m360_ray.hip), which
// must add in the same order.
__device__ __forceinline__ float row8_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror: lane i <-> 7 - i
    return x;
}

// ---- frustum -> gaussian ------------------------------------------------------------------
// intern/parameterization.py:99-107
__device__ __forceinline__ void frustum_moments(float t0, float t1, float radius, float &t_mean,
                                                float &t_var, float &r_var) {
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw;
    const float hw4 = hw2 * hw2;
    const float den = 3.0f * mu2 + hw2;
    t_mean = mu + (2.0f * mu * hw2) / den;
    t_var = hw2 / 3.0f - (4.0f / 15.0f) * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
    r_var = (radius * radius) * (mu2 / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / den);
}

// intern/parameterization.py:44-46,55-62: mean = d t_mean, cov = t_var d d^T + r_var (I - d (d/|d|^2)^T)
__device__ __forceinline__ void lift_to_xyz(const float d[3], float t_mean, float t_var, float r_var,
                                            float mean[3], float cov[9]) {
    const float mag = fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        mean[i] = d[i] * t_mean;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float null_ij = (i == j ? 1.0f : 0.0f) - d[i] * (d[j] / mag);
            cov[3 * i + j] = t_var * (d[i] * d[j]) + r_var * null_ij;
        }
    }
}

// global contraction scale (intern/parameterization.py:23-29 applied to the whole tensor)
__device__ __forceinline__ void contract_mean(float mean[3], float gnorm) {
    if (gnorm <= 1.0f) return;
    const float s = 2.0f - 1.0f / gnorm;
#pragma unroll
    for (int i = 0; i < 3; ++i) mean[i] = s * (mean[i] / gnorm);
}

// cov <- J cov J^T with J = d contract / dy at the (already contracted) mean y; closed form of
// what intern/parameterization.py:76-81 obtains from autograd per sample.
__device__ __forceinline__ void contract_cov(const float y[3], float cov[9]) {
    const float r = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    if (r <= 1.0f) return;  // J = I.  A NaN norm takes the other branch, as `if x_norm <= 1` does in contract(): NaN Jacobian
    const float r2 = r * r;
    const float a = 2.0f / r - 1.0f / r2;
    const float b = 2.0f / (r2 * r2) - 2.0f / (r2 * r);
    float J[9], M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) J[3 * i + j] = (i == j ? a : 0.0f) + b * (y[i] * y[j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            M[3 * i + j] = J[3 * i] * cov[j] + J[3 * i + 1] * cov[3 + j] + J[3 * i + 2] * cov[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            cov[3 * i + j] = M[3 * i] * J[3 * j] + M[3 * i + 1] * J[3 * j + 1] + M[3 * i + 2] * J[3 * j + 2];
}

// 21 unit directions, intern/encoding.py:9-30
__device__ static const float kIpeBasis[kIpeDirs][3] = {
    {0.8506508f, 0.f, 0.5257311f},   {0.809017f, 0.5f, 0.309017f},    {0.5257311f, 0.8506508f, 0.f},
    {1.f, 0.f, 0.f},                 {0.809017f, 0.5f, -0.309017f},   {0.8506508f, 0.f, -0.5257311f},
    {0.309017f, 0.809017f, -0.5f},   {0.f, 0.5257311f, -0.8506508f},  {0.5f, 0.309017f, -0.809017f},
    {0.f, 1.f, 0.f},                 {-0.5257311f, 0.8506508f, 0.f},  {-0.309017f, 0.809017f, -0.5f},
    {0.f, 0.5257311f, 0.8506508f},   {-0.309017f, 0.809017f, 0.5f},   {0.309017f, 0.809017f, 0.5f},
    {0.5f, 0.309017f, 0.809017f},    {0.5f, -0.309017f, 0.809017f},   {0.f, 0.f, 1.f},
    {-0.5f, 0.309017f, 0.809017f},   {-0.809017f, 0.5f, 0.309017f},   {-0.809017f, 0.5f, -0.309017f}};

// sin and cos of one argument with |x| <= 8192, <= 1e-7 absolute error: three-constant Cody-Waite reduction by pi/2
// with fused multiply-adds + the minimax polynomials of cephes sinf / cosf.  (The library sincosf costs ~190 instructions
// and a branch per call: at 21 directions per sample it made encode_features ALU-bound at 2.2 TB/s, profiles/r02.)
__device__ __forceinline__ void sincos_small(float x, float *s, float *c) {
    const float kf = rintf(x * 0.63661977236758134f);
    float r = fmaf(-kf, 1.5703125f, x);
    r = fmaf(-kf, 4.837512969970703125e-4f, r);
    r = fmaf(-kf, 7.549789948768648e-8f, r);
    const float z = r * r;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float sn = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float cs = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = (int)kf & 3;
    const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}

// intern/encoding.py:43-56 for one sample: out[k] = exp(-sigma_k/2) sin(gamma_k), out[21+k] = ... cos
// STATIC_K: the caller's store() needs compile-time channel indices (values kept in registers): the rare path is then
// unrolled as well (dead weight in the code object, never fetched by the contracted path)
template <bool HAS_COV, bool STATIC_K = false, typename Store>
__device__ __forceinline__ void ipe_sample(const float mean[3], const float cov[9], Store &&store) {
    // |gamma_k| <= |mean| (unit directions): ONE range test per sample picks the short sin / cos for all 21 directions;
    // far-away / non-finite means (never produced by the contracted path) take the library routines in a rolled loop
    const float m2 = mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2];
    if (!(m2 <= 6.0e7f)) {
        auto slow = [&](int k) __attribute__((always_inline)) {
            const float p0 = kIpeBasis[k][0], p1 = kIpeBasis[k][1], p2 = kIpeBasis[k][2];
            const float gamma = p0 * mean[0] + p1 * mean[1] + p2 * mean[2];
            float sn, cs;
            sincosf(gamma, &sn, &cs);
            if (HAS_COV) {
                const float a0 = cov[0] * p0 + cov[1] * p1 + cov[2] * p2;
                const float a1 = cov[3] * p0 + cov[4] * p1 + cov[5] * p2;
                const float a2 = cov[6] * p0 + cov[7] * p1 + cov[8] * p2;
                const float damp = expf(-0.5f * (p0 * a0 + p1 * a1 + p2 * a2));
                sn *= damp;
                cs *= damp;
            }
            store(k, sn);
            store(kIpeDirs + k, cs);
        };
        if (STATIC_K) {
#pragma unroll
            for (int k = 0; k < kIpeDirs; ++k) slow(k);
        } else {
#pragma unroll 1
            for (int k = 0; k < kIpeDirs; ++k) slow(k);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kIpeDirs; ++k) {
        const float p0 = kIpeBasis[k][0], p1 = kIpeBasis[k][1], p2 = kIpeBasis[k][2];
        const float gamma = p0 * mean[0] + p1 * mean[1] + p2 * mean[2];
        float sn, cs;
        sincos_small(gamma, &sn, &cs);
        if (HAS_COV) {
            const float a0 = cov[0] * p0 + cov[1] * p1 + cov[2] * p2;
            const float a1 = cov[3] * p0 + cov[4] * p1 + cov[5] * p2;
            const float a2 = cov[6] * p0 + cov[7] * p1 + cov[8] * p2;
            const float sigma = p0 * a0 + p1 * a1 + p2 * a2;
            // hardware exp2 (v_exp_f32, ~1 ulp): relative error of damp <= (1 + |sigma| / 2) * 1.2e-7, against the 2e-6
            // absolute tolerance of the encoding and |damp * sin| <= exp(-sigma / 2)
            const float damp = __expf(-0.5f * sigma);
            sn *= damp;
            cs *= damp;
        }
        store(k, sn);
        store(kIpeDirs + k, cs);
    }
}

}  // namespace m360
