// Per-ray kernels: transmittance scan (density -> weights), max-blur + piecewise-constant
// inverse-CDF resampling, alpha compositing, and the two fused "stage finishers" that also
// apply the last (hidden -> 1 / hidden -> 4) linear heads.
//
// Mapping: one wavefront (64 lanes) owns one ray; per-ray arrays (t, weights, cdf, ...) live
// in that wave's private LDS slice; prefix sums are wave-level Hillis-Steele scans in fp64
// (torch's CPU cumsum accumulates fp32 in double; this also keeps the CDF monotone) with a
// carry across 64-sample chunks, so any N works.  All of these kernels are HBM-bound.
#include "m360_common.hip.h"

namespace m360 {

// intra-wave LDS ordering: slices are wave-private, so a wave-scope fence suffices
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float dir_norm(const float *__restrict__ dirs, int b) {
    const float x = dirs[3 * b], y = dirs[3 * b + 1], z = dirs[3 * b + 2];
    return sqrtf(x * x + y * y + z * z);
}

// model.py:71-77 / intern/ray.py:173-183: w_i = (1 - exp(-x_i)) exp(-sum_{j<i} x_j), x = rho * (dt * |d|)
__device__ __forceinline__ void wave_weights(const float *t, const float *rho, int rho_stride,
                                             float dnorm, int N, float *w) {
    const int l = lane_id();
    double carry = 0.0;
    for (int base = 0; base < N; base += kWave) {
        const int i = base + l;
        float x = 0.0f;
        if (i < N) x = rho[i * rho_stride] * ((t[i + 1] - t[i]) * dnorm);
        const double incl = wave_incl_scan_d((double)x);
        // exclusive prefix = the previous lane's inclusive one (not incl - x: an infinite or NaN x_i must not reach its
        // own transmittance; the reference's cumsum runs over x[:-1])
        double prev = __shfl_up(incl, 1, kWave);
        if (l == 0) prev = 0.0;
        const float excl = (float)(carry + prev);
        if (i < N) w[i] = (1.0f - expf(-x)) * expf(-excl);
        carry += __shfl(incl, kWave - 1, kWave);
    }
}

// intern/ray.py:137-142: replicate-pad, pairwise max, pairwise mean, + padding
__device__ __forceinline__ void wave_blur(const float *w, int N, float padding, float *out) {
    for (int i = lane_id(); i < N; i += kWave) {
        const float c = w[i];
        const float lo = nan_maxf_(w[i > 0 ? i - 1 : 0], c);  // torch.maximum: NaN-propagating
        const float hi = nan_maxf_(c, w[i < N - 1 ? i + 1 : N - 1]);
        out[i] = 0.5f * (lo + hi) + padding;
    }
}

// intern/ray.py:12-57.  bins[nb] and w[nb-1] in LDS (w is overwritten), cdf[nb] LDS scratch.
// u_rand_row: the ray's uniforms (randomized=True with a materialised tensor), or NULL and rng.on: drawn here from the Philox stream
// (stream id 1, element rng_base + j), or NULL and !rng.on: the deterministic linspace
__device__ __forceinline__ void wave_sorted_pdf(const float *bins, float *w, float *cdf, int nb, int ns,
                                                const float *__restrict__ u_rand_row,
                                                float *__restrict__ out_row, const rng_t &rng = rng_t{0, 0, 0}, long rng_base = 0) {
    const int l = lane_id();
    const int nw = nb - 1;
    float part = 0.0f;
    for (int i = l; i < nw; i += kWave) part += w[i];
    float wsum = wave_sum(part);
    const float pad = nan_maxf_(0.0f, 1e-5f - wsum);
    const float padw = pad / (float)nw;
    wsum = wsum + pad;
    // cdf[0] = 0, cdf[i+1] = min(1, cumsum(pdf)[i]) for i < nw-1, cdf[nw] = 1
    double carry = 0.0;
    for (int base = 0; base < nw - 1; base += kWave) {
        const int i = base + l;
        float pdf = 0.0f;
        if (i < nw - 1) pdf = (w[i] + padw) / wsum;
        const double incl = wave_incl_scan_d((double)pdf);
        if (i < nw - 1) cdf[i + 1] = nan_minf_(1.0f, (float)(carry + incl));
        carry += __shfl(incl, kWave - 1, kWave);
    }
    if (l == 0) {
        cdf[0] = 0.0f;
        cdf[nw] = 1.0f;
    }
    wave_sync();
    // A cdf that is not a finite non-decreasing sequence (NaN / Inf / negative weights: never from the path's own
    // weights, fixture G18) has no "sorted" bracket; the reference's comparison table (intern/ray.py:43-50) still defines
    // one, evaluated literally below.
    int bad = 0;
    for (int i = l; i < nw; i += kWave) {
        const float a = cdf[i], c = cdf[i + 1];
        bad |= !(c >= a) || !(fabsf(c) <= FLT_MAX);
    }
    bad = __any(bad);
    const float f32eps = 1.1920928955078125e-07f;
    const float umax = 1.0f - f32eps;
    for (int j = l; j < ns; j += kWave) {
        float u;
        if (u_rand_row == nullptr && !rng.on) {
            u = linspacef_(0.0f, umax, ns, j);
        } else {  // intern/ray.py:30-35, including the `u + u` doubling
            const float s = 1.0f / (float)ns;
            const float base = (float)j * s;
            const float ur = u_rand_row != nullptr ? u_rand_row[j] : philox_uniform(rng, 1u, (unsigned long long)(rng_base + j));
            u = fminf(base + base + ur * (s - f32eps), umax);
        }
        float c0, c1, b0, b1;
        if (!bad) {
            // last index i with cdf[i] <= u  (cdf[0] = 0 <= u always)
            int lo = 0, hi = nb;  // invariant: cdf[lo] <= u, (hi == nb or cdf[hi] > u)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= u) lo = mid; else hi = mid;
            }
            const int i0 = lo, i1 = lo + 1 < nb ? lo + 1 : nb - 1;
            c0 = cdf[i0], c1 = cdf[i1];
            b0 = bins[i0], b1 = bins[i1];
        } else {
            // x0 = max_i (u >= cdf_i ? x_i : x_0), x1 = min_i (u >= cdf_i ? x_last : x_i), NaN-propagating like torch.max / min
            c0 = cdf[0], b0 = bins[0], c1 = cdf[nb - 1], b1 = bins[nb - 1];
            for (int i = 0; i < nb; ++i) {
                const bool m = u >= cdf[i];
                c0 = nan_maxf_(c0, m ? cdf[i] : cdf[0]);
                b0 = nan_maxf_(b0, m ? bins[i] : bins[0]);
                c1 = nan_minf_(c1, m ? cdf[nb - 1] : cdf[i]);
                b1 = nan_minf_(b1, m ? bins[nb - 1] : bins[i]);
            }
        }
        float tt = nan_to_numf_((u - c0) / (c1 - c0));
        tt = fminf(fmaxf(tt, 0.0f), 1.0f);
        out_row[j] = b0 + tt * (b1 - b0);
    }
}

struct Composite { float r, g, b, dist, acc; };

// intern/ray.py:171-191 given per-sample rgb (LDS, stride 3 floats starting at rgb[0]) and weights w (LDS)
__device__ __forceinline__ Composite wave_composite(const float *t, const float *w, const float *rgb,
                                                    int rgb_stride, int N, bool white_bkgd) {
    float sr = 0, sg = 0, sb = 0, sa = 0, sd = 0;
    for (int i = lane_id(); i < N; i += kWave) {
        const float wi = w[i];
        sr += wi * rgb[i * rgb_stride + 0];
        sg += wi * rgb[i * rgb_stride + 1];
        sb += wi * rgb[i * rgb_stride + 2];
        sa += wi;
        sd += wi * (0.5f * (t[i] + t[i + 1]));
    }
    Composite c;
    c.r = wave_sum(sr);
    c.g = wave_sum(sg);
    c.b = wave_sum(sb);
    c.acc = wave_sum(sa);
    const float d = nan_to_numf_(wave_sum(sd) / c.acc);
    c.dist = fminf(fmaxf(d, t[0]), t[N]);
    if (white_bkgd) {
        const float bg = 1.0f - c.acc;
        c.r += bg;
        c.g += bg;
        c.b += bg;
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// standalone per-ray kernels: 4 waves / block, one ray per wave
constexpr int kRayWaves = 4;

__global__ __launch_bounds__(kRayWaves *kWave) void density_to_weight_kernel(
    const float *__restrict__ t_vals, const float *__restrict__ density,
    const float *__restrict__ dirs, int B, int N, float *__restrict__ weights) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int b = blockIdx.x * kRayWaves + wave;
    if (b >= B) return;
    float *t = smem + wave * (3 * N + 1), *rho = t + N + 1, *w = rho + N;
    for (int i = l; i <= N; i += kWave) t[i] = t_vals[(long)b * (N + 1) + i];
    for (int i = l; i < N; i += kWave) rho[i] = density[(long)b * N + i];
    wave_sync();
    wave_weights(t, rho, 1, dir_norm(dirs, b), N, w);
    wave_sync();
    for (int i = l; i < N; i += kWave) weights[(long)b * N + i] = w[i];
}

template <bool BLUR>
__global__ __launch_bounds__(kRayWaves *kWave) void resample_kernel(
    const float *__restrict__ bins_g, const float *__restrict__ weights_g,
    const float *__restrict__ u_rand, int B, int nb, int ns, float padding,
    float *__restrict__ samples, rng_t rng = rng_t{0, 0, 0}) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int b = blockIdx.x * kRayWaves + wave;
    if (b >= B) return;
    const int nw = nb - 1;
    float *bins = smem + wave * (4 * nb), *w = bins + nb, *w2 = w + nb, *cdf = w2 + nb;
    for (int i = l; i < nb; i += kWave) bins[i] = bins_g[(long)b * nb + i];
    for (int i = l; i < nw; i += kWave) w[i] = weights_g[(long)b * nw + i];
    wave_sync();
    float *wsrc = w;
    if (BLUR) {
        wave_blur(w, nw, padding, w2);
        wave_sync();
        wsrc = w2;
    }
    wave_sorted_pdf(bins, wsrc, cdf, nb, ns, u_rand ? u_rand + (long)b * ns : nullptr,
                    samples + (long)b * ns, rng, (long)b * ns);
}

__global__ __launch_bounds__(kRayWaves *kWave) void volumetric_rendering_kernel(
    const float *__restrict__ rgb_g, const float *__restrict__ density,
    const float *__restrict__ t_vals, const float *__restrict__ dirs, int B, int N, int white_bkgd,
    float *__restrict__ comp_rgb, float *__restrict__ distance, float *__restrict__ acc,
    float *__restrict__ weights) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int b = blockIdx.x * kRayWaves + wave;
    if (b >= B) return;
    float *t = smem + wave * (6 * N + 1), *rho = t + N + 1, *w = rho + N, *rgb = w + N;
    for (int i = l; i <= N; i += kWave) t[i] = t_vals[(long)b * (N + 1) + i];
    for (int i = l; i < N; i += kWave) rho[i] = density[(long)b * N + i];
    for (int i = l; i < 3 * N; i += kWave) rgb[i] = rgb_g[(long)b * 3 * N + i];
    wave_sync();
    wave_weights(t, rho, 1, dir_norm(dirs, b), N, w);
    wave_sync();
    const Composite c = wave_composite(t, w, rgb, 3, N, white_bkgd != 0);
    if (l == 0) {
        comp_rgb[3 * b] = c.r;
        comp_rgb[3 * b + 1] = c.g;
        comp_rgb[3 * b + 2] = c.b;
        distance[b] = c.dist;
        acc[b] = c.acc;
    }
    if (weights != nullptr)
        for (int i = l; i < N; i += kWave) weights[(long)b * N + i] = w[i];
}

// intern/utils.py:17-20
__global__ void to8b_kernel(const float *__restrict__ x, long n, uint8_t *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float v = fminf(fmaxf(nan_to_numf_(x[idx]), 0.0f), 1.0f);
    out[idx] = (uint8_t)(255.0f * v);
}

// ------------------------------------------------------------------------------------------
// Fused stage finishers: one 256-thread workgroup per ray.  Phase 1: the 4 waves stream the
// ray's N activation rows (k_pad floats each, 16 B per lane, fully coalesced) and reduce the
// H head dot-products per row; head weights sit in LDS.  Phase 2: wave 0 runs the scans.
constexpr int kFinishThreads = 256;

// one 16-byte chunk of an activation row as floats: 4 fp32 or 8 bf16 values
template <typename T> struct ActChunk;
template <> struct ActChunk<float> {
    static constexpr int kElems = 4;
    static __device__ __forceinline__ void load(const float *row, int c, float (&v)[4]) {
        const float4 x = reinterpret_cast<const float4 *>(row)[c];
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    }
};
typedef __bf16 bf16x8_r __attribute__((ext_vector_type(8)));
template <> struct ActChunk<__bf16> {
    static constexpr int kElems = 8;
    static __device__ __forceinline__ void load(const __bf16 *row, int c, float (&v)[8]) {
        const bf16x8_r x = reinterpret_cast<const bf16x8_r *>(row)[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
    }
};

template <int H, typename T>
__device__ __forceinline__ void head_dots(const T *__restrict__ act_ray, int ld,
                                          const float *hw /*LDS [H][k_pad]*/,
                                          const float *__restrict__ hb, int k_pad, int N,
                                          float *raw /*LDS [N][H]*/) {
    constexpr int E = ActChunk<T>::kElems;
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int nwaves = blockDim.x >> 6;
    const int kc = k_pad / E;
    float bias[H];
#pragma unroll
    for (int h = 0; h < H; ++h) bias[h] = hb[h];
    for (int n = wave; n < N; n += 2 * nwaves) {
        const int n2 = n + nwaves;
        const bool has2 = n2 < N;
        const T *x1 = act_ray + (long)n * ld;
        const T *x2 = act_ray + (long)(has2 ? n2 : n) * ld;
        float a1[H], a2[H];
#pragma unroll
        for (int h = 0; h < H; ++h) a1[h] = a2[h] = 0.0f;
        for (int c = l; c < kc; c += kWave) {
            float v1[E], v2[E];
            ActChunk<T>::load(x1, c, v1);
            ActChunk<T>::load(x2, c, v2);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float *wp = hw + h * k_pad + c * E;
#pragma unroll
                for (int e = 0; e < E; e += 4) {
                    const float4 w = *reinterpret_cast<const float4 *>(wp + e);
                    a1[h] += v1[e] * w.x + v1[e + 1] * w.y + v1[e + 2] * w.z + v1[e + 3] * w.w;
                    a2[h] += v2[e] * w.x + v2[e + 1] * w.y + v2[e + 2] * w.z + v2[e + 3] * w.w;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            a1[h] = wave_sum(a1[h]);
            a2[h] = wave_sum(a2[h]);
        }
        if (l == 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                raw[n * H + h] = a1[h] + bias[h];
                if (has2) raw[n2 * H + h] = a2[h] + bias[h];
            }
        }
    }
}

// bf16 activation rows: 8 lanes per row, 8 rows per wave.  A lane walks every 8th 16-byte chunk of its row (consecutive
// lanes read consecutive chunks: whole 128-byte lines per row), the head rows come from LDS (the 8 rows of a wave read the
// same addresses: broadcast), and the 8 lanes of a row are reduced by three DPP steps - no ds_bpermute chains.  The
// 64-lanes-per-row form above spends most of its time in its six-step shuffle reductions once a row is only 2 KB
// (profiles/r02: 3.3 TB/s on the bf16 path).
// SPLIT (bf16x3 mode): a row is [hi(k_pad) | lo(k_pad)] and the value is hi + lo (two bf16 terms = 16 significant bits).
template <int H, bool SPLIT = false>
__device__ __forceinline__ void head_dots_rows8_bf16(const __bf16 *__restrict__ act_ray, int ld, const float *hw /*LDS [H][k_pad]*/,
                                                     const float *__restrict__ hb, int k_pad, int N, float *raw /*LDS [N][H]*/) {
    const int wave = threadIdx.x >> 6, l = lane_id(), nwaves = blockDim.x >> 6;
    const int lane8 = l & 7, rsub = l >> 3;
    const int kc = k_pad / 8;
    float bias[H];
#pragma unroll
    for (int h = 0; h < H; ++h) bias[h] = hb[h];
    for (int n0 = wave * 8; n0 < N; n0 += nwaves * 8) {
        const int n = n0 + rsub;
        const bool valid = n < N;
        const __bf16 *x = act_ray + (long)(valid ? n : N - 1) * ld;
        float a[H];
#pragma unroll
        for (int h = 0; h < H; ++h) a[h] = 0.0f;
        for (int c = lane8; c < kc; c += 8) {
            float v[8];
            ActChunk<__bf16>::load(x, c, v);
            if (SPLIT) {
                float lo[8];
                ActChunk<__bf16>::load(x + k_pad, c, lo);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += lo[e];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float4 w0 = *reinterpret_cast<const float4 *>(hw + h * k_pad + c * 8);
                const float4 w1 = *reinterpret_cast<const float4 *>(hw + h * k_pad + c * 8 + 4);
                a[h] += v[0] * w0.x + v[1] * w0.y + v[2] * w0.z + v[3] * w0.w + v[4] * w1.x + v[5] * w1.y + v[6] * w1.z + v[7] * w1.w;
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            a[h] = row8_sum(a[h]);
            if (lane8 == 0 && valid) raw[n * H + h] = a[h] + bias[h];
        }
    }
}

// Head products of activation rows in EXACTLY the order of the fused epilogue (m360_linear_persist.hip.h, HEADS > 0): per
// 128-column wave tile ("slot") the 8 lanes of a row each chain 4 column blocks x 4 columns of fused multiply-adds, the
// lanes are reduced by the same DPP butterfly, the slots are added 0, 1, ...  Used for the rows the fused layer did not
// cover (ragged tail of the batch), so a row's result does not depend on which side of `fused_rows` it fell - a chunk
// rendered alone and the same chunk inside a larger launch stay bit-identical.  One wave per sample, 8 lanes per slot.
template <int H>
__device__ __forceinline__ void head_dots_fused_order(const float *__restrict__ act_rows, int ld, const float *hw /*LDS [H][k_pad]*/,
                                                      const float *__restrict__ hb, int k_pad, int slots, int count,
                                                      float *raw /*LDS [count][H]*/) {
    const int wave = threadIdx.x >> 6, l = lane_id(), nwaves = blockDim.x >> 6;
    const int q = l >> 3, lane8 = l & 7;
    for (int n = wave; n < count; n += nwaves) {
        float accq[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) accq[hh] = 0.0f;
        if (q < slots) {
            const float *x = act_rows + (long)n * ld + q * 128 + 4 * lane8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(x + 32 * j);
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(hw + hh * k_pad + q * 128 + 32 * j + 4 * lane8);
                    accq[hh] = fmaf(v.w, w4.w, fmaf(v.z, w4.z, fmaf(v.y, w4.y, fmaf(v.x, w4.x, accq[hh]))));
                }
            }
        }
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            accq[hh] = row8_sum(accq[hh]);
            float a = 0.0f;
            for (int s = 0; s < slots; ++s) a += __shfl(accq[hh], 8 * s, kWave);
            if (l == 0) raw[n * H + hh] = a + hb[hh];
        }
    }
}

// Head products of one ray's N samples -> raw[N][H] (LDS).  Samples whose global row (b N + n) lies below `fused_rows`
// take them from the partial sums the fused last layer left (m360_linear_heads: head_part[row][slots][H], added slot
// 0, 1, ... then the bias); the others are computed from the activation rows as in the unfused path.
template <int H, typename T, bool SPLIT = false>
__device__ __forceinline__ void ray_heads(const T *__restrict__ act, int ld, const float *__restrict__ head_part,
                                          long fused_rows, int slots, const float *__restrict__ head_w,
                                          const float *__restrict__ head_b, int k_pad, int b, int N, float *hw /*LDS [H][k_pad]*/,
                                          float *raw /*LDS [N][H]*/) {
    const long s0 = (long)b * N;
    long nfl = fused_rows - s0;
    const int nf = nfl <= 0 ? 0 : (nfl >= N ? N : (int)nfl);  // block-uniform
    if (nf < N)
        for (int i = threadIdx.x; i < H * k_pad; i += blockDim.x) hw[i] = head_w[i];
    if (sizeof(T) == 2 && H == 4 && slots % 8 == 0) {
        // bf16 / bf16x3 layers leave 8 slots per 256 columns (32 at width 1024: 512 contiguous bytes per sample): 8 lanes per
        // sample, each adds slots / 8 consecutive float4, then a DPP butterfly over the 8 lanes - whole lines per wave read
        // instead of 16-byte pieces 512 bytes apart
        const int per = slots / 8, lane8 = threadIdx.x & 7;
        for (int n = threadIdx.x >> 3; n < nf; n += blockDim.x >> 3) {
            const float4 *p = reinterpret_cast<const float4 *>(head_part + ((s0 + n) * slots + lane8 * per) * 4);
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int q = 0; q < per; ++q) {
                const float4 v = p[q];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            a.x = row8_sum(a.x); a.y = row8_sum(a.y); a.z = row8_sum(a.z); a.w = row8_sum(a.w);
            if (lane8 == 0) {
                raw[n * 4 + 0] = a.x + head_b[0];
                raw[n * 4 + 1] = a.y + head_b[1];
                raw[n * 4 + 2] = a.z + head_b[2];
                raw[n * 4 + 3] = a.w + head_b[3];
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < nf * H; idx += blockDim.x) {
            const int n = idx / H, hh = idx % H;
            const float *p = head_part + ((s0 + n) * slots) * H + hh;
            float a = 0.0f;
            for (int q = 0; q < slots; ++q) a += p[q * H];
            raw[idx] = a + head_b[hh];
        }
    }
    __syncthreads();
    if (nf < N) {
        if constexpr (sizeof(T) == 4) {  // fp32: widths the fused layer takes (slots > 0) keep ITS summation order in the tail rows
            if (slots > 0 && slots <= 8 && k_pad == 128 * slots) head_dots_fused_order<H>(reinterpret_cast<const float *>(act) + (s0 + nf) * ld, ld, hw, head_b, k_pad, slots, N - nf, raw + nf * H);
            else head_dots<H, T>(act + (s0 + nf) * ld, ld, hw, head_b, k_pad, N - nf, raw + nf * H);
        } else {
            head_dots_rows8_bf16<H, SPLIT>(reinterpret_cast<const __bf16 *>(act) + (s0 + nf) * ld, ld, hw, head_b, k_pad, N - nf, raw + nf * H);
        }
    }
    __syncthreads();
}

// The same sums for a ray whose N samples ALL lie below `fused_rows`, formed by ONE wave (the wave-per-ray path of the
// finishers below): every element is added by one lane in the order of ray_heads - slots 0, 1, ... then the bias; on the
// 8-slots-per-256-columns layout 8 lanes per sample and the same DPP butterfly - so both paths give the same bits.
template <int H, typename T>
__device__ __forceinline__ void ray_heads_fused_wave(const float *__restrict__ head_part, int slots, const float *__restrict__ head_b,
                                                     long s0, int N, float *raw /*LDS [N][H], this wave's*/) {
    const int l = lane_id();
    if (sizeof(T) == 4 && H == 4 && slots == 8) {
        // fp32 NeRF stage at width 1024: a sample's 8 slots x 4 heads are 128 contiguous bytes = one line.  Round 5: ONE LANE PER SAMPLE adds
        // its line's eight float4 in order - ((((0 + x0) + x1) + x2) ... + x7) + bias, the order of ray_heads - 36 adds per 64 samples where
        // rounds 3-4 spread a sample over 8 lanes (whole lines per load instruction) and paid 7 DPP adds x 4 heads per 8 samples: the kernel
        // is bound by its vector instructions, not by how its 64 KB per ray arrive; all of a lane's loads are in flight at once.
#pragma unroll 1
        for (int n = l; n < N; n += kWave) {
            const float4 *p = reinterpret_cast<const float4 *>(head_part + (s0 + n) * 32);
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = p[q];
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            raw[n * 4 + 0] = a.x + head_b[0];  // (raw is 4-byte aligned only: t[N + 1] lies before it)
            raw[n * 4 + 1] = a.y + head_b[1];
            raw[n * 4 + 2] = a.z + head_b[2];
            raw[n * 4 + 3] = a.w + head_b[3];
        }
    } else if (H == 1 && slots == 2) {
        // proposal stage at width 256 (fp32 kernel and the bf16 ring kernel alike): two partial sums per sample, one float2 per lane
        for (int n = l; n < N; n += kWave) {
            const float2 v = *reinterpret_cast<const float2 *>(head_part + (s0 + n) * 2);
            raw[n] = ((0.0f + v.x) + v.y) + head_b[0];
        }
    } else if (sizeof(T) == 2 && H == 4 && slots % 8 == 0) {
        // (round 5: the loads of four of these sample groups issued before the first add - 16 float4 per lane in flight - measured no gain,
        // 29.8 against 29.9 us: the kernel's time is its ~1900 vector instructions per ray, profiles/r05/hbm_kernels_sq_counters.json)
        const int per = slots / 8, lane8 = l & 7;
        for (int n = l >> 3; n < N; n += kWave >> 3) {
            const float4 *p = reinterpret_cast<const float4 *>(head_part + ((s0 + n) * slots + lane8 * per) * 4);
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int q = 0; q < per; ++q) {
                const float4 v = p[q];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            a.x = row8_sum(a.x); a.y = row8_sum(a.y); a.z = row8_sum(a.z); a.w = row8_sum(a.w);
            if (lane8 == 0) {
                raw[n * 4 + 0] = a.x + head_b[0];
                raw[n * 4 + 1] = a.y + head_b[1];
                raw[n * 4 + 2] = a.z + head_b[2];
                raw[n * 4 + 3] = a.w + head_b[3];
            }
        }
    } else {
        for (int idx = l; idx < N * H; idx += kWave) {
            const int n = idx / H, hh = idx % H;
            const float *p = head_part + ((s0 + n) * slots) * H + hh;
            float a = 0.0f;
            for (int q = 0; q < slots; ++q) a += p[q * H];
            raw[idx] = a + head_b[hh];
        }
    }
}

// Rays per workgroup of the stage finishers.  Round 4: a workgroup takes kFinishRays rays.  When the fused last layer covered
// all of their samples (every ray of the rendering forward but the last partial 256-row tile) each WAVE finishes one ray on its
// own - partial head sums, activations, fp64 transmittance scan, resampling / composite - with no workgroup barrier; until
// round 3 one 256-thread workgroup per ray parked three of its four waves after the head sums (0.07 / 0.27 of the HBM peak).
// A workgroup that holds rows the fused layer did not cover runs its rays one after the other in the old workgroup-wide form.
constexpr int kFinishRays = kFinishThreads / kWave;

// model.py:52,92-93 + intern/ray.py:136-149
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(kFinishThreads) void prop_finish_kernel(
    const T *__restrict__ act, int ld, const float *__restrict__ head_part, long fused_rows, int slots,
    const float *__restrict__ head_w, const float *__restrict__ head_b, int k_pad, float density_bias,
    const float *__restrict__ t_vals, const float *__restrict__ dirs, const float *__restrict__ u_rand, int B, int N, int ns,
    float padding, float *__restrict__ weights, float *__restrict__ t_new, int rpb /* rays per workgroup: kFinishRays, or 1 when four rays' buffers exceed the LDS */,
    const unsigned char *__restrict__ nanflag /* bf16 modes: 1 = the sample had a NaN feature (encode_features_wave_kernel), or NULL */,
    rng_t rng /* randomized inverse CDF drawn here (u_rand == NULL and rng.on) */,
    float *__restrict__ raw_out /* [B*N]: the head sums as formed here, for finish_backward_kernel (tape-keeping forward), or NULL */) {
    extern __shared__ float smem[];
    const int l = lane_id(), wave = threadIdx.x >> 6;
    const int nb = N + 1;
    const int b0 = blockIdx.x * rpb;
    const int b_end = (b0 + rpb < B) ? b0 + rpb : B;
    // the tail of one ray: activation, weights (model.py:59-78), blur + inverse-CDF resampling (intern/ray.py:136-149), by one wave
    auto finish_ray = [&](int b, float *t, float *rho, float *w, float *w2, float *cdf) __attribute__((always_inline)) {
        // bf16 modes: the head output of a sample with a NaN feature is NaN, as nn.ReLU would have carried it here (the bf16 pipe's
        // ReLU drops every NaN: see the encoder)
        for (int i = l; i < N; i += kWave) {
            if (raw_out != nullptr) raw_out[(long)b * N + i] = rho[i];
            const float raw = (nanflag != nullptr && nanflag[(long)b * N + i]) ? __builtin_nanf("") : rho[i];
            rho[i] = softplusf_(raw + density_bias);
        }
        wave_sync();
        wave_weights(t, rho, 1, dir_norm(dirs, b), N, w);
        wave_sync();
        for (int i = l; i < N; i += kWave) weights[(long)b * N + i] = w[i];
        if (t_new == nullptr) return;
        wave_blur(w, N, padding, w2);
        wave_sync();
        wave_sorted_pdf(t, w2, cdf, nb, ns, u_rand ? u_rand + (long)b * ns : nullptr, t_new + (long)b * ns, rng, (long)b * ns);
    };
    if (fused_rows >= (long)b_end * N) {  // workgroup-uniform: one wave per ray, no workgroup barrier
        const int b = b0 + wave;
        if (wave >= rpb || b >= B) return;
        float *t = smem + wave * 5 * nb, *rho = t + nb, *w = rho + nb, *w2 = w + nb, *cdf = w2 + nb;
        for (int i = l; i < nb; i += kWave) t[i] = t_vals[(long)b * nb + i];
        ray_heads_fused_wave<1, T>(head_part, slots, head_b, (long)b * N, N, rho);
        wave_sync();
        finish_ray(b, t, rho, w, w2, cdf);
        return;
    }
    float *hw = smem, *t = hw + k_pad, *rho = t + nb, *w = rho + nb, *w2 = w + nb, *cdf = w2 + nb;
    for (int b = b0; b < b_end; ++b) {
        for (int i = threadIdx.x; i < nb; i += blockDim.x) t[i] = t_vals[(long)b * nb + i];
        ray_heads<1, T, SPLIT>(act, ld, head_part, fused_rows, slots, head_w, head_b, k_pad, b, N, hw, rho);
        if (threadIdx.x < kWave) finish_ray(b, t, rho, w, w2, cdf);
        __syncthreads();  // the next ray reuses the buffers
    }
}

// model.py:150-158,180-186 + intern/ray.py:155-191
// (4 waves per SIMD asked of the compiler - at most 128 VGPRs: the launch is B / 4 workgroups of one wave per SIMD each, 4 per CU at 4096
// rays; the fp32 form had 132 and ran its last quarter of workgroups in a second round)
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(kFinishThreads, 4) void nerf_finish_kernel(
    const T *__restrict__ act, int ld, const float *__restrict__ head_part, long fused_rows, int slots,
    const float *__restrict__ head_w, const float *__restrict__ head_b, int k_pad, float density_bias,
    float rgb_padding, const float *__restrict__ t_vals, const float *__restrict__ dirs, int B, int N, int white_bkgd,
    float *__restrict__ comp_rgb, float *__restrict__ distance, float *__restrict__ acc,
    float *__restrict__ weights, float *__restrict__ t_out, float *__restrict__ s_out, const float *__restrict__ near,
    const float *__restrict__ far, int ts_calls, int rpb, const unsigned char *__restrict__ nanflag,
    float *__restrict__ raw_out /* [B*N][4]: the head sums as formed here, for finish_backward_kernel (tape-keeping forward), or NULL */) {
    extern __shared__ float smem[];
    const int l = lane_id(), wave = threadIdx.x >> 6;
    const int nb = N + 1;
    const int b0 = blockIdx.x * rpb;
    const int b_end = (b0 + rpb < B) ? b0 + rpb : B;
    // the tail of one ray by one wave: head activations (model.py:180-186), composite (intern/ray.py:155-191) and the two
    // tensors nerf_net.forward returns beside it (model.py:194-196): t_vals + 1e-6 (what g() inside t_to_s leaves behind) and
    // s_vals = t_to_s(t_vals, near, far) - until round 3 two more launches (add_eps_kernel, t_to_s_kernel), same arithmetic
    auto finish_ray = [&](int b, float *t, float *raw, float *w) __attribute__((always_inline)) {
        for (int i = l; i < N; i += kWave) {
            if (raw_out != nullptr) *reinterpret_cast<float4 *>(raw_out + ((long)b * N + i) * 4) = make_float4(raw[4 * i], raw[4 * i + 1], raw[4 * i + 2], raw[4 * i + 3]);
            if (nanflag != nullptr && nanflag[(long)b * N + i]) {  // bf16 modes: see prop_finish_kernel
#pragma unroll
                for (int c = 0; c < 4; ++c) raw[4 * i + c] = __builtin_nanf("");
            }
            raw[4 * i] = softplusf_(sigmoidf_(raw[4 * i]) + density_bias);
#pragma unroll
            for (int c = 1; c < 4; ++c)
                raw[4 * i + c] = sigmoidf_(raw[4 * i + c]) * (1.0f + 2.0f * rgb_padding) - rgb_padding;
        }
        wave_sync();
        wave_weights(t, raw, 4, dir_norm(dirs, b), N, w);
        wave_sync();
        const Composite c = wave_composite(t, w, raw + 1, 4, N, white_bkgd != 0);
        if (l == 0) {
            comp_rgb[3 * b] = c.r;
            comp_rgb[3 * b + 1] = c.g;
            comp_rgb[3 * b + 2] = c.b;
            distance[b] = c.dist;
            acc[b] = c.acc;
        }
        if (weights != nullptr)
            for (int i = l; i < N; i += kWave) weights[(long)b * N + i] = w[i];
        if (t_out != nullptr)
            for (int i = l; i < nb; i += kWave) t_out[(long)b * nb + i] = t[i] + kEpsG;
        if (s_out != nullptr) {
            const float gn1 = g_calls(near[b], ts_calls + 1), gf = g_calls(far[b], ts_calls + 1), gn2 = g_calls(near[b], ts_calls + 2);
            for (int i = l; i < nb; i += kWave) s_out[(long)b * nb + i] = (1.0f / (t[i] + kEpsG) - gn1) / (gf - gn2);
        }
    };
    if (fused_rows >= (long)b_end * N) {  // workgroup-uniform: one wave per ray, no workgroup barrier
        const int b = b0 + wave;
        if (wave >= rpb || b >= B) return;
        float *t = smem + wave * (2 * nb + 4 * N), *raw = t + nb, *w = raw + 4 * N;
        for (int i = l; i < nb; i += kWave) t[i] = t_vals[(long)b * nb + i];
        ray_heads_fused_wave<4, T>(head_part, slots, head_b, (long)b * N, N, raw);
        wave_sync();
        finish_ray(b, t, raw, w);
        return;
    }
    float *hw = smem, *t = hw + 4 * k_pad, *raw = t + nb, *w = raw + 4 * N;
    for (int b = b0; b < b_end; ++b) {
        for (int i = threadIdx.x; i < nb; i += blockDim.x) t[i] = t_vals[(long)b * nb + i];
        ray_heads<4, T, SPLIT>(act, ld, head_part, fused_rows, slots, head_w, head_b, k_pad, b, N, hw, raw);
        if (threadIdx.x < kWave) finish_ray(b, t, raw, w);
        __syncthreads();  // the next ray reuses the buffers
    }
}


// ------------------------------------------------------------------------------------------
// Backward of the stage finishers: what autograd does for model.py:52,92-93 (H = 1) and
// model.py:150-158,180-186 + intern/ray.py:171-191 (H = 4) under train.py:62,80.  One 256-thread workgroup per ray.
//   phase 1  the H head sums of the ray's N samples: the forward's own, kept on the training tape (raw_in; round 6: the stage backwards of
//            m360_capi.hip - one pass over the ray's activation rows less), or re-evaluated (head_dots; the stand-alone C-ABI entry points)
//   phase 2  wave 0: activations, transmittance scan, then the reverse scan
//              x_i = rho_i delta_i,  w_i = (1 - e^{-x_i}) T_i,  T_i = exp(-sum_{j<i} x_j)
//              dL/dx_i = g_i T_{i+1} - sum_{k>i} g_k w_k      (g = total gradient reaching w)
//            and the chain through softplus / sigmoid / rgb padding -> d raw[N][H]
//   phase 3  dz[s,:] = (sum_h d raw[s,h] head_w[h,:]) * a(1 - a)  (a = sigmoid output of the last hidden layer),
//            per-ray partial sums of d head_w[h,:] = sum_s d raw[s,h] a[s,:] and d head_b (reduced later, fixed order)
// T = float, or __bf16 (the bf16 training path, round 5): the stored activations are read and dz is written in T, everything in between is fp32
template <typename T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct Quad<__bf16> {
    typedef __bf16 bf16x4_q __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float4 load(const __bf16 *p) {
        const bf16x4_q v = *reinterpret_cast<const bf16x4_q *>(p);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    static __device__ __forceinline__ void store(__bf16 *p, float4 v) {
        const bf16x4_q o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<bf16x4_q *>(p) = o;
    }
};
template <int H, typename T = float>
__global__ __launch_bounds__(kFinishThreads) void finish_backward_kernel(
    const T *__restrict__ act, int ld, const float *__restrict__ head_w, const float *__restrict__ head_b,
    int k_pad, float density_bias, float rgb_padding, const float *__restrict__ t_vals,
    const float *__restrict__ dirs, int N, int white_bkgd, const float *__restrict__ g_rgb,
    const float *__restrict__ g_dist, const float *__restrict__ g_acc, const float *__restrict__ g_w,
    T *__restrict__ dz, float *__restrict__ part_hw /*[B*groups][H*k_pad]*/,
    float *__restrict__ part_hb /*[B][H]*/, int groups, const float *__restrict__ raw_in /*[B*N][H]: the forward's head sums, or NULL*/) {
    extern __shared__ float smem[];
    const int b = blockIdx.x, l = lane_id();
    const int nb = N + 1;
    float *hw = smem, *t = hw + H * k_pad, *raw = t + nb, *w = raw + H * N, *tnext = w + N, *gw = tnext + N,
          *draw = gw + N;
    for (int i = threadIdx.x; i < H * k_pad; i += blockDim.x) hw[i] = head_w[i];
    for (int i = threadIdx.x; i < nb; i += blockDim.x) t[i] = t_vals[(long)b * nb + i];
    __syncthreads();
    const T *act_ray = act + (long)b * N * ld;
    if (raw_in != nullptr) {
        for (int i = threadIdx.x; i < N * H; i += blockDim.x) raw[i] = raw_in[(long)b * N * H + i];
    } else {
        head_dots<H, T>(act_ray, ld, hw, head_b, k_pad, N, raw);
    }
    __syncthreads();
    if (threadIdx.x < kWave) {
        const float dnorm = dir_norm(dirs, b);
        // pass A: weights, T_{i+1}, accumulated opacity and the distance numerator
        double carry = 0.0;
        float sa = 0.0f, sd = 0.0f;
        for (int base = 0; base < N; base += kWave) {
            const int i = base + l;
            float x = 0.0f;
            if (i < N) {
                const float rho = H == 1 ? softplusf_(raw[i] + density_bias) : softplusf_(sigmoidf_(raw[H * i]) + density_bias);
                x = rho * ((t[i + 1] - t[i]) * dnorm);
            }
            const double incl = wave_incl_scan_d((double)x);
            if (i < N) {
                const float wi = (1.0f - expf(-x)) * expf(-(float)(carry + incl - (double)x));
                w[i] = wi;
                tnext[i] = expf(-(float)(carry + incl));
                sa += wi;
                sd += wi * (0.5f * (t[i] + t[i + 1]));
            }
            carry += __shfl(incl, kWave - 1, kWave);
        }
        sa = wave_sum(sa);
        sd = wave_sum(sd);
        // pass B: total gradient reaching each weight
        float gr = 0.0f, gg = 0.0f, gb = 0.0f, ga = 0.0f, gd_scale = 0.0f, dval = 0.0f;
        if (H == 4) {
            if (g_rgb) {
                gr = g_rgb[3 * b];
                gg = g_rgb[3 * b + 1];
                gb = g_rgb[3 * b + 2];
            }
            if (g_acc) ga = g_acc[b];
            if (white_bkgd) ga -= gr + gg + gb;  // comp_rgb += 1 - acc
            if (g_dist) {                        // distance = clamp(nan_to_num(sd / sa), t_0, t_N): gradient only inside
                dval = sd / sa;
                if (dval == dval && fabsf(dval) <= 3.4028234e38f && dval >= t[0] && dval <= t[N]) gd_scale = g_dist[b] / sa;
                else dval = 0.0f;
            }
        }
        wave_sync();
        double total = 0.0;
        for (int base = 0; base < N; base += kWave) {
            const int i = base + l;
            float g = 0.0f;
            if (i < N) {
                if (g_w) g = g_w[(long)b * N + i];
                if (H == 4) {
                    const float k = 1.0f + 2.0f * rgb_padding;
                    g += ga + gr * (sigmoidf_(raw[4 * i + 1]) * k - rgb_padding) + gg * (sigmoidf_(raw[4 * i + 2]) * k - rgb_padding) +
                         gb * (sigmoidf_(raw[4 * i + 3]) * k - rgb_padding);
                    g += gd_scale * (0.5f * (t[i] + t[i + 1]) - dval);
                }
                gw[i] = g;
            }
            total += (double)(i < N ? g * w[i] : 0.0f);
        }
        total = wave_sum_d(total);
        // pass C: reverse scan and the activation chain
        double pcarry = 0.0;
        for (int base = 0; base < N; base += kWave) {
            const int i = base + l;
            const float gwi = i < N ? gw[i] * w[i] : 0.0f;
            const double incl = wave_incl_scan_d((double)gwi);
            if (i < N) {
                const float suffix = (float)(total - (pcarry + incl));
                const float dx = gw[i] * tnext[i] - suffix;
                const float drho = dx * ((t[i + 1] - t[i]) * dnorm);
                if (H == 1) {
                    draw[i] = drho * sigmoidf_(raw[i] + density_bias);
                } else {
                    const float s0 = sigmoidf_(raw[4 * i]);
                    draw[4 * i] = drho * sigmoidf_(s0 + density_bias) * s0 * (1.0f - s0);
                    const float k = (1.0f + 2.0f * rgb_padding) * w[i];
                    const float s1 = sigmoidf_(raw[4 * i + 1]), s2 = sigmoidf_(raw[4 * i + 2]), s3 = sigmoidf_(raw[4 * i + 3]);
                    draw[4 * i + 1] = k * gr * s1 * (1.0f - s1);
                    draw[4 * i + 2] = k * gg * s2 * (1.0f - s2);
                    draw[4 * i + 3] = k * gb * s3 * (1.0f - s3);
                }
            }
            pcarry += __shfl(incl, kWave - 1, kWave);
        }
        wave_sync();
        // head bias gradient of this ray
        float hb[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) hb[hh] = 0.0f;
        for (int i = l; i < N; i += kWave)
#pragma unroll
            for (int hh = 0; hh < H; ++hh) hb[hh] += draw[H * i + hh];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            hb[hh] = wave_sum(hb[hh]);
            if (l == 0) part_hb[(long)b * H + hh] = hb[hh];
        }
    }
    __syncthreads();
    // phase 3: thread = (sample group, 16-byte column chunk); every group walks its samples n = group, group+groups, ..
    const int kc = k_pad / 4;
    const int per = kc < (int)blockDim.x ? kc : (int)blockDim.x;  // threads per group
    const int group = threadIdx.x / per, tc = threadIdx.x % per;
    if (group >= groups) return;
    T *__restrict__ dz_ray = dz + (long)b * N * ld;
    for (int c = tc; c < kc; c += per) {
        float4 hwc[H], accw[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            hwc[hh] = *reinterpret_cast<const float4 *>(hw + hh * k_pad + 4 * c);
            accw[hh] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        for (int n = group; n < N; n += groups) {
            const float4 a = Quad<T>::load(act_ray + (long)n * ld + 4 * c);
            float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int hh = 0; hh < H; ++hh) {
                const float dr = draw[H * n + hh];
                d.x += dr * hwc[hh].x;
                d.y += dr * hwc[hh].y;
                d.z += dr * hwc[hh].z;
                d.w += dr * hwc[hh].w;
                accw[hh].x += dr * a.x;
                accw[hh].y += dr * a.y;
                accw[hh].z += dr * a.z;
                accw[hh].w += dr * a.w;
            }
            d.x *= a.x * (1.0f - a.x);
            d.y *= a.y * (1.0f - a.y);
            d.z *= a.z * (1.0f - a.z);
            d.w *= a.w * (1.0f - a.w);
            Quad<T>::store(dz_ray + (long)n * ld + 4 * c, d);
        }
#pragma unroll
        for (int hh = 0; hh < H; ++hh)
            *reinterpret_cast<float4 *>(part_hw + ((long)b * groups + group) * (H * k_pad) + hh * k_pad + 4 * c) = accw[hh];
    }
}

}  // namespace m360

// =========================================================================================
namespace m360 {
int resample_t_any(const float *t_vals, const float *weights, const float *u_rand, int B, int N, int num_out, float resample_padding, float *t_new,
                   const rng_t &rng, m360_stream_t stream);
static int sorted_pdf_any(const float *bins, const float *weights, const float *u_rand, int B, int nb, int num_samples, float *samples, const rng_t &rng, m360_stream_t stream);
}  // namespace m360
using namespace m360;

static inline hipStream_t S_(m360_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
#define M360_RAY_TRY(expr)          \
    do {                            \
        const int rc_ = (expr);     \
        if (rc_ != M360_OK) return rc_; \
    } while (0)
static constexpr size_t kMaxDynLds = 64 * 1024;

extern "C" {

int m360_density_to_weight(const float *t_vals, const float *density, const float *dirs, int B, int N,
                           float *weights, m360_stream_t stream) {
    if (!t_vals || !density || !dirs || !weights || B < 0 || N < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_density_to_weight: bad argument");
    if (B == 0) return M360_OK;
    const size_t lds = (size_t)kRayWaves * (3 * N + 1) * sizeof(float);
    if (lds > kMaxDynLds) return fail(M360_ERR_INVALID_ARGUMENT, "m360_density_to_weight: N=%d too large for LDS", N);
    hipLaunchKernelGGL(density_to_weight_kernel, dim3((B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * kWave), lds, S_(stream), t_vals, density, dirs, B, N, weights);
    return check_launch("density_to_weight");
}

int m360_sorted_pdf(const float *bins, const float *weights, const float *u_rand, int B, int nb,
                    int num_samples, float *samples, m360_stream_t stream) {
    return sorted_pdf_any(bins, weights, u_rand, B, nb, num_samples, samples, rng_t{0, 0, 0}, stream);
}

int m360_sorted_pdf_philox(const float *bins, const float *weights, int B, int nb, int num_samples, unsigned long long seed,
                           unsigned long long offset, float *samples, m360_stream_t stream) {
    return sorted_pdf_any(bins, weights, nullptr, B, nb, num_samples, samples, rng_t{seed, offset, 1}, stream);
}

int m360_resample_t(const float *t_vals, const float *weights, const float *u_rand, int B, int N,
                    float resample_padding, float *t_new, m360_stream_t stream) {
    return m360_resample_t_n(t_vals, weights, u_rand, B, N, N + 1, resample_padding, t_new, stream);
}

int m360_resample_t_n(const float *t_vals, const float *weights, const float *u_rand, int B, int N, int num_out,
                      float resample_padding, float *t_new, m360_stream_t stream) {
    return resample_t_any(t_vals, weights, u_rand, B, N, num_out, resample_padding, t_new, rng_t{0, 0, 0}, stream);
}

int m360_resample_t_philox(const float *t_vals, const float *weights, int B, int N, int num_out, float resample_padding,
                           unsigned long long seed, unsigned long long offset, float *t_new, m360_stream_t stream) {
    return resample_t_any(t_vals, weights, nullptr, B, N, num_out, resample_padding, t_new, rng_t{seed, offset, 1}, stream);
}

int m360_volumetric_rendering(const float *rgb, const float *density, const float *t_vals,
                              const float *dirs, int B, int N, int white_bkgd, float *comp_rgb,
                              float *distance, float *acc, float *weights, m360_stream_t stream) {
    if (!rgb || !density || !t_vals || !dirs || !comp_rgb || !distance || !acc || B < 0 || N < 1)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_volumetric_rendering: bad argument");
    if (B == 0) return M360_OK;
    const size_t lds = (size_t)kRayWaves * (6 * N + 1) * sizeof(float);
    if (lds > kMaxDynLds) return fail(M360_ERR_INVALID_ARGUMENT, "m360_volumetric_rendering: N=%d too large for LDS", N);
    hipLaunchKernelGGL(volumetric_rendering_kernel, dim3((B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * kWave), lds, S_(stream), rgb, density, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights);
    return check_launch("volumetric_rendering");
}

int m360_to8b(const float *x, long n, uint8_t *out, m360_stream_t stream) {
    if (!x || !out || n < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_to8b: bad argument");
    if (n == 0) return M360_OK;
    hipLaunchKernelGGL(to8b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), x, n, out);
    return check_launch("to8b");
}

int m360_prop_finish(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                     float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                     int B, int N, float resample_padding, float *weights, float *t_new,
                     m360_stream_t stream) {
    return m360_prop_finish_n(act, ld, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, N + 1, resample_padding, weights, t_new, stream);
}

static int prop_finish_any(const void *act, int bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                           float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                           int B, int N, int num_out, float resample_padding, float *weights, float *t_new,
                           m360_stream_t stream, const float *head_part = nullptr, long fused_rows = 0, int slots = 0,
                           const unsigned char *nanflag = nullptr, const rng_t &rng = rng_t{0, 0, 0}, float *raw_out = nullptr);

int m360_prop_finish_n(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                       float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                       int B, int N, int num_out, float resample_padding, float *weights, float *t_new,
                       m360_stream_t stream) {
    return prop_finish_any(act, 0, ld, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, num_out, resample_padding, weights, t_new, stream);
}

int m360_prop_finish_bf16(const void *act_bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                          float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                          int B, int N, int num_out, float resample_padding, float *weights, float *t_new,
                          m360_stream_t stream) {
    return prop_finish_any(act_bf16, 1, ld, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, num_out, resample_padding, weights, t_new, stream);
}

int m360_prop_finish_fused(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots,
                           const float *head_w, const float *head_b, int k_pad, float density_bias,
                           const float *t_vals, const float *dirs, const float *u_rand, int B, int N, int num_out,
                           float resample_padding, float *weights, float *t_new, m360_stream_t stream) {
    if (fused_rows < 0 || (fused_rows > 0 && (!head_part || slots < 1))) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_finish_fused: fused_rows=%ld slots=%d head_part=%p", fused_rows, slots, (const void *)head_part);
    return prop_finish_any(act, act_bf16, ld, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, num_out, resample_padding, weights, t_new, stream, head_part, fused_rows, slots);
}

static int prop_finish_any(const void *act, int bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                           float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                           int B, int N, int num_out, float resample_padding, float *weights, float *t_new,
                           m360_stream_t stream, const float *head_part, long fused_rows, int slots, const unsigned char *nanflag,
                           const rng_t &rng, float *raw_out) {
    if (num_out < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_finish: num_out=%d", num_out);
    const int align = bf16 ? 8 : 4;
    if (!act || !head_w || !head_b || !t_vals || !dirs || !weights || B < 0 || N < 1 || k_pad < align || k_pad % align != 0 || ld < (bf16 == 2 ? 2 * k_pad : k_pad) || ld % align != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_finish: bad argument");
    if (B == 0) return M360_OK;
    // workgroup-wide form: head row + one ray's buffers; wave-per-ray form: kFinishRays rays' buffers
    const size_t lds_wg = ((size_t)k_pad + 5 * (N + 1)) * sizeof(float), lds_wave = (size_t)kFinishRays * 5 * (N + 1) * sizeof(float);
    const int rpb = (lds_wave <= kMaxDynLds) ? kFinishRays : 1;  // very long rays: one ray per workgroup, as before round 4
    const size_t lds = rpb == 1 ? lds_wg : (lds_wg > lds_wave ? lds_wg : lds_wave);
    if (lds > kMaxDynLds) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_finish: k_pad=%d N=%d too large for LDS", k_pad, N);
    const dim3 grid((unsigned)((B + rpb - 1) / rpb));
    if (bf16 == 2) hipLaunchKernelGGL((prop_finish_kernel<__bf16, true>), grid, dim3(kFinishThreads), lds, S_(stream), static_cast<const __bf16 *>(act), ld, head_part, fused_rows, slots, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, num_out, resample_padding, weights, t_new, rpb, nanflag, rng, raw_out);
    else if (bf16) hipLaunchKernelGGL(prop_finish_kernel<__bf16>, grid, dim3(kFinishThreads), lds, S_(stream), static_cast<const __bf16 *>(act), ld, head_part, fused_rows, slots, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, num_out, resample_padding, weights, t_new, rpb, nanflag, rng, raw_out);
    else hipLaunchKernelGGL(prop_finish_kernel<float>, grid, dim3(kFinishThreads), lds, S_(stream), static_cast<const float *>(act), ld, head_part, fused_rows, slots, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, num_out, resample_padding, weights, t_new, rpb, nanflag, rng, raw_out);
    return check_launch("prop_finish");
}

struct FinishExtras {  // what nerf_net.forward returns beside the composite (model.py:194-196), written by the same kernel
    float *t_out = nullptr, *s_out = nullptr;
    const float *near = nullptr, *far = nullptr;
    int calls = 1;  // applications of the reference's in-place g() that near / far have behind them (t_to_s_kernel)
    const unsigned char *nanflag = nullptr;  // bf16 modes: per-sample NaN flags of the encoder
    float *raw_out = nullptr;                // tape-keeping forward: the head sums [B*N][4] for the backward
};
static int nerf_finish_any(const void *act, int bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                           float density_bias, float rgb_padding, const float *t_vals, const float *dirs, int B,
                           int N, int white_bkgd, float *comp_rgb, float *distance, float *acc, float *weights,
                           m360_stream_t stream, const float *head_part = nullptr, long fused_rows = 0, int slots = 0,
                           FinishExtras ex = FinishExtras());

int m360_nerf_finish(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                     float density_bias, float rgb_padding, const float *t_vals, const float *dirs, int B,
                     int N, int white_bkgd, float *comp_rgb, float *distance, float *acc, float *weights,
                     m360_stream_t stream) {
    return nerf_finish_any(act, 0, ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, stream);
}

int m360_nerf_finish_bf16(const void *act_bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                          float density_bias, float rgb_padding, const float *t_vals, const float *dirs, int B,
                          int N, int white_bkgd, float *comp_rgb, float *distance, float *acc, float *weights,
                          m360_stream_t stream) {
    return nerf_finish_any(act_bf16, 1, ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, stream);
}

int m360_nerf_finish_fused(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots,
                           const float *head_w, const float *head_b, int k_pad, float density_bias, float rgb_padding,
                           const float *t_vals, const float *dirs, int B, int N, int white_bkgd, float *comp_rgb,
                           float *distance, float *acc, float *weights, m360_stream_t stream) {
    if (fused_rows < 0 || (fused_rows > 0 && (!head_part || slots < 1))) return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_finish_fused: fused_rows=%ld slots=%d head_part=%p", fused_rows, slots, (const void *)head_part);
    return nerf_finish_any(act, act_bf16, ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, stream, head_part, fused_rows, slots);
}

int m360_nerf_finish_outputs(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots,
                             const float *head_w, const float *head_b, int k_pad, float density_bias, float rgb_padding,
                             const float *t_vals, const float *dirs, const float *near, const float *far, int near_far_calls,
                             int B, int N, int white_bkgd, float *comp_rgb, float *distance, float *acc, float *weights,
                             float *t_vals_out, float *s_vals_out, m360_stream_t stream) {
    if (fused_rows < 0 || (fused_rows > 0 && (!head_part || slots < 1))) return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_finish_outputs: fused_rows=%ld slots=%d head_part=%p", fused_rows, slots, (const void *)head_part);
    if (s_vals_out && (!near || !far || near_far_calls < 0)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_finish_outputs: s_vals needs near, far and near_far_calls >= 0");
    FinishExtras ex;
    ex.t_out = t_vals_out, ex.s_out = s_vals_out, ex.near = near, ex.far = far, ex.calls = near_far_calls;
    return nerf_finish_any(act, act_bf16, ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, stream, head_part, fused_rows, slots, ex);
}

static int nerf_finish_any(const void *act, int bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                           float density_bias, float rgb_padding, const float *t_vals, const float *dirs, int B,
                           int N, int white_bkgd, float *comp_rgb, float *distance, float *acc, float *weights,
                           m360_stream_t stream, const float *head_part, long fused_rows, int slots, FinishExtras ex) {
    const int align = bf16 ? 8 : 4;
    if (!act || !head_w || !head_b || !t_vals || !dirs || !comp_rgb || !distance || !acc || B < 0 || N < 1 || k_pad < align || k_pad % align != 0 || ld < (bf16 == 2 ? 2 * k_pad : k_pad) || ld % align != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_finish: bad argument");
    if (B == 0) return M360_OK;
    const size_t ray_floats = (size_t)2 * (N + 1) + 4 * N;
    const size_t lds_wg = ((size_t)4 * k_pad + ray_floats) * sizeof(float), lds_wave = (size_t)kFinishRays * ray_floats * sizeof(float);
    const int rpb = (lds_wave <= kMaxDynLds) ? kFinishRays : 1;
    const size_t lds = rpb == 1 ? lds_wg : (lds_wg > lds_wave ? lds_wg : lds_wave);
    if (lds > kMaxDynLds) return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_finish: k_pad=%d N=%d too large for LDS", k_pad, N);
    const dim3 grid((unsigned)((B + rpb - 1) / rpb));
    if (bf16 == 2) hipLaunchKernelGGL((nerf_finish_kernel<__bf16, true>), grid, dim3(kFinishThreads), lds, S_(stream), static_cast<const __bf16 *>(act), ld, head_part, fused_rows, slots, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, ex.t_out, ex.s_out, ex.near, ex.far, ex.calls, rpb, ex.nanflag, ex.raw_out);
    else if (bf16) hipLaunchKernelGGL(nerf_finish_kernel<__bf16>, grid, dim3(kFinishThreads), lds, S_(stream), static_cast<const __bf16 *>(act), ld, head_part, fused_rows, slots, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, ex.t_out, ex.s_out, ex.near, ex.far, ex.calls, rpb, ex.nanflag, ex.raw_out);
    else hipLaunchKernelGGL(nerf_finish_kernel<float>, grid, dim3(kFinishThreads), lds, S_(stream), static_cast<const float *>(act), ld, head_part, fused_rows, slots, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, ex.t_out, ex.s_out, ex.near, ex.far, ex.calls, rpb, ex.nanflag, ex.raw_out);
    return check_launch("nerf_finish");
}

// ---- backward of the stage finishers (training path)
static int finish_groups(int k_pad) {
    const int kc = k_pad / 4;
    return kc >= kFinishThreads ? 1 : kFinishThreads / kc;
}
static constexpr int kFinishSlices = 64;
static inline size_t fb_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t m360_finish_backward_workspace_bytes(int B, int heads, int k_pad) {
    if (B < 0 || heads < 1 || k_pad < 4) return 0;
    const size_t C = (size_t)heads * k_pad;
    return fb_up((size_t)B * finish_groups(k_pad) * C * sizeof(float)) + fb_up((size_t)B * heads * sizeof(float)) +
           fb_up((size_t)kFinishSlices * C * sizeof(float));
}

extern "C++" {
template <int H, typename T = float>
static int finish_backward_any(const T *act, int ld, const float *head_w, const float *head_b, int k_pad,
                               float density_bias, float rgb_padding, const float *t_vals, const float *dirs, int B,
                               int N, int white_bkgd, const float *g_rgb, const float *g_dist, const float *g_acc,
                               const float *g_w, T *dz, float *grad_head_w, float *grad_head_b, void *workspace,
                               size_t workspace_bytes, m360_stream_t stream, const char *who, const float *raw_in = nullptr) {
    constexpr int kAlign = sizeof(T) == 2 ? 8 : 4;  // bf16 rows are read in 16-byte chunks by head_dots
    if (!act || !head_w || !head_b || !t_vals || !dirs || !dz || !grad_head_w || !grad_head_b || B < 0 || N < 1 || k_pad < kAlign || k_pad % kAlign || ld < k_pad || ld % kAlign)
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: bad argument", who);
    if (B == 0) return M360_OK;
    const size_t need = m360_finish_backward_workspace_bytes(B, H, k_pad);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "%s: workspace %zu < %zu bytes (or not 256-byte aligned)", who, workspace_bytes, need);
    const size_t lds = ((size_t)H * k_pad + (N + 1) + (size_t)(2 * H + 3) * N) * sizeof(float);
    if (lds > kMaxDynLds) return fail(M360_ERR_INVALID_ARGUMENT, "%s: k_pad=%d N=%d too large for LDS", who, k_pad, N);
    const int groups = finish_groups(k_pad), C = H * k_pad;
    char *ws = static_cast<char *>(workspace);
    float *part_hw = reinterpret_cast<float *>(ws);
    float *part_hb = reinterpret_cast<float *>(ws + fb_up((size_t)B * groups * C * sizeof(float)));
    float *slices = reinterpret_cast<float *>(reinterpret_cast<char *>(part_hb) + fb_up((size_t)B * H * sizeof(float)));
    hipLaunchKernelGGL((finish_backward_kernel<H, T>), dim3(B), dim3(kFinishThreads), lds, S_(stream), act, ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, N, white_bkgd, g_rgb, g_dist, g_acc, g_w, dz, part_hw, part_hb, groups, raw_in);
    const int rc = check_launch(who);
    if (rc != M360_OK) return rc;
    M360_RAY_TRY(launch_colsum(part_hw, (long)B * groups, C, C, slices, kFinishSlices, grad_head_w, S_(stream)));
    return launch_colsum(part_hb, (long)B, H, H, slices, kFinishSlices, grad_head_b, S_(stream));
}
}  // extern "C++"

int m360_prop_finish_backward(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                              float density_bias, const float *t_vals, const float *dirs, int B, int N,
                              const float *grad_weights, float *dz, float *grad_head_w, float *grad_head_b,
                              void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    if (!grad_weights) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_finish_backward: grad_weights is required");
    return finish_backward_any<1>(act, ld, head_w, head_b, k_pad, density_bias, 0.0f, t_vals, dirs, B, N, 0, nullptr, nullptr, nullptr, grad_weights, dz, grad_head_w, grad_head_b, workspace, workspace_bytes, stream, "m360_prop_finish_backward");
}

int m360_nerf_finish_backward(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                              float density_bias, float rgb_padding, const float *t_vals, const float *dirs, int B,
                              int N, int white_bkgd, const float *grad_rgb, const float *grad_distance,
                              const float *grad_acc, const float *grad_weights, float *dz, float *grad_head_w,
                              float *grad_head_b, void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    return finish_backward_any<4>(act, ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, grad_rgb, grad_distance, grad_acc, grad_weights, dz, grad_head_w, grad_head_b, workspace, workspace_bytes, stream, "m360_nerf_finish_backward");
}

}  // extern "C"

namespace m360 {
// Stage backwards (m360_capi.hip; not part of the C-ABI): the finishers' backward on the activations of the training tape - fp32, or bf16
// (act_bf16 = 1: dz written in bf16 as well) - with the head sums the forward's finisher left on the tape (raw_in; NULL: re-evaluated)
int prop_finish_backward_stage(const void *act, int act_bf16, int ld, const float *head_w, const float *head_b, int k_pad, float density_bias, const float *t_vals,
                               const float *dirs, int B, int N, const float *grad_weights, void *dz, float *grad_head_w, float *grad_head_b,
                               void *workspace, size_t workspace_bytes, const float *raw_in, m360_stream_t stream) {
    if (!grad_weights) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_backward: grad_w_hat is required");
    if (act_bf16)
        return finish_backward_any<1, __bf16>(static_cast<const __bf16 *>(act), ld, head_w, head_b, k_pad, density_bias, 0.0f, t_vals, dirs, B, N, 0, nullptr, nullptr, nullptr, grad_weights,
                                              static_cast<__bf16 *>(dz), grad_head_w, grad_head_b, workspace, workspace_bytes, stream, "m360_prop_backward (bf16 finisher)", raw_in);
    return finish_backward_any<1, float>(static_cast<const float *>(act), ld, head_w, head_b, k_pad, density_bias, 0.0f, t_vals, dirs, B, N, 0, nullptr, nullptr, nullptr, grad_weights,
                                         static_cast<float *>(dz), grad_head_w, grad_head_b, workspace, workspace_bytes, stream, "m360_prop_backward (finisher)", raw_in);
}
int nerf_finish_backward_stage(const void *act, int act_bf16, int ld, const float *head_w, const float *head_b, int k_pad, float density_bias, float rgb_padding,
                               const float *t_vals, const float *dirs, int B, int N, int white_bkgd, const float *grad_rgb, const float *grad_distance,
                               const float *grad_acc, const float *grad_weights, void *dz, float *grad_head_w, float *grad_head_b, void *workspace,
                               size_t workspace_bytes, const float *raw_in, m360_stream_t stream) {
    if (act_bf16)
        return finish_backward_any<4, __bf16>(static_cast<const __bf16 *>(act), ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, grad_rgb, grad_distance,
                                              grad_acc, grad_weights, static_cast<__bf16 *>(dz), grad_head_w, grad_head_b, workspace, workspace_bytes, stream, "m360_nerf_backward (bf16 finisher)", raw_in);
    return finish_backward_any<4, float>(static_cast<const float *>(act), ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, grad_rgb, grad_distance,
                                         grad_acc, grad_weights, static_cast<float *>(dz), grad_head_w, grad_head_b, workspace, workspace_bytes, stream, "m360_nerf_backward (finisher)", raw_in);
}
// Stage drivers only (m360_capi.hip; not part of the C-ABI): the fused finishers with the encoder's per-sample NaN flags (bf16 / bf16x3
// modes; NULL = none) - and, for the NeRF stage, the t_vals + 1e-6 / s_vals outputs of m360_nerf_finish_outputs.
int prop_finish_stage(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots, const float *head_w,
                      const float *head_b, int k_pad, float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                      int B, int N, int num_out, float resample_padding, float *weights, float *t_new, const unsigned char *nanflag,
                      m360_stream_t stream, const rng_t &rng, float *raw_out) {
    if (fused_rows < 0 || (fused_rows > 0 && (!head_part || slots < 1))) return fail(M360_ERR_INVALID_ARGUMENT, "prop_finish_stage: fused_rows=%ld slots=%d", fused_rows, slots);
    return prop_finish_any(act, act_bf16, ld, head_w, head_b, k_pad, density_bias, t_vals, dirs, u_rand, B, N, num_out, resample_padding, weights, t_new, stream, head_part, fused_rows, slots, nanflag, rng, raw_out);
}
// m360_resample_t_n with the inverse CDF's uniforms from a tensor (u_rand), from the Philox stream (rng.on) or deterministic
int resample_t_any(const float *t_vals, const float *weights, const float *u_rand, int B, int N, int num_out, float resample_padding, float *t_new,
                   const rng_t &rng, m360_stream_t stream) {
    if (!t_vals || !weights || !t_new || B < 0 || N < 1 || num_out < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_resample_t: bad argument");
    if (B == 0) return M360_OK;
    const int nb = N + 1;
    const size_t lds = (size_t)kRayWaves * 4 * nb * sizeof(float);
    if (lds > kMaxDynLds) return fail(M360_ERR_INVALID_ARGUMENT, "m360_resample_t: N=%d too large for LDS", N);
    hipLaunchKernelGGL(resample_kernel<true>, dim3((B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * kWave), lds, S_(stream), t_vals, weights, u_rand, B, nb, num_out, resample_padding, t_new, rng);
    return check_launch("resample_t");
}
static int sorted_pdf_any(const float *bins, const float *weights, const float *u_rand, int B, int nb, int num_samples, float *samples, const rng_t &rng, m360_stream_t stream) {
    if (!bins || !weights || !samples || B < 0 || nb < 2 || num_samples < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_sorted_pdf: bad argument");
    if (B == 0) return M360_OK;
    const size_t lds = (size_t)kRayWaves * 4 * nb * sizeof(float);
    if (lds > kMaxDynLds) return fail(M360_ERR_INVALID_ARGUMENT, "m360_sorted_pdf: nb=%d too large for LDS", nb);
    hipLaunchKernelGGL(resample_kernel<false>, dim3((B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * kWave), lds, S_(stream), bins, weights, u_rand, B, nb, num_samples, 0.0f, samples, rng);
    return check_launch("sorted_pdf");
}
int nerf_finish_stage(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots, const float *head_w,
                      const float *head_b, int k_pad, float density_bias, float rgb_padding, const float *t_vals, const float *dirs,
                      const float *near, const float *far, int near_far_calls, int B, int N, int white_bkgd, float *comp_rgb,
                      float *distance, float *acc, float *weights, float *t_vals_out, float *s_vals_out, const unsigned char *nanflag,
                      m360_stream_t stream, float *raw_out) {
    if (fused_rows < 0 || (fused_rows > 0 && (!head_part || slots < 1))) return fail(M360_ERR_INVALID_ARGUMENT, "nerf_finish_stage: fused_rows=%ld slots=%d", fused_rows, slots);
    if (s_vals_out && (!near || !far || near_far_calls < 0)) return fail(M360_ERR_INVALID_ARGUMENT, "nerf_finish_stage: s_vals needs near, far and near_far_calls >= 0");
    FinishExtras ex;
    ex.t_out = t_vals_out, ex.s_out = s_vals_out, ex.near = near, ex.far = far, ex.calls = near_far_calls, ex.nanflag = nanflag, ex.raw_out = raw_out;
    return nerf_finish_any(act, act_bf16, ld, head_w, head_b, k_pad, density_bias, rgb_padding, t_vals, dirs, B, N, white_bkgd, comp_rgb, distance, acc, weights, stream, head_part, fused_rows, slots, ex);
}
}  // namespace m360
