// Sampling, conical-frustum gaussians, the reference's whole-tensor "contraction" and the
// encodings.  All kernels here are HBM/VALU-bound elementwise or
// reduction kernels: one thread per sample, coalesced SoA ray loads, LDS-staged feature rows.
#include "m360_common.hip.h"

namespace m360 {

constexpr int kNormPartials = 1024;  // fixed partition => deterministic reduction order
constexpr int kMaxNormGroups = 1024;  // chunks per launch in grouped mode
constexpr long kSmallGroup = 131072;  // samples: groups up to this size are reduced by ONE workgroup (see norm_group_*)

struct NormScratch {  // layout of the contraction workspace
    double partial[kNormPartials];
    float gnorm;  // Frobenius norm of the un-contracted means (whole batch, = gnorms[0] for a single group)
    float pad0;
    double sumsq;  // gnorm^2 before the square root (fp64): what ranks exchange when one batch is sharded
    float pad[12];
    float gnorms[kMaxNormGroups];  // grouped mode: one norm per chunk of `group_rays` rays
};

// ------------------------------------------------------------------------------------------
// intern/ray.py:99-110
__global__ void sample_t_kernel(const float *__restrict__ near, const float *__restrict__ far,
                                const float *__restrict__ t_rand, int B, int N,
                                float *__restrict__ t_vals, rng_t rng = rng_t{0, 0, 0}) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int M = N + 1;
    if (idx >= (long)B * M) return;
    const int b = (int)(idx / M), i = (int)(idx % M);
    const float gf = 1.0f / (far[b] + kEpsG), gn = 1.0f / (near[b] + kEpsG);
    auto t_at = [&](int j) {
        const float s = linspacef_(0.0f, 1.0f, M, j);
        const float mix = s * gf + (1.0f - s) * gn;
        return 1.0f / (mix + kEpsG);
    };
    float t = t_at(i);
    if (t_rand != nullptr || rng.on) {  // intern/ray.py:103-108
        const float lower = (i == 0) ? t : 0.5f * (t + t_at(i - 1));
        const float upper = (i == N) ? t : 0.5f * (t_at(i + 1) + t);
        t = lower + (upper - lower) * (t_rand != nullptr ? t_rand[idx] : philox_uniform(rng, 0u, (unsigned long long)idx));
    }
    t_vals[idx] = t;
}

// intern/parameterization.py:5-8 as called from model.py:196
__global__ void t_to_s_kernel(const float *__restrict__ t_vals, const float *__restrict__ near,
                              const float *__restrict__ far, int B, int M, int near_calls,
                              int far_calls, float *__restrict__ s_vals) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * M) return;
    const int b = (int)(idx / M);
    const float gt = 1.0f / (t_vals[idx] + kEpsG);
    const float gn1 = g_calls(near[b], near_calls + 1);
    const float gf = g_calls(far[b], far_calls + 1);
    const float gn2 = g_calls(near[b], near_calls + 2);
    s_vals[idx] = (gt - gn1) / (gf - gn2);
}

__global__ void g_kernel(const float *__restrict__ x, long n, float *__restrict__ y) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) y[idx] = 1.0f / (x[idx] + kEpsG);
}

// intern/parameterization.py:10-13
__global__ void s_to_t_kernel(const float *__restrict__ s_vals, const float *__restrict__ near,
                              const float *__restrict__ far, int B, int M, float *__restrict__ t_vals) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * M) return;
    const int b = (int)(idx / M);
    const float s = s_vals[idx];
    const float gf = 1.0f / (far[b] + kEpsG), gn = 1.0f / (near[b] + kEpsG);
    const float mix = s * gf + (1.0f - s) * gn;
    t_vals[idx] = 1.0f / (mix + kEpsG);
}

__global__ void contract_vec_kernel(const float *__restrict__ x, long n, const NormScratch *__restrict__ ws,
                                    float *__restrict__ y) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float gn = ws->gnorm;
    y[idx] = gn <= 1.0f ? x[idx] : (2.0f - 1.0f / gn) * (x[idx] / gn);
}

__global__ void frustum_moments_kernel(const float *__restrict__ t0, const float *__restrict__ t1,
                                       const float *__restrict__ radii, int B, int N,
                                       float *__restrict__ t_mean, float *__restrict__ t_var,
                                       float *__restrict__ r_var) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * N) return;
    const int b = (int)(idx / N);
    float tm, tv, rv;
    frustum_moments(t0[idx], t1[idx], radii[b], tm, tv, rv);
    t_mean[idx] = tm;
    t_var[idx] = tv;
    r_var[idx] = rv;
}

// intern/parameterization.py:108-113 (stable=False): the direct moment formulas (catastrophic cancellation for thin
// frusta - the reference's own docstring warns - but a public branch all the same)
__global__ void frustum_moments_unstable_kernel(const float *__restrict__ t0, const float *__restrict__ t1,
                                                const float *__restrict__ radii, int B, int N,
                                                float *__restrict__ t_mean, float *__restrict__ t_var,
                                                float *__restrict__ r_var) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * N) return;
    const int b = (int)(idx / N);
    const float a = t0[idx], c = t1[idx], r = radii[b];
    const float a2 = a * a, c2 = c * c;
    const float a3 = a2 * a, c3 = c2 * c, a4 = a2 * a2, c4 = c2 * c2, a5 = a4 * a, c5 = c4 * c;
    const float d3 = c3 - a3, d5 = c5 - a5;
    const float tm = (3.0f * (c4 - a4)) / (4.0f * d3);
    const float rv = (r * r) * ((3.0f / 20.0f) * d5 / d3);
    const float mosq = (3.0f / 5.0f) * d5 / d3;
    t_mean[idx] = tm;
    r_var[idx] = rv;
    t_var[idx] = mosq - tm * tm;
}

// intern/parameterization.py:48-54 (diag=True): mean = d t_mean, diagonal of the covariance only
__global__ void gaussian_to_xyz_diag_kernel(const float *__restrict__ d, const float *__restrict__ t_mean,
                                            const float *__restrict__ t_var, const float *__restrict__ r_var,
                                            int B, int N, float *__restrict__ mean, float *__restrict__ cov_diag) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * N) return;
    const int b = (int)(idx / N);
    const float dd[3] = {d[3 * b], d[3 * b + 1], d[3 * b + 2]};
    const float mag = fmaxf(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2], 1e-10f);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float outer = dd[i] * dd[i];
        mean[3 * idx + i] = dd[i] * t_mean[idx];
        cov_diag[3 * idx + i] = t_var[idx] * outer + r_var[idx] * (1.0f - outer / mag);
    }
}

__global__ void gaussian_to_xyz_kernel(const float *__restrict__ d, const float *__restrict__ t_mean,
                                       const float *__restrict__ t_var,
                                       const float *__restrict__ r_var, int B, int N,
                                       float *__restrict__ mean, float *__restrict__ cov) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * N) return;
    const int b = (int)(idx / N);
    const float dd[3] = {d[3 * b], d[3 * b + 1], d[3 * b + 2]};
    float m[3], c[9];
    lift_to_xyz(dd, t_mean[idx], t_var[idx], r_var[idx], m, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) mean[3 * idx + i] = m[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) cov[9 * idx + i] = c[i];
}

// ------------------------------------------------------------------------------------------
// Whole-tensor Frobenius norm, two deterministic passes (fixed partition, fp64 partials).
__device__ __forceinline__ void block_store_partial(double v, double *__restrict__ partial) {
    __shared__ double red[8];
    v = wave_sum_d(v);
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        partial[blockIdx.x] = s;
    }
}

// sum of squares of mean = d * t_mean straight from t_vals (no means materialised)
__global__ __launch_bounds__(256) void norm_partial_from_t_kernel(
    const float *__restrict__ t_vals, const float *__restrict__ directions,
    const float *__restrict__ radii, int B, int N, NormScratch *__restrict__ ws) {
    const long S = (long)B * N;
    double acc = 0.0;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < S;
         idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / N), n = (int)(idx % N);
        const float t0 = t_vals[(long)b * (N + 1) + n], t1 = t_vals[(long)b * (N + 1) + n + 1];
        float tm, tv, rv;
        frustum_moments(t0, t1, radii[b], tm, tv, rv);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float m = directions[3 * b + i] * tm;
            acc += (double)m * (double)m;
        }
    }
    block_store_partial(acc, ws->partial);
}

// One workgroup per group of `group_rays` consecutive rays: the summation order depends only on the group's own
// samples (thread t takes local samples t, t + 256, ...; fp64), so a chunk's norm is bit-identical whether the chunk
// is launched alone or as one of many groups of a larger launch.  Used for every group of <= kSmallGroup samples.
__global__ __launch_bounds__(256) void norm_group_from_t_kernel(
    const float *__restrict__ t_vals, const float *__restrict__ directions,
    const float *__restrict__ radii, int B, int N, int group_rays, NormScratch *__restrict__ ws) {
    __shared__ double red[4];
    const int r0 = blockIdx.x * group_rays;
    const int rays = (B - r0 < group_rays) ? (B - r0) : group_rays;
    const long count = (long)rays * N;
    double acc = 0.0;
    for (long j = threadIdx.x; j < count; j += blockDim.x) {
        const int b = r0 + (int)(j / N), n = (int)(j % N);
        const float t0 = t_vals[(long)b * (N + 1) + n], t1 = t_vals[(long)b * (N + 1) + n + 1];
        float tm, tv, rv;
        frustum_moments(t0, t1, radii[b], tm, tv, rv);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float m = directions[3 * b + i] * tm;
            acc += (double)m * (double)m;
        }
    }
    acc = wave_sum_d(acc);
    if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ss = red[0] + red[1] + red[2] + red[3];
        const float g = (float)sqrt(ss);
        ws->gnorms[blockIdx.x] = g;
        if (gridDim.x == 1) {
            ws->gnorm = g;
            ws->sumsq = ss;
        }
    }
}

__global__ __launch_bounds__(256) void norm_partial_from_mean_kernel(const float *__restrict__ mean,
                                                                      long count,
                                                                      NormScratch *__restrict__ ws) {
    double acc = 0.0;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < count;
         idx += (long)gridDim.x * blockDim.x) {
        const float m = mean[idx];
        acc += (double)m * (double)m;
    }
    block_store_partial(acc, ws->partial);
}

__global__ __launch_bounds__(256) void norm_final_kernel(NormScratch *__restrict__ ws, int nparts) {
    __shared__ double red[4];
    double v = 0.0;
    for (int p = threadIdx.x; p < nparts; p += blockDim.x) v += ws->partial[p];
    v = wave_sum_d(v);
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        ws->sumsq = red[0] + red[1] + red[2] + red[3];
        ws->gnorm = (float)sqrt(ws->sumsq);
    }
}

// norm_final_kernel's reduction run by EVERY 256-thread workgroup of a consumer kernel (round 4: one launch less per stage):
// thread t adds partials t, t + 256, ...; wave_sum_d; the four wave sums are added 0 + 1 + 2 + 3 - the same order, hence the same
// bits in every workgroup and the same bits as the separate kernel.  The partials (<= 8 KB) sit in L2.  All 256 threads must call.
__device__ __forceinline__ float block_norm_from_partials(const NormScratch *__restrict__ ws, int nparts, double *sumsq_out = nullptr) {
    __shared__ double red_[4];
    double v = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) v += ws->partial[p];
    v = wave_sum_d(v);
    if (lane_id() == 0) red_[threadIdx.x >> 6] = v;
    __syncthreads();
    const double ss = red_[0] + red_[1] + red_[2] + red_[3];
    if (sumsq_out) *sumsq_out = ss;
    return (float)sqrt(ss);
}

// Proposal-stage prologue of the rendering forward in ONE launch (round 4; until round 3: sample_t, viewdir_enc, norm_partial,
// norm_final and a memset): on the grid of the norm's partial sums (`parts` workgroups of 256 threads, grid-stride)
//   * t_vals = the deterministic samples of intern/ray.py:99-101 (sample_t_kernel's arithmetic);
//   * the view-direction encoding of intern/encoding.py:69-90 (viewdir_enc_kernel's arithmetic);
//   * the partial sums of the whole-chunk contraction norm (parameterization.py:25) in norm_partial_from_t_kernel's partition
//     and order, with t0 / t1 RECOMPUTED from near / far - the same instructions as the stored values, so the same bits - instead
//     of read back; the final sum is taken by the encoder's workgroups (block_norm_from_partials);
//   * workgroup 0 clears the tile-queue words of both stages' balanced linear launches.
__global__ __launch_bounds__(256) void stage_prologue_kernel(
    const float *__restrict__ near, const float *__restrict__ far, const float *__restrict__ viewdirs,
    const float *__restrict__ directions, const float *__restrict__ radii, int B, int N, int min_deg, int L,
    float *__restrict__ t_vals, float *__restrict__ vdenc, unsigned *__restrict__ queue_words, int n_queue_words,
    NormScratch *__restrict__ ws, rng_t rng) {
    const int M = N + 1;
    const long stride = (long)gridDim.x * blockDim.x, first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    auto t_lin = [&](float gn, float gf, int j) {
        const float sl = linspacef_(0.0f, 1.0f, M, j);
        const float mix = sl * gf + (1.0f - sl) * gn;
        return 1.0f / (mix + kEpsG);
    };
    // randomized=True (intern/ray.py:103-108): sample j of ray b jittered inside its half-intervals by the Philox uniform of element
    // b M + j - a pure function of the element, so the norm's loop below redraws what the t loop drew (sample_t_kernel's arithmetic)
    auto t_at = [&](float gn, float gf, int j, long b = 0) {
        const float t = t_lin(gn, gf, j);
        if (!rng.on) return t;
        const float lower = (j == 0) ? t : 0.5f * (t + t_lin(gn, gf, j - 1));
        const float upper = (j == N) ? t : 0.5f * (t_lin(gn, gf, j + 1) + t);
        return lower + (upper - lower) * philox_uniform(rng, 0u, (unsigned long long)(b * M + j));
    };
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_queue_words; i += blockDim.x) queue_words[i] = 0u;
    for (long idx = first; idx < (long)B * M; idx += stride) {
        const int b = (int)(idx / M), i = (int)(idx % M);
        t_vals[idx] = t_at(1.0f / (near[b] + kEpsG), 1.0f / (far[b] + kEpsG), i, b);
    }
    if (L > 0)
        for (long b = first; b < B; b += stride) {
            const float x = viewdirs[3 * b], y = viewdirs[3 * b + 1], z = viewdirs[3 * b + 2];
            const float theta = acosf(z);
            const float phi = atanf(y / (x + 1e-6f));
            float *row = vdenc + b * 4 * L;
            for (int i = 0; i < L; ++i) {
                const float sc = ldexpf(1.0f, min_deg + i);
                float sn, cs;
                sincosf(sc * theta, &sn, &cs);
                row[i] = sn;
                row[L + i] = cs;
                sincosf(sc * phi, &sn, &cs);
                row[2 * L + i] = sn;
                row[3 * L + i] = cs;
            }
        }
    const long S = (long)B * N;
    double acc = 0.0;
    for (long idx = first; idx < S; idx += stride) {
        const int b = (int)(idx / N), n = (int)(idx % N);
        const float gn = 1.0f / (near[b] + kEpsG), gf = 1.0f / (far[b] + kEpsG);
        const float t0 = t_at(gn, gf, n, b), t1 = t_at(gn, gf, n + 1, b);
        float tm, tv, rv;
        frustum_moments(t0, t1, radii[b], tm, tv, rv);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float m = directions[3 * b + i] * tm;
            acc += (double)m * (double)m;
        }
    }
    block_store_partial(acc, ws->partial);
}

// intern/parameterization.py:64-83 on materialised tensors
__global__ void contract_apply_kernel(const float *__restrict__ mean_in,
                                      const float *__restrict__ cov_in, long S,
                                      const NormScratch *__restrict__ ws,
                                      float *__restrict__ mean_out, float *__restrict__ cov_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S) return;
    float m[3], c[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) m[i] = mean_in[3 * idx + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = cov_in[9 * idx + i];
    contract_mean(m, ws->gnorm);
    contract_cov(m, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) mean_out[3 * idx + i] = m[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) cov_out[9 * idx + i] = c[i];
}

// one sample: t interval + ray -> contracted gaussian (+ origin), intern/parameterization.py:119-135
__device__ __forceinline__ void sample_gaussian(const float *__restrict__ t_vals,
                                                const float *__restrict__ origins,
                                                const float *__restrict__ directions,
                                                const float *__restrict__ radii, int N, int b, int n,
                                                float gnorm, float mean[3], float cov[9]) {
    const float t0 = t_vals[(long)b * (N + 1) + n], t1 = t_vals[(long)b * (N + 1) + n + 1];
    float tm, tv, rv;
    frustum_moments(t0, t1, radii[b], tm, tv, rv);
    const float d[3] = {directions[3 * b], directions[3 * b + 1], directions[3 * b + 2]};
    lift_to_xyz(d, tm, tv, rv, mean, cov);
    contract_mean(mean, gnorm);
    contract_cov(mean, cov);
#pragma unroll
    for (int i = 0; i < 3; ++i) mean[i] = mean[i] + origins[3 * b + i];
}

__global__ void para_rays_kernel(const float *__restrict__ t_vals, const float *__restrict__ origins,
                                 const float *__restrict__ directions,
                                 const float *__restrict__ radii, int B, int N,
                                 const NormScratch *__restrict__ ws, float *__restrict__ means,
                                 float *__restrict__ covs) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * N) return;
    float m[3], c[9];
    sample_gaussian(t_vals, origins, directions, radii, N, (int)(idx / N), (int)(idx % N), ws->gnorm,
                    m, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) means[3 * idx + i] = m[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) covs[9 * idx + i] = c[i];
}

// ------------------------------------------------------------------------------------------
// intern/encoding.py:33-61
template <bool HAS_COV>
__global__ void ipe_kernel(const float *__restrict__ mean, const float *__restrict__ cov, long S,
                           float *__restrict__ enc) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S) return;
    float m[3], c[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) m[i] = mean[3 * idx + i];
    if (HAS_COV) {
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] = cov[9 * idx + i];
    }
    float *row = enc + idx * kIpeCh;
    ipe_sample<HAS_COV>(m, c, [&](int k, float v) { row[k] = v; });
}

// intern/encoding.py:69-90: theta = acos(z), phi = atan(y / (x + 1e-6)); [sin th, cos th, sin ph, cos ph]
__global__ void viewdir_enc_kernel(const float *__restrict__ viewdirs, int B, int min_deg, int L,
                                   float *__restrict__ enc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float x = viewdirs[3 * b], y = viewdirs[3 * b + 1], z = viewdirs[3 * b + 2];
    const float theta = acosf(z);
    const float phi = atanf(y / (x + 1e-6f));
    float *row = enc + (long)b * 4 * L;
    for (int i = 0; i < L; ++i) {
        const float sc = ldexpf(1.0f, min_deg + i);
        float s, c;
        sincosf(sc * theta, &s, &c);
        row[i] = s;
        row[L + i] = c;
        sincosf(sc * phi, &s, &c);
        row[2 * L + i] = s;
        row[3 * L + i] = c;
    }
}

// ------------------------------------------------------------------------------------------
// Fused para_rays + IPE + view-direction repeat + concat -> MLP input rows.
// One thread per sample; each block stages its 128 x ld_feat tile in LDS (row stride ld+1 =>
// conflict-free per-row writes) and streams it out as whole 16-byte-per-lane coalesced stores.
constexpr int kEncThreads = 128;

// intra-wave LDS ordering (wave-private tiles): a wave-scope fence suffices
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// BF16: 0 = fp32 rows, 1 = bf16 rows, 2 = rows [hi(ld) | lo(ld)], two bf16 terms per value (split_bf16_), 3 = "x6" rows
// [lo | mid | hi | mid | hi | hi] of ld columns each, three bf16 terms per value (split3_bf16_): the MLP input of the bf16 and
// bf16x3 modes since round 4 - their first layers multiply all 24 bits of every feature (m360_pack_linear_bf16x6)
template <int BF16>
__global__ __launch_bounds__(kEncThreads) void encode_features_kernel(
    const float *__restrict__ t_vals, const float *__restrict__ origins,
    const float *__restrict__ directions, const float *__restrict__ radii,
    const float *__restrict__ vdenc, int vd_ch, int B, int N, const NormScratch *__restrict__ ws,
    void *__restrict__ feat_out, int ld, int group_rays, const float *__restrict__ ext_norm, unsigned char *__restrict__ nanflag) {
    extern __shared__ float tile[];  // [kEncThreads][ld + 1]
    const long S = (long)B * N;
    const long s0 = (long)blockIdx.x * kEncThreads;
    const long idx = s0 + threadIdx.x;
    const int lds_ld = ld + 1;
    float *row = tile + threadIdx.x * lds_ld;
    if (idx < S) {
        const int b = (int)(idx / N), n = (int)(idx % N);
        float m[3], c[9];
        const float gn = ext_norm ? *ext_norm : (group_rays > 0 ? ws->gnorms[b / group_rays] : ws->gnorm);
        sample_gaussian(t_vals, origins, directions, radii, N, b, n, gn, m, c);
        // NaN features are written as +NaN (canon_nanf_): the MLP's ReLU keeps exactly those
        int bad = 0;  // does this sample carry a NaN feature?  (see encode_features_wave_kernel)
        ipe_sample<true>(m, c, [&](int k, float v) { bad |= (v != v); row[k] = canon_nanf_(v); });
        for (int k = 0; k < vd_ch; ++k) {
            const float v = vdenc[(long)b * vd_ch + k];
            bad |= (v != v);
            row[kIpeCh + k] = canon_nanf_(v);
        }
        for (int k = kIpeCh + vd_ch; k < ld; ++k) row[k] = 0.0f;
        if (nanflag) nanflag[idx] = (unsigned char)bad;
    }
    __syncthreads();
    const long rows = (S - s0 < kEncThreads) ? (S - s0) : kEncThreads;
    if (!BF16) {
        const long total4 = rows * ld / 4;  // ld is a multiple of 32
        float4 *out = reinterpret_cast<float4 *>(static_cast<float *>(feat_out) + s0 * ld);
        for (long q = threadIdx.x; q < total4; q += kEncThreads) {
            const int r = (int)((q * 4) / ld), col = (int)((q * 4) % ld);
            const float *src = tile + r * lds_ld + col;
            out[q] = make_float4(src[0], src[1], src[2], src[3]);
        }
    } else {
        const long total8 = rows * ld / 8;
        const int ldo = BF16 == 3 ? 6 * ld : (BF16 == 2 ? 2 * ld : ld);
        __bf16 *outb = static_cast<__bf16 *>(feat_out) + s0 * ldo;
        for (long q = threadIdx.x; q < total8; q += kEncThreads) {
            const int r = (int)((q * 8) / ld), col = (int)((q * 8) % ld);
            const float *src = tile + r * lds_ld + col;
            bf16x8_t o, lo, mid;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (BF16 == 3) {
                    __bf16 h_, m_, l_;
                    split3_bf16_(src[e], h_, m_, l_);
                    o[e] = h_, mid[e] = m_, lo[e] = l_;
                } else {
                    o[e] = (__bf16)src[e];
                    if (BF16 == 2) lo[e] = bf16_lo_(src[e], o[e]);
                }
            }
            __bf16 *dst = outb + (long)r * ldo + col;
            if (BF16 == 3) {
                *reinterpret_cast<bf16x8_t *>(dst) = lo;
                *reinterpret_cast<bf16x8_t *>(dst + ld) = mid;
                *reinterpret_cast<bf16x8_t *>(dst + 2 * ld) = o;
                *reinterpret_cast<bf16x8_t *>(dst + 3 * ld) = mid;
                *reinterpret_cast<bf16x8_t *>(dst + 4 * ld) = o;
                *reinterpret_cast<bf16x8_t *>(dst + 5 * ld) = o;
            } else {
                *reinterpret_cast<bf16x8_t *>(dst) = o;
                if (BF16 == 2) *reinterpret_cast<bf16x8_t *>(dst + ld) = lo;
            }
        }
    }
}


// Wave-tiled form of the kernel above for ld = 64 / 96 (every model of the path): one sample per lane, the 42 IPE values
// stay in registers and go out in 32-channel passes through a wave-private [64][36]-float LDS tile (9 KiB per wave instead
// of 33 KiB per 128 samples, no workgroup barrier): 4 x the resident waves per CU of the kernel above, whose 2.6 TB/s were
// occupancy-bound once the short sin / cos had removed the ALU bound.  Each pass writes whole 128-byte (fp32) or 64-byte
// (bf16) row segments with 16-byte lanes.  Same values as the kernel above, bit for bit.
// (4 workgroups per CU = 128 VGPRs with 10-12 of them spilled: measured against 3 per CU = 139 VGPRs, no scratch, in round 6 - 39.6 against 43.3 us
// in fp32, 47.2 against 51.5 us in bf16, alternating builds on one box: the occupancy is worth more than the spills cost.)
constexpr int kEncWaves = 4;
constexpr int kEncTileLd = 36;

template <int BF16, int NPASS>
__global__ __launch_bounds__(kEncWaves *kWave, 4) void encode_features_wave_kernel(
    const float *__restrict__ t_vals, const float *__restrict__ origins,
    const float *__restrict__ directions, const float *__restrict__ radii,
    const float *__restrict__ vdenc, int vd_ch, int B, int N, const NormScratch *__restrict__ ws,
    void *__restrict__ feat_out, int group_rays, const float *__restrict__ ext_norm, int norm_parts,
    unsigned char *__restrict__ nanflag) {
    // nanflag (bf16 / bf16x3 stage drivers): one byte per sample, 1 = some feature of the sample is NaN.  torch.relu(NaN) is NaN, so in
    // the reference such a sample stays NaN through every layer (model.py:43-53,131-148); the fp32 kernels keep it too (integer-max ReLU
    // on +NaN).  The bf16 matrix pipe answers ANY NaN operand with the default NaN 0xFFC00000 - sign bit set (tools/diag/nan_bits_bf16.py) -
    // which the packed integer-max ReLU reads as a negative number: no NaN survives a ReLU layer there.  The finishers therefore poison
    // the head outputs of flagged samples, which is where the reference's NaN would have arrived.
    __shared__ __attribute__((aligned(16))) float tiles[kEncWaves][kWave * kEncTileLd];
    constexpr int ld = 32 * NPASS;
    const long S = (long)B * N;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    // norm_parts > 0: the whole-chunk norm is still `norm_parts` partial sums - every workgroup takes the final sum itself, in
    // norm_final_kernel's order (before any wave leaves: the reduction has a workgroup barrier)
    float gn_block = 0.0f;
    if (norm_parts > 0) {
        double ss;
        gn_block = block_norm_from_partials(ws, norm_parts, &ss);
        if (blockIdx.x == 0 && threadIdx.x == 0) {  // kept for readers of the scratch (tests, m360_mean_sumsq's layout)
            NormScratch *wsw = const_cast<NormScratch *>(ws);
            wsw->sumsq = ss;
            wsw->gnorm = gn_block;
        }
    }
    const long s0 = ((long)blockIdx.x * kEncWaves + wave) * kWave;  // first sample of this wave
    if (s0 >= S) return;
    const long idx = s0 + lane;
    const bool live = idx < S;
    float *tile = tiles[wave];
    float v[kIpeCh];
    int b = 0;
    if (live) {
        b = (int)(idx / N);
        const int n = (int)(idx % N);
        float m[3], c[9];
        const float gn = norm_parts > 0 ? gn_block : (ext_norm ? *ext_norm : (group_rays > 0 ? ws->gnorms[b / group_rays] : ws->gnorm));
        sample_gaussian(t_vals, origins, directions, radii, N, b, n, gn, m, c);
        ipe_sample<true, true>(m, c, [&](int k, float val) { v[k] = val; });
    } else {
#pragma unroll
        for (int k = 0; k < kIpeCh; ++k) v[k] = 0.0f;
    }
    const long rows = (S - s0 < kWave) ? (S - s0) : kWave;
    int bad = 0;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) {
            const int ch = 32 * p + cc;  // compile-time
            float val;
            if (ch < kIpeCh) val = v[ch];
            else val = (live && ch - kIpeCh < vd_ch) ? vdenc[(long)b * vd_ch + (ch - kIpeCh)] : 0.0f;
            if (BF16) bad |= (val != val);
            tile[lane * kEncTileLd + cc] = canon_nanf_(val);  // NaN features go out as +NaN: the MLP's ReLU keeps exactly those
        }
        wave_sync_lds();
        if (!BF16) {  // 8 lanes x 16 B = one 128-byte row segment; 8 rows per instruction
            float *out = static_cast<float *>(feat_out) + s0 * ld + 32 * p;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 8 * i + (lane >> 3), col = 4 * (lane & 7);
                if (r < rows) *reinterpret_cast<float4 *>(out + (long)r * ld + col) = *reinterpret_cast<const float4 *>(tile + r * kEncTileLd + col);
            }
        } else {      // 4 lanes x 16 B (8 bf16) = one 64-byte row segment; 16 rows per instruction
            constexpr int ldo = BF16 == 3 ? 6 * ld : (BF16 == 2 ? 2 * ld : ld);  // [hi | lo] pairs / x6 rows
            __bf16 *out = static_cast<__bf16 *>(feat_out) + s0 * ldo + 32 * p;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * i + (lane >> 2), col = 8 * (lane & 3);
                const float4 lo = *reinterpret_cast<const float4 *>(tile + r * kEncTileLd + col);
                const float4 hi = *reinterpret_cast<const float4 *>(tile + r * kEncTileLd + col + 4);
                const float e8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                bf16x8_t o, l2, m2;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (BF16 == 3) {
                        __bf16 h_, m_, l_;
                        split3_bf16_(e8[e], h_, m_, l_);
                        o[e] = h_, m2[e] = m_, l2[e] = l_;
                    } else {
                        o[e] = (__bf16)e8[e];
                        if (BF16 == 2) l2[e] = bf16_lo_(e8[e], o[e]);
                    }
                }
                if (r < rows) {
                    __bf16 *dst = out + (long)r * ldo + col;
                    if (BF16 == 3) {  // [lo | mid | hi | mid | hi | hi]: the order the first layer accumulates its six products in
                        *reinterpret_cast<bf16x8_t *>(dst) = l2;
                        *reinterpret_cast<bf16x8_t *>(dst + ld) = m2;
                        *reinterpret_cast<bf16x8_t *>(dst + 2 * ld) = o;
                        *reinterpret_cast<bf16x8_t *>(dst + 3 * ld) = m2;
                        *reinterpret_cast<bf16x8_t *>(dst + 4 * ld) = o;
                        *reinterpret_cast<bf16x8_t *>(dst + 5 * ld) = o;
                    } else {
                        *reinterpret_cast<bf16x8_t *>(dst) = o;
                        if (BF16 == 2) *reinterpret_cast<bf16x8_t *>(dst + ld) = l2;
                    }
                }
            }
        }
        wave_sync_lds();  // the tile is rewritten by the next pass
    }
    if (BF16 && nanflag != nullptr && live) nanflag[idx] = (unsigned char)bad;
}

}  // namespace m360

// =========================================================================================
using namespace m360;

static inline hipStream_t S_(m360_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned blocks_for(long n, int threads) { return (unsigned)((n + threads - 1) / threads); }

extern "C" {  // (defined inside the extern "C" block below)
static int norm_parts(long work);
static int encode_features_any(const float *t_vals, const float *origins, const float *directions,
                               const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat,
                               int ld_feat, int bf16, void *workspace, size_t workspace_bytes, m360_stream_t stream,
                               int group_rays, const float *ext_norm, int prepared_parts, unsigned char *nanflag);
}

namespace m360 {
// Stage drivers only (m360_capi.hip; not part of the C-ABI).  stage_prologue: one launch for t_vals, the view-direction encoding,
// the norm's partial sums and the tile-queue words; returns the number of partial sums (> 0) to hand to encode_prepared, 0 when
// the chunk is not of the shape the fused prologue takes (the caller then runs the separate entry points), < 0 on a launch error.
int stage_prologue(const m360_rays_t *r, int B, int N, int min_deg, int max_deg, float *t_vals, float *vdenc,
                   unsigned *queue_words, int n_queue_words, int ld_feat, void *norm_ws, m360_stream_t stream, const rng_t &rng) {
    if ((long)B * N <= kSmallGroup || !(ld_feat == 64 || ld_feat == 96)) return 0;
    const int parts = norm_parts((long)B * N);
    hipLaunchKernelGGL(stage_prologue_kernel, dim3(parts), dim3(256), 0, S_(stream), r->near, r->far, r->viewdirs, r->directions, r->radii,
                       B, N, min_deg, max_deg - min_deg, t_vals, vdenc, queue_words, n_queue_words, static_cast<NormScratch *>(norm_ws), rng);
    return check_launch("stage_prologue") == M360_OK ? parts : -1;
}
// m360_sample_t with the jitter's uniforms from a tensor (t_rand), from the Philox stream (rng.on) or not at all
int sample_t_any(const float *near, const float *far, const float *t_rand, int B, int N, float *t_vals, const rng_t &rng, m360_stream_t stream) {
    if (!near || !far || !t_vals || B < 0 || N < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_sample_t: bad argument (B=%d N=%d)", B, N);
    if (B == 0) return M360_OK;
    const long n = (long)B * (N + 1);
    hipLaunchKernelGGL(sample_t_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, S_(stream), near, far, t_rand, B, N, t_vals, rng);
    return check_launch("sample_t");
}
__global__ void philox_uniform_kernel(rng_t rng, unsigned stream_id, long n, float *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = philox_uniform(rng, stream_id, (unsigned long long)idx);
}
// the encode step of a stage: per-chunk norms (group_rays > 0), a norm given by the caller (ext_norm), partial sums left by
// stage_prologue (prepared_parts > 0), or computed here; nanflag: one byte per sample for the bf16 modes' finishers (or NULL)
int encode_stage(const float *t_vals, const float *origins, const float *directions, const float *radii, const float *vdenc,
                 int vd_ch, int B, int N, void *feat, int ld_feat, int row_format, int group_rays, const float *ext_norm,
                 int prepared_parts, unsigned char *nanflag, void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    if (group_rays < 0) return fail(M360_ERR_INVALID_ARGUMENT, "encode_stage: group_rays=%d", group_rays);
    if (group_rays > 0 && ((long)group_rays * N > kSmallGroup || (B + group_rays - 1) / group_rays > kMaxNormGroups))
        return fail(M360_ERR_INVALID_ARGUMENT, "encode_stage: group_rays=%d x N=%d exceeds %ld samples per group, or more than %d groups", group_rays, N, kSmallGroup, kMaxNormGroups);
    return encode_features_any(t_vals, origins, directions, radii, vdenc, vd_ch, B, N, feat, ld_feat, row_format, workspace, workspace_bytes, stream, group_rays, ext_norm, prepared_parts, nanflag);
}
}  // namespace m360

extern "C" {

int m360_sample_t(const float *near, const float *far, const float *t_rand, int B, int N,
                  float *t_vals, m360_stream_t stream) {
    return sample_t_any(near, far, t_rand, B, N, t_vals, rng_t{0, 0, 0}, stream);
}

int m360_sample_t_philox(const float *near, const float *far, int B, int N, unsigned long long seed, unsigned long long offset,
                         float *t_vals, m360_stream_t stream) {
    return sample_t_any(near, far, nullptr, B, N, t_vals, rng_t{seed, offset, 1}, stream);
}

int m360_philox_uniform(unsigned long long seed, unsigned long long offset, int stream_id, long n, float *out, m360_stream_t stream) {
    if (!out || n < 0 || stream_id < 0 || stream_id > 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_philox_uniform: bad argument (n=%ld stream_id=%d)", n, stream_id);
    if (n == 0) return M360_OK;
    hipLaunchKernelGGL(philox_uniform_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, S_(stream), rng_t{seed, offset, 1}, (unsigned)stream_id, n, out);
    return check_launch("philox_uniform");
}

int m360_t_to_s(const float *t_vals, const float *near, const float *far, int B, int M,
                int near_calls, int far_calls, float *s_vals, m360_stream_t stream) {
    if (!t_vals || !near || !far || !s_vals || B < 0 || M < 1 || near_calls < 0 || far_calls < 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_t_to_s: bad argument");
    if (B == 0) return M360_OK;
    hipLaunchKernelGGL(t_to_s_kernel, dim3(blocks_for((long)B * M, 256)), dim3(256), 0, S_(stream), t_vals, near, far, B, M, near_calls, far_calls, s_vals);
    return check_launch("t_to_s");
}

int m360_g(const float *x, long n, float *y, m360_stream_t stream) {
    if (!x || !y || n < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_g: bad argument");
    if (n == 0) return M360_OK;
    hipLaunchKernelGGL(g_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, S_(stream), x, n, y);
    return check_launch("g");
}

int m360_s_to_t(const float *s_vals, const float *near, const float *far, int B, int M, float *t_vals,
                m360_stream_t stream) {
    if (!s_vals || !near || !far || !t_vals || B < 0 || M < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_s_to_t: bad argument");
    if (B == 0) return M360_OK;
    hipLaunchKernelGGL(s_to_t_kernel, dim3(blocks_for((long)B * M, 256)), dim3(256), 0, S_(stream), s_vals, near, far, B, M, t_vals);
    return check_launch("s_to_t");
}

int m360_frustum_moments(const float *t0, const float *t1, const float *radii, int B, int N,
                         float *t_mean, float *t_var, float *r_var, m360_stream_t stream) {
    if (!t0 || !t1 || !radii || !t_mean || !t_var || !r_var || B < 0 || N < 1)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_frustum_moments: bad argument");
    if (B == 0) return M360_OK;
    hipLaunchKernelGGL(frustum_moments_kernel, dim3(blocks_for((long)B * N, 256)), dim3(256), 0, S_(stream), t0, t1, radii, B, N, t_mean, t_var, r_var);
    return check_launch("frustum_moments");
}

int m360_frustum_moments_unstable(const float *t0, const float *t1, const float *radii, int B, int N,
                                  float *t_mean, float *t_var, float *r_var, m360_stream_t stream) {
    if (!t0 || !t1 || !radii || !t_mean || !t_var || !r_var || B < 0 || N < 1)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_frustum_moments_unstable: bad argument");
    if (B == 0) return M360_OK;
    hipLaunchKernelGGL(frustum_moments_unstable_kernel, dim3(blocks_for((long)B * N, 256)), dim3(256), 0, S_(stream), t0, t1, radii, B, N, t_mean, t_var, r_var);
    return check_launch("frustum_moments_unstable");
}

int m360_gaussian_to_xyz_diag(const float *d, const float *t_mean, const float *t_var, const float *r_var,
                              int B, int N, float *mean, float *cov_diag, m360_stream_t stream) {
    if (!d || !t_mean || !t_var || !r_var || !mean || !cov_diag || B < 0 || N < 1)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_gaussian_to_xyz_diag: bad argument");
    if (B == 0) return M360_OK;
    hipLaunchKernelGGL(gaussian_to_xyz_diag_kernel, dim3(blocks_for((long)B * N, 256)), dim3(256), 0, S_(stream), d, t_mean, t_var, r_var, B, N, mean, cov_diag);
    return check_launch("gaussian_to_xyz_diag");
}

int m360_gaussian_to_xyz(const float *d, const float *t_mean, const float *t_var, const float *r_var,
                         int B, int N, float *mean, float *cov, m360_stream_t stream) {
    if (!d || !t_mean || !t_var || !r_var || !mean || !cov || B < 0 || N < 1)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_gaussian_to_xyz: bad argument");
    if (B == 0) return M360_OK;
    hipLaunchKernelGGL(gaussian_to_xyz_kernel, dim3(blocks_for((long)B * N, 256)), dim3(256), 0, S_(stream), d, t_mean, t_var, r_var, B, N, mean, cov);
    return check_launch("gaussian_to_xyz");
}

size_t m360_contract_workspace_bytes(void) { return sizeof(NormScratch); }

static int norm_parts(long work) {
    long p = (work + 255) / 256;
    return (int)(p < 1 ? 1 : (p > kNormPartials ? kNormPartials : p));
}

int m360_contract(const float *x, long n, float *y, void *workspace, size_t workspace_bytes,
                  m360_stream_t stream) {
    if (!x || !y || n < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_contract: bad argument");
    if (!workspace || workspace_bytes < sizeof(NormScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_contract: workspace %zu < %zu", workspace_bytes, sizeof(NormScratch));
    if (n == 0) return M360_OK;
    NormScratch *ws = static_cast<NormScratch *>(workspace);
    const int parts = norm_parts(n);
    hipLaunchKernelGGL(norm_partial_from_mean_kernel, dim3(parts), dim3(256), 0, S_(stream), x, n, ws);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, S_(stream), ws, parts);
    hipLaunchKernelGGL(contract_vec_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, S_(stream), x, n, ws, y);
    return check_launch("contract");
}

int m360_gaussian_contract(const float *mean_in, const float *cov_in, long S, float *mean_out,
                           float *cov_out, void *workspace, size_t workspace_bytes,
                           m360_stream_t stream) {
    if (!mean_in || !cov_in || !mean_out || !cov_out || S < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_gaussian_contract: bad argument");
    if (!workspace || workspace_bytes < sizeof(NormScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_gaussian_contract: workspace %zu < %zu", workspace_bytes, sizeof(NormScratch));
    if (S == 0) return M360_OK;
    NormScratch *ws = static_cast<NormScratch *>(workspace);
    const int parts = norm_parts(3 * S);
    hipLaunchKernelGGL(norm_partial_from_mean_kernel, dim3(parts), dim3(256), 0, S_(stream), mean_in, 3 * S, ws);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, S_(stream), ws, parts);
    hipLaunchKernelGGL(contract_apply_kernel, dim3(blocks_for(S, 256)), dim3(256), 0, S_(stream), mean_in, cov_in, S, ws, mean_out, cov_out);
    return check_launch("gaussian_contract");
}

// norm pre-pass shared by para_rays / encode_features
// defer_final: leave the last step (norm_final_kernel) to the consumer, which then gets the number of partial sums (the return
// value; 0 = the norm(s) are final in ws, nothing deferred)
static int launch_norm_from_t(const float *t_vals, const float *directions, const float *radii, int B,
                              int N, NormScratch *ws, hipStream_t st, int group_rays = 0, bool defer_final = false) {
    if (group_rays > 0 || (long)B * N <= kSmallGroup) {  // per-chunk norms, or one small chunk: same kernel, same order
        const int gr = group_rays > 0 ? group_rays : B;
        hipLaunchKernelGGL(norm_group_from_t_kernel, dim3((B + gr - 1) / gr), dim3(256), 0, st, t_vals, directions, radii, B, N, gr, ws);
        return 0;
    }
    const int parts = norm_parts((long)B * N);
    hipLaunchKernelGGL(norm_partial_from_t_kernel, dim3(parts), dim3(256), 0, st, t_vals, directions, radii, B, N, ws);
    if (defer_final) return parts;
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, ws, parts);
    return 0;
}

int m360_para_rays(const float *t_vals, const float *origins, const float *directions,
                   const float *radii, int B, int N, float *means, float *covs, void *workspace,
                   size_t workspace_bytes, m360_stream_t stream) {
    if (!t_vals || !origins || !directions || !radii || !means || !covs || B < 0 || N < 1)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_para_rays: bad argument");
    if (!workspace || workspace_bytes < sizeof(NormScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_para_rays: workspace %zu < %zu", workspace_bytes, sizeof(NormScratch));
    if (B == 0) return M360_OK;
    NormScratch *ws = static_cast<NormScratch *>(workspace);
    launch_norm_from_t(t_vals, directions, radii, B, N, ws, S_(stream));
    hipLaunchKernelGGL(para_rays_kernel, dim3(blocks_for((long)B * N, 256)), dim3(256), 0, S_(stream), t_vals, origins, directions, radii, B, N, ws, means, covs);
    return check_launch("para_rays");
}

int m360_ipe(const float *mean, const float *cov, long S, float *enc, m360_stream_t stream) {
    if (!mean || !enc || S < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_ipe: bad argument");
    if (S == 0) return M360_OK;
    if (cov) hipLaunchKernelGGL(ipe_kernel<true>, dim3(blocks_for(S, 256)), dim3(256), 0, S_(stream), mean, cov, S, enc);
    else hipLaunchKernelGGL(ipe_kernel<false>, dim3(blocks_for(S, 256)), dim3(256), 0, S_(stream), mean, cov, S, enc);
    return check_launch("ipe");
}

int m360_viewdir_enc(const float *viewdirs, int B, int min_deg, int max_deg, float *enc,
                     m360_stream_t stream) {
    if (!viewdirs || !enc || B < 0 || max_deg < min_deg) return fail(M360_ERR_INVALID_ARGUMENT, "m360_viewdir_enc: bad argument");
    if (B == 0 || max_deg == min_deg) return M360_OK;
    hipLaunchKernelGGL(viewdir_enc_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, S_(stream), viewdirs, B, min_deg, max_deg - min_deg, enc);
    return check_launch("viewdir_enc");
}

static int encode_features_any(const float *t_vals, const float *origins, const float *directions,
                               const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat,
                               int ld_feat, int bf16, void *workspace, size_t workspace_bytes, m360_stream_t stream,
                               int group_rays = 0, const float *ext_norm = nullptr, int prepared_parts = 0,
                               unsigned char *nanflag = nullptr);  // (defaults: this declaration)

// one logical batch sharded over several devices: this shard's sum of squares of the un-contracted
// means, reduced exactly like the norm the encode stage would compute for these rays alone
int m360_mean_sumsq(const float *t_vals, const float *directions, const float *radii, int B, int N, double *sumsq,
                    void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    if (!t_vals || !directions || !radii || !sumsq || B < 0 || N < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_mean_sumsq: bad argument");
    if (!workspace || workspace_bytes < sizeof(NormScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_mean_sumsq: workspace %zu < %zu", workspace_bytes, sizeof(NormScratch));
    NormScratch *ws = static_cast<NormScratch *>(workspace);
    if (B == 0) {
        if (hipMemsetAsync(sumsq, 0, sizeof(double), S_(stream)) != hipSuccess) return fail(M360_ERR_LAUNCH, "m360_mean_sumsq: memset failed");
        return M360_OK;
    }
    launch_norm_from_t(t_vals, directions, radii, B, N, ws, S_(stream));
    if (hipMemcpyAsync(sumsq, &ws->sumsq, sizeof(double), hipMemcpyDeviceToDevice, S_(stream)) != hipSuccess) return fail(M360_ERR_LAUNCH, "m360_mean_sumsq: copy failed");
    return check_launch("mean_sumsq");
}

// m360_encode_features[_bf16] with the contraction norm supplied by the caller (device float)
int m360_encode_features_ext_norm(const float *t_vals, const float *origins, const float *directions,
                                  const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat,
                                  int ld_feat, int bf16, const float *norm, void *workspace, size_t workspace_bytes,
                                  m360_stream_t stream) {
    if (!norm) return fail(M360_ERR_INVALID_ARGUMENT, "m360_encode_features_ext_norm: norm is required");
    return encode_features_any(t_vals, origins, directions, radii, vdenc, vd_ch, B, N, feat, ld_feat, bf16, workspace, workspace_bytes, stream, 0, norm);
}

int m360_encode_features_grouped(const float *t_vals, const float *origins, const float *directions,
                                 const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat,
                                 int ld_feat, int bf16, int group_rays, void *workspace, size_t workspace_bytes,
                                 m360_stream_t stream) {
    if (group_rays < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_encode_features_grouped: group_rays=%d", group_rays);
    if (group_rays > 0 && ((long)group_rays * N > kSmallGroup || (B + group_rays - 1) / group_rays > kMaxNormGroups))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_encode_features_grouped: group_rays=%d x N=%d exceeds %ld samples per group, or more than %d groups", group_rays, N, kSmallGroup, kMaxNormGroups);
    return encode_features_any(t_vals, origins, directions, radii, vdenc, vd_ch, B, N, feat, ld_feat, bf16, workspace, workspace_bytes, stream, group_rays);
}

int m360_encode_features(const float *t_vals, const float *origins, const float *directions,
                         const float *radii, const float *vdenc, int vd_ch, int B, int N, float *feat,
                         int ld_feat, void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    return encode_features_any(t_vals, origins, directions, radii, vdenc, vd_ch, B, N, feat, ld_feat, 0, workspace, workspace_bytes, stream);
}

int m360_encode_features_bf16(const float *t_vals, const float *origins, const float *directions,
                              const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat_bf16,
                              int ld_feat, void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    return encode_features_any(t_vals, origins, directions, radii, vdenc, vd_ch, B, N, feat_bf16, ld_feat, 1, workspace, workspace_bytes, stream);
}

static int encode_features_any(const float *t_vals, const float *origins, const float *directions,
                               const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat,
                               int ld_feat, int bf16, void *workspace, size_t workspace_bytes, m360_stream_t stream,
                               int group_rays, const float *ext_norm, int prepared_parts, unsigned char *nanflag) {
    if (!t_vals || !origins || !directions || !radii || !feat || B < 0 || N < 1 || vd_ch < 0 || (vd_ch > 0 && !vdenc))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_encode_features: bad argument");
    if (ld_feat % 32 != 0 || ld_feat < kIpeCh + vd_ch) return fail(M360_ERR_INVALID_ARGUMENT, "m360_encode_features: ld_feat=%d must be a multiple of 32 and >= %d", ld_feat, kIpeCh + vd_ch);
    if (bf16 < 0 || bf16 > 3) return fail(M360_ERR_INVALID_ARGUMENT, "m360_encode_features: row format %d (0 fp32, 1 bf16, 2 [hi | lo], 3 x6)", bf16);
    if (!workspace || workspace_bytes < sizeof(NormScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_encode_features: workspace %zu < %zu", workspace_bytes, sizeof(NormScratch));
    if (B == 0) return M360_OK;
    NormScratch *ws = static_cast<NormScratch *>(workspace);
    const bool wave_kernel = ld_feat == 64 || ld_feat == 96;  // every model of the path
    // the whole-chunk norm: given (ext_norm), already `prepared_parts` partial sums in ws (stage_prologue), or computed here - with
    // the final sum left to the wave kernel's workgroups when it is the large-chunk two-step reduction
    int parts = prepared_parts;
    if (!ext_norm && prepared_parts == 0) parts = launch_norm_from_t(t_vals, directions, radii, B, N, ws, S_(stream), group_rays, wave_kernel);
    if (wave_kernel) {
        const dim3 grid(blocks_for((long)B * N, kEncWaves * kWave)), block(kEncWaves * kWave);
#define M360_ENC(BF, NP) hipLaunchKernelGGL((encode_features_wave_kernel<BF, NP>), grid, block, 0, S_(stream), t_vals, origins, directions, radii, vdenc, vd_ch, B, N, ws, feat, group_rays, ext_norm, parts, nanflag)
        if (ld_feat == 64) { if (bf16 == 3) M360_ENC(3, 2); else if (bf16 == 2) M360_ENC(2, 2); else if (bf16) M360_ENC(1, 2); else M360_ENC(0, 2); }
        else { if (bf16 == 3) M360_ENC(3, 3); else if (bf16 == 2) M360_ENC(2, 3); else if (bf16) M360_ENC(1, 3); else M360_ENC(0, 3); }
#undef M360_ENC
        return check_launch("encode_features");
    }
    if (parts > 0) hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, S_(stream), ws, parts);  // prepared partials, generic kernel
    const size_t lds = (size_t)kEncThreads * (ld_feat + 1) * sizeof(float);
    if (bf16 == 3) hipLaunchKernelGGL(encode_features_kernel<3>, dim3(blocks_for((long)B * N, kEncThreads)), dim3(kEncThreads), lds, S_(stream), t_vals, origins, directions, radii, vdenc, vd_ch, B, N, ws, feat, ld_feat, group_rays, ext_norm, nanflag);
    else if (bf16 == 2) hipLaunchKernelGGL(encode_features_kernel<2>, dim3(blocks_for((long)B * N, kEncThreads)), dim3(kEncThreads), lds, S_(stream), t_vals, origins, directions, radii, vdenc, vd_ch, B, N, ws, feat, ld_feat, group_rays, ext_norm, nanflag);
    else if (bf16) hipLaunchKernelGGL(encode_features_kernel<1>, dim3(blocks_for((long)B * N, kEncThreads)), dim3(kEncThreads), lds, S_(stream), t_vals, origins, directions, radii, vdenc, vd_ch, B, N, ws, feat, ld_feat, group_rays, ext_norm, nanflag);
    else hipLaunchKernelGGL(encode_features_kernel<0>, dim3(blocks_for((long)B * N, kEncThreads)), dim3(kEncThreads), lds, S_(stream), t_vals, origins, directions, radii, vdenc, vd_ch, B, N, ws, feat, ld_feat, group_rays, ext_norm, nanflag);
    return check_launch("encode_features");
}

}  // extern "C"
