// Training-side loss kernels (the "two loss loops"): forward values and the analytic
// gradients with respect to the losses' direct inputs.  One wavefront per ray, everything in that wave's LDS
// slice; per-ray partial losses are written out and summed by a deterministic fp64 reduction.
//   * proposal / envelope loss:   intern/distillation.py:4-51, intern/loss.py:6-21
//   * distortion loss:            intern/regularization.py:3-19, intern/loss.py:42-54 (the O(N^2) Python loop)
//   * reconstruction (log-MSE):   intern/loss.py:23-40,57-59
#include "m360_common.hip.h"

namespace m360 {

__device__ __forceinline__ void lwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kLossWaves = 4;

// Step 1 of bounds(): per ray b and proposal interval i, sum_j w[b,j] [not (t0[b,j] > T1[b,i] or t1[b,j] < T0[b,i])].
__global__ __launch_bounds__(kLossWaves *kWave) void prop_overlap_kernel(
    const float *__restrict__ t, const float *__restrict__ w, const float *__restrict__ t_hat, int B, int Nf, int Np,
    float *__restrict__ per_ray) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int b = blockIdx.x * kLossWaves + wave;
    if (b >= B) return;
    float *tf = smem + wave * (2 * Nf + 1), *wf = tf + Nf + 1;
    for (int j = l; j <= Nf; j += kWave) tf[j] = t[(long)b * (Nf + 1) + j];
    for (int j = l; j < Nf; j += kWave) wf[j] = w[(long)b * Nf + j];
    lwave_sync();
    for (int i = l; i < Np; i += kWave) {
        const float L = t_hat[(long)b * (Np + 1) + i], R = t_hat[(long)b * (Np + 1) + i + 1];
        float bi = 0.0f;
        for (int j = 0; j < Nf; ++j)
            if (!((tf[j] > R) || (tf[j + 1] < L))) bi += wf[j];
        per_ray[(long)b * Np + i] = bi;
    }
}

// Step 2: the reference indexes fine_weights[..., mask] with a [B, Nf] mask (distillation.py:29), which flattens over
// the rays: every ray receives the BATCH TOTAL.  Column sums in fp64, fixed order, in two levels (round 6): workgroup (x, g) adds the rays
// b = g, g + kColGroups, ... of its 64 columns (16 row classes by its waves, then 0..15) into part[g][Np]; prop_colsum_final_kernel adds the
// kColGroups rows.  (One workgroup per 64 columns over all 4096 rays was 76 us: 1 MB through ONE CU's loads in flight.)
constexpr int kColWaves = 16, kColGroups = 32;
__global__ __launch_bounds__(kColWaves *kWave) void prop_colsum_kernel(const float *__restrict__ per_ray, int B, int Np,
                                                                       double *__restrict__ part) {
    __shared__ double red[kColWaves][kWave];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int i = blockIdx.x * kWave + l;
    const int stride = kColGroups * kColWaves;
    double acc = 0.0;
    if (i < Np) {  // 8 loads in flight per wave, added in ascending order
        for (int b = blockIdx.y + kColGroups * wave; b < B; b += 8 * stride) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = b + e * stride < B ? per_ray[(long)(b + e * stride) * Np + i] : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (b + e * stride < B) acc += (double)v[e];
        }
    }
    red[wave][l] = acc;
    __syncthreads();
    if (wave == 0 && i < Np) {
        double s = 0.0;
        for (int k = 0; k < kColWaves; ++k) s += red[k][l];
        part[(long)blockIdx.y * Np + i] = s;
    }
}
__global__ __launch_bounds__(kWave) void prop_colsum_final_kernel(const double *__restrict__ part, int Np, float *__restrict__ total) {
    const int i = blockIdx.x * kWave + threadIdx.x;
    if (i >= Np) return;
    double v[kColGroups], s = 0.0;
#pragma unroll
    for (int g = 0; g < kColGroups; ++g) v[g] = part[(long)g * Np + i];
#pragma unroll
    for (int g = 0; g < kColGroups; ++g) s += v[g];
    total[i] = (float)s;
}

// Step 3 (distillation.py:35-51): ray loss = sum_i relu(bounds_i - what_i)^2 / (what_i + 1e-6); bounds row stride 0 =
// one shared vector (the batch total), Np = caller-supplied matrix.
__global__ __launch_bounds__(kLossWaves *kWave) void loss_prop_kernel(
    const float *__restrict__ bnd, int bnd_stride, const float *__restrict__ w_hat, int B, int Np, float inv_batch,
    float *__restrict__ bounds_out, float *__restrict__ loss_ray, float *__restrict__ grad_w_hat) {
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int b = blockIdx.x * kLossWaves + wave;
    if (b >= B) return;
    float part = 0.0f;
    for (int i = l; i < Np; i += kWave) {
        const float bi = bnd[(long)b * bnd_stride + i];
        const float wh = w_hat[(long)b * Np + i];
        const float r = fmaxf(bi - wh, 0.0f), den = wh + 1e-6f;
        part += (r * r) / den;
        if (bounds_out) bounds_out[(long)b * Np + i] = bi;
        if (grad_w_hat) grad_w_hat[(long)b * Np + i] = inv_batch * (-(2.0f * r * den + r * r) / (den * den));
    }
    part = wave_sum(part);
    if (l == 0) loss_ray[b] = part;
}

// ray loss = sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 (s_{i+1} - s_i),  m_i = (s_i + s_{i+1}) / 2
__global__ __launch_bounds__(kLossWaves *kWave) void loss_dist_kernel(
    const float *__restrict__ s_vals, const float *__restrict__ weights, int B, int N, float *__restrict__ loss_ray,
    float *__restrict__ grad_w, float *__restrict__ grad_s) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int b = blockIdx.x * kLossWaves + wave;
    if (b >= B) return;
    float *s = smem + wave * (4 * N + 1), *w = s + N + 1, *m = w + N, *gm = m + N;
    for (int j = l; j <= N; j += kWave) s[j] = s_vals[(long)b * (N + 1) + j];
    for (int j = l; j < N; j += kWave) w[j] = weights[(long)b * N + j];
    lwave_sync();
    for (int j = l; j < N; j += kWave) m[j] = (s[j] + s[j + 1]) / 2.0f;
    lwave_sync();
    float part = 0.0f;
    for (int i = l; i < N; i += kWave) {
        const float wi = w[i], mi = m[i];
        float acc = 0.0f, gsign = 0.0f;
        for (int j = 0; j < N; ++j) {
            const float d = mi - m[j];
            acc += w[j] * fabsf(d);
            gsign += w[j] * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
        }
        const float ds = s[i + 1] - s[i];
        part += wi * acc + (1.0f / 3.0f) * wi * wi * ds;
        if (grad_w) grad_w[(long)b * N + i] = 2.0f * acc + (2.0f / 3.0f) * wi * ds;
        gm[i] = 2.0f * wi * gsign;  // d/dm_i of the pair term
    }
    part = wave_sum(part);
    if (l == 0) loss_ray[b] = part;
    if (grad_s) {
        lwave_sync();
        for (int k = l; k <= N; k += kWave) {
            float g = 0.0f;
            if (k < N) g += 0.5f * gm[k] - (1.0f / 3.0f) * w[k] * w[k];
            if (k > 0) g += 0.5f * gm[k - 1] + (1.0f / 3.0f) * w[k - 1] * w[k - 1];
            grad_s[(long)b * (N + 1) + k] = g;
        }
    }
}

constexpr int kSumParts = 256;
struct SumScratch {
    double partial[kSumParts];
};

__global__ __launch_bounds__(256) void sum_partial_kernel(const float *__restrict__ x, long n, int squared_diff,
                                                          const float *__restrict__ y, SumScratch *__restrict__ ws) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double v = (double)x[i];
        if (squared_diff) {
            const float d = x[i] - y[i];
            v = (double)(d * d);
        }
        acc += v;
    }
    acc = wave_sum_d(acc);
    if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws->partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// mode 0: out = scale * sum;  mode 1 (Loss_nerf): mse = scale * sum, out[0] = 10 log10(mse) + 30, out[1] = -10 log10(mse), out[2] = mse
__global__ void sum_final_kernel(const SumScratch *__restrict__ ws, int parts, double scale, int mode,
                                 float *__restrict__ out) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int p = 0; p < parts; ++p) s += ws->partial[p];
    const float v = (float)(s * scale);
    if (mode == 0) {
        out[0] = v;
    } else {
        const float psnr = -10.0f * log10f(v);
        out[0] = -psnr + 30.0f;
        out[1] = psnr;
        out[2] = v;
    }
}

// d(10 log10(mse) + 30)/d input = (10 / ln 10) / mse * 2 (input - target) / batch
__global__ void loss_nerf_grad_kernel(const float *__restrict__ input, const float *__restrict__ target, long n,
                                      const float *__restrict__ out3, float inv_batch, float *__restrict__ grad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mse = out3[2];
    grad[i] = (4.3429448190325175f / mse) * 2.0f * (input[i] - target[i]) * inv_batch;
}

}  // namespace m360

using namespace m360;

static inline hipStream_t S_(m360_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static int launch_sum(const float *x, const float *y, long n, int squared_diff, double scale, int mode, float *out,
                      void *workspace, size_t workspace_bytes, hipStream_t st, const char *who) {
    if (!workspace || workspace_bytes < sizeof(SumScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "%s: workspace %zu < %zu", who, workspace_bytes, sizeof(SumScratch));
    long parts = (n + 255) / 256;
    parts = parts < 1 ? 1 : (parts > kSumParts ? kSumParts : parts);
    SumScratch *ws = static_cast<SumScratch *>(workspace);
    hipLaunchKernelGGL(sum_partial_kernel, dim3((unsigned)parts), dim3(256), 0, st, x, n, squared_diff, y, ws);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, st, ws, (int)parts, scale, mode, out);
    return check_launch(who);
}

extern "C" {

// workspace: [SumScratch | pad to 256] [loss_ray: B floats | pad] [total: N floats | pad] [per_ray: B*N floats]
static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t m360_loss_workspace_bytes(int B, int N) {
    const size_t b = B > 0 ? B : 0, n = N > 0 ? N : 0;
    return up256(sizeof(SumScratch)) + up256(b * sizeof(float)) + up256(n * sizeof(float)) + up256(b * n * sizeof(float)) + up256((size_t)kColGroups * n * sizeof(double));
}
static inline float *ws_loss_ray(void *ws) { return reinterpret_cast<float *>(static_cast<char *>(ws) + up256(sizeof(SumScratch))); }

int m360_loss_prop(const float *t, const float *w, const float *t_hat, const float *w_hat, int B, int Nf, int Np,
                   float *bounds, float *loss, float *grad_w_hat, void *workspace, size_t workspace_bytes,
                   m360_stream_t stream) {
    if (!w || !w_hat || !loss || B < 1 || Nf < 1 || Np < 1 || (t && !t_hat) || (!t && Nf != Np)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_loss_prop: bad argument");
    if (!workspace || workspace_bytes < m360_loss_workspace_bytes(B, Np)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_loss_prop: workspace %zu < %zu", workspace_bytes, m360_loss_workspace_bytes(B, Np));
    float *loss_ray = ws_loss_ray(workspace);
    const float *bnd = w;
    int bnd_stride = Np;
    const dim3 grid((B + kLossWaves - 1) / kLossWaves), block(kLossWaves * kWave);
    if (t) {
        const size_t lds = (size_t)kLossWaves * (2 * Nf + 1) * sizeof(float);
        if (lds > 64 * 1024) return fail(M360_ERR_INVALID_ARGUMENT, "m360_loss_prop: Nf=%d too large for LDS", Nf);
        float *total = reinterpret_cast<float *>(reinterpret_cast<char *>(loss_ray) + up256((size_t)B * sizeof(float)));
        float *per_ray = reinterpret_cast<float *>(reinterpret_cast<char *>(total) + up256((size_t)Np * sizeof(float)));
        hipLaunchKernelGGL(prop_overlap_kernel, grid, block, lds, S_(stream), t, w, t_hat, B, Nf, Np, per_ray);
        double *part = reinterpret_cast<double *>(reinterpret_cast<char *>(per_ray) + up256((size_t)B * Np * sizeof(float)));
        hipLaunchKernelGGL(prop_colsum_kernel, dim3((Np + kWave - 1) / kWave, kColGroups), dim3(kColWaves * kWave), 0, S_(stream), per_ray, B, Np, part);
        hipLaunchKernelGGL(prop_colsum_final_kernel, dim3((Np + kWave - 1) / kWave), dim3(kWave), 0, S_(stream), part, Np, total);
        bnd = total;
        bnd_stride = 0;
    }
    hipLaunchKernelGGL(loss_prop_kernel, grid, block, 0, S_(stream), bnd, bnd_stride, w_hat, B, Np, 1.0f / (float)B, t ? bounds : nullptr, loss_ray, grad_w_hat);
    return launch_sum(loss_ray, nullptr, B, 0, 1.0 / (double)B, 0, loss, workspace, workspace_bytes, S_(stream), "loss_prop");
}

int m360_loss_dist(const float *s_vals, const float *weights, int B, int N, float *loss, float *grad_w, float *grad_s,
                   void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    if (!s_vals || !weights || !loss || B < 1 || N < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_loss_dist: bad argument");
    if (!workspace || workspace_bytes < m360_loss_workspace_bytes(B, 0)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_loss_dist: workspace %zu < %zu", workspace_bytes, m360_loss_workspace_bytes(B, 0));
    const size_t lds = (size_t)kLossWaves * (4 * N + 1) * sizeof(float);
    if (lds > 64 * 1024) return fail(M360_ERR_INVALID_ARGUMENT, "m360_loss_dist: N=%d too large for LDS", N);
    float *loss_ray = ws_loss_ray(workspace);
    hipLaunchKernelGGL(loss_dist_kernel, dim3((B + kLossWaves - 1) / kLossWaves), dim3(kLossWaves * kWave), lds, S_(stream), s_vals, weights, B, N, loss_ray, grad_w, grad_s);
    return launch_sum(loss_ray, nullptr, B, 0, 1.0, 0, loss, workspace, workspace_bytes, S_(stream), "loss_dist");
}

int m360_loss_nerf(const float *input, const float *target, int B, int C, float *out3, float *grad_input,
                   void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    if (!input || !target || !out3 || B < 1 || C < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_loss_nerf: bad argument");
    const long n = (long)B * C;
    const int rc = launch_sum(input, target, n, 1, 1.0 / (double)B, 1, out3, workspace, workspace_bytes, S_(stream), "loss_nerf");
    if (rc != M360_OK || !grad_input) return rc;
    hipLaunchKernelGGL(loss_nerf_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), input, target, n, out3, 1.0f / (float)B, grad_input);
    return check_launch("loss_nerf_grad");
}

}  // extern "C"
