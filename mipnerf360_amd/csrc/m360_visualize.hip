// Frame post-processing on the device: the reference's visualize_depth,
// visualize_normals, depth_to_normals and sinebow (intern/pose.py:112-212), so that test.py / video.py
// style callers get finished frames without a trip through NumPy / scipy / matplotlib.
// All kernels are per-pixel and HBM-bound; the image statistics (min / max / variances) are a
// deterministic two-level fp64 reduction (fixed partition), like the contraction norm.
#include <string.h>


#include "m360_common.hip.h"
#include "m360_turbo_lut.h"

namespace m360 {

constexpr int kVisParts = 256;
constexpr int kVisStats = 10;  // count, sx, sxx, sy, syy, sd, sdd, min, max, n_nan

struct VisScratch {
    double partial[kVisParts][kVisStats];
    double stats[kVisStats];
};

__global__ __launch_bounds__(256) void vis_stats_partial_kernel(const float *__restrict__ depth, int h, int w,
                                                                VisScratch *__restrict__ ws) {
    __shared__ double red[4][kVisStats];
    double s[kVisStats] = {0, 0, 0, 0, 0, 0, 0, 1e300, -1e300, 0};
    const long n = (long)h * w;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const float d = depth[idx];
        if (isnan(d)) {
            s[9] += 1.0;
            continue;
        }
        const double x = (double)(idx % w), y = (double)(idx / w), dd = (double)d;
        s[0] += 1.0;
        s[1] += x;
        s[2] += x * x;
        s[3] += y;
        s[4] += y * y;
        s[5] += dd;
        s[6] += dd * dd;
        s[7] = fmin(s[7], dd);
        s[8] = fmax(s[8], dd);
    }
#pragma unroll
    for (int k = 0; k < kVisStats; ++k) {
        double v = s[k];
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(v, o, kWave);
            v = (k == 7) ? fmin(v, other) : (k == 8) ? fmax(v, other) : v + other;
        }
        if (lane_id() == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kVisStats) {
        const int k = threadIdx.x;
        double v = red[0][k];
        for (int wv = 1; wv < 4; ++wv) v = (k == 7) ? fmin(v, red[wv][k]) : (k == 8) ? fmax(v, red[wv][k]) : v + red[wv][k];
        ws->partial[blockIdx.x][k] = v;
    }
}

__global__ void vis_stats_final_kernel(VisScratch *__restrict__ ws, int parts) {
    const int k = threadIdx.x;
    if (k >= kVisStats) return;
    double v = ws->partial[0][k];
    for (int p = 1; p < parts; ++p) v = (k == 7) ? fmin(v, ws->partial[p][k]) : (k == 8) ? fmax(v, ws->partial[p][k]) : v + ws->partial[p][k];
    ws->stats[k] = v;
}

// intern/pose.py:112-121 (scipy convolve2d mode='same', zero fill, TRUE convolution => flipped taps)
__device__ __forceinline__ void normals_at(const float *__restrict__ depth, int h, int w, int y, int x, float scale,
                                           float n[3]) {
    auto at = [&](int yy, int xx) -> float {
        return (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0.0f : scale * depth[(long)yy * w + xx];
    };
    // blurred neighbours: [1 2 1]/4 across, [-1 0 1]/2 along (convolution flips the edge taps)
    const float up = 0.25f * at(y - 1, x - 1) + 0.5f * at(y - 1, x) + 0.25f * at(y - 1, x + 1);
    const float dn = 0.25f * at(y + 1, x - 1) + 0.5f * at(y + 1, x) + 0.25f * at(y + 1, x + 1);
    const float lf = 0.25f * at(y - 1, x - 1) + 0.5f * at(y, x - 1) + 0.25f * at(y + 1, x - 1);
    const float rt = 0.25f * at(y - 1, x + 1) + 0.5f * at(y, x + 1) + 0.25f * at(y + 1, x + 1);
    // the zero-weight taps still propagate NaN / inf in the reference (0 * NaN = NaN inside convolve2d): the middle
    // row of the dy stencil and the middle column of the dx stencil, i.e. additionally the centre pixel itself
    const float zc = 0.0f * at(y, x);
    const float dy = 0.5f * (up - dn) + 0.0f * (at(y, x - 1) + at(y, x + 1)) + zc;
    const float dx = 0.5f * (lf - rt) + 0.0f * (at(y - 1, x) + at(y + 1, x)) + zc;
    const float inv = 1.0f / sqrtf(1.0f + dx * dx + dy * dy);
    n[0] = dx * inv;
    n[1] = dy * inv;
    n[2] = inv;
}

__global__ void depth_to_normals_kernel(const float *__restrict__ depth, int h, int w, float *__restrict__ normals) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)h * w) return;
    float n[3];
    normals_at(depth, h, w, (int)(idx / w), (int)(idx % w), 1.0f, n);
#pragma unroll
    for (int c = 0; c < 3; ++c) normals[3 * idx + c] = n[c];
}

// intern/pose.py:127-146
__global__ void visualize_normals_kernel(const float *__restrict__ depth, const float *__restrict__ acc, int h, int w,
                                         const VisScratch *__restrict__ ws, float *__restrict__ vis) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)h * w) return;
    const double *s = ws->stats;
    const double cnt = s[0];
    const double var_x = s[2] / cnt - (s[1] / cnt) * (s[1] / cnt), var_y = s[4] / cnt - (s[3] / cnt) * (s[3] / cnt);
    const double var_z = s[6] / cnt - (s[5] / cnt) * (s[5] / cnt);
    const float scale = (float)sqrt(0.5 * (var_x + var_y) / var_z);
    float n[3];
    normals_at(depth, h, w, (int)(idx / w), (int)(idx % w), scale, n);
    const float a = acc ? acc[idx] : 1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = isnan(n[c]) ? 1.0f : (n[c] + 1.0f) / 2.0f;  // isnan(n) + nan_to_num((n + 1) / 2, 0)
        vis[3 * idx + c] = acc ? v * a + (1.0f - a) : v;
    }
}

__device__ __forceinline__ float sinebow_f(float x) {
    const float s = sinf(3.14159265358979323846f * x);
    return s * s;
}

// intern/pose.py:122-125
__global__ void sinebow_kernel(const float *__restrict__ hval, long n, float *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float hh = hval[idx];
    out[3 * idx] = sinebow_f(3.0f / 6.0f - hh);
    out[3 * idx + 1] = sinebow_f(5.0f / 6.0f - hh);
    out[3 * idx + 2] = sinebow_f(7.0f / 6.0f - hh);
}

// intern/pose.py:148-212 with the default curve -log(x + eps) and ignore_frac = 0
// `curved`: depth / near / far already went through the caller's own curve_fn (a host callable in the reference's API);
// `value_out`: write the normalised value [h,w] (the argument of the colormap) instead of colours - for a caller-supplied
// colormap callable; `planes` (device float[2], may be NULL): the automatic planes of m360_visualize_depth_ex's
// ignore_frac > 0 selection, else ws->stats.
__global__ void visualize_depth_kernel(const float *__restrict__ depth, const float *__restrict__ acc, int h, int w,
                                       float near, float far, int near_auto, int far_auto, float modulus,
                                       const VisScratch *__restrict__ ws, float *__restrict__ vis, int curved = 0,
                                       float *__restrict__ value_out = nullptr, const float *__restrict__ planes = nullptr) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)h * w) return;
    const float eps = 1.1920928955078125e-07f;
    // automatic planes: lowest / highest value of the depth-sorted map (NaNs sort last => far becomes NaN)
    if (near_auto) near = planes ? planes[0] : (float)ws->stats[7] - eps;
    if (far_auto) far = planes ? planes[1] : (ws->stats[9] > 0.0 ? NAN : (float)ws->stats[8]) + eps;
    const float d = depth[idx];
    float a = acc ? acc[idx] : 1.0f;
    if (isnan(d)) a = 0.0f;
    const float cd = curved ? d : -logf(d + eps), cn = curved ? near : -logf(near + eps), cf = curved ? far : -logf(far + eps);
    float rgb[3];
    if (modulus > 0.0f) {
        float m = fmodf(cd, modulus);  // np.mod: result takes the sign of the divisor
        if (m != 0.0f && ((m < 0.0f) != (modulus < 0.0f))) m += modulus;
        const float value = m / modulus;
        if (value_out) {
            value_out[idx] = value;
            return;
        }
        rgb[0] = sinebow_f(3.0f / 6.0f - value);
        rgb[1] = sinebow_f(5.0f / 6.0f - value);
        rgb[2] = sinebow_f(7.0f / 6.0f - value);
    } else {
        float value = (cd - fminf(cn, cf)) / fabsf(cf - cn);
        value = nan_to_numf_(fminf(fmaxf(value, 0.0f), 1.0f));
        if (isnan(cd) || isnan(cn) || isnan(cf)) value = 0.0f;  // np.clip propagates NaN, nan_to_num zeroes it
        if (value_out) {
            value_out[idx] = value;
            return;
        }
        int li = (int)(value * 256.0f);                          // matplotlib ListedColormap lookup (N = 256)
        li = li > 255 ? 255 : (li < 0 ? 0 : li);
        rgb[0] = kTurboLut[li][0];
        rgb[1] = kTurboLut[li][1];
        rgb[2] = kTurboLut[li][2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) vis[3 * idx + c] = rgb[c] * a + (1.0f - a);
}

// ---- ignore_frac > 0 (intern/pose.py:177-192): the planes are the first / last depth of the depth-sorted map whose
// running sum of acc lies inside [f, 1 - f] of the total.  The running sum is numpy's float32 cumsum: SEQUENTIAL fp32
// adds - reproduced by one lane walking the sorted array (a few ms for a megapixel frame; exactness over speed in a
// visualisation helper).
__global__ void vis_prepare_sort_kernel(const float *__restrict__ depth, const float *__restrict__ acc, long n,
                                        float *__restrict__ keys, float *__restrict__ vals) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float d = depth[idx];
    keys[idx] = d;
    vals[idx] = isnan(d) ? 0.0f : (acc ? acc[idx] : 1.0f);
}

// The sort itself (round 5: hand-written - until then rocPRIM's radix_sort_pairs, the build's last vendor-library call).  A frame is at
// most a few million (depth, acc) pairs, sorted once per visualised frame, off the render path: a plain LSD radix sort, four passes of
// 8 bits over the order-preserving integer image of the float keys (sign bit flipped for positives, all bits for negatives: -x < -0 < +0 <
// +x < +Inf < +NaN, the order rocPRIM's float sort gave and numpy's argsort gives up to ties).  Per pass: per-tile digit counts, one
// workgroup turns them into global offsets ([digit][tile], digit-major), then every tile - ONE wave walking its 1024 elements in index
// order, 64 at a time - places its elements: lanes with the same digit find each other with eight ballots, their rank among
// themselves is a popcount below the lane: STABLE, hence deterministic, no atomics on the data path.
constexpr int kSortTile = 1024;
__device__ __forceinline__ unsigned sort_key_bits(float f) {
    const unsigned b = __builtin_bit_cast(unsigned, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(64) void vis_sort_count_kernel(const float *__restrict__ keys, long n, int shift, int ntiles, unsigned *__restrict__ counts /*[256][ntiles]*/) {
    __shared__ unsigned hist[256];
    const int l = threadIdx.x;
    for (int d = l; d < 256; d += 64) hist[d] = 0u;
    __syncthreads();
    const long base = (long)blockIdx.x * kSortTile;
    for (int c = 0; c < kSortTile; c += 64) {
        const long i = base + c + l;
        if (i < n) atomicAdd(&hist[(sort_key_bits(keys[i]) >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = l; d < 256; d += 64) counts[(long)d * ntiles + blockIdx.x] = hist[d];
}
__global__ __launch_bounds__(256) void vis_sort_offsets_kernel(unsigned *__restrict__ counts, int ntiles) {
    // exclusive prefix over the digit-major array [256][ntiles]: thread d walks its digit's tiles, then the digits' totals are chained
    __shared__ unsigned total[256];
    const int d = threadIdx.x;
    unsigned run = 0u;
    for (int t = 0; t < ntiles; ++t) {
        const unsigned c = counts[(long)d * ntiles + t];
        counts[(long)d * ntiles + t] = run;
        run += c;
    }
    total[d] = run;
    __syncthreads();
    unsigned before = 0u;
    for (int e = 0; e < d; ++e) before += total[e];
    for (int t = 0; t < ntiles; ++t) counts[(long)d * ntiles + t] += before;
}
__global__ __launch_bounds__(64) void vis_sort_scatter_kernel(const float *__restrict__ keys, const float *__restrict__ vals, long n, int shift, int ntiles,
                                                               const unsigned *__restrict__ offsets, float *__restrict__ keys_out, float *__restrict__ vals_out) {
    __shared__ unsigned next[256];  // where the tile's next element of each digit goes
    const int l = threadIdx.x;
    for (int d = l; d < 256; d += 64) next[d] = offsets[(long)d * ntiles + blockIdx.x];
    __syncthreads();
    const long base = (long)blockIdx.x * kSortTile;
    const unsigned long long below = (1ull << l) - 1ull;
    for (int c = 0; c < kSortTile; c += 64) {
        const long i = base + c + l;
        const bool live = i < n;
        const float k = live ? keys[i] : 0.0f, v = live ? vals[i] : 0.0f;
        const unsigned digit = (sort_key_bits(k) >> shift) & 255u;
        unsigned long long same = __ballot(live);  // lanes holding an element with MY digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long set = __ballot(live && ((digit >> b) & 1u));
            same &= ((digit >> b) & 1u) ? set : ~set;
        }
        unsigned pos = 0u;
        if (live) pos = next[digit] + (unsigned)__popcll(same & below);
        __syncthreads();  // every lane has read next[] before the group leaders advance it
        if (live && (same & below) == 0ull) next[digit] += (unsigned)__popcll(same);
        __syncthreads();
        if (live) {
            keys_out[pos] = k;
            vals_out[pos] = v;
        }
    }
}

__global__ void vis_cumsum_seq_kernel(const float *__restrict__ vals, long n, float *__restrict__ cum, int *__restrict__ range) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float run = 0.0f;
    for (long i = 0; i < n; ++i) {
        run += vals[i];
        cum[i] = run;
    }
    range[0] = 0x7fffffff;  // first / last index inside the kept band
    range[1] = -1;
}

__global__ void vis_band_kernel(const float *__restrict__ cum, long n, float frac, int *__restrict__ range) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float total = cum[n - 1];
    const float lo = total * frac, hi = total * (float)(1.0 - (double)frac);  // np.float32 * python float (NEP 50)
    const float c = cum[idx];
    if (c >= lo && c <= hi) {
        atomicMin(&range[0], (int)idx);
        atomicMax(&range[1], (int)idx);
    }
}

__global__ void vis_planes_kernel(const float *__restrict__ keys_sorted, const int *__restrict__ range, float *__restrict__ planes) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float eps = 1.1920928955078125e-07f;
    // an empty band raises IndexError in the reference; here both planes become NaN (an all-white frame)
    planes[0] = range[1] >= 0 ? keys_sorted[range[0]] - eps : NAN;
    planes[1] = range[1] >= 0 ? keys_sorted[range[1]] + eps : NAN;
}

__global__ void vis_planes_from_stats_kernel(const VisScratch *__restrict__ ws, float *__restrict__ planes) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float eps = 1.1920928955078125e-07f;
    planes[0] = (float)ws->stats[7] - eps;
    planes[1] = (ws->stats[9] > 0.0 ? NAN : (float)ws->stats[8]) + eps;
}

// intern/pose.py:207-210: vis = colormap(value)[..., :3] * acc + (1 - acc), acc zeroed where depth is NaN
__global__ void vis_composite_kernel(const float *__restrict__ colors, const float *__restrict__ acc,
                                     const float *__restrict__ depth, long n, float *__restrict__ vis) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float a = acc ? acc[idx] : 1.0f;
    if (depth && isnan(depth[idx])) a = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) vis[3 * idx + c] = colors[3 * idx + c] * a + (1.0f - a);
}

static inline size_t vis_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace m360

using namespace m360;

static inline hipStream_t S_(m360_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static int launch_stats(const float *depth, int h, int w, VisScratch *ws, hipStream_t st) {
    const long n = (long)h * w;
    long parts = (n + 255) / 256;
    if (parts > kVisParts) parts = kVisParts;
    hipLaunchKernelGGL(vis_stats_partial_kernel, dim3((unsigned)parts), dim3(256), 0, st, depth, h, w, ws);
    hipLaunchKernelGGL(vis_stats_final_kernel, dim3(1), dim3(64), 0, st, ws, (int)parts);
    return check_launch("vis_stats");
}

extern "C" {

size_t m360_visualize_workspace_bytes(void) { return sizeof(VisScratch); }

int m360_depth_to_normals(const float *depth, int h, int w, float *normals, m360_stream_t stream) {
    if (!depth || !normals || h < 1 || w < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_depth_to_normals: bad argument");
    const long n = (long)h * w;
    hipLaunchKernelGGL(depth_to_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), depth, h, w, normals);
    return check_launch("depth_to_normals");
}

int m360_sinebow(const float *hval, long n, float *rgb, m360_stream_t stream) {
    if (!hval || !rgb || n < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_sinebow: bad argument");
    if (n == 0) return M360_OK;
    hipLaunchKernelGGL(sinebow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), hval, n, rgb);
    return check_launch("sinebow");
}

int m360_visualize_normals(const float *depth, const float *acc, int h, int w, float *vis, void *workspace,
                           size_t workspace_bytes, m360_stream_t stream) {
    if (!depth || !vis || h < 1 || w < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_visualize_normals: bad argument");
    if (!workspace || workspace_bytes < sizeof(VisScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_visualize_normals: workspace %zu < %zu", workspace_bytes, sizeof(VisScratch));
    VisScratch *ws = static_cast<VisScratch *>(workspace);
    const int rc = launch_stats(depth, h, w, ws, S_(stream));
    if (rc != M360_OK) return rc;
    const long n = (long)h * w;
    hipLaunchKernelGGL(visualize_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), depth, acc, h, w, ws, vis);
    return check_launch("visualize_normals");
}

int m360_visualize_depth(const float *depth, const float *acc, int h, int w, float near, float far, int near_auto,
                         int far_auto, float modulus, float *vis, void *workspace, size_t workspace_bytes,
                         m360_stream_t stream) {
    if (!depth || !vis || h < 1 || w < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_visualize_depth: bad argument");
    if (!workspace || workspace_bytes < sizeof(VisScratch)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_visualize_depth: workspace %zu < %zu", workspace_bytes, sizeof(VisScratch));
    VisScratch *ws = static_cast<VisScratch *>(workspace);
    if (near_auto || far_auto) {
        const int rc = launch_stats(depth, h, w, ws, S_(stream));
        if (rc != M360_OK) return rc;
    }
    const long n = (long)h * w;
    hipLaunchKernelGGL(visualize_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), depth, acc, h, w, near, far, near_auto, far_auto, modulus, ws, vis);
    return check_launch("visualize_depth");
}

static size_t sort_temp_bytes(long n) {  // digit counts / offsets of one pass: [256][tiles]
    return (size_t)256 * (size_t)((n + kSortTile - 1) / kSortTile) * sizeof(unsigned);
}

size_t m360_visualize_depth_ex_workspace_bytes(int h, int w) {
    if (h < 1 || w < 1) return 0;
    const long n = (long)h * w;
    // VisScratch | planes[2] + range[2] | keys, vals, keys / vals of the other pass, cum | the sort's digit offsets
    return vis_up(sizeof(VisScratch)) + 256 + 5 * vis_up((size_t)n * sizeof(float)) + vis_up(sort_temp_bytes(n));
}

int m360_visualize_depth_ex(const float *depth, const float *acc, int h, int w, float near, float far, int near_auto,
                            int far_auto, float ignore_frac, int curved, float modulus, float *vis, float *value_out,
                            float *planes_out, void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    if (!depth || h < 1 || w < 1 || !(ignore_frac >= 0.0f) || ignore_frac > 0.5f)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_visualize_depth_ex: bad argument (ignore_frac=%g)", (double)ignore_frac);
    if (curved && (near_auto || far_auto) && (vis || value_out))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_visualize_depth_ex: a pre-curved depth map needs explicit (curved) near / far");
    const size_t need = m360_visualize_depth_ex_workspace_bytes(h, w);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_visualize_depth_ex: workspace %zu < %zu (or not 256-byte aligned)", workspace_bytes, need);
    const long n = (long)h * w;
    hipStream_t st = S_(stream);
    char *base = static_cast<char *>(workspace);
    VisScratch *ws = reinterpret_cast<VisScratch *>(base);
    float *planes = reinterpret_cast<float *>(base + vis_up(sizeof(VisScratch)));
    int *range = reinterpret_cast<int *>(planes + 2);
    const size_t arr = vis_up((size_t)n * sizeof(float));
    float *keys = reinterpret_cast<float *>(base + vis_up(sizeof(VisScratch)) + 256);
    float *vals = reinterpret_cast<float *>(reinterpret_cast<char *>(keys) + arr);
    float *keys_s = reinterpret_cast<float *>(reinterpret_cast<char *>(keys) + 2 * arr);
    float *vals_s = reinterpret_cast<float *>(reinterpret_cast<char *>(keys) + 3 * arr);
    float *cum = reinterpret_cast<float *>(reinterpret_cast<char *>(keys) + 4 * arr);
    void *temp = reinterpret_cast<char *>(keys) + 5 * arr;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (near_auto || far_auto) {
        if (ignore_frac > 0.0f) {
            hipLaunchKernelGGL(vis_prepare_sort_kernel, dim3(blocks), dim3(256), 0, st, depth, acc, n, keys, vals);
            // four stable 8-bit passes, ping-pong between (keys, vals) and (keys_s, vals_s): the sorted pairs end up in (keys, vals)
            const int ntiles = (int)((n + kSortTile - 1) / kSortTile);
            unsigned *offsets = static_cast<unsigned *>(temp);
            float *ki = keys, *vi = vals, *ko = keys_s, *vo = vals_s;
            for (int shift = 0; shift < 32; shift += 8) {
                hipLaunchKernelGGL(vis_sort_count_kernel, dim3((unsigned)ntiles), dim3(64), 0, st, ki, n, shift, ntiles, offsets);
                hipLaunchKernelGGL(vis_sort_offsets_kernel, dim3(1), dim3(256), 0, st, offsets, ntiles);
                hipLaunchKernelGGL(vis_sort_scatter_kernel, dim3((unsigned)ntiles), dim3(64), 0, st, ki, vi, n, shift, ntiles, offsets, ko, vo);
                float *t = ki; ki = ko; ko = t;
                t = vi; vi = vo; vo = t;
            }
            hipLaunchKernelGGL(vis_cumsum_seq_kernel, dim3(1), dim3(64), 0, st, vi, n, cum, range);
            hipLaunchKernelGGL(vis_band_kernel, dim3(blocks), dim3(256), 0, st, cum, n, ignore_frac, range);
            hipLaunchKernelGGL(vis_planes_kernel, dim3(1), dim3(64), 0, st, ki, range, planes);
        } else {
            const int rc = launch_stats(depth, h, w, ws, st);
            if (rc != M360_OK) return rc;
            hipLaunchKernelGGL(vis_planes_from_stats_kernel, dim3(1), dim3(64), 0, st, ws, planes);
        }
        if (planes_out && hipMemcpyAsync(planes_out, planes, 2 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(M360_ERR_LAUNCH, "m360_visualize_depth_ex: copy of the planes failed");
    }
    if (vis || value_out)
        hipLaunchKernelGGL(visualize_depth_kernel, dim3(blocks), dim3(256), 0, st, depth, acc, h, w, near, far, near_auto, far_auto, modulus, ws, vis, curved, value_out, (near_auto || far_auto) ? planes : nullptr);
    return check_launch("visualize_depth_ex");
}

int m360_visualize_composite(const float *colors, const float *acc, const float *depth, int h, int w, float *vis,
                             m360_stream_t stream) {
    if (!colors || !vis || h < 1 || w < 1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_visualize_composite: bad argument");
    const long n = (long)h * w;
    hipLaunchKernelGGL(vis_composite_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), colors, acc, depth, n, vis);
    return check_launch("visualize_composite");
}

}  // extern "C"
