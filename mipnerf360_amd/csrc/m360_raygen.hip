// Ray generation on the device: pinhole rays + radii for every pixel of every
// camera (dataset.py:109-145 of the reference), optional NDC conversion with NDC radii
// (dataset.py:364-387, intern/ray.py:59-79).  Removes the 48 B/ray host->device stream for frame /
// video rendering: only the 48-byte pose per camera crosses PCIe.  One thread per pixel, everything
// recomputed from the pose (neighbour pixels included), 48 B/ray of coalesced SoA stores: HBM-bound.
#include "m360_common.hip.h"

namespace m360 {

struct Cam {
    float r[3][3], t[3];
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ c2w, int c) {
    Cam m;
    const float *p = c2w + 12 * c;  // [3][4] row-major
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) m.r[i][j] = p[4 * i + j];
        m.t[i] = p[4 * i + 3];
    }
    return m;
}

// dataset.py:113-123: camera-frame direction of pixel (x, y) rotated into the world frame
__device__ __forceinline__ void pixel_dir(const Cam &m, int x, int y, int h, int w, float focal, float d[3]) {
    const float cx = ((float)x - (float)w * 0.5f + 0.5f) / focal;
    const float cy = -((float)y - (float)h * 0.5f + 0.5f) / focal;
    const float cz = -1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = (cx * m.r[i][0] + cy * m.r[i][1]) + cz * m.r[i][2];
}

// intern/ray.py:59-79 (sx = 2 focal / w, sy = 2 focal / h evaluated in double on the host)
__device__ __forceinline__ void to_ndc(const float o_in[3], const float d[3], float sx, float sy, float near,
                                       float o[3], float dn[3]) {
    const float t = -(near + o_in[2]) / (d[2] + 1e-15f);
    float p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = o_in[i] + t * d[i];
    const float oz = p[2] + 1e-15f, dz = d[2] + 1e-15f;
    o[0] = -sx * (p[0] / oz);
    o[1] = -sy * (p[1] / oz);
    o[2] = 1.0f + 2.0f * near / oz;
    dn[0] = -sx * (d[0] / dz - p[0] / oz);
    dn[1] = -sy * (d[1] / dz - p[1] / oz);
    dn[2] = -2.0f * near / oz;
}

__device__ __forceinline__ float dist3(const float a[3], const float b[3]) {
    const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return sqrtf((x * x + y * y) + z * z);
}

template <bool NDC>
__global__ void generate_rays_kernel(const float *__restrict__ c2w, long first, long count, int h, int w, float focal,
                                     float sx, float sy, float near, float far, float ndc_near,
                                     float *__restrict__ origins, float *__restrict__ directions,
                                     float *__restrict__ viewdirs, float *__restrict__ radii,
                                     float *__restrict__ near_out, float *__restrict__ far_out) {
    // idx = row of the OUTPUT arrays; pix = flat pixel index over all cameras (the span starts at `first`)
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per_cam = (long)h * w;
    if (idx >= count) return;
    const long pix = first + idx;
    const int c = (int)(pix / per_cam);
    const int y = (int)((pix % per_cam) / w), x = (int)(pix % w);
    const Cam m = load_cam(c2w, c);
    float d[3];
    pixel_dir(m, x, y, h, w, focal, d);
    const float dn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    float o[3] = {m.t[0], m.t[1], m.t[2]};
    float rad;
    if (!NDC) {
        // dataset.py:129-135: distance to the neighbour one ROW down; the last row gets `dx[:, -2:-1]`, which is
        // the SECOND-to-last of the h-1 differences, i.e. the pair (h-3, h-2)
        const int y0 = y < h - 1 ? y : h - 3;
        float a[3], b[3];
        pixel_dir(m, x, y0, h, w, focal, a);
        pixel_dir(m, x, y0 + 1, h, w, focal, b);
        rad = dist3(a, b) * 2.0f / 3.4641016151377544f;
#pragma unroll
        for (int i = 0; i < 3; ++i) directions[3 * idx + i] = d[i];
    } else {
        // dataset.py:367-377: radii from the NDC origins of the row / column neighbours
        auto ndc_origin = [&](int xx, int yy, float out[3]) {
            float dd[3], dtmp[3];
            pixel_dir(m, xx, yy, h, w, focal, dd);
            to_ndc(m.t, dd, sx, sy, ndc_near, out, dtmp);
        };
        const int y0 = y < h - 1 ? y : h - 3, x0 = x < w - 1 ? x : w - 3;  // same `[-2:-1]` padding rule
        float a[3], b[3];
        ndc_origin(x, y0, a);
        ndc_origin(x, y0 + 1, b);
        const float dx = dist3(a, b);
        ndc_origin(x0, y, a);
        ndc_origin(x0 + 1, y, b);
        const float dy = dist3(a, b);
        rad = (0.5f * (dx + dy)) * 2.0f / 3.4641016151377544f;
        float on[3], dnn[3];
        to_ndc(m.t, d, sx, sy, ndc_near, on, dnn);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            o[i] = on[i];
            directions[3 * idx + i] = dnn[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        origins[3 * idx + i] = o[i];
        viewdirs[3 * idx + i] = d[i] / dn;  // dataset.py:126: unit world-space direction (also for LLFF)
    }
    radii[idx] = rad;
    near_out[idx] = near;
    far_out[idx] = far;
}

__global__ void convert_to_ndc_kernel(const float *__restrict__ o_in, const float *__restrict__ d_in, long n,
                                      float sx, float sy, float near, float *__restrict__ o_out,
                                      float *__restrict__ d_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float oi[3] = {o_in[3 * idx], o_in[3 * idx + 1], o_in[3 * idx + 2]};
    const float di[3] = {d_in[3 * idx], d_in[3 * idx + 1], d_in[3 * idx + 2]};
    float o[3], d[3];
    to_ndc(oi, di, sx, sy, near, o, d);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o_out[3 * idx + i] = o[i];
        d_out[3 * idx + i] = d[i];
    }
}

}  // namespace m360

using namespace m360;

extern "C" {

int m360_generate_rays_span(const float *cam_to_world, int n_cams, int h, int w, float focal, float near, float far,
                            int ndc, float ndc_near, long first, long count, float *origins, float *directions,
                            float *viewdirs, float *radii, float *near_out, float *far_out, m360_stream_t stream) {
    if (n_cams < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_generate_rays: negative camera count");
    if (h < 3 || w < 3 || !(focal > 0.0f)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_generate_rays: h=%d w=%d must be >= 3 (the reference's radii padding reads difference n-3), focal=%g > 0", h, w, (double)focal);
    const long n = (long)n_cams * h * w;
    if (first < 0 || count < 0 || first + count > n)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_generate_rays_span: span [%ld, %ld) outside the %ld pixels of %d camera(s)", first, first + count, n, n_cams);
    if (count == 0) return M360_OK;  // an empty span (no cameras, or a rank without chunks) needs no buffers
    if (!cam_to_world || !origins || !directions || !viewdirs || !radii || !near_out || !far_out)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_generate_rays: null pointer");
    const float sx = (float)((2.0 * (double)focal) / (double)w), sy = (float)((2.0 * (double)focal) / (double)h);
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (ndc) hipLaunchKernelGGL(generate_rays_kernel<true>, grid, block, 0, st, cam_to_world, first, count, h, w, focal, sx, sy, near, far, ndc_near, origins, directions, viewdirs, radii, near_out, far_out);
    else hipLaunchKernelGGL(generate_rays_kernel<false>, grid, block, 0, st, cam_to_world, first, count, h, w, focal, sx, sy, near, far, ndc_near, origins, directions, viewdirs, radii, near_out, far_out);
    return check_launch("generate_rays");
}

int m360_generate_rays(const float *cam_to_world, int n_cams, int h, int w, float focal, float near, float far,
                       int ndc, float ndc_near, float *origins, float *directions, float *viewdirs, float *radii,
                       float *near_out, float *far_out, m360_stream_t stream) {
    return m360_generate_rays_span(cam_to_world, n_cams, h, w, focal, near, far, ndc, ndc_near, 0,
                                   n_cams < 0 ? 0 : (long)n_cams * h * w, origins, directions, viewdirs, radii, near_out,
                                   far_out, stream);
}

int m360_convert_to_ndc(const float *origins, const float *directions, long n, float focal, int w, int h, float near,
                        float *origins_out, float *directions_out, m360_stream_t stream) {
    if (!origins || !directions || !origins_out || !directions_out || n < 0 || w < 1 || h < 1)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_convert_to_ndc: bad argument");
    if (n == 0) return M360_OK;
    const float sx = (float)((2.0 * (double)focal) / (double)w), sy = (float)((2.0 * (double)focal) / (double)h);
    hipLaunchKernelGGL(convert_to_ndc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), origins, directions, n, sx, sy, near, origins_out, directions_out);
    return check_launch("convert_to_ndc");
}

}  // extern "C"
