// libm360 C-ABI glue: error reporting and the stage / whole-forward drivers that chain the
// kernels of m360_sample_encode.hip, m360_linear.hip and m360_ray.hip on ONE caller-owned HIP
// stream without allocating or synchronising (mipNeRF360.forward, model.py:247-252).
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "m360_common.cuh"

namespace m360 {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return M360_OK;
    return fail(M360_ERR_LAUNCH, "%s: kernel launch failed: %s", what, hipGetErrorString(e));
}

struct ProfRec {
    hipEvent_t start, stop;
    long M;
    int n_pad, k_pad;
};
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;
static bool g_prof_on = false;

int prof_begin(hipStream_t st, long M, int n_pad, int k_pad) {
    if (!g_prof_on || g_prof_used >= g_prof.size()) return -1;
    ProfRec &r = g_prof[g_prof_used];
    r.M = M;
    r.n_pad = n_pad;
    r.k_pad = k_pad;
    if (hipEventRecord(r.start, st) != hipSuccess) return -1;
    return (int)g_prof_used++;
}

void prof_end(int idx, hipStream_t st) {
    if (idx >= 0) (void)hipEventRecord(g_prof[idx].stop, st);
}

__global__ void add_eps_kernel(const float *__restrict__ x, long n, float *__restrict__ y) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) y[idx] = x[idx] + kEpsG;
}

static inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct FwdLayout {
    size_t norm, vdenc, t1, t0, what, feat, act_a, act_b, total;
};

static FwdLayout layout_for(int B, int N, const m360_model_t *m) {
    FwdLayout L;
    const size_t S = (size_t)B * N;
    const int vd_ch = m->in_ch - kIpeCh;
    const size_t wmax = (size_t)(m->hp_pad > m->hn_pad ? m->hp_pad : m->hn_pad);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    L.norm = take(m360_contract_workspace_bytes());
    L.vdenc = take((size_t)B * (vd_ch > 0 ? vd_ch : 1) * sizeof(float));
    L.t1 = take((size_t)B * (N + 1) * sizeof(float));
    L.t0 = take((size_t)B * (N + 1) * sizeof(float));
    L.what = take((size_t)B * N * sizeof(float));
    L.feat = take(S * m->in_pad * sizeof(float));
    L.act_a = take(S * wmax * sizeof(float));
    L.act_b = take(S * wmax * sizeof(float));
    L.total = off;
    return L;
}

static inline int n_fine(const m360_hyper_t *h) { return h->num_samples_fine > 0 ? h->num_samples_fine : h->num_samples; }
static inline int n_max(const m360_hyper_t *h) { return n_fine(h) > h->num_samples ? n_fine(h) : h->num_samples; }

static int validate(const m360_rays_t *r, const m360_model_t *m, const m360_hyper_t *h, int B,
                    const void *ws, size_t ws_bytes, const char *who) {
    if (!r || !m || !h) return fail(M360_ERR_INVALID_ARGUMENT, "%s: null descriptor", who);
    if (B < 0 || h->num_samples < 1 || h->num_samples_fine < 0) return fail(M360_ERR_INVALID_ARGUMENT, "%s: B=%d num_samples=%d num_samples_fine=%d", who, B, h->num_samples, h->num_samples_fine);
    if (B == 0) return M360_OK;  // empty batch: nothing is dereferenced
    if (!r->origins || !r->directions || !r->viewdirs || !r->radii || !r->near || !r->far)
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: null ray field", who);
    const int vd_ch = 4 * (h->viewdir_max_deg - h->viewdir_min_deg);
    if (vd_ch < 0 || m->in_ch != kIpeCh + vd_ch || m->in_pad < m->in_ch || m->in_pad % 32 || m->hp_pad % 32 || m->hn_pad % 32 || m->hp_pad < 32 || m->hn_pad < 32)
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: model dims inconsistent (in_ch=%d in_pad=%d hp_pad=%d hn_pad=%d vd_ch=%d)", who, m->in_ch, m->in_pad, m->hp_pad, m->hn_pad, vd_ch);
    if (m->mlp_bf16 && (m->in_pad % 64 || m->hp_pad % 64 || m->hn_pad % 64))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: the bf16 MLP needs in_pad/hp_pad/hn_pad multiples of 64", who);
    const FwdLayout L = layout_for(B, n_max(h), m);
    if (B > 0 && (!ws || ws_bytes < L.total)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "%s: workspace %zu < required %zu bytes", who, ws_bytes, L.total);
    if ((uintptr_t)ws & 255) return fail(M360_ERR_INVALID_ARGUMENT, "%s: workspace must be 256-byte aligned", who);
    return M360_OK;
}

#define M360_TRY(expr)            \
    do {                          \
        const int rc_ = (expr);   \
        if (rc_ != M360_OK) return rc_; \
    } while (0)

// sample (or take) t -> features -> 4 proposal layers -> head + weights (+ fused resample)
static int prop_stage(const m360_rays_t *r, const m360_model_t *m, const m360_hyper_t *h, int B,
                      const float *t_rand, float *t_hat, float *w_hat, float *t_new, char *ws,
                      m360_stream_t st) {
    const int N = h->num_samples;
    const FwdLayout L = layout_for(B, n_max(h), m);
    const int vd_ch = m->in_ch - kIpeCh;
    float *vdenc = reinterpret_cast<float *>(ws + L.vdenc);
    float *feat = reinterpret_cast<float *>(ws + L.feat);
    float *a = reinterpret_cast<float *>(ws + L.act_a), *b = reinterpret_cast<float *>(ws + L.act_b);
    const long S = (long)B * N;
    M360_TRY(m360_sample_t(r->near, r->far, t_rand, B, N, t_hat, st));
    M360_TRY(m360_viewdir_enc(r->viewdirs, B, h->viewdir_min_deg, h->viewdir_max_deg, vdenc, st));
    const int hp = m->hp_pad;
    if (m->mlp_bf16) {  // opt-in: bf16 features / weights / activations, fp32 accumulation (same buffers, half the bytes)
        M360_TRY(m360_encode_features_bf16(t_hat, r->origins, r->directions, r->radii, vdenc, vd_ch, B, N, feat, m->in_pad, ws + L.norm, m360_contract_workspace_bytes(), st));
        M360_TRY(m360_linear_bf16(feat, S, m->in_pad, m->prop_w[0], m->prop_b[0], hp, m->in_pad, M360_ACT_RELU, a, hp, st));
        M360_TRY(m360_linear_bf16(a, S, hp, m->prop_w[1], m->prop_b[1], hp, hp, M360_ACT_RELU, b, hp, st));
        M360_TRY(m360_linear_bf16(b, S, hp, m->prop_w[2], m->prop_b[2], hp, hp, M360_ACT_RELU, a, hp, st));
        M360_TRY(m360_linear_bf16(a, S, hp, m->prop_w[3], m->prop_b[3], hp, hp, M360_ACT_SIGMOID, b, hp, st));
        return m360_prop_finish_bf16(b, hp, m->prop_head_w, m->prop_head_b, hp, h->density_bias, t_hat, r->directions, nullptr, B, N, n_fine(h) + 1, h->resample_padding, w_hat, t_new, st);
    }
    M360_TRY(m360_encode_features(t_hat, r->origins, r->directions, r->radii, vdenc, vd_ch, B, N, feat, m->in_pad, ws + L.norm, m360_contract_workspace_bytes(), st));
    M360_TRY(m360_linear(feat, S, m->in_pad, m->prop_w[0], m->prop_b[0], hp, m->in_pad, M360_ACT_RELU, a, hp, st));
    M360_TRY(m360_linear(a, S, hp, m->prop_w[1], m->prop_b[1], hp, hp, M360_ACT_RELU, b, hp, st));
    M360_TRY(m360_linear(b, S, hp, m->prop_w[2], m->prop_b[2], hp, hp, M360_ACT_RELU, a, hp, st));
    M360_TRY(m360_linear(a, S, hp, m->prop_w[3], m->prop_b[3], hp, hp, M360_ACT_SIGMOID, b, hp, st));
    return m360_prop_finish_n(b, hp, m->prop_head_w, m->prop_head_b, hp, h->density_bias, t_hat, r->directions, nullptr, B, N, n_fine(h) + 1, h->resample_padding, w_hat, t_new, st);
}

// resampled t -> features -> 8 NeRF layers -> heads + composite
static int nerf_stage(const m360_rays_t *r, const m360_model_t *m, const m360_hyper_t *h, int B,
                      const float *t1, const m360_outputs_t *out, char *ws, m360_stream_t st) {
    const int N = n_fine(h);  // the NeRF stage runs on the resampled intervals
    const FwdLayout L = layout_for(B, n_max(h), m);
    const int vd_ch = m->in_ch - kIpeCh;
    float *vdenc = reinterpret_cast<float *>(ws + L.vdenc);
    float *feat = reinterpret_cast<float *>(ws + L.feat);
    float *a = reinterpret_cast<float *>(ws + L.act_a), *b = reinterpret_cast<float *>(ws + L.act_b);
    const long S = (long)B * N;
    M360_TRY(m360_viewdir_enc(r->viewdirs, B, h->viewdir_min_deg, h->viewdir_max_deg, vdenc, st));
    const int hn = m->hn_pad;
    float *src = a, *dst = b;
    if (m->mlp_bf16) {
        M360_TRY(m360_encode_features_bf16(t1, r->origins, r->directions, r->radii, vdenc, vd_ch, B, N, feat, m->in_pad, ws + L.norm, m360_contract_workspace_bytes(), st));
        M360_TRY(m360_linear_bf16(feat, S, m->in_pad, m->nerf_w[0], m->nerf_b[0], hn, m->in_pad, M360_ACT_RELU, a, hn, st));
        for (int layer = 1; layer < 8; ++layer) {
            M360_TRY(m360_linear_bf16(src, S, hn, m->nerf_w[layer], m->nerf_b[layer], hn, hn, layer == 7 ? M360_ACT_SIGMOID : M360_ACT_RELU, dst, hn, st));
            float *tmp = src; src = dst; dst = tmp;
        }
        M360_TRY(m360_nerf_finish_bf16(src, hn, m->nerf_head_w, m->nerf_head_b, hn, h->density_bias, h->rgb_padding, t1, r->directions, B, N, h->white_bkgd, out->rgb, out->distance, out->acc, out->fine_w, st));
    } else {
    M360_TRY(m360_encode_features(t1, r->origins, r->directions, r->radii, vdenc, vd_ch, B, N, feat, m->in_pad, ws + L.norm, m360_contract_workspace_bytes(), st));
    M360_TRY(m360_linear(feat, S, m->in_pad, m->nerf_w[0], m->nerf_b[0], hn, m->in_pad, M360_ACT_RELU, a, hn, st));
    for (int layer = 1; layer < 8; ++layer) {
        M360_TRY(m360_linear(src, S, hn, m->nerf_w[layer], m->nerf_b[layer], hn, hn, layer == 7 ? M360_ACT_SIGMOID : M360_ACT_RELU, dst, hn, st));
        float *tmp = src; src = dst; dst = tmp;
    }
    M360_TRY(m360_nerf_finish(src, hn, m->nerf_head_w, m->nerf_head_b, hn, h->density_bias, h->rgb_padding, t1, r->directions, B, N, h->white_bkgd, out->rgb, out->distance, out->acc, out->fine_w, st));
    }
    if (out->t_vals) {  // model.py:194,196: g() inside t_to_s bumps the stored t_vals by 1e-6
        const long n = (long)B * (N + 1);
        hipLaunchKernelGGL(add_eps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(st), t1, n, out->t_vals);
        M360_TRY(check_launch("add_eps"));
    }
    if (out->s_vals) M360_TRY(m360_t_to_s(t1, r->near, r->far, B, N + 1, 1, 1, out->s_vals, st));
    return M360_OK;
}

}  // namespace m360

using namespace m360;

extern "C" {

int m360_version(void) { return M360_VERSION; }
const char *m360_last_error(void) { return g_err; }

int m360_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int m360_prof_enable(int capacity) {
    for (ProfRec &r : g_prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    g_prof.clear();
    g_prof_used = 0;
    g_prof_on = false;
    if (capacity <= 0) return M360_OK;
    g_prof.resize((size_t)capacity);
    for (ProfRec &r : g_prof) {
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) {
            g_prof.clear();
            return fail(M360_ERR_LAUNCH, "m360_prof_enable: hipEventCreate failed");
        }
    }
    g_prof_on = true;
    return M360_OK;
}

int m360_prof_count(void) { return (int)g_prof_used; }

int m360_prof_reset(void) {
    g_prof_used = 0;
    return M360_OK;
}

int m360_prof_read(int i, float *ms, long *M, int *n_pad, int *k_pad) {
    if (i < 0 || (size_t)i >= g_prof_used || !ms) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prof_read: bad index %d", i);
    ProfRec &r = g_prof[i];
    if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(ms, r.start, r.stop) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_prof_read: event query failed");
    if (M) *M = r.M;
    if (n_pad) *n_pad = r.n_pad;
    if (k_pad) *k_pad = r.k_pad;
    return M360_OK;
}

size_t m360_forward_workspace_bytes(int B, int N, const m360_model_t *model_host) {
    if (!model_host || B < 0 || N < 1) return 0;
    return layout_for(B, N, model_host).total;
}

int m360_prop_forward(const m360_rays_t *rays, const m360_model_t *model, const m360_hyper_t *hyper, int B,
                      const float *t_rand, float *t_hat, float *w_hat, void *workspace,
                      size_t workspace_bytes, m360_stream_t stream) {
    M360_TRY(validate(rays, model, hyper, B, workspace, workspace_bytes, "m360_prop_forward"));
    if (B > 0 && (!t_hat || !w_hat)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_forward: t_hat and w_hat are required");
    if (B == 0) return M360_OK;
    return prop_stage(rays, model, hyper, B, t_rand, t_hat, w_hat, nullptr, static_cast<char *>(workspace), stream);
}

int m360_nerf_forward(const m360_rays_t *rays, const m360_model_t *model, const m360_hyper_t *hyper, int B,
                      const float *t_hat, const float *w_hat, const float *u_rand,
                      const m360_outputs_t *out, void *workspace, size_t workspace_bytes,
                      m360_stream_t stream) {
    M360_TRY(validate(rays, model, hyper, B, workspace, workspace_bytes, "m360_nerf_forward"));
    if (B == 0) return M360_OK;
    if (!t_hat || !w_hat || !out || !out->rgb || !out->distance || !out->acc)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_forward: t_hat, w_hat, out.rgb/distance/acc are required");
    if (B == 0) return M360_OK;
    char *ws = static_cast<char *>(workspace);
    const FwdLayout L = layout_for(B, n_max(hyper), model);
    float *t1 = reinterpret_cast<float *>(ws + L.t1);
    M360_TRY(m360_resample_t_n(t_hat, w_hat, u_rand, B, hyper->num_samples, n_fine(hyper) + 1, hyper->resample_padding, t1, stream));
    return nerf_stage(rays, model, hyper, B, t1, out, ws, stream);
}

int m360_forward(const m360_rays_t *rays, const m360_model_t *model, const m360_hyper_t *hyper, int B,
                 const m360_outputs_t *out, void *workspace, size_t workspace_bytes,
                 m360_stream_t stream) {
    M360_TRY(validate(rays, model, hyper, B, workspace, workspace_bytes, "m360_forward"));
    if (B == 0) return M360_OK;
    if (!out || !out->rgb || !out->distance || !out->acc) return fail(M360_ERR_INVALID_ARGUMENT, "m360_forward: out.rgb/distance/acc are required");
    if (B == 0) return M360_OK;
    char *ws = static_cast<char *>(workspace);
    const FwdLayout L = layout_for(B, n_max(hyper), model);
    float *t0 = out->t_hat ? out->t_hat : reinterpret_cast<float *>(ws + L.t0);
    float *what = out->w_hat ? out->w_hat : reinterpret_cast<float *>(ws + L.what);
    float *t1 = reinterpret_cast<float *>(ws + L.t1);
    M360_TRY(prop_stage(rays, model, hyper, B, nullptr, t0, what, t1, ws, stream));
    return nerf_stage(rays, model, hyper, B, t1, out, ws, stream);
}

}  // extern "C"
