// libm360 C-ABI glue: error reporting and the stage / whole-forward drivers that chain the
// kernels of m360_sample_encode.hip, m360_linear.hip and m360_ray.hip on ONE caller-owned HIP
// stream without allocating or synchronising (mipNeRF360.forward, model.py:247-252).
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "m360_common.hip.h"

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

}  // namespace m360

// Caller-owned event recorder (include/m360.h "measurement"): no state lives in the library.
struct m360_prof {
    struct Rec {
        hipEvent_t start, stop;
        int kind;
        long M;
        int n_pad, k_pad;
    };
    std::vector<Rec> recs;
    size_t used = 0;
};

// Caller-owned second stream + fork / join events (include/m360.h, m360_side_t): no stream, event or handle lives in the library.
struct m360_side {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int device = -1;
};

namespace m360 {

// m360_sample_encode.hip (stage drivers only, not part of the C-ABI)
int stage_prologue(const m360_rays_t *r, int B, int N, int min_deg, int max_deg, float *t_vals, float *vdenc,
                   unsigned *queue_words, int n_queue_words, int ld_feat, void *norm_ws, m360_stream_t stream, const rng_t &rng);
int sample_t_any(const float *near, const float *far, const float *t_rand, int B, int N, float *t_vals, const rng_t &rng, m360_stream_t stream);
// m360_ray.hip
int resample_t_any(const float *t_vals, const float *weights, const float *u_rand, int B, int N, int num_out, float resample_padding, float *t_new,
                   const rng_t &rng, m360_stream_t stream);
int encode_stage(const float *t_vals, const float *origins, const float *directions, const float *radii, const float *vdenc,
                 int vd_ch, int B, int N, void *feat, int ld_feat, int row_format, int group_rays, const float *ext_norm,
                 int prepared_parts, unsigned char *nanflag, void *workspace, size_t workspace_bytes, m360_stream_t stream);
// m360_ray.hip (stage drivers only)
int prop_finish_stage(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots, const float *head_w,
                      const float *head_b, int k_pad, float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                      int B, int N, int num_out, float resample_padding, float *weights, float *t_new, const unsigned char *nanflag,
                      m360_stream_t stream, const rng_t &rng, float *raw_out);
int nerf_finish_stage(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots, const float *head_w,
                      const float *head_b, int k_pad, float density_bias, float rgb_padding, const float *t_vals, const float *dirs,
                      const float *near, const float *far, int near_far_calls, int B, int N, int white_bkgd, float *comp_rgb,
                      float *distance, float *acc, float *weights, float *t_vals_out, float *s_vals_out, const unsigned char *nanflag,
                      m360_stream_t stream, float *raw_out);

int prop_finish_backward_stage(const void *act, int act_bf16, int ld, const float *head_w, const float *head_b, int k_pad, float density_bias, const float *t_vals,
                               const float *dirs, int B, int N, const float *grad_weights, void *dz, float *grad_head_w, float *grad_head_b,
                               void *workspace, size_t workspace_bytes, const float *raw_in, m360_stream_t stream);
int nerf_finish_backward_stage(const void *act, int act_bf16, int ld, const float *head_w, const float *head_b, int k_pad, float density_bias, float rgb_padding,
                               const float *t_vals, const float *dirs, int B, int N, int white_bkgd, const float *grad_rgb, const float *grad_distance,
                               const float *grad_acc, const float *grad_weights, void *dz, float *grad_head_w, float *grad_head_b, void *workspace,
                               size_t workspace_bytes, const float *raw_in, m360_stream_t stream);
// m360_linear.hip: m360_linear_wgrad_bf16 with operand rows that overlap (the first layer: see mlp_backward_bf16)
int linear_wgrad_bf16_rows(const void *dz, int ldz, const void *x, int ldx, long M, int n_pad, int k_pad, float *grad_w, float *grad_b,
                           void *workspace, size_t workspace_bytes, unsigned tuning, m360_stream_t stream, bool x_rows_overlap);
// m360_linear.hip: the hidden-layer chain of the bf16 mode (one launch) and its gated layer-by-layer re-run
int mlp_chain_bf16_launch(const void *x_in, void *act0, void *act1, long M, int ld, const void *const *w_packed, const float *const *b_packed,
                          int layers, int width, void *ws, m360_stream_t stream, const m360_hyper_t *opts, bool x3);
int mlp_chain_bf16_rerun(const void *x_in, void *act0, void *act1, long M, int ld, const void *const *w_packed, const float *const *b_packed,
                         int layers, int width, void *ws, m360_stream_t stream, const m360_hyper_t *opts, bool x3);
// m360_linear.hip: the ReLU mask of m360_linear_dgrad_bf16 on its own (mlp_backward_bf16 overlaps it with the weight gradient)
int relu_mask_bf16(void *dx, const void *relu_out, long M, int k_pad, int ldx, m360_stream_t stream, int blocks /* > 0: that many striding workgroups */,
                   float *sums_part /* != NULL: [blocks][k_pad] column sums of the masked rows per workgroup */);
bool relu_mask_bf16_sums_ok(long M, int k_pad, int blocks);
int relu_mask_bf16_sums_reduce(const float *part, int blocks, int k_pad, float *grad_b, m360_stream_t stream);
size_t relu_mask_bf16_sums_bytes(int blocks);

// brackets the launches of ONE public entry point with two HIP events on the launch stream
struct ProfScope {
    m360_prof *p;
    hipStream_t st;
    int idx = -1;
    ProfScope(const m360_hyper_t *h, m360_stream_t stream, int kind, long M, int n_pad, int k_pad)
        : p(h ? static_cast<m360_prof *>(h->prof) : nullptr), st(reinterpret_cast<hipStream_t>(stream)) {
        if (!p || p->used >= p->recs.size()) return;
        m360_prof::Rec &r = p->recs[p->used];
        r.kind = kind;
        r.M = M;
        r.n_pad = n_pad;
        r.k_pad = k_pad;
        if (hipEventRecord(r.start, st) == hipSuccess) idx = (int)p->used++;
    }
    int done(int rc) {
        if (idx >= 0) (void)hipEventRecord(p->recs[idx].stop, st);
        idx = -1;
        return rc;
    }
};
#define M360_PROF(h, st, kind, M, n, k, expr)                \
    do {                                                     \
        ProfScope ps_((h), (st), (kind), (M), (n), (k));     \
        const int rc_ = ps_.done(expr);                      \
        if (rc_ != M360_OK) return rc_;                      \
    } while (0)

static inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct FwdLayout {
    size_t norm, vdenc, t1, t0, what, feat, act_a, act_b, act_c, hpart, queues, nanflag, chain, total;
};
// tile-queue words of the balanced linear launches of one stage (m360_linear_balanced): 16 words, one per 64-byte line
constexpr int kQueueSlots = 16, kQueueStride = 16;
struct TileQueues {
    unsigned *base = nullptr;
    int next = 0;
    unsigned *take() { return (base && next < kQueueSlots) ? base + kQueueStride * next++ : nullptr; }
};
// the workspace holds one set of queue words per stage (set 0: proposal, set 1: NeRF); `cleared`: the proposal stage's fused
// prologue has already zeroed both sets in this forward (stage_prologue) - no memset launch
constexpr int kQueueWords = kQueueSlots * kQueueStride;
static int queues_begin(TileQueues *q, char *ws, size_t off, int set, bool cleared, m360_stream_t st) {
    q->base = reinterpret_cast<unsigned *>(ws + off) + set * kQueueWords;
    q->next = 0;
    if (!cleared && hipMemsetAsync(q->base, 0, (size_t)kQueueWords * sizeof(unsigned), reinterpret_cast<hipStream_t>(st)) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "tile queues: memset failed");
    return M360_OK;
}

static FwdLayout layout_for(int B, int N, const m360_model_t *m) {
    FwdLayout L;
    const size_t S = (size_t)B * N;
    const int vd_ch = m->in_ch - kIpeCh;
    const size_t wmax = (size_t)(m->hp_pad > m->hn_pad ? m->hp_pad : m->hn_pad);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    // offset 0 of EVERY forward workspace: the 128-byte status block (m360_workspace_init / m360_workspace_status), in bf16 mode followed
    // by the counters of the hidden-layer chain (m360_mlp_chain_bf16_workspace = status + counters)
    L.chain = take(m->mlp_bf16 == 1 && S >= 32768 ? m360_mlp_chain_bf16_workspace((long)(S / 32768) * 32768, 6) : 128);
    L.norm = take(m360_contract_workspace_bytes());
    L.vdenc = take((size_t)B * (vd_ch > 0 ? vd_ch : 1) * sizeof(float));
    L.t1 = take((size_t)B * (N + 1) * sizeof(float));
    L.t0 = take((size_t)B * (N + 1) * sizeof(float));
    L.what = take((size_t)B * N * sizeof(float));
    // MLP input rows: fp32 [in_pad]; bf16: [hi | lo] pairs of in_pad bf16 each; bf16x3: x6 rows of 6 in_pad bf16 (see p_linear_first)
    L.feat = take(S * m->in_pad * (m->mlp_bf16 == 2 ? 6 * sizeof(unsigned short) : sizeof(float)));  // bf16: [hi | lo] pairs = 4 bytes per value too
    L.act_a = take(S * wmax * sizeof(float));
    L.act_b = take(S * wmax * sizeof(float));
    // (bf16x3 mode: m360_mlp_chain_bf16x3_safe exists and is bit-identical, but the forward does not use it - measured in round 6: 16.98 against
    // 17.01 ms per step with six launches (three matrix passes per byte of activations: the round trip through HBM is not what bounds those
    // layers), and its [hi | lo] rows fill a and b completely, so the chain's untouched third input buffer would cost 2.1 GB more workspace)
    L.act_c = 0;
    // partial head sums of the fused last layer: [S][slots][heads] fp32 (67 MB at 4096 x 128, width 1024)
    // sized by the rows the fused epilogue really covers: none in bf16 mode or at widths it does not take
    const size_t hp_slots = (size_t)m360_linear_heads_slots(m->hp_pad, m->mlp_bf16), hn_slots = (size_t)m360_linear_heads_slots(m->hn_pad, m->mlp_bf16);
    const size_t hp_rows = (size_t)m360_linear_heads_fused_rows((long)S, m->hp_pad, m->mlp_bf16), hn_rows = (size_t)m360_linear_heads_fused_rows((long)S, m->hn_pad, m->mlp_bf16);
    const size_t hp_b = hp_rows * hp_slots * 1 * sizeof(float), hn_b = hn_rows * hn_slots * 4 * sizeof(float);
    L.hpart = take(hp_b > hn_b ? hp_b : hn_b);
    L.queues = take((size_t)2 * kQueueSlots * kQueueStride * sizeof(unsigned));  // one set per stage
    // bf16 modes: one byte per sample, "a feature of this sample is NaN" (the bf16 pipe's ReLU drops NaN: the finishers restore it)
    L.nanflag = take(m->mlp_bf16 ? S : 0);
    L.total = off;
    return L;
}

static inline int n_fine(const m360_hyper_t *h) { return h->num_samples_fine > 0 ? h->num_samples_fine : h->num_samples; }
// m360_hyper_t.randomized: bit 0 = the proposal samples' stratified jitter (intern/ray.py:103-108), bit 1 = the randomized inverse CDF
// (intern/ray.py:30-35), both drawn inside the kernels from the Philox stream (rng_seed, rng_offset) whenever the entry point's own
// t_rand / u_rand tensor is NULL
static inline rng_t rng_jitter(const m360_hyper_t *h, const float *t_rand) { return rng_t{h->rng_seed, h->rng_offset, (!t_rand && (h->randomized & 1)) ? 1 : 0}; }
static inline rng_t rng_cdf(const m360_hyper_t *h, const float *u_rand) { return rng_t{h->rng_seed, h->rng_offset, (!u_rand && (h->randomized & 2)) ? 1 : 0}; }
static inline int n_max(const m360_hyper_t *h) { return n_fine(h) > h->num_samples ? n_fine(h) : h->num_samples; }

static int validate(const m360_rays_t *r, const m360_model_t *m, const m360_hyper_t *h, int B,
                    const void *ws, size_t ws_bytes, const char *who) {
    if (!r || !m || !h) return fail(M360_ERR_INVALID_ARGUMENT, "%s: null descriptor", who);
    if (B < 0 || h->num_samples < 1 || h->num_samples_fine < 0 || h->norm_group_rays < 0) return fail(M360_ERR_INVALID_ARGUMENT, "%s: B=%d num_samples=%d num_samples_fine=%d norm_group_rays=%d", who, B, h->num_samples, h->num_samples_fine, h->norm_group_rays);
    if (B == 0) return M360_OK;  // empty batch: nothing is dereferenced
    if (!r->origins || !r->directions || !r->viewdirs || !r->radii || !r->near || !r->far)
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: null ray field", who);
    const int vd_ch = 4 * (h->viewdir_max_deg - h->viewdir_min_deg);
    if (vd_ch < 0 || m->in_ch != kIpeCh + vd_ch || m->in_pad < m->in_ch || m->in_pad % 32 || m->hp_pad % 32 || m->hn_pad % 32 || m->hp_pad < 32 || m->hn_pad < 32)
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: model dims inconsistent (in_ch=%d in_pad=%d hp_pad=%d hn_pad=%d vd_ch=%d)", who, m->in_ch, m->in_pad, m->hp_pad, m->hn_pad, vd_ch);
    if (m->mlp_bf16 < 0 || m->mlp_bf16 > 2 || (m->mlp_bf16 ? m->packed_layout != M360_PACKED_LAYOUT : (m->packed_layout != 0 && m->packed_layout != M360_PACKED_LAYOUT)))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: m360_model_t.mlp_bf16=%d with packed_layout=%d: this library reads layout %d (first layers of the bf16 modes as [n_pad, 3 in_pad] / [n_pad, 6 in_pad]: re-pack with m360_pack_linear_bf16x3 / _bf16x6)", who, m->mlp_bf16, m->packed_layout, M360_PACKED_LAYOUT);
    if (m->mlp_bf16 && (m->in_pad % 64 || m->hp_pad % 64 || m->hn_pad % 64))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: the bf16 MLP needs in_pad/hp_pad/hn_pad multiples of 64", who);
    const FwdLayout L = layout_for(B, n_max(h), m);
    if (B > 0 && (!ws || ws_bytes < L.total)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "%s: workspace %zu < required %zu bytes", who, ws_bytes, L.total);
    if ((uintptr_t)ws & 255) return fail(M360_ERR_INVALID_ARGUMENT, "%s: workspace must be 256-byte aligned", who);
#ifndef M360_DIAG
    if (h->chain_debug_wait_ticks != 0 || h->chain_debug_fault != 0 || (h->tuning & M360_TUNE_CHAIN_UNGATED))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: m360_hyper_t.chain_debug_* / M360_TUNE_CHAIN_UNGATED are test hooks of the diagnostics build (libm360_diag.so); this library has none", who);
#endif
    return M360_OK;
}

#define M360_TRY(expr)            \
    do {                          \
        const int rc_ = (expr);   \
        if (rc_ != M360_OK) return rc_; \
    } while (0)

// the entry points the stage drivers chain, each bracketed by the caller's event recorder (hyper->prof, may be NULL)
static int p_linear(const m360_hyper_t *h, TileQueues *q, const float *x, long M, int ldx, const float *w, const float *b, int n_pad, int k_pad, int act, float *y, int ldy, m360_stream_t st) {
    ProfScope ps(h, st, M360_K_LINEAR, M, n_pad, k_pad);
    return ps.done(m360_linear_balanced(x, M, ldx, w, b, n_pad, k_pad, act, y, ldy, q->take(), st));
}
// First layer of the bf16 (mode 1) / bf16x3 (mode 2) MLP: it sees more bits of the features than the hidden layers do.  bf16x3: the
// encoder hands it "x6" rows - every feature as THREE bf16 terms, all 24 bits - and the weights are packed the same way
// (m360_pack_linear_bf16x6), so one plain bf16 contraction of length 6 in_pad forms the fp32 product up to 2^-24 terms, written out as
// [hi | lo] pairs.  Why: the reference contracts a whole
// chunk by its Frobenius norm (intern/parameterization.py:23-29), which squeezes a ray's samples into ~1e-2 of the unit ball, so a
// net that resolves anything along a ray has first-layer gains of ~1e3-1e4 on DIFFERENCES of the sin / cos features: bf16
// features (8 bits) put the density shells of fixture G19 at the wrong depth (PSNR off by 3-6 dB), two-term features (16 bits)
// leave the bf16x3 mode at |d rgb| ~2e-4, outside the fp32 tolerance.
// The bf16 mode rounds the layer's output to bf16 anyway and gets by with 16 bits of each operand (PSNR within 0.013 dB on G19):
// [hi | lo] features, [Wh | Wh | Wl] weights, the ring kernel's three-product loop with the plain bf16 epilogue - one 64-deep block per
// tile, store-bound like round 3's first layer (0.30 instead of the x6 form's 0.53 ms for the NeRF net).
static inline int first_row_format(int mode) { return mode == 2 ? 3 : (mode == 1 ? 2 : 0); }  // encoder row format of the MLP input
// Paired rows (m360.h) between the layers of one bf16 / bf16x3 MLP: only when EVERY layer of it runs its full tiles on the one-wave ring
// kernel - first layer out, hidden layers in and out, fused-heads last layer in (rendering forward: the tape-keeping one keeps plain rows)
// (Round 6: the process-wide switches that lived here - hidden chain, paired rows, row blocks, backward overlap - are per-call bits
// of m360_hyper_t.tuning now, the library-owned second stream a caller-owned handle; the row-block form of the NeRF MLP is gone with its switch:
// profiles/HISTORY.md.)
static bool mlp_rows_pairable(const m360_hyper_t *h, int mode, int width, int in_pad) {
    if (h->tuning & M360_TUNE_PLAIN_ROWS) return false;  // A/B of the two layouts (same bits either way)
    if (mode == 2) return m360_linear_bf16_rows_pairable(M360_PAIRABLE_SPLIT, width, 6 * in_pad) && m360_linear_bf16_rows_pairable(M360_PAIRABLE_X3, width, width) && m360_linear_bf16_rows_pairable(M360_PAIRABLE_HEADS_X3, width, width);
    return m360_linear_bf16_rows_pairable(M360_PAIRABLE_X3_BF16OUT, width, in_pad) && m360_linear_bf16_rows_pairable(M360_PAIRABLE_LINEAR, width, width) && m360_linear_bf16_rows_pairable(M360_PAIRABLE_HEADS, width, width);
}

static int p_linear_first(const m360_hyper_t *h, int mode, const void *feat, long M, const void *w0, const float *b, int n_pad, int in_pad, void *y, int pair, m360_stream_t st, int temporal = 0) {
    ProfScope ps(h, st, M360_K_LINEAR_BF16, M, n_pad, (mode == 2 ? 6 : 3) * in_pad);
    const int act = M360_ACT_RELU | (pair ? M360_ROWS_PAIRED_OUT : 0) | (temporal ? M360_STORES_TEMPORAL : 0);
    if (mode == 2) return ps.done(m360_linear_bf16_split(feat, M, 6 * in_pad, w0, b, n_pad, 6 * in_pad, act, y, 2 * n_pad, st));
    return ps.done(m360_linear_bf16x3_bf16out(feat, M, 2 * in_pad, w0, b, n_pad, in_pad, act, y, n_pad, st));
}

// mode 1: bf16 rows of k_pad / n_pad columns; mode 2 (bf16x3): [hi | lo] rows of 2 k_pad / 2 n_pad columns
static int p_linear_bf16(const m360_hyper_t *h, int mode, const void *x, long M, const void *w, const float *b, int n_pad, int k_pad, int act, void *y, int pair, m360_stream_t st, int temporal = 0) {
    ProfScope ps(h, st, M360_K_LINEAR_BF16, M, n_pad, mode == 2 ? 3 * k_pad : k_pad);
    if (pair) act |= M360_ROWS_PAIRED_IN | M360_ROWS_PAIRED_OUT;
    if (temporal) act |= M360_STORES_TEMPORAL;
    if (mode == 2) return ps.done(m360_linear_bf16x3(x, M, 2 * k_pad, w, b, n_pad, k_pad, act, y, 2 * n_pad, st));
    return ps.done(m360_linear_bf16(x, M, k_pad, w, b, n_pad, k_pad, act, y, n_pad, st));
}
// the encode step of a stage (encode_stage: per-chunk norms, an external norm, or partial sums left by stage_prologue); flags: the
// per-sample NaN flags of the bf16 modes (NULL in fp32)
static int p_encode(const m360_hyper_t *h, const float *t, const m360_rays_t *r, const float *vdenc, int vd_ch, int B, int N, void *feat, int ld, int row_format, int group, const float *ext_norm, int prepared_parts, unsigned char *flags, void *ws, size_t wsb, m360_stream_t st) {
    ProfScope ps(h, st, M360_K_ENCODE, (long)B * N, ld, row_format);
    return ps.done(encode_stage(t, r->origins, r->directions, r->radii, vdenc, vd_ch, B, N, feat, ld, row_format, group, ext_norm, prepared_parts, flags, ws, wsb, st));
}
// last hidden layer + heads fused (fp32 or bf16): partial head sums to `part`, y written only when store_y
static int p_linear_heads(const m360_hyper_t *h, int bf16, const void *x, long M, int ldx, const void *w, const float *b, int n_pad, int k_pad, void *y, int ldy, int store_y, const float *head_w, int heads, float *part, m360_stream_t st, int pair = 0) {
    ProfScope ps(h, st, M360_K_LINEAR_HEADS, M, n_pad, bf16 ? -k_pad : k_pad);
    const int sig = M360_ACT_SIGMOID | (pair ? M360_ROWS_PAIRED_IN : 0);
    if (bf16 == 2) return ps.done(m360_linear_heads_bf16x3(x, M, ldx, w, b, n_pad, k_pad, sig, y, ldy, store_y, head_w, heads, part, st));
    if (bf16) return ps.done(m360_linear_heads_bf16(x, M, ldx, w, b, n_pad, k_pad, sig, y, ldy, store_y, head_w, heads, part, st));
    return ps.done(m360_linear_heads(static_cast<const float *>(x), M, ldx, static_cast<const float *>(w), b, n_pad, k_pad, M360_ACT_SIGMOID, static_cast<float *>(y), ldy, store_y, head_w, heads, part, st));
}
static int p_prop_finish_fused(const m360_hyper_t *h, const void *act, int bf16, int ld, const float *part, long fused_rows, int slots, const float *hw, const float *hb, int k_pad, const float *t, const float *dirs, int B, int N, float *w_hat, float *t_new, m360_stream_t st, const unsigned char *flags = nullptr, float *raw_out = nullptr) {
    ProfScope ps(h, st, M360_K_PROP_FINISH, (long)B * N, k_pad, bf16);
    // the fused resample (t_new != NULL: m360_forward / the sharded batch) draws its uniforms itself when the model is randomized
    return ps.done(prop_finish_stage(act, bf16, ld, part, fused_rows, slots, hw, hb, k_pad, h->density_bias, t, dirs, nullptr, B, N, n_fine(h) + 1, h->resample_padding, w_hat, t_new, flags, st, rng_cdf(h, nullptr), raw_out));
}
// heads + composite, and in the same launch the t_vals + 1e-6 and s_vals nerf_net.forward returns (model.py:194-196; until round 3
// an add_eps and a t_to_s launch).  near / far went through g() once in sample_along_rays (numerically, or physically when
// rays_mutated: then t_to_s starts from the values it is handed).
static int p_nerf_finish_fused(const m360_hyper_t *h, const void *act, int bf16, int ld, const float *part, long fused_rows, int slots, const float *hw, const float *hb, int k_pad, const float *t, const m360_rays_t *r, int B, int N, const m360_outputs_t *out, m360_stream_t st, const unsigned char *flags = nullptr, float *raw_out = nullptr) {
    ProfScope ps(h, st, M360_K_NERF_FINISH, (long)B * N, k_pad, bf16);
    return ps.done(nerf_finish_stage(act, bf16, ld, part, fused_rows, slots, hw, hb, k_pad, h->density_bias, h->rgb_padding, t, r->directions, r->near, r->far, h->rays_mutated ? 0 : 1, B, N, h->white_bkgd, out->rgb, out->distance, out->acc, out->fine_w, out->t_vals, out->s_vals, flags, st, raw_out));
}

// Training tape of one stage (caller-owned): everything the backward needs from the forward.
struct TapeLayout {
    size_t t, feat, act[8], raw, total;  // raw: the head sums [B*N][1 or 4] as the forward's finisher formed them (round 6: the backward's finisher
    int layers;                          // starts from them instead of re-reading the last layer's rows for the head dot products)
};
static TapeLayout tape_for(int B, int N, const m360_model_t *m, int stage) {
    TapeLayout T;
    const size_t S = (size_t)B * N;
    const int width = stage == 0 ? m->hp_pad : m->hn_pad;
    T.layers = stage == 0 ? 4 : 8;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    T.t = take((size_t)B * (N + 1) * sizeof(float));
    T.feat = take(S * m->in_pad * sizeof(float));  // bf16 mode: [hi | lo] pairs of in_pad bf16 each - the same 4 bytes per value
    const size_t el = m->mlp_bf16 == 1 ? sizeof(unsigned short) : sizeof(float);  // bf16 mode: every layer's output as bf16 rows
    for (int l = 0; l < 8; ++l) T.act[l] = l < T.layers ? take(S * width * el) : 0;
    T.raw = take(S * (stage == 0 ? 1 : 4) * sizeof(float));
    T.total = off;
    return T;
}

// sample (or take) t -> features -> 4 proposal layers -> head + weights (+ fused resample)
// fused: out-flag, set when this call ran the one-launch prologue (stage_prologue): the view-direction encoding in the workspace
// and BOTH stages' tile-queue words are then ready for a NeRF stage that follows in the same forward (m360_forward)
static int prop_stage(const m360_rays_t *r, const m360_model_t *m, const m360_hyper_t *h, int B,
                      const float *t_rand, float *t_hat, float *w_hat, float *t_new, char *ws,
                      m360_stream_t st, char *tape = nullptr, const float *ext_norm = nullptr, bool *fused = nullptr) {
    const int N = h->num_samples;
    const FwdLayout L = layout_for(B, n_max(h), m);
    const int vd_ch = m->in_ch - kIpeCh;
    float *vdenc = reinterpret_cast<float *>(ws + L.vdenc);
    float *feat = reinterpret_cast<float *>(ws + L.feat);
    float *a = reinterpret_cast<float *>(ws + L.act_a), *b = reinterpret_cast<float *>(ws + L.act_b);
    float *hpart = reinterpret_cast<float *>(ws + L.hpart);
    const long S = (long)B * N;
    unsigned char *flags = m->mlp_bf16 ? reinterpret_cast<unsigned char *>(ws + L.nanflag) : nullptr;  // per-sample NaN flags (bf16 modes)
    TileQueues tq;
    // rendering forward on a chunk of its own norm, deterministic samples: ONE prologue launch (t, view directions, norm partials,
    // queue words) instead of five; the encoder's workgroups finish the norm themselves
    int parts = 0;
    if (!tape && !ext_norm && !t_rand && h->norm_group_rays == 0) {
        parts = stage_prologue(r, B, N, h->viewdir_min_deg, h->viewdir_max_deg, t_hat, vdenc, reinterpret_cast<unsigned *>(ws + L.queues), 2 * kQueueWords, m->in_pad, ws + L.norm, st, rng_jitter(h, t_rand));
        if (parts < 0) return M360_ERR_LAUNCH;
    }
    if (fused) *fused = parts > 0;
    M360_TRY(queues_begin(&tq, ws, L.queues, 0, parts > 0, st));
    if (tape && m->mlp_bf16 == 1) {  // bf16 training (round 5): plain rows, every layer's bf16 output kept; the encoder's [hi | lo] feature rows too
        const TapeLayout T = tape_for(B, N, m, 0);
        const int hp = m->hp_pad;
        float *tt = reinterpret_cast<float *>(tape + T.t);
        void *tf = tape + T.feat;
        void *act[4];
        for (int l = 0; l < 4; ++l) act[l] = tape + T.act[l];
        M360_TRY(sample_t_any(r->near, r->far, t_rand, B, N, tt, rng_jitter(h, t_rand), st));
        if (hipMemcpyAsync(t_hat, tt, (size_t)B * (N + 1) * sizeof(float), hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(st)) != hipSuccess)
            return fail(M360_ERR_LAUNCH, "m360_prop_forward_train: copy of t_hat failed");
        M360_TRY(m360_viewdir_enc(r->viewdirs, B, h->viewdir_min_deg, h->viewdir_max_deg, vdenc, st));
        M360_TRY(p_encode(h, tt, r, vdenc, vd_ch, B, N, tf, m->in_pad, first_row_format(1), h->norm_group_rays, nullptr, 0, flags, ws + L.norm, m360_contract_workspace_bytes(), st));
        M360_TRY(p_linear_first(h, 1, tf, S, m->prop_w[0], m->prop_b[0], hp, m->in_pad, act[0], 0, st));
        for (int l = 1; l < 3; ++l)
            M360_TRY(p_linear_bf16(h, 1, act[l - 1], S, m->prop_w[l], m->prop_b[l], hp, hp, M360_ACT_RELU, act[l], 0, st));
        M360_TRY(p_linear_heads(h, 1, act[2], S, hp, m->prop_w[3], m->prop_b[3], hp, hp, act[3], hp, 1, m->prop_head_w, 1, hpart, st));
        return p_prop_finish_fused(h, act[3], 1, hp, hpart, m360_linear_heads_fused_rows(S, hp, 1), m360_linear_heads_slots_bf16(hp, hp, 1, 1), m->prop_head_w, m->prop_head_b, hp, tt, r->directions, B, N, w_hat, t_new, st, flags, reinterpret_cast<float *>(tape + T.raw));
    }
    if (tape) {  // training, fp32: every layer output kept
        const TapeLayout T = tape_for(B, N, m, 0);
        const int hp = m->hp_pad;
        float *tt = reinterpret_cast<float *>(tape + T.t), *tf = reinterpret_cast<float *>(tape + T.feat);
        float *act[4];
        for (int l = 0; l < 4; ++l) act[l] = reinterpret_cast<float *>(tape + T.act[l]);
        M360_TRY(sample_t_any(r->near, r->far, t_rand, B, N, tt, rng_jitter(h, t_rand), st));
        if (hipMemcpyAsync(t_hat, tt, (size_t)B * (N + 1) * sizeof(float), hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(st)) != hipSuccess)
            return fail(M360_ERR_LAUNCH, "m360_prop_forward_train: copy of t_hat failed");
        M360_TRY(m360_viewdir_enc(r->viewdirs, B, h->viewdir_min_deg, h->viewdir_max_deg, vdenc, st));
        M360_TRY(p_encode(h, tt, r, vdenc, vd_ch, B, N, tf, m->in_pad, 0, h->norm_group_rays, nullptr, 0, nullptr, ws + L.norm, m360_contract_workspace_bytes(), st));
        M360_TRY(p_linear(h, &tq, tf, S, m->in_pad, m->prop_w[0], m->prop_b[0], hp, m->in_pad, M360_ACT_RELU, act[0], hp, st));
        for (int l = 1; l < 3; ++l)
            M360_TRY(p_linear(h, &tq, act[l - 1], S, hp, m->prop_w[l], m->prop_b[l], hp, hp, M360_ACT_RELU, act[l], hp, st));
        // last hidden layer + head fused; the tape keeps the layer output (store_y = 1), same partial sums as when rendering
        M360_TRY(p_linear_heads(h, 0, act[2], S, hp, m->prop_w[3], m->prop_b[3], hp, hp, act[3], hp, 1, m->prop_head_w, 1, hpart, st));
        return p_prop_finish_fused(h, act[3], 0, hp, hpart, m360_linear_heads_fused_rows(S, hp, 0), m360_linear_heads_slots(hp, 0), m->prop_head_w, m->prop_head_b, hp, tt, r->directions, B, N, w_hat, t_new, st, nullptr, reinterpret_cast<float *>(tape + T.raw));
    }
    if (parts == 0) {
        if (!ext_norm) M360_TRY(sample_t_any(r->near, r->far, t_rand, B, N, t_hat, rng_jitter(h, t_rand), st));  // sharded batch: t_hat is given
        M360_TRY(m360_viewdir_enc(r->viewdirs, B, h->viewdir_min_deg, h->viewdir_max_deg, vdenc, st));
    }
    const int hp = m->hp_pad;
    if (ext_norm) {  // the caller supplies the (all-reduced) contraction norm; the layers below are shared
        M360_TRY(p_encode(h, t_hat, r, vdenc, vd_ch, B, N, feat, m->in_pad, first_row_format(m->mlp_bf16), 0, ext_norm, 0, flags, ws + L.norm, m360_contract_workspace_bytes(), st));  /* row format of the mode */
    }
    if (m->mlp_bf16) {  // opt-in: bf16 features / weights / activations, fp32 accumulation (same buffers; mode 2 = bf16x3: [hi | lo] pairs)
        const int mode = m->mlp_bf16;
        if (!ext_norm) M360_TRY(p_encode(h, t_hat, r, vdenc, vd_ch, B, N, feat, m->in_pad, first_row_format(mode), h->norm_group_rays, nullptr, parts, flags, ws + L.norm, m360_contract_workspace_bytes(), st));
        const int pair = mlp_rows_pairable(h, mode, hp, m->in_pad) ? 1 : 0;  // paired rows between the layers (m360.h)
        M360_TRY(p_linear_first(h, mode, feat, S, m->prop_w[0], m->prop_b[0], hp, m->in_pad, a, pair, st));
        // (the proposal MLP's two hidden layers stay two launches: at width 256 the chain - a workgroup owns whole rows there - measured
        // 0.245 against 0.201 ms, profiles/r04/chain_bench_w256_SLOWER.jsonl: its tiles are too short for the hand-over's fixed costs)
        M360_TRY(p_linear_bf16(h, mode, a, S, m->prop_w[1], m->prop_b[1], hp, hp, M360_ACT_RELU, b, pair, st));
        M360_TRY(p_linear_bf16(h, mode, b, S, m->prop_w[2], m->prop_b[2], hp, hp, M360_ACT_RELU, a, pair, st));
        // last hidden layer + head on the matrix pipe (full 256-row tiles; tail rows through y): ld of y = hp (bf16) / 2 hp ([hi | lo])
        const int ldl = mode == 2 ? 2 * hp : hp;
        M360_TRY(p_linear_heads(h, mode, a, S, ldl, m->prop_w[3], m->prop_b[3], hp, hp, b, ldl, 0, m->prop_head_w, 1, hpart, st, pair));
        return p_prop_finish_fused(h, b, mode, ldl, hpart, m360_linear_heads_fused_rows(S, hp, mode), m360_linear_heads_slots_bf16(hp, hp, mode, 0), m->prop_head_w, m->prop_head_b, hp, t_hat, r->directions, B, N, w_hat, t_new, st, flags);
    }
    if (!ext_norm) M360_TRY(p_encode(h, t_hat, r, vdenc, vd_ch, B, N, feat, m->in_pad, 0, h->norm_group_rays, nullptr, parts, nullptr, ws + L.norm, m360_contract_workspace_bytes(), st));
    M360_TRY(p_linear(h, &tq, feat, S, m->in_pad, m->prop_w[0], m->prop_b[0], hp, m->in_pad, M360_ACT_RELU, a, hp, st));
    M360_TRY(p_linear(h, &tq, a, S, hp, m->prop_w[1], m->prop_b[1], hp, hp, M360_ACT_RELU, b, hp, st));
    M360_TRY(p_linear(h, &tq, b, S, hp, m->prop_w[2], m->prop_b[2], hp, hp, M360_ACT_RELU, a, hp, st));
    // last hidden layer + head fused: its 537 MB output never goes to HBM (store_y = 0; ragged tail rows excepted)
    M360_TRY(p_linear_heads(h, 0, a, S, hp, m->prop_w[3], m->prop_b[3], hp, hp, b, hp, 0, m->prop_head_w, 1, hpart, st));
    return p_prop_finish_fused(h, b, 0, hp, hpart, m360_linear_heads_fused_rows(S, hp, 0), m360_linear_heads_slots(hp, 0), m->prop_head_w, m->prop_head_b, hp, t_hat, r->directions, B, N, w_hat, t_new, st);
}

// resampled t -> features -> 8 NeRF layers -> heads + composite
// after_fused_prop: the proposal stage of the same forward ran the fused prologue - the view-direction encoding (identical in
// both stages, intern/encoding.py:69-90) and this stage's queue words are in place
static int nerf_stage(const m360_rays_t *r, const m360_model_t *m, const m360_hyper_t *h, int B,
                      const float *t1, const m360_outputs_t *out, char *ws, m360_stream_t st, char *tape = nullptr,
                      const float *ext_norm = nullptr, bool after_fused_prop = false) {
    const int N = n_fine(h);  // the NeRF stage runs on the resampled intervals
    const FwdLayout L = layout_for(B, n_max(h), m);
    const int vd_ch = m->in_ch - kIpeCh;
    float *vdenc = reinterpret_cast<float *>(ws + L.vdenc);
    float *feat = reinterpret_cast<float *>(ws + L.feat);
    float *a = reinterpret_cast<float *>(ws + L.act_a), *b = reinterpret_cast<float *>(ws + L.act_b);
    float *hpart = reinterpret_cast<float *>(ws + L.hpart);
    const long S = (long)B * N;
    unsigned char *flags = m->mlp_bf16 ? reinterpret_cast<unsigned char *>(ws + L.nanflag) : nullptr;  // per-sample NaN flags (bf16 modes)
    TileQueues tq;
    M360_TRY(queues_begin(&tq, ws, L.queues, 1, after_fused_prop, st));
    if (!after_fused_prop) M360_TRY(m360_viewdir_enc(r->viewdirs, B, h->viewdir_min_deg, h->viewdir_max_deg, vdenc, st));
    const int hn = m->hn_pad;
    const int slots = m360_linear_heads_slots(hn, m->mlp_bf16);
    float *src = a, *dst = b;
    if (tape && m->mlp_bf16 == 1) {  // bf16 training (round 5): plain rows, every layer's bf16 output kept (t1 already lives in the tape)
        const TapeLayout T = tape_for(B, N, m, 1);
        void *tf = tape + T.feat;
        void *act[8];
        for (int l = 0; l < 8; ++l) act[l] = tape + T.act[l];
        M360_TRY(p_encode(h, t1, r, vdenc, vd_ch, B, N, tf, m->in_pad, first_row_format(1), h->norm_group_rays, nullptr, 0, flags, ws + L.norm, m360_contract_workspace_bytes(), st));
        M360_TRY(p_linear_first(h, 1, tf, S, m->nerf_w[0], m->nerf_b[0], hn, m->in_pad, act[0], 0, st));
        for (int l = 1; l < 7; ++l)
            M360_TRY(p_linear_bf16(h, 1, act[l - 1], S, m->nerf_w[l], m->nerf_b[l], hn, hn, M360_ACT_RELU, act[l], 0, st));
        M360_TRY(p_linear_heads(h, 1, act[6], S, hn, m->nerf_w[7], m->nerf_b[7], hn, hn, act[7], hn, 1, m->nerf_head_w, 4, hpart, st));
        M360_TRY(p_nerf_finish_fused(h, act[7], 1, hn, hpart, m360_linear_heads_fused_rows(S, hn, 1), m360_linear_heads_slots_bf16(hn, hn, 1, 1), m->nerf_head_w, m->nerf_head_b, hn, t1, r, B, N, out, st, flags, reinterpret_cast<float *>(tape + T.raw)));
    } else if (tape) {  // training, fp32: every layer output kept (t1 already lives in the tape)
        const TapeLayout T = tape_for(B, N, m, 1);
        float *tf = reinterpret_cast<float *>(tape + T.feat);
        float *act[8];
        for (int l = 0; l < 8; ++l) act[l] = reinterpret_cast<float *>(tape + T.act[l]);
        M360_TRY(p_encode(h, t1, r, vdenc, vd_ch, B, N, tf, m->in_pad, 0, h->norm_group_rays, nullptr, 0, nullptr, ws + L.norm, m360_contract_workspace_bytes(), st));
        M360_TRY(p_linear(h, &tq, tf, S, m->in_pad, m->nerf_w[0], m->nerf_b[0], hn, m->in_pad, M360_ACT_RELU, act[0], hn, st));
        for (int l = 1; l < 7; ++l)
            M360_TRY(p_linear(h, &tq, act[l - 1], S, hn, m->nerf_w[l], m->nerf_b[l], hn, hn, M360_ACT_RELU, act[l], hn, st));
        M360_TRY(p_linear_heads(h, 0, act[6], S, hn, m->nerf_w[7], m->nerf_b[7], hn, hn, act[7], hn, 1, m->nerf_head_w, 4, hpart, st));
        M360_TRY(p_nerf_finish_fused(h, act[7], 0, hn, hpart, m360_linear_heads_fused_rows(S, hn, 0), slots, m->nerf_head_w, m->nerf_head_b, hn, t1, r, B, N, out, st, nullptr, reinterpret_cast<float *>(tape + T.raw)));
    } else if (m->mlp_bf16) {
        const int mode = m->mlp_bf16;
        M360_TRY(p_encode(h, t1, r, vdenc, vd_ch, B, N, feat, m->in_pad, first_row_format(mode), ext_norm ? 0 : h->norm_group_rays, ext_norm, 0, flags, ws + L.norm, m360_contract_workspace_bytes(), st));
        const int pair = mlp_rows_pairable(h, mode, hn, m->in_pad) ? 1 : 0;  // paired rows between the layers (m360.h)
        const int ldl = mode == 2 ? 2 * hn : hn;
        {
        // the six hidden layers: ONE launch for the rows the chain takes (bf16 mode, paired rows, width 1024, multiples of 32768 rows:
        // m360_mlp_chain_bf16 - the activations handed over through the XCDs' L2s instead of six kernel boundaries), layer by layer the rest
        const long Mc = (mode == 1 && pair && !(h->tuning & M360_TUNE_NO_HIDDEN_CHAIN)) ? (S / 32768) * 32768 : 0;
        const bool chain = Mc > 0 && m360_mlp_chain_bf16_supported(Mc, hn, 6);
        // With the chain the first layer writes to a THIRD buffer - the upper half of `a`, which is sized for fp32 rows and holds bf16 ones
        // here - that no hidden layer writes: what the chain read stays intact, so the launch can be repeated layer by layer when it reports
        // that one of its assumptions did not hold (gated launches queued behind it: empty unless its error word is set).
        float *first = !chain ? a : mode == 2 ? reinterpret_cast<float *>(ws + L.act_c) : reinterpret_cast<float *>(reinterpret_cast<char *>(a) + (size_t)S * ldl * 2);
        M360_TRY(p_linear_first(h, mode, feat, S, m->nerf_w[0], m->nerf_b[0], hn, m->in_pad, first, pair, st));
        if (chain) {
            const void *cw[6];
            const float *cb[6];
            for (int layer = 1; layer < 7; ++layer) { cw[layer - 1] = m->nerf_w[layer]; cb[layer - 1] = m->nerf_b[layer]; }
            {
                ProfScope ps(h, st, M360_K_LINEAR_BF16, Mc, hn, (mode == 2 ? 18 : 6) * hn);  // k_pad = 6 hn (bf16x3: 6 x 3 hn): six layers in one record (the chain kernel alone)
                M360_TRY(ps.done(mlp_chain_bf16_launch(first, a, b, Mc, ldl, cw, cb, 6, hn, ws + L.chain, st, h, mode == 2)));
            }
            M360_TRY(mlp_chain_bf16_rerun(first, a, b, Mc, ldl, cw, cb, 6, hn, ws + L.chain, st, h, mode == 2));
        }
        const long r0 = chain ? Mc : 0;  // rows [r0, S) layer by layer; both parts end in `a`: first -> b -> a -> b -> a -> b -> a
        if (r0 < S) {
            const size_t ro = (size_t)r0 * ldl * 2;
            float *ts = reinterpret_cast<float *>(reinterpret_cast<char *>(first) + ro);
            for (int layer = 1; layer < 7; ++layer) {
                float *td = reinterpret_cast<float *>(reinterpret_cast<char *>((layer & 1) ? b : a) + ro);
                M360_TRY(p_linear_bf16(h, mode, ts, S - r0, m->nerf_w[layer], m->nerf_b[layer], hn, hn, M360_ACT_RELU, td, pair, st));
                ts = td;
            }
        }
        M360_TRY(p_linear_heads(h, mode, src, S, ldl, m->nerf_w[7], m->nerf_b[7], hn, hn, dst, ldl, 0, m->nerf_head_w, 4, hpart, st, pair));
        }
        M360_TRY(p_nerf_finish_fused(h, dst, mode, ldl, hpart, m360_linear_heads_fused_rows(S, hn, mode), m360_linear_heads_slots_bf16(hn, hn, mode, 0), m->nerf_head_w, m->nerf_head_b, hn, t1, r, B, N, out, st, flags));
    } else {
    M360_TRY(p_encode(h, t1, r, vdenc, vd_ch, B, N, feat, m->in_pad, 0, ext_norm ? 0 : h->norm_group_rays, ext_norm, 0, nullptr, ws + L.norm, m360_contract_workspace_bytes(), st));
    M360_TRY(p_linear(h, &tq, feat, S, m->in_pad, m->nerf_w[0], m->nerf_b[0], hn, m->in_pad, M360_ACT_RELU, a, hn, st));
    for (int layer = 1; layer < 7; ++layer) {
        M360_TRY(p_linear(h, &tq, src, S, hn, m->nerf_w[layer], m->nerf_b[layer], hn, hn, M360_ACT_RELU, dst, hn, st));
        float *tmp = src; src = dst; dst = tmp;
    }
    // last hidden layer + the 4 heads fused: its 2.15 GB output never goes to HBM (store_y = 0; ragged tail rows excepted)
    M360_TRY(p_linear_heads(h, 0, src, S, hn, m->nerf_w[7], m->nerf_b[7], hn, hn, dst, hn, 0, m->nerf_head_w, 4, hpart, st));
    M360_TRY(p_nerf_finish_fused(h, dst, 0, hn, hpart, m360_linear_heads_fused_rows(S, hn, 0), slots, m->nerf_head_w, m->nerf_head_b, hn, t1, r, B, N, out, st));
    }
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

int m360_side_create(m360_side_t **out) {
    if (!out) return fail(M360_ERR_INVALID_ARGUMENT, "m360_side_create: null out pointer");
    *out = nullptr;
    m360_side *sd = new m360_side();
    if (hipGetDevice(&sd->device) != hipSuccess || hipStreamCreateWithFlags(&sd->s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sd->join, hipEventDisableTiming) != hipSuccess) {
        const int rc = fail(M360_ERR_LAUNCH, "m360_side_create: %s", hipGetErrorString(hipGetLastError()));
        m360_side_destroy(sd);
        return rc;
    }
    *out = sd;
    return M360_OK;
}

void m360_side_destroy(m360_side_t *sd) {
    if (!sd) return;
    if (sd->s) { (void)hipStreamSynchronize(sd->s); (void)hipStreamDestroy(sd->s); }
    if (sd->fork) (void)hipEventDestroy(sd->fork);
    if (sd->join) (void)hipEventDestroy(sd->join);
    delete sd;
}

m360_prof_t *m360_prof_create(int capacity) {
    if (capacity <= 0) return nullptr;
    m360_prof *p = new m360_prof();
    p->recs.resize((size_t)capacity);
    for (size_t i = 0; i < p->recs.size(); ++i) {
        if (hipEventCreate(&p->recs[i].start) != hipSuccess || hipEventCreate(&p->recs[i].stop) != hipSuccess) {
            (void)hipGetLastError();
            p->recs.resize(i);  // events [0, i) exist (a half-made pair leaks one event at worst)
            m360_prof_destroy(p);
            fail(M360_ERR_LAUNCH, "m360_prof_create: hipEventCreate failed");
            return nullptr;
        }
    }
    return p;
}

void m360_prof_destroy(m360_prof_t *p) {
    if (!p) return;
    for (m360_prof::Rec &r : p->recs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    delete p;
}

int m360_prof_count(const m360_prof_t *p) { return p ? (int)p->used : 0; }

int m360_prof_reset(m360_prof_t *p) {
    if (p) p->used = 0;
    return M360_OK;
}

int m360_prof_read(m360_prof_t *p, int i, float *ms, int *kind, long *M, int *n_pad, int *k_pad) {
    if (!p || i < 0 || (size_t)i >= p->used || !ms) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prof_read: bad index %d", i);
    m360_prof::Rec &r = p->recs[i];
    if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(ms, r.start, r.stop) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_prof_read: event query failed");
    if (kind) *kind = r.kind;
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
    M360_TRY(resample_t_any(t_hat, w_hat, u_rand, B, hyper->num_samples, n_fine(hyper) + 1, hyper->resample_padding, t1, rng_cdf(hyper, u_rand), stream));
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
    bool fused = false;
    M360_TRY(prop_stage(rays, model, hyper, B, nullptr, t0, what, t1, ws, stream, nullptr, nullptr, &fused));
    return nerf_stage(rays, model, hyper, B, t1, out, ws, stream, nullptr, nullptr, fused);
}

/* ------------------------------------------------------------------ training path (row f3) */

size_t m360_train_tape_bytes(int B, int N, const m360_model_t *model_host, int stage) {
    if (!model_host || B < 0 || N < 1 || (stage != 0 && stage != 1)) return 0;
    return tape_for(B, N, model_host, stage).total;
}

struct BwdLayout {
    size_t dz_a, dz_b, gemm, finish, feat_wide, first, total;
    int first_k;  // bf16 mode: columns of the first layer's operand as the weight-gradient kernel sees it (see mlp_backward_bf16)
};
// bf16 mode, first layer: the features are [hi | lo] pair rows of 2 in_pad columns.  The MFMA weight-gradient kernel takes contractions over
// 256-column tiles, so the rows are handed over as overlapping 256-column rows (linear_wgrad_bf16_rows; widths it does not take go through the
// fp32 kernel as they are); either way the result [n_pad, first_k] is folded into grad_w[0][n][k] = r[n][k] + r[n][in_pad + k]
static int bf16_first_k(const m360_model_t *m, int width) { return (width % 256 == 0 && 2 * m->in_pad <= 256) ? 256 : 2 * m->in_pad; }
static BwdLayout bwd_layout_for(int B, int N, const m360_model_t *m, int stage) {
    BwdLayout L;
    const size_t S = (size_t)B * N;
    const int width = stage == 0 ? m->hp_pad : m->hn_pad;
    const int kmax = width > m->in_pad ? width : m->in_pad;
    const bool b16 = m->mlp_bf16 == 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    (void)take(128);  // a caller may hand in the buffer it uses as forward workspace: its status block (offset 0) is left alone
    L.first_k = b16 ? bf16_first_k(m, width) : 0;
    L.dz_a = take(S * width * (b16 ? 2 : sizeof(float)));
    L.dz_b = take(S * width * (b16 ? 2 : sizeof(float)));
    if (b16) {
        const size_t g1 = m360_linear_wgrad_bf16_workspace_bytes((long)S, width, width), g0 = m360_linear_wgrad_bf16_workspace_bytes((long)S, width, L.first_k);
        L.gemm = take(g1 > g0 ? g1 : g0);
    } else {
        L.gemm = take(m360_linear_wgrad_workspace_bytes((long)S, width, kmax));
    }
    L.finish = take(m360_finish_backward_workspace_bytes(B, stage == 0 ? 1 : 4, width));
    L.feat_wide = take(0);  // (until round 6: a zero-padded [S, first_k] copy of the feature rows; the offset still ends the finishers' scratch)
    L.first = take(b16 ? (size_t)width * L.first_k * sizeof(float) : 0);
    L.total = off;
    return L;
}

size_t m360_backward_workspace_bytes(int B, int N, const m360_model_t *model_host, int stage) {
    if (!model_host || B < 0 || N < 1 || (stage != 0 && stage != 1)) return 0;
    return bwd_layout_for(B, N, model_host, stage).total;
}

static int validate_train(const m360_model_t *m, const void *tape, size_t tape_bytes, size_t need, const char *who) {
    if (m->mlp_bf16 == 2) return fail(M360_ERR_INVALID_ARGUMENT, "%s: the training path takes mlp_bf16 = 0 (fp32) or 1 (bf16); the bf16x3 mode is forward-only", who);
    if (!tape || tape_bytes < need || ((uintptr_t)tape & 255)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "%s: tape %zu < required %zu bytes (or not 256-byte aligned)", who, tape_bytes, need);
    return M360_OK;
}

int m360_prop_forward_train(const m360_rays_t *rays, const m360_model_t *model, const m360_hyper_t *hyper, int B,
                            const float *t_rand, float *t_hat, float *w_hat, void *tape, size_t tape_bytes,
                            void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    M360_TRY(validate(rays, model, hyper, B, workspace, workspace_bytes, "m360_prop_forward_train"));
    if (B == 0) return M360_OK;
    if (!t_hat || !w_hat) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_forward_train: t_hat and w_hat are required");
    M360_TRY(validate_train(model, tape, tape_bytes, tape_for(B, hyper->num_samples, model, 0).total, "m360_prop_forward_train"));
    return prop_stage(rays, model, hyper, B, t_rand, t_hat, w_hat, nullptr, static_cast<char *>(workspace), stream, static_cast<char *>(tape));
}

int m360_nerf_forward_train(const m360_rays_t *rays, const m360_model_t *model, const m360_hyper_t *hyper, int B,
                            const float *t_hat, const float *w_hat, const float *u_rand, const m360_outputs_t *out,
                            void *tape, size_t tape_bytes, void *workspace, size_t workspace_bytes,
                            m360_stream_t stream) {
    M360_TRY(validate(rays, model, hyper, B, workspace, workspace_bytes, "m360_nerf_forward_train"));
    if (B == 0) return M360_OK;
    if (!t_hat || !w_hat || !out || !out->rgb || !out->distance || !out->acc)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_forward_train: t_hat, w_hat, out.rgb/distance/acc are required");
    const TapeLayout T = tape_for(B, n_fine(hyper), model, 1);
    M360_TRY(validate_train(model, tape, tape_bytes, T.total, "m360_nerf_forward_train"));
    float *t1 = reinterpret_cast<float *>(static_cast<char *>(tape) + T.t);
    M360_TRY(resample_t_any(t_hat, w_hat, u_rand, B, hyper->num_samples, n_fine(hyper) + 1, hyper->resample_padding, t1, rng_cdf(hyper, u_rand), stream));
    return nerf_stage(rays, model, hyper, B, t1, out, static_cast<char *>(workspace), stream, static_cast<char *>(tape));
}

// dz (gradient at the pre-activation of the last hidden layer, already in `dz`) -> all weight / bias gradients
static int mlp_backward(const m360_hyper_t *h, int layers, const float *const *w_t, float *const *grad_w, float *const *grad_b, const float *feat,
                        int in_pad, float *const *act, int width, long S, float *dz, float *dz_other, void *gemm_ws,
                        size_t gemm_ws_bytes, m360_stream_t st, const char *who) {
    for (int l = layers - 1; l >= 0; --l) {
        const float *x = l == 0 ? feat : act[l - 1];
        const int k = l == 0 ? in_pad : width;
        if (!grad_w[l] || !grad_b[l]) return fail(M360_ERR_INVALID_ARGUMENT, "%s: gradient buffer of layer %d is null", who, l);
        M360_PROF(h, st, M360_K_WGRAD, S, width, k, m360_linear_wgrad(dz, width, x, k, S, width, k, grad_w[l], grad_b[l], gemm_ws, gemm_ws_bytes, st));
        if (l > 0) {
            if (!w_t[l]) return fail(M360_ERR_INVALID_ARGUMENT, "%s: transposed weight of layer %d is null", who, l);
            M360_PROF(h, st, M360_K_DGRAD, S, width, width, m360_linear_dgrad(dz, S, width, w_t[l], width, width, act[l - 1], dz_other, width, st));
            float *tmp = dz; dz = dz_other; dz_other = tmp;
        }
    }
    return M360_OK;
}

// bf16 mode: grad_w[0] = fold of dZ^T [feat_hi | feat_lo]
__global__ void fold_first_layer_kernel(const float *__restrict__ r, int n_pad, int ld, int in_pad, float *__restrict__ gw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pad * in_pad) return;
    const int n = idx / in_pad, k = idx % in_pad;
    gw[idx] = r[(long)n * ld + k] + r[(long)n * ld + in_pad + k];
}
// The same chain in the bf16 mode (round 5): dz, the stored activations and the transposed weights are bf16, every product accumulates in
// fp32 on the bf16 matrix pipe, the gradients come out in fp32 in the packed [n_pad, k_pad] layouts of the fp32 path (layer 0: [n_pad, in_pad]).
// Round 5: the ReLU mask of a layer's input gradient (an HBM pass over [S, width] bf16: 0.52 ms at 524 288 x 1024) runs on a SECOND stream
// beside the layer's weight gradient (MFMA-bound, 1.0 ms, reads dz and the stored activations - not dx): input-gradient GEMM, then fork -
// {mask | weight gradient} - join.  The second stream is the CALLER's (m360_hyper_t.side, a m360_side_t); without one: one stream, the mask behind
// its GEMM (same weight gradients bit for bit).
// The mask runs THROTTLED there - one striding workgroup per CU, two 16-byte pieces per thread in flight: at full rate (6 TB/s) it stretched the
// weight gradient beside it from 1.04 to 1.45 ms and the pair gained 0.1 ms; 4096 x 128, NeRF backward + forward, same box, ms:
//   one stream 26.5 | mask workgroups: all 25.4, 4096 25.7, 1024 25.6, 512 24.5, 384 26.3, 320 26.1, 256 23.7-23.9, 192 24.9, 128 27.5, 64 36.9
// (four pieces in flight: no better; profiles/r05/backward_overlap_ab.jsonl)
#ifdef M360_DIAG  // diagnostics build: M360_MASK_BLOCKS re-tunes the throttle (A/B runs: tools/train_step_bench.py with M360_LIB=libm360_diag.so)
static const int kBackwardMaskBlocksWide = getenv("M360_MASK_BLOCKS") ? atoi(getenv("M360_MASK_BLOCKS")) : 512;
#else
// one striding workgroup of the throttled mask kernel per CU.  Re-swept in round 6 beside the faster weight gradient
// (profiles/r06/backward_mask_blocks_sweep_after_new_wgrad.txt: NeRF / proposal update in ms - 256: 23.4 / 9.4, 320: 25.4 / 9.6, 384: 25.0 / 9.8,
// 448: 24.3 / 9.9, 512: 22.7 / 10.0, 640: 23.4 / 10.1): counts that are not a multiple of the CU count put two mask workgroups on some CUs
// and slow those CUs' weight-gradient tiles; 512 gains on the 1024-wide layers what it loses on the 256-wide ones.  The pair is bound by HBM:
// mask 3.2 GB + weight gradient 2.15 GB in ~0.9 ms is the 6.3 TB/s a copy reaches on this chip.
constexpr int kBackwardMaskBlocksWide = 512;
#endif
// ... hence by width: two mask workgroups per CU beside the one-wave weight gradient of the 1024-wide layers (whose 0.85 ms the one-per-CU mask
// outlasted: 1.2 ms), one per CU beside the 8-wave kernel of the narrow ones
static inline int backward_mask_blocks(int width) { return width >= 1024 ? kBackwardMaskBlocksWide : 256; }
static int mlp_backward_bf16(const m360_hyper_t *h, int layers, const float *const *w_t, float *const *grad_w, float *const *grad_b, const void *feat,
                             int in_pad, void *const *act, int width, long S, void *dz, void *dz_other, char *ws, const BwdLayout &L,
                             m360_stream_t st, const char *who) {
    void *gemm_ws = ws + L.gemm;
    const size_t gemm_bytes = L.finish - L.gemm;
    m360_side *ss = (h && h->side && S >= 32768 && !h->prof) ? static_cast<m360_side *>(h->side) : nullptr;  // (a recorder brackets launches of ONE stream)
    if (ss) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != ss->device)
            return fail(M360_ERR_INVALID_ARGUMENT, "%s: m360_hyper_t.side was created for device %d, the call runs on device %d", who, ss->device, dev);
    }
    // The masked rows the second stream writes ARE the next layer's dz: their column sums - that layer's bias gradient - come out of the mask kernel
    // (it meets every element anyway and runs beside, not in front of, the matrix work), where the weight-gradient kernel pays ~0.1 ms of its 1.0 for
    // them.  Its partial sums (2 MB) live in the finishers' backward scratch (read for the last time before this function runs).
    float *mask_part = reinterpret_cast<float *>(ws + L.finish);
    const int kBackwardMaskBlocks = backward_mask_blocks(width);
    const bool mask_sums = ss && relu_mask_bf16_sums_ok(S, width, kBackwardMaskBlocks) && relu_mask_bf16_sums_bytes(kBackwardMaskBlocks) <= L.feat_wide - L.finish;
    const unsigned tuning = h ? h->tuning : 0u;
    bool have_bias = false;  // grad_b[l] already written (by the mask that produced this layer's dz)
    for (int l = layers - 1; l >= 0; --l) {
        if (!grad_w[l] || !grad_b[l]) return fail(M360_ERR_INVALID_ARGUMENT, "%s: gradient buffer of layer %d is null", who, l);
        float *gb = have_bias ? nullptr : grad_b[l];
        have_bias = false;
        if (l > 0 && ss) {
            if (!w_t[l]) return fail(M360_ERR_INVALID_ARGUMENT, "%s: transposed weight of layer %d is null", who, l);
            hipStream_t hs = reinterpret_cast<hipStream_t>(st);
            m360_stream_t s2 = reinterpret_cast<m360_stream_t>(ss->s);
            M360_TRY(m360_linear_dgrad_bf16(dz, S, width, w_t[l], width, width, nullptr, dz_other, width, st));
            if (hipEventRecord(ss->fork, hs) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) return fail(M360_ERR_LAUNCH, "%s: fork to the second stream failed: %s", who, hipGetErrorString(hipGetLastError()));
            // From here to the join the second stream may hold work that writes dz_other / mask_part / grad_b[l - 1]: WHATEVER fails in between, the
            // caller's stream is ordered behind that work before this call returns (an early return used to leave it unordered, and the
            // caller free to recycle those buffers under it).
            int rc = relu_mask_bf16(dz_other, act[l - 1], S, width, width, s2, kBackwardMaskBlocks, mask_sums ? mask_part : nullptr);
            if (rc == M360_OK && mask_sums) {
                rc = relu_mask_bf16_sums_reduce(mask_part, kBackwardMaskBlocks, width, grad_b[l - 1], s2);
                have_bias = true;
            }
            if (rc == M360_OK) rc = m360_linear_wgrad_bf16(dz, width, act[l - 1], width, S, width, width, grad_w[l], gb, gemm_ws, gemm_bytes, tuning, st);
            if (hipEventRecord(ss->join, ss->s) != hipSuccess || hipStreamWaitEvent(hs, ss->join, 0) != hipSuccess) {
                (void)hipStreamSynchronize(ss->s);  // no event to order the streams by: wait the second one out on the host
                if (rc == M360_OK) rc = fail(M360_ERR_LAUNCH, "%s: join of the second stream failed: %s", who, hipGetErrorString(hipGetLastError()));
            }
            if (rc != M360_OK) return rc;
            void *tmp = dz; dz = dz_other; dz_other = tmp;
        } else if (l > 0) {
            M360_PROF(h, st, M360_K_WGRAD, S, width, -width, m360_linear_wgrad_bf16(dz, width, act[l - 1], width, S, width, width, grad_w[l], gb, gemm_ws, gemm_bytes, tuning, st));
            if (!w_t[l]) return fail(M360_ERR_INVALID_ARGUMENT, "%s: transposed weight of layer %d is null", who, l);
            M360_PROF(h, st, M360_K_DGRAD, S, width, -width, m360_linear_dgrad_bf16(dz, S, width, w_t[l], width, width, act[l - 1], dz_other, width, st));
            void *tmp = dz; dz = dz_other; dz_other = tmp;
        } else {
            hipStream_t hs = reinterpret_cast<hipStream_t>(st);
            // The [hi | lo] feature rows are 2 in_pad = 128 bf16 wide; the MFMA kernels tile the contraction in 256 columns.  first_k = 256: the rows are
            // handed over as they lie, OVERLAPPING (linear_wgrad_bf16_rows: columns 128 .. 255 of a row are the next row's features; behind the last
            // row lies the tape's first layer output) - the fold below reads columns < 2 in_pad of the result only.  (Until round 6: a zero-padded
            // [S, 256] copy per backward, a 268 MB memset + a strided copy.)
            float *r = reinterpret_cast<float *>(ws + L.first);
            M360_PROF(h, st, M360_K_WGRAD, S, width, -L.first_k, linear_wgrad_bf16_rows(dz, width, feat, 2 * in_pad, S, width, L.first_k, r, gb, gemm_ws, gemm_bytes, tuning, st, L.first_k != 2 * in_pad));
            hipLaunchKernelGGL(fold_first_layer_kernel, dim3((unsigned)((width * in_pad + 255) / 256)), dim3(256), 0, hs, r, width, L.first_k, in_pad, grad_w[0]);
            M360_TRY(check_launch(who));
        }
    }
    return M360_OK;
}

int m360_prop_backward(const m360_rays_t *rays, const m360_model_t *model, const m360_mlp_transposed_t *wt,
                       const m360_hyper_t *hyper, int B, const void *tape, size_t tape_bytes,
                       const float *grad_w_hat, const m360_mlp_grads_t *grads, void *workspace,
                       size_t workspace_bytes, m360_stream_t stream) {
    if (!rays || !model || !wt || !hyper || !grads || B < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_backward: null descriptor");
    if (B == 0) return M360_OK;
    const int N = hyper->num_samples;
    const TapeLayout T = tape_for(B, N, model, 0);
    M360_TRY(validate_train(model, tape, tape_bytes, T.total, "m360_prop_backward"));
    const BwdLayout L = bwd_layout_for(B, N, model, 0);
    if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_prop_backward: workspace %zu < required %zu bytes", workspace_bytes, L.total);
    if (!grad_w_hat || !rays->directions || !grads->head_w || !grads->head_b) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_backward: grad_w_hat, rays.directions and the head gradient buffers are required");
    char *tp = const_cast<char *>(static_cast<const char *>(tape)), *ws = static_cast<char *>(workspace);
    const int hp = model->hp_pad;
    float *act[4];
    for (int l = 0; l < 4; ++l) act[l] = reinterpret_cast<float *>(tp + T.act[l]);
    float *dz = reinterpret_cast<float *>(ws + L.dz_a), *dz2 = reinterpret_cast<float *>(ws + L.dz_b);
    if (model->mlp_bf16 == 1) {
        void *actb[4];
        for (int l = 0; l < 4; ++l) actb[l] = tp + T.act[l];
        M360_TRY(prop_finish_backward_stage(actb[3], 1, hp, model->prop_head_w, model->prop_head_b, hp, hyper->density_bias, reinterpret_cast<const float *>(tp + T.t), rays->directions, B, N, grad_w_hat, dz, grads->head_w, grads->head_b, ws + L.finish, L.feat_wide - L.finish, reinterpret_cast<const float *>(tp + T.raw), stream));
        return mlp_backward_bf16(hyper, 4, wt->w_t, grads->w, grads->b, tp + T.feat, model->in_pad, actb, hp, (long)B * N, dz, dz2, ws, L, stream, "m360_prop_backward");
    }
    M360_TRY(prop_finish_backward_stage(act[3], 0, hp, model->prop_head_w, model->prop_head_b, hp, hyper->density_bias, reinterpret_cast<const float *>(tp + T.t), rays->directions, B, N, grad_w_hat, dz, grads->head_w, grads->head_b, ws + L.finish, L.total - L.finish, reinterpret_cast<const float *>(tp + T.raw), stream));
    return mlp_backward(hyper, 4, wt->w_t, grads->w, grads->b, reinterpret_cast<const float *>(tp + T.feat), model->in_pad, act, hp, (long)B * N, dz, dz2, ws + L.gemm, L.finish - L.gemm, stream, "m360_prop_backward");
}

int m360_nerf_backward(const m360_rays_t *rays, const m360_model_t *model, const m360_mlp_transposed_t *wt,
                       const m360_hyper_t *hyper, int B, const void *tape, size_t tape_bytes, const float *grad_rgb,
                       const float *grad_distance, const float *grad_acc, const float *grad_weights,
                       const m360_mlp_grads_t *grads, void *workspace, size_t workspace_bytes,
                       m360_stream_t stream) {
    if (!rays || !model || !wt || !hyper || !grads || B < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_backward: null descriptor");
    if (B == 0) return M360_OK;
    const int N = n_fine(hyper);
    const TapeLayout T = tape_for(B, N, model, 1);
    M360_TRY(validate_train(model, tape, tape_bytes, T.total, "m360_nerf_backward"));
    const BwdLayout L = bwd_layout_for(B, N, model, 1);
    if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_nerf_backward: workspace %zu < required %zu bytes", workspace_bytes, L.total);
    if (!rays->directions || !grads->head_w || !grads->head_b) return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_backward: rays.directions and the head gradient buffers are required");
    char *tp = const_cast<char *>(static_cast<const char *>(tape)), *ws = static_cast<char *>(workspace);
    const int hn = model->hn_pad;
    float *act[8];
    for (int l = 0; l < 8; ++l) act[l] = reinterpret_cast<float *>(tp + T.act[l]);
    float *dz = reinterpret_cast<float *>(ws + L.dz_a), *dz2 = reinterpret_cast<float *>(ws + L.dz_b);
    if (model->mlp_bf16 == 1) {
        void *actb[8];
        for (int l = 0; l < 8; ++l) actb[l] = tp + T.act[l];
        M360_TRY(nerf_finish_backward_stage(actb[7], 1, hn, model->nerf_head_w, model->nerf_head_b, hn, hyper->density_bias, hyper->rgb_padding, reinterpret_cast<const float *>(tp + T.t), rays->directions, B, N, hyper->white_bkgd, grad_rgb, grad_distance, grad_acc, grad_weights, dz, grads->head_w, grads->head_b, ws + L.finish, L.feat_wide - L.finish, reinterpret_cast<const float *>(tp + T.raw), stream));
        return mlp_backward_bf16(hyper, 8, wt->w_t, grads->w, grads->b, tp + T.feat, model->in_pad, actb, hn, (long)B * N, dz, dz2, ws, L, stream, "m360_nerf_backward");
    }
    M360_TRY(nerf_finish_backward_stage(act[7], 0, hn, model->nerf_head_w, model->nerf_head_b, hn, hyper->density_bias, hyper->rgb_padding, reinterpret_cast<const float *>(tp + T.t), rays->directions, B, N, hyper->white_bkgd, grad_rgb, grad_distance, grad_acc, grad_weights, dz, grads->head_w, grads->head_b, ws + L.finish, L.total - L.finish, reinterpret_cast<const float *>(tp + T.raw), stream));
    return mlp_backward(hyper, 8, wt->w_t, grads->w, grads->b, reinterpret_cast<const float *>(tp + T.feat), model->in_pad, act, hn, (long)B * N, dz, dz2, ws + L.gemm, L.finish - L.gemm, stream, "m360_nerf_backward");
}

/* ------------------------------------------------------------------ one batch sharded over devices */

int m360_prop_forward_from_t(const m360_rays_t *rays, const m360_model_t *model, const m360_hyper_t *hyper, int B,
                             const float *t_hat, const float *norm, float *w_hat, float *t_new, void *workspace,
                             size_t workspace_bytes, m360_stream_t stream) {
    M360_TRY(validate(rays, model, hyper, B, workspace, workspace_bytes, "m360_prop_forward_from_t"));
    if (B == 0) return M360_OK;
    if (!t_hat || !norm || !w_hat) return fail(M360_ERR_INVALID_ARGUMENT, "m360_prop_forward_from_t: t_hat, norm and w_hat are required");
    return prop_stage(rays, model, hyper, B, nullptr, const_cast<float *>(t_hat), w_hat, t_new, static_cast<char *>(workspace), stream, nullptr, norm);
}

int m360_nerf_forward_from_t(const m360_rays_t *rays, const m360_model_t *model, const m360_hyper_t *hyper, int B,
                             const float *t_new, const float *norm, const m360_outputs_t *out, void *workspace,
                             size_t workspace_bytes, m360_stream_t stream) {
    M360_TRY(validate(rays, model, hyper, B, workspace, workspace_bytes, "m360_nerf_forward_from_t"));
    if (B == 0) return M360_OK;
    if (!t_new || !norm || !out || !out->rgb || !out->distance || !out->acc) return fail(M360_ERR_INVALID_ARGUMENT, "m360_nerf_forward_from_t: t_new, norm, out.rgb/distance/acc are required");
    return nerf_stage(rays, model, hyper, B, t_new, out, static_cast<char *>(workspace), stream, nullptr, norm);
}

}  // extern "C"
