// m360_pack_many: every packing of a parameter set in ONE launch (include/m360.h).
//
// A training step re-packs both networks on every forward (the parameters changed) and their transposes on every backward: per forward that
// was one 4 us kernel per layer behind a 56 us NaN scan of the same tensors, ~7 us of dependency gap each - 0.1-0.15 ms of small launches in
// front of a 6 ms forward.  Here one kernel walks a table of up to kPackMax items handed over BY VALUE in its kernel arguments; every element
// is formed by the same expressions as in the per-layer kernels (m360_linear.hip: pack_linear_kernel, pack_linear_t_kernel;
// m360_linear_bf16.hip.h: pack_linear_bf16 / _bf16x3 / _bf16x6_kernel; m360_linear_tn_bf16.hip.h: pack_linear_bf16_t_kernel) - the same bits -
// and, reading every source value anyway, raises the NaN flag the bf16 modes refuse parameters by (m360_params_nan_flag).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/m360.h"
#include "m360_common.hip.h"

namespace m360 {
constexpr int kPackMax = 16;
struct pack_entry_t {
    const float *w, *b;
    void *wp;
    float *bp;
    int n_out, k_in, n_pad, k_pad, format;
    unsigned first_block;  // of this item in the launch's grid
};
struct pack_table_t {
    pack_entry_t e[kPackMax];
    int count;
};

__global__ __launch_bounds__(256) void pack_many_kernel(pack_table_t t, unsigned *__restrict__ nan_flag) {
    int it = 0;
#pragma unroll 1
    for (int i = 1; i < t.count; ++i)
        if (blockIdx.x >= t.e[i].first_block) it = i;  // (first_block ascends: the last item at or below this workgroup)
    const pack_entry_t &e = t.e[it];
    const long idx = (long)(blockIdx.x - e.first_block) * 256 + threadIdx.x;
    const int n_out = e.n_out, k_in = e.k_in, n_pad = e.n_pad, k_pad = e.k_pad;
    const float *__restrict__ w = e.w;
    bool bad = false;
    if (idx < (long)n_pad * k_pad) {
        const bool transposed = e.format == M360_PACK_F32_T || e.format == M360_PACK_BF16_T;
        // [n_pad, k_pad] packings: idx = n k_pad + k; transposed [k_pad, n_pad]: idx = k n_pad + n
        const int n = transposed ? (int)(idx % n_pad) : (int)(idx / k_pad), k = transposed ? (int)(idx / n_pad) : (int)(idx % k_pad);
        const bool inside = n < n_out && k < k_in;
        const float raw = inside ? w[(long)n * k_in + k] : 0.0f;
        bad = raw != raw;
        switch (e.format) {
            case M360_PACK_F32: static_cast<float *>(e.wp)[idx] = inside ? canon_nanf_(raw) : 0.0f; break;
            case M360_PACK_BF16: static_cast<__bf16 *>(e.wp)[idx] = (__bf16)(inside ? canon_nanf_(raw) : 0.0f); break;
            case M360_PACK_BF16X3: {
                __bf16 hi, lo;
                split_bf16_(inside ? canon_nanf_(raw) : 0.0f, hi, lo);
                __bf16 *row = static_cast<__bf16 *>(e.wp) + (long)n * 3 * k_pad;
                row[k] = hi;
                row[k_pad + k] = hi;
                row[2 * k_pad + k] = lo;
                break;
            }
            case M360_PACK_BF16X6: {
                __bf16 hi, mid, lo;
                split3_bf16_(inside ? canon_nanf_(raw) : 0.0f, hi, mid, lo);
                __bf16 *row = static_cast<__bf16 *>(e.wp) + (long)n * 6 * k_pad + k;
                row[0] = hi;
                row[k_pad] = mid;
                row[2 * k_pad] = lo;
                row[3 * k_pad] = hi;
                row[4 * k_pad] = mid;
                row[5 * k_pad] = hi;
                break;
            }
            case M360_PACK_F32_T: static_cast<float *>(e.wp)[idx] = raw; break;          // (the transposes keep a NaN's bits, as their kernels do)
            case M360_PACK_BF16_T: static_cast<__bf16 *>(e.wp)[idx] = (__bf16)raw; break;
            default: break;
        }
    }
    if (e.bp != nullptr && idx < n_pad) {
        const float rb = (e.b != nullptr && idx < n_out) ? e.b[idx] : 0.0f;
        bad |= rb != rb;
        e.bp[idx] = canon_nanf_(rb);
    }
    if (nan_flag != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1u);
}
}  // namespace m360
using namespace m360;

extern "C" {
int m360_pack_many(const m360_pack_item_t *items, int count, unsigned *nan_flag, m360_stream_t stream) {
    if (count < 0 || (count > 0 && !items)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: null item list or negative count");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (nan_flag && hipMemsetAsync(nan_flag, 0, sizeof(unsigned), st) != hipSuccess) return fail(M360_ERR_LAUNCH, "m360_pack_many: memset of the flag failed");
    if (count == 0) return M360_OK;  // (the flag, if any, is cleared: nothing read, nothing NaN)
    // every item is checked before the first launch: a refused list packs nothing
    for (int i = 0; i < count; ++i) {
        const m360_pack_item_t &it = items[i];
        const bool t = it.format == M360_PACK_F32_T || it.format == M360_PACK_BF16_T;
        if (it.format < M360_PACK_F32 || it.format > M360_PACK_BF16_T) return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: item %d: format %d", i, it.format);
        if (!it.w || !it.w_packed || it.n_out < 1 || it.k_in < 1 || it.n_pad < it.n_out || it.k_pad < it.k_in)
            return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: item %d: bad argument (n_out=%d k_in=%d n_pad=%d k_pad=%d)", i, it.n_out, it.k_in, it.n_pad, it.k_pad);
        if (t && (it.b || it.b_packed)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: item %d: a transposed packing has no bias", i);
        // the pads the per-layer entry points ask for (m360_pack_linear_bf16: k_pad % 64; _bf16x3: n_pad % 32 too; _bf16_transposed: n_pad % 64)
        if ((it.format == M360_PACK_BF16 || it.format == M360_PACK_BF16X3 || it.format == M360_PACK_BF16X6) && it.k_pad % 64 != 0)
            return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: item %d: k_pad=%d must be a multiple of 64 for the bf16 packings", i, it.k_pad);
        if (it.format == M360_PACK_BF16X3 && it.n_pad % 32 != 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: item %d: n_pad=%d must be a multiple of 32 (bf16x3)", i, it.n_pad);
        if (it.format == M360_PACK_BF16_T && it.n_pad % 64 != 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: item %d: n_pad=%d must be a multiple of 64 (transposed bf16: the input gradient's contraction)", i, it.n_pad);
        if ((long)it.n_pad * it.k_pad > (1l << 37)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: item %d: %d x %d elements", i, it.n_pad, it.k_pad);
    }
    for (int i0 = 0; i0 < count; i0 += kPackMax) {
        pack_table_t t{};
        t.count = count - i0 < kPackMax ? count - i0 : kPackMax;
        unsigned long long blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const m360_pack_item_t &it = items[i0 + i];
            pack_entry_t &e = t.e[i];
            e.w = it.w, e.b = it.b, e.wp = it.w_packed, e.bp = it.b_packed;
            e.n_out = it.n_out, e.k_in = it.k_in, e.n_pad = it.n_pad, e.k_pad = it.k_pad, e.format = it.format;
            e.first_block = (unsigned)blocks;
            blocks += ((unsigned long long)it.n_pad * it.k_pad + 255) / 256;
        }
        if (blocks > 0x7fffffffull) return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_many: %llu workgroups in one launch", blocks);
        hipLaunchKernelGGL(pack_many_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t, nan_flag);
    }
    return check_launch("pack_many");
}
}  // extern "C"
