// Pure C-ABI use of libm360 (no Python, no torch): pack random weights, run mipNeRF360.forward (model.py:247-252)
// on a synthetic ray batch, dump inputs + outputs to a file so that the Python mirror can be checked against it
// (tests/test_gpu_parity.py::test_c_abi_without_python).
//
//   hipcc -O2 -I include examples/forward_c_abi.cpp -L mipnerf360_amd -lm360 -Wl,-rpath,$PWD/mipnerf360_amd -o /tmp/forward_c_abi
//   /tmp/forward_c_abi out.bin [rays] [samples] [hidden_prop] [hidden_nerf]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "m360.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define M360_CHECK(x)                                                              \
    do {                                                                           \
        if ((x) != M360_OK) {                                                      \
            fprintf(stderr, "%s failed: %s\n", #x, m360_last_error());             \
            return 3;                                                              \
        }                                                                          \
    } while (0)

static uint64_t g_state = 0x9e3779b97f4a7c15ull;
static float uniform(float lo, float hi) {  // splitmix64 -> [lo, hi)
    uint64_t z = (g_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return lo + (hi - lo) * (float)((z >> 40) * (1.0 / 16777216.0));
}

template <typename T>
static T *to_device(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Layer {
    std::vector<float> w, b;
    int n_out, k_in;
};

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s out.bin [rays samples hidden_prop hidden_nerf]\n", argv[0]);
        return 1;
    }
    const int B = argc > 2 ? atoi(argv[2]) : 200, N = argc > 3 ? atoi(argv[3]) : 32;
    const int hp = argc > 4 ? atoi(argv[4]) : 64, hn = argc > 5 ? atoi(argv[5]) : 128;
    const int in_ch = 42 + 16, in_pad = round_up(in_ch, 32), hp_pad = round_up(hp, 32), hn_pad = round_up(hn, 32);
    if (m360_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 4;
    }
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    // ---- weights in the reference's state_dict order (model.py:43-53, 131-158), Kaiming-uniform-like
    std::vector<Layer> layers;
    auto add = [&](int n_out, int k_in) {
        Layer l;
        l.n_out = n_out;
        l.k_in = k_in;
        const float bound = sqrtf(6.0f / k_in);
        l.w.resize((size_t)n_out * k_in);
        l.b.resize(n_out);
        for (float &v : l.w) v = uniform(-bound, bound);
        for (float &v : l.b) v = uniform(-0.1f, 0.1f);
        layers.push_back(l);
    };
    add(hp, in_ch); add(hp, hp); add(hp, hp); add(hp, hp); add(1, hp);                      // prop_net.model.{0,2,4,6,8}
    add(hn, in_ch); for (int i = 0; i < 7; ++i) add(hn, hn); add(1, hn); add(3, hn);        // nerf_net.model.*, final_density, final_color

    m360_model_t model = {};
    model.in_ch = in_ch; model.in_pad = in_pad; model.hp_pad = hp_pad; model.hn_pad = hn_pad; model.mlp_bf16 = 0; model.packed_layout = M360_PACKED_LAYOUT;
    auto pack = [&](const Layer &l, int n_pad, int k_pad, float **wp, float **bp) -> int {
        float *w = to_device(l.w), *b = to_device(l.b);
        if (!w || !b) return 1;
        if (hipMalloc(wp, (size_t)n_pad * k_pad * sizeof(float)) != hipSuccess) return 1;
        if (bp && hipMalloc(bp, (size_t)n_pad * sizeof(float)) != hipSuccess) return 1;
        if (m360_pack_linear(w, bp ? b : nullptr, l.n_out, l.k_in, n_pad, k_pad, *wp, bp ? *bp : nullptr, (m360_stream_t)stream) != M360_OK) return 1;
        return 0;
    };
    float *tmp_w, *tmp_b;
    for (int i = 0; i < 4; ++i) {
        if (pack(layers[i], hp_pad, i == 0 ? in_pad : hp_pad, &tmp_w, &tmp_b)) return 5;
        model.prop_w[i] = tmp_w; model.prop_b[i] = tmp_b;
    }
    if (pack(layers[4], 1, hp_pad, &tmp_w, nullptr)) return 5;
    model.prop_head_w = tmp_w; model.prop_head_b = to_device(layers[4].b);
    for (int i = 0; i < 8; ++i) {
        if (pack(layers[5 + i], hn_pad, i == 0 ? in_pad : hn_pad, &tmp_w, &tmp_b)) return 5;
        model.nerf_w[i] = tmp_w; model.nerf_b[i] = tmp_b;
    }
    {   // heads: [4, hn_pad] = density row, then the 3 colour rows; biases [4]
        Layer h4;
        h4.n_out = 4; h4.k_in = hn;
        h4.w = layers[13].w; h4.w.insert(h4.w.end(), layers[14].w.begin(), layers[14].w.end());
        h4.b = layers[13].b; h4.b.insert(h4.b.end(), layers[14].b.begin(), layers[14].b.end());
        if (pack(h4, 4, hn_pad, &tmp_w, nullptr)) return 5;
        model.nerf_head_w = tmp_w; model.nerf_head_b = to_device(h4.b);
    }

    // ---- rays (NDC-like, near 0 / far 1)
    std::vector<float> o(3 * B), d(3 * B), vd(3 * B), rad(B), nr(B, 0.0f), fr(B, 1.0f);
    for (int b = 0; b < B; ++b) {
        o[3 * b] = uniform(-1, 1); o[3 * b + 1] = uniform(-1, 1); o[3 * b + 2] = -1.0f;
        d[3 * b] = uniform(-0.5f, 0.5f); d[3 * b + 1] = uniform(-0.5f, 0.5f); d[3 * b + 2] = 2.0f;
        float vx = uniform(-0.4f, 0.4f), vy = uniform(-0.4f, 0.4f), vz = -1.0f, inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
        vd[3 * b] = vx * inv; vd[3 * b + 1] = vy * inv; vd[3 * b + 2] = vz * inv;
        rad[b] = uniform(1e-3f, 3e-3f);
    }
    m360_rays_t rays = {to_device(o), to_device(d), to_device(vd), to_device(rad), to_device(nr), to_device(fr)};
    m360_hyper_t hyper = {};
    hyper.num_samples = N; hyper.viewdir_min_deg = 0; hyper.viewdir_max_deg = 4; hyper.white_bkgd = 0;
    hyper.density_bias = -1.0f; hyper.rgb_padding = 0.001f; hyper.resample_padding = 0.01f;

    const size_t ws_bytes = m360_forward_workspace_bytes(B, N, &model);
    void *ws;
    HIP_OK(hipMalloc(&ws, ws_bytes));
    float *rgb, *dist, *acc;
    HIP_OK(hipMalloc(&rgb, 3 * B * sizeof(float)));
    HIP_OK(hipMalloc(&dist, B * sizeof(float)));
    HIP_OK(hipMalloc(&acc, B * sizeof(float)));
    m360_outputs_t out = {};
    out.rgb = rgb; out.distance = dist; out.acc = acc;
    M360_CHECK(m360_forward(&rays, &model, &hyper, B, &out, ws, ws_bytes, (m360_stream_t)stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<float> h_rgb(3 * B), h_dist(B), h_acc(B);
    HIP_OK(hipMemcpy(h_rgb.data(), rgb, h_rgb.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_dist.data(), dist, h_dist.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_acc.data(), acc, h_acc.size() * 4, hipMemcpyDeviceToHost));

    // ---- dump: header (B, N, hp, hn, n_layers) then per layer (n_out, k_in, w, b), rays, outputs
    FILE *f = fopen(argv[1], "wb");
    if (!f) return 6;
    const int32_t hdr[5] = {B, N, hp, hn, (int32_t)layers.size()};
    fwrite(hdr, 4, 5, f);
    for (const Layer &l : layers) {
        const int32_t dims[2] = {l.n_out, l.k_in};
        fwrite(dims, 4, 2, f);
        fwrite(l.w.data(), 4, l.w.size(), f);
        fwrite(l.b.data(), 4, l.b.size(), f);
    }
    const std::vector<float> *arrs[] = {&o, &d, &vd, &rad, &nr, &fr, &h_rgb, &h_dist, &h_acc};
    for (const auto *a : arrs) fwrite(a->data(), 4, a->size(), f);
    fclose(f);
    double s = 0;
    for (float v : h_rgb) s += v;
    printf("forward_c_abi: %d rays x %d samples, mean rgb %.6f, libm360 version %d\n", B, N, s / h_rgb.size(), m360_version());
    return 0;
}
