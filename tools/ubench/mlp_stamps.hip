// Developer tool (not part of the product): where the waves of one workgroup of the fused block-tail kernel
// spend their time.  Builds the kernel file with EQUSS_MLP_STAMPS, runs the benchmark shape (768 tiles) and
// prints the phase durations per wave.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mlp_stamps mlp_stamps.hip
#define EQUSS_MLP_STAMPS 1
#include "../../expand-and-quantize-for-unsupervised-semantic-segmentation_amd/csrc/mlp_fused.hip"
#include "../../expand-and-quantize-for-unsupervised-semantic-segmentation_amd/csrc/capi.hip"
#include <vector>
#include <random>
#include <cstring>

static unsigned short bf16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char**) {
    const int rows = 768 * 128, c = 384, hid = 1536;
    std::mt19937 gen(1);
    std::normal_distribution<float> nd;
    auto dev_bf16 = [&](size_t n, float sc) { std::vector<unsigned short> h(n); for (auto& v : h) v = bf16(nd(gen) * sc); void* d; hipMalloc(&d, n * 2); hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice); return d; };
    auto dev_f32 = [&](size_t n, float sc, float off) { std::vector<float> h(n); for (auto& v : h) v = nd(gen) * sc + off; float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; };
    void* att = dev_bf16((size_t)rows * c, 1.f);
    void* wp = dev_bf16((size_t)c * c, 0.05f); void* w1 = dev_bf16((size_t)hid * c, 0.05f); void* w2 = dev_bf16((size_t)c * hid, 0.025f);
    float* bp = dev_f32(c, 1.f, 0.f); float* b1 = dev_f32(hid, 1.f, 0.f); float* b2 = dev_f32(c, 1.f, 0.f);
    float* g2 = dev_f32(c, 0.f, 1.f); float* be2 = dev_f32(c, 0.f, 0.f); float* ga = dev_f32(c, 0.f, 1.f); float* be = dev_f32(c, 0.f, 0.f);
    float* x = dev_f32((size_t)rows * c, 1.f, 0.f);
    void *mid, *ln; hipMalloc(&mid, (size_t)rows * c * 2); hipMalloc(&ln, (size_t)rows * c * 2);
    char* img = nullptr;   // argument "images": the weights as ring-stage images (what the ViT forward uses)
    if (argc > 1) {
        hipMalloc(&img, equss::tail_weight_image_bytes());
        const void* wps[1] = {wp}; const void* w1s[1] = {w1}; const void* w2s[1] = {w2};
        printf("pack rc %d\n", equss::pack_tail_weights(wps, w1s, w2s, 1, img, nullptr));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        int rc;
        if (img) rc = equss::block_fused(att, c, wp, bp, g2, be2, mid, c, w1, b1, w2, b2, x, c, ga, be, ln, c, rows, 1e-6f, nullptr, img, img + equss::tail_weight_wp_bytes());
        else rc = equss_block_tail_bf16(att, c, wp, bp, g2, be2, mid, c, w1, b1, w2, b2, x, c, ga, be, ln, c, rows, c, hid, 1e-6f, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rc %d, %.1f us\n", rc, ms * 1e3);
    }
    static unsigned long long st[NW * kStampMax];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_mlp_stamps), sizeof(st));
    const int per_tile = 5 + 36 + 1;
    for (int w = 0; w < NW; ++w) {
        const unsigned long long* s = st + w * kStampMax;
        printf("wave %d (times in us since the wave's first stamp)\n", w);
        for (int t = 0; t < 3; ++t) {
            const unsigned long long* q = s + t * per_tile;
            auto us = [&](int a, int b) { return (q[b] - q[a]) * 0.01; };
            double g1 = 0, gelu = 0, g2s = 0;
            for (int ch = 0; ch < 12; ++ch) {
                const int b = 5 + ch * 3;
                g1 += us(b - 1, b); gelu += us(b, b + 1); g2s += us(b + 1, b + 2);
            }
            printf("  tile %d @%7.2f: issue x loads %.2f | wait ring+x %.2f | proj 12 steps %.2f | LN2 write+reissue %.2f | G1 %.2f (per chunk %.2f) | gelu %.2f (%.2f) | G2 %.2f (%.2f) | epilogue %.2f | total %.2f\n",
                   t, (q[0] - s[0]) * 0.01, us(0, 1), us(1, 2), us(2, 3), us(3, 4), g1, g1 / 12, gelu, gelu / 12, g2s, g2s / 12,
                   us(per_tile - 2, per_tile - 1), us(0, per_tile - 1));
        }
    }
    // per-chunk detail of wave 0, tile 1
    {
        const unsigned long long* q = st + per_tile;
        printf("wave 0 tile 1 chunks (G1, gelu, G2):");
        for (int ch = 0; ch < 12; ++ch) { const int b = 5 + ch * 3; printf(" [%.2f %.2f %.2f]", (q[b] - q[b - 1]) * 0.01, (q[b + 1] - q[b]) * 0.01, (q[b + 2] - q[b + 1]) * 0.01); }
        printf("\n");
    }
    return 0;
}
