// Developer tool (not part of the product): where one wave of the screened PQ kernel spends its time.
// Builds the kernel file with EQUSS_PQ_STAMPS, runs the benchmark shape and prints the phase durations.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pq_stamps pq_stamps.hip
#define EQUSS_PQ_STAMPS 1
#include "../../expand-and-quantize-for-unsupervised-semantic-segmentation_amd/csrc/pq_assign.hip"
#include "../../expand-and-quantize-for-unsupervised-semantic-segmentation_amd/csrc/capi.hip"
#include <vector>
#include <random>

int main() {
    const int rows = 64 * 1600, num_pq = 8, K = 256, d = 128;
    std::vector<float> code((size_t)rows * num_pq * d), books((size_t)num_pq * K * d);
    std::mt19937 gen(1);
    std::normal_distribution<float> nd;
    std::uniform_real_distribution<float> ud(-0.5f, 0.5f);
    for (auto& v : code) v = nd(gen);
    for (auto& v : books) v = ud(gen);
    float *dcode, *dbooks, *dzq, *dloss; int64_t* didx; void* ws;
    hipMalloc(&dcode, code.size() * 4); hipMalloc(&dzq, code.size() * 4); hipMalloc(&dbooks, books.size() * 4);
    hipMalloc(&dloss, 64); hipMalloc(&didx, (size_t)rows * num_pq * 8);
    const size_t wsb = equss_pq_workspace_bytes(rows, num_pq, K, d);
    hipMalloc(&ws, wsb);
    hipMemcpy(dcode, code.data(), code.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbooks, books.data(), books.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        int rc = equss_pq_assign(dcode, num_pq * d, dbooks, didx, dzq, num_pq * d, dloss, ws, wsb, rows, num_pq, K, d, 1, 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rc %d, %.1f us\n", rc, ms * 1e3);
    }
    unsigned long long st[64 * 8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pq_stamps), sizeof(st));
    const char* names[6] = {"normalise+lds", "pass1", "pass2", "survivors", "rescoring", "gather+store"};
    double sum[7] = {0};
    int n = 0;
    for (int t = 1; t < 20; ++t) {
        if (!st[t * 8 + 6]) break;
        printf("tile %2d:", t);
        for (int p = 0; p < 6; ++p) { double us = (st[t * 8 + p + 1] - st[t * 8 + p]) * 0.01; sum[p] += us; printf(" %s %.2f", names[p], us); }
        const double whole = (st[t * 8 + 6] - st[(t - 1) * 8 + 6]) * 0.01;
        sum[6] += whole; ++n;
        printf("  | tile-to-tile %.2f us\n", whole);
    }
    printf("mean:");
    for (int p = 0; p < 6; ++p) printf(" %s %.2f", names[p], sum[p] / n);
    printf(" | tile %.2f us (%d tiles)\n", sum[6] / n, n);
    return 0;
}
