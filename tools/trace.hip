// tools/trace.hip — developer tool: per-wave phase timeline of step_kernel (needs a GYMRS_TRACE_TIMES build).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "gymrs_amd.h"
extern "C" gymrs_status gymrs_dev_set_trace(gymrs_engine* e, unsigned long long* buf);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv)
{
    const size_t n = (size_t)1 << (argc > 3 ? atoi(argv[3]) : 20);
    const uint32_t flags = argc > 1 ? atoi(argv[1]) : 3;
    // steps of the traced call: >= 8 go through a chain (unless GYMRS_AQL=0); the stamps of the LAST step survive.  13 after the 200 warm-up
    // steps ends on tick 212: not a folding launch (those are ticks = 7 mod 8)
    const uint32_t traced = argc > 2 ? atoi(argv[2]) : 13;
    gymrs_engine* e;
    if (gymrs_engine_create(GYMRS_CARTPOLE, n, 0, 0, nullptr, flags, &e)) { printf("%s\n", gymrs_last_error()); return 1; }
    unsigned char* act; CK(hipMalloc(&act, n * 8));
    for (int b = 0; b < 8; ++b) gymrs_fill_actions(e, act + (size_t)b * n, 1, b);
    gymrs_reset(e, 1, 0, nullptr, nullptr);
    gymrs_step_many(e, act, n, 8, 200, 0);
    gymrs_sync(e);
    const size_t waves = n / 256;
    unsigned long long* tr; CK(hipMalloc(&tr, waves * 8 * 8)); CK(hipMemset(tr, 0, waves * 8 * 8));
    gymrs_dev_set_trace(e, tr);
    gymrs_step_many(e, act, n, 8, traced, 0);
    gymrs_sync(e);
    std::vector<unsigned long long> h(waves * 8);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < waves; ++w) t0 = std::min(t0, h[w * 8]);
    const char* names[7] = {"start", "loads issued", "loads landed", "physics done", "reset done", "stores issued", "end (stores acked)"};
    printf("per-wave s_memtime stamps relative to the earliest wave start (s_memtime ticks = shader-clock cycles, ~2.4 GHz; start stamps of different XCDs are not comparable, phase durations are)\n");
    for (int s = 0; s < 7; ++s) {
        std::vector<long long> v(waves);
        for (size_t w = 0; w < waves; ++w) v[w] = (long long)(h[w * 8 + s] - t0);
        std::sort(v.begin(), v.end());
        printf("%-20s min %6lld  p10 %6lld  median %6lld  p90 %6lld  max %6lld\n", names[s], v[0], v[waves / 10], v[waves / 2], v[waves * 9 / 10], v[waves - 1]);
    }
    const char* dn[6] = {"issue loads", "wait for loads", "physics", "auto-reset", "issue stores", "store ack"};
    for (int s = 0; s < 6; ++s) {
        std::vector<long long> v(waves);
        for (size_t w = 0; w < waves; ++w) v[w] = (long long)(h[w * 8 + s + 1] - h[w * 8 + s]);
        std::sort(v.begin(), v.end());
        printf("phase %-16s min %6lld  median %6lld  p90 %6lld  max %6lld\n", dn[s], v[0], v[waves / 2], v[waves * 9 / 10], v[waves - 1]);
    }
    if (argc > 4) {
        // timeline (argv[4] = bucket width in cycles): how many waves are in which phase at a time.  s_memtime has one time base per CU (or a few CUs): the
        // start stamps fall into clusters millions of cycles apart.  Every CU receives waves of the first generation within ~1 us of the launch's
        // start, so each cluster is shifted to ITS earliest start and the clusters are added up: a chip-wide timeline good to about +-1 us.
        const long long bucket = atoll(argv[4]);
        std::vector<size_t> order;
        for (size_t w = 0; w < waves; ++w)
            if (h[w * 8]) order.push_back(w);
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return h[x * 8] < h[y * 8]; });
        std::vector<unsigned long long> origin(waves, 0);
        size_t clusters = 0, first = 0;
        for (size_t k = 0; k <= order.size(); ++k) {
            if (k == order.size() || (k > first && h[order[k] * 8] - h[order[k - 1] * 8] > 1000000ull)) {
                for (size_t j = first; j < k; ++j) origin[order[j]] = h[order[first] * 8];
                ++clusters;
                first = k;
            }
        }
        long long span = 0;
        for (size_t w : order) span = std::max(span, (long long)(h[w * 8 + 6] - origin[w]));
        const size_t nb = (size_t)(span / bucket) + 1;
        if (nb > 4000) { printf("timeline: a span of %lld cycles -- clusters of time bases overlap, no timeline\n", span); return 0; }
        std::vector<int> cnt(nb * 4, 0);
        const int lo[3] = {0, 2, 4}, hi[3] = {2, 4, 6}; // waiting for loads | physics + auto-reset | storing
        for (size_t w : order) {
            for (int ph = 0; ph < 3; ++ph) {
                const size_t b0 = (size_t)((h[w * 8 + lo[ph]] - origin[w]) / bucket), b1 = (size_t)((h[w * 8 + hi[ph]] - origin[w]) / bucket);
                for (size_t b = b0; b <= b1 && b < nb; ++b) cnt[b * 4 + ph]++;
            }
            cnt[(size_t)((h[w * 8] - origin[w]) / bucket) * 4 + 3]++;
        }
        printf("timeline of the chip (%zu stamped waves in %zu clusters of time bases), buckets of %lld cycles: waves waiting for their loads | computing | storing | waves that START in the bucket\n",
               order.size(), clusters, bucket);
        for (size_t b = 0; b < nb; ++b) printf("t=%7lld  load %5d  compute %5d  store %5d  starts %5d\n", (long long)(b * bucket), cnt[b * 4], cnt[b * 4 + 1], cnt[b * 4 + 2], cnt[b * 4 + 3]);
    }
    return 0;
}
