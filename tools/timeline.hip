// timeline.hip -- where does the time of the two fused kernels go?  (measurement tool, not product code)
// Builds the whole library as one translation unit with -DOCEAN_TIMELINE: lane 0 of every workgroup
// stamps the 100 MHz wall clock at phase boundaries (ocean_device_intrinsics.hpp: OCEAN_TL).  Prints
// per-phase statistics and the number of workgroups inside each phase over time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DOCEAN_TIMELINE -I gfx_ocean_amd/csrc tools/timeline.hip -o tools/timeline
#include "../gfx_ocean_amd/csrc/ocean_api.hip"
#include <algorithm>
#include <chrono>
#include <map>
#include <random>

static void ocean_debug_pass(OceanContext* c, int which, float t) {
    OCEAN_DISPATCH(c->n, {
        if (which == 1) L::pass1(c, t, c->default_domain, c->stream);
        else L::pass2(c, c->stream);
    });
}

static void stats(const char* name, std::vector<double>& v) {
    if (v.empty()) return;
    std::sort(v.begin(), v.end());
    double s = 0; for (double x : v) s += x;
    printf("  %-34s n=%5zu  mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", name, v.size(), s / v.size(),
           v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096;
    OceanContext* ctx = nullptr;
    if (ocean_context_create(0, N, &ctx) != OCEAN_OK) { printf("create failed: %s\n", ocean_last_error(nullptr)); return 1; }
    std::vector<float> h0((size_t)N * N * 2), om((size_t)N * N);
    std::mt19937 rng(N);
    std::normal_distribution<float> nd(0.f, 1e-3f);
    std::uniform_real_distribution<float> ud(0.1f, 4.f);
    for (auto& x : h0) x = nd(rng);
    for (auto& x : om) x = ud(rng);
    if (ocean_upload_spectrum(ctx, h0.data(), om.data()) != OCEAN_OK) { printf("upload failed: %s\n", ocean_last_error(ctx)); return 1; }
    const int maxblocks = 2 * N + 8;
    unsigned long long* d_tl = nullptr;
    hipMalloc(&d_tl, (size_t)maxblocks * 16 * 8 * 2);
    unsigned long long* d_tl2 = d_tl + (size_t)maxblocks * 16;
    hipMemcpyToSymbol(HIP_SYMBOL(ocean_tl), &d_tl, sizeof(d_tl));
    for (int i = 0; i < 5; ++i) ocean_frame(ctx, i / 60.f, nullptr);
    ocean_sync(ctx);
    hipMemset(d_tl, 0, (size_t)maxblocks * 16 * 8 * 2);
    float ms = 0;
    // pass 1 alone (probes -> d_tl), then pass 2 alone (probes -> d_tl2)
    ocean_debug_pass(ctx, 1, 0.5f); ocean_sync(ctx);
    hipMemcpyToSymbol(HIP_SYMBOL(ocean_tl), &d_tl2, sizeof(d_tl2));
    ocean_debug_pass(ctx, 2, 0.5f); ocean_sync(ctx);
    ocean_time_frames(ctx, 20, 0.f, 1.f / 60, &ms);
    printf("N=%d frame %.1f us (with probes)\n", N, ms / 20 * 1000);
    {   // which side bounds a frame loop: the host's submission of the two launches, or the GPU?
        const int K = 4000;
        ocean_sync(ctx);
        const auto h0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K; ++i) ocean_frame(ctx, i / 60.f, nullptr);
        const auto h1 = std::chrono::steady_clock::now();
        ocean_sync(ctx);
        const auto h2 = std::chrono::steady_clock::now();
        {   // the same frame as a captured hipGraph (fixed time: a timing experiment): does a graph launch shorten the
            // two launch boundaries of a frame?
            hipGraph_t graph; hipGraphExec_t exec;
            hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeGlobal);
            ocean_frame(ctx, 0.5f, nullptr);
            hipStreamEndCapture(ctx->stream, &graph);
            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            for (int i = 0; i < 50; ++i) hipGraphLaunch(exec, ctx->stream);
            ocean_sync(ctx);
            const auto g0 = std::chrono::steady_clock::now();
            for (int i = 0; i < K; ++i) hipGraphLaunch(exec, ctx->stream);
            const auto g1 = std::chrono::steady_clock::now();
            ocean_sync(ctx);
            const auto g2 = std::chrono::steady_clock::now();
            printf("the frame as a hipGraph (%d launches): host submit %.2f us/frame, until the GPU is idle %.2f us/frame\n", K,
                   std::chrono::duration<double, std::micro>(g1 - g0).count() / K, std::chrono::duration<double, std::micro>(g2 - g0).count() / K);
            hipGraphExecDestroy(exec); hipGraphDestroy(graph);
        }
        printf("frame loop of %d frames: host submit %.2f us/frame, until the GPU is idle %.2f us/frame\n", K,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / K, std::chrono::duration<double, std::micro>(h2 - h0).count() / K);
    }
    std::vector<unsigned long long> tl((size_t)maxblocks * 16 * 2);
    hipMemcpy(tl.data(), d_tl, tl.size() * 8, hipMemcpyDeviceToHost);
    for (int pass = 1; pass <= 2; ++pass) {
        const unsigned long long* t = tl.data() + (pass == 2 ? (size_t)maxblocks * 16 : 0);
        const int first = 0;
        int nb = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        const int last = (pass == 1) ? 7 : 6;
        int nyq_block = -1;
        for (int b = first; b < maxblocks; ++b) {
            if (t[(size_t)b * 16 + 8]) nyq_block = b;
            bool ok = true;
            for (int k = 0; k <= last; ++k) ok = ok && (pass == 2 && k == 3 ? true : t[(size_t)b * 16 + k] != 0);
            if (!ok) continue;
            nb = b + 1;
            t0 = std::min(t0, t[(size_t)b * 16]);
            t1 = std::max(t1, t[(size_t)b * 16 + last]);
        }
        if (nyq_block >= 0) {
            const unsigned long long* q = t + (size_t)nyq_block * 16;
            printf("pass %d: Nyquist block %d: start %.2f us, loads done %.2f, end %.2f (relative to the first regular start)\n", pass, nyq_block,
                   ((double)q[8] - (double)t0) * 0.01, ((double)q[9] - (double)t0) * 0.01, ((double)q[10] - (double)t0) * 0.01);
        }
        auto complete = [&](int b) {
            for (int k = 0; k <= last; ++k) if (!(pass == 2 && k == 3) && t[(size_t)b * 16 + k] == 0) return false;
            return true;
        };
        int ncomplete = 0;
        for (int b = first; b < nb; ++b) ncomplete += complete(b);
        printf("pass %d: %d workgroups with complete probes, first start -> last probe %.2f us\n", pass, ncomplete, (t1 - t0) * 0.01);
        {   // the five workgroups that finish last
            std::vector<std::pair<double, int>> fin;
            for (int b = first; b < nb; ++b) if (complete(b)) fin.push_back({(t[(size_t)b * 16 + last] - t0) * 0.01, b});
            std::sort(fin.begin(), fin.end());
            printf("  last to finish:");
            for (size_t i = fin.size() > 5 ? fin.size() - 5 : 0; i < fin.size(); ++i)
                printf("  block %d start %.1f end %.1f;", fin[i].second, (t[(size_t)fin[i].second * 16] - t0) * 0.01, fin[i].first);
            printf("\n");
        }
        const char* names1[] = {"load+propagate", "spectrum+FFT height", "chunk stores height", "normalise+spectrum+FFT disp_x", "chunk stores disp_x", "spectrum+FFT disp_z", "chunk stores disp_z"};   // (N <= 512: one field per wave group, in field order)
        const char* names2[] = {"gather h + LDS expand", "FFT h", "gather dx,dz + LDS expand", "(read LDS)", "FFT dx+i dz", "RGBA stores"};
        const char* names2r[] = {"loads h + pair through LDS", "transform h, take the store mapping", "pair + transform disp_x", "", "pair + transform disp_z", "RGBA stores"};   // k_half_pass2_real (N >= 8192)
        const int nph = (pass == 1) ? 7 : 6;
        for (int k = 0; k < nph; ++k) {
            std::vector<double> v;
            for (int b = first; b < nb; ++b) {
                const unsigned long long a = t[(size_t)b * 16 + k], c = t[(size_t)b * 16 + k + 1];
                if (a && c) v.push_back((c - a) * 0.01);
            }
            if (pass == 2 && k == 2) { // probes 2 -> 4 (slot 3 unused)
                v.clear();
                for (int b = first; b < nb; ++b) if (complete(b)) v.push_back((t[(size_t)b * 16 + 4] - t[(size_t)b * 16 + 2]) * 0.01);
            }
            if (pass == 2 && k == 3) continue;
            stats(pass == 1 ? names1[k] : (N > 4096 ? names2r[k] : names2[k]), v);
        }
        std::vector<double> life, start;
        for (int b = first; b < nb; ++b) if (complete(b)) { life.push_back((t[(size_t)b * 16 + last] - t[(size_t)b * 16]) * 0.01); start.push_back((t[(size_t)b * 16] - t0) * 0.01); }
        stats("workgroup lifetime", life);
        stats("start time", start);
        // occupancy of each phase over time, 5 us bins
        const double total = (t1 - t0) * 0.01;
        const int bins = (int)(total / 5) + 1;
        printf("  t(us):  #WGs in [load | compute | storing] per 5 us bin\n");
        for (int bi = 0; bi < bins; ++bi) {
            const double tm = bi * 5 + 2.5;
            int c[3] = {0, 0, 0};
            for (int b = first; b < nb; ++b) {
                if (!complete(b)) continue;
                const unsigned long long* q = t + (size_t)b * 16;
                auto rel = [&](int k) { return (q[k] - t0) * 0.01; };
                if (tm < rel(0) || tm >= rel(last)) continue;
                if (pass == 1) {
                    if (tm < rel(1)) c[0]++;
                    else if ((tm >= rel(2) && tm < rel(3)) || (tm >= rel(4) && tm < rel(5)) || (tm >= rel(6) && tm < rel(7))) c[2]++;
                    else c[1]++;
                } else {
                    if (tm < rel(1) || (tm >= rel(2) && tm < rel(4))) c[0]++;
                    else if (tm >= rel(5)) c[2]++;
                    else c[1]++;
                }
            }
            printf("  %6.1f  %4d %4d %4d\n", tm, c[0], c[1], c[2]);
        }
        // workgroups per CU
        std::map<unsigned long long, int> cus;
        for (int b = first; b < nb; ++b) {
            if (!complete(b)) continue;
            const unsigned long long id = t[(size_t)b * 16 + 15];
            const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
            cus[((unsigned long long)xcc << 16) | (hw & 0xff00)]++;
        }
        std::map<int, int> hist;
        for (auto& kv : cus) hist[kv.second]++;
        printf("  distinct (xcc, se/sh/cu) ids: %zu;  workgroups per id:", cus.size());
        for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
        printf("\n");
    }
    ocean_context_destroy(ctx);
    return 0;
}
