// tune_rotate.hip -- developer harness: times launch-shape variants of the rotate kernels on the headline
// workload (B=50, 128x128, A angles, P=184).  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tune_rotate.bin tune_rotate.hip
#include "../ct_pvae_amd/csrc/core.hip"
#include "../ct_pvae_amd/csrc/rotate.hip"
#include "../ct_pvae_amd/csrc/rotate_plan.hip"
#include <vector>
#include <cstdio>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F> static float time_us(F f, int n = 200)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / n;
}

__global__ void empty_kernel() {}

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 50, A = argc > 2 ? atoi(argv[2]) : 20, N = argc > 3 && atoi(argv[3]) > 0 ? atoi(argv[3]) : 128;
    const int P = ctpvae_num_proj_pix(N, N), pad = (P - N) / 2;
    std::vector<float> theta(A), img((size_t)S * N * N), g((size_t)S * A * P);
    for (int a = 0; a < A; ++a) theta[a] = (float)(M_PI * (a * (180 / A)) / 180.0);
    unsigned s = 1;
    for (auto &v : img) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    for (auto &v : g) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    float *d_theta, *d_img, *d_g, *d_T, *d_Ti, *d_sino, *d_gimg;
    CK(hipMalloc(&d_theta, A * 4)); CK(hipMalloc(&d_img, img.size() * 4)); CK(hipMalloc(&d_g, g.size() * 4));
    CK(hipMalloc(&d_T, A * 32)); CK(hipMalloc(&d_Ti, A * 32)); CK(hipMalloc(&d_sino, g.size() * 4)); CK(hipMalloc(&d_gimg, img.size() * 4));
    CK(hipMemcpy(d_theta, theta.data(), A * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_g, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    if (ctpvae_rotate_transforms_f32(d_theta, A, P, P, d_T, d_Ti, nullptr)) { printf("%s\n", ctpvae_last_error()); return 1; }
    CK(hipDeviceSynchronize());

    printf("S=%d A=%d N=%d P=%d\n", S, A, N, P);
    if (N > 256) {   // slices larger than LDS: the tiled forward and the segment backward
        const long long wsb = ctpvae_rotate_fwd_tiled_workspace_bytes(S, N, N, P, P, A, 0);
        void *ws; CK(hipMalloc(&ws, wsb));
        printf("tiled fwd: %.2f us (workspace %.1f MB)\n", time_us([&] { if (ctpvae_rotate_fwd_tiled_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, ws, d_sino, nullptr)) { printf("%s\n", ctpvae_last_error()); exit(1); } }, 50), wsb / 1e6);
        printf("segment bwd: %.2f us\n", time_us([&] { ctpvae_rotate_bwd_f32(d_g, S, A, P, P, d_Ti, 0, 0, N, N, pad, pad, d_gimg, nullptr); }, 50));
#ifdef CTPVAE_TUNE_STAMPS
        {
            CK(hipDeviceSynchronize());
            ctpvae_rotate_fwd_tiled_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, ws, d_sino, nullptr);
            CK(hipDeviceSynchronize());
            const int nw = getenv("CTPVAE_TUNE_NW") ? atoi(getenv("CTPVAE_TUNE_NW")) : 192 * 16;
            std::vector<long long> st(8 * nw);
            CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(ctpvae::g_stamps), st.size() * 8));
            long long t0min = st[0], tend = 0;
            for (int w = 0; w < nw; ++w) { t0min = std::min(t0min, st[8 * w]); tend = std::max(tend, st[8 * w + 5]); }
            double seg[3] = {0, 0, 0}, start = 0, km = 0, endsk = 0, lastwalk = 0;
            for (int w = 0; w < nw; ++w) lastwalk += (double)(st[8 * w + 4] - st[8 * w + 3]);
            double lastsetup = 0; for (int w = 0; w < nw; ++w) lastsetup += (double)(st[8 * w + 3] - st[8 * w + 7]);
            printf("last task of each wave: setup %.0f cycles, walk %.0f cycles, end-of-walk -> wave end %.0f\n", lastsetup / nw, lastwalk / nw, [&]{ double e = 0; for (int w = 0; w < nw; ++w) e += (double)(st[8 * w + 5] - st[8 * w + 4]); return e / nw; }());
            for (int w = 0; w < nw; ++w) {
                seg[0] += (double)(st[8 * w + 1] - st[8 * w]); seg[1] += (double)(st[8 * w + 2] - st[8 * w + 1]); seg[2] += (double)(st[8 * w + 5] - st[8 * w + 2]);
                start += (double)(st[8 * w] - t0min); km += (double)st[8 * w + 6]; endsk += (double)(tend - st[8 * w + 5]);
            }
            printf("tiled stamps over %d waves (cycles): start-skew %.0f | stage-issue %.0f | barrier %.0f | tasks %.0f | idle-after-end %.0f | last kmax %.1f | first-start->last-end %lld\n",
                   nw, start / nw, seg[0] / nw, seg[1] / nw, seg[2] / nw, endsk / nw, km / nw, tend - t0min);
        }
#endif
        return 0;
    }
    if (argc > 4) {   // counter mode: a few launches of the product kernels only
        void *fp, *bp;
        CK(hipMalloc(&fp, ctpvae_rotate_plan_bytes(N, N, P, P, A, 0))); CK(hipMalloc(&bp, ctpvae_rotate_plan_bytes(N, N, P, P, A, 1)));
        ctpvae_rotate_plan_build_f32(d_T, d_Ti, A, N, N, P, P, pad, pad, fp, bp, nullptr);
        for (int i = 0; i < 3; ++i) {
            ctpvae_rotate_fwd_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, 0, d_sino, nullptr);
            ctpvae_rotate_bwd_f32(d_g, S, A, P, P, d_Ti, 0, 0, N, N, pad, pad, d_gimg, nullptr);
            ctpvae_rotate_fwd_planned_f32(d_img, S, N, N, P, P, A, fp, d_sino, nullptr);
            ctpvae_rotate_bwd_planned_f32(d_g, S, N, N, P, P, A, bp, d_gimg, nullptr);
        }
        CK(hipDeviceSynchronize());
        return 0;
    }
    printf("empty kernel back-to-back: %.2f us\n", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }));
    printf("product fwd : %.2f us\n", time_us([&] { ctpvae_rotate_fwd_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, 0, d_sino, nullptr); }));
    printf("product bwd : %.2f us\n", time_us([&] { ctpvae_rotate_bwd_f32(d_g, S, A, P, P, d_Ti, 0, 0, N, N, pad, pad, d_gimg, nullptr); }));
    {
        void *fp, *bp;
        CK(hipMalloc(&fp, ctpvae_rotate_plan_bytes(N, N, P, P, A, 0))); CK(hipMalloc(&bp, ctpvae_rotate_plan_bytes(N, N, P, P, A, 1)));
        printf("plan build (fwd+bwd): %.2f us\n", time_us([&] { ctpvae_rotate_plan_build_f32(d_T, d_Ti, A, N, N, P, P, pad, pad, fp, bp, nullptr); }));
        printf("planned fwd : %.2f us\n", time_us([&] { if (ctpvae_rotate_fwd_planned_f32(d_img, S, N, N, P, P, A, fp, d_sino, nullptr)) { printf("%s\n", ctpvae_last_error()); exit(1); } }));
        printf("planned bwd : %.2f us\n", time_us([&] { if (ctpvae_rotate_bwd_planned_f32(d_g, S, N, N, P, P, A, bp, d_gimg, nullptr)) { printf("%s\n", ctpvae_last_error()); exit(1); } }));
        for (int ns : {1, 2}) for (int G : {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}) {
            char b[16]; snprintf(b, 16, "%d", G); ctpvae_tune_set("G", atoi(b));
            snprintf(b, 16, "%d", ns); ctpvae_tune_set("NS", atoi(b));
            printf("planned fwd NS=%d G=%d: %.2f us\n", ns, G, time_us([&] { ctpvae_rotate_fwd_planned_f32(d_img, S, N, N, P, P, A, fp, d_sino, nullptr); }, 100));
        }
        ctpvae_tune_set("NS", -1);
        ctpvae_tune_set("G", -1);
        for (int bw : {2, 4, 8, 16}) {
            char b[16]; snprintf(b, 16, "%d", bw); ctpvae_tune_set("BW", atoi(b));
            printf("planned bwd waves=%d: %.2f us\n", bw, time_us([&] { ctpvae_rotate_bwd_planned_f32(d_g, S, N, N, P, P, A, bp, d_gimg, nullptr); }, 100));
        }
        ctpvae_tune_set("BW", -1);
        ctpvae_tune_set("G", -1);
    }
    printf("product fwd bilinear: %.2f us\n", time_us([&] { ctpvae_rotate_fwd_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, 1, d_sino, nullptr); }));
    printf("product bwd bilinear: %.2f us\n", time_us([&] { ctpvae_rotate_bwd_f32(d_g, S, A, P, P, d_Ti, 1, 0, N, N, pad, pad, d_gimg, nullptr); }));

    const ctpvae::RotGeom geo{S, N, N, P, P, pad, pad, A};
    const size_t lds = (size_t)(N + 2) * 161 * 4;
#ifdef CTPVAE_TUNE_STAMPS
    for (int cfg : {15, 25, 24}) {   // (slices per workgroup, groups)
        const int Gs = cfg % 10, nsl = cfg / 10;
        char gb[16]; snprintf(gb, 16, "%d", Gs); ctpvae_tune_set("G", atoi(gb));
        snprintf(gb, 16, "%d", nsl); ctpvae_tune_set("NS", atoi(gb));
        void *fp; CK(hipMalloc(&fp, ctpvae_rotate_plan_bytes(N, N, P, P, A, 0)));
        ctpvae_rotate_plan_build_f32(d_T, d_Ti, A, N, N, P, P, pad, pad, fp, nullptr, nullptr);
        for (int rep = 0; rep < 2; ++rep) { ctpvae_rotate_fwd_planned_f32(d_img, S, N, N, P, P, A, fp, d_sino, nullptr); CK(hipDeviceSynchronize()); }
        const int G = Gs;
        const int units = (S + nsl - 1) / nsl;
        const int T = A * 3, wv = std::min(16, std::max(nsl == 2 ? 16 : 8, (T + 2 * G - 1) / (2 * G))), nw = units * 2 * G * wv;
        printf("NS=%d G=%d: %d workgroups x %d waves\n", nsl, G, units * 2 * G, wv);
        std::vector<long long> st(8 * nw);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(ctpvae::g_pstamps), st.size() * 8));
        double seg[3] = {0, 0, 0};
        for (int w = 0; w < nw; ++w) for (int q = 0; q < 3; ++q) seg[q] += (double)(st[8 * w + q + 1] - st[8 * w + q]);
        printf("planned fwd stamps over %d waves (cycles): fill-issue %.0f | barrier wait %.0f | tasks %.0f\n", nw, seg[0] / nw, seg[1] / nw, seg[2] / nw);
        {   // global timeline in 10 ns ticks (s_memrealtime, 100 MHz)
            long long t0 = st[4], t1 = 0; std::vector<long long> starts, ends;
            for (int w = 0; w < nw; ++w) { t0 = std::min(t0, st[8 * w + 4]); t1 = std::max(t1, st[8 * w + 5]); }
            for (int w = 0; w < nw; ++w) { starts.push_back(st[8 * w + 4] - t0); ends.push_back(st[8 * w + 5] - t0); }
            std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
            printf("  timeline (us): span %.2f | wave starts p0 %.2f p50 %.2f p90 %.2f p100 %.2f | wave ends p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f\n",
                   (t1 - t0) * 0.01, starts[0] * 0.01, starts[nw / 2] * 0.01, starts[nw * 9 / 10] * 0.01, starts[nw - 1] * 0.01,
                   ends[0] * 0.01, ends[nw / 10] * 0.01, ends[nw / 2] * 0.01, ends[nw * 9 / 10] * 0.01, ends[nw - 1] * 0.01);
        }
    }
    {
        ctpvae_rotate_fwd_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, 0, d_sino, nullptr);
        CK(hipDeviceSynchronize());
        ctpvae_rotate_fwd_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, 0, d_sino, nullptr);
        CK(hipDeviceSynchronize());
        const int rpb_ = getenv("CTPVAE_TUNE_RPB") ? atoi(getenv("CTPVAE_TUNE_RPB")) : 768;
        const int nw = S * ((A * P + rpb_ - 1) / rpb_) * (rpb_ / 64);
        std::vector<long long> st(8 * nw);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(ctpvae::g_stamps), st.size() * 8));
        long long t0min = st[0], tend = 0;
        for (int w = 0; w < nw; ++w) { t0min = std::min(t0min, st[8 * w]); tend = std::max(tend, st[8 * w + 5]); }
        double seg[5] = {0, 0, 0, 0, 0}, start = 0, km = 0;
        for (int w = 0; w < nw; ++w) {
            for (int q = 0; q < 5; ++q) seg[q] += (double)(st[8 * w + q + 1] - st[8 * w + q]);
            start += (double)(st[8 * w] - t0min); km += (double)st[8 * w + 6];
        }
        printf("stamps over %d waves (cycles): start-skew %.0f | fill %.0f | barrier %.0f | setup %.0f | loop %.0f | store %.0f | kmax %.1f | first-start->last-end %lld\n",
               nw, start / nw, seg[0] / nw, seg[1] / nw, seg[2] / nw, seg[3] / nw, seg[4] / nw, km / nw, tend - t0min);
        return 0;
    }
#endif
    auto kern = ctpvae::rotate_fwd_fast_kernel<0, false, false, 1>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int nrays = A * P;
    for (int rpb : {384, 768, 1024, 1856}) {
        if (rpb > nrays) continue;
        for (int block : {384, 768, 1024}) {
            if (block > rpb) continue;
            dim3 grid((nrays + rpb - 1) / rpb, S);
            float t = time_us([&] { hipLaunchKernelGGL(kern, grid, dim3(block), lds, 0, d_img, geo, ctpvae::TileSpec{}, d_T, rpb, d_sino); }, 100);
            printf("fwd rpb=%4d block=%4d grid=%4u x %d : %.2f us\n", rpb, block, grid.x, S, t);
        }
    }
    return 0;
}
