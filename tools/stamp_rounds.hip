// stamp_rounds.hip -- developer harness (round 5): what does a launch of SEVERAL ROUNDS of workgroups wait for?
// Runs the u16 planned forward at S slices x A angles (128 x 128) in a -DCTPVAE_TUNE_STAMPS build and writes one line per
// workgroup -- start, fill issued, barrier passed, end (10 ns ticks of s_memrealtime / shader cycles of s_memtime), the CU it ran
// on -- to a text file; tools/analyse_rounds.py turns the file into the per-CU timeline and the summary of profiles/r05_rounds.txt.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DCTPVAE_TUNE_STAMPS -o stamp_rounds.bin stamp_rounds.hip
//   ./stamp_rounds.bin S A out.txt [G] [NS]
#include "../ct_pvae_amd/csrc/core.hip"
#include "../ct_pvae_amd/csrc/rotate.hip"
#include "../ct_pvae_amd/csrc/rotate_bilin.hip"
#include "../ct_pvae_amd/csrc/rotate_plan.hip"
#include <vector>
#include <cstdio>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    if (argc < 4) { printf("usage: stamp_rounds.bin S A out.txt [G] [NS]\n"); return 1; }
    const int S = atoi(argv[1]), A = atoi(argv[2]), N = 128;
    if (argc > 4 && atoi(argv[4]) > 0) ctpvae_tune_set("G", atoi(argv[4]));
    if (argc > 5 && atoi(argv[5]) > 0) ctpvae_tune_set("NS", atoi(argv[5]));
    const int P = ctpvae_num_proj_pix(N, N), pad = (P - N) / 2;
    std::vector<float> theta(A), img((size_t)S * N * N);
    for (int a = 0; a < A; ++a) theta[a] = (float)(M_PI * a / A);
    unsigned s = 1;
    for (auto &v : img) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    float *d_theta, *d_img, *d_T, *d_Ti, *d_sino;
    CK(hipMalloc(&d_theta, A * 4)); CK(hipMalloc(&d_img, img.size() * 4));
    CK(hipMalloc(&d_T, A * 32)); CK(hipMalloc(&d_Ti, A * 32)); CK(hipMalloc(&d_sino, (size_t)S * A * P * 4));
    CK(hipMemcpy(d_theta, theta.data(), A * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    if (ctpvae_rotate_transforms_f32(d_theta, A, P, P, d_T, d_Ti, nullptr)) { printf("%s\n", ctpvae_last_error()); return 1; }
    void *fp;
    CK(hipMalloc(&fp, ctpvae_rotate_plan_bytes(N, N, P, P, A, 0)));
    if (ctpvae_rotate_plan_build_f32(d_T, d_Ti, A, N, N, P, P, pad, pad, fp, nullptr, nullptr)) { printf("%s\n", ctpvae_last_error()); return 1; }
    // CTPVAE_STAMP_BWD=1: the planned backward instead (the cotangents = the forward's sinograms)
    const bool bwd = getenv("CTPVAE_STAMP_BWD") != nullptr;
    void *bp = nullptr;
    float *d_gimg = nullptr;
    if (bwd) {
        CK(hipMalloc(&bp, ctpvae_rotate_plan_bytes(N, N, P, P, A, 1)));
        CK(hipMalloc(&d_gimg, img.size() * 4));
        if (ctpvae_rotate_plan_build_f32(d_T, d_Ti, A, N, N, P, P, pad, pad, nullptr, bp, nullptr)) { printf("%s\n", ctpvae_last_error()); return 1; }
        if (ctpvae_rotate_fwd_planned_f32(d_img, S, N, N, P, P, A, fp, d_sino, nullptr)) { printf("%s\n", ctpvae_last_error()); return 1; }
    }
    const bool bilin = getenv("CTPVAE_STAMP_BILIN") != nullptr;   // the bilinear forward of whole slices instead
    auto run = [&] {
        if (bilin) {
            if (ctpvae_rotate_fwd_f32(d_img, S, N, N, P, P, pad, pad, d_T, A, CTPVAE_BILINEAR, d_sino, nullptr)) { printf("%s\n", ctpvae_last_error()); exit(1); }
            return;
        }
        const int rc = bwd ? ctpvae_rotate_bwd_planned_f32(d_sino, S, N, N, P, P, A, bp, d_gimg, nullptr)
                           : ctpvae_rotate_fwd_planned_f32(d_img, S, N, N, P, P, A, fp, d_sino, nullptr);
        if (rc) { printf("%s\n", ctpvae_last_error()); exit(1); }
    };
    for (int i = 0; i < 20; ++i) run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) run();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // the launch that is looked at: the last of a back-to-back series (clocks and caches warm)
    long long *dst; CK(hipGetSymbolAddress((void **)&dst, HIP_SYMBOL(ctpvae::g_pstamps)));
    CK(hipMemset(dst, 0, sizeof(long long) * 8 * 65536));
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
    std::vector<long long> st(8 * 65536);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    // workgroups = runs of waves that share a block: the kernel numbers waves blockIdx * nwaves + wave; nwaves = first gap
    int nw = 0;
    while (nw < 65536 && st[8 * nw + 4] != 0) ++nw;
    FILE *f = fopen(argv[3], "w");
    fprintf(f, "# S %d A %d waves %d us_per_launch_stamped_build %.2f units %d wgs_per_unit %d waves_per_wg %d ns %d affine %d units1 %d wgs_per_unit2 %d units2 %d wgs_per_unit3 %d\n", S, A, nw, ms * 1000.0 / 50,
            ctpvae::g_pshape[0], ctpvae::g_pshape[1], ctpvae::g_pshape[2], ctpvae::g_pshape[3], ctpvae::g_pshape[4], ctpvae::g_pshape[5], ctpvae::g_pshape[6], ctpvae::g_pshape[7], ctpvae::g_pshape[8]);
    fprintf(f, "# wave start_rt end_rt start_cyc fill_cyc barrier_cyc end_cyc hwid xcc issued_cyc\n");
    for (int w = 0; w < nw; ++w)
        fprintf(f, "%d %lld %lld %lld %lld %lld %lld %lld %lld %lld\n", w, st[8 * w + 4], st[8 * w + 5], st[8 * w], st[8 * w + 1], st[8 * w + 2], st[8 * w + 3],
                st[8 * w + 6] & 0xffffffffll, st[8 * w + 6] >> 32, st[8 * w + 7]);
    fclose(f);
    printf("S=%d A=%d: %d waves stamped, %.2f us per launch (stamped build)\n", S, A, nw, ms * 1000.0 / 50);
    return 0;
}
