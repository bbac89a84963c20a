// dft_mfma_probe.hip -- round 5's go / no-go probe for VERDICT r4 "next" #1: the 480-point complex DFT behind the 960-point real
// transforms as two matrix products on the matrix cores (nnn_dft_mfma.h) against today's three-pass LDS transform (fft480_regs),
// same harness, same inputs: time per transform and error against a double-precision DFT.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nnnoiseless_amd/csrc scripts/ubench/dft_mfma_probe.hip -o dft_probe
//   ./dft_probe [blocks_per_cu] [iters]
// Also builds against the test-only SIMT interpreter (tests/hostsim; -DNNN_PROBE_HOSTSIM): logic and accuracy without a GPU.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define NNN_WEIGHTS_PATH ""
#include "nnn_kernels.hip"
#include "nnn_dft_mfma.h"

using namespace nnn;

constexpr int WPB = 4;   // waves per block, as k_fft_xp / k_synth

// MODE 0: fft480_regs (input order n = j + 60 r on lanes j < 60); 1: DftF16; 2: DftBf16 (input order dft_in_n)
template <int MODE>
__global__ void __launch_bounds__(64 * WPB) k_probe(const float2 *in, float2 *out, const float2 *tw960, const void *img, int iters, int nsets, int out_sets)
{
    __shared__ float2 tw[NFFT];
    __shared__ float2 Zs[WPB][NFFT_BUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = (int)blockIdx.x * WPB + wave, nw = (int)gridDim.x * WPB;
    float2 *Z = Zs[wave];
    if (MODE == 0) {
        for (int i = threadIdx.x; i < NFFT; i += blockDim.x) tw[i] = tw960[i];
        __syncthreads();
    }
    DftRegs<DftF16> c16;
    DftRegs<DftBf16> cb;
    if (MODE == 1) dft_regs_load(c16, img, lane);
    if (MODE == 2) dft_regs_load(cb, img, lane);
    for (int it = 0; it < iters; it++) {
        const int set = (int)(((long long)w + (long long)it * nw) % nsets);
        const float2 *x = in + (size_t)set * NFFT;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = MODE == 0 ? (lane < FFT_P1 ? lane : FFT_P1 - 1) + FFT_P1 * r : dft_in_n_clamped(lane, r);
            v[r] = x[n];
        }
        if (MODE == 0) fft480_regs(v, Z, tw, lane);
        else {
            if (MODE == 1) dft480_mfma<DftF16>(v, Z, c16, lane);
            else dft480_mfma<DftBf16>(v, Z, cb, lane);
            wave_lds_sync();
        }
        float2 *o = out + (size_t)(w % out_sets) * NFFT;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = lane + 64 * u;
            if (k < NFFT) o[k] = Z[k];
        }
        wave_lds_sync();
    }
}

static void ref_dft(const float2 *x, double *re, double *im)
{
    static std::vector<double> c, s;
    if (c.empty()) {
        c.resize(NFFT); s.resize(NFFT);
        for (int i = 0; i < NFFT; i++) { c[i] = cos(2.0 * M_PI * i / NFFT); s[i] = sin(2.0 * M_PI * i / NFFT); }
    }
    for (int k = 0; k < NFFT; k++) {
        double ar = 0, ai = 0;
        for (int n = 0; n < NFFT; n++) {
            const int j = (int)(((long long)k * n) % NFFT);
            ar += x[n].x * c[j] + x[n].y * s[j];
            ai += x[n].y * c[j] - x[n].x * s[j];
        }
        re[k] = ar; im[k] = ai;
    }
}

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int bpc = argc > 1 ? atoi(argv[1]) : 8, iters = argc > 2 ? atoi(argv[2]) : 64;
#ifdef NNN_PROBE_HOSTSIM
    const int ncu = 1, nver = 8;
#else
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, nver = 512;
    printf("device: %s, %d CUs, clock %d MHz\n", prop.name, ncu, prop.clockRate / 1000);
#endif
    // ---- inputs: (a) windowed sine + noise packed as 480 complex, (b) Gaussian, (c) a spectrum-like set (a few huge bins over a floor)
    const int nsets = 3 * nver;
    std::vector<float2> hin((size_t)nsets * NFFT);
    srand(12345);
    auto urand = [] { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); };
    auto grand = [&] { return sqrt(-2.0 * log(urand())) * cos(2.0 * M_PI * urand()); };
    for (int s = 0; s < nsets; s++) {
        const int kind = s / nver;
        const double f = 80.0 + 920.0 * urand(), ph = 6.28 * urand(), A = 500.0 + 11500.0 * urand(), sg = 50.0 + 2950.0 * urand();
        for (int n = 0; n < NFFT; n++) {
            float2 z;
            if (kind == 0) {
                double v[2];
                for (int h = 0; h < 2; h++) {
                    const int i = 2 * n + h;
                    const double win = sin(0.5 * M_PI * pow(sin(0.5 * M_PI * (i + 0.5) / 480.0), 2.0));
                    v[h] = win * 0.5 * (A * sin(2.0 * M_PI * f * i / 48000.0 + ph) + sg * grand());
                }
                z = make_float2((float)v[0], (float)v[1]);
            } else if (kind == 1) z = make_float2((float)(3000.0 * grand()), (float)(3000.0 * grand()));
            else {
                const bool big = n % 97 == s % 97;
                z = make_float2((float)((big ? 4.0e6 : 30.0) * grand()), (float)((big ? 4.0e6 : 30.0) * grand()));
            }
            hin[(size_t)s * NFFT + n] = z;
        }
    }
    std::vector<float2> htw(960);
    for (int i = 0; i < 960; i++) htw[i] = make_float2((float)cos(2.0 * M_PI * i / 960.0), (float)-sin(2.0 * M_PI * i / 960.0));
    std::vector<unsigned char> img16(DftImage<DftF16>::BYTES), imgb(DftImage<DftBf16>::BYTES);
    dft_mfma_image<DftF16>(img16.data());
    dft_mfma_image<DftBf16>(imgb.data());
    float2 *din, *dout, *dtw;
    void *dimg16, *dimgb;
    CHK(hipMalloc((void **)&din, hin.size() * sizeof(float2)));
    CHK(hipMalloc((void **)&dout, (size_t)nsets * NFFT * sizeof(float2)));
    CHK(hipMalloc((void **)&dtw, 960 * sizeof(float2)));
    CHK(hipMalloc(&dimg16, img16.size()));
    CHK(hipMalloc(&dimgb, imgb.size()));
    CHK(hipMemcpy(din, hin.data(), hin.size() * sizeof(float2), hipMemcpyHostToDevice));
    CHK(hipMemcpy(dtw, htw.data(), 960 * sizeof(float2), hipMemcpyHostToDevice));
    CHK(hipMemcpy(dimg16, img16.data(), img16.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(dimgb, imgb.data(), imgb.size(), hipMemcpyHostToDevice));
    // ---- reference
    std::vector<double> rre((size_t)nsets * NFFT), rim((size_t)nsets * NFFT);
    for (int s = 0; s < nsets; s++) ref_dft(&hin[(size_t)s * NFFT], &rre[(size_t)s * NFFT], &rim[(size_t)s * NFFT]);
    const char *names[3] = {"fft480_regs (radix 8x6x10, LDS)", "dft480_mfma<DftF16>  (2 planes, 36 MFMA)", "dft480_mfma<DftBf16> (3 planes, 72 MFMA)"};
    const char *kinds[3] = {"windowed sine+noise", "gaussian", "sparse spectrum (4e6 over 30)"};
    std::vector<float2> hout((size_t)nsets * NFFT);
    double t_us[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; mode++) {
        const void *img = mode == 1 ? dimg16 : dimgb;
        auto launch = [&](int blocks, int it, int ns, int os) {
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(64 * WPB), 0, 0, din, dout, dtw, img, it, ns, os);
            if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(64 * WPB), 0, 0, din, dout, dtw, img, it, ns, os);
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(64 * WPB), 0, 0, din, dout, dtw, img, it, ns, os);
        };
        // verification: one transform per wave, wave w -> set w
        CHK(hipMemset(dout, 0, (size_t)nsets * NFFT * sizeof(float2)));
        launch(nsets / WPB, 1, nsets, nsets);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(hout.data(), dout, hout.size() * sizeof(float2), hipMemcpyDeviceToHost));
        printf("%s\n", names[mode]);
        for (int kind = 0; kind < 3; kind++) {
            double num = 0, den = 0, worst = 0, relbin_sum = 0;
            long nb = 0;
            for (int s = kind * nver; s < (kind + 1) * nver; s++) {
                double n1 = 0, d1 = 0;
                for (int k = 0; k < NFFT; k++) {
                    const size_t i = (size_t)s * NFFT + k;
                    const double er = hout[i].x - rre[i], ei = hout[i].y - rim[i];
                    n1 += er * er + ei * ei;
                    d1 += rre[i] * rre[i] + rim[i] * rim[i];
                    relbin_sum += sqrt((er * er + ei * ei) / (rre[i] * rre[i] + rim[i] * rim[i] + 1e-300));
                    nb++;
                }
                num += n1; den += d1;
                if (sqrt(n1 / d1) > worst) worst = sqrt(n1 / d1);
            }
            printf("    %-32s rel-rms error %.3e   worst transform %.3e   mean per-bin relative error %.3e\n", kinds[kind], sqrt(num / den), worst, relbin_sum / nb);
        }
#ifndef NNN_PROBE_HOSTSIM
        hipFuncAttributes fa;
        const void *fp = mode == 0 ? (const void *)k_probe<0> : (mode == 1 ? (const void *)k_probe<1> : (const void *)k_probe<2>);
        CHK(hipFuncGetAttributes(&fa, fp));
        int occ = 0;
        CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fp, 64 * WPB, 0));
        const int blocks = ncu * bpc;
        launch(blocks, 4, nver, nsets);   // warm-up
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHK(hipEventRecord(e0, 0));
            launch(blocks, iters, nver, nsets);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double ntr = (double)blocks * WPB * iters;
        t_us[mode] = best * 1e3;
        printf("    registers %d, LDS %zu B, blocks per CU %d;  %d blocks x %d waves x %d transforms: %.1f us -> %.2f ns per transform (chip), "
               "%.0f SIMD-cycles per transform at 2.4 GHz\n", fa.numRegs, (size_t)fa.sharedSizeBytes, occ, blocks, WPB, iters, best * 1e3,
               best * 1e6 / ntr, best * 1e-3 * 2.4e9 * ncu * 4 / ntr);
#endif
    }
#ifndef NNN_PROBE_HOSTSIM
    printf("time ratio  DftF16 / fft480_regs = %.3f   DftBf16 / fft480_regs = %.3f\n", t_us[1] / t_us[0], t_us[2] / t_us[0]);
#endif
    return 0;
}
