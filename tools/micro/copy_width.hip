// What a plain streaming copy reaches on MI355X by access width and pattern (the kernels of this library
// move 8 bytes per lane): hipcc --offload-arch=gfx950 -O3 copy_width.hip -o copy_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <typename V>
__global__ __launch_bounds__(256) void copy_k(V *__restrict__ dst, const V *__restrict__ src, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
// the access pattern of sw2d_fused_kernel: one wave per workgroup marches down a 64-column strip of NF
// arrays [H][W], reading a row of each and writing a row of each (one row ahead in flight)
template <int NF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void strip_k(double *const *dst, const double *const *src, int W, int H, int rows_per_band) {
    const int strips = W / (64 * WAVES);
    const int band = blockIdx.x / strips, i = (blockIdx.x % strips) * 64 * WAVES + threadIdx.x;
    const int j0 = band * rows_per_band, j1 = min(H, j0 + rows_per_band);
    double cur[NF], nxt[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[f] = src[f][(long)j0 * W + i];
    for (int j = j0; j < j1; ++j) {
        const int jn = min(j + 1, j1 - 1);
#pragma unroll
        for (int f = 0; f < NF; ++f) nxt[f] = src[f][(long)jn * W + i];
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f][(long)j * W + i] = cur[f] * 1.0000001;
#pragma unroll
        for (int f = 0; f < NF; ++f) cur[f] = nxt[f];
    }
}
// variants of the strip march: MODE 0 = read + write, 1 = read only (one store per wave at the end), 2 = write only;
// NT = nontemporal loads / stores; D = rows requested per iteration (D rows of every array in flight)
template <int NF, int MODE, bool NT, int D>
__global__ __launch_bounds__(64) void strip2_k(double *const *dst, const double *const *src, int W, int H, int rows_per_band) {
    const int strips = W / 64;
    const int band = blockIdx.x / strips, i = (blockIdx.x % strips) * 64 + threadIdx.x;
    const int j0 = band * rows_per_band, j1 = min(H, j0 + rows_per_band);
    double cur[D][NF], nxt[D][NF], acc = 0.0;
    const auto ld = [&](int f, int j) { const double *p = &src[f][(long)min(j, j1 - 1) * W + i]; return NT ? __builtin_nontemporal_load(p) : *p; };
    const auto st = [&](int f, int j, double v) { if (j < j1) { double *p = &dst[f][(long)j * W + i]; if (NT) __builtin_nontemporal_store(v, p); else *p = v; } };
    if (MODE != 2) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int f = 0; f < NF; ++f) cur[d][f] = ld(f, j0 + d);
    }
    for (int j = j0; j < j1; j += D) {
        if (MODE != 2) {
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int f = 0; f < NF; ++f) nxt[d][f] = ld(f, j + D + d);
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (MODE == 0) st(f, j + d, cur[d][f] * 1.0000001);
                if (MODE == 1) acc += cur[d][f];
                if (MODE == 2) st(f, j + d, 1.5 * j);
            }
        if (MODE != 2) {
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int f = 0; f < NF; ++f) cur[d][f] = nxt[d][f];
        }
    }
    if (MODE == 1 && acc == 12345.678) dst[0][i] = acc;
}
// reads one row per iteration (one row ahead in flight, nontemporal), writes held back and issued WB rows at a time
template <int NF, int WB, bool NTL, bool NTS>
__global__ __launch_bounds__(64) void strip3_k(double *const *dst, const double *const *src, int W, int H, int rows_per_band) {
    const int strips = W / 64;
    const int band = blockIdx.x / strips, i = (blockIdx.x % strips) * 64 + threadIdx.x;
    const int j0 = band * rows_per_band, j1 = min(H, j0 + rows_per_band);
    double cur[NF], nxt[NF], held[WB][NF];
    const auto ld = [&](int f, int j) { const double *p = &src[f][(long)min(j, j1 - 1) * W + i]; return NTL ? __builtin_nontemporal_load(p) : *p; };
    const auto st = [&](int f, int j, double v) { if (j < j1) { double *p = &dst[f][(long)j * W + i]; if (NTS) __builtin_nontemporal_store(v, p); else *p = v; } };
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[f] = ld(f, j0);
    for (int j = j0; j < j1; j += WB) {
#pragma unroll
        for (int d = 0; d < WB; ++d) {
#pragma unroll
            for (int f = 0; f < NF; ++f) nxt[f] = ld(f, j + d + 1);
#pragma unroll
            for (int f = 0; f < NF; ++f) held[d][f] = cur[f] * 1.0000001;
#pragma unroll
            for (int f = 0; f < NF; ++f) cur[f] = nxt[f];
        }
#pragma unroll
        for (int d = 0; d < WB; ++d)
#pragma unroll
            for (int f = 0; f < NF; ++f) st(f, j + d, held[d][f]);
    }
}
// the strip march with a row pitch different from the row width (W columns used, P doubles between rows)
template <int NF>
__global__ __launch_bounds__(64) void strip_pitch_k(double *const *dst, const double *const *src, int W, int P, int H, int rows_per_band) {
    const int strips = W / 64;
    const int band = blockIdx.x / strips, i = (blockIdx.x % strips) * 64 + threadIdx.x;
    const int j0 = band * rows_per_band, j1 = min(H, j0 + rows_per_band);
    double cur[NF], nxt[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[f] = src[f][(long)j0 * P + i];
    for (int j = j0; j < j1; ++j) {
        const int jn = min(j + 1, j1 - 1);
#pragma unroll
        for (int f = 0; f < NF; ++f) nxt[f] = src[f][(long)jn * P + i];
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f][(long)j * P + i] = cur[f] * 1.0000001;
#pragma unroll
        for (int f = 0; f < NF; ++f) cur[f] = nxt[f];
    }
}
// the strip march southwards (up = 0) or northwards (up = 1): consecutive steps ping-pong src / dst as the model does;
// alternating the direction lets a step begin on the rows the previous one touched last (Infinity Cache reuse?)
template <int NF>
__global__ __launch_bounds__(64) void strip_dir_k(double *const *dst, const double *const *src, int W, int H, int rows_per_band, int up, int rev = 0) {
    // 13 KB of LDS per wave: 12 waves per CU, the residency of the real kernel (3 per SIMD), so that a grid of more
    // than 3072 waves runs in rounds, in workgroup-id order
    __shared__ double occ[1664];
    occ[threadIdx.x] = 0.0;
    const int strips = W / 64;
    const int tile = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int band = tile / strips, i = (tile % strips) * 64 + threadIdx.x;
    const int j0 = band * rows_per_band, j1 = min(H, j0 + rows_per_band);
    const int n = j1 - j0;
    const auto row = [&](int t) { return up ? j1 - 1 - t : j0 + t; };
    double cur[NF], nxt[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[f] = src[f][(long)row(0) * W + i];
    for (int t = 0; t < n; ++t) {
        const int jn = row(min(t + 1, n - 1));
#pragma unroll
        for (int f = 0; f < NF; ++f) nxt[f] = src[f][(long)jn * W + i];
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f][(long)row(t) * W + i] = cur[f] * 1.0000001;
#pragma unroll
        for (int f = 0; f < NF; ++f) cur[f] = nxt[f];
    }
}
int main() {
    const int W = 4096, H = 2048, NF = 5;
    const long n = (long)W * H;            // doubles per field
    std::vector<double *> s(NF), d(NF);
    double *sall, *dall;
    const long slack = 1 << 20;            // doubles of slack per array for the skew experiments
    CK(hipMalloc(&sall, sizeof(double) * (n + slack) * NF)); CK(hipMalloc(&dall, sizeof(double) * (n + slack) * NF));
    CK(hipMemset(sall, 0x11, sizeof(double) * (n + slack) * NF));
    for (int f = 0; f < NF; ++f) { s[f] = sall + f * n; d[f] = dall + f * n; }
    double **ds, **dd;
    CK(hipMalloc(&ds, sizeof(double *) * NF)); CK(hipMalloc(&dd, sizeof(double *) * NF));
    CK(hipMemcpy(ds, s.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
    CK(hipMemcpy(dd, d.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * sizeof(double) * n * NF;
    auto time = [&](const char *name, auto &&launch) -> int {
        for (int w = 0; w < 20; ++w) launch();
        CK(hipEventRecord(e0, 0));
        const int reps = 200;
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-62s %7.3f ms  %6.2f TB/s\n", name, ms / reps, bytes * reps / (ms * 1e-3) / 1e12);
        return 0;
    };
    for (int blocks : {2048, 8192, 32768}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride copy,  4 B per lane, %5d workgroups", blocks);
        time(nm, [&] { hipLaunchKernelGGL(copy_k<float>, dim3(blocks), dim3(256), 0, 0, (float *)dall, (const float *)sall, n * NF * 2); });
        snprintf(nm, sizeof nm, "grid-stride copy,  8 B per lane, %5d workgroups", blocks);
        time(nm, [&] { hipLaunchKernelGGL(copy_k<double>, dim3(blocks), dim3(256), 0, 0, dall, (const double *)sall, n * NF); });
        snprintf(nm, sizeof nm, "grid-stride copy, 16 B per lane, %5d workgroups", blocks);
        time(nm, [&] { hipLaunchKernelGGL(copy_k<double2>, dim3(blocks), dim3(256), 0, 0, (double2 *)dall, (const double2 *)sall, n * NF / 2); });
    }
    // the same with the arrays skewed against each other: array f starts skew * f (src) / skew * (f + NF) (dst)
    // bytes past its power-of-two-aligned position
    for (long skew : {0L, 256L, 512L, 4096L, 4096L + 256L, 65536L + 4096L + 256L, 1048576L + 4096L + 256L}) {
        for (int f = 0; f < NF; ++f) { s[f] = sall + f * n + skew / 8 * f; d[f] = dall + f * n + skew / 8 * (f + NF); }
        CK(hipMemcpy(ds, s.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
        CK(hipMemcpy(dd, d.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
        char nm[128];
        snprintf(nm, sizeof nm, "strip march, 47 rows per band, arrays skewed by %8ld B each", skew);
        const int bands = (H + 46) / 47;
        time(nm, [&] { hipLaunchKernelGGL((strip_k<NF, 1>), dim3((W / 64) * bands), dim3(64), 0, 0, dd, (const double *const *)ds, W, H, 47); });
    }
    for (int f = 0; f < NF; ++f) { s[f] = sall + f * n; d[f] = dall + f * n; }
    CK(hipMemcpy(ds, s.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
    CK(hipMemcpy(dd, d.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
    for (int rpb : {47}) {
        char nm[128];
        snprintf(nm, sizeof nm, "strip march (sw2d_fused pattern), 5 + 5 arrays, %4d rows per band", rpb);
        const int bands = (H + rpb - 1) / rpb;
        time(nm, [&] { hipLaunchKernelGGL((strip_k<NF, 1>), dim3((W / 64) * bands), dim3(64), 0, 0, dd, (const double *const *)ds, W, H, rpb); });
        snprintf(nm, sizeof nm, "  the same, 2 adjacent strips per workgroup (128 threads), %4d rows", rpb);
        time(nm, [&] { hipLaunchKernelGGL((strip_k<NF, 2>), dim3((W / 128) * bands), dim3(128), 0, 0, dd, (const double *const *)ds, W, H, rpb); });
        snprintf(nm, sizeof nm, "  the same, 4 adjacent strips per workgroup (256 threads), %4d rows", rpb);
        time(nm, [&] { hipLaunchKernelGGL((strip_k<NF, 4>), dim3((W / 256) * bands), dim3(256), 0, 0, dd, (const double *const *)ds, W, H, rpb); });
    }
    {
        const int rpb = 47, bands = (H + rpb - 1) / rpb;
        const dim3 g((W / 64) * bands), b(64);
        const double *const *cs = (const double *const *)ds;
#define RUN(NAME, ...) time(NAME, [&] { hipLaunchKernelGGL((strip2_k<__VA_ARGS__>), g, b, 0, 0, dd, cs, W, H, rpb); })
        RUN("strip2 copy, D = 1 (the pattern above)", NF, 0, false, 1);
        RUN("strip2 copy, D = 1, nontemporal", NF, 0, true, 1);
        RUN("strip2 copy, D = 2 rows per iteration", NF, 0, false, 2);
        RUN("strip2 copy, D = 2, nontemporal", NF, 0, true, 2);
        RUN("strip2 copy, D = 4 rows per iteration", NF, 0, false, 4);
        printf("(read-only / write-only lines move HALF the bytes the TB/s column assumes)\n");
        RUN("strip2 read only, D = 1", NF, 1, false, 1);
        RUN("strip2 read only, D = 2", NF, 1, false, 2);
        RUN("strip2 read only, D = 1, nontemporal", NF, 1, true, 1);
        RUN("strip2 write only, D = 1", NF, 2, false, 1);
        RUN("strip2 write only, D = 1, nontemporal", NF, 2, true, 1);
#undef RUN
#define RUN3(NAME, ...) time(NAME, [&] { hipLaunchKernelGGL((strip3_k<__VA_ARGS__>), g, b, 0, 0, dd, cs, W, H, rpb); })
        RUN3("strip3: reads row by row, writes 1 row at a time", NF, 1, false, false);
        RUN3("strip3: reads row by row (nt), writes 1 row at a time", NF, 1, true, false);
        RUN3("strip3: reads row by row (nt), writes 2 rows at a time", NF, 2, true, false);
        RUN3("strip3: reads row by row (nt), writes 3 rows at a time", NF, 3, true, false);
        RUN3("strip3: reads row by row, writes 3 rows at a time", NF, 3, false, false);
        RUN3("strip3: reads row by row (nt), writes 4 rows at a time", NF, 4, true, false);
        RUN3("strip3: reads row by row (nt), writes 6 rows at a time", NF, 6, true, false);
        RUN3("strip3: reads row by row (nt), writes 8 rows at a time", NF, 8, true, false);
        RUN3("strip3: reads row by row (nt), writes 12 rows at a time", NF, 12, true, false);
        RUN3("strip3: reads row by row (nt), writes 8 rows at a time (nt)", NF, 8, true, true);
#undef RUN3
        {
            int step = 0;
            const double *const *csd = (const double *const *)dd;
            time("ping-pong steps, every step southwards", [&] {
                if (step++ & 1) hipLaunchKernelGGL((strip_dir_k<NF>), g, b, 0, 0, ds, csd, W, H, rpb, 0);
                else hipLaunchKernelGGL((strip_dir_k<NF>), g, b, 0, 0, dd, cs, W, H, rpb, 0);
            });
            for (int rr : {47, 24, 16}) {
                const dim3 g2((W / 64) * ((H + rr - 1) / rr));
                char nm[128];
                snprintf(nm, sizeof nm, "ping-pong steps, %d rows per band (%d waves, 3072 resident), same tile order", rr, (int)g2.x);
                step = 0;
                time(nm, [&] {
                    if (step++ & 1) hipLaunchKernelGGL((strip_dir_k<NF>), g2, b, 0, 0, ds, csd, W, H, rr, 0, 0);
                    else hipLaunchKernelGGL((strip_dir_k<NF>), g2, b, 0, 0, dd, cs, W, H, rr, 0, 0);
                });
                snprintf(nm, sizeof nm, "ping-pong steps, %d rows per band, tile order reversed every other step", rr);
                step = 0;
                time(nm, [&] {
                    const int rv = step & 1;
                    if (step++ & 1) hipLaunchKernelGGL((strip_dir_k<NF>), g2, b, 0, 0, ds, csd, W, H, rr, 0, rv);
                    else hipLaunchKernelGGL((strip_dir_k<NF>), g2, b, 0, 0, dd, cs, W, H, rr, 0, rv);
                });
            }
            step = 0;
            time("ping-pong steps, direction alternating", [&] {
                const int up = step & 1;
                if (step++ & 1) hipLaunchKernelGGL((strip_dir_k<NF>), g, b, 0, 0, ds, csd, W, H, rpb, up);
                else hipLaunchKernelGGL((strip_dir_k<NF>), g, b, 0, 0, dd, cs, W, H, rpb, up);
            });
        }
        // row pitch: arrays re-based at (n + slack) doubles apart so that a padded pitch fits
        for (int pad : {0, 16, 32, 512}) {   // H * pad <= slack
            const int P = W + pad;
            if ((long)H * pad > slack) {          // a pitch the arrays were not spaced for would run past them: say so, do not launch
                printf("strip march, row pitch = 4096 + %4d doubles: skipped (H * pad = %ld > slack = %ld doubles)\n", pad, (long)H * pad, slack);
                continue;
            }
            for (int f = 0; f < NF; ++f) { s[f] = sall + f * (n + slack); d[f] = dall + f * (n + slack); }
            CK(hipMemcpy(ds, s.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
            CK(hipMemcpy(dd, d.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
            char nm[128];
            snprintf(nm, sizeof nm, "strip march, row pitch = 4096 + %4d doubles", pad);
            time(nm, [&] { hipLaunchKernelGGL((strip_pitch_k<NF>), g, b, 0, 0, dd, cs, W, P, H, rpb); });
        }
        // strip-major tiles: the same bytes as [W / 64 strips][H rows][64 columns] -- a wave walks 47 x 512 contiguous
        // bytes of each array (strip_pitch_k with W = P = 64 and H * 64 rows)
        for (int f = 0; f < NF; ++f) { s[f] = sall + f * n; d[f] = dall + f * n; }
        CK(hipMemcpy(ds, s.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
        CK(hipMemcpy(dd, d.data(), sizeof(double *) * NF, hipMemcpyHostToDevice));
        for (int rows : {47, 94}) {
            const int HH = H * (W / 64), bb = (HH + rows - 1) / rows;
            char nm[128];
            snprintf(nm, sizeof nm, "strip-major tiles: a wave walks %d x 512 contiguous bytes per array", rows);
            time(nm, [&] { hipLaunchKernelGGL((strip_pitch_k<NF>), dim3(bb), b, 0, 0, dd, cs, 64, 64, HH, rows); });
        }
    }
    return 0;
}
