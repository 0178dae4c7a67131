// What stream-ordering primitives cost between two small kernels on MI355X (the band step of GCM_PE25D is a
// chain of small kernels on two streams): hipcc --offload-arch=gfx950 -O2 sync_cost.hip -o sync_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
// does a wait for a stopEvent really wait for the kernel?  a: spin, then write n; b: read the cell at its very start
__global__ void spin_then_set(long long ticks, int *cell, int v) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) { __threadfence_system(); *cell = v; }
}
__global__ void read_at_start(const int *cell, int *out, int n) { if (threadIdx.x == 0) out[n] = *(volatile const int *)cell; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e0, e1, ev, done_b;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&done_b, hipEventDisableTiming));
    const int N = 400;
    const long long T = 500;   // 5 us at 100 MHz
    auto run = [&](const char *name, auto &&between) -> int {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            CK(hipEventRecord(e0, a));
            for (int n = 0; n < N; ++n) {
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, T);
                between(n);
            }
            CK(hipEventRecord(e1, a));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) printf("%-72s %7.2f us per iteration (kernel itself 5.0)\n", name, ms * 1e3 / N);
        }
        return 0;
    };
    // an event on stream b that completed long ago
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, T);
    CK(hipEventRecord(done_b, b));
    CK(hipStreamSynchronize(b));
    run("back-to-back kernels, one stream", [&](int) {});
    run("+ hipEventRecord (no-timing event) between", [&](int) { (void)hipEventRecord(ev, a); });
    run("+ hipStreamWaitEvent on an event of another stream, complete long ago", [&](int) { (void)hipStreamWaitEvent(a, done_b, 0); });
    run("+ two such waits", [&](int) { (void)hipStreamWaitEvent(a, done_b, 0); (void)hipStreamWaitEvent(a, done_b, 0); });
    run("+ record on a, stream b waits for it and runs a kernel (fork, not joined)", [&](int) {
        (void)hipEventRecord(ev, a); (void)hipStreamWaitEvent(b, ev, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, T); });
    run("+ fork to b (5 us kernel there) and join back before the next kernel", [&](int) {
        (void)hipEventRecord(ev, a); (void)hipStreamWaitEvent(b, ev, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, T);
        (void)hipEventRecord(done_b, b); (void)hipStreamWaitEvent(a, done_b, 0); });
    run("+ ping-pong: kernel on a, then kernel on b that waits for it, then back", [&](int) {
        (void)hipEventRecord(ev, a); (void)hipStreamWaitEvent(b, ev, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, T);
        (void)hipEventRecord(done_b, b); (void)hipStreamWaitEvent(a, done_b, 0); });
    // round 4: the kernel's OWN completion signal as the event (hipExtLaunchKernelGGL's stopEvent) instead of a
    // record packet behind it
    {
        auto run_ext = [&](const char *name, bool fork) -> int {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
                CK(hipEventRecord(e0, a));
                for (int n = 0; n < N; ++n) {
                    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, nullptr, ev, 0, T);
                    if (fork) { (void)hipStreamWaitEvent(b, ev, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, T); }
                }
                CK(hipEventRecord(e1, a));
                CK(hipEventSynchronize(e1));
                CK(hipStreamSynchronize(b));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 1) printf("%-72s %7.2f us per iteration (kernel itself 5.0)\n", name, ms * 1e3 / N);
            }
            return 0;
        };
        run_ext("kernels launched with a stopEvent (hipExtLaunchKernelGGL), no record", false);
        run_ext("the same, stream b waits for that stopEvent and runs a kernel (fork)", true);
    }
    {
        // correctness of the stopEvent dependency: 200 rounds, b must see the value a's kernel wrote at its END
        int *cell = nullptr, *out = nullptr;
        CK(hipMalloc(&cell, sizeof(int))); CK(hipMalloc(&out, 256 * sizeof(int)));
        CK(hipMemset(cell, 0, sizeof(int))); CK(hipMemset(out, 0xff, 256 * sizeof(int)));
        hipEvent_t back;
        CK(hipEventCreateWithFlags(&back, hipEventDisableTiming));
        for (int n = 0; n < 200; ++n) {
            hipExtLaunchKernelGGL(spin_then_set, dim3(1), dim3(64), 0, a, nullptr, ev, 0, 2000LL, cell, n + 1);     // 20 us
            CK(hipStreamWaitEvent(b, ev, 0));
            hipExtLaunchKernelGGL(read_at_start, dim3(1), dim3(64), 0, b, nullptr, back, 0, (const int *)cell, out, n);
            CK(hipStreamWaitEvent(a, back, 0));                   // (a's next write must not overtake b's read)
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        int h[200];
        CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int n = 0; n < 200; ++n) bad += h[n] != n + 1;
        printf("stopEvent dependency check: %d of 200 reads on b saw a stale value (0 = the wait is a real wait)\n", bad);
    }
    run("+ hipMemcpyAsync D2D 2 MB between", [&](int) {
        static void *p = nullptr, *q = nullptr;
        if (!p) { (void)hipMalloc(&p, 2 << 20); (void)hipMalloc(&q, 2 << 20); }
        (void)hipMemcpyAsync(q, p, 2 << 20, hipMemcpyDeviceToDevice, a); });
    // a wait whose event is NOT complete when the host enqueues it, but long complete when the waiting stream
    // gets there: b runs 5 us then records; a runs 30 us, waits, runs 5 us.  Everything is enqueued behind a
    // 2 ms spin on both streams, so the host is far ahead of the device.
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t evs[64];
        for (auto &evn : evs) CK(hipEventCreateWithFlags(&evn, hipEventDisableTiming));
        for (int with = 0; with < 2; ++with) {
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 200000LL);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 200000LL);
            CK(hipEventRecord(e0, a));
            for (int n = 0; n < 64; ++n) {
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, T);
                CK(hipEventRecord(evs[n], b));
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 6 * T);
                if (with) CK(hipStreamWaitEvent(a, evs[n], 0));
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, T);
            }
            CK(hipEventRecord(e1, a));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) printf("a: 30 us + 5 us kernels per iteration, %s a wait for b's event (done 25 us earlier): %7.2f us per iteration\n", with ? "with   " : "without", ms * 1e3 / 64);
        }
    }
    return 0;
}
