// Device scratch of the stand-alone operator entry points (gcm_sw2d_op, gcm_pe25d_op, gcm_pe1d_op,
// gcm_flux_limiter, gcm_pgf2d, gcm_advect2d ...): host arrays in, host arrays out, one call at a time.
// They used to hipMalloc / hipFree every operand of every call; now each calling thread keeps ONE
// grow-only arena and a call carves its operands out of it (256-byte aligned).  When an arena turns
// out too small the call takes extra blocks, and the end of the call replaces them by a single
// block of the combined size, so the next call of that shape allocates nothing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

namespace gcm {

struct DevArena {
    struct Block { char *p; size_t cap, used; };
    std::vector<Block> blocks;
    int device = -1;
    ~DevArena() { release(); }
    void release() {
        for (Block &b : blocks) (void)hipFree(b.p);
        blocks.clear();
    }
    void *take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev != device) {                       // the thread moved to another device: start afresh
            release();
            device = dev;
        }
        if (!blocks.empty()) {
            Block &b = blocks.back();
            if (b.used + bytes <= b.cap) {
                void *r = b.p + b.used;
                b.used += bytes;
                return r;
            }
        }
        size_t cap = bytes;
        // (doubling only while the arena is small: an operand of a production-size field is taken at its own size)
        if (!blocks.empty() && 2 * blocks.back().cap > cap && blocks.back().cap < kKeepCap) cap = 2 * blocks.back().cap;
        if (cap < ((size_t)1 << 20)) cap = (size_t)1 << 20;
        void *d = nullptr;
        if (hipMalloc(&d, cap) != hipSuccess) {
            if (cap == bytes || hipMalloc(&d, bytes) != hipSuccess) return nullptr;
            cap = bytes;
        }
        blocks.push_back(Block{(char *)d, cap, bytes});
        return d;
    }
    // end of a call: everything is handed back; several blocks become one of their combined size -- unless that
    // is more than kKeepCap: a call on a production-size field (2880x1440x40 fp64: 1.3 GB per operand) does not pin
    // gigabytes of HBM for the rest of the thread's life, it allocates again next time
    static constexpr size_t kKeepCap = (size_t)256 << 20;
    void reset() {
        size_t held = 0;
        for (const Block &b : blocks) held += b.cap;
        if (held > kKeepCap) {
            release();
            return;
        }
        if (blocks.size() > 1) {
            size_t total = 0;
            for (const Block &b : blocks) total += b.cap;
            release();
            void *d = nullptr;
            if (hipMalloc(&d, total) == hipSuccess) blocks.push_back(Block{(char *)d, total, 0});
        }
        for (Block &b : blocks) b.used = 0;
    }
};

inline DevArena &thread_arena() {
    thread_local DevArena a;
    return a;
}

// the operands of ONE call: device doubles carved from the calling thread's arena, optionally
// filled from a host array (blocking copy); the arena is reset when the call ends
struct DevScratch {
    ~DevScratch() { thread_arena().reset(); }
    double *get(size_t n, const double *src = nullptr) {
        void *d = thread_arena().take(n * sizeof(double));
        if (!d) return nullptr;
        if (src && hipMemcpy(d, src, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return (double *)d;
    }
};

}  // namespace gcm
