// slice_stream_probe.hip -- does a BatchNorm-style streaming pass keep its bandwidth when a block owns a 64-CHANNEL SLICE of the
// [R, C] bf16 activation (128-byte row segments, rows C * 2 bytes apart) instead of whole rows?  (docs/history/DESIGN_r1-r4.md section 7: a channel-
// sliced block map would let the apply pass sum the few partials of its own slice and drop the 208 `final` launches of a step.)
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/slice_probe tools/hip/slice_stream_probe.hip && /tmp/slice_probe
// Two passes per map: read-only column sums (the statistics / reduce pass) and read + write (the apply pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct __attribute__((aligned(16))) U4 { unsigned x, y, z, w; };

__device__ inline float sum8(const U4& u) {
    return __uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u) + __uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u) +
           __uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u) + __uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u);
}

// row map (what bnorm.hip does): a block covers all channels of its row range; thread = 8 channels, 256 / (C / 8) row lanes
template <bool WRITE>
__global__ __launch_bounds__(256) void row_map(const U4* __restrict__ x, U4* __restrict__ y, long long R, int C, int rows_per_block, float* out) {
    const int tpr = C >> 3, tprb = tpr < 256 ? tpr : 256, rpp = 256 / tprb;
    const int rsub = threadIdx.x / tprb, c8 = threadIdx.x % tprb;
    const long long rb = (long long)blockIdx.x * rows_per_block, re = min(R, rb + rows_per_block);
    float acc = 0.f;
    for (long long r = rb + rsub; r < re; r += rpp) {
        const U4 u = x[r * tpr + c8];
        acc += sum8(u);
        if (WRITE) y[r * tpr + c8] = u;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// slice map: a block owns 64 channels (8 threads) x 32 row lanes of its row range
template <bool WRITE>
__global__ __launch_bounds__(256) void slice_map(const U4* __restrict__ x, U4* __restrict__ y, long long R, int C, int rows_per_block, float* out) {
    const int tpr = C >> 3;
    const int c8 = blockIdx.y * 8 + (threadIdx.x & 7), rsub = threadIdx.x >> 3;
    const long long rb = (long long)blockIdx.x * rows_per_block, re = min(R, rb + rows_per_block);
    float acc = 0.f;
    for (long long r = rb + rsub; r < re; r += 32) {
        const U4 u = x[r * tpr + c8];
        acc += sum8(u);
        if (WRITE) y[r * tpr + c8] = u;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <class P, class L>
static float time_ms(P pre, L launch, int iters) {                // the flush is outside the timed interval
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    pre(); launch();
    hipDeviceSynchronize();
    float tot = 0.f;
    for (int i = 0; i < iters; ++i) {
        pre();
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        tot += ms;
    }
    return tot / iters;
}

int main() {
    const long long shapes[4][2] = {{802816, 256}, {200704, 512}, {50176, 1024}, {50176, 256}};
    float* out;
    hipMalloc(&out, 64);
    for (auto& sh : shapes) {
        const long long R = sh[0];
        const int C = (int)sh[1];
        const size_t bytes = (size_t)R * C * 2;
        U4 *x, *y, *flush;
        hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&flush, 512u << 20);
        hipMemset(x, 0, bytes);
        auto cold = [&] { hipMemsetAsync(flush, 1, 512u << 20, 0); };        // push x out of the Infinity Cache between launches
        for (int wr = 0; wr < 2; ++wr) {
            const int nblk = 768;
            const int rpb = (int)((R + nblk - 1) / nblk);
            const float t_row = time_ms(cold, [&] { if (wr) row_map<true><<<nblk, 256>>>(x, y, R, C, rpb, out); else row_map<false><<<nblk, 256>>>(x, y, R, C, rpb, out); }, 10);
            const int ny = C / 64, nx = (768 + ny - 1) / ny;
            const int rpb2 = (int)((R + nx - 1) / nx);
            const float t_sl = time_ms(cold, [&] { if (wr) slice_map<true><<<dim3(nx, ny), 256>>>(x, y, R, C, rpb2, out); else slice_map<false><<<dim3(nx, ny), 256>>>(x, y, R, C, rpb2, out); }, 10);
            const float t_cold = 0.f;
            const double gb = bytes * (wr ? 2.0 : 1.0) / 1e9;
            printf("{\"R\": %lld, \"C\": %d, \"pass\": \"%s\", \"MB\": %.0f, \"row_map_us\": %.1f, \"slice_map_us\": %.1f, \"row_map_TBps\": %.2f, \"slice_map_TBps\": %.2f}\n",
                   R, C, wr ? "read+write" : "read", bytes / 1e6, (t_row - t_cold) * 1e3, (t_sl - t_cold) * 1e3, gb / (t_row - t_cold), gb / (t_sl - t_cold));
        }
        hipFree(x); hipFree(y); hipFree(flush);
    }
    return 0;
}
