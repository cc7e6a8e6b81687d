// Probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950 (the guide gives the effect for one layout only).
// Fills LDS with u16[i] = i, reads with several per-lane address patterns, prints what every lane received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, unsigned short* out) {
    __shared__ unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int a = addr[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    for (int pat = 0; pat < 4; ++pat) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, p = l & 15;
            if (pat == 0) a[l] = 4 * l;                                    // contiguous
            if (pat == 1) a[l] = g * 1024 + (p >> 2) * 64 + (p & 3) * 4;   // 4 rows x 16 cols, row stride 64 elements
            if (pat == 2) a[l] = g * 1024 + (p & 3) * 64 + (p >> 2) * 4;   // lane p -> row p&3, col block p>>2
            if (pat == 3) a[l] = g * 2048 + p * 128;                       // 16 rows x 4 cols (each lane its own row)
        }
        hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d_addr, d_out);
        std::vector<unsigned short> o(256);
        hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %5d ->", l, a[l]);
            for (int j = 0; j < 4; ++j) printf(" %5d", o[l * 4 + j]);
            printf("\n");
        }
    }
    return 0;
}
