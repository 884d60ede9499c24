// Micro-benchmark 2: cost of a GEMM-tile epilogue's global stores on one CU while sibling waves stream LDS-DMA loads.
// 512-thread WG: waves 0..3 = loaders (global_load_lds dwordx4, as the GEMM main loop), waves 4..7 = storers
// (64 KiB per "epilogue", 16-B stores, rows of `row_bytes` contiguous bytes `ld` bytes apart).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__global__ __launch_bounds__(512) void k(const char* src, char* out, long long* cyc, int row_bytes, long ld, int reps, int loaders_on, int storer_waves, int nt, int win_bytes, int throttle) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ int done;
    if (tid == 0) done = 0;
    __syncthreads();
    if (wave < 4) {
        if (!loaders_on) return;
        // stream 1-KiB pieces from a 2-MiB window per WG (L2 resident after the first pass)
        const char* base = src + (long)blockIdx.x * win_bytes;
        long off = wave * 8192;
        while (__atomic_load_n(&done, __ATOMIC_RELAXED) < storer_waves) {
#pragma unroll
            for (int j = 0; j < 8; ++j) glds16(base + ((off + j * 1024) & (win_bytes - 1)) + lane * 16, smem + wave * 8192 + j * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            off += 32768;
            if (throttle) __builtin_amdgcn_s_sleep(32);
        }
    } else if (wave - 4 < storer_waves) {
        const int sw = wave - 4;
        const int lanes_per_row = row_bytes / 16;          // 2, 4, 8 or 64
        const int lpr_sh = __builtin_ctz(lanes_per_row);
        const int rows_per_instr = 64 >> lpr_sh;
        const int cpt_sh = __builtin_ctz(256 / row_bytes);
        const long bytes_per_wave = 65536 / storer_waves;
        const int instrs = bytes_per_wave / 1024;
        f4 v = {1.f * tid, 2.f, 3.f, 4.f};
        long long t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) {
            // tile r of this WG: 256 rows x 512 B region; this wave's strip
            char* tile = out + ((long)blockIdx.x * reps + r) * 256 * ld;
            for (int i = 0; i < instrs; ++i) {
                const int g = sw * instrs + i;                      // instruction index within the group epilogue
                const int row = g * rows_per_instr + (lane >> lpr_sh);  // rows of row_bytes
                // rows are laid out so that a 256-row tile has 65536/256 = 256 B per row: row_bytes chunks side by side
                const int trow = row >> cpt_sh, chunk = row & ((1 << cpt_sh) - 1);
                char* p = tile + (long)trow * ld + chunk * row_bytes + (lane & (lanes_per_row - 1)) * 16;
                if (nt) __builtin_nontemporal_store(v, (f4*)p); else *(f4*)p = v;
            }
            v[0] += 1.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t1 = __builtin_readcyclecounter();
        if (lane == 0) { cyc[blockIdx.x * 4 + sw] = t1 - t0; __atomic_fetch_add(&done, 1, __ATOMIC_RELAXED); }
    }
}
int main() {
    char *src, *out; long long* cyc;
    const int reps = 16;
    const long ld = 4608;
    hipMalloc(&src, 512L << 20); hipMemset(src, 1, 512L << 20);
    hipMalloc(&out, 256L * reps * 256 * ld + (1 << 20)); hipMalloc(&cyc, 256 * 4 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    std::vector<long long> h(1024);
    for (int grid : {1, 256})
    for (int loaders : {0, 1, 2, 3, 4})
    for (int sw : {4, 1})
    for (int nt : {0})
    for (int row_bytes : {32, 128, 256}) {
        const int win = (loaders == 1 || loaders == 3) ? (2 << 20) : (64 << 10);
        const int thr = loaders >= 3;
        for (int it = 0; it < 2; ++it) {
            hipMemset(cyc, 0, 256 * 4 * 8);
            k<<<grid, 512, 128 * 1024>>>(src, out, cyc, row_bytes, ld, reps, loaders, sw, nt, win, thr);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
        double avg = 0; int n = 0; for (int b = 0; b < grid; ++b) for (int w = 0; w < sw; ++w) { avg += h[b * 4 + w]; ++n; }
        avg /= n;
        printf("grid=%3d loaders=%d(1 hbm,2 l2,3 hbm-throttled,4 l2-throttled) storers=%d nt=%d row=%3dB: %.0f cycles per 64-KiB epilogue (%.1f B/clk/CU)\n", grid, loaders, sw, nt, row_bytes, avg / reps, 65536.0 * reps / avg);
    }
    return 0;
}
