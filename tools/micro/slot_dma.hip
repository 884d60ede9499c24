// Micro-benchmark 4: staggered read-slot / MFMA-slot schedule of gemm16_s256 with the LDS-DMA stream added.
//   nR = global_load_lds per wave at the head of the read slot, nM = per wave spread between the MFMAs of the MFMA slot.
//   pos: 0 = glds before the ds_reads, 1 = glds after the ds_reads, 2 = glds interleaved with the ds_reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
#define SB() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int NR, int NM, int POS, int ADDR = 0>
__global__ __launch_bounds__(512) void k(const char* src, long long* cyc, float* sink, int iters, int win) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = wave >> 2;
    for (int i = tid; i < 128 * 1024 / 4; i += 512) ((float*)smem)[i] = 0.001f * (i & 1023);
    __syncthreads();
    f16v acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    h8 fr[24];
    for (int j = 0; j < 24; ++j) fr[j] = *(const h8*)(smem + lane * 16 + j * 1024);
    const char* base = smem + (wave & 3) * 24 * 1024 + lane * 16;
    // GEMM-like: fragment row = lane&31 (128-B pitch), 16-B slot (2*ks + (lane>>5)) ^ ((row>>1)&7); group g reads its A half
    const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
    const char* gA = smem + grp * 16384 + frow * 128;                  // + mi*4096
    const char* gW = smem + 32768 + (wave & 3) * 8192 + frow * 128;      // + ni*4096
    unsigned stage = 0;
    char* dst = smem + 96 * 1024 + wave * 4096;          // DMA destination (not read: timing only)
    const char* g = src + (long)blockIdx.x * win + wave * 8192 + lane * 16;
    unsigned off = 0;
    auto dma = [&](int j) { glds16(g + ((off + j * 1024) & (win - 1)), dst + (j & 3) * 1024); };
    long long t0 = __builtin_readcyclecounter();
    auto body = [&]() {
        // ---- read slot ----
        if (POS == 0) { for (int j = 0; j < NR; ++j) dma(j); }
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            if (ADDR == 0) fr[j] = *(const h8*)(base + j * 1024);
            else {
                // order as in gemm16_s256: per ks: wf[0], wf[1], xf[0..3]  -> fr index: W: (ni)*4+ks, A: 8 + mi*4 + ks
                const int ks = j / 6, w = j % 6;
                const int slot = ((2 * ks + fh) ^ fsw) << 4;
                if (w < 2) fr[w * 4 + ks] = *(const h8*)(gW + stage + w * 4096 + slot);
                else fr[8 + (w - 2) * 4 + ks] = *(const h8*)(gA + stage + (w - 2) * 4096 + slot);
            }
            if (POS == 2 && NR > 0 && (j % (24 / (NR > 0 ? NR : 1))) == 0 && j / (24 / (NR > 0 ? NR : 1)) < NR) { FENCE(); dma(j / (24 / (NR > 0 ? NR : 1))); FENCE(); }
        }
        if (POS == 1) { FENCE(); for (int j = 0; j < NR; ++j) dma(j); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SB();
        // ---- MFMA slot ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[(a & 1) * 4 + ks], fr[8 + (a >> 1) * 4 + ks], acc[a], 0, 0, 0);
                if (NM > 0 && ((ks * 8 + a) % (32 / (NM > 0 ? NM : 1))) == 0) { FENCE(); dma(NR + (ks * 8 + a) / (32 / (NM > 0 ? NM : 1))); FENCE(); }
            }
        }
        FENCE();
        if (NM == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NM == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (NM == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (NM == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        off += 65536;
        stage ^= 65536;
        SB();
    };
    if (grp == 0) { for (int it = 0; it < iters; ++it) body(); SB(); }
    else { SB(); for (int it = 0; it < iters; ++it) body(); }
    long long t1 = __builtin_readcyclecounter();
    f16v sv = acc[0];
    for (int a = 1; a < 8; ++a) sv += acc[a];
    float s = sv[0] + sv[5];
    if (s == 12345.f) for (int j = 0; j < 24; ++j) s += (float)fr[j][0];
    sink[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
const char* src; long long* cyc; float* sink;
template <int NR, int NM, int POS, int ADDR = 0>
void run(int grid, int win) {
    hipFuncSetAttribute((const void*)k<NR, NM, POS, ADDR>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    const int iters = 4000;
    std::vector<long long> h(2048);
    float ms = 0;
    for (int it = 0; it < 2; ++it) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<NR, NM, POS, ADDR><<<grid, 512, 136 * 1024>>>(src, cyc, sink, iters, win);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h.data(), cyc, grid * 64, hipMemcpyDeviceToHost);
    double c = 0; for (int b = 0; b < grid; ++b) c += h[b * 8];
    c /= grid * (double)iters;
    printf("grid=%3d win=%7d nR=%d nM=%d pos=%d addr=%d: %.0f cycles per k-step (MFMA-bound = 2048), %.3f GHz, %.0f TFLOP/s\n", grid, win, NR, NM, POS, ADDR, c, c * iters / (ms * 1e6),
           (double)grid * 8 * 32 * 32768.0 * iters / (ms * 1e-3) / 1e12);
}
int main() {
    char* s; hipMalloc(&s, 1L << 30); hipMemset(s, 1, 1L << 30); src = s;
    hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    for (int grid : {1, 256})
    for (int win : {65536}) {
        run<0, 0, 0>(grid, win);
        run<0, 0, 0, 1>(grid, win);
        run<0, 8, 1, 1>(grid, win);
        run<8, 0, 1>(grid, win);
        run<6, 2, 1>(grid, win);
        run<4, 4, 1>(grid, win);
        run<2, 6, 1>(grid, win);
        run<0, 8, 1>(grid, win);
        run<4, 0, 1>(grid, win);
        run<0, 4, 1>(grid, win);
    }
    return 0;
}
