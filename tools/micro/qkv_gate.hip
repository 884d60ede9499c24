// Micro-benchmark (the "gate" of the fused QKV-projection + attention kernel): can the K loop of a workgroup that owns ONE
// (item, head) pair — C[208 x 192] = x_item[208 x 768] . W_head[192 x 768]^T, the tile a fused kernel would need so that Q / K / V of a head never
// leave the CU — hold the rate the production QKV product has WITHOUT its epilogue (1,290 - 1,340 TFLOP/s on 256 x 256 tiles)?
// The gate asks for >= 1,150 TFLOP/s (executed FLOPs, operands from L2 / MALL).  K loop only: accumulators are never stored in the timed launches.
//
// Tiling.  197 tokens pad to 208 = 13 blocks of 16 rows: the only MFMA row granularity that keeps the padding tax at 5.6 % (224 = 7 x 32 would be
// 13.7 %), so the product runs on v_mfma_f32_16x16x32_f16.  A head's 192 output columns (q | k | v x 64) = 12 blocks of 16.  Four waves (one per
// SIMD, the most favourable register budget: 512 per lane), wave w owns all 13 token blocks x column blocks 3w .. 3w + 2: 39 MFMAs per 32-deep K
// slice, 156 accumulator registers, and per slice 13 + 3 fragment reads (ds_read_b128) — each wave reads the whole x tile.
//   * operands L2 -> LDS by global_load_lds_dwordx4, FOUR stages of BK = 32: 208 + 192 rows of 64 bytes = 25,600 B per stage; a piece = 16 rows
//     x 64 B; 25 pieces per half-step (13 of x, 12 of W: wave w brings x pieces w, w+4, w+8 and 12 — the 13th by every wave, identical bytes — and W pieces w, w+4, w+8);
//   * the pieces of half-step h+3 are issued during half-step h, waited for at the end of h+1, certified by the barrier there; the fragments of
//     half-step h+1 are requested between the MFMAs of half-step h (two fragment register sets);
//   * LDS rows are 64 bytes; 16-byte slot s of row r lives at slot s ^ ((-(r >> 2)) & 3): the 16-lane groups of a ds_read_b128
//     ({0-3, 12-15, 20-27}, ...) of a 16-row x 32-deep fragment (lane -> row lane & 15, slot lane >> 4) cover all 64 banks once.
// Workgroups are persistent: workgroup b walks (item, head) pairs b, b + G, ... in head-minor order, so that the 12 heads of an item run on
// neighbouring workgroups at about the same time (x comes from L2 / MALL for 11 of them) and every XCD sees every W slice.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/qkv_gate.hip -o /tmp/qkv_gate && /tmp/qkv_gate [items]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstdint>
#include <type_traits>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int S = 197, SP = 208, HD = 64, HEADS = 12, D = HEADS * HD, K = 768;
constexpr int NB = 192;                       // columns of a head's q | k | v
constexpr int BK = 32, NST = 4;
constexpr int A_BYTES = SP * BK * 2, W_BYTES = NB * BK * 2, STAGE = A_BYTES + W_BYTES;      // 13,312 + 12,288 = 25,600
constexpr int NH = K / BK;                    // 24 half-steps per tile
constexpr int MB = SP / 16, WB = 3;           // 13 token blocks, 3 column blocks per wave

template <bool CHK>
__global__ __launch_bounds__(256, 1) void qkv_gate(const _Float16* __restrict__ X, const _Float16* __restrict__ W, float* __restrict__ out, int pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my = (pairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int H = my * NH;
    if (H <= 0) return;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;
    // DMA: lane -> row-in-piece lane >> 2, physical slot lane & 3; logical slot = physical ^ ((-(row >> 2)) & 3)
    const int prow = lane >> 2, pslot = lane & 3;
    const uint32_t voff = (uint32_t)(prow * K * 2 + ((pslot ^ ((-(prow >> 2)) & 3)) * 16));
    int i_kh = 0, i_t = blockIdx.x, i_h = 0;
    uint64_t i_gx = 0, i_gw = 0;
    auto i_tile = [&]() {
        const int item = i_t / HEADS, head = i_t - item * HEADS;
        i_gx = (uint64_t)(X + (size_t)item * S * K);
        i_gw = (uint64_t)(W + (size_t)head * HD * K);
    };
    i_tile();
    // piece j of the current issue half-step: j = 0..3 -> x piece wave + 4 j (j = 3: wave 0 only), 4..6 -> W piece wave + 4 (j - 4)
    auto piece = [&](int j) {
        uint64_t gb; uint32_t lds = smem_lds + (i_h & (NST - 1)) * STAGE;
        if (j < 4) {
            const int p = j < 3 ? wave + 4 * j : 12;          // (the 13th x piece is brought by every wave: identical bytes, no branch between the MFMAs)
            gb = i_gx + (uint64_t)p * 16 * K * 2 + (uint64_t)i_kh * (BK * 2);
            lds += p * 1024;
        } else {
            const int p = wave + 4 * (j - 4);                 // 0..11: q | k | v third p / 4, rows 16 (p % 4) of the head's 64
            gb = i_gw + ((uint64_t)(p >> 2) * D + (uint64_t)(p & 3) * 16) * K * 2 + (uint64_t)i_kh * (BK * 2);
            lds += A_BYTES + p * 1024;
        }
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds), "v"(voff), "s"(gb) : "memory");
    };
    auto i_next = [&]() {
        ++i_h;
        if (++i_kh == NH) { i_kh = 0; i_t += gridDim.x; i_tile(); }
    };
    f4 acc[MB][WB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < WB; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 64 + ((fg ^ ((-(frow >> 2)) & 3)) << 4);          // the same for every 16-row block
    const int offW = A_BYTES + wave * WB * 1024 + foff;
    h8 xf[2][MB], wf[2][WB];
    // fragment q of half-step h into register set buf: q = 0..2 -> W blocks, 3..15 -> x blocks
    auto load1 = [&](int h, int buf, int q) {
        const char* st = smem + (h & (NST - 1)) * STAGE;
        if (q < WB) wf[buf][q] = *(const h8*)(st + offW + q * 1024);
        else xf[buf][q - WB] = *(const h8*)(st + foff + (q - WB) * 1024);
    };
    auto issue_all = [&]() {                      // 7 pieces per wave and half-step
#pragma unroll
        for (int j = 0; j < 7; ++j) piece(j);
    };
    // prologue: half-steps 0, 1, 2 in flight; 0 and 1 landed and certified
    for (int p = 0; p < 3; ++p) {
        issue_all();
        if (i_h + 1 < H) i_next(); else ++i_h;
    }
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MB + WB; ++q) load1(0, 0, q);
    auto half = [&](int h, auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        // 39 MFMAs of half-step h; between them the 16 fragment reads of half-step h + 1 (stage certified at the end of h - 1) and this
        // wave's 6 / 7 pieces of half-step h + 3
#pragma unroll
        for (int n = 0; n < MB * WB; ++n) {
            const int i = n / WB, j = n - i * WB;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cur][j], xf[cur][i], acc[i][j], 0, 0, 0);
            FENCE();
            if (n < MB + WB) load1(h + 1, cur ^ 1, n);
            else if (n < MB + WB + 7) piece(n - (MB + WB));
            FENCE();
        }
        if (i_h + 1 < H) i_next(); else ++i_h;
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __syncthreads();
    };
    for (int h = 0; h < H; h += 2) {
        half(h, std::integral_constant<int, 0>{});
        half(h + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (CHK) {
        // D = A_op(W rows) x B_op(x rows)^T: lane (j = lane & 15, g = lane >> 4) holds token j of the block, output columns 4 g + r
        const int t = blockIdx.x;
        const int item = t / HEADS, head = t - item * HEADS;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < WB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = i * 16 + frow;                               // token row within the item's 208
                    const int n = (wave * WB + j) * 16 + 4 * fg + r;           // column within the head's 192
                    out[((size_t)t * SP + m) * NB + n] = acc[i][j][r];
                    (void)item; (void)head;
                }
    } else {
        float* o = out + (size_t)blockIdx.x * (256 * MB * WB * 4) + tid;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < WB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[((i * WB + j) * 4 + r) * 256] = acc[i][j][r];
    }
}

int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 1408;
    const int pairs = items * HEADS;
    const size_t xrows = (size_t)items * S + 16;           // the last item's pad rows read 11 rows past its end
    _Float16 *X, *W; float* out;
    CHECK(hipMalloc(&X, xrows * K * 2)); CHECK(hipMalloc(&W, (size_t)3 * D * K * 2));
    CHECK(hipMalloc(&out, (size_t)256 << 20));
    std::vector<_Float16> hX(xrows * K), hW((size_t)3 * D * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto& v : hX) v = (_Float16)rnd();
    for (auto& v : hW) v = (_Float16)(0.1f * rnd());
    CHECK(hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void*)qkv_gate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    CHECK(hipFuncSetAttribute((const void*)qkv_gate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)qkv_gate<false>));
    printf("qkv_gate: %d registers per thread, %zu bytes of scratch per thread, %d bytes of LDS\n", fa.numRegs, (size_t)fa.localSizeBytes, NST * STAGE);
    // 1. correctness: the first 36 (item, head) pairs, one per workgroup
    {
        const int tt = 36;
        hipLaunchKernelGGL(qkv_gate<true>, dim3(tt), dim3(256), NST * STAGE, 0, X, W, out, tt);
        CHECK(hipDeviceSynchronize());
        std::vector<float> ho((size_t)tt * SP * NB);
        CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int q = 0; q < 6000; ++q) {
            s = s * 1664525u + 1013904223u; const int t = (s >> 8) % tt;
            s = s * 1664525u + 1013904223u; const int m = (s >> 8) % S;
            s = s * 1664525u + 1013904223u; const int n = (s >> 8) % NB;
            const int item = t / HEADS, head = t % HEADS;
            const size_t xr = (size_t)item * S + m, wr = (size_t)(n / 64) * D + head * HD + (n % 64);
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)(float)hX[xr * K + k] * (double)(float)hW[wr * K + k];
            worst = std::max(worst, std::fabs(ref - ho[((size_t)t * SP + m) * NB + n]));
        }
        printf("check: worst |C - ref| over 6000 samples %.3e (values ~ %.2f)\n", worst, std::sqrt((double)K) * 0.058);
    }
    // 2. the K loop at production size: `items` x 12 pairs over 256 persistent workgroups
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(qkv_gate<false>, dim3(256), dim3(256), NST * STAGE, 0, X, W, out, pairs);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / 10 * 1e3;
        printf("%d items x 12 heads: %.1f us per launch; executed (208-row tiles) %.0f TFLOP/s, useful (197 rows) %.0f TFLOP/s   [production QKV product: 921-928 us with its epilogue]\n",
               items, us, 2.0 * pairs * SP * NB * K / (us * 1e-6) / 1e12, 2.0 * pairs * S * NB * K / (us * 1e-6) / 1e12);
    }
    return 0;
}
