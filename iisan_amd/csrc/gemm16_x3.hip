// Split-operand GEMM with SHARED operand planes (round 6): the product behind split.hip,
//     C = (Ah·Bh^T + Ah·Bl^T + Al·Bh^T) / (sA sB)        (x·s = hi + lo in fp16, split.hip),
// used to run as ONE 16-bit GEMM over K' = 3K on images A' = [Ah | Ah | Al], B' = [Bh | Bl | Bh] through gemm16_kernel<F16, EPI_F32>:
// six operand tiles staged per 64 contraction steps for three tile products, Ah and Bh twice.  Every LDS-DMA-fed GEMM on this part is bound by
// operand delivery, and Versa's dim-align product was outright traffic-bound (422 MB per launch at 5.8 TB/s).  Here the images hold
// each plane ONCE — A2 = [Ah | Al], B2 = [Bh | Bl], rows of 2 kp elements — and a K-step stages the four 128 x 32 tiles Ah, Al, Bh, Bl and runs the
// three products from them: a third fewer operand bytes (half for an operand whose lo plane is all zero, e.g. taps cached in fp16: its lo tile is
// neither staged nor multiplied), and the split passes write two planes instead of three.
//   * 128 x 128 output tile per 256-thread workgroup (2 x 2 waves, 64 x 64 per wave = 4 x 4 fragments of v_mfma_f32_16x16x32_f16), BK = 32:
//     4 tiles x 8 KiB per stage, two stages = 64 KiB: two workgroups per CU, as the kernel it replaces;
//   * LDS rows are 64 bytes; 16-byte slot s of row r lives at s ^ ((-(r >> 2)) & 3): the 16-lane groups of a ds_read_b128 of a 16-row x 32-deep
//     fragment (lane -> row lane & 15, slot lane >> 4) cover all 64 banks once (tools/micro/qkv_gate.hip);
//   * operand swap, W-row permutation, fp32 epilogue (scales, bias / residual, split-K partials) are those of gemm16_kernel<F16, EPI_F32>;
//   * summation order: per K-step hh, then hl, then lh — fixed, bit-reproducible (it differs from the K'-concatenated order of the old route in
//     the last bits: the parity tests hold the product to the fp32 definition, not to the old route).
#include "common.h"

namespace {

constexpr int XBM = 128, XBK = 32;
constexpr int XTILE = XBM * XBK * 2;          // 8 KiB: an A-operand tile (128 rows x 64 B)

__device__ __forceinline__ int nperm_x(int q) { return (q & ~31) + 8 * ((q & 15) >> 2) + 4 * ((q >> 4) & 1) + (q & 3); }

// NB: 32-column groups per wave — 2: 128 x 128 tiles; 3: 128 x 192 tiles (round 6, second step: the chip holds 2 x CUs = 512 workgroups of this
// kernel, and the Cached fc products are 88 x 6 = 528 tiles of 128 x 128 — a full round and one of 16 workgroups; as 88 x 4 = 352 tiles of
// 128 x 192 they are ONE round)
template <int NB>
__global__ __launch_bounds__(256, 2) void gemm16_x3p_kernel(X3pArgs p) {
    constexpr int XBN = 64 * NB;                       // 128 / 192 columns
    constexpr int WTILE = XBN * XBK * 2;               // 8 / 12 KiB
    constexpr int XSTAGE = 2 * XTILE + 2 * WTILE;      // Ah | Al | Bh | Bl: 32 / 40 KiB
    __shared__ __attribute__((aligned(16))) char smem[2 * XSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int tiles_n = (p.N + XBN - 1) / XBN;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
    const int64_t m0 = (int64_t)tile_m * XBM;
    const int n0 = tile_n * XBN;
    const bool a_lo = p.lo_a == nullptr || p.lo_a[0] != 0, b_lo = p.lo_b == nullptr || p.lo_b[0] != 0;     // (block-uniform)
    const int64_t ld = (int64_t)2 * p.kp * 2;                            // bytes per image row
    const char* Ag = (const char*)p.A2 + m0 * ld;
    const char* Bg = (const char*)p.B2 + (int64_t)n0 * ld;
    // staging: wave w brings pieces 2w, 2w+1 (16 rows x 64 B each) of every tile; lane -> row-in-piece lane >> 2, physical slot lane & 3
    const int prow = lane >> 2, slog = (lane & 3) ^ ((-(prow >> 2)) & 3);
    int64_t a_off[2], b_off[NB];
#pragma unroll
    for (int j = 0; j < 2; ++j) a_off[j] = (int64_t)((2 * wave + j) * 16 + prow) * ld + slog * 16;
#pragma unroll
    for (int j = 0; j < NB; ++j) b_off[j] = (int64_t)nperm_x((NB * wave + j) * 16 + prow) * ld + slog * 16;
    const int64_t plane = (int64_t)p.kp * 2;                             // byte offset of the lo plane inside a row
    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * XSTAGE + (2 * wave) * 1024;
        char* sb = smem + buf * XSTAGE + 2 * XTILE + (NB * wave) * 1024;
        const int64_t kb = (int64_t)kt * (XBK * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(Ag + a_off[j] + kb, sa + j * 1024);
        if (a_lo) {
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(Ag + a_off[j] + kb + plane, sa + XTILE + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) glds16(Bg + b_off[j] + kb, sb + j * 1024);
        if (b_lo) {
#pragma unroll
            for (int j = 0; j < NB; ++j) glds16(Bg + b_off[j] + kb + plane, sb + WTILE + j * 1024);
        }
    };
    f4 acc[4][NB][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[b][nb][a] = (f4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 64 + ((fg ^ ((-(frow >> 2)) & 3)) << 4);
    const int xoff = (wave_m * 64) * 64 + foff, woff = (wave_n * 32 * NB) * 64 + foff;

    int nk = p.kp / XBK, k_first = 0;
    {
        const int per = (nk + (int)gridDim.y - 1) / (int)gridDim.y;
        k_first = (int)blockIdx.y * per;
        int k_end = k_first + per;
        if (k_end > nk) k_end = nk;
        nk = k_end - k_first;
        if (nk < 0) nk = 0;
    }
    if (nk > 0) stage(k_first, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto loop = [&](auto ALO, auto BLO) {
        constexpr bool AL = decltype(ALO)::value, BL = decltype(BLO)::value;
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(k_first + kt + 1, cur ^ 1);
            const char* s = smem + cur * XSTAGE;
            h8 xh[4], xl[4], wh[NB][2], wl[NB][2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                xh[b] = *(const h8*)(s + xoff + b * 1024);
                if (AL) xl[b] = *(const h8*)(s + XTILE + xoff + b * 1024);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    wh[nb][a] = *(const h8*)(s + 2 * XTILE + woff + (nb * 2 + a) * 1024);
                    if (BL) wl[nb][a] = *(const h8*)(s + 2 * XTILE + WTILE + woff + (nb * 2 + a) * 1024);
                }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        acc[b][nb][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nb][a], xh[b], acc[b][nb][a], 0, 0, 0);
                        if (BL) acc[b][nb][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nb][a], xh[b], acc[b][nb][a], 0, 0, 0);
                        if (AL) acc[b][nb][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nb][a], xl[b], acc[b][nb][a], 0, 0, 0);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    using std::true_type; using std::false_type;
    if (a_lo && b_lo) loop(true_type{}, true_type{});
    else if (b_lo) loop(false_type{}, true_type{});
    else if (a_lo) loop(true_type{}, false_type{});
    else loop(false_type{}, false_type{});

    // ---- epilogue (gemm16_kernel<F16, EPI_F32>): lane (j = lane & 15, g = lane >> 4) owns row m0 + wave_m*64 + b*16 + j, columns n .. n+7 ----
    const float ia = p.inv_a ? p.inv_a[0] : 1.f, ib = p.inv_b ? p.inv_b[0] : 1.f;
    const bool first = gridDim.y == 1;                 // split-K partial products: bias / residual are the reducer's
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t m = m0 + wave_m * 64 + b * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + wave_n * 32 * NB + nb * 32 + 8 * fg;
            if (n + 8 > p.N) continue;                 // N % 8 == 0: a lane's 8 columns are all in or all out
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[b][nb][0][r] * ia * ib;
                v[4 + r] = acc[b][nb][1][r] * ia * ib;
            }
            float* op = p.out + (int64_t)blockIdx.y * p.split_stride + m * p.ldo + n;
            if (p.bias && first) {
                const f4 b0 = *(const f4*)(p.bias + n), b1 = *(const f4*)(p.bias + n + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; }
            }
            if (p.resid && first) {
                const float* rp = p.resid + m * p.ldo + n;
                const f4 r0 = *(const f4*)rp, r1 = *(const f4*)(rp + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
            }
            *(f4*)op = (f4){v[0], v[1], v[2], v[3]};
            *(f4*)(op + 4) = (f4){v[4], v[5], v[6], v[7]};
        }
    }
}

}  // namespace

// tile width of a product: 192 columns where that turns a partly filled last round into none (x3p_tile_n, shared with the K-split choice of split.hip)
static int g_x3_bn = 0;                  // 128 / 192: this tile width wherever it divides N (sweeps); 0 = by the rule below
IISAN_DEV_KNOB(x3_force_bn, g_x3_bn);
int x3p_tile_n(int64_t M, int64_t N) {
    if (g_x3_bn == 128 || (g_x3_bn == 192 && N % 192 == 0)) return g_x3_bn;
    const int64_t slots = (int64_t)2 * iisan_cu_count(), rows = ceil_div(M, XBM);
    const int64_t t128 = rows * ceil_div(N, 128), t192 = rows * ceil_div(N, 192);
    if (N % 192 != 0) return 128;                      // (whole tiles only: the W image is padded to 128-row multiples)
    // rounds of equal-length workgroups, a 192-wide tile taking 1.5x the time of a 128-wide one
    const double c128 = (double)ceil_div(t128, slots), c192 = 1.5 * (double)ceil_div(t192, slots);
    return c192 < c128 ? 192 : 128;
}

int launch_gemm16_x3p(const X3pArgs& a, int ksplit, hipStream_t s) {
    IISAN_CHECK_SHAPE(a.M > 0 && a.N > 0 && a.kp > 0 && a.kp % 64 == 0 && a.N % 8 == 0 && a.ldo % 4 == 0, "gemm16_x3p: bad shape");
    IISAN_CHECK_SHAPE(ksplit >= 1 && (ksplit == 1 || a.split_stride > 0), "gemm16_x3p: split-K needs a partial-product buffer");
    const int bn = x3p_tile_n(a.M, a.N);
    const int64_t tiles = ceil_div(a.M, XBM) * ceil_div(a.N, bn);
    IISAN_CHECK_SHAPE(tiles < (1ll << 31), "gemm16_x3p: grid too large");
    if (bn == 192) {
        hipLaunchKernelGGL(gemm16_x3p_kernel<3>, dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(gemm16_x3p_kernel<2>, dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), 0, s, a);
    }
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}
