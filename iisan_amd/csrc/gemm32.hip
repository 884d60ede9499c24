// fp32 GEMM on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains at the fp32 vector
// rate) for everything TRAINABLE on the hot path: SANB down/up projections, fc_* heads, com_dense, SASRec
// projections/FFN and all their backward products (dX = dY·W, dW += dY^T·X).  These are <0.1 % of the step's
// FLOPs but carry the parity budget, hence fp32 end to end.
//
//   C[M,N] (=|+=) epilogue( op(A)[M,K] · op(B)[K,N] )
//   op(A): A stored [M,K] (default) or [K,M] (G32_TA)       op(B): B stored [N,K] (default, torch Linear.weight)
//                                                                     or [K,N] (G32_TB)
// 64x64x64 tile per 256-thread workgroup (4 waves, 32x32 each = 2x2 MFMA fragments); operands go HBM -> registers
// (16-byte loads) -> LDS (66-float rows: conflict-free fragment reads), and the registers of K-tile t+1 are filled
// while tile t is multiplied (the first version loaded a 16-wide K slice, synchronised, multiplied, synchronised: 48
// exposed HBM latencies for K = 768 — 80 us for a 1.1-GFLOP product, rocprofv3); up to 4 independent problems per launch
// (blockIdx.z) so the cv / text / mm towers of the side network go out together; split-K (blockIdx.y) with fp32
// atomics for the weight-gradient products whose M,N are tiny and K = number of item slots.
//
// Tried and dropped (round 2, MI355X): the same K loop on the 16-bit matrix cores with bf16x3 split operands (each fp32
// fragment element cut into three bf16 pieces by truncation after the LDS read, six MFMA terms, fp32 accumulate — 6/16 of
// the matrix cycles, all 175 GPU tests green at unchanged tolerances) ran the Cached step in 7.37 ms against 7.44 and the
// Versa step in 10.7 ms against 9.8: this kernel is bound by its staging (global -> registers -> LDS -> fragments, two
// barriers per K-tile), not by the matrix pipe, and the split added VALU work to it.
#include "common.h"

namespace {

constexpr int TN = 64, TK = 64, LD = 66;
// Operands whose memory layout is K-major (G32_TA / G32_TB: contraction index along the rows) are staged K-MAJOR in LDS
// too: a thread's 16-byte load (4 consecutive rows at one k) becomes ONE ds_write_b128 and the MFMA fragment read walks a
// row of the image (16 consecutive floats per quarter-wave: conflict-free with the k-row stride = 16 mod 32 banks).  The
// first version transposed while storing — 16 scalar LDS writes per operand and thread and K-tile, 4- to 8-way bank
// conflicts — which cost the weight-gradient products (both operands K-major) about as much as their MFMAs.
__host__ __device__ constexpr int ldt(int rows) { return rows == 16 ? 48 : rows + 16; }
__host__ __device__ constexpr int smem_floats(int rows) { return rows * LD > TK * ldt(rows) ? rows * LD : TK * ldt(rows); }
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr int G32_MAXP = 6;          // problems per launch (round 6: the two weight-gradient products of a SANB step x three towers share one)
struct Gemm32Batch {
    Gemm32Prob p[G32_MAXP];
};

// A 64 x 64 operand tile travels as 4 x float4 per thread.  Source stored [rows, K] (default): thread -> row
// (tid>>4)+16i, k (tid&15)*4; `trans` (source stored [K, rows], row index contiguous): k (tid>>4)+16i, rows (tid&15)*4..
template <int ROWS>
__device__ __forceinline__ void load_tile(f4 (&v)[ROWS / 16], const float* __restrict__ src, int ld, bool trans, int64_t row0,
                                          int64_t rows, int64_t k0, int64_t kend, int tid) {
    constexpr int TPR = ROWS / 4;                // trans: threads per K-row
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        v[i] = (f4){0.f, 0.f, 0.f, 0.f};
        if (!trans) {
            const int64_t row = row0 + (tid >> 4) + 16 * i, k = k0 + (tid & 15) * 4;
            if (row < rows && k < kend) {
                const float* p = src + row * ld + k;
                if (k + 3 < kend && (((uintptr_t)p) & 15) == 0) {
                    v[i] = *(const f4*)p;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < kend) v[i][e] = p[e];
                }
            }
        } else {
            const int64_t k = k0 + tid / TPR + (256 / TPR) * i, row = row0 + (tid % TPR) * 4;
            if (k < kend && row < rows) {
                const float* p = src + k * ld + row;
                if (row + 3 < rows && (((uintptr_t)p) & 15) == 0) {
                    v[i] = *(const f4*)p;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row + e < rows) v[i][e] = p[e];
                }
            }
        }
    }
}
template <int ROWS>
__device__ __forceinline__ void store_tile(float* S, const f4 (&v)[ROWS / 16], bool trans, int tid) {
    constexpr int TPR = ROWS / 4;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        if (!trans) {
            float* d = S + ((tid >> 4) + 16 * i) * LD + (tid & 15) * 4;         // 8-byte aligned (row stride 264 B)
            *(f2v*)d = (f2v){v[i][0], v[i][1]};
            *(f2v*)(d + 2) = (f2v){v[i][2], v[i][3]};
        } else {                                                                 // K-major image: row k, 4 consecutive operand rows
            *(f4*)(S + (tid / TPR + (256 / TPR) * i) * ldt(ROWS) + (tid % TPR) * 4) = v[i];
        }
    }
}

// structural flags (operand layouts, atomic accumulate) are compile time; epilogue flags are run time
// TMV = rows of the output tile (64 / 32 / 16).  A wave's MFMA chain is (TMV x 64 x K) / 4 waves long at 32 cycles per
// 16x16x4 step: the skinny products of the side network ([M,768] -> 64: one column tile, K = 768) took 40 us with 64-row
// tiles whatever M was (tools/gemm32_time.py) — 22 workgroups at M = 1408, each a 12-us dependent MFMA chain plus LDS
// traffic — so tiles shrink until the launch fills the chip.
// FAST: every tile of every problem of the launch is full, K-ranges are whole K-tiles and all operand rows are 16-byte
// aligned (checked on the host): the fetch is then one pointer per operand, advanced per K-tile, and unconditional
// 16-byte loads — the generic fetch spends ~160 VALU + 80 SALU instructions per K-tile on indices, bounds and alignment
// (PMC), beside 64 MFMAs.
template <int FLAGS, int TMV, bool FAST, int DEPTHV = 2>
__global__ __launch_bounds__(256) void gemm32_kernel(Gemm32Batch batch, int epi) {
    constexpr int TM = TMV;
    constexpr int WM = TMV >= 32 ? 2 : 1, WN = 4 / WM, FM = TMV / 16 / WM, FN = 4 / WN;
    __shared__ __attribute__((aligned(16))) float As_[smem_floats(TM)];
    __shared__ __attribute__((aligned(16))) float Bs_[smem_floats(TN)];
    float (*As)[LD] = (float (*)[LD])As_;           // row-major view: operand staging (not transposed) and the epilogue tile
    const Gemm32Prob& p = batch.p[blockIdx.z];
    const int tiles_n = (p.N + TN - 1) / TN;
    const int64_t tiles_m = (p.M + TM - 1) / TM;
    if ((int64_t)blockIdx.x >= tiles_m * tiles_n) return;
    const int64_t tile_m = blockIdx.x / tiles_n;
    const int tile_n = blockIdx.x - tile_m * tiles_n;
    const int64_t m0 = tile_m * TM;
    const int n0 = tile_n * TN;

    // K range of this split
    const int64_t ktiles = (p.K + TK - 1) / TK;
    const int64_t per = (ktiles + gridDim.y - 1) / gridDim.y;
    const int64_t kbeg = (int64_t)blockIdx.y * per * TK;
    int64_t kend = kbeg + per * TK;
    if (kend > p.K) kend = p.K;
    if (kbeg >= kend && blockIdx.y > 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 15, fk = lane >> 4;
    constexpr bool TA = (FLAGS & G32_TA) != 0, TB = (FLAGS & G32_TB) != 0;

    // KA independent accumulator sets over alternating K-slices: with one fragment per wave every MFMA would wait for
    // the previous one (same accumulator, ~40 cycles each)
    constexpr int KA = (FM * FN >= 4) ? 1 : 4 / (FM * FN);
    f4 acc[KA][FM][FN];
#pragma unroll
    for (int q = 0; q < KA; ++q)
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b) acc[q][a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // Three K-tiles in flight (register ring, statically indexed by unrolling x3): the products are tiny (one K-tile is
    // 64 MFMAs, ~0.3 us) and an HBM/L2 round trip is 1-2 us, so with one tile of look-ahead every iteration still
    // waited for most of a round trip.
    // (DEPTH 3 only for long-K weight-gradient launches (K >= 4096, chosen by the host), 2 elsewhere: the third stage costs
    //  32-64 registers and a wave of occupancy, which the short skinny products need more than a deeper ring.  Same-box A/B:
    //  depth 2 everywhere Versa 6.13 -> 5.83 ms but Cached 6.40 -> 6.45; depth 4 worse on both.)
    constexpr int DEPTH = DEPTHV;
    f4 ra[DEPTH][TM / 16], rb[DEPTH][4];
    // FAST: this thread's element of K-tile 0 and the strides between its ROWS/16 loads / between K-tiles
    constexpr int TPRA = TM / 4, TPRB = TN / 4;
    const float* pa = TA ? p.A + (kbeg + tid / TPRA) * p.lda + m0 + (tid % TPRA) * 4 : p.A + (m0 + (tid >> 4)) * p.lda + kbeg + (tid & 15) * 4;
    const float* pb = TB ? p.B + (kbeg + tid / TPRB) * p.ldb + n0 + (tid % TPRB) * 4 : p.B + (int64_t)(n0 + (tid >> 4)) * p.ldb + kbeg + (tid & 15) * 4;
    const int64_t sia = TA ? (int64_t)(256 / TPRA) * p.lda : (int64_t)16 * p.lda, sib = TB ? (int64_t)(256 / TPRB) * p.ldb : (int64_t)16 * p.ldb;
    const int64_t ska = TA ? (int64_t)TK * p.lda : TK, skb = TB ? (int64_t)TK * p.ldb : TK;
    const int64_t last_t = (kend - kbeg - 1) / TK;
    auto fetch = [&](int slot, int64_t k) {
        if constexpr (FAST) {
            // Unconditional: a K-tile past the end re-reads the last one (never staged).  Behind `if (k < kend)` every load
            // was a conditional one, the waitcnt pass lost count at the joins and drained the whole ring before each
            // stage (s_waitcnt vmcnt(7..0) in front of the stage's eight LDS writes although 16 younger loads were in
            // flight): the 3-deep ring worked as a 1-deep one.
            int64_t t = (k - kbeg) / TK;
            t = t < last_t ? t : last_t;
#pragma unroll
            for (int i = 0; i < TM / 16; ++i) ra[slot][i] = *(const f4*)(pa + t * ska + i * sia);
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[slot][i] = *(const f4*)(pb + t * skb + i * sib);
            return;
        }
        if (k < kend) {
            if constexpr (FAST) {
            } else {
                load_tile<TM>(ra[slot], p.A, p.lda, TA, m0, p.M, k, kend, tid);
                load_tile<TN>(rb[slot], p.B, p.ldb, TB, n0, p.N, k, kend, tid);
            }
        }
    };
    // FAST only: one of the NP = TM/16 + 4 loads of a K-tile.  Issued back to back ahead of the MFMAs, eight 16-byte loads per
    // thread cost the f32 product loop a quarter of its matrix rate even when every one hits L1 (tools/micro/f32_loop.hip:
    // 152 -> 115 TF; spread one by one among the MFMAs: 154) — so the fenced path below requests piece i after k-step 2 i + 1.
    constexpr int NP = TM / 16 + 4;
    // Addresses as wave-uniform base (SGPRs: operand + tile origin + K-tile) + one 32-bit lane offset per piece: between MFMAs a
    // 64-bit per-lane address add per load costs matrix time (the first spread version made the weight-gradient products slower).
    const float* const pa_u = TA ? p.A + kbeg * p.lda + m0 : p.A + m0 * p.lda + kbeg;
    const float* const pb_u = TB ? p.B + kbeg * p.ldb + n0 : p.B + (int64_t)n0 * p.ldb + kbeg;
    uint32_t voa[TM / 16], vob[4];
#pragma unroll
    for (int i = 0; i < TM / 16; ++i)
        voa[i] = (uint32_t)(((TA ? (int64_t)(tid / TPRA) * p.lda + (tid % TPRA) * 4 : (int64_t)(tid >> 4) * p.lda + (tid & 15) * 4) + i * sia) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        vob[i] = (uint32_t)(((TB ? (int64_t)(tid / TPRB) * p.ldb + (tid % TPRB) * 4 : (int64_t)(tid >> 4) * p.ldb + (tid & 15) * 4) + i * sib) * 4);
    auto fetch_piece = [&](int slot, int64_t k, int i) {
        int64_t t = (k - kbeg) / TK;
        t = t < last_t ? t : last_t;
        if (i < TM / 16) ra[slot][i] = *(const f4*)((const char*)(pa_u + t * ska) + voa[i]);
        else rb[slot][i - TM / 16] = *(const f4*)((const char*)(pb_u + t * skb) + vob[i - TM / 16]);
    };
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) fetch(u, kbeg + (int64_t)u * TK);
    for (int64_t kb = kbeg; kb < kend; kb += (int64_t)DEPTH * TK) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int64_t k0 = kb + (int64_t)u * TK;
            if (k0 < kend) {                          // wave-uniform
                store_tile<TM>(As_, ra[u], TA, tid);
                store_tile<TN>(Bs_, rb[u], TB, tid);
                __syncthreads();
                if constexpr (!FAST) fetch(u, k0 + (int64_t)DEPTH * TK);
                const int ksteps = (kend - k0 >= TK) ? TK / 4 : (int)((kend - k0 + 3) / 4);      // zero-filled tail
                auto kstep = [&](int ks0) {
#pragma unroll
                    for (int q = 0; q < KA; ++q) {
                        const int ks = ks0 + q;               // rows beyond the tail are zero-filled: harmless
                        float a[FM], b[FN];
#pragma unroll
                        for (int f = 0; f < FM; ++f)
                            a[f] = TA ? As_[(ks * 4 + fk) * ldt(TM) + wm * 16 * FM + f * 16 + fi] : As_[(wm * 16 * FM + f * 16 + fi) * LD + ks * 4 + fk];
#pragma unroll
                        for (int f = 0; f < FN; ++f)
                            b[f] = TB ? Bs_[(ks * 4 + fk) * ldt(TN) + wn * 16 * FN + f * 16 + fi] : Bs_[(wn * 16 * FN + f * 16 + fi) * LD + ks * 4 + fk];
#pragma unroll
                        for (int mf = 0; mf < FM; ++mf)
#pragma unroll
                            for (int nf = 0; nf < FN; ++nf)
                                acc[q][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mf], b[nf], acc[q][mf][nf], 0, 0, 0);
                    }
                };
                if (FAST || ksteps == TK / 4) {
                    // full tile: the fragments of k-slice s+1 are read before the MFMAs of slice s, fenced — left alone hipcc
                    // puts each slice's reads directly in front of its MFMAs behind an lgkmcnt(0) (the LDS latency exposed
                    // TK/4 times per tile); reading the whole tile's fragments up front costs 60 registers and a wave of occupancy
                    float ac[FM], bc[FN];
                    auto frag = [&](int ks, float (&a)[FM], float (&b)[FN]) {
#pragma unroll
                        for (int f = 0; f < FM; ++f)
                            a[f] = TA ? As_[(ks * 4 + fk) * ldt(TM) + wm * 16 * FM + f * 16 + fi] : As_[(wm * 16 * FM + f * 16 + fi) * LD + ks * 4 + fk];
#pragma unroll
                        for (int f = 0; f < FN; ++f)
                            b[f] = TB ? Bs_[(ks * 4 + fk) * ldt(TN) + wn * 16 * FN + f * 16 + fi] : Bs_[(wn * 16 * FN + f * 16 + fi) * LD + ks * 4 + fk];
                    };
                    frag(0, ac, bc);
#pragma unroll
                    for (int ks = 0; ks < TK / 4; ++ks) {
                        float an[FM], bn[FN];
                        frag(ks + 1 < TK / 4 ? ks + 1 : ks, an, bn);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mf = 0; mf < FM; ++mf)
#pragma unroll
                            for (int nf = 0; nf < FN; ++nf)
                                acc[ks % KA][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[mf], bc[nf], acc[ks % KA][mf][nf], 0, 0, 0);
                        if constexpr (FAST) {
                            if ((ks & 1) && (ks >> 1) < NP) fetch_piece(u, k0 + (int64_t)DEPTH * TK, ks >> 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int f = 0; f < FM; ++f) ac[f] = an[f];
#pragma unroll
                        for (int f = 0; f < FN; ++f) bc[f] = bn[f];
                    }
                } else {
                    for (int ks0 = 0; ks0 < ksteps; ks0 += KA) kstep(ks0);
                }
                __syncthreads();
            }
        }
    }

    const bool first_split = blockIdx.y == 0;
    float* const Cp = p.C + (int64_t)blockIdx.y * p.ksplit_stride;     // ksplit_stride != 0: raw partial product of this K-split
    // Epilogue.  In the MFMA layout a lane owns 4 rows x 1 column of a fragment: stores (and the residual / activation
    // reads) straight from it are 4-byte accesses, 256 B per wave instruction — the [M,64]·[64,768] products of the side
    // network (3 x 35 MB out, 35 MB residual in) ran at 1 TB/s, 5x their HBM time (rocprofv3, Cached step).  So the tile
    // goes through LDS once (As is free after the K loop) and every thread handles float4 runs of a row: 16-byte
    // accesses, 256 contiguous bytes per row.  Atomic accumulation (split-K weight gradients) keeps the direct path.
    constexpr bool ACC = (FLAGS & G32_ACCUM) != 0;
    const bool vec_ok = !ACC && n0 + TN <= p.N && (p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0 &&
                        (!p.resid || ((p.ldr & 3) == 0 && ((uintptr_t)p.resid & 15) == 0)) &&
                        (!p.act_src || ((uintptr_t)p.act_src & 15) == 0) && (!p.bias || ((uintptr_t)p.bias & 15) == 0);
    if (vec_ok) {            // block-uniform
#pragma unroll
        for (int mf = 0; mf < FM; ++mf)
#pragma unroll
            for (int nf = 0; nf < FN; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[0][mf][nf][r];
#pragma unroll
                    for (int q = 1; q < KA; ++q) v += acc[q][mf][nf][r];
                    As[wm * 16 * FM + mf * 16 + fk * 4 + r][wn * 16 * FN + nf * 16 + fi] = v;
                }
        __syncthreads();
        for (int idx = tid; idx < TM * 16; idx += 256) {
            const int row = idx >> 4, c4 = (idx & 15) * 4;
            const int64_t m = m0 + row;
            if (m >= p.M) continue;
            const int n = n0 + c4;
            const f2v lo = *(const f2v*)&As[row][c4], hi = *(const f2v*)&As[row][c4 + 2];
            f4 v = {lo[0], lo[1], hi[0], hi[1]};
            if (p.bias && first_split) v += *(const f4*)(p.bias + n);
            const int64_t ci = m * p.ldc + n;
            if (epi & G32_PREACT) *(f4*)((float*)p.act_src + ci) = v;
            f4 act = {0.f, 0.f, 0.f, 0.f};
            if (epi & (G32_MUL_RELU_MASK | G32_MUL_GELU_GRAD)) act = *(const f4*)(p.act_src + ci);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e];
                if (epi & G32_RELU) x = fmaxf(x, 0.f);
                if (epi & G32_GELU) x = gelu_erf(x);
                if (epi & G32_MUL_RELU_MASK) x = act[e] > 0.f ? x : 0.f;
                if (epi & G32_MUL_GELU_GRAD) x *= gelu_erf_grad(act[e]);
                if (epi & G32_DROPOUT) x *= drop_scale(p.drop.seed, p.drop.site, (uint64_t)(ci + e), p.drop.thr24, p.drop.inv_keep);
                v[e] = x;
            }
            if (p.resid && first_split) v += *(const f4*)(p.resid + m * p.ldr + n);
            *(f4*)(Cp + ci) = v;
        }
        return;
    }
#pragma unroll
    for (int mf = 0; mf < FM; ++mf)
#pragma unroll
        for (int nf = 0; nf < FN; ++nf) {
            const int n = n0 + wn * 16 * FN + nf * 16 + fi;
            if (n >= p.N) continue;
            const float bias = (p.bias && first_split) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + wm * 16 * FM + mf * 16 + fk * 4 + r;
                if (m >= p.M) continue;
                float v = acc[0][mf][nf][r];
#pragma unroll
                for (int q = 1; q < KA; ++q) v += acc[q][mf][nf][r];
                v += bias;
                if (epi & G32_PREACT) ((float*)p.act_src)[m * p.ldc + n] = v;
                if (epi & G32_RELU) v = fmaxf(v, 0.f);
                if (epi & G32_GELU) v = gelu_erf(v);
                if (epi & G32_MUL_RELU_MASK) v = p.act_src[m * p.ldc + n] > 0.f ? v : 0.f;
                if (epi & G32_MUL_GELU_GRAD) v *= gelu_erf_grad(p.act_src[m * p.ldc + n]);
                if (epi & G32_DROPOUT) v *= drop_scale(p.drop.seed, p.drop.site, (uint64_t)(m * p.ldc + n), p.drop.thr24, p.drop.inv_keep);
                if (p.resid && first_split) v += p.resid[m * p.ldr + n];
                if constexpr (ACC)
                    atomicAdd(p.C + m * p.ldc + n, v);
                else
                    Cp[m * p.ldc + n] = v;
            }
        }
}

// ---- weight-gradient products: both operands K-major, K = number of rows (item slots) ---------------------------------------
//   P_y[M, N] = A^T[M, Kr] · B[Kr, N]   over the K range Kr of split y;   A stored [K, M], B stored [K, N]   (G32_TA | G32_TB, raw
//   split-K partials for gemm32_reduce_kernel — the "+=" products dWu += dO^T·A, dWd += dU^T·F of every SANB and the fc layers)
// gemm32_kernel ran these at 65-75 TF: 64 ds_read_b32 and two barriers per K-tile, ring loads in a burst.  Here, per 64 x 64
// output tile and K-tile of 64 rows:
//   * both operand tiles are 64 K-rows of 64 floats: 16-byte loads (one piece per 8 MFMAs, two K-tiles ahead), ds_write_b128
//     into a double-buffered [k][68] image — one barrier per K-tile;
//   * the four waves split the K-TILE (wave w contracts rows 16 w ..+15 of it), each accumulates the whole 64 x 64 tile
//     (16 accumulators): per k-step one ds_read_b128 per operand — lane (i, kq) takes the four consecutive columns 4 i ..+3 of
//     row 16 w + 4 s + kq, which feed the four 16-row fragments {4 i' + e} — and 16 MFMAs: 8 LDS reads per 64 MFMAs;
//   * the four waves' tiles meet in LDS once, at the end (fixed order: bit-reproducible), 16-byte stores.
// Fragment e of A holds rows {4 i + e}, fragment e' of B columns {4 j + e'}: accumulator (e, e') register r of lane (j, g) is
// element (16 g + 4 r + e, 4 j + e') — four consecutive columns over e'.
__global__ __launch_bounds__(256, 2) void gemm32_dw_kernel(Gemm32Batch batch) {
    __shared__ __attribute__((aligned(16))) float sm[2][2][64 * 64];          // [buffer][operand][k][64]   (64 KB)
    const Gemm32Prob& p = batch.p[blockIdx.z];
    const int tiles_n = p.N >> 6;
    const int64_t tiles_m = p.M >> 6;
    if ((int64_t)blockIdx.x >= tiles_m * tiles_n) return;
    const int64_t tile_m = blockIdx.x / tiles_n;
    const int tile_n = (int)(blockIdx.x - tile_m * tiles_n);
    const int64_t m0 = tile_m * 64;
    const int n0 = tile_n * 64;
    const int64_t ktiles = p.K / TK;
    const int64_t per = (ktiles + gridDim.y - 1) / gridDim.y;
    const int64_t kt0 = (int64_t)blockIdx.y * per;
    int64_t kt1 = kt0 + per;
    if (kt1 > ktiles) kt1 = ktiles;
    if (kt0 >= kt1) return;                                         // (the reducer knows which splits own rows)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    // staging: thread -> K-rows (tid >> 4) + 16 q, columns 4 (tid & 15) ..+3; wave-uniform base + one 32-bit lane offset per piece
    const float* const a_u = p.A + kt0 * TK * p.lda + m0;
    const float* const b_u = p.B + kt0 * TK * p.ldb + n0;
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        voa[q] = (uint32_t)((((int64_t)(tid >> 4) + 16 * q) * p.lda + (tid & 15) * 4) * 4);
        vob[q] = (uint32_t)((((int64_t)(tid >> 4) + 16 * q) * p.ldb + (tid & 15) * 4) * 4);
    }
    const int64_t nt = kt1 - kt0;
    f4 st[2][8];                                                    // two K-tiles of pieces in flight: [slot][A 0..3 | B 0..3]
    auto piece = [&](int slot, int64_t t, int q) {
        t = t < nt ? t : nt - 1;                                    // past the end: the last tile again (never staged)
        if (q < 4) st[slot][q] = *(const f4*)((const char*)(a_u + t * TK * p.lda) + voa[q]);
        else st[slot][q] = *(const f4*)((const char*)(b_u + t * TK * p.ldb) + vob[q - 4]);
    };
#pragma unroll
    for (int q = 0; q < 8; ++q) piece(0, 0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) piece(1, 1, q);
    f4 acc[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[e][f] = (f4){0.f, 0.f, 0.f, 0.f};
    const int sw = (tid >> 4) * 64 + (tid & 15) * 4;                // staging write offset (+ 16 * 64 q)
    const int rd = (16 * wave + kq) * 64 + 4 * i16;                 // fragment read offset (+ 4 * 64 s)
    // column sums of the A operand (the bias gradient of the same dY): the workgroups of column tile 0 add up the pieces they stage
    const bool do_cs = p.colsum_a != nullptr && tile_n == 0;
    f4 csum = {0.f, 0.f, 0.f, 0.f};
    auto tile = [&](int64_t t, int slot) {
        float* bufA = sm[t & 1][0];
        float* bufB = sm[t & 1][1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(f4*)(bufA + sw + 16 * 64 * q) = st[slot][q];
            *(f4*)(bufB + sw + 16 * 64 * q) = st[slot][4 + q];
            if (do_cs) csum += st[slot][q];
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        f4 a = *(const f4*)(bufA + rd), b = *(const f4*)(bufB + rd);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f4 an = *(const f4*)(bufA + rd + 4 * 64 * (s < 3 ? s + 1 : s)), bn = *(const f4*)(bufB + rd + 4 * 64 * (s < 3 ? s + 1 : s));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[f], acc[e][f], 0, 0, 0);
                if (e & 1) {                                        // one piece of tile t + 2 after every 8th MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    piece(slot, t + 2, 2 * s + (e >> 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            a = an; b = bn;
        }
    };
    int64_t t = 0;
#pragma unroll 1
    for (; t + 2 <= nt; t += 2) { tile(t, 0); tile(t + 1, 1); }
    if (t < nt) tile(t, 0);
    // ---- the four waves' tiles meet in LDS (64 KB = 4 x 16 KB), summed in wave order -----------------------------------------
    __syncthreads();
    float* red = &sm[0][0][0] + wave * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *(f4*)(red + (16 * kq + 4 * r + e) * 64 + 4 * i16) = (f4){acc[e][0][r], acc[e][1][r], acc[e][2][r], acc[e][3][r]};
    __syncthreads();
    float* const Cp = p.C + (int64_t)blockIdx.y * p.ksplit_stride;
    const float* all = &sm[0][0][0];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = tid + 256 * q;                               // 1024 pieces of 16 bytes: row id >> 4, columns 4 (id & 15)
        const int o = (id >> 4) * 64 + (id & 15) * 4;
        const f4 v = (*(const f4*)(all + o) + *(const f4*)(all + 4096 + o)) + (*(const f4*)(all + 8192 + o) + *(const f4*)(all + 12288 + o));
        *(f4*)(Cp + (m0 + (id >> 4)) * p.ldc + n0 + (id & 15) * 4) = v;
    }
    if (do_cs) {            // sixteen threads hold partial sums of the same four columns: through LDS, fixed order, behind the C partial
        __syncthreads();
        float* cs = &sm[0][0][0];
        *(f4*)(cs + (tid >> 4) * 64 + (tid & 15) * 4) = csum;
        __syncthreads();
        if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += cs[r * 64 + tid];
            Cp[p.M * p.N + m0 + tid] = v;
        }
    }
}

// ---- fusion-fed down projection (separate SANB launches, forward) -----------------------------------------------------------
//   F = fuse(a, b, prev)   (type 0: g·a + (1-g)·prev;  type 1: prev + g·a + (1-g)·b;  not gated: plain sums)      [M, K]
//   U = F · W^T + bias,  A = act(U)                                                                                   [M, 64]
// fuse_fwd_kernel wrote F and the down projection read it back; here a lane loads the 16-byte pieces of a / prev / b that make up ITS
// piece of the MFMA operand (row j, k = k0 + 16 s + 4 g ..+3 — the contraction-index assignment of gemm32_k64_kernel, W [N, 64] case),
// forms F in registers, stores it (the only copy that ever travels) and multiplies.  64 rows per workgroup (wave w rows 16 w ..+15),
// the 64 x 64 weight tile through a double-buffered LDS image (one barrier per K-tile), all loads of the next K-tile requested one by
// one among the MFMAs of the current one.  Split-K (blockIdx.y, a K range = a column range of F) writes raw partials for
// gemm32_reduce_kernel, which applies bias / activation.
struct N64FProb {
    const float* a; const float* b; const float* prev; int64_t lda, ldb, ldp; const float* gate; int32_t type;
    float* F; const float* W; int32_t ldw; const float* bias; float* U; float* A; float* P; int64_t pstride; int64_t M; int32_t K;
};
struct N64FBatch { N64FProb p[3]; int32_t gelu; int32_t rpw; };      // rpw: rows per workgroup (<= 64; waves whose 16 rows start past it only stage)

__global__ __launch_bounds__(256, 2) void gemm32_n64f_kernel(N64FBatch batch) {
    __shared__ __attribute__((aligned(16))) float wl[2][64 * 68];
    const N64FProb& p = batch.p[blockIdx.z];
    const int64_t m0 = (int64_t)blockIdx.x * batch.rpw;
    if (m0 >= p.M) return;
    const int ktiles = p.K >> 6;
    const int per = (ktiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt0 = (int)blockIdx.y * per;
    int kt1 = kt0 + per;
    if (kt1 > ktiles) kt1 = ktiles;
    if (kt0 >= kt1) return;
    const int nt = kt1 - kt0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int64_t mj = m0 + 16 * wave + j;
    const int64_t mrow = mj < p.M ? mj : p.M - 1;                   // rows past M compute row M-1 again and store nothing
    const bool live = mj < p.M && 16 * wave + j < batch.rpw;
    const bool wave_on = 16 * wave < batch.rpw;                     // (wave-uniform) a wave past the tile only helps staging the weights
    const bool gated = p.gate != nullptr;
    const float gv = gated ? 1.0f / (1.0f + __expf(-p.gate[0] / 0.1f)) : 1.0f;
    const float ca = gated ? gv : 1.f, cb = gated ? 1.f - gv : 1.f;
    const bool has_b = p.type == 1;
    const float cp = p.prev ? (p.type == 0 ? cb : 1.f) : 0.f;
    // lane pieces: row mrow, floats k0 + 4 g + 16 s ..+3.  A missing prev / b reads `a` (weight 0 / never used): no branch between loads
    const float* arow = p.a + mrow * p.lda + 4 * g;
    const float* prow = p.prev ? p.prev + mrow * p.ldp + 4 * g : arow;
    const float* brow = has_b ? p.b + mrow * p.ldb + 4 * g : arow;
    float* frow = p.F + mrow * (int64_t)p.K + 4 * g;
    // weight-tile staging: thread -> feature (tid >> 4) + 16 q, floats 4 (tid & 15) ..+3 of the K-tile
    const float* wsrc = p.W + (int64_t)(tid >> 4) * p.ldw + (tid & 15) * 4;
    const int64_t wq = (int64_t)16 * p.ldw;
    const int sw = (tid >> 4) * 68 + (tid & 15) * 4;
    f4 wst[4], an[4], pn[4], bn[4];
    auto piece = [&](int t, int q) {                                 // one of the 16 loads of K-tile t (t past the end: the last again)
        const int kk = (kt0 + (t < nt ? t : nt - 1)) * 64;
        const int s = q & 3;
        if (q < 4) an[s] = *(const f4*)(arow + kk + 16 * s);
        else if (q < 8) pn[s] = *(const f4*)(prow + kk + 16 * s);
        else if (q < 12) { if (has_b) bn[s] = *(const f4*)(brow + kk + 16 * s); }
        else wst[s] = *(const f4*)(wsrc + kk + s * wq);
    };
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(0, q);
    f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f4){0.f, 0.f, 0.f, 0.f};
    const int lw = j * 68 + 4 * g;
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        float* buf = wl[t & 1];
        const int kk = (kt0 + t) * 64;
        f4 xa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *(f4*)(buf + sw + 16 * 68 * s) = wst[s];
            f4 f = cp * pn[s] + ca * an[s];
            if (has_b) f += cb * bn[s];
            xa[s] = f;
            if (live) *(f4*)(frow + kk + 16 * s) = f;
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        const float* wb = buf + lw;
        if (wave_on) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f4 w[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) w[f] = *(const f4*)(wb + 16 * f * 68 + 16 * s);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[f][e], xa[s][e], acc[f], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    piece(t + 1, 4 * s + e);                        // 16 pieces, one after every 4th MFMA
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int q = 12; q < 16; ++q) piece(t + 1, q);          // its share of the next weight tile
        }
    }
    if (!live) return;
    // epilogue from the fragments: piece f = columns 16 f + 4 g ..+3 of row j
    if (gridDim.y > 1) {
        float* pr = p.P + (int64_t)blockIdx.y * p.pstride + mrow * 64 + 4 * g;
#pragma unroll
        for (int f = 0; f < 4; ++f) *(f4*)(pr + 16 * f) = acc[f];
        return;
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int n = 16 * f + 4 * g;
        f4 v = acc[f] + *(const f4*)(p.bias + n);
        *(f4*)(p.U + mrow * 64 + n) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = batch.gelu ? gelu_erf(v[e]) : fmaxf(v[e], 0.f);
        *(f4*)(p.A + mrow * 64 + n) = v;
    }
}

// ---- K = 64 products with a wide N: the up projection of a SANB and its dF product -------------------------------------------
//   C[M, N] = A[M, 64] · op(W) (+ bias) (+ resid),   op(W): W stored [N, 64] (default) or [64, N] (TBV)
// The tiled kernel above gives these ONE K-tile per workgroup — operand round trip, 64 MFMAs, epilogue, exit — with nothing in
// flight meanwhile: Versa's text tower ([1408, 64] x [64 -> 8192] + F, 92 MB of HBM traffic, 1.5 GFLOP) took 67 us, 34 us even
// without the residual (tools/gemm32_wide.py), against ~18 us of HBM time and ~10 us of f32-matrix time.  Here a workgroup owns
// 64 rows x (64 * nblk) columns: its slice of W goes through LDS ONCE and is shared by the four waves, wave w keeps rows
// 16 w ..+15 of A in registers as the B operand of TRANSPOSED products (weights are the A operand, so a lane ends up with
// consecutive output columns of one row) and walks the slice in 64-column blocks: 16 ds_read_b128, the block's residual / bias
// reads, 64 MFMAs, 16-byte stores (64 contiguous bytes per row and instruction).  (A first version streamed W from L2 per
// 16-ROW tile, four waves side by side on different columns: 36 TF at any size — every 16 rows re-read all of W, 1.5 GB per
// launch at M = 11,264, and the chip delivers ~5 TB/s of such reads)
// Any assignment of the contraction index to (MFMA step, lane group g) works as long as both operands use the same one; it is
// chosen per layout so that every LDS read is a 16-byte one:
//   W [N, 64]:  LDS [col][68]; step (s, e) contracts k = 16 s + 4 g + e — lane (i, g) reads W[16 f + i][16 s + 4 g ..+3] for fragment f
//   W [64, N]:  LDS [k][cols + 4]; step s contracts k = 16 g + s — lane (i, g) reads W[16 g + s][4 q(i) ..+3], q(i) = 4 (i & 3) + (i >> 2):
//               the four values feed four fragments e whose feature rows are {4 q(i') + e}; a lane's results for register r
//               are then the columns 16 r + 4 g ..+3 of the block — contiguous over e.
// Exact fp32 (v_mfma_f32_16x16x4_f32), fixed summation order.  In place (C == resid) is fine: a lane reads exactly the elements
// it later writes.
struct K64Prob {
    const float* A; const float* W; const float* bias; const float* resid; float* C; int64_t M; int32_t N, lda, ldw, ldc, ldr;
    // GATE variant (the dF product of a gated SANB step with the fusion's backward folded in, launch_gemm32_k64_gate):
    //   dF = acc + resid;   dtheta += <dF, ga - go> g (1 - g) / 0.1   (go null: zeros);   C = dF * (scale_prev ? 1 - g : 1), if store
    const float* gate; const float* ga; const float* go; int64_t ldga, ldgo; float* dgate; int32_t scale_prev, store;
    float* d2; int64_t ldd2; int32_t d2_is_b;      // optional second output  d2 = (d2_is_b ? 1 - g : g) · dF  (Versa: gradient wrt the dim-aligned tap)
};
struct K64Batch { K64Prob p[4]; };

template <bool TBV, bool GATE = false>
__global__ __launch_bounds__(256, 2) void gemm32_k64_kernel(K64Batch batch, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    __shared__ float gred[4];
    const K64Prob& p = batch.p[blockIdx.z];
    const int c0 = (int)blockIdx.y * nblk * 64;                     // first column of the slice
    if ((int64_t)blockIdx.x * 64 >= p.M || c0 >= p.N) return;       // block-uniform
    int nb = (p.N - c0) >> 6;
    nb = nb < nblk ? nb : nblk;
    const int cols = nb * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    // ---- the slice of W -> LDS (all requests first, then the writes) ---------------------------------------------------------
    const int stride = TBV ? cols + 4 : 68;
    {
        const int per_row = TBV ? cols / 4 : 16;                    // 16-byte pieces per source row
        const int pieces = 16 * cols;                               // 64 x cols floats
        f4 t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int id = tid + 256 * q;
            id = id < pieces ? id : pieces - 1;
            const int r = id / per_row, c = id - r * per_row;
            t[q] = TBV ? *(const f4*)(p.W + (int64_t)r * p.ldw + c0 + 4 * c) : *(const f4*)(p.W + (int64_t)(c0 + r) * p.ldw + 4 * c);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int id = tid + 256 * q;
            if (id < pieces) {
                const int r = id / per_row, c = id - r * per_row;
                *(f4*)(wlds + r * stride + 4 * c) = t[q];
            }
        }
    }
    const bool has_r = p.resid != nullptr, has_b = p.bias != nullptr;
    float gpart = 0.f, gval = 0.f;
    if constexpr (GATE) gval = 1.0f / (1.0f + __expf(-p.gate[0] / 0.1f));
    const float cdp = (GATE && p.scale_prev) ? 1.f - gval : 1.f;
    const float cgo = (GATE && p.go) ? 1.f : 0.f;
    const float cd2 = GATE ? (p.d2_is_b ? 1.f - gval : gval) : 0.f;
    const float* wl = wlds + (TBV ? (16 * g) * stride + 4 * (4 * (j & 3) + (j >> 2)) : j * stride + 4 * g);
    __syncthreads();
    // ---- the workgroup's row tiles (64 rows each: wave w rows 16 w ..+15), the slice stays in LDS ------------------------------
#pragma unroll 1
    for (int64_t m0 = (int64_t)blockIdx.x * 64; m0 < p.M; m0 += (int64_t)gridDim.x * 64) {
    const int64_t mj = m0 + 16 * wave + j;
    const int64_t mrow = mj < p.M ? mj : p.M - 1;                   // rows past M compute row M-1 again and are not stored
    const bool live = mj < p.M;
    // the wave's operand: this lane's 16 contraction values of row j
    f4 xa[4];
    {
        const float* ar = p.A + mrow * p.lda + (TBV ? 16 * g : 4 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = *(const f4*)(ar + (TBV ? 4 * q : 16 * q));
    }
    const int ocol = c0 + 4 * g;                                    // + 64 * block + 16 * piece
    float* crow = p.C + mrow * p.ldc + ocol;
    // a missing residual / bias reads C / W instead and is discarded by a select: no branch between the loads
    const float* rrow = has_r ? p.resid + mrow * p.ldr + ocol : crow;
    const float* brow = has_b ? p.bias + ocol : p.W + 4 * g;
    const float* garow = GATE ? p.ga + mrow * p.ldga + ocol : nullptr;
    const float* gorow = GATE ? (p.go ? p.go + mrow * p.ldgo + ocol : p.ga + mrow * p.ldga + ocol) : nullptr;      // missing: read ga, weight 0
    f4 rn[4];                                                       // residual of the next block
#pragma unroll
    for (int q = 0; q < 4; ++q) rn[q] = *(const f4*)(rrow + 16 * q);
#pragma unroll 1
    for (int t = 0; t < nb; ++t) {
        f4 wc[16];
        {
            const float* wb = wl + (TBV ? 64 * t : 64 * t * stride);
#pragma unroll
            for (int q1 = 0; q1 < (TBV ? 16 : 4); ++q1)
#pragma unroll
                for (int q2 = 0; q2 < (TBV ? 1 : 4); ++q2)
                    wc[TBV ? q1 : 4 * q1 + q2] = *(const f4*)(wb + q1 * (TBV ? stride : 16 * stride) + 16 * q2);
        }
        f4 rv[4], bv[4];
        const int tn = t + 1 < nb ? t + 1 : t;                      // unconditional: the last block re-requests itself
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rv[q] = rn[q];
            bv[q] = *(const f4*)(brow + 64 * t + 16 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) rn[q] = *(const f4*)(rrow + 64 * tn + 16 * q);
        f4 gav[4], gov[4];                                          // GATE: this block's fusion operands, consumed after the MFMAs
        if constexpr (GATE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gav[q] = *(const f4*)(garow + 64 * t + 16 * q);
                gov[q] = *(const f4*)(gorow + 64 * t + 16 * q);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (f4){0.f, 0.f, 0.f, 0.f};
        if constexpr (TBV) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[s][e], xa[s >> 2][s & 3], acc[e], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[4 * f + s][e], xa[s][e], acc[f], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // piece q = columns 64 t + 16 q + 4 g ..+3 of row j:  TBV: register q of the four fragments;  else fragment q
            const f4 v = TBV ? (f4){acc[0][q], acc[1][q], acc[2][q], acc[3][q]} : acc[q];
            const f4 z = {0.f, 0.f, 0.f, 0.f};
            const f4 o = v + ((has_b ? bv[q] : z) + (has_r ? rv[q] : z));
            if constexpr (GATE) {
                if (live) {
                    const f4 d = gav[q] - cgo * gov[q];
                    gpart += (o[0] * d[0] + o[1] * d[1]) + (o[2] * d[2] + o[3] * d[3]);
                    if (p.store) *(f4*)(crow + 64 * t + 16 * q) = o * cdp;
                    if (p.d2) *(f4*)(p.d2 + mrow * p.ldd2 + ocol + 64 * t + 16 * q) = o * cd2;
                }
            } else {
                if (live) *(f4*)(crow + 64 * t + 16 * q) = o;
            }
        }
    }
    }
    if constexpr (GATE) {                       // one atomic per workgroup (same-address atomics serialise at ~12 ns)
        gpart = wave_sum(gpart);
        if (lane == 0) gred[wave] = gpart;
        __syncthreads();
        if (tid == 0) atomicAdd(p.dgate, ((gred[0] + gred[1]) + (gred[2] + gred[3])) * gval * (1.f - gval) / 0.1f);
    }
}

// column sums: out[n] += sum_m X[m][n]  (bias gradients); up to 4 problems per launch
struct ColsumBatch {
    const float* X[4];
    float* out[4];
    int64_t M[4];
    int32_t N[4], ld[4];
};
__global__ __launch_bounds__(256) void colsum_kernel(ColsumBatch b, int rows_per_block) {
    __shared__ float part[4][64];
    const int z = blockIdx.z;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > b.M[z]) r1 = b.M[z];
    float s = 0.f;
    if (n < b.N[z]) {
        // eight loads in flight per thread (one per iteration — what the plain loop compiles to — left each launch
        // latency-bound: 19 us for a 35 MB operand); fixed order: partial sums of the eight lanes combined at the end
        const float* xp = b.X[z] + n;
        const int64_t ld = b.ld[z];
        float part8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int64_t r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) part8[u] += xp[(r + 4 * u) * ld];
        }
        for (; r < r1; r += 4) s += xp[r * ld];
        s += ((part8[0] + part8[1]) + (part8[2] + part8[3])) + ((part8[4] + part8[5]) + (part8[6] + part8[7]));
    }
    part[rl][c] = s;
    __syncthreads();
    if (rl == 0 && n < b.N[z] && r0 < b.M[z]) atomicAdd(b.out[z] + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

// C = epilogue(sum_y P[y] + bias) (+ resid): the split-K partial products of up to G32_MAXP problems, fixed summation order
// ns[z] = the K-splits of problem z that own a non-empty K range: only those wrote a partial (a shorter K than its launch
// mates' leaves the rest of the scratch slots untouched — they are never read, so the scratch needs no zeroing)
struct ReduceBatch { Gemm32Prob p[G32_MAXP]; const float* P[G32_MAXP]; int64_t stride[G32_MAXP]; int32_t ns[G32_MAXP]; float* cs[G32_MAXP]; };     // cs: colsum_a of problem z (its partials follow the C partial of every split)
__global__ __launch_bounds__(256) void gemm32_reduce_kernel(ReduceBatch rb, int ks_launch, int epi) {
    const Gemm32Prob& p = rb.p[blockIdx.z];
    const int ks = ks_launch < rb.ns[blockIdx.z] ? ks_launch : rb.ns[blockIdx.z];
    const float* P = rb.P[blockIdx.z];
    const int64_t stride = rb.stride[blockIdx.z], total = p.M * p.N;
    if (rb.cs[blockIdx.z]) {         // column sums of the A operand: M values per split, fixed order (y ascending)
        // one value per thread over the first ceil(M / 256) workgroups.  (Until round 6 workgroup 0 walked all M values alone: at Versa's
        // M = 8192 that is 32 dependent rounds of loads by one workgroup — 33.8 us for a launch whose other workgroups finish in 6.)
        for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += (int64_t)gridDim.x * blockDim.x) {
            float v = 0.f;
            for (int y = 0; y < ks; ++y) v += P[(int64_t)y * stride + total + m];
            rb.cs[blockIdx.z][m] += v;
        }
    }
    if (epi == 0 && !p.bias && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (!p.resid || (p.ldr & 3) == 0)) {
        // plain sum (+ old C): the weight-gradient reductions.  16-byte accesses, eight partials in flight per thread; the
        // summation order is fixed (y ascending within each of the eight lanes of the unroll, lanes combined in order)
        const int64_t total4 = total / 4;
        const int n4 = p.N / 4;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t m = i / n4;
            const int n = (int)(i - m * n4) * 4;
            f4 acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = (f4){0.f, 0.f, 0.f, 0.f};
            int y = 0;
            for (; y + 8 <= ks; y += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += *(const f4*)(P + (int64_t)(y + u) * stride + 4 * i);
            }
            for (int u = 0; y < ks; ++y, ++u) acc[u] += *(const f4*)(P + (int64_t)y * stride + 4 * i);
            f4 v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            if (p.resid) v += *(const f4*)(p.resid + m * p.ldr + n);
            *(f4*)(p.C + m * p.ldc + n) = v;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / p.N;
        const int n = (int)(i - m * p.N);
        float v = P[i];
        for (int y = 1; y < ks; ++y) v += P[y * stride + i];
        if (p.bias) v += p.bias[n];
        const int64_t ci = m * p.ldc + n;
        if (epi & G32_PREACT) ((float*)p.act_src)[ci] = v;
        if (epi & G32_RELU) v = fmaxf(v, 0.f);
        if (epi & G32_GELU) v = gelu_erf(v);
        if (epi & G32_MUL_RELU_MASK) v = p.act_src[ci] > 0.f ? v : 0.f;
        if (epi & G32_MUL_GELU_GRAD) v *= gelu_erf_grad(p.act_src[ci]);
        if (epi & G32_DROPOUT) v *= drop_scale(p.drop.seed, p.drop.site, (uint64_t)ci, p.drop.thr24, p.drop.inv_keep);
        if (p.resid) v += p.resid[m * p.ldr + n];
        p.C[ci] = v;
    }
}

// registered by an executor for the duration of its call; thread_local: two host threads driving two streams must not see
// each other's scratch
thread_local float* g_scratch = nullptr;
thread_local size_t g_scratch_floats = 0;
constexpr int g_accum_via_scratch = 1;       // (weight-gradient split-K sums go through the executor's scratch; the atomics route's switch was retired in round 5)

// launch-shape heuristics (constants since round 5; tools/step_ab.py measured them): workgroups wanted before the row tile shrinks /
// before split-K stops adding slices
static constexpr int g_tm_thresh = 512, g_splitk_target = 1024;   // tools/step_ab.py (MI355X): row tiles shrink below 512 workgroups: Versa 9.88 -> 9.55 ms, Cached unchanged; split-K target 512 or 2048: no gain

template <int FLAGS>
int launch_flags(const Gemm32Batch& b, dim3 grid, int tm, int epi, bool fast, bool deep, hipStream_t s) {
    if constexpr ((FLAGS & (G32_TA | G32_TB)) == (G32_TA | G32_TB)) {
        if (deep && fast) {          // long-K weight gradients: three K-tiles in flight
            if (tm == 64) hipLaunchKernelGGL((gemm32_kernel<FLAGS, 64, true, 3>), grid, dim3(256), 0, s, b, epi);
            else if (tm == 32) hipLaunchKernelGGL((gemm32_kernel<FLAGS, 32, true, 3>), grid, dim3(256), 0, s, b, epi);
            else hipLaunchKernelGGL((gemm32_kernel<FLAGS, 16, true, 3>), grid, dim3(256), 0, s, b, epi);
            IISAN_LAUNCH_OK();
            return IISAN_OK;
        }
    }
    if (fast) {
        if (tm == 64) hipLaunchKernelGGL((gemm32_kernel<FLAGS, 64, true>), grid, dim3(256), 0, s, b, epi);
        else if (tm == 32) hipLaunchKernelGGL((gemm32_kernel<FLAGS, 32, true>), grid, dim3(256), 0, s, b, epi);
        else hipLaunchKernelGGL((gemm32_kernel<FLAGS, 16, true>), grid, dim3(256), 0, s, b, epi);
    } else if (tm == 64) hipLaunchKernelGGL((gemm32_kernel<FLAGS, 64, false>), grid, dim3(256), 0, s, b, epi);
    else if (tm == 32) hipLaunchKernelGGL((gemm32_kernel<FLAGS, 32, false>), grid, dim3(256), 0, s, b, epi);
    else hipLaunchKernelGGL((gemm32_kernel<FLAGS, 16, false>), grid, dim3(256), 0, s, b, epi);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

}  // namespace

// 1 (default): K = 64 products with a wide N and a plain epilogue take gemm32_k64_kernel; 0: the tiled kernel.  Test / bench knob.
static int g_use_k64 = 1, g_use_dw = 1;
static int64_t g_cnt_n64f = 0, g_cnt_k64 = 0, g_cnt_dw = 0;      // launches by kernel (route counters, common.h)
IISAN_DEV_COUNTER(gemm32_n64f, g_cnt_n64f);
IISAN_DEV_COUNTER(gemm32_k64, g_cnt_k64);
IISAN_DEV_COUNTER(gemm32_dw, g_cnt_dw);
IISAN_DEV_KNOB(gemm32_dw, g_use_dw);
static int g_dw_splits = 0;              // > 0: the split count of every gemm32_dw_kernel launch (sweeps)
IISAN_DEV_KNOB(gemm32_dw_splits, g_dw_splits);
IISAN_DEV_KNOB(gemm32_k64, g_use_k64);

void gemm32_set_scratch(float* ws, size_t floats) { g_scratch = ws; g_scratch_floats = floats; }

static int g_use_n64f = 1;
IISAN_DEV_KNOB(gemm32_n64f, g_use_n64f);
bool gemm32_n64f_ok(const N64FDesc* d, int n) {
    if (!g_use_n64f || n < 1 || n > 3) return false;
    for (int i = 0; i < n; ++i) {
        const N64FDesc& q = d[i];
        if (q.K < 256 || (q.K & 63) || (q.lda & 3) || (q.ldw & 3) || (q.prev && (q.ldp & 3)) || (q.type == 1 && (!q.b || (q.ldb & 3))) || !q.a || !q.bias ||
            (((uintptr_t)q.a | (uintptr_t)q.b | (uintptr_t)q.prev | (uintptr_t)q.F | (uintptr_t)q.W | (uintptr_t)q.bias | (uintptr_t)q.U | (uintptr_t)q.A) & 15))
            return false;
    }
    return true;
}
static int launch_gemm32_n64f_impl(const N64FDesc* d, int n, int gelu, hipStream_t s) {
    IISAN_CHECK_SHAPE(gemm32_n64f_ok(d, n), "gemm32_n64f: unsupported problem");
    N64FBatch nb{};
    nb.gelu = gelu;
    int64_t maxM = 0;
    int max_k = 0;
    for (int i = 0; i < n; ++i) {
        N64FProb& q = nb.p[i];
        q.a = d[i].a; q.b = d[i].b; q.prev = d[i].prev; q.lda = d[i].lda; q.ldb = d[i].ldb; q.ldp = d[i].ldp; q.gate = d[i].gate; q.type = d[i].type;
        q.F = d[i].F; q.W = d[i].W; q.ldw = d[i].ldw; q.bias = d[i].bias; q.U = d[i].U; q.A = d[i].A; q.M = d[i].M; q.K = d[i].K;
        if (d[i].M > maxM) maxM = d[i].M;
        if (d[i].K > max_k) max_k = d[i].K;
    }
    // rows per workgroup: 64, or — when the rows divide into one workgroup per CU and tower with 33..63 rows each (Cached,
    // bs = 1024: 11,264 = 256 x 44) — that count: 768 workgroups = exactly three per CU instead of 528 = 2.06 (same box: 5.72 ->
    // 5.63 ms per Cached step; knob value 2 = always 64).  The same tiling made gemm32_k64_kernel SLOWER (44 rows x 3 blocks: 5.79
    // against 5.62 ms): there the fourth wave's idle MFMA share and the extra weight staging outweigh the balance.
    const int cus = iisan_cu_count();
    int rpw = 64;
    if (g_use_n64f != 2 && maxM % cus == 0 && maxM / cus > 32 && maxM / cus < 64) rpw = (int)(maxM / cus);
    nb.rpw = rpw;
    const int64_t rt = ceil_div(maxM, rpw);
    IISAN_CHECK_SHAPE(rt < (1ll << 31), "gemm32: grid too large");
    int ks = 1;
    if (g_scratch && rt * n < 384) {           // too few 64-row tiles to fill the chip: split K through the executor's scratch
        ks = max_k / (4 * TK);
        if (ks > 16) ks = 16;
        while (ks >= 2) {
            int64_t need = 0;
            for (int i = 0; i < n; ++i) need += (int64_t)ks * (int64_t)align_up((size_t)(d[i].M * 64), 64);
            if ((size_t)need <= g_scratch_floats) break;
            ks >>= 1;
        }
        if (ks < 2) ks = 1;
    }
    if (ks >= 2) {
        int64_t off = 0;
        for (int i = 0; i < n; ++i) {
            nb.p[i].P = g_scratch + off;
            nb.p[i].pstride = (int64_t)align_up((size_t)(d[i].M * 64), 64);
            off += ks * nb.p[i].pstride;
        }
    }
    ++g_cnt_n64f;
    hipLaunchKernelGGL(gemm32_n64f_kernel, dim3((unsigned)rt, (unsigned)ks, (unsigned)n), dim3(256), 0, s, nb);
    IISAN_LAUNCH_OK();
    if (ks < 2) return IISAN_OK;
    ReduceBatch rb{};
    int64_t max_mn = 0;
    for (int i = 0; i < n; ++i) {
        Gemm32Prob q{};
        q.C = d[i].A; q.ldc = 64; q.bias = d[i].bias; q.act_src = d[i].U; q.M = d[i].M; q.N = 64; q.K = d[i].K; q.ldr = 64;
        rb.p[i] = q;
        rb.P[i] = nb.p[i].P;
        rb.stride[i] = nb.p[i].pstride;
        const int64_t ktiles = d[i].K / TK, per = ceil_div(ktiles, (int64_t)ks);
        rb.ns[i] = (int32_t)ceil_div(ktiles, per);
        if (d[i].M * 64 > max_mn) max_mn = d[i].M * 64;
    }
    int64_t blocks = ceil_div(max_mn, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm32_reduce_kernel, dim3((unsigned)blocks, 1, (unsigned)n), dim3(256), 0, s, rb, ks, (gelu ? G32_GELU : G32_RELU) | G32_PREACT);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

// Shapes the K = 64 kernel takes (shared by launch_gemm32's own dispatch test and the gate-fused entry below)
static bool k64_shape_ok(const Gemm32Prob& q) {
    return q.K == 64 && q.N >= 256 && (q.N & 63) == 0 && !q.act_src && q.ksplit_stride == 0 &&
           (q.lda & 3) == 0 && (q.ldb & 3) == 0 && (q.ldc & 3) == 0 && (!q.resid || (q.ldr & 3) == 0) &&
           (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C | (uintptr_t)q.resid | (uintptr_t)q.bias) & 15) == 0;
}
static int g_use_k64_gate = 1;
IISAN_DEV_KNOB(gemm32_k64_gate, g_use_k64_gate);
bool gemm32_k64_gate_ok(const Gemm32Prob* probs, const K64Gate* gates, int nprob) {
    if (!g_use_k64 || !g_use_k64_gate || nprob < 1 || nprob > 4) return false;
    for (int i = 0; i < nprob; ++i) {
        const K64Gate& g = gates[i];
        if (!k64_shape_ok(probs[i]) || !probs[i].resid || !g.gate || !g.ga || !g.dgate || (g.ldga & 3) || (g.go && (g.ldgo & 3)) || (g.d2 && ((g.ldd2 & 3) || ((uintptr_t)g.d2 & 15))) ||
            (((uintptr_t)g.ga | (uintptr_t)g.go) & 15))
            return false;
    }
    return true;
}
// dF = A[M, 64] · W[64, N] + resid with the gated fusion's backward folded into the epilogue (sidenet.hip, separate SANB launches):
// the gate gradient <dF, a - other> and the (1 - g) scaling of dprev happen while dF is in registers — fuse_bwd_kernel's pass
// over dF (read + write of every [M, D] state gradient) disappears.  W stored [64, N] (G32_TB).
static int launch_gemm32_k64_gate_impl(const Gemm32Prob* probs, const K64Gate* gates, int nprob, hipStream_t s) {
    IISAN_CHECK_SHAPE(gemm32_k64_gate_ok(probs, gates, nprob), "gemm32_k64_gate: unsupported problem");
    K64Batch kb{};
    int64_t maxM = 0;
    int maxnb = 0;
    for (int i = 0; i < nprob; ++i) {
        const Gemm32Prob& q = probs[i];
        K64Prob& k = kb.p[i];
        k.A = q.A; k.W = q.B; k.bias = q.bias; k.resid = q.resid; k.C = q.C; k.M = q.M; k.N = q.N;
        k.lda = q.lda; k.ldw = q.ldb; k.ldc = q.ldc; k.ldr = q.ldr;
        k.gate = gates[i].gate; k.ga = gates[i].ga; k.go = gates[i].go; k.ldga = gates[i].ldga; k.ldgo = gates[i].ldgo;
        k.dgate = gates[i].dgate; k.scale_prev = gates[i].scale_prev; k.store = gates[i].store;
        k.d2 = gates[i].d2; k.ldd2 = gates[i].ldd2; k.d2_is_b = gates[i].d2_is_b;
        if (q.M > maxM) maxM = q.M;
        if ((q.N >> 6) > maxnb) maxnb = q.N >> 6;
    }
    const int64_t rt = ceil_div(maxM, 64);
    int nblk = 4;
    while (nblk > 1 && rt * ceil_div(maxnb, nblk) * nprob < 1024) nblk >>= 1;
    IISAN_CHECK_SHAPE(rt < (1ll << 31), "gemm32: grid too large");
    const dim3 grid((unsigned)rt, (unsigned)ceil_div(maxnb, nblk), (unsigned)nprob);
    const size_t lds = (size_t)64 * (64 * nblk + 4) * sizeof(float);
    static OncePerDevice attr;
    if (attr.first())
        IISAN_HIP_OK(hipFuncSetAttribute((const void*)gemm32_k64_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 68 * 4));
    ++g_cnt_k64;
    hipLaunchKernelGGL((gemm32_k64_kernel<true, true>), grid, dim3(256), lds, s, kb, nblk);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

int launch_colsum(const float* const* X, float* const* out, const int64_t* M, const int32_t* N, const int32_t* ld,
                  int nprob, hipStream_t s);
static int launch_gemm32_impl(const Gemm32Prob* probs, int nprob, int flags, hipStream_t s) {
    IISAN_CHECK_SHAPE(nprob >= 1 && nprob <= G32_MAXP, "gemm32: 1..%d problems per launch (got %d)", G32_MAXP, nprob);
    Gemm32Batch b{};
    int64_t min_k = INT64_MAX;
    for (int i = 0; i < nprob; ++i) {
        b.p[i] = probs[i];
        IISAN_CHECK_SHAPE(probs[i].M > 0 && probs[i].N > 0 && probs[i].K > 0, "gemm32: empty problem %d", i);
        if (probs[i].K < min_k) min_k = probs[i].K;
    }
    // K = 64 products with a wide N and a plain epilogue (bias / residual): gemm32_k64_kernel
    if ((flags & ~G32_TB) == 0 && g_use_k64 && nprob > 1) {
        // a group that mixes K = 64 products with others (Versa's fc backward: dO = dY Wf is [1408, 64] x [64 -> 1024 | 8192] for the two
        // towers and [1408, 1024] x [1024 -> 1024] for the inter-modal one) used to go to the tiled kernel as a whole — one K-tile per
        // workgroup for the K = 64 members: 68.6 us for 46 MB of output (round 6).  The K = 64 members get their own launch.
        Gemm32Prob yes[G32_MAXP], no[G32_MAXP];
        int ny = 0, nn = 0;
        for (int i = 0; i < nprob; ++i) { if (k64_shape_ok(probs[i])) yes[ny++] = probs[i]; else no[nn++] = probs[i]; }
        if (ny > 0 && nn > 0) {
            IISAN_TRY(launch_gemm32_impl(yes, ny, flags, s));
            return launch_gemm32_impl(no, nn, flags, s);
        }
    }
    if ((flags & ~G32_TB) == 0 && g_use_k64 && nprob <= 4) {      // (K64Batch holds four)
        bool ok = true;
        int64_t maxM = 0;
        int maxnb = 0;
        K64Batch kb{};
        for (int i = 0; i < nprob && ok; ++i) {
            const Gemm32Prob& q = probs[i];
            ok = q.K == 64 && q.N >= 256 && (q.N & 63) == 0 && !q.act_src && q.ksplit_stride == 0 &&
                 (q.lda & 3) == 0 && (q.ldb & 3) == 0 && (q.ldc & 3) == 0 && (!q.resid || (q.ldr & 3) == 0) &&
                 (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C | (uintptr_t)q.resid | (uintptr_t)q.bias) & 15) == 0;
            kb.p[i] = K64Prob{};
            kb.p[i].A = q.A; kb.p[i].W = q.B; kb.p[i].bias = q.bias; kb.p[i].resid = q.resid; kb.p[i].C = q.C; kb.p[i].M = q.M; kb.p[i].N = q.N;
            kb.p[i].lda = q.lda; kb.p[i].ldw = q.ldb; kb.p[i].ldc = q.ldc; kb.p[i].ldr = q.ldr;
            if (q.M > maxM) maxM = q.M;
            if ((q.N >> 6) > maxnb) maxnb = q.N >> 6;
        }
        if (ok) {
            const int64_t rt = ceil_div(maxM, 64);
            int nblk = 4;                                    // 64-column blocks per workgroup: fewer while the launch is small
            while (nblk > 1 && rt * ceil_div(maxnb, nblk) * nprob < 1024) nblk >>= 1;
            // One 64-row tile per workgroup.  (Persistent workgroups that keep their W slice and walk several row tiles were
            // slower at every size tried — Versa 5.86 vs 5.65 ms per step, Cached on this route 6.41 vs 6.25: a tile's operand
            // and first residual round trips are exposed once per tile and the hardware's own workgroup dispatch balances better.)
            const int64_t ny = ceil_div(maxnb, nblk);
            IISAN_CHECK_SHAPE(rt < (1ll << 31), "gemm32: grid too large");
            const dim3 grid((unsigned)rt, (unsigned)ny, (unsigned)nprob);
            const size_t lds = (size_t)64 * (64 * nblk + 4) * sizeof(float) > (size_t)64 * nblk * 68 * sizeof(float)
                                   ? (size_t)64 * (64 * nblk + 4) * sizeof(float) : (size_t)64 * nblk * 68 * sizeof(float);
            static OncePerDevice attr;
            if (attr.first()) {
                IISAN_HIP_OK(hipFuncSetAttribute((const void*)gemm32_k64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 68 * 4));
                IISAN_HIP_OK(hipFuncSetAttribute((const void*)gemm32_k64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 68 * 4));
            }
            ++g_cnt_k64;
            if (flags & G32_TB) hipLaunchKernelGGL(gemm32_k64_kernel<true>, grid, dim3(256), lds, s, kb, nblk);
            else hipLaunchKernelGGL(gemm32_k64_kernel<false>, grid, dim3(256), lds, s, kb, nblk);
            IISAN_LAUNCH_OK();
            return IISAN_OK;
        }
    }
    auto tiles_for = [&](int tm) {
        int64_t mt = 0;
        for (int i = 0; i < nprob; ++i) {
            const int64_t t = ceil_div(probs[i].M, tm) * ceil_div(probs[i].N, TN);
            if (t > mt) mt = t;
        }
        return mt;
    };
    // row-tile height: the tallest that still gives the chip ~one workgroup per CU (weight-gradient launches spread
    // K instead and keep 64)
    int TM = 64;
    if (!(flags & G32_ACCUM)) {
        if (tiles_for(64) * nprob < g_tm_thresh) TM = tiles_for(32) * nprob >= g_tm_thresh ? 32 : 16;
    }
    const int64_t max_tiles = tiles_for(TM);
    IISAN_CHECK_SHAPE(max_tiles < (1ll << 31), "gemm32: grid too large");
    // weight-gradient shape on whole 64-tiles: gemm32_dw_kernel (needs the scratch route for its raw split-K partials)
    bool dw_ok = g_use_dw && (flags & (G32_TA | G32_TB | G32_ACCUM)) == (G32_TA | G32_TB | G32_ACCUM) && (flags & ~(G32_TA | G32_TB | G32_ACCUM)) == 0 &&
                 g_scratch && g_accum_via_scratch;
    for (int i = 0; i < nprob && dw_ok; ++i) {
        const Gemm32Prob& q = probs[i];
        dw_ok = (q.M & 63) == 0 && (q.N & 63) == 0 && (q.K & 63) == 0 && q.K >= 4 * TK && (q.lda & 3) == 0 && (q.ldb & 3) == 0 &&
                (((uintptr_t)q.A | (uintptr_t)q.B) & 15) == 0;
    }
    int splitk = 1;
    if (flags & G32_ACCUM) {   // weight-gradient shape: few tiles, long K -> spread K over the chip
        int64_t want = ceil_div(g_splitk_target, max_tiles * nprob);
        if (dw_ok) {
            // gemm32_dw_kernel: as many splits as give every CU ONE workgroup (nearest count).  Same-box sweep (tools/dw_ab.py):
            // Cached (36 tiles, 176 K-tiles) 7 splits = 252 workgroups 5.95 ms, 14 = 504 (two per CU) 5.95, but 8 = 288 6.20,
            // 10 = 360 6.07, 15 = 540 6.09 (a second, nearly empty round: 45.7 us per launch against 35.7) and 5 = 180 6.13;
            // Versa (160 tiles, 22 K-tiles) 2 splits = 320 workgroups 5.45 ms, 3 = 480 5.55, 1 = 160 5.77.
            int64_t sum_tiles = 0;
            for (int i = 0; i < nprob; ++i) sum_tiles += (probs[i].M >> 6) * (probs[i].N >> 6);
            const int cus = iisan_cu_count();
            want = ((int64_t)cus + sum_tiles / 2) / sum_tiles;
            // more than three problems = both weight gradients of a SANB step in one launch (round 6): TWO workgroups per CU (nearest count) —
            // same-box sweeps: Cached (72 tiles, 176 K-tiles) 7 splits = 504 workgroups 4.78 - 4.81 ms per step, 6 4.86, 14 4.88 - 4.90,
            // 3 4.91, 5 / 10 4.94 - 4.96, 4 = 288 (what the one-per-CU rule picks: a second, nearly empty round) 5.07; the two launches
            // before 4.93 - 4.94.  Versa (320 tiles, 22 K-tiles) 2 splits 4.30 - 4.31 ms, 1 4.36 - 4.41, 3 4.38 - 4.39.
            if (nprob > 3) want = ((int64_t)2 * cus + sum_tiles / 2) / sum_tiles;
            if (g_dw_splits > 0 && nprob > 3) want = g_dw_splits;          // (sweeps of the merged SANB launches)
        }
        const int64_t maxs = ceil_div(min_k, 2 * TK);
        splitk = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
        if (splitk < 1) splitk = 1;
    }
    // Skinny long-K products that are not "+=" (the [M, 8192] -> 64 projections of Versa's text tower: 88 workgroups, each a
    // serial chain of 128 K-tiles — 77 us for 1.5 GFLOP): spread K over the chip through the executor's scratch buffer and
    // let a reducer apply the epilogue to the complete sums.
    Gemm32Batch orig = b;
    bool via_scratch = false;
    if (!(flags & G32_ACCUM) && g_scratch && max_tiles <= 192) {
        int64_t max_k = 0, need = 0;
        for (int i = 0; i < nprob; ++i) if (probs[i].K > max_k) max_k = probs[i].K;
        int ks = (int)(max_k / (8 * TK));
        if (ks > 8) ks = 8;
        for (int i = 0; i < nprob; ++i) need += (int64_t)ks * (int64_t)align_up((size_t)(probs[i].M * probs[i].N), 64);
        if (ks >= 2 && (size_t)need <= g_scratch_floats) {
            int64_t off = 0;
            for (int i = 0; i < nprob; ++i) {
                Gemm32Prob& q = b.p[i];
                q.C = g_scratch + off; q.ldc = q.N; q.bias = nullptr; q.resid = nullptr; q.act_src = nullptr;
                q.ksplit_stride = (int64_t)align_up((size_t)(q.M * q.N), 64);
                off += ks * q.ksplit_stride;
            }
            splitk = ks;
            via_scratch = true;
        }
    }
    // Weight-gradient products ("+=", K = number of item slots): the split-K partial products used to be added with fp32
    // atomics — 29 splits x 49k outputs x 3 towers = 4.3 M atomics per launch, which is what the launch took (45 us at
    // K = 11264 and 48 us at K = 4373: independent of K; rocprofv3, Cached step).  Through the scratch buffer instead: plain
    // 16-byte stores of the partials and a reducer that adds them to C in a fixed order — faster, and the weight gradients
    // become bit-reproducible.  Falls back to atomics when the scratch buffer is missing or too small.
    int structural = flags & (G32_TA | G32_TB | G32_ACCUM);
    // (round 6: gemm32_dw_kernel also when ONE split is wanted — a group whose tiles already fill the chip, like Versa's fc weight
    //  gradients (400 tiles), fell back to the tiled kernel: 22 K-tiles in a row per workgroup, 77 us; the raw partial costs one more pass
    //  of the reducer over the output and is still the faster route)
    if ((flags & G32_ACCUM) && g_scratch && g_accum_via_scratch && (splitk >= 2 || dw_ok)) {
        {   // only NON-EMPTY K ranges: the kernel gives split y the K-tiles [y, y+1) * ceil(ktiles / splits) and a split whose
            // range is empty writes nothing — its partial would be read uninitialised by the reducer
            const int64_t ktiles = ceil_div(min_k, (int64_t)TK), per = ceil_div(ktiles, (int64_t)splitk);
            splitk = (int)ceil_div(ktiles, per);
        }
        // (+ M floats behind every C partial for the A operand's column sums, where gemm32_dw_kernel computes them)
        auto cs_floats = [&](const Gemm32Prob& q) { return (dw_ok && q.colsum_a) ? q.M : 0; };
        int64_t need = 0;
        for (int i = 0; i < nprob; ++i) need += (int64_t)splitk * (int64_t)align_up((size_t)(probs[i].M * probs[i].N + cs_floats(probs[i])), 64);
        if ((size_t)need <= g_scratch_floats) {
            int64_t off = 0;
            for (int i = 0; i < nprob; ++i) {
                Gemm32Prob& q = b.p[i];
                q.C = g_scratch + off; q.ldc = q.N; q.bias = nullptr; q.resid = nullptr; q.act_src = nullptr;
                q.ksplit_stride = (int64_t)align_up((size_t)(q.M * q.N + cs_floats(q)), 64);
                off += splitk * q.ksplit_stride;
                orig.p[i].resid = orig.p[i].C;          // the reducer adds the old C:  C = sum of partials + C
                orig.p[i].ldr = orig.p[i].ldc;
                orig.p[i].bias = nullptr;
            }
            via_scratch = true;
            structural &= ~G32_ACCUM;
        }
    }
    dim3 grid((unsigned)max_tiles, (unsigned)splitk, (unsigned)nprob);
    // FAST fetch: full tiles, whole K-tiles per split, 16-byte aligned operand rows — for every problem of the launch
    bool fast = true;
    for (int i = 0; i < nprob && fast; ++i) {
        const Gemm32Prob& q = probs[i];
        fast = q.M % TM == 0 && q.N % TN == 0 && q.K % TK == 0 &&
               (q.lda & 3) == 0 && (q.ldb & 3) == 0 && ((uintptr_t)q.A & 15) == 0 && ((uintptr_t)q.B & 15) == 0;
    }
    const int epi = via_scratch ? 0 : (flags & ~(G32_TA | G32_TB | G32_ACCUM));
    int rc;
    bool cs_folded = false;          // the A operand's column sums came out of the product kernel
    if (dw_ok && via_scratch && structural == (G32_TA | G32_TB)) {
        ++g_cnt_dw;
        hipLaunchKernelGGL(gemm32_dw_kernel, grid, dim3(256), 0, s, b);
        IISAN_LAUNCH_OK();
        rc = IISAN_OK;
        cs_folded = true;
    } else
    switch (structural) {
#define G32_CASE(F) case (F): rc = launch_flags<(F)>(b, grid, TM, epi, fast, min_k >= 4096, s); break
        G32_CASE(0);
        G32_CASE(G32_TA);
        G32_CASE(G32_TB);
        G32_CASE(G32_TA | G32_TB);
        G32_CASE(G32_ACCUM);
        G32_CASE(G32_TA | G32_ACCUM);
        G32_CASE(G32_TB | G32_ACCUM);
        G32_CASE(G32_TA | G32_TB | G32_ACCUM);
#undef G32_CASE
        default: iisan_set_error("gemm32: bad flags 0x%x", flags); return IISAN_EBADSHAPE;
    }
    if (rc == IISAN_OK && !cs_folded && (flags & G32_TA)) {          // every other route: the column sums as a launch of their own
        const float* X[G32_MAXP]; float* O[G32_MAXP]; int64_t Ms[G32_MAXP]; int32_t Ns[G32_MAXP], lds[G32_MAXP];
        int n = 0;
        for (int i = 0; i < nprob; ++i)
            if (probs[i].colsum_a) { X[n] = probs[i].A; O[n] = probs[i].colsum_a; Ms[n] = probs[i].K; Ns[n] = (int32_t)probs[i].M; lds[n] = probs[i].lda; ++n; }
        for (int i = 0; i < n; i += 4) IISAN_TRY(launch_colsum(X + i, O + i, Ms + i, Ns + i, lds + i, n - i < 4 ? n - i : 4, s));
    }
    if (rc != IISAN_OK || !via_scratch) return rc;
    ReduceBatch rb{};
    int64_t max_mn = 0;
    for (int i = 0; i < nprob; ++i) {
        rb.p[i] = orig.p[i];
        rb.P[i] = b.p[i].C;
        rb.stride[i] = b.p[i].ksplit_stride;
        rb.cs[i] = cs_folded ? orig.p[i].colsum_a : nullptr;
        {   // the kernel gives split y the K-tiles [y, y+1) * ceil(ktiles / splits)
            const int64_t ktiles = ceil_div(orig.p[i].K, (int64_t)TK), per = ceil_div(ktiles, (int64_t)splitk);
            rb.ns[i] = (int32_t)ceil_div(ktiles, per);
        }
        if (orig.p[i].M * orig.p[i].N > max_mn) max_mn = orig.p[i].M * orig.p[i].N;
    }
    int64_t blocks = ceil_div(max_mn, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm32_reduce_kernel, dim3((unsigned)blocks, 1, (unsigned)nprob), dim3(256), 0, s, rb, splitk, flags & ~(G32_TA | G32_TB | G32_ACCUM));
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

// ---- public launchers: the implementation above, bracketed by HIP events when bench.py times this kernel family (class 2).
// A launcher that re-enters another (split-K through the scratch buffer) is timed once, as the caller sees it: product +
// reducer together.
namespace { thread_local int g_timing_depth = 0; }
struct TimedG32 {
    hipStream_t s; bool on;
    TimedG32(hipStream_t s_, double flops, double bytes) : s(s_), on(iisan_timing_class() == 2 && g_timing_depth == 0) {
        ++g_timing_depth;
        if (on) iisan_timing_pre(s, flops, bytes);
    }
    ~TimedG32() {
        --g_timing_depth;
        if (on) iisan_timing_post(s);
    }
};
int launch_gemm32(const Gemm32Prob* probs, int nprob, int flags, hipStream_t s) {
    double fl = 0, by = 0;
    for (int i = 0; i < nprob && i < G32_MAXP; ++i) {
        fl += 2.0 * (double)probs[i].M * probs[i].N * (double)probs[i].K;
        by += 4.0 * ((double)probs[i].M * probs[i].K + (double)probs[i].N * probs[i].K + (double)probs[i].M * probs[i].N);
    }
    TimedG32 t(s, fl, by);
    return launch_gemm32_impl(probs, nprob, flags, s);
}
int launch_gemm32_n64f(const N64FDesc* d, int n, int gelu, hipStream_t s) {
    double fl = 0, by = 0;
    for (int i = 0; i < n; ++i) {       // reads tap (+ second tap) + previous state, writes F, U, A; the product is [M, K] x [K, 64]
        fl += 2.0 * (double)d[i].M * 64.0 * d[i].K;
        by += 4.0 * ((double)d[i].M * d[i].K * ((d[i].a ? 1 : 0) + (d[i].b ? 1 : 0) + (d[i].prev ? 1 : 0) + 1) + 64.0 * d[i].K + 2.0 * d[i].M * 64.0);
    }
    TimedG32 t(s, fl, by);
    return launch_gemm32_n64f_impl(d, n, gelu, s);
}
int launch_gemm32_k64_gate(const Gemm32Prob* probs, const K64Gate* gates, int nprob, hipStream_t s) {
    double fl = 0, by = 0;
    for (int i = 0; i < nprob; ++i) {   // dF = dU W (K = 64) with the gated fusion's backward in the epilogue: tap + state in, (1 - g) dF out
        fl += 2.0 * (double)probs[i].M * probs[i].N * 64.0;
        by += 4.0 * ((double)probs[i].M * 64.0 + (double)probs[i].N * 64.0 + 4.0 * (double)probs[i].M * probs[i].N);
    }
    TimedG32 t(s, fl, by);
    return launch_gemm32_k64_gate_impl(probs, gates, nprob, s);
}

int launch_colsum(const float* const* X, float* const* out, const int64_t* M, const int32_t* N, const int32_t* ld,
                  int nprob, hipStream_t s) {
    IISAN_CHECK_SHAPE(nprob >= 1 && nprob <= 4, "colsum: 1..4 problems per launch");
    ColsumBatch b{};
    int64_t maxM = 0;
    int maxN = 0;
    for (int i = 0; i < nprob; ++i) {
        b.X[i] = X[i]; b.out[i] = out[i]; b.M[i] = M[i]; b.N[i] = N[i]; b.ld[i] = ld[i];
        if (M[i] > maxM) maxM = M[i];
        if (N[i] > maxN) maxN = N[i];
    }
    const int rows_per_block = 256;
    dim3 grid((unsigned)ceil_div(maxN, 64), (unsigned)ceil_div(maxM, rows_per_block), (unsigned)nprob);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, b, rows_per_block);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

extern "C" int iisan_gemm32(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N, int64_t K,
                            int32_t ta, int32_t tb, int32_t relu, int32_t accumulate, void* stream) {
    Gemm32Prob p{};
    p.A = A; p.B = B; p.bias = bias; p.resid = nullptr; p.act_src = nullptr; p.C = C;
    p.M = M; p.N = N; p.K = K;
    p.lda = ta ? (int32_t)M : (int32_t)K;
    p.ldb = tb ? N : (int32_t)K;
    p.ldc = N; p.ldr = N;
    int flags = (ta ? G32_TA : 0) | (tb ? G32_TB : 0) | (relu ? G32_RELU : 0) | (accumulate ? G32_ACCUM : 0);
    return launch_gemm32(&p, 1, flags, (hipStream_t)stream);
}

extern "C" int iisan_linear_fwd(const float* x, const float* w, const float* b, float* y, int64_t M, int32_t K, int32_t N,
                                void* stream) {
    Gemm32Prob p{};
    p.A = x; p.B = w; p.bias = b; p.C = y; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.ldr = N;
    return launch_gemm32(&p, 1, 0, (hipStream_t)stream);
}

// dx = dy · W ; dW += dy^T · x ; db += colsum(dy)   (dx / dw / db may be NULL to skip)
extern "C" int iisan_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                int64_t M, int32_t K, int32_t N, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        Gemm32Prob p{};
        p.A = dy; p.B = w; p.C = dx; p.M = M; p.N = K; p.K = N; p.lda = N; p.ldb = K; p.ldc = K; p.ldr = K;
        IISAN_TRY(launch_gemm32(&p, 1, G32_TB, s));
    }
    if (dw) {
        Gemm32Prob p{};
        p.A = dy; p.B = x; p.C = dw; p.M = N; p.N = K; p.K = M; p.lda = N; p.ldb = K; p.ldc = K; p.ldr = K;
        IISAN_TRY(launch_gemm32(&p, 1, G32_TA | G32_TB | G32_ACCUM, s));
    }
    if (db) {
        const float* X[1] = {dy};
        float* O[1] = {db};
        int64_t Ms[1] = {M};
        int32_t Ns[1] = {N}, lds[1] = {N};
        IISAN_TRY(launch_colsum(X, O, Ms, Ns, lds, 1, s));
    }
    return IISAN_OK;
}
