// Fourth-generation 16-bit MFMA GEMM: the staggered 256x256x64 kernel (gemm16_s256.hip) with a HALF-SLOT TILE BOUNDARY.
//
// What s256 loses: with K = 768 a tile is only 12 K-steps long, and at every tile boundary the epilogue (bias,
// GELU, 16-bit convert, 128 KiB of stores per tile) got slots of its own —
//        A: M(L) | E1 | E2+R(0') | M(0')         one slot with NO MFMA at all per tile, two more stretched by the stores:
//        B: R(L) | M(L) | E1     | E2+R(0')      QKV 986 / FC1 873-950 TFLOP/s against 1,260-1,280 without an epilogue.
// An epilogue can only overlap the sibling group's MFMAs while its accumulators are not needed by the own group's next
// MFMAs.  So the LAST K-step L of a tile and the FIRST K-step 0' of the next are each cut into two half-steps by output
// rows: "lo" = the group's rows 0..63 (acc[0..1][*]), "hi" = rows 64..127 (acc[2..3][*]):
//
//        A:  Rlo(L) Mlo(L) [Rhi(L)+Elo] Mhi(L) | Rlo(0') Mlo(0') [Rhi(0')+Ehi] Mhi(0') | R(1') M(1') ...
//        B:  one slot behind, same program
//
// The lo accumulators are final after Mlo(L): their epilogue Elo runs in the NEXT read slot, beside the sibling's MFMAs, and
// must be done before Mlo(0') overwrites them; the hi accumulators are final after Mhi(L) and are flushed in the read slot
// between Mlo(0') and Mhi(0').  Every slot of the boundary carries MFMAs (16 instead of 32), no slot is epilogue-only, and no
// extra register is needed (a half-step holds 64 fragment registers instead of 96).  Cost: three more barriers per tile.
//
// LDS-DMA schedule (ring, pieces and their owners as in s256; plan(s) = 8 pieces: 0..3 / 4..7):
//        group A plan(s): 0..3 = B's A-operand half of step s+1, 4..7 = A's own half of step s+2
//        group B plan(s): 0..3 = W rows 0..127 of step s+2,      4..7 = W rows 128..255 of step s+2
//   middle step  M(s)   : all 8 pieces of plan(s), one after every 4 MFMAs (as s256)
//   last step    Mlo(L) : pieces 0..3 of plan(L);            Mhi(L)  : none
//   first step   Mlo(0'): pieces 4..7 of plan(L) + 0..3 of plan(0') (one after every 2 MFMAs);  Mhi(0'): 4..7 of plan(0')
// Nothing is issued in Mhi(L): a load issued there would be YOUNGER than Elo's stores, and `vmcnt` retires in order on gfx9
// — the wait for it at the end of Rlo(0') would wait for the stores as well.  As scheduled, every wait at the end of an
// [R+E] slot is `vmcnt(8)` (the 8 stores of that half-epilogue stay in flight), Rlo(0') needs no wait at all, and only the
// wait at the end of R(1') covers stores — Ehi's, two slots old.  Buffer safety (slot numbers: A's Rlo(L) = 1):
//   A Mlo(L)  [2]: B's half of 0'  -> read by B in [6],[8]; buffer last read by B in [0];      waited by A at the end of [3]
//   A Mlo(0') [6]: own half of 1'  -> read by A in [9];     buffer last read by A in [1],[3];  waited at the end of [7]
//                  B's half of 1'  -> read by B in [10];    buffer last read by B in [2],[4]
//   A Mhi(0') [8]: own half of 2'  -> read by A in [11];    buffer last read by A in [5],[7];  waited at the end of [9]
//   B Mlo(L)  [3]: W 0..127 of 1'  -> read in [9],[10];     W buffer last read in [1],[2];     waited by B at the end of [4]
//   B Mlo(0') [7]: W 128..255 of 1'-> read in [9],[10];     ditto;                             waited at the end of [8]
//                  W 0..127 of 2'  -> read in [11],[12];    W buffer last read in [5],[6]
//   B Mhi(0') [9]: W 128..255 of 2'-> read in [11],[12];                                       waited at the end of [10]
// (the whole W fragment set of a step is read in Rlo, the A-operand rows 0..63 in Rlo and 64..127 in Rhi).
// Tile walk, LDS swizzle, operand swap / W-row permutation, fences and the 16-bit epilogues are those of gemm16_s256.hip.
//
// Round 4, second half — two epilogue families that remove every kernel between the products of a pre-LN block:
//   LNA (template flag; EPI_QKVH16 / EPI_GELU16, fp16): A is the un-normalised residual stream, W the gamma-folded, centred weights
//        (rowops.hip: fold_ln_weights_kernel); the epilogue multiplies by one rstd per row — the tile's 256 values arrive by one LDS-DMA
//        instruction at the top of the tile's first [Rhi + E] slot, two buffers — where the plain epilogue adds the bias.
//   EPI_STREAM16 (fp16, N <= 1024): the O / FC2 products add into the fp16 stream in place (x = fp16(x + acc + b)) and leave per-slice
//        row sums for stream_stats_finalize (rowops.hip).  The old stream values of a half-epilogue are requested two slots ahead
//        (stream_load: inline-asm loads, waits by hand — `vmcnt` is in order and the LDS-DMA pieces queue behind them).
#include "common.h"
#include <type_traits>

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int SBM = 256, SBN = 256, SBK = 64;
constexpr int S_OP_BYTES = SBM * SBK * 2;        // 32 KiB per operand tile
constexpr int S_STAGE_BYTES = 2 * S_OP_BYTES;    // 64 KiB per K-step
constexpr int STG_BIAS_BYTES = 8192 * 4;           // LDS after the ring: the layer's bias vector (N <= 8192 floats; 160 KiB in all)
// LNA (LayerNorm applied in the epilogue, Gemm16Args::rowstat): two 1-KiB buffers of row statistics (rstd of the tile's 256 rows, one
// buffer per tile, alternating) sit at 24 KiB of the bias area
constexpr int LNA_STAT_OFF = 24576, LNA_MAX_N = 6144;
constexpr int STG_TILE_BYTES = 0;

template <typename T> struct Mfma32s;
template <> struct Mfma32s<F16> {
    static __device__ __forceinline__ f16v run(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma32s<BF16> {
    static __device__ __forceinline__ f16v run(b8 a, b8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ int nperm32s(int q) { return (q & ~31) + 16 * ((q >> 2) & 1) + 4 * ((q & 31) >> 3) + (q & 3); }

#define S256_FENCE() __builtin_amdgcn_sched_barrier(0)
#define S256_BARRIER()                                   \
    do {                                                 \
        S256_FENCE();                                    \
        asm volatile("s_barrier" ::: "memory");          \
        S256_FENCE();                                    \
    } while (0)
// vmcnt(n) / lgkmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14), not inline asm:
// hipcc's own wait insertion then knows what has retired.  With the asm form it re-waited `vmcnt(0)` at the first touch of a
// register it had reloaded from scratch slots earlier — in the middle of the epilogue arithmetic, behind the LDS-DMA issued since.
#define S256_VMCNT(n)                                                                              \
    do {                                                                                           \
        S256_FENCE();                                                                              \
        __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14));  \
        S256_FENCE();                                                                              \
    } while (0)
#define S256_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)      /* lgkmcnt(0) */
// the stores of a half-epilogue that may stay in flight over the wait at the end of its slot: 8 per wave, 9 with the row statistics
#define S256_VMCNT_EPI() do { if constexpr (EPI == EPI_STREAM16) S256_VMCNT(9); else S256_VMCNT(8); } while (0)

template <typename T, int EPI, bool LNA>
__global__ __launch_bounds__(512, 2) void gemm16_h256_kernel(Gemm16Args p, int tiles_m, int tiles_n, uint32_t qkv_magic) {
    typedef typename T::v8 V8;
    // Ablation / experiment bits (Gemm16Args::debug, set through the dev switch gemm16_variant = 4 | bits << 8; tools/gemm_walk.py, gemm_stream.py):
    // compiled in only with -DGEMM16_DEBUG_BITS (`make EXTRA=-DGEMM16_DEBUG_BITS`).  The product build folds every test away: each
    // run-time bit test in the epilogue costs scalar registers the kernel does not have (round 5: SGPR spills 17 / 18 -> 11 / 16 in the
    // LayerNorm-epilogue instantiations with two of them gone).
#ifdef GEMM16_DEBUG_BITS
    const int dbg_bits = p.debug;
#else
    constexpr int dbg_bits = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * S_STAGE_BYTES ring + up to 8192 floats of bias
    float* sBias = (float*)(smem + 2 * S_STAGE_BYTES);
    float* sStat = (float*)(smem + 2 * S_STAGE_BYTES + LNA_STAT_OFF);             // LNA: [2][256] row statistics (rstd)
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;      // LDS byte address of the dynamic segment

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 0 = A (tile rows 0..127), 1 = B (rows 128..255).  The YOUNGER half of the workgroup (waves 4..7) is group A: on every
    // SIMD the younger wave loses arbitration to its older sibling, so its MFMA slot runs longer (slot timelines: 1,640 vs
    // 1,252 cycles with the older waves as A; 1,400 vs 1,276 this way round) — it gets the lighter DMA duty (A-operand
    // halves; group B streams the W tile that every CU hammers) and the leading position.
    const int grp = 1 - (wave >> 2);
    const int wq = wave & 3;              // wave within the group = its 64-column slice of the tile

    const int G = gridDim.x;
    const int nk = p.K / SBK;
    // ---- tile walk ----
    // The dispatcher places workgroup b on XCD b % 8 (observed, used for speed only: any placement computes the same tiles).
    // Legacy walk (walk_c == 0): XCD x takes 32 consecutive tiles of the row-major tile list per round — at 9..12 column tiles
    // that is 2.7..3.6 row tiles x EVERY column tile: the whole W (3.5 / 4.7 MB for QKV / FC1) passes through the XCD's 4-MB
    // L2 every round.  Panel walk (walk_c = C > 0): XCD x owns the row tiles [x tiles_m / 8, (x+1) tiles_m / 8) and walks them
    // in sub-slabs of walk_h row tiles; inside a sub-slab it goes panel by panel (C column tiles wide), row-major inside a
    // panel, 32 consecutive positions per round: a round is ~32/C row tiles x C column tiles, the W panel (C x 393 KB at
    // K = 768) is the same for walk_h * C / 32 rounds, and the A rows of a sub-slab come back once per panel.
    const int xcd = blockIdx.x & 7, lidx = blockIdx.x >> 3, L = G >> 3;
    // walk_c == 0: one slab of every row tile, one panel of every column tile, position pid + ti * G — the row-major list
    const bool panel = p.walk_c > 0;
    const int wC = panel ? p.walk_c : tiles_n, wH = panel ? p.walk_h : tiles_m;
    const int slab0 = panel ? (int)(((int64_t)xcd * tiles_m) >> 3) : 0;
    const int slab_rows = panel ? (int)(((int64_t)(xcd + 1) * tiles_m) >> 3) - slab0 : tiles_m;
    const int pid = xcd * L + lidx;
    const int ntiles = slab_rows * tiles_n;
    const int first = panel ? lidx : pid, stride = panel ? L : G;
    const int my_tiles = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
    const int nsteps = my_tiles * nk;
    if (nsteps == 0) return;
    // coordinates of this workgroup's ti-th tile (wave-uniform; computed once per tile, in a read slot)
    auto walk = [&](int ti, int& tm, int& tn) {
        const int n = first + ti * stride;
        const int HT = wH * tiles_n;
        const int sb = n / HT, rem = n - sb * HT;
        const int left = slab_rows - sb * wH;
        const int hrows = left < wH ? left : wH;
        const int PT = hrows * wC;
        const int pn = rem / PT, rem2 = rem - pn * PT;
        const int cleft = tiles_n - pn * wC;
        const int cw = cleft < wC ? cleft : wC;
        const int row = rem2 / cw;
        tm = __builtin_amdgcn_readfirstlane(slab0 + sb * wH + row);
        tn = __builtin_amdgcn_readfirstlane(pn * wC + (rem2 - row * cw));
    };
    int cur_tm, cur_tn, nxt_tm, nxt_tn, prev_tm = 0, prev_tn = 0;
    walk(0, cur_tm, cur_tn);
    nxt_tm = cur_tm; nxt_tn = cur_tn;
    if (my_tiles > 1) walk(1, nxt_tm, nxt_tn);
    // tile and K-step of flat step (current tile, kt + d), d = 1, 2 (nk >= 2: never beyond the next tile).  After the last tile
    // `nxt` repeats `cur`: past the end of the workgroup's steps the plan re-loads steps 0 / 1 of the last tile into buffers
    // nobody reads any more (any valid address will do).
    auto step_at = [&](int kt, int d, int& tm, int& tn, int& k) {
        k = kt + d;
        tm = cur_tm; tn = cur_tn;
        if (k >= nk) { tm = nxt_tm; tn = nxt_tn; k -= nk; }
    };

    // staging offsets, kept to 5 VGPRs: lane -> row-in-chunk r8 = lane>>3, physical slot lane&7; the logical slot is
    // physical ^ ((row>>1)&7) = s0 ^ 4*(j&1) for chunk j of a 32-row-aligned run, s0 = (lane&7) ^ (lane>>4).
    const int r8 = lane >> 3;
    const int s0 = (lane & 7) ^ (lane >> 4);
    const int slotx0 = s0 * 16, slotx1 = (s0 ^ 4) * 16;
    const int rowA = r8 * p.lda * 2;                                             // A-operand: LDS row == global row
    const int rowW = (16 * ((r8 >> 2) & 1) + (r8 & 3)) * p.ldw * 2;              // W: permuted rows (nperm32s)
    const char* Abase = (const char*)p.A;
    const char* Wbase = (const char*)p.W;

    // One 1-KiB piece (8 LDS rows) of the A-operand half `h` / the W half `h` of flat step s; wave wq owns rows
    // h*128 + wq*32 .. +31 of a half, piece j = rows +8j.
    auto piece_A = [&](int s, int h, int j) {      // prologue only: s = 0, 1 lie in the first tile (nk >= 2)
        const int kt = s, tm = cur_tm;
        const int q0 = h * 128 + wq * 32;
        const char* Ag = Abase + (((int64_t)tm * SBM + q0 + 8 * j) * p.lda + (int64_t)kt * SBK) * 2 + rowA;
        glds16(Ag + ((j & 1) ? slotx1 : slotx0), smem + (s & 1) * S_STAGE_BYTES + (q0 + 8 * j) * 128);
    };
    auto piece_W = [&](int s, int h, int j) {   // LDS rows q0+8j.. hold W rows q0 + 4j + {0,16} + {0..3} (nperm32s)
        const int kt = s, tn = cur_tn;
        const int q0 = h * 128 + wq * 32;
        const char* Wg = Wbase + (((int64_t)tn * SBN + q0 + 4 * j) * p.ldw + (int64_t)kt * SBK) * 2 + rowW;
        glds16(Wg + ((j & 1) ? slotx1 : slotx0), smem + (s & 1) * S_STAGE_BYTES + S_OP_BYTES + (q0 + 8 * j) * 128);
    };

    f16v acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    int xoff[4], woff2[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) xoff[mi] = (grp * 128 + mi * 32 + frow) * 128;   // group g reads only its A-half
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) woff2[ni] = (wq * 64 + ni * 32 + frow) * 128;

    V8 wf[2][4], xf[4][4];
    auto read_step = [&](int s) {
        const char* sA = smem + (s & 1) * S_STAGE_BYTES;
        const char* sW = sA + S_OP_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((2 * ks + fh) ^ fsw) << 4;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) wf[ni][ks] = *(const V8*)(sW + woff2[ni] + slot);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) xf[mi][ks] = *(const V8*)(sA + xoff[mi] + slot);
        }
    };
    // half-steps of a tile's last / first K-step: Rlo = every W fragment of the step + the A-operand rows 0..63 of the
    // group's half (16 reads), Rhi = rows 64..127 (8 reads)
    auto read_lo = [&](int s) {
        const char* sA = smem + (s & 1) * S_STAGE_BYTES;
        const char* sW = sA + S_OP_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((2 * ks + fh) ^ fsw) << 4;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) wf[ni][ks] = *(const V8*)(sW + woff2[ni] + slot);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) xf[mi][ks] = *(const V8*)(sA + xoff[mi] + slot);
        }
    };
    auto read_hi = [&](int s) {
        const char* sA = smem + (s & 1) * S_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ((2 * ks + fh) ^ fsw) << 4;
#pragma unroll
            for (int mi = 2; mi < 4; ++mi) xf[mi][ks] = *(const V8*)(sA + xoff[mi] + slot);
        }
    };
    // ---- DMA plan of one MFMA slot: 8 wave-uniform (global address, LDS offset) pairs, computed in the preceding READ
    // slot (which idles ~1000 cycles at its barrier).  Nothing but the MFMAs, one 64-bit add, `s_mov m0` and the
    // `global_load_lds` itself may sit in the MFMA slot: slot timelines showed every scalar instruction or branch between
    // MFMAs to cost matrix-pipe time (an MFMA slot took 1440 cycles with 8 skipped `if`s, 1540 with 32, 1046 bare).
    // The plan is therefore unconditional: past the end of the workgroup's steps it re-loads the last step into
    // buffers that nobody reads any more.
    struct Plan {
        uint32_t g1lo, g1hi, g2lo, g2hi;      // global byte address of pieces 0..3 / 4..7 (SGPRs)
        uint32_t l1, l2;                      // LDS byte offset of piece 0 / piece 4
    };
    // per-lane source offset of piece j (j = pc & 3): row-in-chunk and swizzled 16-B slot, plus j chunks of 8 (A operand)
    // or 4 (permuted W) rows
    uint32_t vj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        vj[j] = (uint32_t)((grp == 0 ? rowA : rowW) + ((j & 1) ? slotx1 : slotx0)) + (uint32_t)j * (grp == 0 ? 8u * p.lda * 2u : 4u * p.ldw * 2u);
    auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // Group A: pieces 0..3 = B's A-operand half of step s+1, 4..7 = A's own half of step s+2;
    // group B: the W tile of step s+2 (pieces 0..3 rows 0..127, 4..7 rows 128..255).
    auto make_plan = [&](int s, int kt) {          // kt = K-step of flat step s inside the current tile (-1: before step 0)
        Plan q;
        uint64_t g1, g2;
        int tm2, tn2, kt2;
        step_at(kt, 2, tm2, tn2, kt2);
        if (grp == 0) {
            int tm1, tn1, kt1;
            step_at(kt, 1, tm1, tn1, kt1);
            g1 = (uint64_t)Abase + (((int64_t)tm1 * SBM + 128 + wq * 32) * p.lda + (int64_t)kt1 * SBK) * 2;
            g2 = (uint64_t)Abase + (((int64_t)tm2 * SBM + wq * 32) * p.lda + (int64_t)kt2 * SBK) * 2;
            q.l1 = ((s + 1) & 1) * S_STAGE_BYTES + (128 + wq * 32) * 128;
            q.l2 = (s & 1) * S_STAGE_BYTES + (wq * 32) * 128;
        } else {
            g1 = (uint64_t)Wbase + (((int64_t)tn2 * SBN + wq * 32) * p.ldw + (int64_t)kt2 * SBK) * 2;
            g2 = g1 + (uint64_t)128 * p.ldw * 2;
            q.l1 = (s & 1) * S_STAGE_BYTES + S_OP_BYTES + (wq * 32) * 128;
            q.l2 = q.l1 + 128 * 128;
        }
        q.g1lo = sgpr((uint32_t)g1); q.g1hi = sgpr((uint32_t)(g1 >> 32));
        q.g2lo = sgpr((uint32_t)g2); q.g2hi = sgpr((uint32_t)(g2 >> 32));
        q.l1 = sgpr(q.l1); q.l2 = sgpr(q.l2);
        return q;
    };
    // 32 MFMAs with this wave's 8 DMA pieces interleaved: piece pc after MFMA 4*pc + 3.  Only the 8 MFMAs of K-slice 0
    // exist in two versions (a tile's first K-step starts from C = 0, an inline-constant operand, instead of zeroing
    // 128 VGPRs): with two full bodies the compiler hoisted the 8 piece addresses above the branch -> 16 VGPRs, spills,
    // and a `s_waitcnt vmcnt(0)` for the reload at the top of every MFMA slot.
    auto mfma_step = [&](const Plan& q, bool first) {
        auto piece = [&](int pc) {
            const int j = pc & 3;
            const uint64_t gb = ((uint64_t)(pc < 4 ? q.g1hi : q.g2hi) << 32) | (pc < 4 ? q.g1lo : q.g2lo);
            S256_FENCE();
            // SGPR base + 32-bit lane offset, written as inline asm: the builtin is selected with a 64-bit VGPR address
            // (a v_lshl_add_u64 per piece between the MFMAs and two address registers read per lane).  Same-box A/B
            // over 6 rounds: QKV 915 -> 921, O 940 -> 942, FC1 921 -> 931, FC2 1146 -> 1162 TFLOP/s.  (M0 is only
            // written here and, in the prologue, by the builtin right before its own use.)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(smem_lds + (pc < 4 ? q.l1 : q.l2) + j * 1024), "v"(vj[j]), "s"(gb) : "memory");
            S256_FENCE();
        };
        auto slice = [&](auto FIRST, int ks) {
            f16v zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = Mfma32s<T>::run(wf[ni][ks], xf[mi][ks], decltype(FIRST)::value ? zero : acc[mi][ni]);
                    if (((mi * 2 + ni) & 3) == 3) piece(ks * 2 + (mi >> 1));
                }
        };
        // The sibling wave of this SIMD is in its read slot (ds_reads, address VALU, plan SALU): with equal priority the
        // arbiter prefers the OLDER wave, and the slot timeline showed group B's MFMA slots at 1,680 cycles against
        // group A's 1,252.  Priority 1 for whoever is on the matrix pipe removes that.
        __builtin_amdgcn_s_setprio(1);
        if (first) slice(std::true_type{}, 0); else slice(std::false_type{}, 0);
        S256_FENCE();
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) slice(std::false_type{}, ks);
        __builtin_amdgcn_s_setprio(0);
    };
    // One half-step: the 16 MFMAs of output rows lo (HI = 0: acc[0..1]) or hi (HI = 1: acc[2..3]) of a K-step, with the DMA
    // pieces the schedule in the file header assigns to it (MODE, compile time — nothing but MFMAs and pieces in the slot):
    //   H_LAST_LO : pieces 0..3 of q (one after every 4 MFMAs)            H_LAST_HI : none
    //   H_FIRST_LO: pieces 4..7 of qp (the previous tile's last plan) and 0..3 of q, one after every 2 MFMAs; C = 0 at ks 0
    //   H_FIRST_HI: pieces 4..7 of q;                                                                      C = 0 at ks 0
    enum { H_LAST_LO = 0, H_LAST_HI = 1, H_FIRST_LO = 2, H_FIRST_HI = 3 };
    auto mfma_half = [&](auto MODE_T, const Plan& q, const Plan& qp) {
        constexpr int MODE = decltype(MODE_T)::value;
        constexpr int HI = MODE & 1;
        constexpr bool FIRST = MODE >= 2;
        auto piece = [&](uint32_t lo32, uint32_t hi32, uint32_t lds, int j) {
            const uint64_t gb = ((uint64_t)hi32 << 32) | lo32;
            S256_FENCE();
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(smem_lds + lds + j * 1024), "v"(vj[j]), "s"(gb) : "memory");
            S256_FENCE();
        };
        f16v zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int mi = 2 * HI + m2;
                    const int n = ks * 4 + m2 * 2 + ni;               // 0..15: index of this MFMA in the half-step
                    acc[mi][ni] = Mfma32s<T>::run(wf[ni][ks], xf[mi][ks], (FIRST && ks == 0) ? zero : acc[mi][ni]);
                    if constexpr (MODE == H_LAST_LO) {
                        if ((n & 3) == 3) piece(q.g1lo, q.g1hi, q.l1, n >> 2);
                    } else if constexpr (MODE == H_FIRST_HI) {
                        if ((n & 3) == 3) piece(q.g2lo, q.g2hi, q.l2, n >> 2);
                    } else if constexpr (MODE == H_FIRST_LO) {
                        if ((n & 1) == 1) {
                            const int pc = n >> 1;                    // 0..7: 0..3 = qp's pieces 4..7, 4..7 = q's pieces 0..3
                            if (pc < 4) piece(qp.g2lo, qp.g2hi, qp.l2, pc); else piece(q.g1lo, q.g1hi, q.l1, pc - 4);
                        }
                    }
                }
            if (FIRST && ks == 0) S256_FENCE();
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // EPI_STREAM16: the old stream values of a half-epilogue — 8 x 16 bytes per lane, the same addresses its stores overwrite — are
    // requested TWO SLOTS before the epilogue that adds them (top of Rlo of the K-step whose [Rhi + E] slot holds that epilogue): the
    // half-slots in between hold 64 fragment registers less than a full step, which is where the 32 registers come from.  Issued
    // inside the epilogue they cost O 341 -> 410 us and FC2 1,090 -> 1,156 us: ~3 us of exposed HBM latency per half-epilogue.
    V8 xo[4][2];
    auto stream_load = [&](int tm, int tn, int half) {
        if constexpr (EPI == EPI_STREAM16) {
            if (dbg_bits & 128) {                // ablation: no stream loads (the sums are garbage).  The registers are DEFINED on this path
                                                // too: left untouched, their previous contents stay live from one stream_load to the next
                                                // epilogue — across the middle K-steps — and the production kernel spilled 12 of them per
                                                // tile (3 scratch stores + 4 reloads, each reload behind an `s_waitcnt vmcnt(0)`; round 5)
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" : "=v"(xo[b][0]), "=v"(xo[b][1]));
                return;
            }
            int l2 = lane;
            asm volatile("" : "+v"(l2));
            const uint32_t lane_off = (uint32_t)((l2 & 31) * p.ldo * 2 + (l2 >> 5) * 32);
            const char* base = (const char*)p.out + ((int64_t)(tm * SBM + grp * 128 + half * 64) * p.ldo + tn * SBN + wq * 64) * 2;
            // (inline asm, waited for by hand in the epilogue: the compiler's own `s_waitcnt` for a plain load does not know the LDS-DMA
            //  pieces issued behind it and came out as `vmcnt(0)` at the top of the epilogue — a wait for the pieces of the MFMA slot
            //  that has just ended)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const char* bp = base + ((int64_t)((b & 1) * 32) * p.ldo + (b >> 1) * 32) * 2;      // wave-uniform
                asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
                             : "=v"(xo[b][0]), "=&v"(xo[b][1]) : "v"(lane_off), "s"(bp) : "memory");
            }
            S256_FENCE();
        }
    };
    bool stores8 = false;      // the last half-epilogue issued exactly 8 store instructions per wave (full row tile, stores enabled)
    // ---- half-epilogue: rows half*64 .. +63 of the group's 128 (acc[2*half .. 2*half+1][*]) of tile (tm, tn) ----
    // What the slot timelines (tools/gemm_slots_h.py, tools/slots_fine.py) showed.  (1) The s256 epilogue inside a half-slot took
    // 2,650-3,400 cycles beside a 700-cycle sibling MFMA slot; its fixed part — three integer divisions for the tile / QKV row
    // coordinates, 64-bit per-lane address products, per-lane row tests with exec masking, eight branch-guarded bias reads —
    // is gone here: tile coordinates are carried incrementally, a QKV row's item is one `v_mul_hi_u32` by a host-computed
    // reciprocal, stores are a wave-uniform base + one 32-bit lane offset, the bias is always in LDS, full tiles skip the row
    // test.  (2) WHAT REMAINS IS THE STORE ISSUE RATE OF A CU: one `global_store_dwordx4` wave-instruction per ~92 cycles,
    // whatever it writes — 8 stores x 4 waves of a group = 2,950 cycles per half-epilogue, the same for one 16-byte piece per
    // lane in 64 different rows (this layout), for 16 rows x 64 contiguous bytes after an LDS transposition (built, measured,
    // dropped: the transposition only added 1,600 cycles of LDS time), with the CUs' tile phases spread over a tile time
    // (debug bit 4: no change — it is not a chip-wide burst limit), and as the guide's T21 note has it ("store-ISSUE-bound, not
    // bandwidth").  128 store instructions per 256 x 256 tile = 11.8k cycles of a K = 768 tile's ~36k; they overlap the sibling
    // group's MFMAs only while the accumulators are not needed again, i.e. over the boundary half-slots.
    // Stores leave block by block, right after their block's arithmetic, so that the queue drains under the next block's.
    // Also built, measured and dropped: PARKING the last block of a half in a wave-private LDS area and storing it one or two read
    // slots later (6 instead of 8 stores per half-epilogue slot: those slots shrank from 2,700-3,000 to 2,300-2,500 cycles) — the
    // stores issued from middle read slots then sit in the CU's in-order vector-memory pipeline in front of the LDS-DMA pieces of
    // the next MFMA slot, which land late: middle slots of 1,500-2,200 cycles appeared and the kernel lost 7 % (QKV 928 against
    // 998 TFLOP/s).  The vector-memory path (128 stores x 92 + 768 DMA pieces per tile) is as busy as the matrix pipe.
    auto epilogue = [&](int tm, int tn, int half, int sbuf, bool tail = false) {
        // The lane-dependent offsets are RECOMPUTED here from a laundered lane id: hoisted out of the K loop they stay live
        // across it, the kernel sits at 256 VGPRs, they are spilled, and every scratch reload is followed by an
        // `s_waitcnt vmcnt(0)` — which also waits for the LDS-DMA in flight and the stores.
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int frow = l2 & 31, fh = l2 >> 5;
        const bool full = (int64_t)(tm + 1) * SBM <= p.M;
        stores8 = full && !(dbg_bits & (1 | 32));        // (EPI_STREAM16: one more, the row statistics — S256_VMCNT_EPI)
        const int row0 = tm * SBM + grp * 128;                 // first row of the group's half of the tile
        const int col0 = tn * SBN + wq * 64;                   // first column of the wave's slice
        if (!(dbg_bits & 1)) {
            auto run = [&](auto FULL_T) {                      // FULL (compile time): no row test
                constexpr bool FULL = decltype(FULL_T)::value;
                // byte offset of the lane's 32 bytes (16 columns) inside a 32 x 32 block at (mi, ni)
                uint32_t lane_off, qk_item0 = 0;
                uint32_t qk_rowadd = 0, qk_istride = 0;
                if constexpr (EPI == EPI_QKVH16) {
                    // head-major QKV [item][head][q|k|v][token][64]: this wave's 64 columns are one (q|k|v, head) pair; element-row
                    // index of row m: R = item * (3 heads - 1) S + m + (3 head + which) S,  item = m / S by reciprocal multiplication
                    const uint32_t qk_S = (uint32_t)p.qkv_S;
                    qk_istride = (uint32_t)(3 * p.qkv_heads - 1) * qk_S;
                    const uint32_t Dm = (uint32_t)p.qkv_heads * 64u, n64 = (uint32_t)col0;
                    const uint32_t wq_ = n64 >= 2 * Dm ? 2u : (n64 >= Dm ? 1u : 0u);
                    qk_rowadd = (((n64 - wq_ * Dm) >> 6) * 3u + wq_ + (uint32_t)p.qkv_which0) * qk_S;
                    lane_off = (uint32_t)(fh * 32);
                    (void)qk_item0;
                } else {
                    lane_off = (uint32_t)(frow * p.ldo * 2 + fh * 32);
                }
                f4 bbv[4];         // the lane's 16 bias values of column block ni (the same for every 32-row block)
                if constexpr (EPI == EPI_STREAM16) {
                    // ---- x += acc + bias in the fp16 stream (in place) + the rows' sums over this wave's 64 columns ----
                    // The old values are in `xo` (stream_load, two slots ago; behind them in the queue: the LDS-DMA pieces of the MFMA slot in
                    // between — 4 after a last step's Mlo, 8 after a first step's).  A block's packed results go out right behind its arithmetic.
                    V8 oo[4][2];
                    if (tail) S256_VMCNT(0); else if (half == 0) S256_VMCNT(4); else S256_VMCNT(8);

                    // the four blocks' sums wait in a wave-private LDS row — 2 KiB per wave behind the bias — instead of four registers
                    f2* sSum = (f2*)(sBias + 1024) + (wave * 4) * 64 + lane;
                    auto blk_ptr = [&](int b) {
                        const int mi = 2 * half + (b & 1), ni = b >> 1;
                        return (char*)p.out + ((int64_t)(row0 + mi * 32) * p.ldo + col0 + ni * 32) * 2 + lane_off;
                    };
                    auto comp = [&](int b) {
                        const int mi = 2 * half + (b & 1), ni = b >> 1;
                        // a CLS row (m = item * S, its stream is the executor's fp32 xc) receives the delta alone
                        const uint32_t m = (uint32_t)(row0 + mi * 32 + frow);
                        const bool cls = m - __umulhi(m, qkv_magic) * (uint32_t)p.qkv_S == 0u;
                        f2 s2 = (f2){0.f, 0.f}, q2 = (f2){0.f, 0.f};        // even / odd columns apart: v_pk_add_f32 / v_pk_fma_f32
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            // eight columns at a time, bias re-read from LDS per block (registers, as in the LNA epilogue)
                            const f4 b0 = *(const f4*)(sBias + col0 + ni * 32 + 16 * fh + 8 * hf), b1 = *(const f4*)(sBias + col0 + ni * 32 + 16 * fh + 8 * hf + 4);
                            // (the launder pins the conversions HERE: left alone they are hoisted above the full / ragged branch, all 16 of
                            //  a block at once, into a slot that has ~30 free registers: 39-67 spilled)
                            V8 x8 = xo[b][hf];
                            asm volatile("" : "+v"(x8));
                            if (cls) x8 = (V8)(_Float16)0;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                f2 t = (f2){acc[mi][ni][8 * hf + 2 * k], acc[mi][ni][8 * hf + 2 * k + 1]} + (k < 2 ? (f2){b0[2 * k], b0[2 * k + 1]} : (f2){b1[2 * k - 4], b1[2 * k - 3]});
                                t += (f2){(float)x8[2 * k], (float)x8[2 * k + 1]};
                                s2 += t;
                                q2 = __builtin_elementwise_fma(t, t, q2);
                                // converted at once: the eight sums of a block never sit in registers together
                                oo[b][hf][2 * k] = T::from_f32(t[0]); oo[b][hf][2 * k + 1] = T::from_f32(t[1]);
                            }
                            S256_FENCE();
                        }
                        sSum[b * 64] = (f2){s2[0] + s2[1], q2[0] + q2[1]};
                        S256_FENCE();
                    };
                    auto store = [&](int b) {
                        const int mi = 2 * half + (b & 1);
                        char* op = blk_ptr(b);
                        if (FULL || (int64_t)(row0 + mi * 32 + frow) < p.M) {
                            *(V8*)op = oo[b][0];
                            *(V8*)(op + 16) = oo[b][1];
                        }
                        S256_FENCE();
                    };
                    comp(0); store(0);
                    comp(1); store(1);
                    comp(2); store(2);
                    comp(3); store(3);
                    // the lane's two rows over this wave's 64 columns: lanes l and l ^ 32 hold the two 16-column halves of both.  One
                    // v_permlane32_swap per statistic leaves row block 2 half in lanes 0..31 and row block 2 half + 1 in lanes 32..63,
                    // one 8-byte store per lane: slice-major [N / 64][Mpad] so that a wave writes two 256-byte runs
                    // (inline asm: this hipcc's __builtin_amdgcn_permlane32_swap returns the first register twice — `v_add_f32 v1, v1, v1`
                    //  behind the swap — so the builtin's sum is 2 x one half; the swap exchanges a[32..63] with b[0..31] in place)
                    auto both = [](float a, float b) {
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
                        return a + b;
                    };
                    const f2 a0 = sSum[0], a1 = sSum[64], a2 = sSum[128], a3 = sSum[192];       // blocks (mi, ni) = (0,0) (1,0) (0,1) (1,1)
                    const f2 st = (f2){both(a0[0] + a2[0], a1[0] + a3[0]), both(a0[1] + a2[1], a1[1] + a3[1])};
                    const int64_t slot = (int64_t)(col0 >> 6) * ((int64_t)tiles_m * SBM);
                    if (!(dbg_bits & 256)) *(f2*)(p.rowpart + (slot + row0 + (2 * half + fh) * 32 + frow) * 2) = st;      // (ablation bit: no statistics store)
                    return;
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int mi = 2 * half + (b & 1), ni = b >> 1;
                    f2 g[8];       // bias added pairwise: v_pk_add_f32 (8 instead of 16 v_add_f32 per block)
                    V8 o0, o1;
                    if (!LNA && (b & 1) == 0) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) bbv[q4] = *(const f4*)(sBias + col0 + ni * 32 + 16 * fh + 4 * q4);
                    }
                    if constexpr (LNA) {
                        // rstd_m acc + bias'_n (the weights are gamma-folded AND centred, rowops.hip: the mean term is gone): one v_pk_fma_f32
                        // per pair where the plain product has a v_pk_add_f32.  The statistic is MATERIALISED as a register pair.  Left to
                        // the compiler it is broadcast by op_sel straight from the ds_read destination (`v_pk_fma_f32 ..., v[st], ...
                        // op_sel:[0,1,0]` right behind the `s_waitcnt lgkmcnt(0)`), and in the first version of this epilogue that form
                        // returned a ZERO product in the low results of lanes 48..63 of the first block of a hi half-epilogue — a few thousand
                        // of 8.5e8 outputs per launch, different ones every run, with or without the statistics DMA in flight
                        // (tools/gemm_lna.py checks every element); with the v_mov in between: clean over every run made since.
                        // (the bias is re-read from LDS per block, eight columns at a time, behind a scheduling fence: held across the two
                        //  row blocks as in the plain epilogue it costs the GELU variant the 3 registers it does not have)
                        S256_FENCE();
                        const float st = sStat[sbuf * 256 + grp * 128 + mi * 32 + frow];
                        f2 sx = (f2){st, st};
                        asm volatile("" : "+v"(sx));
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const f4 b0 = *(const f4*)(sBias + col0 + ni * 32 + 16 * fh + 8 * hf), b1 = *(const f4*)(sBias + col0 + ni * 32 + 16 * fh + 8 * hf + 4);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                g[4 * hf + k] = __builtin_elementwise_fma((f2){acc[mi][ni][8 * hf + 2 * k], acc[mi][ni][8 * hf + 2 * k + 1]}, sx,
                                                                          k < 2 ? (f2){b0[2 * k], b0[2 * k + 1]} : (f2){b1[2 * k - 4], b1[2 * k - 3]});
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            g[k] = (f2){acc[mi][ni][2 * k], acc[mi][ni][2 * k + 1]} + (f2){bbv[k >> 1][2 * (k & 1)], bbv[k >> 1][2 * (k & 1) + 1]};
                    }
                    // (four pairs at a time: the eight-pair form needs 48 temporaries on top of the fragments that stay live
                    //  across this slot, and spilled)
                    if constexpr (EPI == EPI_GELU16) { gelu_erf_fast2xN<4>(g); gelu_erf_fast2xN<4>(g + 4); }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        o0[e] = T::from_f32(g[e / 2][0]); o0[e + 1] = T::from_f32(g[e / 2][1]);
                        o1[e] = T::from_f32(g[4 + e / 2][0]); o1[e + 1] = T::from_f32(g[4 + e / 2][1]);
                    }
#ifdef S256_TIMELINE
                    if (dbg_bits & 32) { asm volatile("" :: "v"(o0), "v"(o1)); continue; }     // ablation: the arithmetic alone
#endif
                    const int mrow = row0 + mi * 32;                       // (wave-uniform) first row of the block
                    char* op;
                    bool ok = true;
                    if constexpr (EPI == EPI_PATCH16) {
                        // patch row m of image m / P -> token row m + m / P + 1 (one v_mul_hi by the host's reciprocal of P)
                        const uint32_t m = (uint32_t)(mrow + frow);
                        const uint32_t R = m + __umulhi(m, qkv_magic) + 1u;
                        op = (char*)p.out + ((uint32_t)(R * (uint32_t)(p.ldo * 2)) + (uint32_t)((col0 + ni * 32) * 2) + (uint32_t)(fh * 32));
                        if (!FULL) ok = (int64_t)m < p.M;
                    } else if constexpr (EPI == EPI_QKVH16) {
                        const uint32_t m = (uint32_t)(mrow + frow);
                        const uint32_t item = __umulhi(m, qkv_magic);        // m / S (exact: m * S < 2^32, checked by the launcher)
                        const uint32_t R = item * qk_istride + m + qk_rowadd;
                        op = (char*)p.out + ((uint32_t)(R * 128u) + (uint32_t)(ni * 64) + lane_off);
                        if (!FULL) ok = (int64_t)m < p.M;
                    } else {
                        char* base = (char*)p.out + ((int64_t)mrow * p.ldo + col0 + ni * 32) * 2;      // wave-uniform
                        op = base + lane_off;
                        if (!FULL) ok = (int64_t)(mrow + frow) < p.M;
                    }
                    if constexpr (EPI == EPI_OUT16 || EPI == EPI_GELU16) {
                        if (FULL && (dbg_bits & 512)) {      // (full row tiles only: the re-addressed rows are not tested against M)
                            // experiment (round 5, timing only — the DATA lands in the wrong places): the same two stores per block, but each
                            // instruction writes 8 complete 128-byte rows of the wave's 64-column slice (8 cache lines) instead of two 16-byte
                            // pieces in each of 32 rows (32 lines).  What would line-complete stores be worth if the rearrangement were free?
                            char* b0 = (char*)p.out + ((int64_t)(mrow + 16 * ni + (l2 >> 3)) * p.ldo + col0) * 2 + (l2 & 7) * 16;
                            *(V8*)b0 = o0;
                            *(V8*)(b0 + (int64_t)8 * p.ldo * 2) = o1;
                            continue;
                        }
                    }
                    if (ok) {
                        if (dbg_bits & 64) {          // experiment: non-temporal stores (the output must not evict the W panel from the L2)
                            __builtin_nontemporal_store(o0, (V8*)op);
                            __builtin_nontemporal_store(o1, (V8*)(op + 16));
                        } else {
                            *(V8*)op = o0;
                            *(V8*)(op + 16) = o1;
                        }
                    }
                }
            };
            if (full) run(std::true_type{}); else run(std::false_type{});
        }
        // the accumulators are dead now (the next K-step on them is a tile's first and starts from C = 0); an empty asm that
        // "defines" them tells the register allocator so — otherwise it copies them before the bias add
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                if ((mi >> 1) == half) asm volatile("" : "=v"(acc[mi][ni]));
    };

    const long long dbg_t0 = __builtin_readcyclecounter();
    // development aid (build with -DS256_TIMELINE, run with debug bit 16): wave 0 of each group stamps the cycle counter
    // at slot boundaries into LDS, dumped to out[] at the end; tools/gemm_slots.py prints the timeline.  Compiled out by
    // default: even the disabled checks cost ~5 % (everything between MFMAs does).
#ifdef S256_TIMELINE
    unsigned* sStamp = (unsigned*)(smem + 2 * S_STAGE_BYTES + STG_BIAS_BYTES - 4096) + grp * 512;     // (timeline builds: the last 4 KiB of the bias area, N <= 3072)
    int dbg_n = 0;
    const bool dbg_on = (dbg_bits & 16) && blockIdx.x == 0 && wq == 0;
    auto stamp = [&]() {
        if (dbg_on && dbg_n < 512) {
            if (lane == 0) sStamp[dbg_n] = (unsigned)(__builtin_readcyclecounter() - dbg_t0);
            ++dbg_n;
        }
    };
    auto stamp2 = [&]() {
#ifdef S256_FINE
        stamp();
#endif
    };
#else
    auto stamp = [] {};
    auto stamp2 = [] {};
    (void)stamp2;
#endif
    if (dbg_bits & 4) {                         // experiment: spread the CUs' tile phases over ~one tile time
        const int units = (int)(((unsigned)pid * 40503u) >> 5) & 63;
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    }
    if (dbg_bits & 8) {                         // experiment: one phase per XCD (L2 sharing inside an XCD is kept), spread over
        const int units = (int)(blockIdx.x & 7) * nk * ((dbg_bits >> 8) & 15) / 12;     // (debug>>8)&15 kilo-cycles per XCD at K=768
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    }
    for (int i = tid; i < p.N; i += 512) sBias[i] = p.bias ? p.bias[i] : 0.f;
    __syncthreads();
    // LNA: the row statistics of the workgroup's ti-th tile (rstd of its 256 rows) come in by LDS-DMA, ONE 1-KiB instruction per tile
    // (wave 0 of group A: lane i brings rows 4i .. 4i+3), issued at the top of the tile's first [Rhi + E] slot of group A — OLDER
    // than that slot's 8 stores, so the slot's own `vmcnt(8)` covers it, and the barrier behind that wait publishes it a whole
    // epilogue before its first reader (group A's lo half-epilogue one K-step later at the earliest; group B runs a slot behind).
    // Two buffers: the previous tile's hi half-epilogues (A in this slot, B in the next) read the other one.
    auto stat_dma = [&](int tm, int sbuf) {
        if constexpr (LNA) {
            if (grp == 0 && wq == 0)
                glds16((const char*)p.rowstat + ((int64_t)tm * SBM + 4 * lane) * 4, (char*)sStat + sbuf * 1024);
        }
    };

    // ---- prologue: what the steady-state rules would have issued before slot 0 ----
    if (grp == 0) {
        for (int j = 0; j < 4; ++j) piece_W(0, 0, j);
        for (int j = 0; j < 4; ++j) piece_A(0, 1, j);
        S256_VMCNT(0);
    } else {
        for (int j = 0; j < 4; ++j) piece_A(0, 0, j);
        for (int j = 0; j < 4; ++j) piece_W(0, 1, j);
        if (nsteps > 1) {
            for (int j = 0; j < 4; ++j) piece_A(1, 0, j);
            for (int j = 0; j < 8; ++j) piece_W(1, j >> 2, j & 3);
            S256_VMCNT(12);
        } else {
            S256_VMCNT(0);
        }
    }
    S256_BARRIER();                                   // P: W(0) and both A-halves of step 0 are in LDS

    // One group's program; B runs the same one slot behind (its extra barrier in front, A's at the end).  `stores8`: the
    // half-epilogue of this slot issued exactly 8 store instructions (full row tile, stores enabled), all YOUNGER than the
    // loads the wait is for: `vmcnt(8)` lets them fly on.
    using std::integral_constant;
    Plan qprev = make_plan(-1, -1);    // pieces 4..7: what the prologue loaded for step 1 (re-issued identically at s = 0)
    int ti = 0;                        // index of the tile being computed in this workgroup's walk
    if (grp == 1) S256_BARRIER();      // B: slot 0 (A is in Rlo(0))
    for (int s = 0; s < nsteps;) {
        {
            // ======== first K-step of a tile: Rlo | Mlo | Rhi + E(hi, previous tile) | Mhi ========
            if (s > 0) stream_load(prev_tm, prev_tn, 1);
            read_lo(s);
            const Plan q = make_plan(s, 0);
            S256_LGKM0();
            S256_FENCE();              // nothing to wait for: the only vector-memory operations in flight are stores
            stamp();
            S256_BARRIER();
            stamp();
            mfma_half(integral_constant<int, H_FIRST_LO>{}, q, qprev);
            stamp();
            S256_BARRIER();
            stamp();
            stat_dma(cur_tm, ti & 1);
            if (s > 0) epilogue(prev_tm, prev_tn, 1, (ti - 1) & 1);
            read_hi(s);
            S256_LGKM0();
            // the 8 pieces of Mlo (and, older, the 8 stores of the lo half); the 8 stores just issued stay in flight
            if (s > 0 && stores8) S256_VMCNT_EPI(); else S256_VMCNT(0);
            stamp();
            S256_BARRIER();
            stamp();
            mfma_half(integral_constant<int, H_FIRST_HI>{}, q, qprev);
            stamp();
            S256_BARRIER();
            ++s;
        }
        for (int kt = 1; kt < nk - 1; ++kt, ++s) {
            // ======== middle K-steps: R | M (as gemm16_s256.hip) ========
            read_step(s);
            const Plan q = make_plan(s, kt);
            S256_LGKM0();
            S256_VMCNT(0);             // the pieces of the previous MFMA slot (after a first step: the hi half's stores too, two slots old)
            stamp();
            S256_BARRIER();
            stamp();
            mfma_step(q, false);
            stamp();
            S256_BARRIER();
        }
        {
            // ======== last K-step of a tile: Rlo | Mlo | Rhi + E(lo) | Mhi ========
            stream_load(cur_tm, cur_tn, 0);
            read_lo(s);
            const Plan q = make_plan(s, nk - 1);
            S256_LGKM0();
            if constexpr (EPI == EPI_STREAM16) S256_VMCNT(8); else S256_VMCNT(0);      // (the 8 stream loads just issued fly on)
            stamp();
            S256_BARRIER();
            stamp();
            mfma_half(integral_constant<int, H_LAST_LO>{}, q, qprev);
            stamp();
            S256_BARRIER();
            stamp();
            epilogue(cur_tm, cur_tn, 0, ti & 1);
            read_hi(s);
            S256_LGKM0();
            if (stores8) S256_VMCNT_EPI(); else S256_VMCNT(0);      // the 4 pieces of Mlo; the 8 stores just issued stay in flight
            stamp();
            S256_BARRIER();
            stamp();
            mfma_half(integral_constant<int, H_LAST_HI>{}, q, qprev);
            stamp();
            S256_BARRIER();
            qprev = q;
            ++s;
            prev_tm = cur_tm; prev_tn = cur_tn;
            cur_tm = nxt_tm; cur_tn = nxt_tn;
            ++ti;
            if (ti + 1 < my_tiles) walk(ti + 1, nxt_tm, nxt_tn);
        }
    }
    stream_load(prev_tm, prev_tn, 1);
    epilogue(prev_tm, prev_tn, 1, (ti - 1) & 1, true);     // the last tile's hi rows
    if (grp == 0) S256_BARRIER();      // matches B's last slot
#ifdef S256_TIMELINE
    if (dbg_on && lane == 0)
        for (int i = 0; i < 512; ++i) ((unsigned*)p.out)[8192 + grp * 1024 + i] = sStamp[i];
#endif
    // The plan is unconditional, so the last MFMA slots issued LDS-DMA loads nobody reads: they must have landed before
    // this workgroup's LDS can be handed to another workgroup.
    S256_VMCNT(0);
    if ((dbg_bits & 16) && tid == 0) {          // development aid: cycles and K-steps of this workgroup into out[]
        ((long long*)p.out)[2 * blockIdx.x] = __builtin_readcyclecounter() - dbg_t0;
        ((long long*)p.out)[2 * blockIdx.x + 1] = nsteps;
    }
}

template <typename T, int EPI, bool LNA = false>
int launch_epi(const Gemm16Args& a, hipStream_t s) {
    static OncePerDevice attr;
    auto kern = gemm16_h256_kernel<T, EPI, LNA>;
    constexpr int LDS = 2 * S_STAGE_BYTES + STG_BIAS_BYTES + STG_TILE_BYTES;       // 160 KiB: the whole LDS of a CU
    if (attr.first())
        IISAN_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int tiles_m = (int)ceil_div(a.M, SBM), tiles_n = a.N / SBN;
    const int64_t ntiles = (int64_t)tiles_m * tiles_n;
    const int cus = iisan_cu_count();
    int grid = (int)(ntiles < cus ? ntiles : cus);
    grid = (grid + 7) / 8 * 8;
    // EPI_QKVH16: item = row / S as one v_mul_hi_u32 by floor(2^32 / S) + 1 — exact while row * S < 2^32 (gemm16_h256_applicable)
    const int div = EPI == EPI_PATCH16 ? a.patch_P : a.qkv_S;       // (EPI_STREAM16: S, for the CLS-row test)
    const uint32_t magic = div > 0 ? (uint32_t)((1ull << 32) / (uint64_t)div) + 1u : 0u;
    Gemm16Args b = a;
    // the panel walk needs at least one row tile per XCD slab and a panel narrower than the tile row
    if (b.walk_c >= tiles_n || tiles_m < 8 || b.walk_c < 0) b.walk_c = 0;
    if (b.walk_c > 0 && (b.walk_h <= 0 || b.walk_h > tiles_m)) b.walk_h = tiles_m;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, b, tiles_m, tiles_n, magic);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

template <typename T>
int launch_t(int mode, const Gemm16Args& a, hipStream_t s) {
    if (a.rowstat) {         // LayerNorm in the epilogue: fp16 operands, the two products that follow a LayerNorm
        if constexpr (std::is_same<T, F16>::value) {
            if (mode == EPI_QKVH16) return launch_epi<F16, EPI_QKVH16, true>(a, s);
            if (mode == EPI_GELU16) return launch_epi<F16, EPI_GELU16, true>(a, s);
        }
        iisan_set_error("gemm16_h256: row statistics with epilogue mode %d / bf16 operands not supported", mode);
        return IISAN_EBADSHAPE;
    }
    switch (mode) {
        case EPI_OUT16: return launch_epi<T, EPI_OUT16>(a, s);
        case EPI_GELU16: return launch_epi<T, EPI_GELU16>(a, s);
        case EPI_QKVH16: return launch_epi<T, EPI_QKVH16>(a, s);
        case EPI_PATCH16: return launch_epi<T, EPI_PATCH16>(a, s);
        case EPI_STREAM16:
            if constexpr (std::is_same<T, F16>::value) return launch_epi<F16, EPI_STREAM16>(a, s);
            iisan_set_error("gemm16_h256: the stream epilogue needs fp16 operands"); return IISAN_EBADSHAPE;
        default: iisan_set_error("gemm16_h256: epilogue mode %d not supported", mode); return IISAN_EBADSHAPE;
    }
}

}  // namespace

bool gemm16_h256_applicable(int mode, const Gemm16Args& a) {
    if (!((mode == EPI_OUT16 || mode == EPI_GELU16 || mode == EPI_QKVH16 || mode == EPI_PATCH16 || mode == EPI_STREAM16) && a.N % SBN == 0 && a.N * 4 <= STG_BIAS_BYTES && a.K % SBK == 0 &&
          a.K / SBK >= 2 && (int64_t)a.lda * 2 * SBM < (1ll << 31) && (int64_t)a.ldw * 2 * SBN < (1ll << 31)))
        return false;
    if (a.rowstat && !((mode == EPI_QKVH16 || mode == EPI_GELU16) && a.N <= LNA_MAX_N)) return false;
    const int64_t rows = ceil_div(a.M, SBM) * SBM;
    if (mode == EPI_QKVH16)      // 32-bit byte offsets into the head-major tensor, exact reciprocal division.  The layout is always
                                 // [item][head][q|k|v][S][64]: element-row index R reaches rows * 3 * heads WHATEVER qkv_which0 is (a
                                 // K/V-only product of the CLS-pruned last block still addresses the full tensor) — ADVICE r3: the bound
                                 // used (3 - which0) and let row counts in [932k, 1.40M) wrap at 12 heads
        return a.qkv_S > 0 && rows * a.qkv_S < (1ll << 32) && rows * 3 * (int64_t)a.qkv_heads * 128 + 128 < (1ll << 32);
    if (mode == EPI_STREAM16)    // exact reciprocal division of the row index; in place: the output row stride is N
        return a.qkv_S > 0 && rows * a.qkv_S < (1ll << 32) && a.rowpart && a.ldo == a.N && a.N <= 1024 && (int64_t)a.ldo * 2 * 16 + 64 < (1ll << 31);
    if (mode == EPI_PATCH16)     // 32-bit byte offsets of the remapped rows, exact reciprocal division
        return a.patch_P > 0 && rows * a.patch_P < (1ll << 32) && (rows + rows / a.patch_P + 2) * (int64_t)a.ldo * 2 < (1ll << 32);
    return (int64_t)a.ldo * 2 * 16 + 64 < (1ll << 31);       // the 32-bit lane offset of a 16-row store
}

int launch_gemm16_h256(int dtype16, int mode, const Gemm16Args& a, hipStream_t s) {
    return dtype16 == IISAN_BF16 ? launch_t<BF16>(mode, a, s) : launch_t<F16>(mode, a, s);
}

// host-side predicate, exported for the CPU tests (tests/test_host_logic.py): no device is touched
extern "C" int32_t iisan_gemm16_h256_applicable(int32_t mode, int64_t M, int32_t N, int32_t K, int32_t qkv_S, int32_t qkv_heads,
                                                int32_t qkv_which0) {
    Gemm16Args a{};
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldo = N; a.qkv_S = qkv_S; a.qkv_heads = qkv_heads; a.qkv_which0 = qkv_which0;
    return gemm16_h256_applicable(mode, a) ? 1 : 0;
}
