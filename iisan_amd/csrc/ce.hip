// Fused in-batch debiased cross-entropy (forward + backward) — the [T, M] logits matrix is never materialised.
// Replaces the loss section of ModelMM.forward (Code_Uncached/model/model.py:81-104), including its O(bs)
// Python masking loop (model.py:92-100) whose autograd graph dominates the reference's Cached step at bs=1024
// (SURVEY.md §3.2).  Exact semantics kept, in the reference's order:
//     z[row,c]  = prec[row]·score[c] - log pop[id_c]
//     z[:,c]    = -1e4   if slot c is history padding (position < S and log_mask == 0)            (model.py:88-89)
//     z[row,c]  = -1e4   if id_c occurs in the row's own sequence and c is not the row's positive  (model.py:92-100)
//     loss      = mean over rows with log_mask != 0 of  logsumexp(z[row,:]) - z[row, label]       (model.py:102-104)
//
// One workgroup owns 16 rows of X (prec rows for fwd / d_prec, score rows for d_score) and streams the other
// matrix in 16-row tiles, four waves taking tiles round-robin.  Logit tiles are computed TRANSPOSED on the f32
// matrix cores (v_mfma_f32_16x16x4_f32, K = E = 64) so a lane owns one X row and 4 Y columns per tile: the online
// log-sum-exp is per-lane + two xor-shuffles, and the dZ registers are already the B operand of the second product
// dX^T = Y^T·dZ^T.  The Y tile is staged once per wave in LDS (272-byte rows: conflict-free ds_read_b128).
#include "common.h"

namespace {

constexpr int E = 64;
constexpr int MAXS1 = 16;         // sequences of up to 16 slots keep their ids in registers / LDS (longer ones take the slow loop)
constexpr int YLD = 68;           // padded LDS row (floats)
constexpr float MASKV = -1e4f;

struct CeBufs {
    int* ids32;      // [M]
    float* debias;   // [M] log pop[id]
    int* colpad;     // [M] 1 if the slot is history padding
    float* lse;      // [T]
    float* rowloss;  // [T]
    float* nvalid;   // [1]
    float* dprec;    // [T, E] d loss / d prec for d_loss = 1, left by the fused forward pass (ce_rowpass_kernel<CE_FUSED>)
};

void carve(WsCarver& c, CeBufs& b, int64_t bs, int S) {
    const size_t M = (size_t)bs * (S + 1), T = (size_t)bs * S;
    b.ids32 = c.take<int>(M);
    b.debias = c.take<float>(M);
    b.colpad = c.take<int>(M);
    b.lse = c.take<float>(T);
    b.rowloss = c.take<float>(T);
    b.nvalid = c.take<float>(4);
    b.dprec = c.take<float>(T * E);
}

__global__ void ce_prep_kernel(const int64_t* __restrict__ ids, const float* __restrict__ log_mask,
                               const float* __restrict__ pop, int64_t n_pop, CeBufs b, int64_t bs, int S) {
    const int64_t M = bs * (S + 1);
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < M; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = c / (S + 1);
        const int p = (int)(c - i * (S + 1));
        const int64_t id = ids[c];
        b.ids32[c] = (int)id;
        b.debias[c] = (id >= 0 && id < n_pop) ? logf(pop[id]) : __builtin_nanf("");     // bad id: loud NaN loss, no wild read
        b.colpad[c] = (p < S && log_mask[i * S + p] == 0.f) ? 1 : 0;
    }
}

// n_valid = #{log_mask != 0}; single block, fixed order
__global__ __launch_bounds__(256) void ce_count_kernel(const float* __restrict__ log_mask, int64_t T, float* out) {
    __shared__ float red[256];
    float s = 0.f;
    // (16-byte loads, all of a thread's requests in flight: one dependent 4-byte load per iteration made this single-workgroup pass 10.7 us at T = 10,240;
    //  the counts are small integers: any summation order is exact)
    const int64_t T4 = (T % 4 == 0 && ((uintptr_t)log_mask & 15) == 0) ? T / 4 : 0;
    for (int64_t i = threadIdx.x; i < T4; i += 256) {
        const f4 v = ((const f4*)log_mask)[i];
        s += (v[0] != 0.f ? 1.f : 0.f) + (v[1] != 0.f ? 1.f : 0.f) + (v[2] != 0.f ? 1.f : 0.f) + (v[3] != 0.f ? 1.f : 0.f);
    }
    for (int64_t i = 4 * T4 + threadIdx.x; i < T; i += 256) s += log_mask[i] != 0.f ? 1.f : 0.f;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ rowloss, int64_t T, const float* nvalid,
                                                        float* loss) {
    __shared__ float red[256];
    float s = 0.f;
    // 16-byte loads (as ce_count_kernel); the order is fixed — thread t adds the groups t, t + 256, ... of four rows, each as (r0 + r1) + (r2 + r3)
    const int64_t T4 = (T % 4 == 0 && ((uintptr_t)rowloss & 15) == 0) ? T / 4 : 0;
    for (int64_t i = threadIdx.x; i < T4; i += 256) {
        const f4 v = ((const f4*)rowloss)[i];
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (int64_t i = 4 * T4 + threadIdx.x; i < T; i += 256) s += rowloss[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / nvalid[0];
}

__device__ __forceinline__ bool id_in_seq(const int* __restrict__ ids32, int64_t seq, int S1, int idc) {
    bool hit = false;
    for (int p = 0; p < S1; ++p) hit |= ids32[seq * S1 + p] == idc;
    return hit;
}

enum { CE_FWD = 0, CE_DPREC = 1, CE_DSCORE = 2, CE_FUSED = 3 };

// X rows: prec (FWD, DPREC) or score (DSCORE).  Y rows: the other matrix.
// RS1: slots per sequence held in registers by the row-fixed passes (11 = the reference's max_seq_len 10 + 1: five fewer
// id compares per logit than the generic 16)
template <int MODE, int RS1 = MAXS1>
__global__ __launch_bounds__(256) void ce_pass_kernel(const float* __restrict__ prec, const float* __restrict__ score,
                                                      const float* __restrict__ log_mask, CeBufs b, int64_t bs, int S,
                                                      float d_loss, float* __restrict__ dX) {
    __shared__ __attribute__((aligned(16))) float sY[4][16 * YLD];
    __shared__ float sRed[4][16][E + 4];
    // per-wave, per-tile metadata (the first version fetched it from global memory per logit: 3 + S1 loads — and in the
    // d_score pass 3 more plus two 64-bit divisions — for each of a lane's 4 logits, ~56 load instructions per 16 MFMAs:
    // the passes ran at 15 TF of the 157 TF f32 MFMA rate at bs=1024).
    //   FWD / DPREC (row fixed per lane): sMeta = ids | pad flags | debias of the tile's 16 columns.
    //   DSCORE (column fixed per lane):   sMeta = ids of the <=4 sequences the tile's 16 rows belong to,
    //                                     sRow  = per row: sequence index in that table | label column | lse | scale.
    __shared__ __attribute__((aligned(16))) int sMeta[4][64];
    __shared__ __attribute__((aligned(16))) int sRow[4][64];
    const int S1 = S + 1;
    const int64_t T = bs * S, M = bs * S1;
    const float* X = MODE == CE_DSCORE ? score : prec;
    const float* Y = MODE == CE_DSCORE ? prec : score;
    const int64_t NX = MODE == CE_DSCORE ? M : T, NY = MODE == CE_DSCORE ? T : M;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int64_t x0 = (int64_t)blockIdx.x * 16;
    const int64_t x = x0 + j;
    const bool xok = x < NX;
    const int64_t xc = xok ? x : NX - 1;

    // X fragments (B operand): e(ks, kq) = 16*kq + ks  -> 16 consecutive floats per lane
    float xb[16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const f4 t = *(const f4*)(X + xc * E + 16 * g + 4 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) xb[4 * v + e] = t[e];
    }

    // per-X-row constants
    int64_t row_seq = 0, row_label = 0;
    float row_lse = 0.f, row_scale = 0.f;
    bool row_valid = false;
    int col_id = 0, col_pad = 0;
    float col_debias = 0.f;
    if (MODE != CE_DSCORE) {
        row_seq = xc / S;
        row_label = row_seq * S1 + (xc - row_seq * S) + 1;
        row_valid = xok && log_mask[xc] != 0.f;
        if (MODE == CE_DPREC) {
            row_lse = b.lse[xc];
            row_scale = row_valid ? d_loss / b.nvalid[0] : 0.f;
        }
    } else {
        col_id = b.ids32[xc];
        col_pad = b.colpad[xc];
        col_debias = b.debias[xc];
    }

    // FWD / DPREC: this lane's row belongs to one sequence for the whole pass: its ids live in registers
    const bool fast = S1 <= RS1 && S >= 5 && M < (1ll << 31);
    int rid[RS1];
#pragma unroll
    for (int p = 0; p < RS1; ++p) rid[p] = -1;
    if (MODE != CE_DSCORE && fast) {
#pragma unroll
        for (int p = 0; p < RS1; ++p)
            if (p < S1) rid[p] = b.ids32[row_seq * S1 + p];
    }
    int* myMeta = sMeta[wave];
    int* myRow = sRow[wave];
    const float dscale = MODE == CE_DSCORE ? d_loss / b.nvalid[0] : 0.f;

    // CE_FUSED: forward and d_prec in ONE pass, flash-attention style: the accumulators hold sum_y exp(z - run_m) * score[y]
    // against a running ROW maximum (common to the four lanes of a row, because the MFMA sums over their columns) and are
    // rescaled when it moves — after the first tiles it almost never does, and the test is one ballot per tile.
    float run_m = MODE == CE_FUSED ? -3.0e38f : -INFINITY, run_l = 0.f, zlab = 0.f;
    f4 dacc[4];
#pragma unroll
    for (int et = 0; et < 4; ++et) dacc[et] = (f4){0.f, 0.f, 0.f, 0.f};

    float* myY = sY[wave];
    const int64_t ntiles = (NY + 15) / 16;
    for (int64_t yt = wave; yt < ntiles; yt += 4) {
        const int64_t y0 = yt * 16;
        // stage the Y tile: 16 rows x 64 floats, lane -> row lane>>2, 16-float chunk lane&3
        {
            const int r = lane >> 2, ch = lane & 3;
            const int64_t yr = y0 + r;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                f4 t = {0.f, 0.f, 0.f, 0.f};
                if (yr < NY) t = *(const f4*)(Y + yr * E + ch * 16 + v * 4);
                *(f4*)(myY + r * YLD + ch * 16 + v * 4) = t;
            }
        }
        unsigned hitmask = 0;                        // DSCORE: bit s = this lane's column id occurs in sequence seq0 + s
        if (fast) {
            if (MODE != CE_DSCORE) {                 // columns y0 .. y0+15
                if (lane < 16) {
                    const int64_t cc = y0 + lane < NY ? y0 + lane : NY - 1;
                    myMeta[lane] = b.ids32[cc];
                    myMeta[16 + lane] = b.colpad[cc];
                    myMeta[32 + lane] = __float_as_int(b.debias[cc]);
                }
            } else {                                 // rows y0 .. y0+15 span at most 4 sequences (S >= 5)
                const unsigned seq0 = (unsigned)y0 / (unsigned)S;
                for (int i = lane; i < 4 * S1; i += 64) {
                    const int64_t sq_ = (int64_t)seq0 + i / S1;
                    myMeta[i] = sq_ < bs ? b.ids32[sq_ * S1 + (i % S1)] : -1;
                }
                if (lane < 16) {
                    const unsigned rw = (unsigned)(y0 + lane < NY ? y0 + lane : NY - 1);
                    const unsigned rs = rw / (unsigned)S;
                    myRow[lane] = (int)(rs - seq0);
                    myRow[16 + lane] = (int)(rs * (unsigned)S1 + (rw - rs * (unsigned)S) + 1u);
                    myRow[32 + lane] = __float_as_int(b.lse[rw]);
                    myRow[48 + lane] = __float_as_int(log_mask[rw] != 0.f ? dscale : 0.f);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (fast && MODE == CE_DSCORE) {
            if (RS1 == 11 && S1 == 11) {             // the reference's shape: 44 ids = 11 x ds_read_b128, indices known at compile time
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    typedef int i4 __attribute__((ext_vector_type(4)));
                    const i4 v = *(const i4*)(myMeta + 4 * i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) hitmask |= v[k] == col_id ? (1u << ((4 * i + k) / 11)) : 0u;
                }
            } else {
                for (int sidx = 0; sidx < 4; ++sidx) {
                    bool h = false;
                    for (int p = 0; p < S1; ++p) h |= myMeta[sidx * S1 + p] == col_id;       // wave-uniform addresses: LDS broadcast
                    hitmask |= h ? (1u << sidx) : 0u;
                }
            }
        }
        // Z^T tile: A = Y rows (i = lane&15), B = X rows
        // two independent accumulator chains (a single one is 16 MFMAs each waiting for the previous result)
        f4 z = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; v += 2) {
            const f4 ya = *(const f4*)(myY + j * YLD + 16 * g + 4 * v);
            const f4 yb = *(const f4*)(myY + j * YLD + 16 * g + 4 * v + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[e], xb[4 * v + e], z, 0, 0, 0);
                z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[e], xb[4 * v + 4 + e], z1, 0, 0, 0);
            }
        }
        z += z1;
        // lane holds z(x, y = y0 + 4g + r)
        float dz[4], fv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t y = y0 + 4 * g + r;
            const bool yok = y < NY;
            const int64_t yc = yok ? y : NY - 1;
            int64_t row, col, seq, label;
            int idc, pad;
            float deb;
            bool hit = false;
            float r_lse = 0.f, r_scale = 0.f;        // DSCORE fast path: per-row values from the tile table
            if (MODE != CE_DSCORE) {
                row = xc; col = yc; seq = row_seq; label = row_label;
                if (fast) {
                    const int cl = (int)(yc - y0);
                    idc = myMeta[cl]; pad = myMeta[16 + cl]; deb = __int_as_float(myMeta[32 + cl]);
                } else {
                    idc = b.ids32[col]; pad = b.colpad[col]; deb = b.debias[col];
                }
            } else {
                row = yc; col = xc;
                idc = col_id; pad = col_pad; deb = col_debias;
                if (fast) {
                    const int rr = (int)(yc - y0);
                    seq = 0;
                    label = myRow[16 + rr];
                    hit = (hitmask >> myRow[rr]) & 1u;
                    r_lse = __int_as_float(myRow[32 + rr]);
                    r_scale = __int_as_float(myRow[48 + rr]);
                } else {
                    seq = row / S; label = seq * S1 + (row - seq * S) + 1;
                }
            }
            float val = z[r] - deb;
            if (pad) val = MASKV;
            else if (col != label) {
                if (MODE != CE_DSCORE && fast) {
#pragma unroll
                    for (int p = 0; p < RS1; ++p) hit |= rid[p] == idc;
                } else if (!fast) {
                    hit = id_in_seq(b.ids32, seq, S1, idc);
                }
                if (hit) val = MASKV;
            }
            if (MODE == CE_FWD) {
                fv[r] = yok ? val : -INFINITY;        // folded into the running (max, sum) after the tile, four at a time
                if (yok && col == label) zlab = val;
            } else {
                float lse, scale;
                if (MODE == CE_DPREC) {
                    lse = row_lse; scale = row_scale;
                } else if (fast) {
                    lse = r_lse; scale = r_scale;
                } else {
                    lse = b.lse[row];
                    scale = log_mask[row] != 0.f ? d_loss / b.nvalid[0] : 0.f;
                }
                const float p = expf(val - lse);
                dz[r] = (yok && xok) ? (p - (col == label ? 1.f : 0.f)) * scale : 0.f;
            }
        }
        if (MODE == CE_FWD) {
            // one rescale per tile instead of a data-dependent branch with an exp on both sides per logit (5 exps per 4
            // logits instead of 8, no divergence); masked logits are finite (MASKV), structural padding is -inf -> 0
            const float m4 = fmaxf(fmaxf(fv[0], fv[1]), fmaxf(fv[2], fv[3]));
            const float mn = fmaxf(run_m, m4);
            if (mn > -INFINITY) {
                float add = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) add += expf(fv[r] - mn);
                run_l = run_l * expf(run_m - mn) + add;
                run_m = mn;
            }
        }
        if (MODE != CE_FWD) {
            // dX^T[e][x] += sum_y Y[y][e] dZ(x,y):  A = Y^T (i = e within tile et, k = y0 + 4kq + r), B = dz[r]
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int et = 0; et < 4; ++et) {
                    const float ya = myY[(4 * g + r) * YLD + 16 * et + j];
                    dacc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya, dz[r], dacc[et], 0, 0, 0);
                }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (MODE == CE_FWD) {
        // combine (m, l) and the label logit across the 4 lane groups and the 4 waves
        auto comb = [](float& m, float& l, float m2, float l2) {
            const float mn = fmaxf(m, m2);
            const float a = m == -INFINITY ? 0.f : l * expf(m - mn);
            const float c = m2 == -INFINITY ? 0.f : l2 * expf(m2 - mn);
            m = mn;
            l = a + c;
        };
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float m2 = __shfl_xor(run_m, o, 64), l2 = __shfl_xor(run_l, o, 64);
            comb(run_m, run_l, m2, l2);
            zlab += __shfl_xor(zlab, o, 64);
        }
        if (g == 0) {
            sRed[wave][j][0] = run_m;
            sRed[wave][j][1] = run_l;
            sRed[wave][j][2] = zlab;
        }
        __syncthreads();
        if (wave == 0 && g == 0 && xok) {
            float m = sRed[0][j][0], l = sRed[0][j][1], zl = sRed[0][j][2];
            for (int w = 1; w < 4; ++w) {
                comb(m, l, sRed[w][j][0], sRed[w][j][1]);
                zl += sRed[w][j][2];
            }
            const float lse = m + logf(l);
            b.lse[x] = lse;
            b.rowloss[x] = row_valid ? lse - zl : 0.f;
        }
    } else {
        // lane (j, g) holds dX[x0+j][16et + 4g + r]; sum over the 4 waves
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) sRed[wave][j][16 * et + 4 * g + r] = dacc[et][r];
        __syncthreads();
        for (int i = tid; i < 16 * E; i += 256) {
            const int rj = i / E, e = i - rj * E;
            if (x0 + rj < NX) dX[(x0 + rj) * E + e] = sRed[0][rj][e] + sRed[1][rj][e] + sRed[2][rj][e] + sRed[3][rj][e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-fixed passes (FWD, DPREC) with the per-logit work cut down (round 2).  rocprofv3 of the Cached step at bs = 1024 had
// the three passes at 476 / 586 / 634 us against 110 / 220 / 220 us of f32 MFMA time: they were VALU-bound — per logit
// eleven id compares for the false-negative mask, 64-bit index arithmetic, three LDS reads and a libm expf (~100 VALU
// instructions, four logits per lane and tile).  Here
//   * the false-negative test is done ONCE PER WAVE AND TILE: the 16 rows of a workgroup belong to at most four sequences
//     (S >= 5), lane l tests column l&15 against the ids of sequence slot l>>4 (held in its registers for the whole pass)
//     and a ballot turns the 64 answers into a mask every lane indexes with (its row's slot, the logit's column);
//   * column padding is a second ballot, the debias of a lane's four columns one 16-byte LDS read;
//   * exp(x) = v_exp_f32(x * log2 e): 2 instructions, relative error < 3e-6 for the |x| <= 40 that matter (the loss
//     tolerance is 2e-5, the gradients' 2e-4; masked logits underflow to 0 exactly as before);
//   * all indices are 32-bit (M < 2^31 is a launch condition).
// Same tile walk, same MFMA layout and the same reduction as ce_pass_kernel, which remains the generic path.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

template <int MODE, int RS1>
__global__ __launch_bounds__(256) void ce_rowpass_kernel(const float* __restrict__ prec, const float* __restrict__ score,
                                                         const float* __restrict__ log_mask, CeBufs b, int bs, int S,
                                                         float d_loss, float* __restrict__ dX, int dbg = 0) {
    static_assert(MODE == CE_FWD || MODE == CE_DPREC || MODE == CE_FUSED, "row-fixed passes only");
    __shared__ __attribute__((aligned(16))) float sY[4][16 * YLD];
    __shared__ float sRed[4][16][E + 4];
    __shared__ __attribute__((aligned(16))) int sMeta[4][64];
    const int S1 = S + 1;
    const int T = bs * S, M = bs * S1;
    const float* X = prec;
    const float* Y = score;
    const int NX = T, NY = M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * 16;
    const int x = x0 + j;
    const bool xok = x < NX;
    const int xc = xok ? x : NX - 1;

    float xb[16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const f4 t = *(const f4*)(X + (int64_t)xc * E + 16 * g + 4 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) xb[4 * v + e] = t[e];
    }
    const int row_seq = xc / S;
    const int row_label = row_seq * S1 + (xc - row_seq * S) + 1;
    const bool row_valid = xok && log_mask[xc] != 0.f;
    const float row_lse = MODE == CE_DPREC ? b.lse[xc] : 0.f;
    const float row_scale = (MODE == CE_DPREC && row_valid) ? d_loss / b.nvalid[0] : 0.f;
    const int seq0 = x0 / S;
    const int row_shift = (row_seq - seq0) * 16 + 4 * g;       // bit of (my row's slot, column 4g) in the tile's hit mask
    // ids of sequence slot g = lane>>4 of this workgroup, for the cooperative false-negative test
    int sid[RS1];
    {
        const int sq = seq0 + g;
#pragma unroll
        for (int p = 0; p < RS1; ++p) sid[p] = (p < S1 && sq < bs) ? b.ids32[sq * S1 + p] : -2;
    }

    // CE_FUSED: forward and d_prec in ONE pass, flash-attention style: the accumulators hold sum_y exp(z - run_m) * score[y]
    // against a running ROW maximum (common to the four lanes of a row, because the MFMA sums over their columns) and are
    // rescaled when it moves — after the first tiles it almost never does, and the test is one ballot per tile.
    float run_m = MODE == CE_FUSED ? -3.0e38f : -INFINITY, run_l = 0.f, zlab = 0.f;
    f4 dacc[4];
#pragma unroll
    for (int et = 0; et < 4; ++et) dacc[et] = (f4){0.f, 0.f, 0.f, 0.f};

    float* myY = sY[wave];
    int* myMeta = sMeta[wave];
    const int ntiles = (NY + 15) / 16;
    // the next tile's rows and column metadata travel in registers while the current tile is multiplied: with the loads at
    // the top of the iteration every tile waited for an L2 round trip (FWD 303 us against 110 us of MFMA time)
    f4 yreg[4];
    int mreg[3] = {0, 0, 0};
    // Branch-free: a tile index past the end re-reads the last tile, a row past NY re-reads row NY-1 (finite values that
    // meet dz = 0) — every conditional load was a saveexec / branch pair in the loop, and a join makes the waitcnt pass drain.
    auto prefetch = [&](int yt_) {
        const int y0_ = (yt_ < ntiles ? yt_ : ntiles - 1) * 16;
        const int r = lane >> 2, ch = lane & 3;
        const int yr = y0_ + r < NY ? y0_ + r : NY - 1;
        const float* yp = Y + (int64_t)yr * E + ch * 16;
#pragma unroll
        for (int v = 0; v < 4; ++v) yreg[v] = *(const f4*)(yp + v * 4);
        const int cc = y0_ + (lane & 15) < NY ? y0_ + (lane & 15) : NY - 1;
        mreg[0] = b.ids32[cc];
        mreg[1] = b.colpad[cc];
        mreg[2] = __float_as_int(b.debias[cc]);
    };
    prefetch(wave);
    for (int yt = wave; yt < ntiles; yt += 4) {
        const int y0 = yt * 16;
        {
            const int r = lane >> 2, ch = lane & 3;
#pragma unroll
            for (int v = 0; v < 4; ++v) *(f4*)(myY + r * YLD + ch * 16 + v * 4) = yreg[v];
        }
        if (lane < 16) {
            myMeta[lane] = mreg[0];
            myMeta[16 + lane] = mreg[1];
            myMeta[32 + lane] = mreg[2];
        }
        prefetch(yt + 4);
        __builtin_amdgcn_wave_barrier();
        // once per wave and tile: (sequence slot g, column j) hit bits and the 16 column-padding bits
        const int idc = myMeta[j];
        bool h = false;
#pragma unroll
        for (int p = 0; p < RS1; ++p) h |= sid[p] == idc;
        const unsigned long long hm = __ballot(h);
        const unsigned pm = (unsigned)__ballot(myMeta[16 + j] != 0);
        const unsigned hit4 = (unsigned)(hm >> row_shift) & 0xfu;
        const unsigned pad4 = (pm >> (4 * g)) & 0xfu;
        const f4 deb4 = *(const f4*)(myMeta + 32 + 4 * g);      // bit patterns of four floats

        f4 z = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; v += 2) {
            const f4 ya = *(const f4*)(myY + j * YLD + 16 * g + 4 * v);
            const f4 yb = *(const f4*)(myY + j * YLD + 16 * g + 4 * v + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#ifdef CE_ABLATE      // build with -DCE_ABLATE for tools/ce_ablate.py: even the untaken branch costs the row pass 130 us (450 -> 585)
                if (dbg & 1) { z[0] += ya[e] * xb[4 * v + e]; continue; }       // ablation (timing only): no logits product
#endif
                z = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[e], xb[4 * v + e], z, 0, 0, 0);
                z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[e], xb[4 * v + 4 + e], z1, 0, 0, 0);
            }
        }
        z += z1;
        // the second product's 16 transposed tile reads are requested NOW, fenced, and land behind the mask / exp arithmetic:
        // left to the compiler each pair sat right in front of its two MFMAs behind an lgkmcnt(0) (eight exposed LDS
        // round trips per tile)
        float yt16[4][4];
        if (MODE != CE_FWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int et = 0; et < 4; ++et) yt16[r][et] = myY[(4 * g + r) * YLD + 16 * et + j];
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool whole = y0 + 16 <= NY;            // wave-uniform
        const int lab_r = row_label - (y0 + 4 * g);  // r with column == label, if in 0..3
        float dz[4], fv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool is_lab = lab_r == r;
            const bool masked = ((pad4 >> r) & 1u) | (((hit4 >> r) & 1u) & (is_lab ? 0u : 1u));
            const float val = masked ? MASKV : z[r] - deb4[r];
            const bool yok = whole || (y0 + 4 * g + r < NY);
            if (MODE == CE_FWD || MODE == CE_FUSED) {
                fv[r] = yok ? val : -INFINITY;
                if (yok && is_lab) zlab = val;
            } else {
                const float pr = fexp(val - row_lse);
                dz[r] = (yok && xok) ? (pr - (is_lab ? 1.f : 0.f)) * row_scale : 0.f;
            }
        }
        if (MODE == CE_FWD) {
            const float m4 = fmaxf(fmaxf(fv[0], fv[1]), fmaxf(fv[2], fv[3]));
            const float mn = fmaxf(run_m, m4);
            if (mn > -INFINITY) {
                float add = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) add += fexp(fv[r] - mn);
                run_l = run_l * fexp(run_m - mn) + add;
                run_m = mn;
            }
        } else {
            if (MODE == CE_FUSED) {
                const float m4 = fmaxf(fmaxf(fv[0], fv[1]), fmaxf(fv[2], fv[3]));
                if (__ballot(m4 > run_m)) {              // wave-uniform and rare
                    float mr = fmaxf(m4, __shfl_xor(m4, 16, 64));
                    mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
                    const float mn = fmaxf(run_m, mr);
                    const float f = fexp(run_m - mn);    // run_m starts at -3e38: f = 0 on the first tile
                    run_l *= f;
#pragma unroll
                    for (int et = 0; et < 4; ++et) dacc[et] *= f;
                    run_m = mn;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dz[r] = fexp(fv[r] - run_m);         // structural padding: exp(-inf) = 0
                    run_l += dz[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int et = 0; et < 4; ++et) {
#ifdef CE_ABLATE
                    if (dbg & 2) { dacc[et][0] += yt16[r][et] * dz[r]; continue; }         // ablation (timing only)
#endif
                    dacc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(yt16[r][et], dz[r], dacc[et], 0, 0, 0);
                }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (MODE == CE_FWD) {
        auto comb = [](float& m, float& l, float m2, float l2) {
            const float mn = fmaxf(m, m2);
            const float a = m == -INFINITY ? 0.f : l * expf(m - mn);
            const float c = m2 == -INFINITY ? 0.f : l2 * expf(m2 - mn);
            m = mn;
            l = a + c;
        };
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float m2 = __shfl_xor(run_m, o, 64), l2 = __shfl_xor(run_l, o, 64);
            comb(run_m, run_l, m2, l2);
            zlab += __shfl_xor(zlab, o, 64);
        }
        if (g == 0) {
            sRed[wave][j][0] = run_m;
            sRed[wave][j][1] = run_l;
            sRed[wave][j][2] = zlab;
        }
        __syncthreads();
        if (wave == 0 && g == 0 && xok) {
            float m = sRed[0][j][0], l = sRed[0][j][1], zl = sRed[0][j][2];
            for (int w = 1; w < 4; ++w) {
                comb(m, l, sRed[w][j][0], sRed[w][j][1]);
                zl += sRed[w][j][2];
            }
            const float lse = m + logf(l);
            b.lse[x] = lse;
            b.rowloss[x] = row_valid ? lse - zl : 0.f;
        }
    } else if (MODE == CE_FUSED) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            run_l += __shfl_xor(run_l, o, 64);
            zlab += __shfl_xor(zlab, o, 64);
        }
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) sRed[wave][j][16 * et + 4 * g + r] = dacc[et][r];
        if (g == 0) {
            sRed[wave][j][E] = run_m;
            sRed[wave][j][E + 1] = run_l;
            sRed[wave][j][E + 2] = zlab;
        }
        __syncthreads();
        // per row: the four waves' partial sums brought to the common maximum; d_prec = (sum_y p_y score_y - score_label) / n
        for (int i = tid; i < 16 * E; i += 256) {
            const int rj = i / E, e = i - rj * E;
            const int xr = x0 + rj;
            if (xr >= NX) continue;
            float m = sRed[0][rj][E];
#pragma unroll
            for (int w = 1; w < 4; ++w) m = fmaxf(m, sRed[w][rj][E]);
            float l = 0.f, acc = 0.f, zl = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = fexp(sRed[w][rj][E] - m);
                l += sRed[w][rj][E + 1] * f;
                acc += sRed[w][rj][e] * f;
                zl += sRed[w][rj][E + 2];
            }
            const bool valid = log_mask[xr] != 0.f;
            const int sq = xr / S, lab = sq * S1 + (xr - sq * S) + 1;
            b.dprec[(int64_t)xr * E + e] = valid ? (acc / l - Y[(int64_t)lab * E + e]) / b.nvalid[0] : 0.f;
            if (e == 0) {
                const float lse = m + logf(l);
                b.lse[xr] = lse;
                b.rowloss[xr] = valid ? lse - zl : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) sRed[wave][j][16 * et + 4 * g + r] = dacc[et][r];
        __syncthreads();
        for (int i = tid; i < 16 * E; i += 256) {
            const int rj = i / E, e = i - rj * E;
            if (x0 + rj < NX) dX[(int64_t)(x0 + rj) * E + e] = sRed[0][rj][e] + sRed[1][rj][e] + sRed[2][rj][e] + sRed[3][rj][e];
        }
    }
}

__global__ void ce_scale_kernel(const float* __restrict__ in, float scale, float* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f4 v = ((const f4*)in)[i];
        v *= scale;
        ((f4*)out)[i] = v;
    }
}

// The column-fixed pass (d_score = dZ^T · prec) in the same style: lane (g, j) owns column x0 + j for the whole pass and, per
// tile of 16 prec rows (at most four sequences), tests ITS column against the ids of the tile's sequence slot g — the four
// lanes of a column cover the four slots — and a ballot publishes the 64 answers; per-row label / lse / scale / slot come
// from a small LDS table, one 16-byte read each.
template <int RS1>
__global__ __launch_bounds__(256) void ce_colpass_kernel(const float* __restrict__ prec, const float* __restrict__ score,
                                                         const float* __restrict__ log_mask, CeBufs b, int bs, int S,
                                                         float d_loss, float* __restrict__ dX) {
    __shared__ __attribute__((aligned(16))) float sY[4][16 * YLD];
    __shared__ float sRed[4][16][E + 4];
    __shared__ __attribute__((aligned(16))) int sMeta[4][64];     // ids of the tile's four sequences, 16 per slot
    __shared__ __attribute__((aligned(16))) int sRow[4][64];      // per row: slot | label column | lse | scale
    const int S1 = S + 1;
    const int T = bs * S, M = bs * S1;
    const float* X = score;
    const float* Y = prec;
    const int NX = M, NY = T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * 16;
    const int x = x0 + j;
    const bool xok = x < NX;
    const int xc = xok ? x : NX - 1;

    float xb[16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const f4 t = *(const f4*)(X + (int64_t)xc * E + 16 * g + 4 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) xb[4 * v + e] = t[e];
    }
    const int col_id = b.ids32[xc];
    const bool col_pad = b.colpad[xc] != 0;
    const float col_debias = b.debias[xc];
    const float dscale = d_loss / b.nvalid[0];

    f4 dacc[4];
#pragma unroll
    for (int et = 0; et < 4; ++et) dacc[et] = (f4){0.f, 0.f, 0.f, 0.f};
    float* myY = sY[wave];
    int* myMeta = sMeta[wave];
    int* myRow = sRow[wave];
    const int ntiles = (NY + 15) / 16;
    f4 yreg[4];
    int idreg = -2;
    float lsereg = 0.f, lmreg = 0.f;
    auto prefetch = [&](int yt_) {
        const int y0_ = yt_ * 16;
        const int r = lane >> 2, ch = lane & 3;
        const int yr = y0_ + r;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            yreg[v] = (f4){0.f, 0.f, 0.f, 0.f};
            if (yt_ < ntiles && yr < NY) yreg[v] = *(const f4*)(Y + (int64_t)yr * E + ch * 16 + v * 4);
        }
        if (yt_ < ntiles) {
            const int sq = y0_ / S + g;                 // lane (g, j): id j of the tile's sequence slot g
            idreg = (j < S1 && sq < bs) ? b.ids32[sq * S1 + j] : -2;
            if (lane < 16) {
                const int rw = y0_ + lane < NY ? y0_ + lane : NY - 1;
                lsereg = b.lse[rw];
                lmreg = log_mask[rw];
            }
        }
    };
    prefetch(wave);
    for (int yt = wave; yt < ntiles; yt += 4) {
        const int y0 = yt * 16;
        {
            const int r = lane >> 2, ch = lane & 3;
#pragma unroll
            for (int v = 0; v < 4; ++v) *(f4*)(myY + r * YLD + ch * 16 + v * 4) = yreg[v];
        }
        myMeta[lane] = idreg;
        if (lane < 16) {
            const int seq0 = y0 / S;
            const int rw = y0 + lane < NY ? y0 + lane : NY - 1;
            const int rs = rw / S;
            myRow[lane] = rs - seq0;
            myRow[16 + lane] = rs * S1 + (rw - rs * S) + 1;
            myRow[32 + lane] = __float_as_int(lsereg);
            myRow[48 + lane] = __float_as_int(lmreg != 0.f ? dscale : 0.f);
        }
        prefetch(yt + 4);
        __builtin_amdgcn_wave_barrier();
        bool h = false;
        {
            typedef int i4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < (RS1 + 3) / 4; ++q) {
                const i4 v = *(const i4*)(myMeta + 16 * g + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * q + k < RS1) h |= v[k] == col_id;
            }
        }
        const unsigned long long hm = __ballot(h);      // bit 16*slot + j
        typedef int i4 __attribute__((ext_vector_type(4)));
        const i4 slot4 = *(const i4*)(myRow + 4 * g);
        const i4 lab4 = *(const i4*)(myRow + 16 + 4 * g);
        const f4 lse4 = *(const f4*)(myRow + 32 + 4 * g);
        const f4 sc4 = *(const f4*)(myRow + 48 + 4 * g);

        f4 z = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; v += 2) {
            const f4 ya = *(const f4*)(myY + j * YLD + 16 * g + 4 * v);
            const f4 yb = *(const f4*)(myY + j * YLD + 16 * g + 4 * v + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[e], xb[4 * v + e], z, 0, 0, 0);
                z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[e], xb[4 * v + 4 + e], z1, 0, 0, 0);
            }
        }
        z += z1;
        const bool whole = y0 + 16 <= NY;
        float dz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool is_lab = lab4[r] == xc;
            const bool hit = (hm >> (16 * slot4[r] + j)) & 1ull;
            const bool masked = col_pad | (hit & !is_lab);
            const float val = masked ? MASKV : z[r] - col_debias;
            const bool yok = whole || (y0 + 4 * g + r < NY);
            const float pr = fexp(val - lse4[r]);
            dz[r] = (yok && xok) ? (pr - (is_lab ? 1.f : 0.f)) * sc4[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int et = 0; et < 4; ++et) {
                const float ya = myY[(4 * g + r) * YLD + 16 * et + j];
                dacc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya, dz[r], dacc[et], 0, 0, 0);
            }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) sRed[wave][j][16 * et + 4 * g + r] = dacc[et][r];
    __syncthreads();
    for (int i = tid; i < 16 * E; i += 256) {
        const int rj = i / E, e = i - rj * E;
        if (x0 + rj < NX) dX[(int64_t)(x0 + rj) * E + e] = sRed[0][rj][e] + sRed[1][rj][e] + sRed[2][rj][e] + sRed[3][rj][e];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same two passes on the 16-bit matrix cores with SPLIT operands (the idea of split.hip, inside the loss).
// At the Cached batch size (bs = 1024: 10,240 x 11,264 logits) the f32-input passes above take 415 + 425 us per step against
// 2 x 220 us of v_mfma_f32_16x16x4_f32 time: 59 GFLOP on a 157 TF pipe.  Here prec and score are split once per call into
// fp16 planes, x s = hi + lo (s a power of two from the tensor's amax), and
//     logits      = (Xh Yh^T + Xh Yl^T + Xl Yh^T) / (sx sy)            3 x v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block
//     dX^T       += (Yh^T Qh + Yh^T Ql + Yl^T Qh) / (sy 2^10)          Q = the exp / probability registers, split the same way
// — a 16 x 16 tile pair costs 12 + 12 sixteen-cycle MFMAs instead of 32 thirty-two-cycle ones, the passes become VALU-bound
// (mask, exp, split: ~25 instructions per logit).  Accuracy as split.hip: 22 mantissa bits of every element within 2^-16 of
// the tensor's amax, the dropped lo x lo term 2^-22 relative.
// Shape: a workgroup = 64 x NSUB X rows (a wave owns NSUB blocks of 16) x one of `ysplits` ranges of Y; every step stages 32 Y
// rows for all four waves — row-major (A operand of the logits) and transposed (A operand of the second product), hi and lo,
// four contiguous 4 KB blocks of the images ce16_split_kernel wrote — double-buffered, one barrier per step.  A wave keeps
// its rows' online-softmax state; the per-range partial results meet in a small combine kernel (fixed order: bit-reproducible).
// The 8 probabilities a lane holds after the two logit tiles of a step ARE the B operand of the second product (K-slot t of
// lane group g <-> Y row 4g + t | 16 + 4g + (t - 4) of the step: the transposed images are stored in that order).
constexpr int C16_YLD = 72;            // halves per LDS row of the row-major tile (128 B + 16: conflict-free ds_read_b128)
constexpr int C16_TLD = 40;            // halves per LDS row of the transposed tile (64 B + 16)
constexpr float C16_QS = 1024.f;       // the probability registers are split at this scale (they are <= 1 in magnitude)
constexpr int C16_MAX_YS = 16;

struct Ce16Bufs {
    uint32_t* amax;            // [2]: bit patterns of max|prec|, max|score|
    _Float16* img[2][2];       // [prec | score][hi | lo]: row-major [N padded to 32][64]
    _Float16* imgT[2][2];      // the same rows transposed per group of 32: [group][64][32 slots]
    float* part_row;           // [ysplits][T][68]: sum_y exp(z - m) score_y (64) | m | l | z_label | -
    float* part_col;           // [ysplits][M][64]
};

void carve16(WsCarver& c, Ce16Bufs& b, int64_t bs, int S) {
    const size_t M = (size_t)bs * (S + 1), T = (size_t)bs * S;
    const size_t n[2] = {align_up(T, 32), align_up(M, 32)};
    b.amax = c.take<uint32_t>(4);
    for (int t = 0; t < 2; ++t)
        for (int pl = 0; pl < 2; ++pl) {
            b.img[t][pl] = c.take<_Float16>(n[t] * E);
            b.imgT[t][pl] = c.take<_Float16>(n[t] * E);
        }
    b.part_row = c.take<float>((size_t)C16_MAX_YS * T * 68);
    b.part_col = c.take<float>((size_t)C16_MAX_YS * M * E);
}

__device__ __forceinline__ float c16_scale_of(uint32_t amax_bits) {       // as split.hip: the largest element lands in [2^13, 2^14)
    const int e = (int)(amax_bits >> 23) & 0xff;
    if (e == 0 || e == 0xff) return 1.0f;
    return __uint_as_float((uint32_t)(267 - e) << 23);
}

__global__ __launch_bounds__(256) void ce16_amax_kernel(const float* __restrict__ prec, int64_t n0, const float* __restrict__ score, int64_t n1,
                                                        uint32_t* amax) {
    __shared__ float red[4];
    const float* x = blockIdx.y ? score : prec;
    const int64_t n4 = (blockIdx.y ? n1 : n0) / 4;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f4 v = ((const f4*)x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(amax + blockIdx.y, __float_as_uint(m));
    }
}

// slot of Y row r (0..31) of a step inside the transposed image: lane group g = slot >> 3 holds rows 4g..4g+3 | 16+4g..16+4g+3
__device__ __forceinline__ int c16_slot(int r) { return r < 16 ? 8 * (r >> 2) + (r & 3) : 8 * ((r - 16) >> 2) + 4 + (r & 3); }

// one workgroup = 32 rows of one tensor (blockIdx.y): the four images of that group
__global__ __launch_bounds__(256) void ce16_split_kernel(const float* __restrict__ prec, int64_t T, const float* __restrict__ score, int64_t M, Ce16Bufs c) {
    __shared__ _Float16 sT[2][E][32 + 2];
    const int t = blockIdx.y;
    const float* x = t ? score : prec;
    const int64_t N = t ? M : T, groups = (N + 31) / 32;
    if ((int64_t)blockIdx.x >= groups) return;
    const float s = c16_scale_of(c.amax[t]);
    const int tid = threadIdx.x, r = tid >> 3, c8 = (tid & 7) * 8;
    const int64_t row = (int64_t)blockIdx.x * 32 + r;
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)0.f; lo[e] = (_Float16)0.f; }
    if (row < N) {
        const f4 v0 = *(const f4*)(x + row * E + c8), v1 = *(const f4*)(x + row * E + c8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = v0[e] * s, bq = v1[e] * s;
            hi[e] = (_Float16)a; lo[e] = (_Float16)(a - (float)hi[e]);
            hi[4 + e] = (_Float16)bq; lo[4 + e] = (_Float16)(bq - (float)hi[4 + e]);
        }
    }
    *(h8*)(c.img[t][0] + row * E + c8) = hi;
    *(h8*)(c.img[t][1] + row * E + c8) = lo;
    const int slot = c16_slot(r);
#pragma unroll
    for (int e = 0; e < 8; ++e) { sT[0][c8 + e][slot] = hi[e]; sT[1][c8 + e][slot] = lo[e]; }
    __syncthreads();
    const int er = tid >> 2, p8 = (tid & 3) * 8;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sT[pl][er][p8 + e];
        *(h8*)(c.imgT[t][pl] + ((int64_t)blockIdx.x * E + er) * 32 + p8) = v;
    }
}

// the staged tiles of one step, shared by the four waves
struct C16Tiles {
    _Float16 yh[32 * C16_YLD], yl[32 * C16_YLD], th[E * C16_TLD], tl[E * C16_TLD];
};
struct C16Stage {      // one thread's share of a step in flight: a 16-byte piece of each of the four tiles
    h8 yh, yl, th, tl;
};
__device__ __forceinline__ void c16_load(C16Stage& st, const Ce16Bufs& c, int ty, int64_t step, int tid) {
    const int64_t o = step * (32 * E) + tid * 8;
    st.yh = *(const h8*)(c.img[ty][0] + o);
    st.yl = *(const h8*)(c.img[ty][1] + o);
    st.th = *(const h8*)(c.imgT[ty][0] + o);
    st.tl = *(const h8*)(c.imgT[ty][1] + o);
}
__device__ __forceinline__ void c16_store(C16Tiles& t, const C16Stage& st, int tid) {
    const int yo = (tid >> 3) * C16_YLD + (tid & 7) * 8, to = (tid >> 2) * C16_TLD + (tid & 3) * 8;
    *(h8*)(t.yh + yo) = st.yh;
    *(h8*)(t.yl + yo) = st.yl;
    *(h8*)(t.th + to) = st.th;
    *(h8*)(t.tl + to) = st.tl;
}
// logits^T of one half (16 Y rows) of the step against the 16 X rows whose fragments the lane holds: lane (j, g) -> X row j, Y rows 4g..4g+3
__device__ __forceinline__ f4 c16_logits(const C16Tiles& t, int half, int j, int g, const h8 (&xh)[2], const h8 (&xl)[2]) {
    f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int o = (16 * half + j) * C16_YLD + 32 * cc + 8 * g;
        const h8 ah = *(const h8*)(t.yh + o), al = *(const h8*)(t.yl + o);
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[cc], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[cc], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[cc], z, 0, 0, 0);
    }
    return z;
}
// dacc[et] += Y^T (16 features of block et x the step's 32 rows) · Q (32 rows x the lane's X row), Q split at C16_QS
__device__ __forceinline__ void c16_second(const C16Tiles& t, int j, int g, const float (&q)[8], f4 (&dacc)[4]) {
    h8 bh, bl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = q[e] * C16_QS;
        bh[e] = (_Float16)v;
        bl[e] = (_Float16)(v - (float)bh[e]);
    }
    // fragments one feature block ahead, fenced: left alone the compiler requests all eight up front (32 registers that cost the pass a wave of occupancy)
    const int o0 = j * C16_TLD + 8 * g;
    h8 ah = *(const h8*)(t.th + o0), al = *(const h8*)(t.tl + o0);
#pragma unroll
    for (int et = 0; et < 4; ++et) {
        h8 nh = ah, nl = al;
        if (et < 3) {
            nh = *(const h8*)(t.th + o0 + 16 * (et + 1) * C16_TLD);
            nl = *(const h8*)(t.tl + o0 + 16 * (et + 1) * C16_TLD);
        }
        __builtin_amdgcn_sched_barrier(0);
        dacc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, dacc[et], 0, 0, 0);
        dacc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, dacc[et], 0, 0, 0);
        dacc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, dacc[et], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ah = nh; al = nl;
    }
}

// Fused forward + d_prec row pass (the online softmax of ce_rowpass_kernel<CE_FUSED>): X = prec, Y = score.  grid (x blocks, ysplits)
template <int RS1, int NSUB>
__global__ __launch_bounds__(256, NSUB == 1 ? 4 : 2) void ce16_rowpass_kernel(const float* __restrict__ log_mask, CeBufs b, Ce16Bufs c, int bs, int S, int steps_per) {
    __shared__ __attribute__((aligned(16))) C16Tiles tiles[2];
    __shared__ __attribute__((aligned(16))) int sMeta[2][96];          // per step: ids[32] | padding[32] | debias[32]
    __shared__ __attribute__((aligned(16))) int sSid[4][NSUB][64];     // per wave and X block: the ids of its four sequence slots, 16 per slot
    const int S1 = S + 1;
    const int T = bs * S, M = bs * S1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float inv_z = 1.0f / (c16_scale_of(c.amax[0]) * c16_scale_of(c.amax[1]));
    const float inv_d = 1.0f / (c16_scale_of(c.amax[1]) * C16_QS);
    h8 xh[NSUB][2], xl[NSUB][2];
    int row_label[NSUB], row_shift[NSUB];
    bool xok[NSUB];
    int xrow[NSUB];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        const int x0 = (blockIdx.x * NSUB + u) * 64 + wave * 16;
        const int x = x0 + j;
        xok[u] = x < T;
        xrow[u] = x;
        const int xc = xok[u] ? x : T - 1;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            xh[u][cc] = *(const h8*)(c.img[0][0] + (int64_t)xc * E + 32 * cc + 8 * g);
            xl[u][cc] = *(const h8*)(c.img[0][1] + (int64_t)xc * E + 32 * cc + 8 * g);
        }
        const int row_seq = xc / S, seq0 = x0 / S;
        row_label[u] = row_seq * S1 + (xc - row_seq * S) + 1;
        row_shift[u] = xok[u] ? (row_seq - seq0) * 16 + 4 * g : 4 * g;
        const int sq = seq0 + g;
        sSid[wave][u][lane] = (j < S1 && sq < bs) ? b.ids32[sq * S1 + j] : -2;      // (read back by the same wave only; the first barrier below orders it)
    }
    float run_m[NSUB], run_l[NSUB], zlab[NSUB];
    f4 dacc[NSUB][4];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        run_m[u] = -3.0e38f; run_l[u] = 0.f; zlab[u] = 0.f;
#pragma unroll
        for (int et = 0; et < 4; ++et) dacc[u][et] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    const int steps_total = (M + 31) / 32;
    const int st0 = blockIdx.y * steps_per;
    int st1 = st0 + steps_per;
    if (st1 > steps_total) st1 = steps_total;
    C16Stage stg;
    int mreg[3] = {0, 0, 0};
    auto fetch = [&](int st) {
        st = st < st1 ? st : st1 - 1;                                   // past the end: the last step again (never used)
        c16_load(stg, c, 1, st, tid);
        if (tid < 32) {
            const int y = st * 32 + tid < M ? st * 32 + tid : M - 1;
            mreg[0] = b.ids32[y];
            mreg[1] = b.colpad[y];
            mreg[2] = __float_as_int(b.debias[y]);
        }
    };
    auto put = [&](int buf) {
        c16_store(tiles[buf], stg, tid);
        if (tid < 32) { sMeta[buf][tid] = mreg[0]; sMeta[buf][32 + tid] = mreg[1]; sMeta[buf][64 + tid] = mreg[2]; }
    };
    if (st0 < st1) {
        fetch(st0);
        put(0);
        __syncthreads();
    }
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        fetch(st + 1);
        const C16Tiles& tl = tiles[buf];
        const int* meta = sMeta[buf];
        const int y0 = st * 32;
        const bool whole = y0 + 32 <= M;                                // block-uniform
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            float fv[8];
            bool lab[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f4 z = c16_logits(tl, h, j, g, xh[u], xl[u]);
                // once per wave and half: (sequence slot g, column j) hit bits and the 16 column-padding bits
                const int idc = meta[16 * h + j];
                bool hit = false;
                {
                    typedef int i4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int qq = 0; qq < (RS1 + 3) / 4; ++qq) {
                        const i4 v = *(const i4*)(sSid[wave][u] + 16 * g + 4 * qq);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (4 * qq + k < RS1) hit |= v[k] == idc;
                    }
                }
                const unsigned long long hm = __ballot(hit);
                const unsigned pm = (unsigned)__ballot(meta[32 + 16 * h + j] != 0);
                const unsigned hit4 = (unsigned)(hm >> row_shift[u]) & 0xfu;
                const unsigned pad4 = (pm >> (4 * g)) & 0xfu;
                const f4 deb4 = *(const f4*)(meta + 64 + 16 * h + 4 * g);
                const int lab_r = row_label[u] - (y0 + 16 * h + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool is_lab = lab_r == r;
                    const bool masked = ((pad4 >> r) & 1u) | (((hit4 >> r) & 1u) & (is_lab ? 0u : 1u));
                    const float val = masked ? MASKV : z[r] * inv_z - deb4[r];
                    const bool yok = whole || (y0 + 16 * h + 4 * g + r < M);
                    fv[4 * h + r] = yok ? val : -INFINITY;
                    lab[4 * h + r] = yok && is_lab;
                }
            }
            float m8 = fv[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) m8 = fmaxf(m8, fv[e]);
            if (__ballot(m8 > run_m[u])) {                               // wave-uniform and rare after the first steps
                float mr = fmaxf(m8, __shfl_xor(m8, 16, 64));
                mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
                const float mn = fmaxf(run_m[u], mr);
                const float f = fexp(run_m[u] - mn);                     // run_m starts at -3e38: f = 0 on the first step
                run_l[u] *= f;
#pragma unroll
                for (int et = 0; et < 4; ++et) dacc[u][et] *= f;
                run_m[u] = mn;
            }
            float q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                q[e] = fexp(fv[e] - run_m[u]);                           // structural padding: exp(-inf) = 0
                run_l[u] += q[e];
                if (lab[e]) zlab[u] = fv[e];
            }
            c16_second(tl, j, g, q, dacc[u]);
        }
        put(buf ^ 1);
        __syncthreads();
    }
    // per-range partial results of the wave's rows (the four lanes of a row share run_m; their sums add up)
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            run_l[u] += __shfl_xor(run_l[u], o, 64);
            zlab[u] += __shfl_xor(zlab[u], o, 64);
        }
        if (!xok[u]) continue;
        float* out = c.part_row + ((int64_t)blockIdx.y * T + xrow[u]) * 68;
#pragma unroll
        for (int et = 0; et < 4; ++et) *(f4*)(out + 16 * et + 4 * g) = dacc[u][et] * inv_d;
        if (g == 0) *(f4*)(out + E) = (f4){run_m[u], run_l[u], zlab[u], 0.f};
    }
}

// lse, row loss and d_prec (for d_loss = 1) from the per-range partial results: the ranges brought to the common maximum, fixed order
__global__ __launch_bounds__(256) void ce16_row_combine_kernel(const float* __restrict__ score, const float* __restrict__ log_mask, CeBufs b, Ce16Bufs c,
                                                               int bs, int S, int ysplits) {
    const int S1 = S + 1, T = bs * S;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int xr = i >> 6, e = i & 63;
    if (xr >= T) return;
    float m = -3.0e38f;
    for (int y = 0; y < ysplits; ++y) m = fmaxf(m, c.part_row[((int64_t)y * T + xr) * 68 + E]);
    float l = 0.f, acc = 0.f, zl = 0.f;
    for (int y = 0; y < ysplits; ++y) {
        const float* p = c.part_row + ((int64_t)y * T + xr) * 68;
        const float f = fexp(p[E] - m);
        l += p[E + 1] * f;
        acc += p[e] * f;
        zl += p[E + 2];
    }
    const bool valid = log_mask[xr] != 0.f;
    const int sq = xr / S, lab = sq * S1 + (xr - sq * S) + 1;
    b.dprec[(int64_t)xr * E + e] = valid ? (acc / l - score[(int64_t)lab * E + e]) / b.nvalid[0] : 0.f;
    if (e == 0) {
        const float lse = m + logf(l);
        b.lse[xr] = lse;
        b.rowloss[xr] = valid ? lse - zl : 0.f;
    }
}

// The column-fixed pass (d_score = dZ^T · prec): X = score (a lane owns one column), Y = prec.  grid (x blocks, ysplits)
template <int RS1, int NSUB>
__global__ __launch_bounds__(256, NSUB == 1 ? 4 : 2) void ce16_colpass_kernel(const float* __restrict__ log_mask, CeBufs b, Ce16Bufs c, int bs, int S, int steps_per) {
    __shared__ __attribute__((aligned(16))) C16Tiles tiles[2];
    __shared__ __attribute__((aligned(16))) int sIds[2][128];          // per half: ids of its four sequence slots, 16 per slot
    __shared__ __attribute__((aligned(16))) int sRow[2][4][32];        // per Y row: slot within its half | label column | lse | valid
    const int S1 = S + 1;
    const int T = bs * S, M = bs * S1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float inv_z = 1.0f / (c16_scale_of(c.amax[0]) * c16_scale_of(c.amax[1]));
    const float inv_d = 1.0f / (c16_scale_of(c.amax[0]) * C16_QS);
    h8 xh[NSUB][2], xl[NSUB][2];
    int col_id[NSUB], xcol[NSUB];
    bool col_pad[NSUB], xok[NSUB];
    float col_debias[NSUB];
    f4 dacc[NSUB][4];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        const int x = (blockIdx.x * NSUB + u) * 64 + wave * 16 + j;
        xok[u] = x < M;
        xcol[u] = x;
        const int xc = xok[u] ? x : M - 1;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            xh[u][cc] = *(const h8*)(c.img[1][0] + (int64_t)xc * E + 32 * cc + 8 * g);
            xl[u][cc] = *(const h8*)(c.img[1][1] + (int64_t)xc * E + 32 * cc + 8 * g);
        }
        col_id[u] = b.ids32[xc];
        col_pad[u] = b.colpad[xc] != 0;
        col_debias[u] = b.debias[xc];
#pragma unroll
        for (int et = 0; et < 4; ++et) dacc[u][et] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    const int steps_total = (T + 31) / 32;
    const int st0 = blockIdx.y * steps_per;
    int st1 = st0 + steps_per;
    if (st1 > steps_total) st1 = steps_total;
    C16Stage stg;
    int mreg[4] = {0, 0, 0, 0};
    auto fetch = [&](int st) {
        st = st < st1 ? st : st1 - 1;
        c16_load(stg, c, 0, st, tid);
        if (tid < 128) {                       // thread (half, slot, id index): id of the half's sequence slot
            const int h = tid >> 6, sl = (tid >> 4) & 3, k = tid & 15;
            const int sq = (st * 32 + 16 * h) / S + sl;
            mreg[0] = (k < S1 && sq < bs) ? b.ids32[sq * S1 + k] : -2;
        } else if (tid < 160) {                // one Y row each
            const int r = tid - 128, y = st * 32 + r;
            const int yc = y < T ? y : T - 1;
            const int rs = yc / S;
            const int sl = rs - (st * 32 + 16 * (r >> 4)) / S;          // 0..3 for real rows (S >= 5); rows past T never count
            mreg[0] = sl < 0 ? 0 : (sl > 3 ? 3 : sl);
            mreg[1] = rs * S1 + (yc - rs * S) + 1;
            mreg[2] = __float_as_int(b.lse[yc]);
            mreg[3] = __float_as_int((y < T && log_mask[yc] != 0.f) ? 1.f : 0.f);
        }
    };
    auto put = [&](int buf) {
        c16_store(tiles[buf], stg, tid);
        if (tid < 128) sIds[buf][tid] = mreg[0];
        else if (tid < 160) {
            const int r = tid - 128;
            sRow[buf][0][r] = mreg[0]; sRow[buf][1][r] = mreg[1]; sRow[buf][2][r] = mreg[2]; sRow[buf][3][r] = mreg[3];
        }
    };
    if (st0 < st1) {
        fetch(st0);
        put(0);
        __syncthreads();
    }
    typedef int i4 __attribute__((ext_vector_type(4)));
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        fetch(st + 1);
        const C16Tiles& tl = tiles[buf];
        const int y0 = st * 32;
        const bool whole = y0 + 32 <= T;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            float q[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f4 z = c16_logits(tl, h, j, g, xh[u], xl[u]);
                bool hit = false;
#pragma unroll
                for (int qq = 0; qq < (RS1 + 3) / 4; ++qq) {
                    const i4 v = *(const i4*)(sIds[buf] + 64 * h + 16 * g + 4 * qq);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (4 * qq + k < RS1) hit |= v[k] == col_id[u];
                }
                const unsigned long long hm = __ballot(hit);            // bit 16 * slot + j
                const i4 slot4 = *(const i4*)(sRow[buf][0] + 16 * h + 4 * g);
                const i4 lab4 = *(const i4*)(sRow[buf][1] + 16 * h + 4 * g);
                const f4 lse4 = *(const f4*)(sRow[buf][2] + 16 * h + 4 * g);
                const f4 ok4 = *(const f4*)(sRow[buf][3] + 16 * h + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool is_lab = lab4[r] == xcol[u];
                    const bool hb = (hm >> (16 * slot4[r] + j)) & 1ull;
                    const bool masked = col_pad[u] | (hb & !is_lab);
                    const float val = masked ? MASKV : z[r] * inv_z - col_debias[u];
                    const bool yok = whole || (y0 + 16 * h + 4 * g + r < T);
                    const float pr = fexp(val - lse4[r]);
                    q[4 * h + r] = (yok && xok[u]) ? (pr - (is_lab ? 1.f : 0.f)) * ok4[r] : 0.f;
                }
            }
            c16_second(tl, j, g, q, dacc[u]);
        }
        put(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        if (!xok[u]) continue;
        float* out = c.part_col + ((int64_t)blockIdx.y * M + xcol[u]) * E;
#pragma unroll
        for (int et = 0; et < 4; ++et) *(f4*)(out + 16 * et + 4 * g) = dacc[u][et] * inv_d;
    }
}

__global__ __launch_bounds__(256) void ce16_col_combine_kernel(CeBufs b, Ce16Bufs c, int64_t n4, int ysplits, float d_loss, float* __restrict__ d_score) {
    const float sc = d_loss / b.nvalid[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 v = ((const f4*)c.part_col)[i];
        for (int y = 1; y < ysplits; ++y) v += ((const f4*)c.part_col)[(int64_t)y * n4 + i];
        ((f4*)d_score)[i] = v * sc;
    }
}

// 1 (default): one fused FWD + DPREC row pass (online softmax) and the cooperative column pass — on the 16-bit matrix cores with split
// operands (ce16_*) from C16_MIN_LOGITS logits on, on the f32 cores below; 2: separate FWD and DPREC row passes (round-2a form); 0: the
// generic kernel everywhere; 3: ce16_* at every size; 4: the f32 fused passes at every size (test knobs)
int g_ce_fast = 1;
constexpr int g_ce_dbg = 0;      // (the ablation bits of the fused row pass: compile-time zero since round 5)
// bs = 128 (1,280 x 1,408 logits): 29 us per f32 pass, launch-sized — the split route's four extra small launches would cost more than it gains
constexpr int64_t C16_MIN_LOGITS = (int64_t)1 << 24;
bool rowpass_ok(int64_t bs, int S) { return g_ce_fast && S >= 5 && S + 1 <= MAXS1 && bs * (int64_t)(S + 1) < (1ll << 31); }
bool fused_ok(int64_t bs, int S) { return rowpass_ok(bs, S) && (g_ce_fast == 1 || g_ce_fast == 3 || g_ce_fast == 4); }
bool ce16_ok(int64_t bs, int S) {
    return rowpass_ok(bs, S) && (g_ce_fast == 3 || (g_ce_fast == 1 && bs * S * bs * (int64_t)(S + 1) >= C16_MIN_LOGITS));
}
// ranges of Y per X block: about seven workgroups per CU (four are resident at a time) and >= 8 steps per range.  Same-box sweep at bs = 1024
// (tools/ce_sweep.py, forward + backward of the loss, both passes with the same count): 4 ranges 402 - 407 us, 6 390, 8 365, 10 355 - 357,
// 11 347 - 348, 12 357, 16 353; the f32 passes 852 - 891.
int g_ce16_ys = 0;               // > 0: this many Y ranges (sweeps)
int ce16_ysplits(int64_t nx, int64_t ny, int nsub) {
    if (g_ce16_ys > 0) return g_ce16_ys > C16_MAX_YS ? C16_MAX_YS : g_ce16_ys;
    const int64_t xb = ceil_div(nx, (int64_t)64 * nsub), steps = ceil_div(ny, (int64_t)32), cus = iisan_cu_count();
    int64_t ys = (7 * cus + xb / 2) / xb;
    if (ys > C16_MAX_YS) ys = C16_MAX_YS;
    while (ys > 1 && steps / ys < 8) --ys;
    return ys < 1 ? 1 : (int)ys;
}
template <int NSUB>
int launch_ce16_row(const float* log_mask, const CeBufs& b, const Ce16Bufs& c, int64_t bs, int S, hipStream_t s, int* ys_out) {
    const int64_t T = bs * S, M = bs * (S + 1);
    const int ys = ce16_ysplits(T, M, NSUB);
    const int steps_per = (int)ceil_div(ceil_div(M, (int64_t)32), (int64_t)ys);
    const dim3 grid((unsigned)ceil_div(T, (int64_t)64 * NSUB), (unsigned)ys);
    if (S + 1 <= 11) hipLaunchKernelGGL((ce16_rowpass_kernel<11, NSUB>), grid, dim3(256), 0, s, log_mask, b, c, (int)bs, S, steps_per);
    else hipLaunchKernelGGL((ce16_rowpass_kernel<MAXS1, NSUB>), grid, dim3(256), 0, s, log_mask, b, c, (int)bs, S, steps_per);
    IISAN_LAUNCH_OK();
    *ys_out = ys;
    return IISAN_OK;
}
template <int NSUB>
int launch_ce16_col(const float* log_mask, const CeBufs& b, const Ce16Bufs& c, int64_t bs, int S, hipStream_t s, int* ys_out) {
    const int64_t T = bs * S, M = bs * (S + 1);
    const int ys = ce16_ysplits(M, T, NSUB);
    const int steps_per = (int)ceil_div(ceil_div(T, (int64_t)32), (int64_t)ys);
    const dim3 grid((unsigned)ceil_div(M, (int64_t)64 * NSUB), (unsigned)ys);
    if (S + 1 <= 11) hipLaunchKernelGGL((ce16_colpass_kernel<11, NSUB>), grid, dim3(256), 0, s, log_mask, b, c, (int)bs, S, steps_per);
    else hipLaunchKernelGGL((ce16_colpass_kernel<MAXS1, NSUB>), grid, dim3(256), 0, s, log_mask, b, c, (int)bs, S, steps_per);
    IISAN_LAUNCH_OK();
    *ys_out = ys;
    return IISAN_OK;
}
int g_ce16_nsub = 1;             // X blocks of 16 rows per wave (1 or 2)

int check(int64_t bs, int S, int Ein) {
    IISAN_CHECK_SHAPE(bs > 0 && S >= 1 && S <= 63, "inbatch_ce: bs %lld / S %d unsupported", (long long)bs, S);
    IISAN_CHECK_SHAPE(Ein == E, "inbatch_ce: embedding_dim must be %d (got %d)", E, Ein);
    return IISAN_OK;
}

}  // namespace

extern "C" size_t iisan_inbatch_ce_ws_bytes(int64_t bs, int32_t S) {
    WsCarver c(nullptr, 0);
    CeBufs b;
    Ce16Bufs b16;
    carve(c, b, bs, S);
    carve16(c, b16, bs, S);
    return c.off;
}

// the forward call's token: a tag, the call shape and the route it took (1 = separate / generic passes, 2 = fused f32 row pass, 3 = split-operand passes)
static uint64_t ce_token(int64_t bs, int32_t S, int route) {
    return 0xCE00000000000000ull | ((uint64_t)(bs & 0xFFFFFFFFll) << 16) | ((uint64_t)(S & 0xFF) << 8) | (uint64_t)(route + 1);
}
IISAN_DEV_KNOB(ce_fast, g_ce_fast);
IISAN_DEV_KNOB(ce16_nsub, g_ce16_nsub);
IISAN_DEV_KNOB(ce16_ys, g_ce16_ys);
static int64_t g_cnt_ce16 = 0;            // forward calls on the split-operand route (route counter, common.h)
IISAN_DEV_COUNTER(ce16, g_cnt_ce16);

extern "C" int iisan_inbatch_ce_fwd(const int64_t* ids, const float* score, const float* prec, const float* log_mask,
                                    const float* pop_prob, int64_t n_pop, int64_t bs, int32_t S, int32_t Ein, float* loss,
                                    void* ws, size_t ws_bytes, uint64_t* fwd_token, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_TRY(check(bs, S, Ein));
    IISAN_CHECK_SHAPE(n_pop > 0, "inbatch_ce_fwd: empty pop_prob table");
    WsCarver c(ws, ws_bytes);
    CeBufs b;
    Ce16Bufs b16;
    carve(c, b, bs, S);
    carve16(c, b16, bs, S);
    if (c.overflow || !ws) {
        iisan_set_error("inbatch_ce_fwd: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    const int64_t T = bs * S, M = bs * (S + 1);
    hipLaunchKernelGGL(ce_prep_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, s, ids, log_mask, pop_prob, n_pop, b, bs, S);
    IISAN_LAUNCH_OK();
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, s, log_mask, T, b.nvalid);
    IISAN_LAUNCH_OK();
    if (ce16_ok(bs, S)) {
        // split-operand route: amax -> images -> fused row pass per Y range -> combine (lse, row loss, d_prec for d_loss = 1 in the workspace)
        IISAN_CHECK_SHAPE(fwd_token != nullptr, "inbatch_ce_fwd: fwd_token must not be null");
        *fwd_token = ce_token(bs, S, 2);
        ++g_cnt_ce16;
        IISAN_HIP_OK(hipMemsetAsync(b16.amax, 0, 16, s));
        hipLaunchKernelGGL(ce16_amax_kernel, dim3(64, 2), dim3(256), 0, s, prec, T * E, score, M * E, b16.amax);
        IISAN_LAUNCH_OK();
        hipLaunchKernelGGL(ce16_split_kernel, dim3((unsigned)ceil_div(M, 32), 2), dim3(256), 0, s, prec, T, score, M, b16);
        IISAN_LAUNCH_OK();
        int ys = 1;
        if (g_ce16_nsub == 2) IISAN_TRY(launch_ce16_row<2>(log_mask, b, b16, bs, S, s, &ys));
        else IISAN_TRY(launch_ce16_row<1>(log_mask, b, b16, bs, S, s, &ys));
        hipLaunchKernelGGL(ce16_row_combine_kernel, dim3((unsigned)ceil_div(T * E, 256)), dim3(256), 0, s, score, log_mask, b, b16, (int)bs, S, ys);
        IISAN_LAUNCH_OK();
        hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, b.rowloss, T, b.nvalid, loss);
        IISAN_LAUNCH_OK();
        return IISAN_OK;
    }
    // fast path: the forward pass leaves d_prec (for d_loss = 1) in the workspace, the backward call only scales it — the
    // route goes back to the caller as a token, so that the backward call follows what THIS call did, not the knob's later
    // value, and the library keeps no per-call state
    IISAN_CHECK_SHAPE(fwd_token != nullptr, "inbatch_ce_fwd: fwd_token must not be null");
    *fwd_token = ce_token(bs, S, fused_ok(bs, S) ? 1 : 0);
    if (fused_ok(bs, S) && S + 1 <= 11)
        hipLaunchKernelGGL((ce_rowpass_kernel<CE_FUSED, 11>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score, log_mask, b, (int)bs, S, 0.f, (float*)nullptr, g_ce_dbg);
    else if (fused_ok(bs, S))
        hipLaunchKernelGGL((ce_rowpass_kernel<CE_FUSED, MAXS1>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score, log_mask, b, (int)bs, S, 0.f, (float*)nullptr);
    else if (rowpass_ok(bs, S) && S + 1 <= 11) hipLaunchKernelGGL((ce_rowpass_kernel<CE_FWD, 11>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score,
                       log_mask, b, (int)bs, S, 0.f, (float*)nullptr);
    else if (rowpass_ok(bs, S)) hipLaunchKernelGGL((ce_rowpass_kernel<CE_FWD, MAXS1>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score,
                       log_mask, b, (int)bs, S, 0.f, (float*)nullptr);
    else if (S + 1 <= 11) hipLaunchKernelGGL((ce_pass_kernel<CE_FWD, 11>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score, log_mask, b, bs, S, 0.f,
                       (float*)nullptr);
    else hipLaunchKernelGGL((ce_pass_kernel<CE_FWD, MAXS1>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score, log_mask, b, bs, S, 0.f,
                       (float*)nullptr);
    IISAN_LAUNCH_OK();
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, b.rowloss, T, b.nvalid, loss);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}

extern "C" int iisan_inbatch_ce_bwd(const int64_t* ids, const float* score, const float* prec, const float* log_mask,
                                    const float* pop_prob, int64_t bs, int32_t S, int32_t Ein, float d_loss, float* d_score,
                                    float* d_prec, void* ws, size_t ws_bytes, uint64_t fwd_token, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    IISAN_TRY(check(bs, S, Ein));
    WsCarver c(ws, ws_bytes);
    CeBufs b;
    Ce16Bufs b16;
    carve(c, b, bs, S);       // ids32/debias/colpad/lse/nvalid were filled by the forward call on the same workspace
    carve16(c, b16, bs, S);   // ... and, on the split-operand route, the operand images and their amax
    if (c.overflow || !ws) {
        iisan_set_error("inbatch_ce_bwd: workspace too small (%zu < %zu)", ws_bytes, c.off);
        return IISAN_EWORKSPACE;
    }
    const int64_t T = bs * S, M = bs * (S + 1);
    const bool split16 = fwd_token == ce_token(bs, S, 2);
    const bool fused = split16 || fwd_token == ce_token(bs, S, 1);
    if (split16) {            // whatever the dev switch says by now: the images this route needs are in the workspace
        hipLaunchKernelGGL(ce_scale_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(T * E / 4, 256), 1024)), dim3(256), 0, s, b.dprec, d_loss, d_prec, T * E / 4);
        IISAN_LAUNCH_OK();
        int ys = 1;
        if (g_ce16_nsub == 2) IISAN_TRY(launch_ce16_col<2>(log_mask, b, b16, bs, S, s, &ys));
        else IISAN_TRY(launch_ce16_col<1>(log_mask, b, b16, bs, S, s, &ys));
        hipLaunchKernelGGL(ce16_col_combine_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(M * E / 4, 256), 2048)), dim3(256), 0, s, b, b16, M * E / 4, ys, d_loss, d_score);
        IISAN_LAUNCH_OK();
        return IISAN_OK;
    }
    if (!fused && fwd_token != ce_token(bs, S, 0)) {
        iisan_set_error("inbatch_ce_bwd: fwd_token %llx is not what inbatch_ce_fwd returns for bs = %lld, S = %d", (unsigned long long)fwd_token, (long long)bs, S);
        return IISAN_EBADSHAPE;
    }
    if (fused)        // d_prec for d_loss = 1 is in the workspace (whatever the dev switch ce_fast says by now)
        hipLaunchKernelGGL(ce_scale_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(T * E / 4, 256), 1024)), dim3(256), 0, s, b.dprec, d_loss, d_prec, T * E / 4);
    else if (rowpass_ok(bs, S) && S + 1 <= 11) hipLaunchKernelGGL((ce_rowpass_kernel<CE_DPREC, 11>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score,
                       log_mask, b, (int)bs, S, d_loss, d_prec);
    else if (rowpass_ok(bs, S)) hipLaunchKernelGGL((ce_rowpass_kernel<CE_DPREC, MAXS1>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score,
                       log_mask, b, (int)bs, S, d_loss, d_prec);
    else if (S + 1 <= 11) hipLaunchKernelGGL((ce_pass_kernel<CE_DPREC, 11>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score, log_mask, b, bs, S,
                       d_loss, d_prec);
    else hipLaunchKernelGGL((ce_pass_kernel<CE_DPREC, MAXS1>), dim3((unsigned)ceil_div(T, 16)), dim3(256), 0, s, prec, score, log_mask, b, bs, S,
                       d_loss, d_prec);
    IISAN_LAUNCH_OK();
    if (rowpass_ok(bs, S) && S + 1 <= 11) hipLaunchKernelGGL((ce_colpass_kernel<11>), dim3((unsigned)ceil_div(M, 16)), dim3(256), 0, s, prec, score,
                       log_mask, b, (int)bs, S, d_loss, d_score);
    else if (rowpass_ok(bs, S)) hipLaunchKernelGGL((ce_colpass_kernel<MAXS1>), dim3((unsigned)ceil_div(M, 16)), dim3(256), 0, s, prec, score,
                       log_mask, b, (int)bs, S, d_loss, d_score);
    else if (S + 1 == 11) hipLaunchKernelGGL((ce_pass_kernel<CE_DSCORE, 11>), dim3((unsigned)ceil_div(M, 16)), dim3(256), 0, s, prec, score, log_mask, b, bs, S,
                       d_loss, d_score);
    else hipLaunchKernelGGL((ce_pass_kernel<CE_DSCORE, MAXS1>), dim3((unsigned)ceil_div(M, 16)), dim3(256), 0, s, prec, score, log_mask, b, bs, S,
                       d_loss, d_score);
    IISAN_LAUNCH_OK();
    return IISAN_OK;
}
