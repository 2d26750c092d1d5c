// fp32 MFMA GEMM for the fully-connected layers (theanet/layer/hidden.py:30, layer.py:83).
//
//   C[M,N] = op(A)[M,K] . op(B)[K,N]     v_mfma_f32_32x32x2_f32 (exact fp32, fmaf chain)
//
// * Block tile (64*WM) x (64*WN), BK = 16, 4 waves (2x2), each wave WM x WN accumulators of 32x32.
// * LDS tiles are [row][k] with k contiguous (row stride 20 floats = 5 x 16 B, coprime with the
//   16 slots of a bank row): a lane's eight k-values of a K-tile are TWO ds_read_b128
//   (lanes 0-31 take k = 0..7, lanes 32-63 k = 8..15 -- any k<->slot assignment is legal as long
//   as A and B agree), so a whole K-tile costs 2*(WM+WN) LDS reads for 8*WM*WN MFMAs.
// * Register-staged prefetch: global loads of tile t+1 are issued before the MFMAs of tile t
//   and written to the other LDS buffer after them; one barrier per K-tile.  Interior tiles use
//   unguarded float4 loads; only edge tiles take the bounds-checked path (uniform branch).
// * XCD-aware block decode: MI355X dispatches block b to XCD b % 8 and every XCD has a private
//   4 MB L2.  Blocks that share an A row-panel (forward / dgrad) or a split-K slice (wgrad) are
//   given ids congruent mod 8, so each XCD streams 1/8 of the big operand from HBM once and
//   serves the re-reads from its own L2.
// * Operands may be row- or column-contiguous (NN / NT / TN): forward, dgrad (dz.W^T) and wgrad
//   (x^T.dz) share the kernel; wgrad splits K (= batch) into 8 slabs reduced in a fixed order
//   (deterministic) and picks up the bias gradient (column sums of dz) from the tiles it stages.
// * Epilogues fuse bias + activation + dropout mask (forward) and activation-gradient + mask
//   (dgrad), so no elementwise pass touches HBM again.
//
// Skinny layers (n_out <= 16, the 10-way softmax layer) would waste >2/3 of a 32-wide MFMA
// tile and leave most CUs idle; they run on dedicated VALU kernels at the end of this file.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "elastic_field.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 16
#define LDK 20      // LDS row stride in floats (5 x 16 B: coprime with the 16 slots of a bank row)

enum { EPI_PLAIN = 0, EPI_FWD = 1, EPI_DGRAD = 2 };

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;          // or split-K workspace
    int M, N, K;
    int lda, ldb, ldc;
    int kchunk;        // K range per split (multiple of BK)
    int S;             // number of K splits
    int MT, NT;        // tiles along M and N
    int epi;
    const float* bias;       // EPI_FWD
    const float* prev_a;     // EPI_DGRAD: output of the layer below (same shape as C)
    const uint8_t* mask;     // EPI_FWD / EPI_DGRAD (may be NULL)
    int act;
    float act_prm;
    float* colsum;     // BSUM: [S][N] partial column sums of B
    int a_vec, b_vec;  // 16-byte vector loads allowed (ld % 4 == 0 and base aligned)
    int c_vec;         // 16-byte epilogue: ldc % 4 == 0, N % 4 == 0, C / prev_a 16-byte, masks 4-byte aligned
    // EPI_FWD dropout generated in the epilogue (c_vec only): keep = u01(philox(seed, step, e)) >= pdrop
    uint8_t* drop_out; // mask bytes written for the backward pass (NULL: no inline dropout)
    float pdrop;
    uint32_t dk0, dk1, dstep;
    const uint32_t* d_step;
    uint64_t elem0;    // global index of C[0][0] (multiple of 4)
    unsigned long long* dbg;   // TN_GEMM_DBG (gemm_f32_dma): per block {life, DMA wait, barrier wait, start} cycles
};

// ---- guarded tile loaders (edge tiles only) ------------------------------------------------
// KC: element (r, k) at src[r*ld + k]
__device__ __forceinline__ float4 load_kc(const float* __restrict__ src, int ld, int r, int rlim, int k,
                                          int klim, int vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rlim) {
        const float* p = src + (size_t)r * ld + k;
        if (vec && k + 3 < klim) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            if (k + 0 < klim) v.x = p[0];
            if (k + 1 < klim) v.y = p[1];
            if (k + 2 < klim) v.z = p[2];
            if (k + 3 < klim) v.w = p[3];
        }
    }
    return v;
}
// RC: element (r, k) at src[k*ld + r]
__device__ __forceinline__ float4 load_rc(const float* __restrict__ src, int ld, int r, int rlim, int k,
                                          int klim, int vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < klim) {
        const float* p = src + (size_t)k * ld + r;
        if (vec && r + 3 < rlim) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            if (r + 0 < rlim) v.x = p[0];
            if (r + 1 < rlim) v.y = p[1];
            if (r + 2 < rlim) v.z = p[2];
            if (r + 3 < rlim) v.w = p[3];
        }
    }
    return v;
}

// ---- shared epilogue ---------------------------------------------------------------------
// leaky-ReLU family / no activation: the only kinds the hot copy of an epilogue contains (see tn_act_fwd4)
__device__ __forceinline__ bool gemm_act_lk(const GemmArgs& g) {
    return g.epi == EPI_PLAIN || g.act == TN_ACT_LEAKY || g.act == TN_ACT_LINEAR;
}
template <bool LK>
__device__ __forceinline__ float gemm_act_fwd1(float z, int act, float prm) {
    if (LK) return act == TN_ACT_LEAKY ? fmaxf(0.f, z) + fminf(0.f, z) * prm : z;
    return tn_act_fwd(z, act, prm);
}
template <bool LK>
__device__ __forceinline__ float gemm_act_grad1(float a, int act, float prm) {
    if (LK) return act == TN_ACT_LEAKY ? (a > 0.f ? 1.f : (a < 0.f ? prm : (prm > 0.f ? 1.f + prm : 0.f))) : 1.f;
    return tn_act_grad_from_out(a, act, prm);
}

template <bool LK, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmArgs& g, f32x16 (&acc)[WM][WN], int m0, int n0,
                                                   int z, int wm, int wn, int lane) {
    const int hi = lane >> 5;
    float* Cz = g.C + (size_t)z * ((g.S > 1) ? (size_t)g.M * g.ldc : 0);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn * 32 * WN + 32 * j + (lane & 31);
        const bool cok = col < g.N;
        const int colc = min(col, g.N - 1);
        const float bias = (g.epi == EPI_FWD && g.bias) ? g.bias[colc] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int rbase = m0 + wm * 32 * WM + 32 * i + 4 * hi;
            float pa[16], pm[16];
            if (g.epi == EPI_DGRAD && g.prev_a) {   // all side loads first (in flight together)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
                    pa[r] = g.prev_a[(size_t)row * g.ldc + colc];
                }
            }
            if (g.mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
                    pm[r] = (float)g.mask[(size_t)row * g.ldc + colc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r];
                if (g.epi == EPI_FWD) {
                    v = gemm_act_fwd1<LK>(v + bias, g.act, g.act_prm);
                    if (g.mask) v *= pm[r];
                } else if (g.epi == EPI_DGRAD) {
                    if (g.prev_a) v *= gemm_act_grad1<LK>(pa[r], g.act, g.act_prm);
                    if (g.mask) v *= pm[r];
                }
                if (cok && row < g.M) Cz[(size_t)row * g.ldc + col] = v;
            }
        }
    }
}

template <int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[WM][WN], int m0, int n0,
                                              int z, int wm, int wn, int lane) {
    if (gemm_act_lk(g)) gemm_epilogue_impl<true, WM, WN>(g, acc, m0, n0, z, wm, wn, lane);
    else gemm_epilogue_impl<false, WM, WN>(g, acc, m0, n0, z, wm, wn, lane);
}

// ---- 16-byte epilogue (g.c_vec) ---------------------------------------------------------------
// The accumulators go through LDS so that every thread owns 4 consecutive columns of a row:
// one float4 store (plus one float4 / one 32-bit side load) instead of four scattered dwords,
// and exactly one Philox counter per thread and row for the inline dropout mask.
template <bool LK, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_vec_impl(const GemmArgs& g, f32x16 (&acc)[WM][WN], float* sC,
                                                       int m0, int n0, int z, int wm, int wn, int lane) {
    constexpr int BM = 64 * WM, BN = 64 * WN, LDC = BN + 4;
    const int hi = lane >> 5, t = threadIdx.x;
    float* Cz = g.C + (size_t)z * ((g.S > 1) ? (size_t)g.M * g.ldc : 0);
    __syncthreads();                                   // the operand tiles are dead
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[(wm * 32 * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi) * LDC + wn * 32 * WN + 32 * j +
                   (lane & 31)] = acc[i][j][r];
    __syncthreads();
    constexpr int CQ = BN / 4;                         // float4 groups per tile row
    constexpr int RP = 256 / CQ;                       // rows per pass
    constexpr int NPASS = BM / RP;
    const int c4 = 4 * (t % CQ), rl0 = t / CQ;
    const int col = n0 + c4;
    const bool cok = col < g.N;                        // N % 4 == 0: the whole group is inside
    const int colc = min(col, g.N - 4);
    float4 pa[NPASS];
    uint32_t pm[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {                  // side loads first (all in flight)
        const size_t o = (size_t)min(m0 + rl0 + RP * p, g.M - 1) * g.ldc + colc;
        if (g.epi == EPI_DGRAD && g.prev_a) pa[p] = *reinterpret_cast<const float4*>(g.prev_a + o);
        pm[p] = g.mask ? *reinterpret_cast<const uint32_t*>(g.mask + o) : 0x01010101u;
    }
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.epi == EPI_FWD && g.bias) {
        bias.x = g.bias[colc]; bias.y = g.bias[colc + 1]; bias.z = g.bias[colc + 2]; bias.w = g.bias[colc + 3];
    }
    const uint32_t dst = g.drop_out ? g.dstep + (g.d_step ? *g.d_step : 0u) : 0u;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int rl = rl0 + RP * p, row = m0 + rl;
        float4 v = *reinterpret_cast<const float4*>(sC + rl * LDC + c4);
        const size_t o = (size_t)min(row, g.M - 1) * g.ldc + colc;
        if (g.epi == EPI_FWD) {
            v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
            tn_act_fwd4<LK>(v, g.act, g.act_prm);
            if (g.drop_out) {
                // element e = elem0 + row*N + col uses word (e & 3) of philox(e >> 2): same
                // numbers as tn_dropout_mask (dropout_mask_kernel)
                const uint64_t cq = (g.elem0 + (uint64_t)min(row, g.M - 1) * (uint64_t)g.N + (uint64_t)colc) >> 2;
                const u32x4 rr = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), dst, TN_STREAM_DROPOUT,
                                            g.dk0, g.dk1);
                const uint32_t m = (tn_u01(rr.x) >= g.pdrop ? 1u : 0u) | (tn_u01(rr.y) >= g.pdrop ? 0x100u : 0u) |
                                   (tn_u01(rr.z) >= g.pdrop ? 0x10000u : 0u) |
                                   (tn_u01(rr.w) >= g.pdrop ? 0x1000000u : 0u);
                pm[p] = m;
                if (cok && row < g.M) *reinterpret_cast<uint32_t*>(g.drop_out + o) = m;
            }
        } else if (g.epi == EPI_DGRAD && g.prev_a) {
            tn_act_grad4<LK>(v, pa[p], g.act, g.act_prm);
        }
        if (g.epi != EPI_PLAIN && (g.mask || g.drop_out)) {
            v.x *= (float)(pm[p] & 0xffu);
            v.y *= (float)((pm[p] >> 8) & 0xffu);
            v.z *= (float)((pm[p] >> 16) & 0xffu);
            v.w *= (float)(pm[p] >> 24);
        }
        if (cok && row < g.M) *reinterpret_cast<float4*>(Cz + (size_t)row * g.ldc + col) = v;
    }
}

template <int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_vec(const GemmArgs& g, f32x16 (&acc)[WM][WN], float* sC,
                                                  int m0, int n0, int z, int wm, int wn, int lane) {
    if (gemm_act_lk(g)) gemm_epilogue_vec_impl<true, WM, WN>(g, acc, sC, m0, n0, z, wm, wn, lane);
    else gemm_epilogue_vec_impl<false, WM, WN>(g, acc, sC, m0, n0, z, wm, wn, lane);
}

// XCD-aware decode of the 1-D block id; returns false for padding blocks
__device__ __forceinline__ bool gemm_decode(const GemmArgs& g, int bid, int& mt, int& nt, int& z) {
    const int xcd = bid & 7, idx = bid >> 3;
    if (g.S == 1) {
        mt = (idx / g.NT) * 8 + xcd;      // all N-tiles of an A row-panel on one XCD
        nt = idx % g.NT;
        z = 0;
        return mt < g.MT;
    }
    const int per = g.MT * g.NT;
    if (g.S == 2 || g.S == 4) {           // a K-slab per group of 8 / S XCDs, the tiles dealt out inside the group
        const int gq = 8 / g.S, tile = idx * gq + xcd / g.S;
        z = xcd % g.S;
        mt = tile / g.NT;
        nt = tile - mt * g.NT;
        return tile < per;
    }
    z = (idx / per) * 8 + xcd;            // one K-slab per XCD
    const int rem = idx % per;
    mt = rem / g.NT;
    nt = rem % g.NT;
    return z < g.S;
}
// blocks of a launch of S K-slabs (S > 1)
static inline int gemm_grid_split(int S, int tiles) {
    return (S == 2 || S == 4) ? 8 * cdiv(tiles, 8 / S) : 8 * cdiv(S, 8) * tiles;
}

// ---- FAST kernel ---------------------------------------------------------------------------
// Preconditions (checked by the host): both operands 16-byte aligned with ld % 4 == 0, and the
// extent of every row-contiguous operand is a multiple of 4.  Then EVERY tile -- edge tiles
// included -- can use unguarded float4 loads with the row index clamped to the last valid
// row/group: the duplicates only feed output elements that are never stored.  The hot loop is
// branch-free, so hipcc emits counted vmcnt waits and the two-tile look-ahead really overlaps:
//   registers R0/R1 hold tiles t+1 / t+2, LDS buffers 0/1 hold tiles t / t+1.
// A K-tail (K % 16 != 0, e.g. dgrad with n_out = 500) is one guarded tile after the loop.
// floats of LDS a block needs: the operand tiles of the main loop, later the C tile of the epilogue
template <int WM, int WN, int BKT>
constexpr int gemm_smem_floats() {
    return 2 * (64 * WM + 64 * WN) * (BKT + 4) > 64 * WM * (64 * WN + 4) ? 2 * (64 * WM + 64 * WN) * (BKT + 4)
                                                                         : 64 * WM * (64 * WN + 4);
}

template <bool AKC, bool BKC, bool BSUM, int WM, int WN, int BKT>
__device__ __forceinline__ void gemm_fast_body(const GemmArgs& g, float* __restrict__ smem, int bid) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int NP = BKT / 16;        // 16-wide k passes per tile
    constexpr int KH = BKT / 2;         // k-values per lane and tile
    constexpr int LDT = BKT + 4;        // LDS row stride (20 or 36 floats: odd multiple of 16 B)
    float (*As)[BM][LDT] = reinterpret_cast<float (*)[BM][LDT]>(smem);
    float (*Bs)[BN][LDT] = reinterpret_cast<float (*)[BN][LDT]>(smem + 2 * BM * LDT);
    int mt, nt, z;
    if (!gemm_decode(g, bid, mt, nt, z)) return;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg) / BKT;              // full tiles
    const bool tail = (kend - kbeg) % BKT != 0;

    const int a_r = AKC ? (t >> 2) : 4 * (t >> 4);
    const int a_k = AKC ? 4 * (t & 3) : (t & 15);
    const int b_r = BKC ? (t >> 2) : 4 * (t >> 4);
    const int b_k = BKC ? 4 * (t & 3) : (t & 15);

    // clamped source pointers of tile 0; advance by `tile * step`
    const float* pA[WM];
    const float* pB[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int r = min(m0 + 64 * i + a_r, AKC ? g.M - 1 : g.M - 4);
        pA[i] = AKC ? g.A + (size_t)r * g.lda + kbeg + a_k : g.A + (size_t)(kbeg + a_k) * g.lda + r;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int r = min(n0 + 64 * j + b_r, BKC ? g.N - 1 : g.N - 4);
        pB[j] = BKC ? g.B + (size_t)r * g.ldb + kbeg + b_k : g.B + (size_t)(kbeg + b_k) * g.ldb + r;
    }
    const size_t stepA = AKC ? BKT : (size_t)BKT * g.lda;
    const size_t stepB = BKC ? BKT : (size_t)BKT * g.ldb;
    const size_t passA = AKC ? 16 : (size_t)16 * g.lda;      // offset of the second 16-wide k pass
    const size_t passB = BKC ? 16 : (size_t)16 * g.ldb;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 csum[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) csum[j] = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 a0[WM][NP], b0[WN][NP], a1[WM][NP], b1[WN][NP];          // staging sets R0 / R1
    const int last = max(nk - 1, 0);
#define GLOAD(RA, RB, TILE)                                                                     \
    {                                                                                           \
        const int tl_ = min((TILE), last);                                                      \
        _Pragma("unroll") for (int i = 0; i < WM; ++i)                                          \
            _Pragma("unroll") for (int p = 0; p < NP; ++p)                                      \
                RA[i][p] = *reinterpret_cast<const float4*>(pA[i] + tl_ * stepA + p * passA);   \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                          \
            _Pragma("unroll") for (int p = 0; p < NP; ++p)                                      \
                RB[j][p] = *reinterpret_cast<const float4*>(pB[j] + tl_ * stepB + p * passB);   \
    }
#define LSTORE(RA, RB, BUF)                                                                     \
    {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < WM; ++i)                                          \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                    \
                if (AKC) {                                                                      \
                    *reinterpret_cast<float4*>(&As[BUF][64 * i + a_r][16 * p + a_k]) = RA[i][p]; \
                } else {                                                                        \
                    As[BUF][64 * i + a_r + 0][16 * p + a_k] = RA[i][p].x;                       \
                    As[BUF][64 * i + a_r + 1][16 * p + a_k] = RA[i][p].y;                       \
                    As[BUF][64 * i + a_r + 2][16 * p + a_k] = RA[i][p].z;                       \
                    As[BUF][64 * i + a_r + 3][16 * p + a_k] = RA[i][p].w;                       \
                }                                                                               \
            }                                                                                   \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                          \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                    \
                if (BKC) {                                                                      \
                    *reinterpret_cast<float4*>(&Bs[BUF][64 * j + b_r][16 * p + b_k]) = RB[j][p]; \
                } else {                                                                        \
                    Bs[BUF][64 * j + b_r + 0][16 * p + b_k] = RB[j][p].x;                       \
                    Bs[BUF][64 * j + b_r + 1][16 * p + b_k] = RB[j][p].y;                       \
                    Bs[BUF][64 * j + b_r + 2][16 * p + b_k] = RB[j][p].z;                       \
                    Bs[BUF][64 * j + b_r + 3][16 * p + b_k] = RB[j][p].w;                       \
                }                                                                               \
            }                                                                                   \
    }
#define CSUM(RB)                                                                                \
    if (BSUM && !BKC) {                                                                         \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                          \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                    \
                csum[j].x += RB[j][p].x; csum[j].y += RB[j][p].y;                               \
                csum[j].z += RB[j][p].z; csum[j].w += RB[j][p].w;                               \
            }                                                                                   \
    }
    const int ar = wm * 32 * WM + (lane & 31), br = wn * 32 * WN + (lane & 31), hi = lane >> 5;
    // fragments of a K-tile: 8 k-values per lane and operand row (two b128 reads each)
#define LDFRAG(BUF, AV, BV)                                                                     \
    {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) {                                        \
            const float4* p_ = reinterpret_cast<const float4*>(&As[BUF][ar + 32 * i][KH * hi]); \
            _Pragma("unroll") for (int q_ = 0; q_ < KH / 4; ++q_) {                             \
                const float4 v_ = p_[q_];                                                       \
                AV[i][4 * q_] = v_.x; AV[i][4 * q_ + 1] = v_.y;                                 \
                AV[i][4 * q_ + 2] = v_.z; AV[i][4 * q_ + 3] = v_.w;                             \
            }                                                                                   \
        }                                                                                       \
        _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                        \
            const float4* p_ = reinterpret_cast<const float4*>(&Bs[BUF][br + 32 * j][KH * hi]); \
            _Pragma("unroll") for (int q_ = 0; q_ < KH / 4; ++q_) {                             \
                const float4 v_ = p_[q_];                                                       \
                BV[j][4 * q_] = v_.x; BV[j][4 * q_ + 1] = v_.y;                                 \
                BV[j][4 * q_ + 2] = v_.z; BV[j][4 * q_ + 3] = v_.w;                             \
            }                                                                                   \
        }                                                                                       \
    }
#define MMA(AV, BV)                                                                             \
    {                                                                                           \
        _Pragma("unroll") for (int s_ = 0; s_ < KH; ++s_)                                       \
            _Pragma("unroll") for (int i = 0; i < WM; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < WN; ++j)                                  \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i][s_], BV[j][s_], acc[i][j], 0, 0, 0); \
    }
    float avA[WM][KH], bvA[WN][KH], avB[WM][KH], bvB[WN][KH];

    // Software pipeline: the fragments of tile t+1 are read from LDS (asynchronously) BEFORE
    // the MFMAs of tile t are issued, so the matrix pipe never waits for an LDS round trip:
    //   store(t+1) ; barrier ; read frags(t+1) ; issue global loads(t+3) ; MFMA(t)
    if (nk > 0) {
        GLOAD(a0, b0, 0);
        GLOAD(a1, b1, 1);
        LSTORE(a0, b0, 0);
        CSUM(b0);
        __syncthreads();
        LDFRAG(0, avA, bvA);
        GLOAD(a0, b0, 2);
        int tile = 0;
        for (; tile + 1 < nk; tile += 2) {           // branch-free body
            LSTORE(a1, b1, 1);                        // tile + 1
            CSUM(b1);
            __syncthreads();
            LDFRAG(1, avB, bvB);
            GLOAD(a1, b1, tile + 3);
            MMA(avA, bvA);                            // tile
            LSTORE(a0, b0, 0);                        // tile + 2 (a clamped duplicate at the end)
            if (tile + 2 < nk) CSUM(b0);
            __syncthreads();
            LDFRAG(0, avA, bvA);
            GLOAD(a0, b0, tile + 4);
            MMA(avB, bvB);                            // tile + 1
        }
        if (nk & 1) MMA(avA, bvA);                    // odd count: the last tile is already in avA
    }
    if (tail) {                                       // guarded K-tail
        const int k0 = kbeg + nk * BKT;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                a0[i][p] = AKC ? load_kc(g.A, g.lda, m0 + 64 * i + a_r, g.M, k0 + 16 * p + a_k, kend, g.a_vec)
                               : load_rc(g.A, g.lda, m0 + 64 * i + a_r, g.M, k0 + 16 * p + a_k, kend, g.a_vec);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                b0[j][p] = BKC ? load_kc(g.B, g.ldb, n0 + 64 * j + b_r, g.N, k0 + 16 * p + b_k, kend, g.b_vec)
                               : load_rc(g.B, g.ldb, n0 + 64 * j + b_r, g.N, k0 + 16 * p + b_k, kend, g.b_vec);
        LSTORE(a0, b0, 0);
        CSUM(b0);
        __syncthreads();
        LDFRAG(0, avA, bvA);
        MMA(avA, bvA);
    }
#undef LDFRAG
#undef MMA
#undef GLOAD
#undef LSTORE
#undef CSUM
#undef COMPUTE

    if (g.c_vec)
        gemm_epilogue_vec<WM, WN>(g, acc, smem, m0, n0, z, wm, wn, lane);
    else
        gemm_epilogue<WM, WN>(g, acc, m0, n0, z, wm, wn, lane);

    if (BSUM && !BKC && mt == 0) {
        // reduce csum over the 16 k-lanes of the staging layout (thread = (k = t&15, q = t>>4))
        __syncthreads();
        float* red = &As[0][0][0];   // reuse: [16][BN]  (16*BN <= BM*LDT)
#pragma unroll
        for (int j = 0; j < WN; ++j)
            *reinterpret_cast<float4*>(&red[(t & 15) * BN + 64 * j + 4 * (t >> 4)]) = csum[j];
        __syncthreads();
        if (t < BN) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += red[k * BN + t];
            if (n0 + t < g.N) g.colsum[(size_t)z * g.N + n0 + t] = sum;
        }
    }
}

template <bool AKC, bool BKC, bool BSUM, int WM, int WN, int BKT>
__global__ __launch_bounds__(256) void gemm_f32_fast(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<WM, WN, BKT>()];
    gemm_fast_body<AKC, BKC, BSUM, WM, WN, BKT>(g, smem, (int)blockIdx.x);
}

// ---- DMA kernel (round 5) --------------------------------------------------------------------
// Same product, same 64 x 64 x 16 tiles, same fmaf chain per output element as the FAST kernel -- but no operand ever
// passes through a vector register on its way to LDS and the tile loop holds NO vector-ALU instruction at all.  An fp32
// MFMA runs at the vector rate and does not overlap vector instructions of its SIMD (profiles/r02_probe_mfma_valu.txt):
// the FAST kernel's address arithmetic, ds_writes' operand moves and column sums came straight out of the matrix rate
// (0.57-0.63 of peak on the fc1 products of mnist.prms).  Here:
//   * both operands are staged by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, 1 KB per wave instruction, LDS
//     destination = M0 + 16 * lane): a wave issues ONE instruction per operand and K-tile.  The global address is
//     SGPR base (advanced by SALU per tile) + a per-lane 32-bit offset fixed in the prologue;
//   * a k-contiguous operand lands as [64 rows][16 k] with 64-byte rows; the lane that owns LDS slot (row, j) fetches
//     source chunk j ^ ((row >> 2) & 3), and the MFMA lanes read chunk c at slot c ^ ((row >> 2) & 3): conflict-free
//     ds_read_b128 without a pad (the FAST kernel pads rows to 80 bytes, which a DMA cannot write);
//   * a row-contiguous operand lands as the plain [16 k][64 rows] image: lane (row, half) reads A[k][row] with one
//     ds_read_b32 per MFMA (immediate offsets; 32 consecutive floats per half-wave);
//   * a ring of NS = 4 stages (8 KB each), tile j+3 in flight while tile j is multiplied; the fragments of tile j are
//     read into registers while the MFMAs of tile j-1 issue; one s_waitcnt vmcnt(N) + s_barrier per tile, counted by
//     hand (hipcc knows nothing of the DMAs);
//   * the tile loop is unrolled over the ring so that every LDS address is a per-lane base + an immediate.
// The bias gradient of a weight-gradient product (column sums of dz, which the FAST kernel adds up from its staging
// registers) moves to gemm_colsum_block: a handful of extra blocks of the same launch, same partition, same order
// of additions, same bits.
#define GD_SB 8192                                   // bytes per stage: A tile, then B tile
template <int NS>
constexpr int gemm_dma_smem_floats() {
    return NS * GD_SB / 4 > 64 * 68 ? NS * GD_SB / 4 : 64 * 68;
}

// 16 bytes per lane, global (sbase + voff) -> LDS (lds_dst + 16 * lane)
// (the LDS destination is formed inside the statement, base + constant: as a plain operand hipcc keeps one SGPR per
// stage and operand live across the loop and spills)
template <int LOFF>
__device__ __forceinline__ void gd_dma(const void* sbase, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base), "n"(LOFF) : "memory", "scc");
}
// this wave's DMAs older than the newest 2 * TILES have landed, its LDS reads have returned; then everybody's
template <int TILES>
__device__ __forceinline__ void gd_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * TILES) : "memory");
}
// the same with cycle stamps (TN_GEMM_DBG)
template <int TILES>
__device__ __forceinline__ void gd_wait_barrier_dbg(unsigned long long& d_wait, unsigned long long& d_bar) {
    const unsigned long long s0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * TILES) : "memory");
    const unsigned long long s1 = __builtin_readcyclecounter();
    asm volatile("s_barrier" ::: "memory");
    const unsigned long long s2 = __builtin_readcyclecounter();
    d_wait += s1 - s0;
    d_bar += s2 - s1;
}
template <int N> using gd_ic = std::integral_constant<int, N>;

// NS = ring stages (even): NS - 1 tiles ahead of the one being multiplied.  2 while many blocks share a CU (the other
// blocks' waves cover a DMA's latency), 8 for launches of two blocks per CU (a block must cover it by itself once its
// CU-mate has left: two waves of a SIMD are served oldest first, so the older block finishes well ahead of the younger).
template <bool AKC, bool BKC, int NS, bool DBG = false>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs& g, float* __restrict__ smem, int bid) {
    static_assert(NS >= 2 && NS % 2 == 0 && NS <= 8, "ring stages");
    int mt, nt, z;
    if (!gemm_decode(g, bid, mt, nt, z)) return;
    unsigned long long d_t0 = 0, d_wait = 0, d_bar = 0, d_loop = 0, d_w0 = 0, d_pro = 0;
    if (DBG) { d_t0 = __builtin_readcyclecounter(); d_w0 = wall_clock64(); }
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = mt * 64, n0 = nt * 64;
    const int kbeg = z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg) >> 4;                       // full tiles (>= 1: the host checks)
    const bool tail = ((kend - kbeg) & 15) != 0;
    char* const sm = reinterpret_cast<char*>(smem);

    // ---- DMA sources: per-lane byte offset (fixed) + wave-uniform base (advances per tile)
    unsigned offA, offB;
    if (AKC) {
        const int row = min(m0 + 16 * wave + (lane >> 2), g.M - 1);
        offA = ((unsigned)row * (unsigned)g.lda + 4u * ((lane & 3) ^ ((lane >> 4) & 3))) * 4u;
    } else {
        const int col = min(m0 + 4 * (lane & 15), g.M - 4);
        offA = ((unsigned)(4 * wave + (lane >> 4)) * (unsigned)g.lda + (unsigned)col) * 4u;
    }
    if (BKC) {
        const int row = min(n0 + 16 * wave + (lane >> 2), g.N - 1);
        offB = ((unsigned)row * (unsigned)g.ldb + 4u * ((lane & 3) ^ ((lane >> 4) & 3))) * 4u;
    } else {
        const int col = min(n0 + 4 * (lane & 15), g.N - 4);
        offB = ((unsigned)(4 * wave + (lane >> 4)) * (unsigned)g.ldb + (unsigned)col) * 4u;
    }
    const char* const gA = reinterpret_cast<const char*>(AKC ? g.A + kbeg : g.A + (size_t)kbeg * g.lda);
    const char* const gB = reinterpret_cast<const char*>(BKC ? g.B + kbeg : g.B + (size_t)kbeg * g.ldb);
    const size_t stepA = AKC ? 64 : (size_t)64 * g.lda;      // bytes per K-tile
    const size_t stepB = BKC ? 64 : (size_t)64 * g.ldb;
    const unsigned ldsw = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm + 1024u * wave;   // this wave's KB of a stage's A tile (B: + 4096)

    // ---- fragment reads: per-lane LDS byte offsets inside a stage
    const int ra = wm * 32 + r, rb = wn * 32 + r;
    const int fa0 = AKC ? ra * 64 + (((2 * hi) ^ ((ra >> 2) & 3)) << 4) : (8 * hi) * 256 + ra * 4;
    const int fb0 = 4096 + (BKC ? rb * 64 + (((2 * hi) ^ ((rb >> 2) & 3)) << 4) : (8 * hi) * 256 + rb * 4);
    const int fa1 = fa0 ^ 16, fb1 = fb0 ^ 16;                // second 16-byte chunk (k-contiguous operands only)

    f32x16 acc[1][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][0][i] = 0.f;
    float av[2][8], bv[2][8];

    // tiles are issued in order: running source pointers
    const char* pA = gA;
    const char* pB = gB;
    auto issue = [&](auto Sc) __attribute__((always_inline)) {
        constexpr int S = decltype(Sc)::value;
        gd_dma<S * GD_SB>(pA, offA, ldsw);
        gd_dma<S * GD_SB + 4096>(pB, offB, ldsw);
        pA += stepA;
        pB += stepB;
    };
    auto frag = [&](auto Sc) __attribute__((always_inline)) {
        constexpr int S = decltype(Sc)::value, SET = S & 1;
        const char* s_ = sm + S * GD_SB;
        if (AKC) {
            const float4 u_ = *reinterpret_cast<const float4*>(s_ + fa0);
            const float4 v_ = *reinterpret_cast<const float4*>(s_ + fa1);
            av[SET][0] = u_.x; av[SET][1] = u_.y; av[SET][2] = u_.z; av[SET][3] = u_.w;
            av[SET][4] = v_.x; av[SET][5] = v_.y; av[SET][6] = v_.z; av[SET][7] = v_.w;
        } else {
#pragma unroll
            for (int q_ = 0; q_ < 8; ++q_) av[SET][q_] = *reinterpret_cast<const float*>(s_ + fa0 + 256 * q_);
        }
        if (BKC) {
            const float4 u_ = *reinterpret_cast<const float4*>(s_ + fb0);
            const float4 v_ = *reinterpret_cast<const float4*>(s_ + fb1);
            bv[SET][0] = u_.x; bv[SET][1] = u_.y; bv[SET][2] = u_.z; bv[SET][3] = u_.w;
            bv[SET][4] = v_.x; bv[SET][5] = v_.y; bv[SET][6] = v_.z; bv[SET][7] = v_.w;
        } else {
#pragma unroll
            for (int q_ = 0; q_ < 8; ++q_) bv[SET][q_] = *reinterpret_cast<const float*>(s_ + fb0 + 256 * q_);
        }
    };
    auto mma = [&](auto Sc) __attribute__((always_inline)) {
        constexpr int SET = decltype(Sc)::value & 1;
#pragma unroll
        for (int q_ = 0; q_ < 8; ++q_)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[SET][q_], bv[SET][q_], acc[0][0], 0, 0, 0);
    };
    // wait until at most n (<= NS - 2) of this wave's tiles are in flight, then the barrier
    auto wait_n = [&](int n) __attribute__((always_inline)) {
#define GD_WB(N) do { if (DBG) gd_wait_barrier_dbg<(N)>(d_wait, d_bar); else gd_wait_barrier<(N)>(); } while (0)
        if (n >= NS - 2) GD_WB(NS - 2);
        else if (NS > 7 && n == 5) GD_WB(NS > 7 ? 5 : 0);
        else if (NS > 6 && n == 4) GD_WB(NS > 6 ? 4 : 0);
        else if (NS > 5 && n == 3) GD_WB(NS > 5 ? 3 : 0);
        else if (NS > 4 && n == 2) GD_WB(NS > 4 ? 2 : 0);
        else if (NS > 3 && n == 1) GD_WB(NS > 3 ? 1 : 0);
        else GD_WB(0);
    };
    // step j (tile j sits in stage S = j % NS, its fragments go to register set j & 1 = S & 1): wait for tile j -- at
    // most NS - 2 younger tiles stay in flight --, read its fragments, refill the stage tile j-1 has just left with tile
    // j + NS - 1, and multiply tile j-1.  sched_barrier: hipcc otherwise moves a step's MFMAs (register-only
    // instructions) below the NEXT step's wait + barrier, where the wave would park with an idle matrix pipe.
    // HOT: steps with NS - 1 more tiles behind them (no tests); otherwise the last steps of a block.
    auto step = [&](auto Sc, auto Hc, int j) __attribute__((always_inline)) {
        constexpr int S = decltype(Sc)::value;
        constexpr bool HOT = decltype(Hc)::value != 0;
        if (HOT) {
            GD_WB(NS - 2);
            frag(gd_ic<S>{});
            issue(gd_ic<(S + NS - 1) % NS>{});
        } else {
            wait_n(nk - 1 - j);
            frag(gd_ic<S>{});
            if (j + NS - 1 < nk) issue(gd_ic<(S + NS - 1) % NS>{});
        }
        mma(gd_ic<(S + 1) & 1>{});
        __builtin_amdgcn_sched_barrier(0);
    };
#undef GD_WB
    // a trip of NS steps starting at stage 1 (j = 1 (mod NS))
    auto trip = [&](auto Hc, int j) __attribute__((always_inline)) {
        step(gd_ic<1 % NS>{}, Hc, j);
        if (NS > 2) { step(gd_ic<2 % NS>{}, Hc, j + 1); step(gd_ic<3 % NS>{}, Hc, j + 2); }
        if (NS > 4) { step(gd_ic<4 % NS>{}, Hc, j + 3); step(gd_ic<5 % NS>{}, Hc, j + 4); }
        if (NS > 6) { step(gd_ic<6 % NS>{}, Hc, j + 5); step(gd_ic<7 % NS>{}, Hc, j + 6); }
        step(gd_ic<0>{}, Hc, j + NS - 1);
    };

    issue(gd_ic<0>{});
    if (NS > 2) { if (1 < nk) issue(gd_ic<1 % NS>{}); if (2 < nk) issue(gd_ic<2 % NS>{}); }
    if (NS > 4) { if (3 < nk) issue(gd_ic<3 % NS>{}); if (4 < nk) issue(gd_ic<4 % NS>{}); }
    if (NS > 6) { if (5 < nk) issue(gd_ic<5 % NS>{}); if (6 < nk) issue(gd_ic<6 % NS>{}); }
    wait_n(min(nk, NS - 1) - 1);
    if (DBG) d_pro = __builtin_readcyclecounter();
    frag(gd_ic<0>{});
    if (NS - 1 < nk) issue(gd_ic<NS - 1>{});
    __builtin_amdgcn_sched_barrier(0);
    int j = 1;
    for (; j + 2 * NS - 1 <= nk; j += NS) trip(gd_ic<1>{}, j);     // (j + NS - 1) + (NS - 1) < nk
    for (; j + NS <= nk; j += NS) trip(gd_ic<0>{}, j);
    if (j < nk) { step(gd_ic<1 % NS>{}, gd_ic<0>{}, j); ++j; }
    if (NS > 2) {
        if (j < nk) { step(gd_ic<2 % NS>{}, gd_ic<0>{}, j); ++j; }
        if (j < nk) { step(gd_ic<3 % NS>{}, gd_ic<0>{}, j); ++j; }
    }
    if (NS > 4) {
        if (j < nk) { step(gd_ic<4 % NS>{}, gd_ic<0>{}, j); ++j; }
        if (j < nk) { step(gd_ic<5 % NS>{}, gd_ic<0>{}, j); ++j; }
    }
    if (NS > 6) {
        if (j < nk) { step(gd_ic<6 % NS>{}, gd_ic<0>{}, j); ++j; }
        if (j < nk) { step(gd_ic<7 % NS>{}, gd_ic<0>{}, j); ++j; }
    }
    if ((nk - 1) & 1) mma(gd_ic<1>{});
    else mma(gd_ic<0>{});
    if (DBG) d_loop = __builtin_readcyclecounter();

    if (tail) {
        // guarded K-tail through registers, written in the DMA's layout (thread t owns the 16 bytes at t * 16 of a tile)
        const int k0 = kbeg + nk * 16;
        __syncthreads();
        float4 ta, tb;
        if (AKC) ta = load_kc(g.A, g.lda, m0 + (t >> 2), g.M, k0 + 4 * ((t & 3) ^ ((t >> 4) & 3)), kend, g.a_vec);
        else ta = load_rc(g.A, g.lda, m0 + 4 * (t & 15), g.M, k0 + (t >> 4), kend, g.a_vec);
        if (BKC) tb = load_kc(g.B, g.ldb, n0 + (t >> 2), g.N, k0 + 4 * ((t & 3) ^ ((t >> 4) & 3)), kend, g.b_vec);
        else tb = load_rc(g.B, g.ldb, n0 + 4 * (t & 15), g.N, k0 + (t >> 4), kend, g.b_vec);
        *reinterpret_cast<float4*>(sm + 16 * t) = ta;
        *reinterpret_cast<float4*>(sm + 4096 + 16 * t) = tb;
        __syncthreads();
        frag(gd_ic<0>{});
        mma(gd_ic<0>{});
    }
    if (g.c_vec) {
        gemm_epilogue_vec<1, 1>(g, acc, smem, m0, n0, z, wm, wn, lane);
    } else {
        __syncthreads();
        gemm_epilogue<1, 1>(g, acc, m0, n0, z, wm, wn, lane);
    }
    if (DBG && g.dbg && lane == 0) {
        unsigned long long* d = g.dbg + 8 * ((size_t)bid * 4 + wave);
        d[0] = __builtin_readcyclecounter() - d_t0; d[1] = d_wait; d[2] = d_bar; d[3] = d_pro - d_t0; d[4] = d_loop - d_t0;
        d[5] = wall_clock64(); d[6] = d_w0;
    }
}

// column sums of B over a K slab (the bias gradient of a weight-gradient product): block idx = (slab z, column tile
// nt).  Thread (row lane t >> 4, column group t & 15) adds rows kbeg + (t >> 4) + 16 i in order, the 16 row lanes are
// added in order -- the partition and the order of the FAST kernel's staging-register sums, hence its bits.
__device__ __forceinline__ void gemm_colsum_block(const GemmArgs& g, float* __restrict__ smem, int idx) {
    const int z = idx / g.NT, nt = idx - z * g.NT;
    const int t = threadIdx.x, rl = t >> 4, cg = t & 15;
    const int n0 = nt * 64;
    const int kbeg = z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int col = min(n0 + 4 * cg, g.N - 4);
    const float* p = g.B + (size_t)(kbeg + rl) * g.ldb + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = kbeg + rl;
    for (; k + 16 * 7 < kend; k += 16 * 8) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(p + (size_t)(16 * i) * g.ldb);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        p += (size_t)(16 * 8) * g.ldb;
    }
    for (; k < kend; k += 16) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        p += (size_t)16 * g.ldb;
    }
    *reinterpret_cast<float4*>(&smem[rl * 64 + 4 * cg]) = s;
    __syncthreads();
    if (t < 64) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += smem[q * 64 + t];
        if (n0 + t < g.N) g.colsum[(size_t)z * g.N + n0 + t] = sum;
    }
}

// blocks [0, nb): the product; [nb, nb + ncs): column sums (BSUM launches)
template <bool AKC, bool BKC, int NS, bool DBG = false>
__global__ __launch_bounds__(256) void gemm_f32_dma(GemmArgs g, int nb) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_dma_smem_floats<NS>()];
    const int bid = blockIdx.x;
    if (bid >= nb) {
        gemm_colsum_block(g, smem, bid - nb);
        return;
    }
    gemm_dma_body<AKC, BKC, NS, DBG>(g, smem, bid);
}

// Two INDEPENDENT products in one launch (e.g. a layer's weight gradient and its input gradient,
// which only share dz): groups of 8 blocks alternate between the problems, so blocks of both are
// co-resident on every CU -- the prologue / epilogue of one product overlaps the main loop of the
// other, and a kernel boundary disappears.
template <bool A1, bool B1, bool S1, bool A2, bool B2, bool S2>
__global__ __launch_bounds__(256, 6) void gemm_f32_pair(GemmArgs g1, GemmArgs g2, int n1, int n2, ElField rider) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<1, 1, 16>()];
    const int bid = blockIdx.x, grp = bid >> 3, l8 = bid & 7;
    if (bid >= n1 + n2) {
        // rider blocks behind the two products: a light independent job of the step (the elastic
        // field of the next minibatch) that disappears under the GEMMs instead of owning a launch
        // (it works in the GEMM tile's LDS: a launch-wide dynamic allocation on top of it would cost every block
        // of the two products a sixth resident block per CU)
        elastic_field_block<true>(rider, smem, bid - n1 - n2);
        return;
    }
    const int G1 = n1 >> 3, G2 = n2 >> 3, Gm = min(G1, G2);
    int prob, idx;
    if (grp < 2 * Gm) {
        prob = grp & 1;
        idx = grp >> 1;
    } else {
        prob = G1 > G2 ? 0 : 1;
        idx = grp - Gm;
    }
    if (prob == 0)
        gemm_fast_body<A1, B1, S1, 1, 1, 16>(g1, smem, idx * 8 + l8);
    else
        gemm_fast_body<A2, B2, S2, 1, 1, 16>(g2, smem, idx * 8 + l8);
}

// the same launch on the DMA bodies; blocks: [n1 + n2 interleaved groups][ncs column-sum blocks of product 1][riders]
template <bool A1, bool B1, bool A2, bool B2, int NS, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_f32_pair_dma(GemmArgs g1, GemmArgs g2, int n1, int n2, int ncs, ElField rider) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_dma_smem_floats<NS>()];
    const int bid = blockIdx.x, grp = bid >> 3, l8 = bid & 7;
    if (bid >= n1 + n2) {
        if (bid < n1 + n2 + ncs) gemm_colsum_block(g1, smem, bid - n1 - n2);
        else elastic_field_block<true>(rider, smem, bid - n1 - n2 - ncs);
        return;
    }
    const int G1 = n1 >> 3, G2 = n2 >> 3, Gm = min(G1, G2);
    int prob, idx;
    if (grp < 2 * Gm) {
        prob = grp & 1;
        idx = grp >> 1;
    } else {
        prob = G1 > G2 ? 0 : 1;
        idx = grp - Gm;
    }
    if (prob == 0)
        gemm_dma_body<A1, B1, NS>(g1, smem, idx * 8 + l8);
    else
        gemm_dma_body<A2, B2, NS>(g2, smem, idx * 8 + l8);
}

// ---- DEEP kernel: few output tiles, a reduction long enough to split ---------------------------------
// A short batch (one rank of a sharded step: 512 x 500 <- 720) has 64 tiles of 64 x 64, each a chain of 45
// dependent K-tiles behind an LDS round trip and a barrier: 64 blocks on 256 CUs and 20-27 us of latency for
// 0.37 GFLOP.  Here a block owns a 32 x 32 tile and its NW waves split the REDUCTION: every wave multiplies its
// own K range with operands loaded straight from global memory into the MFMA register layout (lane (r, hi) of
// v_mfma_f32_32x32x2_f32 supplies A[r][k] and B[k][r] for the k of its half: with k-contiguous A that is two
// 16-byte loads per K-tile, with row-contiguous B eight dword loads that are 128 B contiguous across lanes),
// DEPTH tiles per trip in flight, no LDS and no barrier in the loop; the NW partial tiles are added in wave order through
// LDS and the 16-byte epilogue of the other kernels (bias + activation + inline dropout / act' * mask) runs once.
// The price is operand re-reads from L2 (no sharing between the tiles of a block row): M*N*K/4 bytes, so the
// host only picks it while that is small (gemm_deep_ok).
template <bool BKC, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_f32_deep(GemmArgs g) {
    constexpr int DEPTH = 6;
    __shared__ __attribute__((aligned(16))) float red[NW][32][36];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, hi = lane >> 5;
    const int mt = blockIdx.x / g.NT, nt = blockIdx.x - mt * g.NT;
    const int m0 = 32 * mt, n0 = 32 * nt;
    // K-tiles of this wave
    const int ktiles = (g.K + 15) >> 4, per = (ktiles + NW - 1) / NW;
    const int tb = w * per, te = min(ktiles, tb + per);
    const int kend = min(g.K, 16 * te);
    const int ntl = max(te - tb, 0), ntp = (ntl + DEPTH - 1) / DEPTH * DEPTH;
    const float* pa = g.A + (size_t)min(m0 + r, g.M - 1) * g.lda;            // A(m, k) = A[m*lda + k]
    const int nc = min(n0 + r, g.N - 1);
    const float* pb = BKC ? g.B + (size_t)nc * g.ldb : g.B + nc;             // BKC: B(k, n) = B[n*ldb + k]

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 ra[DEPTH][2];
    float rb[DEPTH][8];
#define DLOAD(SLOT, TILE)                                                                         \
    {                                                                                             \
        const int k0_ = 16 * (tb + (TILE)) + 8 * hi;                                              \
        const int ka_ = min(k0_, g.K - 4), kb_ = min(k0_ + 4, g.K - 4);                           \
        ra[SLOT][0] = *reinterpret_cast<const float4*>(pa + ka_);                                 \
        ra[SLOT][1] = *reinterpret_cast<const float4*>(pa + kb_);                                 \
        if (BKC) {                                                                                \
            const float4 u_ = *reinterpret_cast<const float4*>(pb + ka_);                         \
            const float4 v_ = *reinterpret_cast<const float4*>(pb + kb_);                         \
            rb[SLOT][0] = u_.x; rb[SLOT][1] = u_.y; rb[SLOT][2] = u_.z; rb[SLOT][3] = u_.w;       \
            rb[SLOT][4] = v_.x; rb[SLOT][5] = v_.y; rb[SLOT][6] = v_.z; rb[SLOT][7] = v_.w;       \
        } else {                                                                                  \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_)                                      \
                rb[SLOT][s_] = pb[(size_t)min(k0_ + s_, g.K - 1) * g.ldb];                        \
        }                                                                                         \
    }
    // tiles past the wave's range (padding of the unrolled loop, the K tail) multiply by zero
#define DMMA(SLOT, TILE)                                                                          \
    {                                                                                             \
        const int k0_ = 16 * (tb + (TILE)) + 8 * hi;                                              \
        const float av_[8] = {ra[SLOT][0].x, ra[SLOT][0].y, ra[SLOT][0].z, ra[SLOT][0].w,         \
                              ra[SLOT][1].x, ra[SLOT][1].y, ra[SLOT][1].z, ra[SLOT][1].w};        \
        _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                        \
            const bool in_ = k0_ + s_ < kend;        /* both operands: 0 * inf would be NaN */    \
            const float bv_ = in_ ? rb[SLOT][s_] : 0.f;                                           \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in_ ? av_[s_] : 0.f, bv_, acc, 0, 0, 0);   \
        }                                                                                         \
    }
    // a trip = DEPTH tiles: all their loads are issued back to back, then the MFMAs follow under counted waits.
    // (Carrying refilled slots from one trip to the next did not survive the compiler: the values were copied
    // at the loop latch, which waits for the loads -- so the look-ahead across trips comes from the block's other
    // waves instead.)
    for (int tile = 0; tile < ntp; tile += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) DLOAD(d, tile + d);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) DMMA(d, tile + d);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef DLOAD
#undef DMMA
    // ---- add the NW partial tiles in wave order; thread = 4 consecutive columns of a row ----------------
#pragma unroll
    for (int i = 0; i < 16; ++i) red[w][(i & 3) + 8 * (i >> 2) + 4 * hi][r] = acc[i];
    __syncthreads();
    if (t >= 256) return;
    const int rl = t >> 3, c4 = 4 * (t & 7);
    float4 v = *reinterpret_cast<const float4*>(&red[0][rl][c4]);
#pragma unroll
    for (int q = 1; q < NW; ++q) {
        const float4 u = *reinterpret_cast<const float4*>(&red[q][rl][c4]);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int row = m0 + rl, col = n0 + c4;
    if (row >= g.M || col >= g.N) return;                 // N % 4 == 0: a group is inside or outside
    const size_t o = (size_t)row * g.ldc + col;
    uint32_t pm = g.mask ? *reinterpret_cast<const uint32_t*>(g.mask + o) : 0x01010101u;
    if (g.epi == EPI_FWD) {
        if (g.bias) {
            v.x += g.bias[col]; v.y += g.bias[col + 1]; v.z += g.bias[col + 2]; v.w += g.bias[col + 3];
        }
        tn_act_fwd4(v, g.act, g.act_prm);
        if (g.drop_out) {      // same numbers as gemm_epilogue_vec / tn_dropout_mask
            const uint32_t dst = g.dstep + (g.d_step ? *g.d_step : 0u);
            const uint64_t cq = (g.elem0 + (uint64_t)row * (uint64_t)g.N + (uint64_t)col) >> 2;
            const u32x4 rr = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), dst, TN_STREAM_DROPOUT, g.dk0, g.dk1);
            pm = (tn_u01(rr.x) >= g.pdrop ? 1u : 0u) | (tn_u01(rr.y) >= g.pdrop ? 0x100u : 0u) |
                 (tn_u01(rr.z) >= g.pdrop ? 0x10000u : 0u) | (tn_u01(rr.w) >= g.pdrop ? 0x1000000u : 0u);
            *reinterpret_cast<uint32_t*>(g.drop_out + o) = pm;
        }
    } else if (g.epi == EPI_DGRAD && g.prev_a) {
        const float4 pa4 = *reinterpret_cast<const float4*>(g.prev_a + o);
        tn_act_grad4(v, pa4, g.act, g.act_prm);
    }
    if (g.epi != EPI_PLAIN && (g.mask || g.drop_out)) {
        v.x *= (float)(pm & 0xffu);
        v.y *= (float)((pm >> 8) & 0xffu);
        v.z *= (float)((pm >> 16) & 0xffu);
        v.w *= (float)(pm >> 24);
    }
    *reinterpret_cast<float4*>(g.C + o) = v;
}

// ---- generic kernel (any alignment / extent): guarded loads, single-stage prefetch ---------
template <bool AKC, bool BKC, bool BSUM>
__global__ __launch_bounds__(256) void gemm_f32_generic(GemmArgs g) {
    constexpr int WM = 1, WN = 1, BM = 64, BN = 64;
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDK];
    int mt, nt, z;
    if (!gemm_decode(g, (int)blockIdx.x, mt, nt, z)) return;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    const int a_r = AKC ? (t >> 2) : 4 * (t >> 4);
    const int a_k = AKC ? 4 * (t & 3) : (t & 15);
    const int b_r = BKC ? (t >> 2) : 4 * (t >> 4);
    const int b_k = BKC ? 4 * (t & 3) : (t & 15);
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra, rb;
    auto gload = [&](int tile) {
        const int k0 = kbeg + tile * BK;
        ra = AKC ? load_kc(g.A, g.lda, m0 + a_r, g.M, k0 + a_k, kend, g.a_vec)
                 : load_rc(g.A, g.lda, m0 + a_r, g.M, k0 + a_k, kend, g.a_vec);
        rb = BKC ? load_kc(g.B, g.ldb, n0 + b_r, g.N, k0 + b_k, kend, g.b_vec)
                 : load_rc(g.B, g.ldb, n0 + b_r, g.N, k0 + b_k, kend, g.b_vec);
    };
    auto lstore = [&](int buf) {
        if (AKC) {
            *reinterpret_cast<float4*>(&As[buf][a_r][a_k]) = ra;
        } else {
            As[buf][a_r + 0][a_k] = ra.x; As[buf][a_r + 1][a_k] = ra.y;
            As[buf][a_r + 2][a_k] = ra.z; As[buf][a_r + 3][a_k] = ra.w;
        }
        if (BKC) {
            *reinterpret_cast<float4*>(&Bs[buf][b_r][b_k]) = rb;
        } else {
            Bs[buf][b_r + 0][b_k] = rb.x; Bs[buf][b_r + 1][b_k] = rb.y;
            Bs[buf][b_r + 2][b_k] = rb.z; Bs[buf][b_r + 3][b_k] = rb.w;
        }
        if (BSUM && !BKC) {
            csum.x += rb.x; csum.y += rb.y; csum.z += rb.z; csum.w += rb.w;
        }
    };
    if (ntiles > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    const int ar = wm * 32 + (lane & 31), br = wn * 32 + (lane & 31), hi = lane >> 5;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) gload(tile + 1);
        const float4* pa = reinterpret_cast<const float4*>(&As[buf][ar][8 * hi]);
        const float4* pb = reinterpret_cast<const float4*>(&Bs[buf][br][8 * hi]);
        const float4 al = pa[0], au = pa[1], bl = pb[0], bu = pb[1];
        const float av[8] = {al.x, al.y, al.z, al.w, au.x, au.y, au.z, au.w};
        const float bv[8] = {bl.x, bl.y, bl.z, bl.w, bu.x, bu.y, bu.z, bu.w};
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s_], bv[s_], acc[0][0], 0, 0, 0);
        if (tile + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    gemm_epilogue<1, 1>(g, acc, m0, n0, z, wm, wn, lane);
    if (BSUM && !BKC && mt == 0) {
        __syncthreads();
        float* red = &As[0][0][0];
        *reinterpret_cast<float4*>(&red[(t & 15) * BN + 4 * (t >> 4)]) = csum;
        __syncthreads();
        if (t < BN) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += red[k * BN + t];
            if (n0 + t < g.N) g.colsum[(size_t)z * g.N + n0 + t] = sum;
        }
    }
}


static inline int vec_ok(const void* p, int ld) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0);
}

static int tn_tune_tile() { return 0; }       // 0 auto (1 / 2 would force 64x64 / 128x64: sweeps of rounds 1-2)

static int tn_tune_bk() { return 16; }        // K-tile depth (32 measured slower at the layer shapes of the configs)

// FAST needs aligned operands; row-contiguous operands also need an extent % 4 == 0 so that
// clamped float4 groups stay inside the matrix
template <bool AKC, bool BKC>
static bool gemm_fast_ok(const GemmArgs& g) {
    return g.a_vec && g.b_vec && (AKC || (g.M % 4 == 0 && g.M >= 4)) &&
           (BKC || (g.N % 4 == 0 && g.N >= 4)) && tn_tune_tile() != 9;
}

// the 16-byte epilogue (and with it the inline dropout) needs 4-column groups that never straddle
// the matrix edge and aligned C / side operands
static bool gemm_cvec_ok(const GemmArgs& g) {
    const bool vec_on = true;
    auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
    return vec_on && g.ldc % 4 == 0 && g.N % 4 == 0 && g.N >= 4 && al(g.C, 15) && al(g.prev_a, 15) &&
           al(g.mask, 3) && al(g.drop_out, 3) && (g.elem0 & 3) == 0;
}

// DEEP pays while the 64 x 64 tiles cannot fill the chip and the L2 re-reads (M*N*K/4 bytes) stay small
static int tn_tune_deep() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TN_GEMM_DEEP");   // 0 off, 1 auto (default), 4 / 8: force that many waves
        v = e ? atoi(e) : 1;
    }
    return v;
}
template <bool BKC>
static bool gemm_deep_ok(tn_ctx* ctx, const GemmArgs& g) {
    if (!tn_tune_deep() || !g.a_vec || !g.b_vec || !gemm_cvec_ok(g) || g.K % 4 != 0 || g.K < 128) return false;
    if (tn_tune_deep() > 1) return true;
    const long long tiles64 = (long long)cdiv(g.M, 64) * cdiv(g.N, 64);
    return tiles64 <= ctx->num_cus && (long long)g.M * g.N * g.K / 4 <= (128ll << 20);
}
template <bool BKC>
static void launch_deep(tn_ctx* ctx, GemmArgs& g) {
    g.S = 1;
    g.c_vec = 1;
    g.MT = cdiv(g.M, 32);
    g.NT = cdiv(g.N, 32);
    const int grid = g.MT * g.NT;
    int nw = (grid <= 2 * ctx->num_cus && g.K >= 512) ? 8 : 4;
    if (tn_tune_deep() == 4 || tn_tune_deep() == 8) nw = tn_tune_deep();
    if (nw == 8)
        gemm_f32_deep<BKC, 8><<<grid, 512, 0, ctx->stream>>>(g);
    else
        gemm_f32_deep<BKC, 4><<<grid, 256, 0, ctx->stream>>>(g);
}

static unsigned long long* gemm_dbg_buf = nullptr;
extern "C" int tn_gemm_dbg_read(tn_ctx* ctx, unsigned long long* host, int nrec) {
    if (!gemm_dbg_buf) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(host, gemm_dbg_buf, (size_t)nrec * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
// DMA kernel: FAST's preconditions, every K slab holds a full tile, per-lane byte offsets fit 32 bits
static int tn_tune_dma() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TN_GEMM_DMA");    // 0: register-staged tiles (gemm_f32_fast); 2: the DMA kernel also where 128 x 64 tiles would be taken
        v = e ? atoi(e) : 1;
    }
    return v;
}
template <bool AKC, bool BKC>
static bool gemm_dma_ok(const GemmArgs& g, int S) {
    if (!tn_tune_dma() || !gemm_fast_ok<AKC, BKC>(g)) return false;
    const long long lastk = (long long)g.K - (long long)(S - 1) * g.kchunk;
    if (g.kchunk < 256 || lastk < 16) return false;      // short reductions (< 16 tiles) gain nothing from the ring: wide6's 128-row weight gradient measured 1.5 % of a step slower
    const long long ea = AKC ? (long long)g.M * g.lda : (long long)g.kchunk * g.lda + g.M;
    const long long eb = BKC ? (long long)g.N * g.ldb : (long long)g.kchunk * g.ldb + g.N;
    return ea * 4 < (1ll << 32) && eb * 4 < (1ll << 32);
}

template <bool AKC, bool BKC, bool BSUM>
static void launch_gemm(tn_ctx* ctx, GemmArgs& g, int S) {
    const bool fast = gemm_fast_ok<AKC, BKC>(g);
    bool big = (long long)cdiv(g.M, 128) * cdiv(g.N, 64) * S >= 2 * ctx->num_cus;
    if (tn_tune_tile() == 1) big = false;
    if (tn_tune_tile() == 2) big = true;
    if (!fast) big = false;
    g.c_vec = fast && gemm_cvec_ok(g);
    g.S = S;
    g.NT = cdiv(g.N, 64);
    g.MT = cdiv(g.M, big ? 128 : 64);
    const int grid = (S == 1) ? 8 * cdiv(g.MT, 8) * g.NT : gemm_grid_split(S, g.MT * g.NT);
    if (!fast)
        gemm_f32_generic<AKC, BKC, BSUM><<<grid, 256, 0, ctx->stream>>>(g);
    else if ((!big || tn_tune_dma() == 2) && gemm_dma_ok<AKC, BKC>(g, S)) {
        g.MT = cdiv(g.M, 64);
        const int nb = (S == 1) ? 8 * cdiv(g.MT, 8) * g.NT : gemm_grid_split(S, g.MT * g.NT);
        static int dbg_on = -1, pad = 0, nsf = 0;
        if (dbg_on < 0) {
            const char* e = getenv("TN_GEMM_DBG");
            dbg_on = e ? atoi(e) : 0;
            e = getenv("TN_GEMM_DMA_PAD");
            pad = e ? atoi(e) : 0;
            e = getenv("TN_GEMM_DMA_NS");
            nsf = e ? atoi(e) : 0;
        }
        const int grid = nb + ((BSUM && !BKC) ? S * g.NT : 0);
        // two blocks per CU or fewer: the deep ring; else four stages
        const int ns = nsf ? nsf : (nb <= 2 * ctx->num_cus ? 8 : 4);
        if (dbg_on) {
            // (cycle stamps: 8 words per wave, 4 waves per block, 65536 records)
            // (this launcher returns nothing: a stamp buffer that cannot be had, or a grid beyond its 65536 records,
            // just runs unstamped -- g.dbg stays NULL and the kernel writes no stamps)
            if (!gemm_dbg_buf && hipMalloc(&gemm_dbg_buf, 8 * sizeof(unsigned long long) * 65536) != hipSuccess) gemm_dbg_buf = nullptr;
            if (gemm_dbg_buf && grid * 4 <= 65536 &&
                hipMemsetAsync(gemm_dbg_buf, 0, 8 * sizeof(unsigned long long) * 65536, ctx->stream) == hipSuccess)
                g.dbg = gemm_dbg_buf;
            if (ns == 8) gemm_f32_dma<AKC, BKC, 8, true><<<grid, 256, pad, ctx->stream>>>(g, nb);
            else if (ns == 2) gemm_f32_dma<AKC, BKC, 2, true><<<grid, 256, pad, ctx->stream>>>(g, nb);
            else gemm_f32_dma<AKC, BKC, 4, true><<<grid, 256, pad, ctx->stream>>>(g, nb);
        } else if (ns == 8)
            gemm_f32_dma<AKC, BKC, 8><<<grid, 256, pad, ctx->stream>>>(g, nb);
        else if (ns == 2)
            gemm_f32_dma<AKC, BKC, 2><<<grid, 256, pad, ctx->stream>>>(g, nb);
        else
            gemm_f32_dma<AKC, BKC, 4><<<grid, 256, pad, ctx->stream>>>(g, nb);
    } else if (big)
        gemm_f32_fast<AKC, BKC, BSUM, 2, 1, 16><<<grid, 256, 0, ctx->stream>>>(g);
    else if (tn_tune_bk() == 32)
        gemm_f32_fast<AKC, BKC, BSUM, 1, 1, 32><<<grid, 256, 0, ctx->stream>>>(g);
    else
        gemm_f32_fast<AKC, BKC, BSUM, 1, 1, 16><<<grid, 256, 0, ctx->stream>>>(g);
}

// prepares g for the 64 x 64 fast kernel; returns its grid size (0: not eligible)
template <bool AKC, bool BKC>
static int gemm_setup_small(GemmArgs& g, int S) {
    if (!gemm_fast_ok<AKC, BKC>(g)) return 0;
    g.c_vec = gemm_cvec_ok(g);
    g.S = S;
    g.NT = cdiv(g.N, 64);
    g.MT = cdiv(g.M, 64);
    return (S == 1) ? 8 * cdiv(g.MT, 8) * g.NT : gemm_grid_split(S, g.MT * g.NT);
}

static int wgrad_splits(int B, int n_in, int n_out) {
    // K-slabs (1, 2, 4 or 8; every slab is written here and read back by the update): as few as give four 64 x 64 blocks
    // per CU, each with >= 8 K-tiles.  mnist.prms (96 tiles): 8; cifar_like (256 tiles): 4 (step 1.4038 ms with 8 slabs,
    // 1.3890 with 4, 1.3980 with 2, same box).
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("TN_FC_WSPLIT");
        force = e ? atoi(e) : 0;
    }
    if (force == 1 || force == 2 || force == 4 || force == 8) return B >= force * 8 * BK ? force : 1;
    const long long tiles = (long long)cdiv(n_in, 64) * cdiv(n_out, 64);
    int S = 1;
    while (S < 8 && tiles * S < 1024 && B >= 2 * S * 8 * BK) S *= 2;
    if (B < 8 * 8 * BK) return 1;
    return S;
}

// =====================================================================================
// skinny layers: n_out <= 16
// =====================================================================================
#define SK_MAX 16

template <int CTRL>
__device__ __forceinline__ float gdpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over each aligned group of 32 lanes (result valid in every lane of the group)
__device__ __forceinline__ float half_allsum(float v) {
    v += gdpp<0xB1>(v);     // quad xor 1
    v += gdpp<0x4E>(v);     // quad xor 2
    v += gdpp<0x141>(v);    // row_half_mirror
    v += gdpp<0x140>(v);    // row_mirror: 16-lane row sums
    v += __shfl_xor(v, 16, 64);
    return v;
}

// forward: block = 8 rows x 32 k-lanes.  Every lane owns k = kl, kl+32, ... : its x values are
// loaded up front (all in flight), W sits in LDS with row stride n_out+1 (conflict-free), the
// n_out partial sums are reduced over the 32 lanes with DPP.
#define SK_XPL 16     // x values per lane and pass (covers n_in <= 512 in one pass)
__global__ __launch_bounds__(256) void fc_skinny_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    float* __restrict__ a, int B, int n_in, int n_out, int act, float prm,
    const uint8_t* __restrict__ mask) {
    extern __shared__ float sW[];       // [n_in][n_out+1]
    const int ldw = n_out + 1;
    const int kl = threadIdx.x & 31;
    const int row = min(blockIdx.x * 8 + (threadIdx.x >> 5), B - 1);
    const float* xr = x + (size_t)row * n_in;
    float xv[SK_XPL];
#pragma unroll
    for (int i = 0; i < SK_XPL; ++i) xv[i] = xr[min(kl + 32 * i, n_in - 1)];   // in flight with W
    for (int t = threadIdx.x; t < n_in * n_out; t += 256) {
        const int k = t / n_out;
        sW[t + k] = W[t];                                  // k*ldw + n == t + k
    }
    __syncthreads();
    float acc[SK_MAX];
#pragma unroll
    for (int n = 0; n < SK_MAX; ++n) acc[n] = 0.f;
    for (int k0 = 0; k0 < n_in; k0 += 32 * SK_XPL) {
        if (k0 > 0) {
#pragma unroll
            for (int i = 0; i < SK_XPL; ++i) xv[i] = xr[min(k0 + kl + 32 * i, n_in - 1)];
        }
#pragma unroll
        for (int i = 0; i < SK_XPL; ++i) {
            const int k = k0 + kl + 32 * i;
            const float xx = (k < n_in) ? xv[i] : 0.f;
            const float* wr = sW + min(k, n_in - 1) * ldw;
#pragma unroll
            for (int n = 0; n < SK_MAX; ++n)
                if (n < n_out) acc[n] = fmaf(xx, wr[n], acc[n]);
        }
    }
    float mine = 0.f;
#pragma unroll
    for (int n = 0; n < SK_MAX; ++n) {
        if (n < n_out) {       // block-uniform
            const float s = half_allsum(acc[n]);
            if (kl == n) mine = s;
        }
    }
    if (kl < n_out && blockIdx.x * 8 + (threadIdx.x >> 5) < B) {
        const size_t o = (size_t)row * n_out + kl;
        float v = tn_act_fwd(mine + (b ? b[kl] : 0.f), act, prm);
        if (mask) v *= (float)mask[o];
        a[o] = v;
    }
}

// dgrad: thread = one input feature k for 16 rows: side loads (prev_a, mask) are all issued
// first, the rows' dz values are wave-uniform scalar loads.
__global__ __launch_bounds__(256) void fc_skinny_dgrad_kernel(
    const float* __restrict__ dz, const float* __restrict__ W, float* __restrict__ dx, int B, int n_in,
    int n_out, const float* __restrict__ prev_a, int act, float prm, const uint8_t* __restrict__ mask) {
    const int k = min(blockIdx.x * 256 + threadIdx.x, n_in - 1);
    const bool live = blockIdx.x * 256 + threadIdx.x < n_in;
    const int row0 = blockIdx.y * 16;
    float w[SK_MAX];
#pragma unroll
    for (int n = 0; n < SK_MAX; ++n) w[n] = (n < n_out) ? W[(size_t)k * n_out + n] : 0.f;
    float pa[16], pm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t o = (size_t)min(row0 + r, B - 1) * n_in + k;
        pa[r] = prev_a ? prev_a[o] : 0.f;
        pm[r] = mask ? (float)mask[o] : 1.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + r;
        const float* dzr = dz + (size_t)min(row, B - 1) * n_out;     // wave-uniform -> scalar loads
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n)
            if (n < n_out) s = fmaf(dzr[n], w[n], s);
        if (prev_a) s *= tn_act_grad_from_out(pa[r], act, prm);
        s *= pm[r];
        if (live && row < B) dx[(size_t)row * n_in + k] = s;
    }
}

// wgrad: thread = one input feature k, block.y = a chunk of SK_WROWS rows.  The chunk's dz rows
// are staged in LDS (read back as wave broadcasts), the x values are loaded 32 at a time (all
// in flight).  partial[chunk][k][n], dbpartial[chunk][n].
#define SK_WROWS 64
__global__ __launch_bounds__(256) void fc_skinny_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    float* __restrict__ dbpartial, int B, int n_in, int n_out) {
    __shared__ float sdz[SK_WROWS * SK_MAX];
    const int k = min(blockIdx.x * 256 + threadIdx.x, n_in - 1);
    const bool live = blockIdx.x * 256 + threadIdx.x < n_in;
    const int row0 = blockIdx.y * SK_WROWS;
    const int nrow = min(SK_WROWS, B - row0);
    for (int t = threadIdx.x; t < SK_WROWS * SK_MAX; t += 256) {
        const int r = t / SK_MAX, n = t - r * SK_MAX;
        sdz[t] = (r < nrow && n < n_out) ? dz[(size_t)(row0 + r) * n_out + n] : 0.f;
    }
    __syncthreads();
    float acc[SK_MAX];
#pragma unroll
    for (int n = 0; n < SK_MAX; ++n) acc[n] = 0.f;
#pragma unroll
    for (int half = 0; half < SK_WROWS / 32; ++half) {
        float xv[32];
#pragma unroll
        for (int r = 0; r < 32; ++r)
            xv[r] = x[(size_t)min(row0 + 32 * half + r, B - 1) * n_in + k];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const float* dr = sdz + (32 * half + r) * SK_MAX;       // zero rows beyond nrow
#pragma unroll
            for (int n = 0; n < SK_MAX; ++n)
                if (n < n_out) acc[n] = fmaf(xv[r], dr[n], acc[n]);
        }
    }
    if (live) {
        float* p = partial + ((size_t)blockIdx.y * n_in + k) * n_out;
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n)
            if (n < n_out) p[n] = acc[n];
    }
    if (blockIdx.x == 0 && threadIdx.x < n_out) {
        float sb = 0.f;
        for (int r = 0; r < SK_WROWS; ++r) sb += sdz[r * SK_MAX + threadIdx.x];
        dbpartial[(size_t)blockIdx.y * n_out + threadIdx.x] = sb;
    }
}


// fc_skinny.hip: 16-byte-access kernels for n_out <= 16
bool tn_fc_skinny_ok(int n_in, int n_out, const void* p0, const void* p1, const void* p2);
int tn_fc_skinny_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B,
                     int n_in, int n_out, int act, float prm, const uint8_t* mask);
int tn_fc_skinny_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B,
                       int n_in, int n_out, float* ws);
int tn_fc_skinny_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in,
                       int n_out, const float* prev_a, int act, float prm, const uint8_t* mask);

int tn_fc_skinny_bwd(tn_ctx* ctx, const float* x, const float* dz, const float* W, float* dW, float* db,
                     float* dx, int B, int n_in, int n_out, float* ws, const float* prev_a, int act,
                     float prm, const uint8_t* mask);
int tn_fc_skinny_softmax_train(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits,
                               int B, int n_in, int n_out, const int32_t* y, int64_t y_row0,
                               const int64_t* d_row0, float* logprob, float* rowloss, int32_t* pred,
                               float* rowp, float* dz, float inv_batch, float* dW, float* db, float* dx,
                               float* ws, int fuse_act, int act, float prm, const uint8_t* mask);
int tn_fc_skinny_softmax(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits,
                         int B, int n_in, int n_out, const int32_t* y, int64_t y_row0,
                         const int64_t* d_row0, float* logprob, float* rowloss, int32_t* pred,
                         float* rowp, float* dz, float inv_batch);

extern "C" {

int tn_fc_softmax_train(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits, int B,
                        int n_in, int n_out, const int32_t* y, int64_t y_row0, const int64_t* d_row0,
                        float* logprob, float* rowloss, int32_t* pred, float* rowp, float* dz,
                        float inv_batch, float* dW, float* db, float* dx, void* ws, const float* prev_a,
                        int prev_act, float prev_act_param, const uint8_t* prev_mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0 && logits && logprob && y && dz && dW && db && dx && ws,
               "tn_fc_softmax_train: bad arguments");
    TN_REQUIRE(prev_a == nullptr || prev_a == x, "tn_fc_softmax_train: prev_a must be the layer input");
    static int fused_on = -1;
    if (fused_on < 0) {
        const char* e = getenv("TN_SOFTMAX_TRAIN");
        fused_on = e ? atoi(e) : 1;
    }
    if (fused_on && tn_fc_skinny_ok(n_in, n_out, x, dx, prev_mask))
        return tn_fc_skinny_softmax_train(ctx, x, W, b, logits, B, n_in, n_out, y, y_row0, d_row0, logprob,
                                          rowloss, pred, rowp, dz, inv_batch, dW, db, dx, (float*)ws,
                                          prev_a != nullptr, prev_act, prev_act_param, prev_mask);
    int rc = tn_fc_softmax_nll(ctx, x, W, b, logits, B, n_in, n_out, y, y_row0, d_row0, logprob, rowloss,
                               pred, rowp, dz, inv_batch);
    if (rc) return rc;
    return tn_fc_bwd(ctx, x, dz, W, dW, db, dx, B, n_in, n_out, ws, prev_a, prev_act, prev_act_param,
                     prev_mask);
}

int tn_fc_softmax_nll(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits, int B,
                      int n_in, int n_out, const int32_t* y, int64_t y_row0, const int64_t* d_row0,
                      float* logprob, float* rowloss, int32_t* pred, float* rowp, float* dz,
                      float inv_batch) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0 && logits && logprob, "tn_fc_softmax_nll: bad arguments");
    TN_REQUIRE(y != nullptr || (rowloss == nullptr && dz == nullptr && rowp == nullptr),
               "tn_fc_softmax_nll: labels required for loss/gradient outputs");
    if (tn_fc_skinny_ok(n_in, n_out, x, nullptr, nullptr))
        return tn_fc_skinny_softmax(ctx, x, W, b, logits, B, n_in, n_out, y, y_row0, d_row0, logprob,
                                    rowloss, pred, rowp, dz, inv_batch);
    int rc = tn_fc_fwd(ctx, x, W, b, logits, B, n_in, n_out, TN_ACT_LINEAR, 0.f, nullptr);
    if (rc) return rc;
    return tn_softmax_nll(ctx, logits, y, y_row0, d_row0, logprob, rowloss, pred, rowp, dz, B, n_out,
                          inv_batch);
}

// ---- short-and-deep forward products (wide6's 128 x 16384 x 1024): too few output tiles to fill
// the chip, so the reduction is split into S slabs (plain partial products) and a finishing kernel
// adds the slabs in order and applies bias + activation (+ dropout mask).
__global__ __launch_bounds__(256) void fc_fwd_finish_kernel(const float* __restrict__ ws, int S, size_t MN,
                                                           int n_out, const float* __restrict__ bias,
                                                           const uint8_t* __restrict__ mask,
                                                           float* __restrict__ out, int act, float prm) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float v = 0.f;
    for (int z = 0; z < S; ++z) v += ws[(size_t)z * MN + i];
    v = tn_act_fwd(v + bias[i % n_out], act, prm);
    if (mask) v = mask[i] ? v : 0.f;
    out[i] = v;
}

static int fc_fwd_splits(tn_ctx* ctx, int B, int n_in, int n_out) {
    const long long tiles = (long long)cdiv(B, 64) * cdiv(n_out, 64);
    if (tiles * 4 > ctx->num_cus || n_in < 64 * BK) return 1;
    int S = (int)(2 * ctx->num_cus / tiles);
    if (S > n_in / (8 * BK)) S = n_in / (8 * BK);
    if (S >= 8) S &= ~7;
    return S < 2 ? 1 : S;
}

// g: the forward GemmArgs (EPI_FWD); runs it as S partial products + the finishing kernel
static int fc_fwd_splitk(tn_ctx* ctx, GemmArgs g, int S, const uint8_t* mask) {
    const size_t MN = (size_t)g.M * g.N;
    float* out = g.C;
    g.kchunk = cdiv(cdiv(g.K, S), BK) * BK;
    const int Sx = cdiv(g.K, g.kchunk);
    float* ws;
    int rc = tn_scratch_get(ctx, (size_t)Sx * MN * sizeof(float), &ws);
    if (rc) return rc;
    g.C = ws; g.epi = EPI_PLAIN; g.mask = nullptr; g.drop_out = nullptr;
    const float* bias = g.bias;
    g.bias = nullptr;
    launch_gemm<true, false, false>(ctx, g, Sx);
    TN_LAUNCH_CHECK();
    fc_fwd_finish_kernel<<<cdiv(MN, 256), 256, 0, ctx->stream>>>(ws, Sx, MN, g.N, bias, mask, out, g.act, g.act_prm);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

__global__ __launch_bounds__(256) void fc_dgrad_finish_kernel(const float* __restrict__ ws, int S, size_t MN,
                                                             const float* __restrict__ prev_a,
                                                             const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                             int act, float prm) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float v = 0.f;
    for (int z = 0; z < S; ++z) v += ws[(size_t)z * MN + i];
    if (prev_a) v *= tn_act_grad_from_out(prev_a[i], act, prm);
    if (mask) v = mask[i] ? v : 0.f;
    out[i] = v;
}

static int fc_dgrad_splits(tn_ctx* ctx, int B, int n_in, int n_out) {
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("TN_FC_DGRAD_SPLIT");
        force = e ? atoi(e) : 0;
    }
    if (force > 0) return force;
    if (B > 256 || n_out < 32 * BK || n_in < 2048) return 1;
    int S = n_out / (8 * BK);                  // >= 8 K-tiles per slab
    if (S > 8) S = 8;
    return S < 2 ? 1 : S;
}

int tn_fc_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B, int n_in,
              int n_out, int act, float act_param, const uint8_t* mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_fwd: bad shape");
    if (ctx->fc_b3 && tn_b3_fc_ok(x, W, B, n_in, n_out))
        return tn_b3_fc_fwd(ctx, x, W, b, a, B, n_in, n_out, act, act_param, mask);
    if (tn_fc_skinny_ok(n_in, n_out, x, nullptr, nullptr))
        return tn_fc_skinny_fwd(ctx, x, W, b, a, B, n_in, n_out, act, act_param, mask);
    const size_t sk_lds = (size_t)n_in * (n_out + 1) * sizeof(float);
    if (n_out <= SK_MAX && sk_lds <= 60 * 1024) {
        fc_skinny_fwd_kernel<<<cdiv(B, 8), 256, sk_lds, ctx->stream>>>(
            x, W, b, a, B, n_in, n_out, act, act_param, mask);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    GemmArgs g{};
    g.A = x; g.B = W; g.C = a;
    g.M = B; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out;
    g.kchunk = cdiv(n_in, BK) * BK;
    g.epi = EPI_FWD; g.bias = b; g.mask = mask; g.act = act; g.act_prm = act_param;
    g.a_vec = vec_ok(x, n_in); g.b_vec = vec_ok(W, n_out);
    const int S = fc_fwd_splits(ctx, B, n_in, n_out);
    if (S > 1) return fc_fwd_splitk(ctx, g, S, mask);
    if (gemm_deep_ok<false>(ctx, g))
        launch_deep<false>(ctx, g);
    else
        launch_gemm<true, false, false>(ctx, g, 1);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_fc_fwd_dropout(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B,
                      int n_in, int n_out, int act, float act_param, uint8_t* mask_out, float pdrop,
                      uint64_t seed, uint32_t step, const uint32_t* d_step, uint64_t elem0) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0 && mask_out != nullptr, "tn_fc_fwd_dropout: bad arguments");
    GemmArgs g{};
    g.A = x; g.B = W; g.C = a;
    g.M = B; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out;
    g.kchunk = cdiv(n_in, BK) * BK;
    g.epi = EPI_FWD; g.bias = b; g.act = act; g.act_prm = act_param;
    g.a_vec = vec_ok(x, n_in); g.b_vec = vec_ok(W, n_out);
    g.drop_out = mask_out; g.pdrop = pdrop; g.dk0 = (uint32_t)seed; g.dk1 = (uint32_t)(seed >> 32);
    g.dstep = step; g.d_step = d_step; g.elem0 = elem0;
    if (!(ctx->fc_b3 && tn_b3_fc_ok(x, W, B, n_in, n_out)) && n_out > SK_MAX && fc_fwd_splits(ctx, B, n_in, n_out) == 1 && gemm_fast_ok<true, false>(g) &&
        gemm_cvec_ok(g)) {
        if (gemm_deep_ok<false>(ctx, g))
            launch_deep<false>(ctx, g);
        else
            launch_gemm<true, false, false>(ctx, g, 1);  // mask drawn in the epilogue
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    int rc = tn_dropout_mask(ctx, mask_out, (size_t)B * n_out, pdrop, seed, step, d_step, elem0);
    if (rc) return rc;
    return tn_fc_fwd(ctx, x, W, b, a, B, n_in, n_out, act, act_param, mask_out);
}

size_t tn_fc_wgrad_ws_bytes(int B, int n_in, int n_out) {
    if (n_out <= SK_MAX) {
        const int chunks = cdiv(B, SK_WROWS);
        const size_t a = ((size_t)chunks * n_in * n_out + (size_t)chunks * n_out) * sizeof(float) + 64;
        const size_t t = (size_t)cdiv(B, B < 2048 ? 4 : 16) * (n_in + 1) * n_out * sizeof(float) + 64;   // softmax_train slabs (fc_skinny.hip sk_train_rb)
        return a > t ? a : t;
    }
    const int S = wgrad_splits(B, n_in, n_out);
    return ((size_t)S * n_in * n_out + (size_t)S * n_out) * sizeof(float) + 64;
}

int tn_fc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B, int n_in,
                int n_out, void* ws) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0 && ws != nullptr, "tn_fc_wgrad: bad arguments");
    if (ctx->fc_b3 && tn_b3_fc_ok(x, dz, B, n_in, n_out))
        return tn_b3_fc_wgrad(ctx, x, dz, dW, db, B, n_in, n_out, (float*)ws, wgrad_splits(B, n_in, n_out));
    if (tn_fc_skinny_ok(n_in, n_out, x, nullptr, nullptr))
        return tn_fc_skinny_wgrad(ctx, x, dz, dW, db, B, n_in, n_out, (float*)ws);
    if (n_out <= SK_MAX) {
        const int chunks = cdiv(B, SK_WROWS);
        float* wsC = (float*)ws;
        float* wsB = wsC + (size_t)chunks * n_in * n_out;
        fc_skinny_wgrad_kernel<<<dim3(cdiv(n_in, 256), chunks), 256, 0, ctx->stream>>>(
            x, dz, wsC, wsB, B, n_in, n_out);
        TN_LAUNCH_CHECK();
        const int MN = n_in * n_out;
        int rc = tn_red_push(ctx, wsC, dW, (uint32_t)MN, (uint32_t)chunks, (uint32_t)MN, 0);
        if (rc) return rc;
        rc = tn_red_push(ctx, wsB, db, (uint32_t)n_out, (uint32_t)chunks, (uint32_t)n_out, 0);
        if (rc) return rc;
        return tn_red_commit(ctx);
    }
    const int S = wgrad_splits(B, n_in, n_out);
    float* wsC = (float*)ws;
    float* wsB = wsC + (size_t)S * n_in * n_out;
    GemmArgs g{};
    g.A = x; g.B = dz;
    g.M = n_in; g.N = n_out; g.K = B;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out;
    g.kchunk = cdiv(cdiv(B, S), BK) * BK;
    g.epi = EPI_PLAIN;
    g.a_vec = vec_ok(x, n_in); g.b_vec = vec_ok(dz, n_out);
    const int Sx = cdiv(B, g.kchunk);
    if (Sx == 1) {
        g.C = dW; g.colsum = db;
    } else {
        g.C = wsC; g.colsum = wsB;
    }
    launch_gemm<false, false, true>(ctx, g, Sx);
    TN_LAUNCH_CHECK();
    if (Sx > 1) {
        const size_t MN = (size_t)n_in * n_out;
        int blocks = cdiv(cdiv(MN, 4), 256);
        if (blocks < cdiv(n_out, 256)) blocks = cdiv(n_out, 256);
        (void)blocks;
        int rc = tn_red_push(ctx, wsC, dW, (uint32_t)MN, (uint32_t)Sx, (uint32_t)MN, 0);
        if (rc) return rc;
        rc = tn_red_push(ctx, wsB, db, (uint32_t)n_out, (uint32_t)Sx, (uint32_t)n_out, 0);
        if (rc) return rc;
        return tn_red_commit(ctx);
    }
    return TN_OK;
}

int tn_fc_bwd(tn_ctx* ctx, const float* x, const float* dz, const float* W, float* dW, float* db,
              float* dx, int B, int n_in, int n_out, void* ws, const float* prev_a, int prev_act,
              float prev_act_param, const uint8_t* prev_mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0 && ws != nullptr && dx != nullptr, "tn_fc_bwd: bad arguments");
    if (ctx->fc_b3 && tn_b3_fc_ok(x, W, B, n_in, n_out) && tn_b3_fc_ok(dz, W, B, n_in, n_out)) {
        int rc = tn_b3_fc_wgrad(ctx, x, dz, dW, db, B, n_in, n_out, (float*)ws, wgrad_splits(B, n_in, n_out));
        if (rc) return rc;
        return tn_b3_fc_dgrad(ctx, dz, W, dx, B, n_in, n_out, prev_a, prev_act, prev_act_param, prev_mask);
    }
    if (tn_fc_skinny_ok(n_in, n_out, x, nullptr, nullptr) &&
        tn_fc_skinny_ok(n_in, n_out, dx, prev_a, prev_mask))
        return tn_fc_skinny_bwd(ctx, x, dz, W, dW, db, dx, B, n_in, n_out, (float*)ws, prev_a, prev_act,
                                prev_act_param, prev_mask);
    if (n_out > SK_MAX && fc_dgrad_splits(ctx, B, n_in, n_out) == 1) {
        // weight gradient (split-K slabs) and input gradient as ONE launch of interleaved blocks
        const int S = wgrad_splits(B, n_in, n_out);
        float* wsC = (float*)ws;
        float* wsB = wsC + (size_t)S * n_in * n_out;
        GemmArgs g1{}, g2{};
        g1.A = x; g1.B = dz;
        g1.M = n_in; g1.N = n_out; g1.K = B;
        g1.lda = n_in; g1.ldb = n_out; g1.ldc = n_out;
        g1.kchunk = cdiv(cdiv(B, S), BK) * BK;
        g1.epi = EPI_PLAIN;
        g1.a_vec = vec_ok(x, n_in); g1.b_vec = vec_ok(dz, n_out);
        const int Sx = cdiv(B, g1.kchunk);
        if (Sx == 1) {
            g1.C = dW; g1.colsum = db;
        } else {
            g1.C = wsC; g1.colsum = wsB;
        }
        g2.A = dz; g2.B = W; g2.C = dx;
        g2.M = B; g2.N = n_in; g2.K = n_out;
        g2.lda = n_out; g2.ldb = n_out; g2.ldc = n_in;
        g2.kchunk = cdiv(n_out, BK) * BK;
        g2.epi = EPI_DGRAD; g2.prev_a = prev_a; g2.mask = prev_mask; g2.act = prev_act;
        g2.act_prm = prev_act_param;
        g2.a_vec = vec_ok(dz, n_out); g2.b_vec = vec_ok(W, n_out);
        const int n1 = gemm_setup_small<false, false>(g1, Sx), n2 = gemm_setup_small<true, true>(g2, 1);
        if (n1 > 0 && n2 > 0) {
            int nrider = 0;
            size_t rlds = 0;
            ElField rider{};
            static int pns = -1;
            if (pns < 0) {
                const char* e = getenv("TN_PAIR_DMA");      // ring stages * 10 + waves per SIMD (experiments)
                pns = e ? atoi(e) : 26;
            }
            const bool use_dma = gemm_dma_ok<false, false>(g1, Sx) && gemm_dma_ok<true, true>(g2, 1);
            // the rider works in the STATIC LDS of the kernel that is launched: the DMA form declares its ring
            // (gemm_dma_smem_floats<NS>: 17 408 bytes at two stages), the register form 20 480 bytes -- a field that needs
            // more than the chosen kernel has runs standalone (tn_elastic_field) instead of riding
            const size_t rider_cap = sizeof(float) * (!use_dma ? (size_t)gemm_smem_floats<1, 1, 16>()
                                                      : pns / 10 >= 8 ? (size_t)gemm_dma_smem_floats<8>()
                                                      : pns / 10 >= 4 ? (size_t)gemm_dma_smem_floats<4>()
                                                                      : (size_t)gemm_dma_smem_floats<2>());
            if (ctx->rider_valid && ctx->rider_lds <= rider_cap) {
                rider = ctx->rider;
                nrider = cdiv(rider.h * rider.w, 4);
                rlds = 0;                                  // the rider works in the tile's static LDS
                ctx->rider_valid = false;
            }
            {
                // Residency cap while two steps are in flight: 8 KB of unused dynamic LDS per block = five blocks per
                // CU instead of six.  Alone on the GPU six are faster (60.2 vs 63.2 us); with two steps in flight the
                // launch shares the chip with the other stream's forward kernels, and five blocks x 80 registers
                // leave them a fifth of the register file (six leave 32 registers per lane: nothing else fits and the
                // streams take turns): 176-177 vs 180 us per step.  "In flight" = the second stream was selected
                // within the last few heavy launches (tn_stream_select); TN_PAIR_LDS_PAD=n forces a pad for an A/B.
                static int pad = -2;
                if (pad == -2) {
                    const char* e = getenv("TN_PAIR_LDS_PAD");
                    pad = e ? atoi(e) : -1;
                }
                if (pad >= 0)
                    rlds += (size_t)pad;
                else if (ctx->heavy_since_side < 3) {
                    ctx->heavy_since_side++;
                    // (launches with many block rounds per CU gain nothing from sharing: cifar_like's 3072-block
                    // fc backward is 0.7 % of a step slower with the pad)
                    if (n1 + n2 <= 8 * ctx->num_cus) rlds += 8192;
                }
            }
            if (use_dma) {
                const int ncs = Sx * g1.NT;
                const int grid = n1 + n2 + ncs + nrider;
#define TN_PAIR_CASE(NS_, W_) case NS_ * 10 + W_: gemm_f32_pair_dma<false, false, true, true, NS_, W_><<<grid, 256, rlds, ctx->stream>>>(g1, g2, n1, n2, ncs, rider); break;
                switch (pns) {
                    TN_PAIR_CASE(2, 6) TN_PAIR_CASE(2, 5) TN_PAIR_CASE(2, 4) TN_PAIR_CASE(4, 5) TN_PAIR_CASE(4, 4) TN_PAIR_CASE(4, 3)
                    TN_PAIR_CASE(8, 2)
                    default: return tn_fail(ctx, TN_E_ARG, "TN_PAIR_DMA: unknown variant %d", pns);
                }
#undef TN_PAIR_CASE
            } else
                gemm_f32_pair<false, false, true, true, true, false><<<n1 + n2 + nrider, 256, rlds, ctx->stream>>>(
                    g1, g2, n1, n2, rider);
            TN_LAUNCH_CHECK();
            if (Sx > 1) {
                const size_t MN = (size_t)n_in * n_out;
                int rc = tn_red_push(ctx, wsC, dW, (uint32_t)MN, (uint32_t)Sx, (uint32_t)MN, 0);
                if (rc) return rc;
                rc = tn_red_push(ctx, wsB, db, (uint32_t)n_out, (uint32_t)Sx, (uint32_t)n_out, 0);
                if (rc) return rc;
                return tn_red_commit(ctx);
            }
            return TN_OK;
        }
    }
    int rc = tn_fc_wgrad(ctx, x, dz, dW, db, B, n_in, n_out, ws);
    if (rc) return rc;
    return tn_fc_dgrad(ctx, dz, W, dx, B, n_in, n_out, prev_a, prev_act, prev_act_param, prev_mask);
}

int tn_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in, int n_out,
                const float* prev_a, int prev_act, float prev_act_param, const uint8_t* prev_mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_dgrad: bad shape");
    if (ctx->fc_b3 && tn_b3_fc_ok(dz, W, B, n_in, n_out))
        return tn_b3_fc_dgrad(ctx, dz, W, dx, B, n_in, n_out, prev_a, prev_act, prev_act_param, prev_mask);
    if (tn_fc_skinny_ok(n_in, n_out, dx, prev_a, prev_mask))
        return tn_fc_skinny_dgrad(ctx, dz, W, dx, B, n_in, n_out, prev_a, prev_act, prev_act_param,
                                  prev_mask);
    if (n_out <= SK_MAX) {
        fc_skinny_dgrad_kernel<<<dim3(cdiv(n_in, 256), cdiv(B, 16)), 256, 0, ctx->stream>>>(
            dz, W, dx, B, n_in, n_out, prev_a, prev_act, prev_act_param, prev_mask);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    GemmArgs g{};
    g.A = dz; g.B = W; g.C = dx;
    g.M = B; g.N = n_in; g.K = n_out;
    g.lda = n_out; g.ldb = n_out; g.ldc = n_in;     // B(k,n) = W[n*n_out + k]: k-contiguous
    g.kchunk = cdiv(n_out, BK) * BK;
    g.epi = EPI_DGRAD; g.prev_a = prev_a; g.mask = prev_mask; g.act = prev_act; g.act_prm = prev_act_param;
    g.a_vec = vec_ok(dz, n_out); g.b_vec = vec_ok(W, n_out);
    const int S = fc_dgrad_splits(ctx, B, n_in, n_out);
    if (S > 1) {
        // few rows, a long reduction and a big weight matrix (wide6's 128 x 16384 <- 1024): a block per output
        // tile is a long latency-bound chain over its 64 weight rows; S slabs of the reduction put S times
        // as many blocks in flight, a finishing kernel adds them in order and applies act' * mask
        const size_t MN = (size_t)B * n_in;
        g.kchunk = cdiv(cdiv(n_out, S), BK) * BK;
        const int Sx = cdiv(n_out, g.kchunk);
        float* ws;
        int rc = tn_scratch_get(ctx, (size_t)Sx * MN * sizeof(float), &ws);
        if (rc) return rc;
        g.C = ws; g.epi = EPI_PLAIN; g.prev_a = nullptr; g.mask = nullptr;
        launch_gemm<true, true, false>(ctx, g, Sx);
        TN_LAUNCH_CHECK();
        fc_dgrad_finish_kernel<<<cdiv(MN, 256), 256, 0, ctx->stream>>>(ws, Sx, MN, prev_a, prev_mask, dx, prev_act,
                                                                      prev_act_param);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    launch_gemm<true, true, false>(ctx, g, 1);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
