// fp32 MFMA GEMM for the fully-connected layers (theanet/layer/hidden.py:30, layer.py:83).
//
//   C[M,N] = op(A)[M,K] . op(B)[K,N]     v_mfma_f32_32x32x2_f32 (exact fp32, fmaf chain)
//
// Block tile (64*WM) x (64*WN), BK=16, 4 waves in a 2x2 arrangement, each wave owning
// WM x WN accumulators of 32x32 (so one A/B fragment read feeds WN/WM MFMAs).  Double-buffered
// LDS with a register-staged prefetch (global loads of tile t+1 are in flight while tile t is
// multiplied), one barrier per K-tile.  Operands may be row- or column-contiguous (NN / NT / TN)
// so that forward, dgrad (dz.W^T) and wgrad (x^T.dz) all run on the same kernel; wgrad uses
// split-K over the batch dimension (M x N is small, K = batch is long) with a deterministic
// slab reduce, and picks up the bias gradient (column sums of dz) from the tiles it stages.
// Epilogues fuse bias + activation + dropout mask (forward) and activation-gradient + mask
// (dgrad), so no elementwise pass touches HBM again.
//
// Skinny layers (n_out <= 16, the 10-way softmax layer) would waste >2/3 of a 32-wide MFMA
// tile and leave most CUs idle; they run on dedicated VALU kernels at the end of this file.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 16
#define LDS_PAD 4   // rows stay 16-byte aligned for b128 stores; fragment reads conflict-free

enum { EPI_PLAIN = 0, EPI_FWD = 1, EPI_DGRAD = 2 };

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;          // or split-K workspace
    int M, N, K;
    int lda, ldb, ldc;
    int kchunk;        // K range per blockIdx.z (multiple of BK)
    int epi;
    const float* bias;       // EPI_FWD
    const float* prev_a;     // EPI_DGRAD: output of the layer below (same shape as C)
    const uint8_t* mask;     // EPI_FWD / EPI_DGRAD (may be NULL)
    int act;
    float act_prm;
    float* colsum;     // BSUM: [gridDim.z][N] partial column sums of B
    int a_vec, b_vec;  // 16-byte vector loads allowed (ld % 4 == 0 and base aligned)
};

// ---- tile loaders: global -> 4 registers -------------------------------------------------
// KC: source is k-contiguous: element (r, k) at src[r*ld + k]; thread -> row r = t>>2, k = 4*(t&3)..+3
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
// RC: source is row(mn)-contiguous: element (r, k) at src[k*ld + r]; thread -> k = t>>4, r = 4*(t&15)..+3
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

template <bool AKC, bool BKC, bool BSUM, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LDA = BM + LDS_PAD, LDB = BN + LDS_PAD;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    // per-thread staging coordinates inside a 64-row slice
    const int a_r = AKC ? (t >> 2) : 4 * (t & 15);
    const int a_k = AKC ? 4 * (t & 3) : (t >> 4);
    const int b_r = BKC ? (t >> 2) : 4 * (t & 15);
    const int b_k = BKC ? 4 * (t & 3) : (t >> 4);

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

    float4 ra[WM], rb[WN];
    auto gload = [&](int tile) {
        const int k0 = kbeg + tile * BK;
#pragma unroll
        for (int i = 0; i < WM; ++i)
            ra[i] = AKC ? load_kc(g.A, g.lda, m0 + 64 * i + a_r, g.M, k0 + a_k, kend, g.a_vec)
                        : load_rc(g.A, g.lda, m0 + 64 * i + a_r, g.M, k0 + a_k, kend, g.a_vec);
#pragma unroll
        for (int j = 0; j < WN; ++j)
            rb[j] = BKC ? load_kc(g.B, g.ldb, n0 + 64 * j + b_r, g.N, k0 + b_k, kend, g.b_vec)
                        : load_rc(g.B, g.ldb, n0 + 64 * j + b_r, g.N, k0 + b_k, kend, g.b_vec);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (AKC) {
                As[buf][a_k + 0][64 * i + a_r] = ra[i].x;
                As[buf][a_k + 1][64 * i + a_r] = ra[i].y;
                As[buf][a_k + 2][64 * i + a_r] = ra[i].z;
                As[buf][a_k + 3][64 * i + a_r] = ra[i].w;
            } else {
                *reinterpret_cast<float4*>(&As[buf][a_k][64 * i + a_r]) = ra[i];
            }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (BKC) {
                Bs[buf][b_k + 0][64 * j + b_r] = rb[j].x;
                Bs[buf][b_k + 1][64 * j + b_r] = rb[j].y;
                Bs[buf][b_k + 2][64 * j + b_r] = rb[j].z;
                Bs[buf][b_k + 3][64 * j + b_r] = rb[j].w;
            } else {
                *reinterpret_cast<float4*>(&Bs[buf][b_k][64 * j + b_r]) = rb[j];
            }
            if (BSUM && !BKC) {   // column sums of B (db = sum_rows dz), first M-tile only
                csum[j].x += rb[j].x;
                csum[j].y += rb[j].y;
                csum[j].z += rb[j].z;
                csum[j].w += rb[j].w;
            }
        }
    };

    if (ntiles > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();

    // wave (wm, wn) owns rows [wm*32*WM, ...) x cols [wn*32*WN, ...) of the block tile
    const int ar = wm * 32 * WM + (lane & 31), br = wn * 32 * WN + (lane & 31), hi = lane >> 5;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) gload(tile + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = As[buf][kk + hi][ar + 32 * i];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = Bs[buf][kk + hi][br + 32 * j];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (tile + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------
    float* Cz = g.C + (size_t)blockIdx.z * ((gridDim.z > 1) ? (size_t)g.M * g.ldc : 0);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn * 32 * WN + 32 * j + (lane & 31);
        if (col >= g.N) continue;
        const float bias = (g.epi == EPI_FWD && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < g.M) {
                    const size_t o = (size_t)row * g.ldc + col;
                    float v = acc[i][j][r];
                    if (g.epi == EPI_FWD) {
                        v = tn_act_fwd(v + bias, g.act, g.act_prm);
                        if (g.mask) v *= (float)g.mask[o];
                    } else if (g.epi == EPI_DGRAD) {
                        if (g.prev_a) v *= tn_act_grad_from_out(g.prev_a[o], g.act, g.act_prm);
                        if (g.mask) v *= (float)g.mask[o];
                    }
                    Cz[o] = v;
                }
            }
        }
    }

    if (BSUM && !BKC && blockIdx.y == 0) {
        // reduce csum over the 16 k-rows of the staging layout (thread = (k = t>>4, q = t&15))
        __syncthreads();
        float* red = &As[0][0][0];   // reuse: [16][BN]  (16*BN <= BK*LDA)
#pragma unroll
        for (int j = 0; j < WN; ++j)
            *reinterpret_cast<float4*>(&red[(t >> 4) * BN + 64 * j + 4 * (t & 15)]) = csum[j];
        __syncthreads();
        if (t < BN) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * BN + t];
            if (n0 + t < g.N) g.colsum[(size_t)blockIdx.z * g.N + n0 + t] = s;
        }
    }
}

// sum split-K partial slabs (and the partial column sums) in a fixed order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws,
                                                           float* __restrict__ C, size_t MN, int S,
                                                           const float* __restrict__ colsum_ws,
                                                           float* __restrict__ colsum, int N) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < MN && (MN % 4 == 0)) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int z = 0; z < S; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)z * MN + i4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(C + i4) = s;
    } else {
        for (size_t i = i4; i < MN && i < i4 + 4; ++i) {
            float s = 0.f;
            for (int z = 0; z < S; ++z) s += ws[(size_t)z * MN + i];
            C[i] = s;
        }
    }
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (colsum && i < (size_t)N) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += colsum_ws[(size_t)z * N + i];
        colsum[i] = s;
    }
}

static inline int vec_ok(const void* p, int ld) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0);
}

// tile choice: 128x64 when that still gives >= ~200 blocks, else 64x64
static inline bool big_tile(int M, int N, int S) {
    return (long long)cdiv(M, 128) * cdiv(N, 64) * S >= 200;
}

template <bool AKC, bool BKC, bool BSUM>
static void launch_gemm(tn_ctx* ctx, GemmArgs& g, int S) {
    if (big_tile(g.M, g.N, S))
        gemm_f32_kernel<AKC, BKC, BSUM, 2, 1><<<dim3(cdiv(g.N, 64), cdiv(g.M, 128), S), 256, 0, ctx->stream>>>(g);
    else
        gemm_f32_kernel<AKC, BKC, BSUM, 1, 1><<<dim3(cdiv(g.N, 64), cdiv(g.M, 64), S), 256, 0, ctx->stream>>>(g);
}

static int wgrad_splits(int B, int n_in, int n_out) {
    const int tiles = cdiv(n_in, 128) * cdiv(n_out, 64);
    int S = cdiv(256, tiles);             // about one block per CU
    const int max_s = cdiv(B, 8 * BK);    // at least 8 K-tiles per split
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    if (S > 16) S = 16;
    return S;
}

// =====================================================================================
// skinny layers: n_out <= 16
// =====================================================================================
#define SK_MAX 16

// forward: one wave per row; lanes split K (coalesced x reads), W staged in LDS (row stride
// n_out+1 words: conflict-free), per-lane partial sums reduced across the wave with DPP.
template <int CTRL>
__device__ __forceinline__ float gdpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_allsum(float v) {
    v += gdpp<0xB1>(v);
    v += gdpp<0x4E>(v);
    v += gdpp<0x141>(v);
    v += gdpp<0x140>(v);
    const int iv = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48));
}

__global__ __launch_bounds__(256) void fc_skinny_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    float* __restrict__ a, int B, int n_in, int n_out, int act, float prm,
    const uint8_t* __restrict__ mask, int rows_per_wave) {
    extern __shared__ float sW[];       // [n_in][n_out+1]
    const int ldw = n_out + 1;
    for (int t = threadIdx.x; t < n_in * n_out; t += 256) {
        const int k = t / n_out, n = t - k * n_out;
        sW[k * ldw + n] = W[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int row = wave_global * rows_per_wave + rr;
        if (row >= B) return;
        float acc[SK_MAX];
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n) acc[n] = 0.f;
        const float* xr = x + (size_t)row * n_in;
        for (int k = lane; k < n_in; k += 64) {
            const float xv = xr[k];
            const float* wr = sW + k * ldw;
#pragma unroll
            for (int n = 0; n < SK_MAX; ++n)
                if (n < n_out) acc[n] = fmaf(xv, wr[n], acc[n]);
        }
        float mine = 0.f;
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n) {
            if (n < n_out) {       // wave-uniform
                const float s = wave_allsum(acc[n]);
                if (lane == n) mine = s;
            }
        }
        if (lane < n_out) {
            const size_t o = (size_t)row * n_out + lane;
            float v = tn_act_fwd(mine + (b ? b[lane] : 0.f), act, prm);
            if (mask) v *= (float)mask[o];
            a[o] = v;
        }
    }
}

// dgrad: thread = one (row, k) element: n_out FMAs with the row's dz (broadcast) and W[k,:]
__global__ __launch_bounds__(256) void fc_skinny_dgrad_kernel(
    const float* __restrict__ dz, const float* __restrict__ W, float* __restrict__ dx, int B, int n_in,
    int n_out, const float* __restrict__ prev_a, int act, float prm, const uint8_t* __restrict__ mask) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int row0 = blockIdx.y * 16;
    if (k >= n_in) return;
    float w[SK_MAX];
#pragma unroll
    for (int n = 0; n < SK_MAX; ++n) w[n] = (n < n_out) ? W[(size_t)k * n_out + n] : 0.f;
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + r;
        if (row >= B) break;
        const float* dzr = dz + (size_t)row * n_out;     // wave-uniform -> scalar loads
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n)
            if (n < n_out) s = fmaf(dzr[n], w[n], s);
        const size_t o = (size_t)row * n_in + k;
        if (prev_a) s *= tn_act_grad_from_out(prev_a[o], act, prm);
        if (mask) s *= (float)mask[o];
        dx[o] = s;
    }
}

// wgrad: thread = one input feature k, block = a chunk of rows; partial[chunk][k][n]
__global__ __launch_bounds__(256) void fc_skinny_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    float* __restrict__ dbpartial, int B, int n_in, int n_out, int rows_per_blk) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int row0 = blockIdx.y * rows_per_blk;
    const int row1 = min(B, row0 + rows_per_blk);
    float acc[SK_MAX], accb[SK_MAX];
#pragma unroll
    for (int n = 0; n < SK_MAX; ++n) acc[n] = accb[n] = 0.f;
    const bool live = k < n_in;
    for (int row = row0; row < row1; ++row) {
        const float xv = live ? x[(size_t)row * n_in + k] : 0.f;
        const float* dzr = dz + (size_t)row * n_out;     // wave-uniform
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n)
            if (n < n_out) {
                const float d = dzr[n];
                acc[n] = fmaf(xv, d, acc[n]);
                accb[n] += d;
            }
    }
    if (live) {
        float* p = partial + ((size_t)blockIdx.y * n_in + k) * n_out;
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n)
            if (n < n_out) p[n] = acc[n];
    }
    if (k == 0) {
#pragma unroll
        for (int n = 0; n < SK_MAX; ++n)
            if (n < n_out) dbpartial[(size_t)blockIdx.y * n_out + n] = accb[n];
    }
}

extern "C" {

int tn_fc_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B, int n_in,
              int n_out, int act, float act_param, const uint8_t* mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_fwd: bad shape");
    const size_t sk_lds = (size_t)n_in * (n_out + 1) * sizeof(float);
    if (n_out <= SK_MAX && sk_lds <= 60 * 1024) {
        const int rpw = 4;
        fc_skinny_fwd_kernel<<<cdiv(B, 4 * rpw), 256, sk_lds, ctx->stream>>>(
            x, W, b, a, B, n_in, n_out, act, act_param, mask, rpw);
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
    launch_gemm<true, false, false>(ctx, g, 1);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

size_t tn_fc_wgrad_ws_bytes(int B, int n_in, int n_out) {
    if (n_out <= SK_MAX) {
        const int chunks = cdiv(B, 256);
        return ((size_t)chunks * n_in * n_out + (size_t)chunks * n_out) * sizeof(float) + 64;
    }
    const int S = wgrad_splits(B, n_in, n_out);
    return ((size_t)S * n_in * n_out + (size_t)S * n_out) * sizeof(float) + 64;
}

int tn_fc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B, int n_in,
                int n_out, void* ws) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0 && ws != nullptr, "tn_fc_wgrad: bad arguments");
    if (n_out <= SK_MAX) {
        const int chunks = cdiv(B, 256);
        float* wsC = (float*)ws;
        float* wsB = wsC + (size_t)chunks * n_in * n_out;
        fc_skinny_wgrad_kernel<<<dim3(cdiv(n_in, 256), chunks), 256, 0, ctx->stream>>>(
            x, dz, wsC, wsB, B, n_in, n_out, 256);
        TN_LAUNCH_CHECK();
        const size_t MN = (size_t)n_in * n_out;
        splitk_reduce_kernel<<<cdiv(cdiv(MN, 4), 256), 256, 0, ctx->stream>>>(wsC, dW, MN, chunks, wsB, db, n_out);
        TN_LAUNCH_CHECK();
        return TN_OK;
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
        splitk_reduce_kernel<<<blocks, 256, 0, ctx->stream>>>(wsC, dW, MN, Sx, wsB, db, n_out);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

int tn_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in, int n_out,
                const float* prev_a, int prev_act, float prev_act_param, const uint8_t* prev_mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_dgrad: bad shape");
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
    launch_gemm<true, true, false>(ctx, g, 1);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
