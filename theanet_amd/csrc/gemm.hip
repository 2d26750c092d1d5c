// fp32 MFMA GEMM for the fully-connected layers (theanet/layer/hidden.py:30, layer.py:83).
//
//   C[M,N] = op(A)[M,K] . op(B)[K,N]     v_mfma_f32_32x32x2_f32 (exact fp32, fmaf chain)
//
// 64x64 block tile, BK=16, 4 waves (one 32x32 accumulator each), double-buffered LDS with a
// register-staged prefetch (global loads of tile t+1 are in flight while tile t is multiplied),
// one barrier per K-tile.  Operands may be row- or column-contiguous (NN / NT / TN) so that
// forward, dgrad (dz.W^T) and wgrad (x^T.dz) all run on the same kernel; wgrad uses split-K
// over the batch dimension (M x N is small, K = batch is long) with a deterministic reduce.
// Epilogues fuse bias + activation + dropout mask (forward) and activation-gradient + mask
// (dgrad), so no elementwise pass touches HBM again.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 16
#define LDS_LD 68   // 64 + 4: rows stay 16-byte aligned for b128 stores, reads conflict-free

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

template <bool AKC, bool BKC, bool BSUM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDS_LD];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    // per-thread staging coordinates
    const int a_r = AKC ? (t >> 2) : 4 * (t & 15);
    const int a_k = AKC ? 4 * (t & 3) : (t >> 4);
    const int b_r = BKC ? (t >> 2) : 4 * (t & 15);
    const int b_k = BKC ? 4 * (t & 3) : (t >> 4);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
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
            As[buf][a_k + 0][a_r] = ra.x;
            As[buf][a_k + 1][a_r] = ra.y;
            As[buf][a_k + 2][a_r] = ra.z;
            As[buf][a_k + 3][a_r] = ra.w;
        } else {
            *reinterpret_cast<float4*>(&As[buf][a_k][a_r]) = ra;
        }
        if (BKC) {
            Bs[buf][b_k + 0][b_r] = rb.x;
            Bs[buf][b_k + 1][b_r] = rb.y;
            Bs[buf][b_k + 2][b_r] = rb.z;
            Bs[buf][b_k + 3][b_r] = rb.w;
        } else {
            *reinterpret_cast<float4*>(&Bs[buf][b_k][b_r]) = rb;
        }
        if (BSUM && !BKC) {   // column sums of B (db = sum_rows dz), first M-tile only
            csum.x += rb.x;
            csum.y += rb.y;
            csum.z += rb.z;
            csum.w += rb.w;
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
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[buf][kk + hi][ar];
            const float b = Bs[buf][kk + hi][br];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (tile + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------
    float* Cz = g.C + (size_t)blockIdx.z * ((gridDim.z > 1) ? (size_t)g.M * g.ldc : 0);
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
        const float bias = (g.epi == EPI_FWD && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < g.M) {
                const size_t o = (size_t)row * g.ldc + col;
                float v = acc[r];
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

    if (BSUM && !BKC && blockIdx.y == 0) {
        // reduce csum over the 16 k-rows of the staging layout (thread = (k = t>>4, q = t&15))
        __syncthreads();
        float* red = &As[0][0][0];   // reuse: [16][64]
        *reinterpret_cast<float4*>(&red[(t >> 4) * 64 + 4 * (t & 15)]) = csum;
        __syncthreads();
        if (t < 64) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * 64 + t];
            if (n0 + t < g.N) g.colsum[(size_t)blockIdx.z * g.N + n0 + t] = s;
        }
    }
}

// sum split-K partial slabs (and the partial column sums) in a fixed order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws,
                                                           float* __restrict__ C, size_t MN, int S,
                                                           const float* __restrict__ colsum_ws,
                                                           float* __restrict__ colsum, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < MN) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += ws[(size_t)z * MN + i];
        C[i] = s;
    }
    if (colsum && i < (size_t)N) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += colsum_ws[(size_t)z * N + i];
        colsum[i] = s;
    }
}

static inline int vec_ok(const void* p, int ld) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0);
}

static int wgrad_splits(int B, int n_in, int n_out) {
    const int tiles = cdiv(n_in, BM) * cdiv(n_out, BN);
    int S = cdiv(1024, tiles);            // aim at ~4 blocks per CU
    const int max_s = cdiv(B, 4 * BK);    // at least 4 K-tiles per split
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    return S;
}

extern "C" {

int tn_fc_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B, int n_in,
              int n_out, int act, float act_param, const uint8_t* mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_fwd: bad shape");
    GemmArgs g{};
    g.A = x; g.B = W; g.C = a;
    g.M = B; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out;
    g.kchunk = cdiv(n_in, BK) * BK;
    g.epi = EPI_FWD; g.bias = b; g.mask = mask; g.act = act; g.act_prm = act_param;
    g.a_vec = vec_ok(x, n_in); g.b_vec = vec_ok(W, n_out);
    gemm_f32_kernel<true, false, false><<<dim3(cdiv(n_out, BN), cdiv(B, BM), 1), 256, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

size_t tn_fc_wgrad_ws_bytes(int B, int n_in, int n_out) {
    const int S = wgrad_splits(B, n_in, n_out);
    return ((size_t)S * n_in * n_out + (size_t)S * n_out) * sizeof(float) + 64;
}

int tn_fc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B, int n_in,
                int n_out, void* ws) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0 && ws != nullptr, "tn_fc_wgrad: bad arguments");
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
    gemm_f32_kernel<false, false, true><<<dim3(cdiv(n_out, BN), cdiv(n_in, BM), Sx), 256, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    if (Sx > 1) {
        const size_t MN = (size_t)n_in * n_out;
        splitk_reduce_kernel<<<cdiv(MN, 256), 256, 0, ctx->stream>>>(wsC, dW, MN, Sx, wsB, db, n_out);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

int tn_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in, int n_out,
                const float* prev_a, int prev_act, float prev_act_param, const uint8_t* prev_mask) {
    TN_REQUIRE(B > 0 && n_in > 0 && n_out > 0, "tn_fc_dgrad: bad shape");
    GemmArgs g{};
    g.A = dz; g.B = W; g.C = dx;
    g.M = B; g.N = n_in; g.K = n_out;
    g.lda = n_out; g.ldb = n_out; g.ldc = n_in;     // B(k,n) = W[n*n_out + k]: k-contiguous
    g.kchunk = cdiv(n_out, BK) * BK;
    g.epi = EPI_DGRAD; g.prev_a = prev_a; g.mask = prev_mask; g.act = prev_act; g.act_prm = prev_act_param;
    g.a_vec = vec_ok(dz, n_out); g.b_vec = vec_ok(W, n_out);
    gemm_f32_kernel<true, true, false><<<dim3(cdiv(n_in, BN), cdiv(B, BM), 1), 256, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
