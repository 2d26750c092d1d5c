// DTYPE 'float16', fully-connected products on fp16-RESIDENT activations (BASELINE.json configs[4]).
// Same products as theanet/layer/hidden.py:30 (a = act(x . W + b)) and their Theano gradients (layer.py:83); the
// reference is float32-only (weights.py:8), so the arithmetic of this mode is the oracle's stored-fp16 restatement:
// the layer input x lives in HBM as halfs (the flattened c8 tensor of the conv stack below, or any (B, n) half
// matrix), master weights W (n_in, n_out), biases, weight gradients and the layer OUTPUT stay fp32; both operands of
// every product are halfs (W rounded nearest-even while it is loaded, dz as fp16(gs * dz)), products exact, fp32
// accumulation.
//
// The input is consumed in ITS order: column k of x is c8 element ((o * HW + p) * 8 + e) = channel 8 o + e at pixel p,
// while the reference flattens NCHW (row (8 o + e) * HW + p of W, neuralnet.py:168-173): the kernels walk W through that
// row map, so neither the activations nor the 64 MB weight matrix are ever re-ordered.  Channels beyond C (zero in a c8
// tensor) read a clamped row.
//
// Shapes here are short and deep (wide6: 128 x 16384 -> 1024): every product streams W once and is bound by that.
//   forward : wave = 128 rows x 64 outputs x a K range, operands straight from global memory into the MFMA layout
//             (x: 16-byte loads; W: 8 dword loads of 128 contiguous bytes per 32 lanes, converted in registers);
//             the block's four K ranges meet in LDS, K slabs in a finishing kernel (bias + act + mask).
//   dgrad   : C = W_tile . dz^T with the rows of the W tile permuted so that a lane's accumulators are whole c8 cells:
//             16-byte stores of fp16(acc * act'(y)) -- y = the pooled output below, in the same order.
//   wgrad   : both operands want 8 consecutive SAMPLES per lane but are stored sample-major: tiles go to LDS as they
//             are (dz converted on the way) and gfx950's transposing LDS read (ds_read_b64_tr_b16) delivers them.
#include "common.h"
#include "elastic_field.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short fc8_short4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half4v fc8_tr16(const char* l) {
    return __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fc8_short4*)l));
}
__host__ __device__ __forceinline__ int fc8_swap23(int j) { return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1); }

struct FC8 {
    const _Float16* x;      // (M, Kc) halfs, Kc = C8 * HW * 8
    const float* W;         // (C * HW, N)
    const float* bias;
    const float* dz;        // (M, N) fp32
    const _Float16* dz16;   // dgrad: fp16(gs dz), (M, Np)
    const _Float16* ya;     // dgrad: output of the layer below in x's order (act' is taken from it) or NULL
    const uint8_t* mask;    // forward: dropout mask (M, N) or NULL
    float* out;             // forward: (M, N) fp32
    _Float16* dx;           // dgrad: (M, Kc) halfs, carries the gradient scale
    _Float16* dz16w;        // wgrad: the rounded dz it stages, written out for the input-gradient product that follows (or NULL)
    float* ws;              // forward: K slabs [S][M][N]; wgrad: sample slabs [S][C*HW][N]
    float* dbws;            // wgrad: [S][N]
    int M, N, Np, Kc, C, HW, S, krange, act;
    int xcd;                // block ids decoded XCD-aware (the host checked the divisibility): see fc8_fwd_kernel / fc8_wgrad_kernel
    unsigned magic;         // 2^32 / HW + 1 (0: HW == 1), see fc8_wrow
    float prm, gs, oscale;
};

// tn_c8_fc_wgrad leaves the rounded dz it wrote for the tn_c8_fc_dgrad call that follows it (same context, stream, dz,
// shape and gradient scale; consumed or dropped by the next tn_c8_fc_dgrad / any other c8 dense call)
struct Fc8Dz16Keep {
    tn_ctx* ctx = nullptr; hipStream_t stream = nullptr; const float* dz = nullptr; _Float16* dz16 = nullptr;
    int B = 0, n_out = 0; float gs = 0.f;
    unsigned long long gen = 0;            // ctx->scratch_gen when it was written: nobody has asked for scratch since
};
static Fc8Dz16Keep fc8_dz16_keep;

// cell (= c8 column >> 3) -> (o, p) = (cell / HW, cell % HW) without an integer division: magic = 2^32 / HW + 1 (exact
// while cell * HW < 2^32, checked by the host); HW == 1: magic 0
__device__ __forceinline__ int fc8_wrow(const FC8& g, int cell, int e, int rows) {
    const int o = g.magic ? (int)__umulhi((unsigned)cell, g.magic) : cell;
    const int p = cell - o * g.HW;
    return min((o * 8 + e) * g.HW + p, rows - 1);
}

__device__ __forceinline__ half8 fc8_cvt8(const float (&v)[8]) {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    return h;
}

typedef unsigned fc8_u4 __attribute__((ext_vector_type(4)));
typedef float fc8_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4v fc8_cvt4(const fc8_f4 v) {
    return half4v{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
}

// ---------------------------------------------------------------------------------------------------------------
// forward: grid = (N / 64 column groups, S K-slabs, M / 128 row groups); block = 128 rows x 64 outputs x one K slab in
// chunks of 64 c8 columns.  A chunk is FOUR float4 of W and four 16-byte pieces of x per thread; a ring of FC8_NST such
// register sets keeps NST - 1 chunks (16 KB of W each) of every block in flight -- the op streams the 64 MB matrix
// once and is bound by that (short batches) -- and one LDS tile at a time feeds the matrix core: x as stored
// ([row][k], 16-byte reads), W converted to halfs as stored ([k][n]) and read through the transposing LDS read.
// ---------------------------------------------------------------------------------------------------------------
struct Fc8Drop {            // dropout drawn by the finishing kernel (tn_c8_fc_fwd_dropout): the numbers of tn_dropout_mask
    uint8_t* mask_out;      // NULL: no inline dropout
    float pdrop;
    uint32_t k0, k1, step;
    const uint32_t* d_step;
    uint64_t elem0;
};
// bias + activation + dropout of FOUR consecutive outputs i .. i+3 of row-major (M, N) (i % 4 == 0: one bias quad): the
// arithmetic of fc8_fwd_finish_kernel for one slab, used by the forward kernel itself when the product has one K slab
__device__ __forceinline__ float4 fc8_finish4(float4 v, size_t i, int n, const float* __restrict__ bias,
                                              const uint8_t* __restrict__ mask, int act, float prm, const Fc8Drop& dr) {
    const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
    float x[4] = {0.f + v.x + b4.x, 0.f + v.y + b4.y, 0.f + v.z + b4.z, 0.f + v.w + b4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = act == TN_ACT_LEAKY ? fmaxf(0.f, x[j]) + fminf(0.f, x[j]) * prm : tn_act_fwd(x[j], act, prm);
    if (dr.mask_out) {
        const uint64_t e = dr.elem0 + i, cq = e >> 2;
        const uint32_t st = dr.step + (dr.d_step ? *dr.d_step : 0u);
        const u32x4 r0 = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), st, TN_STREAM_DROPOUT, dr.k0, dr.k1);
        uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, 0u, 0u, 0u, 0u};
        const int sh = (int)(e & 3);
        if (sh) {                               // (elem0 not a multiple of 4: the quad straddles two Philox blocks)
            const uint64_t c1 = cq + 1;
            const u32x4 r1 = philox4x32((uint32_t)c1, (uint32_t)(c1 >> 32), st, TN_STREAM_DROPOUT, dr.k0, dr.k1);
            w[4] = r1.x; w[5] = r1.y; w[6] = r1.z; w[7] = r1.w;
        }
        uint32_t m4 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t wj = sh == 0 ? w[j] : sh == 1 ? w[j + 1] : sh == 2 ? w[j + 2] : w[j + 3];
            const bool keep = tn_u01(wj) >= dr.pdrop;
            m4 |= (keep ? 1u : 0u) << (8 * j);
            x[j] = keep ? x[j] : 0.f;
        }
        *reinterpret_cast<uint32_t*>(dr.mask_out + i) = m4;
    } else if (mask) {
        const uint32_t m4 = *reinterpret_cast<const uint32_t*>(mask + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = ((m4 >> (8 * j)) & 0xffu) ? x[j] : 0.f;
    }
    return make_float4(x[0], x[1], x[2], x[3]);
}

static int fc8_xcd_on() {             // TN_FC8_XCD=0: plain block decode (A/B)
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("TN_FC8_XCD");
        on = e ? atoi(e) : 1;
    }
    return on;
}
#define FC8_NST 4
#define FC8F_XS 144         // x tile row stride (64 halfs + 16 bytes: 9 x 16 B, every 16-byte read of 32 rows on its own banks)
#define FC8F_WS 192         // W tile row stride (64 halfs + 64 bytes = 64 (mod 128): the 4 rows of a transposing read on disjoint banks)
// FIN: the product has ONE K slab and whole 64-column tiles: bias, activation and dropout happen on the way out and the
// outputs go straight to g.out (no slab, no finishing launch)
template <bool FIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fc8_fwd_kernel(FC8 g, Fc8Drop dr) {
    __shared__ __attribute__((aligned(16))) char xs[128 * FC8F_XS];
    __shared__ __attribute__((aligned(16))) char wsm[64 * FC8F_WS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    // Blocks are handed to the 8 XCDs round robin by linear id and every XCD has its own L2: with the plain decode the
    // column tiles that share an x tile (same row tile, same K slab) sat on eight XCDs and each fetched it from HBM
    // (cifar_like: 79.8 MB per launch for 28 algorithmic).  XCD-aware: the column tiles of a (K slab, row tile) pair
    // are consecutive blocks of ONE XCD (15.3 -> 14.0 us one step at a time on cifar_like, 21.5 -> 20.6 on wide6: the
    // re-reads were mostly served by the memory-side cache already).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd) {
        const int L = bx + gridDim.x * (by + gridDim.y * bz), xc = L & 7, j = L >> 3;
        bx = j % (int)gridDim.x;
        const int grp = (j / (int)gridDim.x) * 8 + xc;           // (K slab, row tile) pair
        by = grp % (int)gridDim.y; bz = grp / (int)gridDim.y;
    }
    const int n0 = bx * 64, m0 = bz * 128;
    const int kbeg = by * g.krange, kend = min(g.Kc, kbeg + g.krange);
    const int nch = (kend - kbeg) >> 6, rows = g.C * g.HW;
    const int wm = wave >> 1, wn = wave & 1;                 // wave = 64 rows x 32 outputs
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // staging roles: x row t >> 1, 32-half part t & 1; W k-rows (t >> 4) + 16 i, columns 4 (t & 15) ..
    const _Float16* xp = g.x + (size_t)min(m0 + (t >> 1), g.M - 1) * g.Kc + (t & 1) * 32;
    char* const xdst = xs + (t >> 1) * FC8F_XS + (t & 1) * 64;
    const float* wcol = g.W + min(n0 + 4 * (t & 15), g.N - 4);
    char* const wdst = wsm + (t >> 4) * FC8F_WS + 8 * (t & 15);
    const int we = (t >> 4) & 7, wc0 = (kbeg >> 3) + (t >> 7);
    fc8_u4 xr[FC8_NST][4];
    fc8_f4 wr[FC8_NST][4];
    auto gload = [&](int st, int c) __attribute__((always_inline)) {
        const int cc = min(c, nch - 1);                      // past the end: the last chunk again (cache hits, never used)
        const _Float16* xq = xp + kbeg + 64 * cc;
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[st][i] = *reinterpret_cast<const fc8_u4*>(xq + 8 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wr[st][i] = *reinterpret_cast<const fc8_f4*>(wcol + (size_t)fc8_wrow(g, wc0 + 8 * cc + 2 * i, we, rows) * g.N);
    };
    auto lstore = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<fc8_u4*>(xdst + 16 * i) = xr[st][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<half4v*>(wdst + 16 * i * FC8F_WS) = fc8_cvt4(wr[st][i]);
    };
    const int grp = lane >> 4, r4 = (lane >> 2) & 3, q4 = lane & 3;
    const char* const ard = xs + (wm * 64 + l31) * FC8F_XS + 16 * hi;
    const char* const brd = wsm + (8 * (grp >> 1) + r4) * FC8F_WS + (wn * 32 + 16 * (grp & 1) + 4 * q4) * 2;
    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half4v b0 = fc8_tr16(brd + 16 * ks * FC8F_WS), b1 = fc8_tr16(brd + (16 * ks + 4) * FC8F_WS);
            const half8 b = half8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const half8 a = *reinterpret_cast<const half8*>(ard + 32 * i * FC8F_XS + 32 * ks);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
        }
    };
#pragma unroll
    for (int u = 0; u < FC8_NST; ++u) gload(u, u);
    for (int c = 0; c < nch; c += FC8_NST) {
#pragma unroll
        for (int u = 0; u < FC8_NST; ++u) {
            lstore(u);
            gload(u, c + u + FC8_NST);
            __syncthreads();
            if (c + u < nch) compute();
            __syncthreads();
        }
    }
    float* const wz = g.ws + (size_t)by * g.M * g.N;
    const int n = n0 + wn * 32 + l31;
    if (n0 + 64 <= g.N && (g.N & 3) == 0) {
        // whole column tiles: 16-byte stores through LDS (a wave's 32 x 32 half at a time, 4 KB per wave), as in the weight
        // gradient below; the operand tiles are dead behind the loop's last barrier
        float* const T = reinterpret_cast<float*>(xs) + wave * (32 * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[i][r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane + 64 * q, row = idx >> 3, c4 = idx & 7;
                const int m = m0 + wm * 64 + 32 * i + row;
                const float4 v = *reinterpret_cast<const float4*>(T + row * 32 + 4 * c4);
                const int nn = n0 + wn * 32 + 4 * c4;
                if (FIN) {
                    if (m < g.M)
                        *reinterpret_cast<float4*>(g.out + (size_t)m * g.N + nn) =
                            fc8_finish4(v, (size_t)m * g.N + nn, nn, g.bias, g.mask, g.act, g.prm, dr);
                } else if (m < g.M) {
                    *reinterpret_cast<float4*>(wz + (size_t)m * g.N + nn) = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < g.M && n < g.N) wz[(size_t)m * g.N + n] = acc[i][r];
        }
}

// out = act(sum of the K slabs + bias) (* mask), slabs added in order; thread = one output (short batches: 128 x 1024
// outputs are only 512 blocks even so -- with four outputs per thread the kernel ran on 128 blocks at 1.5 TB/s)
__global__ __launch_bounds__(256) void fc8_fwd_finish_kernel(const float* __restrict__ ws, int S, size_t MN, int N,
                                                            const float* __restrict__ bias, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ out, int act, float prm, Fc8Drop dr) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float v = 0.f;
#pragma unroll 8
    for (int z = 0; z < S; ++z) v += ws[(size_t)z * MN + i];
    v += bias[i % N];
    v = act == TN_ACT_LEAKY ? fmaxf(0.f, v) + fminf(0.f, v) * prm : tn_act_fwd(v, act, prm);
    if (dr.mask_out) {
        // element e = elem0 + i uses word (e & 3) of philox(e >> 2): one call per thread (a thread per output keeps the
        // launch wide for short batches; the four-fold redundant calls are ~100 instructions beside S slab reads)
        const uint64_t e = dr.elem0 + i, cq = e >> 2;
        const u32x4 r = philox4x32((uint32_t)cq, (uint32_t)(cq >> 32), dr.step + (dr.d_step ? *dr.d_step : 0u),
                                   TN_STREAM_DROPOUT, dr.k0, dr.k1);
        const uint32_t w = ((e & 3) == 0) ? r.x : ((e & 3) == 1) ? r.y : ((e & 3) == 2) ? r.z : r.w;
        const bool keep = tn_u01(w) >= dr.pdrop;
        dr.mask_out[i] = keep ? 1 : 0;
        v = keep ? v : 0.f;
    } else if (mask) {
        v = mask[i] ? v : 0.f;
    }
    out[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// dgrad: dx16[m][k] = fp16(act'(ya[m][k]) * sum_n fp16(gs dz[m][n]) * fp16(W[row(k)][n])); grid = (Kc / 64, M / 128);
// block = 64 c8 columns x 128 rows, the reduction in chunks of 64 outputs: the block's 64 rows of W (256 contiguous
// bytes each per chunk) and the dz rows go through a ring of register sets into one LDS tile pair as halfs, [row][n].
// C = W_tile . dz^T with the rows of the W tile in the order MFMA row j <-> column base + swap23(j): a lane's accumulators
// are whole c8 cells.  wave = (32-column tile, 64-row half).
// ---------------------------------------------------------------------------------------------------------------
// dz16[m][n] = fp16(gs * dz[m][n]), rows padded with zeros to a multiple of 64 outputs: the operand every block of the
// input-gradient product re-reads (Kc / 64 times) at half the bytes, converted once
__global__ __launch_bounds__(256) void fc8_dz16_kernel(const float* __restrict__ dz, _Float16* __restrict__ out, int M, int N,
                                                      int Np, float gs) {
    const int i = blockIdx.x * 256 + threadIdx.x;            // one 4-column group
    const int q = Np >> 2, m = i / q, n = 4 * (i - m * q);
    if (m >= M) return;
    fc8_f4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < N) v = *reinterpret_cast<const fc8_f4*>(dz + (size_t)m * N + n) * gs;
    *reinterpret_cast<half4v*>(out + (size_t)m * Np + n) = fc8_cvt4(v);
}

template <int NST>                  // ring depth
__global__ __launch_bounds__(512) void fc8_dgrad_kernel(FC8 g) {
    constexpr int NC = 64, RS = NC * 2 + 16;                 // outputs per chunk; row stride: an odd number of 16-byte slots
    __shared__ __attribute__((aligned(16))) char wl[64 * RS];
    __shared__ __attribute__((aligned(16))) char dl[128 * RS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int kb0 = blockIdx.x * 64, m0 = blockIdx.y * 128;
    const int wk = wave & 1, wm = wave >> 1;                 // wave = 32 columns x 32 rows
    const int rows = g.C * g.HW, nch = g.Np / NC;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging: W row (t >> 4) + 32 i, 4 outputs (t & 15); dz16 row (t >> 3) + 64 i, 8 outputs (t & 7)
    const int n4 = 4 * (t & 15);
    const float* wp[2];
    const _Float16* dp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = (t >> 4) + 32 * i, kcol = kb0 + (rr & 32) + fc8_swap23(rr & 31);
        wp[i] = g.W + (size_t)fc8_wrow(g, kcol >> 3, kcol & 7, rows) * g.N;
        dp[i] = g.dz16 + (size_t)min(m0 + (t >> 3) + 64 * i, g.M - 1) * g.Np + 8 * (t & 7);
    }
    char* const wdst = wl + (t >> 4) * RS + 2 * n4;
    char* const ddst = dl + (t >> 3) * RS + 16 * (t & 7);
    fc8_f4 wr[NST][2];
    fc8_u4 dr[NST][2];
    auto gload = [&](int st, int c) __attribute__((always_inline)) {
        const int n = NC * min(c, nch - 1);
        const int nn = min(n + n4, g.N - 4);                 // W columns past N meet zeros of dz16
#pragma unroll
        for (int i = 0; i < 2; ++i) wr[st][i] = *reinterpret_cast<const fc8_f4*>(wp[i] + nn);
#pragma unroll
        for (int i = 0; i < 2; ++i) dr[st][i] = *reinterpret_cast<const fc8_u4*>(dp[i] + n);
    };
    auto lstore = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<half4v*>(wdst + 32 * i * RS) = fc8_cvt4(wr[st][i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<fc8_u4*>(ddst + 64 * i * RS) = dr[st][i];
    };
    const char* const ard = wl + (wk * 32 + l31) * RS + 16 * hi;
    const char* const brd = dl + (wm * 32 + l31) * RS + 16 * hi;
    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < NC / 16; ++ks) {
            const half8 a = *reinterpret_cast<const half8*>(ard + 32 * ks);
            const half8 b = *reinterpret_cast<const half8*>(brd + 32 * ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    };
#pragma unroll
    for (int u = 0; u < NST; ++u) gload(u, u);
    for (int c = 0; c < nch; c += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            lstore(u);
            gload(u, c + u + NST);
            __syncthreads();
            if (c + u < nch) compute();
            __syncthreads();
        }
    }
    // lane (sample l31 of tile i): registers 0-7 / 8-15 are the cells 8 (hi) / 8 (2 + hi) of the 32 columns
    const int kbase = kb0 + wk * 32;
    const float tie = g.prm > 0.f ? 1.f + g.prm : 0.f;
    {
        const int m = m0 + wm * 32 + l31;
        if (m >= g.M) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t o = (size_t)m * g.Kc + kbase + 8 * (2 * h + hi);
            half8 y8;
            if (g.ya) y8 = *reinterpret_cast<const half8*>(g.ya + o);
            half8 o8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = acc[h * 8 + e];
                if (g.ya) {
                    const float y = (float)y8[e];
                    v *= g.act == TN_ACT_LEAKY ? (y > 0.f ? 1.f : (y < 0.f ? g.prm : tie)) : tn_act_grad_from_out(y, g.act, g.prm);
                }
                o8[e] = (_Float16)v;
            }
            *reinterpret_cast<half8*>(g.dx + o) = o8;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad: dW[row(k)][n] = (1/gs) sum_m x16[m][k] * fp16(gs dz[m][n]); grid = (Kc / 128, N / 128, S sample slabs);
// block tile 128 x 128, wave = 64 x 64 (2 x 2 MFMA tiles); sample chunks of 64 go through LDS as they are stored
// ([sample][128 columns] halfs, row stride 320 bytes) and come out transposed.
// ---------------------------------------------------------------------------------------------------------------
#define FC8_RS 320          // row stride = 64 (mod 256) bytes: the 4 x 32-byte rows of both 16-lane groups of a half-wave on disjoint banks
// RIDER: grid layers z >= S carry a light independent job of the step -- the elastic field of the NEXT minibatch
// (tn_rider_elastic_field; inlayers.py:72-125) -- as extra blocks behind the product, like the fp32 nets' paired GEMM
// launch does: a float16 net has no such launch, and the field cost its stream a 14 us launch of its own per step.
__global__ __launch_bounds__(256) void fc8_wgrad_kernel(FC8 g, ElField rider, int nrider) {
    __shared__ __attribute__((aligned(16))) char lds[2][64 * FC8_RS];        // [x | dz][sample][column]
    if ((int)blockIdx.z >= g.S) {
        const int rb = (((int)blockIdx.z - g.S) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (rb < nrider) elastic_field_block<true>(rider, reinterpret_cast<float*>(&lds[0][0]), rb);
        return;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    // (an XCD-aware decode -- all blocks of a sample slab on one XCD -- was measured: no change, the re-reads of the
    // other seven L2s are served by the memory-side cache)
    const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 128, z = blockIdx.z;
    const int mchunk = g.krange;                     // samples per slab (a multiple of 64)
    const int mbeg = z * mchunk, mend = min(g.M, mbeg + mchunk);
    const int wk = (wave & 1) * 64, wn = (wave >> 1) * 64;
    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
    }
    const bool want_db = blockIdx.x == 0 && (wave & 1) == 0;
    const half8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f,
                        (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    // staging: thread -> sample row t >> 2 (64 rows), 32-column quarter t & 3
    const int sr = t >> 2, sq = (t & 3) * 32;
    // transposing reads: group of 16 lanes = 4 samples x 16 columns; lane supplies sample r4, columns 4 q .. 4 q + 3
    const int grp = lane >> 4, r4 = (lane >> 2) & 3, q4 = lane & 3;
    const int rd = (8 * (grp >> 1) + r4) * FC8_RS + (16 * (grp & 1) + 4 * q4) * 2;
    uint4 xv[4];
    float4 dv[8];
    auto gload = [&](int mc) __attribute__((always_inline)) {      // the chunk's rows as stored (clamped: masked when staged)
        const int mm_ = min(mc + sr, g.M - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xv[i] = *reinterpret_cast<const uint4*>(g.x + (size_t)mm_ * g.Kc + min(k0 + sq + 8 * i, g.Kc - 8));
#pragma unroll
        for (int i = 0; i < 8; ++i)
            dv[i] = *reinterpret_cast<const float4*>(g.dz + (size_t)mm_ * g.N + min(n0 + sq + 4 * i, g.N - 4));
    };
    gload(mbeg);
    for (int mc = mbeg; mc < mend; mc += 64) {
        const bool ok = mc + sr < mend;
        __syncthreads();                       // the previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = ok && k0 + sq + 8 * i < g.Kc;
            *reinterpret_cast<uint4*>(lds[0] + sr * FC8_RS + (sq + 8 * i) * 2) = in ? xv[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const bool in = ok && n0 + sq + 4 * i < g.N;
            const float s = in ? g.gs : 0.f;
            const float f[8] = {dv[i].x * s, dv[i].y * s, dv[i].z * s, dv[i].w * s, dv[i + 1].x * s, dv[i + 1].y * s, dv[i + 1].z * s, dv[i + 1].w * s};
            const half8 h8 = fc8_cvt8(f);
            *reinterpret_cast<half8*>(lds[1] + sr * FC8_RS + (sq + 4 * i) * 2) = h8;
            // the blocks of row tile 0 keep what they rounded: fp16(gs dz), (M, Np), zeros beyond N -- exactly what
            // fc8_dz16_kernel writes for tn_c8_fc_dgrad (which then skips that 5 us launch)
            if (g.dz16w && blockIdx.x == 0 && ok && n0 + sq + 4 * i < g.Np)
                *reinterpret_cast<half8*>(g.dz16w + (size_t)(mc + sr) * g.Np + n0 + sq + 4 * i) = h8;
        }
        gload(min(mc + 64, mend - 1));         // the next chunk travels during this chunk's products
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {       // 16 samples per step
            half8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* ap = lds[0] + ks * 16 * FC8_RS + rd + (wk + 32 * i) * 2;
                const half4v a0 = fc8_tr16(ap), a1 = fc8_tr16(ap + 4 * FC8_RS);
                a[i] = half8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const char* bp = lds[1] + ks * 16 * FC8_RS + rd + (wn + 32 * i) * 2;
                const half4v b0 = fc8_tr16(bp), b1 = fc8_tr16(bp + 4 * FC8_RS);
                b[i] = half8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (want_db) {
                accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, b[0], accb[0], 0, 0, 0);
                accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, b[1], accb[1], 0, 0, 0);
            }
        }
    }
    const int rows = g.C * g.HW;
    float* const wz = g.ws + (size_t)z * rows * g.N;
    if (n0 + 128 <= g.N && (g.N & 3) == 0) {
        // whole column tiles: a wave's 32 x 64 half goes through LDS ([row][64 columns], 8 KB per wave) and leaves as
        // 16-byte stores, 256 contiguous bytes per row -- straight from the accumulators it was 64 4-byte stores per lane
        // (the launch writes 67 MB on wide6: the stores are what it does)
        __syncthreads();                       // the operand tiles are dead
        float* const T = reinterpret_cast<float*>(&lds[0][0]) + wave * (32 * 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + 32 * j + l31] = acc[i][j][r] * g.oscale;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = lane + 64 * q, row = idx >> 4, c4 = idx & 15;
                const int k = k0 + wk + 32 * i + row;
                const int cell = k >> 3, o = cell / g.HW, p = cell - o * g.HW, ch = o * 8 + (k & 7);
                const float4 v = *reinterpret_cast<const float4*>(T + row * 64 + 4 * c4);
                if (k < g.Kc && ch < g.C)
                    *reinterpret_cast<float4*>(wz + (size_t)(ch * g.HW + p) * g.N + n0 + wn + 4 * c4) = v;
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + wk + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int cell = k >> 3, o = cell / g.HW, p = cell - o * g.HW, ch = o * 8 + (k & 7);
            if (k < g.Kc && ch < g.C) {
                float* wr = wz + (size_t)(ch * g.HW + p) * g.N;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn + 32 * j + l31;
                    if (n < g.N) wr[n] = acc[i][j][r] * g.oscale;
                }
            }
        }
    if (want_db && hi == 0) {                   // row 0 of the product with ones (every row is the column sum)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + l31;
            if (n < g.N) g.dbws[(size_t)z * g.N + n] = accb[j][0] * g.oscale;
        }
    }
}

static unsigned fc8_magic(int HW) { return HW == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)HW + 1u); }

static int fc8_check(tn_ctx* ctx, int B, int C, int HW, int n_out, const char* what) {
    const int Kc = ((C + 7) / 8) * HW * 8;
    TN_REQUIRE(B > 0 && C > 0 && HW > 0 && n_out > 0, "%s: bad shape", what);
    TN_REQUIRE(Kc % 64 == 0 && n_out % 32 == 0, "%s: %d inputs (c8) and %d outputs must be multiples of 64 / 32", what, Kc, n_out);
    TN_REQUIRE((uint64_t)(Kc / 8) * (uint64_t)HW < (1ull << 32), "%s: %d x %d inputs: too many", what, C, HW);
    return TN_OK;
}

extern "C" {

// 1 if the fp16-resident FC products take this layer (input = C maps of HW pixels, c8 order; HW = 1: a plain matrix)
int tn_c8_fc_supported(int B, int C, int HW, int n_out) {
    const int Kc = ((C + 7) / 8) * HW * 8;
    return B > 0 && C > 0 && HW > 0 && n_out > 0 && Kc % 64 == 0 && n_out % 32 == 0;
}

// a (B, n_out) fp32 = act(x16 . W + b) (* mask); x16 (B, C8*HW*8) halfs in c8 order, W (C*HW, n_out) fp32 in the reference's
// NCHW-flattened row order (hidden.py:30, neuralnet.py:168-173)
static int fc8_fwd_run(tn_ctx* ctx, const void* x, const float* W, const float* b, float* a, int B, int C, int HW, int n_out,
                       int act, float act_param, const uint8_t* mask, const Fc8Drop& dr) {
    int rc = fc8_check(ctx, B, C, HW, n_out, "tn_c8_fc_fwd");
    if (rc) return rc;
    FC8 g{};
    g.x = static_cast<const _Float16*>(x); g.W = W; g.M = B; g.N = n_out; g.C = C; g.HW = HW;
    g.Kc = ((C + 7) / 8) * HW * 8;
    g.magic = fc8_magic(HW);
    const int colg = cdiv(n_out, 64), rowg = cdiv(B, 128);
    // K slabs: one block per CU (two measured 15 % slower: twice the slab traffic), at least four chunks of 64 columns per block
    int S = cdiv(ctx->num_cus, colg * rowg);
    // TN_FC8_FWD_HALF=1 (round 6 A/B): a product whose tiles fill at least half the CUs takes ONE K slab -- no slab written
    // and read back, no finishing launch, and with two steps in flight the other stream's launches fill the idle half
    // (what paid for the fp16 weight gradients, DESIGN.md lesson 18): cifar_like's 2048 x 2048 -> 512 dense layer, 128 tiles
    static int half_ok = -1;
    if (half_ok < 0) {
        const char* e = getenv("TN_FC8_FWD_HALF");
        half_ok = e ? atoi(e) : 0;
    }
    if (half_ok && 2 * colg * rowg >= ctx->num_cus) S = 1;
    if (S > g.Kc / 256) S = g.Kc / 256;
    if (S < 1) S = 1;
    g.krange = cdiv(cdiv(g.Kc, S), 64) * 64;
    S = cdiv(g.Kc, g.krange);
    g.S = S;
    g.xcd = fc8_xcd_on() && (S * rowg) % 8 == 0;
    static int fin_on = -1;                // TN_FC8_FIN=0: the finishing launch also for one slab (A/B)
    if (fin_on < 0) {
        const char* e = getenv("TN_FC8_FIN");
        fin_on = e ? atoi(e) : 1;
    }
    if (S == 1 && n_out % 64 == 0 && fin_on && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)mask | (uintptr_t)dr.mask_out) & 15) == 0) {
        g.bias = b; g.mask = mask; g.out = a; g.act = act; g.prm = act_param;
        fc8_fwd_kernel<true><<<dim3(colg, 1, rowg), 256, 0, ctx->stream>>>(g, dr);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    rc = tn_scratch_get(ctx, (size_t)S * B * n_out * sizeof(float), &g.ws);
    if (rc) return rc;
    fc8_fwd_kernel<false><<<dim3(colg, S, rowg), 256, 0, ctx->stream>>>(g, dr);
    TN_LAUNCH_CHECK();
    const size_t MN = (size_t)B * n_out;
    fc8_fwd_finish_kernel<<<cdiv(MN, 256), 256, 0, ctx->stream>>>(g.ws, S, MN, n_out, b, mask, a, act, act_param, dr);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
int tn_c8_fc_fwd(tn_ctx* ctx, const void* x, const float* W, const float* b, float* a, int B, int C, int HW, int n_out,
                 int act, float act_param, const uint8_t* mask) {
    return fc8_fwd_run(ctx, x, W, b, a, B, C, HW, n_out, act, act_param, mask, Fc8Drop{});
}
// the same with the dropout mask drawn by the finishing kernel (dropout.py:12: keep = u01 >= pdrop, no rescale) and
// written to mask_out for the backward pass: the numbers of tn_dropout_mask(mask_out, B * n_out, pdrop, seed, step,
// d_step, elem0), one launch and one pass over the mask less
int tn_c8_fc_fwd_dropout(tn_ctx* ctx, const void* x, const float* W, const float* b, float* a, int B, int C, int HW,
                         int n_out, int act, float act_param, uint8_t* mask_out, float pdrop, uint64_t seed, uint32_t step,
                         const uint32_t* d_step, uint64_t elem0) {
    TN_REQUIRE(mask_out != nullptr, "tn_c8_fc_fwd_dropout: NULL mask");
    Fc8Drop dr{mask_out, pdrop, (uint32_t)seed, (uint32_t)(seed >> 32), step, d_step, elem0};
    return fc8_fwd_run(ctx, x, W, b, a, B, C, HW, n_out, act, act_param, nullptr, dr);
}

// dx16 (B, C8*HW*8) halfs = fp16(gs * dz . W^T * act'(y16)): dz (B, n_out) fp32 = d cost / d z of this layer, y16 = output
// of the layer below in x's order (NULL: none), (act, prm) its activation.  dx carries the gradient scale.
int tn_c8_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, void* dx, int B, int C, int HW, int n_out, const void* y,
                   int act, float act_param) {
    int rc = fc8_check(ctx, B, C, HW, n_out, "tn_c8_fc_dgrad");
    if (rc) return rc;
    FC8 g{};
    g.dz = dz; g.W = W; g.dx = static_cast<_Float16*>(dx); g.ya = static_cast<const _Float16*>(y);
    g.M = B; g.N = n_out; g.C = C; g.HW = HW; g.Kc = ((C + 7) / 8) * HW * 8;
    g.act = act; g.prm = act_param; g.gs = ctx->grad_scale;
    g.magic = fc8_magic(HW);
    g.Np = cdiv(n_out, 64) * 64;
    const Fc8Dz16Keep kp = fc8_dz16_keep;
    fc8_dz16_keep = Fc8Dz16Keep{};
    if (kp.ctx == ctx && kp.stream == ctx->stream && kp.dz == dz && kp.B == B && kp.n_out == n_out && kp.gs == g.gs &&
        kp.gen == ctx->scratch_gen) {
        g.dz16 = kp.dz16;                    // the weight-gradient launch right in front of this call wrote it
    } else {
        _Float16* dz16;
        rc = tn_scratch_get(ctx, (size_t)B * g.Np * sizeof(_Float16), reinterpret_cast<float**>(&dz16));
        if (rc) return rc;
        g.dz16 = dz16;
        fc8_dz16_kernel<<<cdiv((size_t)B * (g.Np / 4), 256), 256, 0, ctx->stream>>>(dz, dz16, B, n_out, g.Np, g.gs);
        TN_LAUNCH_CHECK();
    }
    fc8_dgrad_kernel<4><<<dim3(g.Kc / 64, cdiv(B, 128)), 512, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// dW (C*HW, n_out), db (n_out) fp32 from x16 and dz (fp32, rounded as fp16(gs * dz) while staged)
int tn_c8_fc_wgrad(tn_ctx* ctx, const void* x, const float* dz, float* dW, float* db, int B, int C, int HW, int n_out) {
    int rc = fc8_check(ctx, B, C, HW, n_out, "tn_c8_fc_wgrad");
    if (rc) return rc;
    FC8 g{};
    g.x = static_cast<const _Float16*>(x); g.dz = dz; g.M = B; g.N = n_out; g.C = C; g.HW = HW;
    g.Kc = ((C + 7) / 8) * HW * 8;
    g.gs = ctx->grad_scale; g.oscale = 1.f / ctx->grad_scale;
    const int kb = cdiv(g.Kc, 128), nb = cdiv(n_out, 128);
    // sample slabs: ONE block per CU (round 5; two per CU until then).  The launch is bound by its traffic, not by its
    // products (cifar_like: 4.3 GFLOP, 83 MB per launch), and every slab is written here and read back by the update:
    // cifar_like float16 step 0.3261 (8 slabs) -> 0.3175 (4) -> 0.3198 (2) ms same-box.  TN_FC8_WSLABS: half blocks per CU.
    static int half_cu = -1;
    if (half_cu < 0) {
        const char* e = getenv("TN_FC8_WSLABS");
        half_cu = e ? atoi(e) : 2;
    }
    int S = cdiv(half_cu * ctx->num_cus / 2, kb * nb);
    if (S > cdiv(B, 64)) S = cdiv(B, 64);
    if (S < 1) S = 1;
    g.krange = cdiv(cdiv(B, S), 64) * 64;
    S = cdiv(B, g.krange);
    g.S = S;
    const size_t n = (size_t)C * HW * n_out;
    // a rider waiting in the context travels as extra grid layers behind the product (it works in the tile's LDS)
    ElField rider{};
    int nrider = 0, zr = 0;
    if (ctx->rider_valid && ctx->rider_lds <= sizeof(char) * 2 * 64 * FC8_RS) {
        rider = ctx->rider;
        nrider = cdiv(rider.h * rider.w, 4);
        zr = cdiv(nrider, kb * nb);
        ctx->rider_valid = false;
    }
    // room for the rounded dz behind the slabs (one request: outside a deferral window every request starts at offset 0)
    g.Np = cdiv(n_out, 64) * 64;
    const size_t slabf = S == 1 ? 0 : (size_t)S * n + (size_t)S * n_out, slabf4 = (slabf + 63) & ~(size_t)63;
    float* scr;
    rc = tn_scratch_get(ctx, slabf4 * sizeof(float) + (size_t)B * g.Np * sizeof(_Float16), &scr);
    if (rc) return rc;
    static int keep_on = -1;               // TN_FC8_DZ16=0: the conversion stays a launch of its own (A/B)
    if (keep_on < 0) {
        const char* e = getenv("TN_FC8_DZ16");
        keep_on = e ? atoi(e) : 1;
    }
    if (keep_on) {
        g.dz16w = reinterpret_cast<_Float16*>(scr + slabf4);
        fc8_dz16_keep = {ctx, ctx->stream, dz, g.dz16w, B, n_out, ctx->grad_scale, ctx->scratch_gen};
    }
    if (S == 1) {
        g.ws = dW; g.dbws = db;
        // (channels beyond C own no row of dW: nothing to clear)
        fc8_wgrad_kernel<<<dim3(kb, nb, 1 + zr), 256, 0, ctx->stream>>>(g, rider, nrider);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    g.ws = scr;
    g.dbws = g.ws + (size_t)S * n;
    fc8_wgrad_kernel<<<dim3(kb, nb, S + zr), 256, 0, ctx->stream>>>(g, rider, nrider);
    TN_LAUNCH_CHECK();
    rc = tn_red_push(ctx, g.ws, dW, (uint32_t)n, (uint32_t)S, (uint32_t)n, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, g.dbws, db, (uint32_t)n_out, (uint32_t)S, (uint32_t)n_out, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

}  // extern "C"
