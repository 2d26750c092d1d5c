// MATMUL 'bf16x3' (opt-in, training param; default off): the fully-connected products of theanet/layer/hidden.py:30 and
// their gradients (layer.py:83) on the bf16 matrix pipe with fp32-grade accuracy.
//
// Every fp32 operand is split EXACTLY into three bf16 terms while its tile is staged into LDS (8 + 8 + 8 mantissa
// bits: x = x0 + x1 + x2, x0 = top half of x, x1 = top half of x - x0, ...), and a product is the six partial products
// a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) on v_mfma_f32_32x32x16_bf16, small terms first, fp32 accumulation;
// the three dropped terms are below 2^-24 of the product.  Measured against float64 (tools/probe/bf16x3.hip,
// profiles/r02_probe_bf16x3.txt): max relative error 5.8e-5 against 4.0e-5 for the fp32 MFMA -- the same tests pass at
// the same tolerances -- but NOT the same bits: the headline configuration stays on the exact fp32 MFMA
// (gemm.hip) and this mode is reported separately.  Six bf16 MFMAs do the work of sixteen fp32 ones (2.5 PFLOP/s / 6
// = 417 TFLOP/s against 157).
//
// One kernel for the three products of a layer: C (M x N) = A (M x K) . B (K x N), each operand k-contiguous or
// k-major in global memory.  Tiles go to LDS in the orientation they are stored in (no transposition while staging);
// a k-contiguous operand is read with ds_read_b128, a k-major one through gfx950's transposing read
// (ds_read_b64_tr_b16).  Block = 128 x 64 x 32, four waves of 64 x 32, one 48 KB LDS stage (three blocks per CU) with
// the next stage prefetched into registers; split-K slabs for the weight gradient (finished by the step's reduction launch / lazy update).
#include "common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short b3_short4 __attribute__((ext_vector_type(4)));
typedef unsigned b3_u4 __attribute__((ext_vector_type(4)));

#define B3_BM 128
#define B3_BN 64
#define B3_BK 32
#define B3_RKC 80           // bytes per row of a k-contiguous tile image (32 bf16 + 16: 16-byte slots of 16 rows all distinct)
#define B3_RA 320           // bytes per k-row of a k-major A image (128 bf16 + 64: = 64 mod 256)
#define B3_RB 192           // ... of a k-major B image (64 bf16 + 64)

struct B3Args {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    int kchunk;              // K range of a slab (multiple of 32); blockIdx.z = slab
    int epi;                 // 0: plain store to C + slab * M * N; 1: bias + act + mask; 2: * act'(prev_a) * mask
    const float* bias; const uint8_t* mask; const float* prev_a;
    int act; float prm;
    float* colsum;           // != NULL: column sums of B over the slab's k -> colsum[slab * N + n] (blocks of row tile 0)
};

// exact three-way split of 8 floats: planes h / m / l as 8 bf16 (16 bytes) each
__device__ __forceinline__ void b3_split8(const float (&v)[8], b3_u4& h, b3_u4& m, b3_u4& l) {
    unsigned uh[8], um[8], ul[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned ua = __float_as_uint(v[j]);
        const float r1 = v[j] - __uint_as_float(ua & 0xffff0000u);
        const unsigned u1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
        uh[j] = ua; um[j] = u1; ul[j] = __float_as_uint(r2);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {           // high halves of two dwords -> one dword
        h[d] = __builtin_amdgcn_perm(uh[2 * d + 1], uh[2 * d], 0x07060302u);
        m[d] = __builtin_amdgcn_perm(um[2 * d + 1], um[2 * d], 0x07060302u);
        l[d] = __builtin_amdgcn_perm(ul[2 * d + 1], ul[2 * d], 0x07060302u);
    }
}

__device__ __forceinline__ bf16x8 b3_tr(const char* p, int rs) {
    const b3_short4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) b3_short4*)p);
    const b3_short4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) b3_short4*)(p + 4 * rs));
    typedef short s8 __attribute__((ext_vector_type(8)));
    const s8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_b3_kernel(B3Args g) {
    // per stage: A image 3 planes, B image 3 planes
    constexpr int APL = AKC ? B3_BM * B3_RKC : B3_BK * B3_RA;        // bytes of one A plane
    constexpr int BPL = BKC ? B3_BN * B3_RKC : B3_BK * B3_RB;
    constexpr int STAGE = 3 * (APL + BPL);
    __shared__ __attribute__((aligned(16))) char lds[STAGE];        // one buffer (48 KB: three blocks per CU), the next stage waits in registers
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    // XCD-aware: the column tiles of a row panel (and a split-K slab) share an L2
    const int m0 = blockIdx.y * B3_BM, n0 = blockIdx.x * B3_BN, slab = blockIdx.z;
    const int kbeg = slab * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;

    // ---- staging units of this thread: 8 consecutive elements along the contiguous dimension --------------
    // k-contiguous: unit = (row, k octet): A 128 x 4 = 512 units (two per thread), B 64 x 4 = 256 (one per thread)
    // k-major     : unit = (k, octet of rows): A 32 x 16 = 512, B 32 x 8 = 256
    float ra[2][2][8], rb[2][8];          // [stage parity]: operands are fetched TWO stages ahead of the matrix work
    auto aunit = [&](int u, int& row, int& k) __attribute__((always_inline)) {
        const int id = t + 256 * u;
        if (AKC) { row = id >> 2; k = (id & 3) * 8; }
        else { k = id >> 4; row = (id & 15) * 8; }
    };
    auto bunit = [&](int& col, int& k) __attribute__((always_inline)) {
        if (BKC) { col = t >> 2; k = (t & 3) * 8; }
        else { k = t >> 3; col = (t & 7) * 8; }
    };
    // Branch-free: both 16-byte loads of a unit always happen, from clamped addresses, and what lies outside the
    // operand is zeroed by selects afterwards (K, M and N are multiples of 4: a float4 is inside or outside as a
    // whole).  With the padding handled by branches hipcc put a wait between the units' loads: three HBM latencies
    // per stage in a row (67 us for mnist's fc1 forward against 33 on the fp32 path).
    auto load8 = [&](const float* base, long ld, bool kc, int row, int k, int rmax, float (&v)[8]) __attribute__((always_inline)) {
        float4 x, y;
        bool okx, oky;
        if (kc) {      // elements k .. k+7 of row `row`
            const float* p = base + (long)min(row, rmax - 1) * ld;
            x = *reinterpret_cast<const float4*>(p + min(k, g.K - 4));
            y = *reinterpret_cast<const float4*>(p + min(k + 4, g.K - 4));
            okx = k + 4 <= kend; oky = k + 8 <= kend;
        } else {       // rows row .. row+7 at reduction index k
            const float* p = base + (long)min(k, g.K - 1) * ld;
            x = *reinterpret_cast<const float4*>(p + min(row, rmax - 4));
            y = *reinterpret_cast<const float4*>(p + min(row + 4, rmax - 4));
            okx = k < kend && row + 4 <= rmax; oky = k < kend && row + 8 <= rmax;
        }
        v[0] = okx ? x.x : 0.f; v[1] = okx ? x.y : 0.f; v[2] = okx ? x.z : 0.f; v[3] = okx ? x.w : 0.f;
        v[4] = oky ? y.x : 0.f; v[5] = oky ? y.y : 0.f; v[6] = oky ? y.z : 0.f; v[7] = oky ? y.w : 0.f;
    };
    auto gload = [&](int ks, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int row, k;
            aunit(u, row, k);
            load8(g.A, g.lda, AKC, m0 + row, ks + k, g.M, ra[P][u]);
        }
        int col, k;
        bunit(col, k);
        load8(g.B, g.ldb, BKC, n0 + col, ks + k, g.N, rb[P]);
    };
    float csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] = 0.f;
    auto lstore = [&](auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        char* sa = lds;
        char* sb = sa + 3 * APL;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int row, k;
            aunit(u, row, k);
            b3_u4 h, m, l;
            b3_split8(ra[P][u], h, m, l);
            char* p = sa + (AKC ? row * B3_RKC + k * 2 : k * B3_RA + row * 2);
            *reinterpret_cast<b3_u4*>(p) = h;
            *reinterpret_cast<b3_u4*>(p + APL) = m;
            *reinterpret_cast<b3_u4*>(p + 2 * APL) = l;
        }
        int col, k;
        bunit(col, k);
        b3_u4 h, m, l;
        b3_split8(rb[P], h, m, l);
        char* p = sb + (BKC ? col * B3_RKC + k * 2 : k * B3_RB + col * 2);
        *reinterpret_cast<b3_u4*>(p) = h;
        *reinterpret_cast<b3_u4*>(p + BPL) = m;
        *reinterpret_cast<b3_u4*>(p + 2 * BPL) = l;
        if (g.colsum) {
#pragma unroll
            for (int j = 0; j < 8; ++j) csum[j] += rb[P][j];
        }
    };

    // three accumulators per tile, one per order of magnitude of the partial products: six MFMAs in a row on ONE
    // accumulator are a dependent chain (each waits for the one before), three chains of 1 / 2 / 3 interleave
    f32x16 acc[2], accm[2], accs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accm[i][r] = 0.f; accs[i][r] = 0.f; }

    // operand read offsets of this lane inside a plane (first 16-deep step; + 32 bytes / + 16 k-rows for the second)
    const int grp = lane >> 4, r4 = (lane >> 2) & 3, q4 = lane & 3;
    const int a_rd = AKC ? (wm + l31) * B3_RKC + hi * 16 : (8 * (grp >> 1) + r4) * B3_RA + (wm + 16 * (grp & 1) + 4 * q4) * 2;
    const int b_rd = BKC ? (wn + l31) * B3_RKC + hi * 16 : (8 * (grp >> 1) + r4) * B3_RB + (wn + 16 * (grp & 1) + 4 * q4) * 2;

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    gload(kbeg, P0{});
    gload(kbeg + B3_BK, P1{});
    lstore(P0{});
    __syncthreads();
    auto body = [&](int ks, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        gload(ks + 2 * B3_BK, Pc);              // (beyond the slab: clamped loads of zeros-to-be, never stored)
        __builtin_amdgcn_sched_barrier(0);      // (hipcc sinks these loads below the MFMAs otherwise: nothing left to hide them)
        const char* sa = lds;
        const char* sb = sa + 3 * APL;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[2][3], b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (AKC) a[i][pl] = *reinterpret_cast<const bf16x8*>(sa + pl * APL + a_rd + i * 32 * B3_RKC + s * 32);
                    else a[i][pl] = b3_tr(sa + pl * APL + a_rd + i * 64 + s * 16 * B3_RA, B3_RA);
                }
                if (BKC) b[pl] = *reinterpret_cast<const bf16x8*>(sb + pl * BPL + b_rd + s * 32);
                else b[pl] = b3_tr(sb + pl * BPL + b_rd + s * 16 * B3_RB, B3_RB);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[0], accs[i], 0, 0, 0);
                accm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[0], accm[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0], acc[i], 0, 0, 0);
                accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[1], accs[i], 0, 0, 0);
                accm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[1], accm[i], 0, 0, 0);
                accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[2], accs[i], 0, 0, 0);
            }
        }
        __syncthreads();                     // everybody has read the stage
        if (ks + B3_BK < kend) lstore(std::integral_constant<int, P ^ 1>{});
        __syncthreads();
    };
    for (int ks = kbeg; ks < kend; ks += 2 * B3_BK) {
        body(ks, P0{});
        if (ks + B3_BK < kend) body(ks + B3_BK, P1{});
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] += accm[i][r] + accs[i][r];      // small terms first

    // ---- column sums of B (the bias gradient of a weight-gradient product): per n, over this slab's k, in k order ----
    if (!BKC && g.colsum && blockIdx.y == 0) {
        float* red = reinterpret_cast<float*>(lds);          // [32 k-rows][64 columns] (k-major B: thread = (k, column octet))
        int col, k;
        bunit(col, k);
#pragma unroll
        for (int j = 0; j < 8; ++j) red[k * 64 + col + j] = csum[j];
        __syncthreads();
        if (t < 64) {
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += red[r * 64 + t];
            if (n0 + t < g.N) g.colsum[(size_t)slab * g.N + n0 + t] = s;
        }
    }

    // ---- epilogue: lane = column n (32 consecutive per half-wave), registers = rows ----
    float* C = g.C + (g.epi == 0 ? (size_t)slab * g.M * g.N : 0);
    const int n = n0 + wn + l31;
    if (n >= g.N) return;
    const float bias = (g.epi == 1 && g.bias) ? g.bias[n] : 0.f;
    const float tie = g.prm > 0.f ? 1.f + g.prm : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m >= g.M) continue;
            const size_t o = (size_t)m * g.ldc + n;
            float v = acc[i][r];
            if (g.epi == 1) {
                v += bias;
                v = g.act == TN_ACT_LEAKY ? fmaxf(0.f, v) + fminf(0.f, v) * g.prm : tn_act_fwd(v, g.act, g.prm);
                if (g.mask) v = g.mask[o] ? v : 0.f;
            } else if (g.epi == 2) {
                if (g.prev_a) {
                    const float y = g.prev_a[o];
                    v *= g.act == TN_ACT_LEAKY ? (y > 0.f ? 1.f : (y < 0.f ? g.prm : tie)) : tn_act_grad_from_out(y, g.act, g.prm);
                }
                if (g.mask) v = g.mask[o] ? v : 0.f;
            }
            C[o] = v;
        }
}

static bool b3_al(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <bool AKC, bool BKC>
static int b3_launch(tn_ctx* ctx, const B3Args& g, int S) {
    gemm_b3_kernel<AKC, BKC><<<dim3(cdiv(g.N, B3_BN), cdiv(g.M, B3_BM), S), 256, 0, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// 1 if the three products of a (B, n_in) -> n_out layer run on this path (wide enough, 16-byte aligned rows)
int tn_b3_fc_ok(const float* x, const float* W, int B, int n_in, int n_out) {
    return n_out > 16 && (n_in & 3) == 0 && (n_out & 3) == 0 && n_in >= 32 && B >= 4 && (B & 3) == 0 && b3_al(x) && b3_al(W);
}

int tn_b3_fc_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B, int n_in, int n_out, int act,
                 float prm, const uint8_t* mask) {
    B3Args g{};
    g.A = x; g.B = W; g.C = a; g.M = B; g.N = n_out; g.K = n_in; g.lda = n_in; g.ldb = n_out; g.ldc = n_out;
    g.kchunk = cdiv(n_in, B3_BK) * B3_BK; g.epi = 1; g.bias = b; g.mask = mask; g.act = act; g.prm = prm;
    return b3_launch<true, false>(ctx, g, 1);
}

int tn_b3_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int B, int n_in, int n_out, const float* prev_a,
                   int act, float prm, const uint8_t* mask) {
    B3Args g{};
    g.A = dz; g.B = W; g.C = dx; g.M = B; g.N = n_in; g.K = n_out; g.lda = n_out; g.ldb = n_out; g.ldc = n_in;
    g.kchunk = cdiv(n_out, B3_BK) * B3_BK; g.epi = 2; g.prev_a = prev_a; g.mask = mask; g.act = act; g.prm = prm;
    return b3_launch<true, true>(ctx, g, 1);
}

// dW (n_in, n_out) = x^T . dz, db = column sums of dz; S sample slabs into ws ([S][n_in * n_out] then [S][n_out]),
// recorded for the step's reduction (tn_red_push) unless S == 1
int tn_b3_fc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int B, int n_in, int n_out, float* ws,
                   int S) {
    B3Args g{};
    g.A = x; g.B = dz; g.M = n_in; g.N = n_out; g.K = B; g.lda = n_in; g.ldb = n_out; g.ldc = n_out;
    g.kchunk = cdiv(cdiv(B, S), B3_BK) * B3_BK;
    const int Sx = cdiv(B, g.kchunk);
    g.epi = 0;
    const size_t MN = (size_t)n_in * n_out;
    if (Sx == 1) {
        g.C = dW; g.colsum = db;
        return b3_launch<false, false>(ctx, g, 1);
    }
    g.C = ws; g.colsum = ws + (size_t)S * MN;
    int rc = b3_launch<false, false>(ctx, g, Sx);
    if (rc) return rc;
    rc = tn_red_push(ctx, ws, dW, (uint32_t)MN, (uint32_t)Sx, (uint32_t)MN, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, g.colsum, db, (uint32_t)n_out, (uint32_t)Sx, (uint32_t)n_out, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}
