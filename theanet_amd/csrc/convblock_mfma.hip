// Matrix-core backward of a conv(3x3) + act + 2x2-max-pool block with few channels and a few
// dozen filters (mnist.prms conv2: 20 x 4 x 3 x 3).  Same contract as convblock.hip; this is the
// path tn_convblock_bwd takes when the shape fits.
//
// One 64-lane WAVE owns one image at a time and never synchronises with other waves inside
// the image loop: its x tile, dz tile and dx tile live in a private LDS slice, ordered only by
// the in-order LDS pipe.  All three products run on v_mfma_f32_16x16x4_f32 (exact f32):
//
//   conv recompute  z[pix][k]     = sum_ckk patch[pix][ckk] * Wf[k][ckk]       M=pix  N=k    red=ckk
//   wgrad           dWf[k][ckk]  += sum_pix dz[k][pix] * patch[pix][ckk]       M=k    N=ckk  red=pix
//   dgrad           T[ckk][x']    = sum_k  Wf[k][ckk] * dz[k][y][x']           M=ckk  N=x'   red=k
//                   dx[c][y+a][x'+b] += T[(c,a,b)][x']: b by DPP lane shifts, a by a rolling row sum
//
// ckk = (c, a, b) indexes the FLIPPED filter Wf[k][c][a][b] = W[k][c][2-a][2-b], so that
// z[k][y][x] = sum Wf * x[c][y+a][x+b] (true convolution, theanet/layer/convpool.py:54-72).
// Pixels are enumerated pooling window by pooling window (4 per window): the conv result of a
// 16-pixel tile leaves the matrix core with one window's 4 outputs in the 4 accumulator
// registers of a lane, so act / max / tie mask (Theano MaxPoolGrad: every tie gets the
// gradient) are in-lane, and the same registers are the A operand of the wgrad product -- dz
// goes to LDS only for the dgrad product, never to HBM.  The filter operands of all three
// products stay in registers for the whole kernel; the next image's x and g are prefetched
// into registers while the current one is computed.  HBM traffic: x, g read once, dx written once.
#include <cstdlib>

#include "common.h"

int tn_conv_wgrad_finish(tn_ctx* ctx, const float* partial, const float* dbpartial, float* dW,
                         float* db, int nblk, int K, int C, int f);

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CM_XR 12        // prefetch registers per lane for x (C*H*W <= 64*CM_XR) and for g (K*Hp*Wp)

struct CmGeom {
    int N, H, Wd, K, pad, Ho, Wo, Hp, Wp;
    int Wx, xplane;      // x / dx tile: C planes of (Ho+3) rows of Wx = Wo+3 floats (one spare row and col)
    int xf;              // floats per x tile incl. the two constant cells (multiple of 4)
    int PT;              // 16-pixel tiles per image = ceil(Hp*Wp / 4)
    int npixp;           // dz row stride (floats), npixp/4 odd
    int dzf;             // floats of the dz tile
    int tabf;            // floats of the per-block tables
    int wavef;           // floats of one wave's slice
    int dbg;             // ablation (TN_CM_DBG): 1 no wgrad, 2 no dgrad, 8 no conv product, 16 cycle stamps -> db
};

__device__ __forceinline__ void cm_wave_sync() {
    // LDS operations of one wave complete in order; only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ f32x4 cm_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// value of the lane N places to the left inside the 16-lane DPP row, 0 beyond its start
template <int N>
__device__ __forceinline__ float cm_shr(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, false));
}

// n / d for 0 <= n < 2^16 given rd = 1/d to 1 ulp (exact: (n + .5) / d stays >= .5/d away from
// the next integer, far more than the rounding of the product)
__device__ __forceinline__ int cm_div(int n, float rd) { return __float2int_rz(((float)n + .5f) * rd); }

template <int ACT>
__device__ __forceinline__ float cm_act(float z, int act, float prm) {
    if (ACT == TN_ACT_LEAKY) return fmaxf(0.f, z) + fminf(0.f, z) * prm;
    return tn_act_fwd(z, act, prm);
}
template <int ACT>
__device__ __forceinline__ float cm_actg(float a, int act, float prm) {
    if (ACT == TN_ACT_LEAKY) return a > 0.f ? 1.f : (a < 0.f ? prm : (prm > 0.f ? 1.f + prm : 0.f));
    return tn_act_grad_from_out(a, act, prm);
}

template <int C, int KS2, int ACT>      // KS2 = ceil(K / 4): reduction steps of the dgrad product
__global__ __launch_bounds__(256, 2) void convblock_bwd_mfma(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
    const float* __restrict__ g, float* __restrict__ dx, float* __restrict__ partial,
    float* __restrict__ dbpartial, CmGeom q, int act, float prm, float* __restrict__ tdbg) {
    constexpr int F = 3, FF = 9, CKK = C * FF;
    constexpr int KS1 = (CKK + 3) / 4;        // reduction steps of the conv product
    constexpr int NT = (CKK + 16) / 16;       // 16-wide ckk tiles incl. the bias column ckk == CKK
    constexpr int NKT = (KS2 + 3) / 4;        // 16-wide filter tiles
    constexpr int NTD = 3;                    // dgrad product rows: (c = row>>2 & 3, ab = 4*tile + (row & 3))
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lo = lane & 15, qd = lane >> 4;
    const long long t0_ = __builtin_readcyclecounter();
    int ts_ = 0;
#define CM_STAMP() if (tdbg && blockIdx.x == (q.dbg >> 8) && threadIdx.x == 0 && ts_ < 20) tdbg[ts_++] = (float)(__builtin_readcyclecounter() - t0_)
    const int K = q.K, HW = q.H * q.Wd, CHW = C * HW, HpWp = q.Hp * q.Wp, KHW = K * HpWp;

    // the first image starts travelling before anything else
    const int gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
    float xr[CM_XR], gr[CM_XR];
    // the image index is wave-uniform: scalar base + per-lane 32-bit offset addressing
#define CM_PREFETCH(N_)                                                                     \
    {                                                                                       \
        const float* xp_ = x + (size_t)(N_) * CHW;                                          \
        const float* gp_ = g + (size_t)(N_) * KHW;                                          \
        _Pragma("unroll") for (int j = 0; j < CM_XR; ++j) {                                 \
            xr[j] = xp_[min(lane + 64 * j, CHW - 1)];                                       \
            gr[j] = gp_[min(lane + 64 * j, KHW - 1)];                                       \
        }                                                                                   \
    }
    int n = __builtin_amdgcn_readfirstlane(gw);
    if (n < q.N) CM_PREFETCH(n);

    int* wtab = reinterpret_cast<int*>(sm);           // [4*PT] window -> (tile offset << 4) | valid bits
    float* wbase = sm + q.tabf + wv * q.wavef;
    float* sx = wbase;                                // x tile (zero padded)
    float* sdx = sx + q.xf;                           // dx tile; doubles as the g tile during phase 1
    float* sg = sdx;
    float* sdz = sdx + q.xf;                          // [4*ks2][npixp], pixels in window order
    const int ZERO = q.xf - 4, ONE = q.xf - 3;        // constant cells behind the x tile
    const int DUMMY = q.xf - 2;                       // write-only cell for out-of-range lanes

    for (int i = threadIdx.x; i < 4 * q.PT; i += 256) {
        int e = 0;
        if (i < HpWp) {
            const int wy = i / q.Wp, wx = i - wy * q.Wp;
            const bool vx = 2 * wx + 1 < q.Wo, vy = 2 * wy + 1 < q.Ho;
            e = ((2 * wy * q.Wx + 2 * wx) << 4) | 1 | (vx ? 2 : 0) | (vy ? 4 : 0) | (vx && vy ? 8 : 0);
        }
        wtab[i] = e;
    }
    // x tile: the pads stay zero for the whole kernel.  dz tile: phase 1 rewrites every pixel cell of
    // the rows < K for each image; the 4 pad cells of each row and the rows K..4*KS2-1 stay zero.
    for (int i = lane * 4; i < q.xf; i += 256) *reinterpret_cast<float4*>(sx + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane; i < 4 * KS2 * 4; i += 64) sdz[(i >> 2) * q.npixp + 16 * q.PT + (i & 3)] = 0.f;
    for (int i = K * q.npixp + lane; i < q.dzf; i += 64) sdz[i] = 0.f;
    CM_STAMP();
    __syncthreads();
    CM_STAMP();
    if (lane == 0) sx[ONE] = 1.f;

    // ---- loop-invariant operands ------------------------------------------------------------
    auto ckk_off = [&](int ckk) {
        const int c = ckk / FF, ab = ckk - c * FF, a = ab / F, bb = ab - a * F;
        return c * q.xplane + a * q.Wx + bb;
    };
    auto wf = [&](int k, int c, int ab) -> float {       // flipped filter, 0 outside
        const bool ok = k < K && c < C && ab < FF;
        const float v = W[((size_t)min(k, K - 1) * C + min(c, C - 1)) * FF + (FF - 1 - min(ab, FF - 1))];
        return ok ? v : 0.f;
    };
    float Bw1[NKT][KS1];       // conv product B operand: Wf[k = 16kt+lo][ckk = 4s+qd]
    int off1[KS1];             // conv product A operand: tile offset of ckk = 4s+qd
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        const int ckk = 4 * s + qd;
        off1[s] = ckk < CKK ? ckk_off(ckk) : 0;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) Bw1[kt][s] = wf(16 * kt + lo, ckk < CKK ? ckk / FF : C, ckk % FF);
    }
    int off3[NT], msk3[NT];    // wgrad B operand: ckk = 16nt+lo (bias column reads the 1.0 cell)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ckk = 16 * nt + lo;
        off3[nt] = ckk < CKK ? ckk_off(ckk) : (ckk == CKK ? ONE : ZERO);
        msk3[nt] = ckk < CKK ? -1 : 0;
    }
    float Aw2[NTD][KS2];       // dgrad product A operand: Wf[k = 4s+qd][c = lo>>2][ab = 4mt + (lo&3)]
#pragma unroll
    for (int mt = 0; mt < NTD; ++mt)
#pragma unroll
        for (int s = 0; s < KS2; ++s) Aw2[mt][s] = wf(4 * s + qd, lo >> 2, 4 * mt + (lo & 3));
    float bk[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) bk[kt] = b[min(16 * kt + lo, K - 1)];
    int tabr[CM_XR];           // image element lane+64j -> offset in the x / dx tile
    const float rHW = __builtin_amdgcn_rcpf((float)HW), rWd = __builtin_amdgcn_rcpf((float)q.Wd);
#pragma unroll
    for (int j = 0; j < CM_XR; ++j) {
        const int i = min(lane + 64 * j, CHW - 1);
        const int c = cm_div(i, rHW), r = i - c * HW, y = cm_div(r, rWd), xx = r - y * q.Wd;
        tabr[j] = (lane + 64 * j < CHW) ? c * q.xplane + (y + q.pad) * q.Wx + xx + q.pad : DUMMY;
    }
    // Pin the operands: they are complete here, so the tile loops never wait on the vector-memory
    // counter and the image prefetch below stays in flight across them.
    CM_STAMP();
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    CM_STAMP();
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) asm volatile("" : "+v"(Bw1[kt][s]));
#pragma unroll
    for (int mt = 0; mt < NTD; ++mt)
#pragma unroll
        for (int s = 0; s < KS2; ++s) asm volatile("" : "+v"(Aw2[mt][s]));
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) asm volatile("" : "+v"(bk[kt]));

    f32x4 accW[NKT][NT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accW[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    CM_STAMP();      // 0: setup done

    const int dyo = ((lo >> 1) & 1) * q.Wx + (lo & 1);      // sub-pixel offset of conv-product row lo
    const float tie = prm > 0.f ? 1.f + prm : 0.f;          // leaky slope at an exact 0
    // dgrad product column lo = tile column x': the dz pixel (y, x') in window order, or the zero pad
    const int dcol = lo < q.Wo ? 4 * (lo >> 1) + (lo & 1) : 16 * q.PT;
    const int dmsk = lo < q.Wo ? -1 : 0;
    for (; n < q.N; n += nw) {
        // branch-free: lanes beyond the image hold a copy of its last element and store it to a
        // dummy cell (x) / the last cell (g)
#pragma unroll
        for (int j = 0; j < CM_XR; ++j) {
            sx[tabr[j]] = xr[j];
            sg[min(lane + 64 * j, KHW - 1)] = gr[j];
        }
        if (n + nw < q.N) CM_PREFETCH(n + nw);
        cm_wave_sync();
        CM_STAMP();  // image staged
        // ---- conv recompute -> dz (registers + LDS) -> wgrad ---------------------------------
        float av[KS1];
        {
            const int poff = (wtab[lo >> 2] >> 4) + dyo;
#pragma unroll
            for (int s = 0; s < KS1; ++s) av[s] = sx[poff + off1[s]];
        }
#pragma unroll 1
        for (int t = 0; t < q.PT; ++t) {
            f32x4 z[NKT];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) z[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(q.dbg & 8)) {
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) z[kt] = cm_mfma(av[s], Bw1[kt][s], z[kt]);
            }
            // operands of the next tile's conv product and of this tile's wgrad product: issued
            // now, they arrive while the matrix core and the epilogue below are busy
            {
                const int tn = min(t + 1, q.PT - 1);
                const int poff = (wtab[4 * tn + (lo >> 2)] >> 4) + dyo;
#pragma unroll
                for (int s = 0; s < KS1; ++s) av[s] = sx[poff + off1[s]];
            }
            // lane (lo = filter, qd = window 4t+qd): the 4 registers are the window's 4 outputs
            const int wi = 4 * t + qd;
            const int e2 = wtab[wi];
            const int woff = e2 >> 4;
            float bv[4][NT];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pw = woff + (r >> 1) * q.Wx + (r & 1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[r][nt] = sx[off3[nt] + (pw & msk3[nt])];
            }
            float dzv[NKT][4];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const int k = 16 * kt + lo;
                float gv = sg[min(k, K - 1) * HpWp + min(wi, HpWp - 1)];
                gv = k < K ? gv : 0.f;
                float a[4];
                float m = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[r] = cm_act<ACT>(z[kt][r] + bk[kt], act, prm);
                    m = (e2 >> r & 1) ? fmaxf(m, a[r]) : m;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float gp;
                    if (ACT == TN_ACT_LEAKY) {
                        gp = a[r] < 0.f ? prm : tie;
                        gp = a[r] > 0.f ? 1.f : gp;
                    } else {
                        gp = tn_act_grad_from_out(a[r], act, prm);
                    }
                    const float sel = ((e2 >> r & 1) && a[r] == m) ? gv : 0.f;
                    dzv[kt][r] = sel * gp;
                }
                if (k < K)
                    *reinterpret_cast<float4*>(sdz + k * q.npixp + 16 * t + 4 * qd) =
                        make_float4(dzv[kt][0], dzv[kt][1], dzv[kt][2], dzv[kt][3]);
            }
            if (!(q.dbg & 1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        accW[kt][nt] = cm_mfma(dzv[kt][r], bv[r][nt], accW[kt][nt]);
            }
        }
        CM_STAMP();  // conv + wgrad done
        if (dx && !(q.dbg & 2)) {
            // ---- dgrad: one dz row y per step.  T[(c,a,b)][x'] = sum_k Wf[k][c][a][b] dz[k][y][x']
            // on the matrix core; lane (x', c) then owns all 9 taps of its channel: the b shifts
            // are DPP lane shifts inside the 16-lane row, the a shifts a rolling 3-row sum, so every
            // dx row is produced complete, in registers, and stored once.
            cm_wave_sync();
            float R1 = 0.f, R2 = 0.f;
            float* dxo = sdx + qd * q.xplane + lo;
            const bool dlive = (qd < C) && (lo < q.Wo + 2);
            float bcur[KS2];
            {
                const float* dzp = sdz + qd * q.npixp + dcol;
#pragma unroll
                for (int s = 0; s < KS2; ++s) bcur[s] = dzp[4 * s * q.npixp];
            }
#pragma unroll 1
            for (int y = 0; y < q.Ho; ++y) {
                // next row's operands travel while this row is on the matrix core
                const int yn = min(y + 1, q.Ho - 1);
                const int prow = 4 * (yn >> 1) * q.Wp + 2 * (yn & 1);
                const float* dzp = sdz + qd * q.npixp + dcol + (prow & dmsk);
                float bnxt[KS2];
#pragma unroll
                for (int s = 0; s < KS2; ++s) bnxt[s] = dzp[4 * s * q.npixp];
                f32x4 T[NTD];
#pragma unroll
                for (int mt = 0; mt < NTD; ++mt) T[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS2; ++s)
#pragma unroll
                    for (int mt = 0; mt < NTD; ++mt) T[mt] = cm_mfma(Aw2[mt][s], bcur[s], T[mt]);
#pragma unroll
                for (int s = 0; s < KS2; ++s) bcur[s] = bnxt[s];
                // T[mt][r] <-> tap ab = 4mt + r = 3a + b
                const float S0 = T[0][0] + cm_shr<1>(T[0][1]) + cm_shr<2>(T[0][2]);
                const float S1 = T[0][3] + cm_shr<1>(T[1][0]) + cm_shr<2>(T[1][1]);
                const float S2 = T[1][2] + cm_shr<1>(T[1][3]) + cm_shr<2>(T[2][0]);
                const float out = S0 + R1;
                R1 = S1 + R2;
                R2 = S2;
                if (dlive) dxo[y * q.Wx] = out;
            }
            if (dlive) {
                dxo[q.Ho * q.Wx] = R1;
                dxo[(q.Ho + 1) * q.Wx] = R2;
            }
            cm_wave_sync();
            float* dxp = dx + (size_t)n * CHW;
#pragma unroll
            for (int j = 0; j < CM_XR; ++j) {
                const int i = lane + 64 * j;
                if (i < CHW) dxp[i] = sdx[tabr[j]];
            }
        }
        cm_wave_sync();
        CM_STAMP();  // dgrad done
    }
#undef CM_PREFETCH
    // ---- one partial slab per block: sum the 4 waves' accumulators in wave order ---------------
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) wbase[((kt * NT + nt) * 4 + r) * 64 + lane] = accW[kt][nt][r];
    __syncthreads();
    for (int e = threadIdx.x; e < K * (CKK + 1); e += 256) {
        const int k = e / (CKK + 1), ckk = e - k * (CKK + 1);
        // accumulator (kt, nt): rows = filters 4*qd + r, cols = ckk-in-tile
        const int idx = (((k >> 4) * NT + (ckk >> 4)) * 4 + (k & 3)) * 64 + ((k & 15) >> 2) * 16 + (ckk & 15);
        const float* src = sm + q.tabf + idx;
        const float s = ((src[0] + src[q.wavef]) + src[2 * q.wavef]) + src[3 * q.wavef];
        if (ckk < CKK)
            partial[(size_t)blockIdx.x * K * CKK + k * CKK + ckk] = s;
        else
            dbpartial[(size_t)blockIdx.x * K + k] = s;
    }
    CM_STAMP();      // slab written
#undef CM_STAMP
}

// ---------------------------------------------------------------------------------------------
// Backward from the forward's pooling mask (tn_convpool_fwd_mask): dz[k][window][r] =
// mask bit r ? g * act'(y) : 0 is formed while the prefetched g / y / mask registers are staged
// into LDS -- no conv recompute, no epilogue.  The rest is the wgrad and dgrad products above.
// ---------------------------------------------------------------------------------------------
// W44: the weight-gradient product on v_mfma_f32_4x4x1_16b_f32 with the CBSZ / ABID broadcast (round 4).  20 filters x 37
// columns sit badly in 16 x 16 tiles (2 x 3 of them: 47 % of the lanes useful).  The 16-block instruction with
// CBSZ = n shares the A block (group base + ABID) among the 2^n blocks of a group (tools/probe/mfma44_cbsz.hip):
//   main product (CBSZ 3): the 64 lanes are 2 pixels x 32 columns; lane (g2, a8, r4) supplies dz[filter 4 a8 + r4] of its
//     pixel as A and x[column 4 a8 + r4] of its pixel as B; instruction ABID = a adds the 4 x 32 outer products of
//     filters 4a .. 4a+3 for both pixels -- ceil(K / 4) instructions per pixel pair, every lane useful;
//   remainder (CBSZ 1, only when C*9 + 1 > 32): 8 pixels x 8 columns (taps 32 .. 35, the bias column, 3 idle);
//     register m supplies filters 8m .. 8m+7, instruction (m, ABID) adds the 4 x 8 outer products of 8 pixels.
// 25 instructions of 8.6 cycles per 8 pixels for mnist.prms conv2 instead of 12 of 32: 27 cycles per pixel against 48
// (useful: 25).  The per-lane halves (2 pixels / 8 pixels) meet in the final slab reduction.
template <int C, int KS2, int ACT, bool W44>
__global__ __launch_bounds__(256, 2) void convblock_bwd_mask_mfma(
    const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ g,
    const float* __restrict__ y, const uint8_t* __restrict__ mask, float* __restrict__ dx,
    float* __restrict__ partial, float* __restrict__ dbpartial, CmGeom q, int act, float prm,
    float* __restrict__ tdbg) {
    constexpr int F = 3, FF = 9, CKK = C * FF;
    constexpr int NKT = (KS2 + 3) / 4;        // 16-wide filter tiles
    constexpr int NT = (CKK + 16) / 16;       // 16-wide ckk tiles incl. the bias column ckk == CKK
    constexpr int NTD = 3;                    // dgrad product rows: (c = row>>2 & 3, ab = 4*tile + (row & 3))
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lo = lane & 15, qd = lane >> 4;
    const long long t0_ = __builtin_readcyclecounter();
    int ts_ = 0;
#define CM_STAMP() if (tdbg && blockIdx.x == (q.dbg >> 8) && threadIdx.x == 0 && ts_ < 20) tdbg[ts_++] = (float)(__builtin_readcyclecounter() - t0_)
    const int K = q.K, HW = q.H * q.Wd, CHW = C * HW, HpWp = q.Hp * q.Wp, KHW = K * HpWp;

    const int gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
    float xr[CM_XR], gr[CM_XR], yr[CM_XR];
    int mr[CM_XR];
#define CM_PREFETCH(N_)                                                                     \
    {                                                                                       \
        const float* xp_ = x + (size_t)(N_) * CHW;                                          \
        const float* gp_ = g + (size_t)(N_) * KHW;                                          \
        const float* yp_ = y + (size_t)(N_) * KHW;                                          \
        const uint8_t* mp_ = mask + (size_t)(N_) * KHW;                                     \
        _Pragma("unroll") for (int j = 0; j < CM_XR; ++j) {                                 \
            xr[j] = xp_[min(lane + 64 * j, CHW - 1)];                                       \
            gr[j] = gp_[min(lane + 64 * j, KHW - 1)];                                       \
            if (ACT != TN_ACT_LEAKY) yr[j] = yp_[min(lane + 64 * j, KHW - 1)];              \
            mr[j] = mp_[min(lane + 64 * j, KHW - 1)];                                       \
        }                                                                                   \
    }
    int n = __builtin_amdgcn_readfirstlane(gw);
    if (n < q.N) CM_PREFETCH(n);

    int* wtab = reinterpret_cast<int*>(sm);           // [4*PT] window -> x tile offset of its top-left
    float* wbase = sm + q.tabf + wv * q.wavef;
    float* sx = wbase;                                // x tile (zero padded)
    float* sdx = sx + q.xf;                           // dx tile
    float* sdz = sdx + q.xf;                          // [4*KS2][npixp], pixels in window order
    const int ZERO = q.xf - 4, ONE = q.xf - 3;        // constant cells behind the x tile
    const int DUMMY = q.xf - 2;                       // write-only cell for out-of-range lanes

    for (int i = threadIdx.x; i < 4 * q.PT; i += 256) {
        const int w = min(i, HpWp - 1);
        const int wy = w / q.Wp, wx = w - wy * q.Wp;
        wtab[i] = 2 * wy * q.Wx + 2 * wx;
    }
    // x tile: the pads stay zero for the whole kernel.  dz tile: the staging rewrites the cells of
    // the K x Hp*Wp windows for each image; everything else (tile padding, rows >= K) stays zero.
    for (int i = lane * 4; i < q.xf; i += 256) *reinterpret_cast<float4*>(sx + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane * 4; i < q.dzf + 4; i += 256) *reinterpret_cast<float4*>(sdz + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (lane == 0) sx[ONE] = 1.f;

    // ---- loop-invariant operands ------------------------------------------------------------
    auto ckk_off = [&](int ckk) {
        const int c = ckk / FF, ab = ckk - c * FF, a = ab / F, bb = ab - a * F;
        return c * q.xplane + a * q.Wx + bb;
    };
    int off3[NT], msk3[NT];    // wgrad B operand: ckk = 16nt+lo (bias column reads the 1.0 cell)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ckk = 16 * nt + lo;
        off3[nt] = ckk < CKK ? ckk_off(ckk) : (ckk == CKK ? ONE : ZERO);
        msk3[nt] = ckk < CKK ? -1 : 0;
    }
    int arow[NKT], amsk[NKT];  // wgrad A operand: dz row of filter 16kt+lo (rows >= K: the zero cells)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int k = 16 * kt + lo;
        arow[kt] = k < K ? k * q.npixp : q.dzf;
        amsk[kt] = k < K ? -1 : 0;
    }
    // W44 operand addresses (see the comment above the kernel)
    constexpr bool REM = CKK + 1 > 32;
    constexpr int NM = (KS2 + 1) / 2;
    const int g2 = lane >> 5, a8 = (lane >> 2) & 7, r4 = lane & 3, g8 = lane >> 3, a2 = (lane >> 2) & 1;
    const int fM = 4 * a8 + r4, tM = lane & 31, tR = 32 + 4 * a2 + r4;
    const int arowM = fM < K ? fM * q.npixp : q.dzf, amskM = fM < K ? -1 : 0;
    const int offM = tM < CKK ? ckk_off(min(tM, CKK - 1)) : (tM == CKK ? ONE : ZERO), mskM = tM < CKK ? -1 : 0;
    const int offR = tR < CKK ? ckk_off(min(tR, CKK - 1)) : (tR == CKK ? ONE : ZERO), mskR = tR < CKK ? -1 : 0;
    int arowR[NM], amskR[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int f = 8 * m + 4 * a2 + r4;
        arowR[m] = f < K ? f * q.npixp : q.dzf;
        amskR[m] = f < K ? -1 : 0;
    }
    f32x4 accM[KS2], accR[KS2];
#pragma unroll
    for (int a_ = 0; a_ < KS2; ++a_) accM[a_] = accR[a_] = f32x4{0.f, 0.f, 0.f, 0.f};
    float Aw2[NTD][KS2];       // dgrad product A operand: Wf[k = 4s+qd][c = lo>>2][ab = 4mt + (lo&3)]
#pragma unroll
    for (int mt = 0; mt < NTD; ++mt)
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            const int k = 4 * s + qd, c = lo >> 2, ab = 4 * mt + (lo & 3);
            const float v = W[((size_t)min(k, K - 1) * C + min(c, C - 1)) * FF + (FF - 1 - min(ab, FF - 1))];
            Aw2[mt][s] = (k < K && c < C && ab < FF) ? v : 0.f;
        }
    int tabr[CM_XR];           // image element lane+64j -> offset in the x / dx tile
    int dzo[CM_XR];            // pooled element lane+64j = (k, window) -> its 4 dz cells
    const float rHW = __builtin_amdgcn_rcpf((float)HW), rWd = __builtin_amdgcn_rcpf((float)q.Wd);
    const float rHpWp = __builtin_amdgcn_rcpf((float)HpWp);
#pragma unroll
    for (int j = 0; j < CM_XR; ++j) {
        const int i = min(lane + 64 * j, CHW - 1);
        const int c = cm_div(i, rHW), r = i - c * HW, yy = cm_div(r, rWd), xx = r - yy * q.Wd;
        tabr[j] = (lane + 64 * j < CHW) ? c * q.xplane + (yy + q.pad) * q.Wx + xx + q.pad : DUMMY;
        const int e = min(lane + 64 * j, KHW - 1);
        const int k = cm_div(e, rHpWp), w = e - k * HpWp;
        dzo[j] = k * q.npixp + 4 * w;       // lanes beyond K*Hp*Wp rewrite the last element's cells
    }
    // Pin the operands: complete here, so the loops below never wait on the vector-memory counter
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
#pragma unroll
    for (int mt = 0; mt < NTD; ++mt)
#pragma unroll
        for (int s = 0; s < KS2; ++s) asm volatile("" : "+v"(Aw2[mt][s]));

    f32x4 accW[NKT][NT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accW[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    CM_STAMP();      // setup done

    const float tie = prm > 0.f ? 1.f + prm : 0.f;          // leaky slope at an exact 0
    // dgrad product column lo = tile column x': the dz pixel (y, x') in window order, or a zero cell
    const int dcol = lo < q.Wo ? 4 * (lo >> 1) + (lo & 1) : 16 * q.PT;
    const int dmsk = lo < q.Wo ? -1 : 0;
    for (; n < q.N; n += nw) {
        // ---- stage: x tile, and dz = mask bit ? g * act'(y) : 0 (branch-free, see tabr / dzo) ----
#pragma unroll
        for (int j = 0; j < CM_XR; ++j) {
            sx[tabr[j]] = xr[j];
            const int m = mr[j];
            float gp;
            if (ACT == TN_ACT_LEAKY) {      // the slope is in the mask's sign bits: y is not read
                gp = (m & 32) ? prm : tie;
                gp = (m & 16) ? 1.f : gp;
            } else {
                gp = tn_act_grad_from_out(yr[j], act, prm);
            }
            const float gy = gr[j] * gp;
            *reinterpret_cast<float4*>(sdz + dzo[j]) =
                make_float4((m & 1) ? gy : 0.f, (m & 2) ? gy : 0.f, (m & 4) ? gy : 0.f, (m & 8) ? gy : 0.f);
        }
        if (n + nw < q.N) CM_PREFETCH(n + nw);
        cm_wave_sync();
        CM_STAMP();  // image staged
        if constexpr (W44) {
            // ---- wgrad on the 16-block MFMA: step Q = pooling windows 2Q and 2Q + 1 (8 pixels) ----
            float4 a4c;
            float b4c[4], arc[NM], brc = 0.f;
#define CM_W4_LOAD(Q_, A_, B_, AR_, BR_)                                                    \
            {                                                                               \
                const int w_ = 2 * (Q_) + g2, wo_ = wtab[w_];                               \
                A_ = *reinterpret_cast<const float4*>(sdz + arowM + ((4 * w_) & amskM));    \
                _Pragma("unroll") for (int r = 0; r < 4; ++r)                               \
                    B_[r] = sx[offM + ((wo_ + (r >> 1) * q.Wx + (r & 1)) & mskM)];          \
                if (REM) {                                                                  \
                    const int wr_ = 2 * (Q_) + (g8 >> 2), er_ = g8 & 3;                     \
                    _Pragma("unroll") for (int m = 0; m < NM; ++m)                          \
                        AR_[m] = sdz[arowR[m] + ((4 * wr_ + er_) & amskR[m])];              \
                    BR_ = sx[offR + ((wtab[wr_] + (er_ >> 1) * q.Wx + (er_ & 1)) & mskR)];  \
                }                                                                           \
            }
            CM_W4_LOAD(0, a4c, b4c, arc, brc);
#pragma unroll 1
            for (int Q = 0; Q < 2 * q.PT; ++Q) {
                float4 a4n;
                float b4n[4], arn[NM], brn = 0.f;
                const int Qn = min(Q + 1, 2 * q.PT - 1);
                CM_W4_LOAD(Qn, a4n, b4n, arn, brn);     // next step's operands travel during this step's products
                if (!(q.dbg & 1)) {
#define CM_M4(AB) if (AB < KS2) accM[AB < KS2 ? AB : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_, b4c[s_], accM[AB < KS2 ? AB : 0], 3, AB, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) {
                        const float a_ = s_ == 0 ? a4c.x : s_ == 1 ? a4c.y : s_ == 2 ? a4c.z : a4c.w;
                        CM_M4(0) CM_M4(1) CM_M4(2) CM_M4(3) CM_M4(4) CM_M4(5) CM_M4(6) CM_M4(7)
                    }
#undef CM_M4
                    if (REM) {
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            accR[2 * m] = __builtin_amdgcn_mfma_f32_4x4x1f32(arc[m], brc, accR[2 * m], 1, 0, 0);
                            if (2 * m + 1 < KS2)
                                accR[2 * m + 1 < KS2 ? 2 * m + 1 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                                    arc[m], brc, accR[2 * m + 1 < KS2 ? 2 * m + 1 : 0], 1, 1, 0);
                        }
                    }
                }
                a4c = a4n;
#pragma unroll
                for (int r = 0; r < 4; ++r) b4c[r] = b4n[r];
#pragma unroll
                for (int m = 0; m < NM; ++m) arc[m] = arn[m];
                brc = brn;
            }
#undef CM_W4_LOAD
        } else {
        // ---- wgrad: dWf[k][ckk] += sum_pix dz[k][pix] * patch[pix][ckk] -------------------------
        float4 ac[NKT];
        float bc[4][NT];
#define CM_WG_LOAD(T_, A_, B_)                                                              \
        {                                                                                   \
            const int po_ = 16 * (T_) + 4 * qd;                                             \
            _Pragma("unroll") for (int kt = 0; kt < NKT; ++kt)                              \
                A_[kt] = *reinterpret_cast<const float4*>(sdz + arow[kt] + (po_ & amsk[kt])); \
            const int wo_ = wtab[4 * (T_) + qd];                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                 \
                const int pw_ = wo_ + (r >> 1) * q.Wx + (r & 1);                            \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                           \
                    B_[r][nt] = sx[off3[nt] + (pw_ & msk3[nt])];                            \
            }                                                                               \
        }
        CM_WG_LOAD(0, ac, bc);
#pragma unroll 1
        for (int t = 0; t < q.PT; ++t) {
            float4 an[NKT];
            float bn[4][NT];
            const int tn = min(t + 1, q.PT - 1);
            CM_WG_LOAD(tn, an, bn);         // next tile's operands travel during this tile's products
            if (!(q.dbg & 1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float a = r == 0 ? ac[kt].x : r == 1 ? ac[kt].y : r == 2 ? ac[kt].z : ac[kt].w;
                        accW[kt][nt] = cm_mfma(a, bc[r][nt], accW[kt][nt]);
                    }
            }
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) ac[kt] = an[kt];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bc[r][nt] = bn[r][nt];
        }
#undef CM_WG_LOAD
        }
        CM_STAMP();  // wgrad done
        if (dx && !(q.dbg & 2)) {
            // ---- dgrad: one dz row per step, see convblock_bwd_mfma --------------------------------
            float R1 = 0.f, R2 = 0.f;
            float* dxo = sdx + qd * q.xplane + lo;
            const bool dlive = (qd < C) && (lo < q.Wo + 2);
            float bcur[KS2];
            {
                const float* dzp = sdz + qd * q.npixp + dcol;
#pragma unroll
                for (int s = 0; s < KS2; ++s) bcur[s] = dzp[4 * s * q.npixp];
            }
#pragma unroll 1
            for (int yy = 0; yy < q.Ho; ++yy) {
                const int yn = min(yy + 1, q.Ho - 1);
                const int prow = 4 * (yn >> 1) * q.Wp + 2 * (yn & 1);
                const float* dzp = sdz + qd * q.npixp + dcol + (prow & dmsk);
                float bnxt[KS2];
#pragma unroll
                for (int s = 0; s < KS2; ++s) bnxt[s] = dzp[4 * s * q.npixp];
                f32x4 T[NTD];
#pragma unroll
                for (int mt = 0; mt < NTD; ++mt) T[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS2; ++s)
#pragma unroll
                    for (int mt = 0; mt < NTD; ++mt) T[mt] = cm_mfma(Aw2[mt][s], bcur[s], T[mt]);
#pragma unroll
                for (int s = 0; s < KS2; ++s) bcur[s] = bnxt[s];
                const float S0 = T[0][0] + cm_shr<1>(T[0][1]) + cm_shr<2>(T[0][2]);
                const float S1 = T[0][3] + cm_shr<1>(T[1][0]) + cm_shr<2>(T[1][1]);
                const float S2 = T[1][2] + cm_shr<1>(T[1][3]) + cm_shr<2>(T[2][0]);
                const float out = S0 + R1;
                R1 = S1 + R2;
                R2 = S2;
                if (dlive) dxo[yy * q.Wx] = out;
            }
            if (dlive) {
                dxo[q.Ho * q.Wx] = R1;
                dxo[(q.Ho + 1) * q.Wx] = R2;
            }
            cm_wave_sync();
            float* dxp = dx + (size_t)n * CHW;
#pragma unroll
            for (int j = 0; j < CM_XR; ++j) {
                const int i = lane + 64 * j;
                if (i < CHW) dxp[i] = sdx[tabr[j]];
            }
        }
        cm_wave_sync();
        CM_STAMP();  // dgrad done
    }
#undef CM_PREFETCH
    // ---- one partial slab per block: sum the 4 waves' accumulators in wave order ---------------
    __syncthreads();
    if constexpr (W44) {
#pragma unroll
        for (int a_ = 0; a_ < KS2; ++a_)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wbase[(a_ * 4 + r) * 64 + lane] = accM[a_][r];
                if (REM) wbase[((KS2 + a_) * 4 + r) * 64 + lane] = accR[a_][r];
            }
        __syncthreads();
        for (int e = threadIdx.x; e < K * (CKK + 1); e += 256) {
            const int k = e / (CKK + 1), ckk = e - k * (CKK + 1);
            float s = 0.f;
            if (!REM || ckk < 32) {       // rows of block k >> 2: lane = pixel parity * 32 + column
                const float* src = sm + q.tabf + ((k >> 2) * 4 + (k & 3)) * 64 + ckk;
#pragma unroll
                for (int w = 0; w < 4; ++w) s += src[w * q.wavef] + src[w * q.wavef + 32];
            } else {                      // remainder: lane = pixel (0..7) * 8 + column - 32
                const float* src = sm + q.tabf + ((KS2 + (k >> 2)) * 4 + (k & 3)) * 64 + (ckk - 32);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    float t = 0.f;
#pragma unroll
                    for (int p8 = 0; p8 < 8; ++p8) t += src[w * q.wavef + 8 * p8];
                    s += t;
                }
            }
            if (ckk < CKK)
                partial[(size_t)blockIdx.x * K * CKK + k * CKK + ckk] = s;
            else
                dbpartial[(size_t)blockIdx.x * K + k] = s;
        }
    } else {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) wbase[((kt * NT + nt) * 4 + r) * 64 + lane] = accW[kt][nt][r];
    __syncthreads();
    for (int e = threadIdx.x; e < K * (CKK + 1); e += 256) {
        const int k = e / (CKK + 1), ckk = e - k * (CKK + 1);
        const int idx = (((k >> 4) * NT + (ckk >> 4)) * 4 + (k & 3)) * 64 + ((k & 15) >> 2) * 16 + (ckk & 15);
        const float* src = sm + q.tabf + idx;
        const float s = ((src[0] + src[q.wavef]) + src[2 * q.wavef]) + src[3 * q.wavef];
        if (ckk < CKK)
            partial[(size_t)blockIdx.x * K * CKK + k * CKK + ckk] = s;
        else
            dbpartial[(size_t)blockIdx.x * K + k] = s;
    }
    }
    CM_STAMP();      // slab written
#undef CM_STAMP
}

static void cm_geometry(CmGeom& q, int C, int K) {
    q.Wx = q.Wo + 3;
    q.xplane = (q.Ho + 3) * q.Wx;
    q.xf = ((C * q.xplane + 3) & ~3) + 4;
    q.PT = (q.Hp * q.Wp + 3) / 4;
    q.npixp = 16 * q.PT + 4;                    // npixp/4 odd: filters land on distinct b128 banks
    q.dzf = 4 * ((K + 3) / 4) * q.npixp;
    q.tabf = (4 * q.PT + 3) & ~3;
    int red = 2 * ((C * 9 + 16) / 16) * 4 * 64;           // final accumulator exchange
    if (red < 2 * ((K + 3) / 4) * 4 * 64) red = 2 * ((K + 3) / 4) * 4 * 64;      // (W44: main + remainder accumulators)
    q.wavef = 2 * q.xf + q.dzf + 4;          // + one zero float4 behind the dz tile
    if (q.wavef < red) q.wavef = red;
}

static size_t cm_lds_bytes(const CmGeom& q) { return ((size_t)q.tabf + 4 * (size_t)q.wavef) * sizeof(float); }

// 1 if the matrix-core backward applies to this block shape
static int cm_supported(int C, int K, int f, int stride, int p, int H, int Wd, int pad_lo, int Ho,
                        int Wo, int Hp, int Wp, bool recompute) {
    if (f != 3 || stride != 1 || p != 2 || C < 1 || C > 4 || K < 1 || K > 32) return 0;
    if (C * H * Wd > 64 * CM_XR || K * Hp * Wp > 64 * CM_XR) return 0;
    if (H + 2 * pad_lo > Ho + 2 || Wd + 2 * pad_lo > Wo + 2) return 0;     // the padded image is the tile
    if (Wo + 2 > 16) return 0;                   // a dz row (+ its 2-pixel spill) is one 16-lane DPP row
    if (Hp != (Ho + 1) / 2 || Wp != (Wo + 1) / 2) return 0;
    CmGeom q;
    q.H = H; q.Wd = Wd; q.Ho = Ho; q.Wo = Wo; q.Hp = Hp; q.Wp = Wp; q.dbg = 0;
    cm_geometry(q, C, K);
    if (recompute && K * Hp * Wp > q.xf) return 0;   // the g tile borrows the dx tile
    return cm_lds_bytes(q) <= 78 * 1024;         // two blocks per CU
}

int tn_convblock_mfma_supported(int C, int K, int f, int stride, int p, int H, int Wd, int pad_lo,
                                int Ho, int Wo, int Hp, int Wp) {
    return cm_supported(C, K, f, stride, p, H, Wd, pad_lo, Ho, Wo, Hp, Wp, true);
}

template <int C, int KS2>
static int launch_cm(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                     float* dx, float* dW, float* db, CmGeom q, int act, float prm) {
    const size_t lds = cm_lds_bytes(q);
    int nblk = 2 * ctx->num_cus;
    if (nblk > cdiv(q.N, 4)) nblk = cdiv(q.N, 4);
    const size_t KCFF = (size_t)q.K * C * 9;
    float* partial;
    int rc = tn_scratch_get(ctx, (size_t)nblk * (KCFF + q.K) * sizeof(float), &partial);
    if (rc) return rc;
    float* dbpartial = partial + (size_t)nblk * KCFF;
    if (act == TN_ACT_LEAKY) {
        auto kern = convblock_bwd_mfma<C, KS2, TN_ACT_LEAKY>;
        static size_t set_for = 0;      // the attribute call is not a stream op: do it once per size
        if (set_for < lds) {
            TN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set_for = lds;
        }
        kern<<<nblk, 256, lds, ctx->stream>>>(x, W, b, g, dx, partial, dbpartial, q, act, prm,
                                              (q.dbg & 16) ? db : nullptr);
    } else {
        auto kern = convblock_bwd_mfma<C, KS2, -1>;
        static size_t set_for = 0;
        if (set_for < lds) {
            TN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set_for = lds;
        }
        kern<<<nblk, 256, lds, ctx->stream>>>(x, W, b, g, dx, partial, dbpartial, q, act, prm,
                                              (q.dbg & 16) ? db : nullptr);
    }
    TN_LAUNCH_CHECK();
    if (q.dbg & 16) return TN_OK;     // timing run: db holds the cycle stamps of block 0
    return tn_conv_wgrad_finish(ctx, partial, dbpartial, dW, db, nblk, q.K, C, 3);
}

template <int C, int KS2>
static int launch_cm_mask(tn_ctx* ctx, const float* x, const float* W, const float* g, const float* y,
                          const uint8_t* mask, float* dx, float* dW, float* db, CmGeom q, int act,
                          float prm) {
    const size_t lds = cm_lds_bytes(q);
    int nblk = 2 * ctx->num_cus;
    if (nblk > cdiv(q.N, 4)) nblk = cdiv(q.N, 4);
    const size_t KCFF = (size_t)q.K * C * 9;
    float* partial;
    int rc = tn_scratch_get(ctx, (size_t)nblk * (KCFF + q.K) * sizeof(float), &partial);
    if (rc) return rc;
    float* dbpartial = partial + (size_t)nblk * KCFF;
    float* tdbg = (q.dbg & 16) ? db : nullptr;
    static int w44 = -1;
    if (w44 < 0) {
        const char* e = getenv("TN_CB_W44");
        w44 = e ? atoi(e) : 1;
    }
    if (act == TN_ACT_LEAKY && w44) {
        auto kern = convblock_bwd_mask_mfma<C, KS2, TN_ACT_LEAKY, true>;
        static size_t set_for = 0;
        if (set_for < lds) {
            TN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set_for = lds;
        }
        kern<<<nblk, 256, lds, ctx->stream>>>(x, W, g, y, mask, dx, partial, dbpartial, q, act, prm, tdbg);
    } else if (act == TN_ACT_LEAKY) {
        auto kern = convblock_bwd_mask_mfma<C, KS2, TN_ACT_LEAKY, false>;
        static size_t set_for = 0;
        if (set_for < lds) {
            TN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set_for = lds;
        }
        kern<<<nblk, 256, lds, ctx->stream>>>(x, W, g, y, mask, dx, partial, dbpartial, q, act, prm, tdbg);
    } else {
        auto kern = convblock_bwd_mask_mfma<C, KS2, -1, false>;
        static size_t set_for = 0;
        if (set_for < lds) {
            TN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set_for = lds;
        }
        kern<<<nblk, 256, lds, ctx->stream>>>(x, W, g, y, mask, dx, partial, dbpartial, q, act, prm, tdbg);
    }
    TN_LAUNCH_CHECK();
    if (q.dbg & 16) return TN_OK;
    return tn_conv_wgrad_finish(ctx, partial, dbpartial, dW, db, nblk, q.K, C, 3);
}

static int cm_dbg() {
    static int dbg = -1;
    if (dbg < 0) {
        const char* e = getenv("TN_CM_DBG");
        dbg = e ? atoi(e) : 0;
    }
    return dbg;
}

extern "C" int tn_convblock_mask_supported(int C, int K, int f, int stride, int p, int H, int Wd,
                                           int pad_lo, int Ho, int Wo, int Hp, int Wp) {
    return cm_supported(C, K, f, stride, p, H, Wd, pad_lo, Ho, Wo, Hp, Wp, false);
}

extern "C" int tn_convblock_bwd_mask(tn_ctx* ctx, const float* x, const float* W, const float* g,
                                     const float* y, const uint8_t* mask, float* dx, float* dW,
                                     float* db, int N, int C, int H, int Wd, int K, int f, int pad_lo,
                                     int Ho, int Wo, int p, int Hp, int Wp, int act, float act_param) {
    TN_REQUIRE(tn_convblock_mask_supported(C, K, f, 1, p, H, Wd, pad_lo, Ho, Wo, Hp, Wp),
               "tn_convblock_bwd_mask: unsupported C=%d K=%d f=%d p=%d %dx%d", C, K, f, p, H, Wd);
    TN_REQUIRE(x && W && g && y && mask && dW && db, "tn_convblock_bwd_mask: null argument");
    CmGeom q;
    q.N = N; q.H = H; q.Wd = Wd; q.K = K; q.pad = pad_lo; q.Ho = Ho; q.Wo = Wo; q.Hp = Hp; q.Wp = Wp;
    cm_geometry(q, C, K);
    q.dbg = cm_dbg();
#define CM_KS(C_, S_) case S_: return launch_cm_mask<C_, S_>(ctx, x, W, g, y, mask, dx, dW, db, q, act, act_param)
#define CM_GO(C_)                                                                                  \
    switch ((K + 3) / 4) {                                                                         \
        CM_KS(C_, 1); CM_KS(C_, 2); CM_KS(C_, 3); CM_KS(C_, 4);                                    \
        CM_KS(C_, 5); CM_KS(C_, 6); CM_KS(C_, 7); default: CM_KS(C_, 8);                           \
    }
    switch (C) {
        case 1: CM_GO(1);
        case 2: CM_GO(2);
        case 3: CM_GO(3);
        default: CM_GO(4);
    }
#undef CM_GO
#undef CM_KS
}

int tn_convblock_mfma_bwd(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                          float* dx, float* dW, float* db, int N, int C, int H, int Wd, int K,
                          int pad_lo, int Ho, int Wo, int Hp, int Wp, int act, float act_param) {
    CmGeom q;
    q.N = N; q.H = H; q.Wd = Wd; q.K = K; q.pad = pad_lo; q.Ho = Ho; q.Wo = Wo; q.Hp = Hp; q.Wp = Wp;
    cm_geometry(q, C, K);
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char* e = getenv("TN_CM_DBG");
            dbg = e ? atoi(e) : 0;
        }
        q.dbg = dbg;
    }
#define CM_KS(C_, S_) case S_: return launch_cm<C_, S_>(ctx, x, W, b, g, dx, dW, db, q, act, act_param)
#define CM_GO(C_)                                                                                  \
    switch ((K + 3) / 4) {                                                                         \
        CM_KS(C_, 1); CM_KS(C_, 2); CM_KS(C_, 3); CM_KS(C_, 4);                                    \
        CM_KS(C_, 5); CM_KS(C_, 6); CM_KS(C_, 7); default: CM_KS(C_, 8);                           \
    }
    switch (C) {
        case 1: CM_GO(1);
        case 2: CM_GO(2);
        case 3: CM_GO(3);
        default: CM_GO(4);
    }
#undef CM_GO
#undef CM_KS
}
