// DTYPE 'float16' (BASELINE.json configs[4]: "fp16 inputs / fp32 accum MFMA"): 3x3 'same' stride-1 convolutions on
// tensors that LIVE in HBM as fp16, in the layout the matrix core wants -- "c8": a logical (N, C, H, W) tensor is
// stored [N][ceil(C/8)][H][W][8] halfs, i.e. one 16-byte cell = the 8 channels of an octet at one pixel (channels
// beyond C are zero).  Same products as theanet/layer/convpool.py:54-72 and their Theano gradients (CorrMM_gradInputs
// / CorrMM_gradWeights); the reference is float32-only (weights.py:8), so the arithmetic of this mode is specified by
// the oracle's stored-fp16 restatement (oracle/theanet_oracle.py, f16 = 'stored'): activations and gradients are
// rounded to IEEE half (nearest-even) when a layer stores them, weights when they are staged (fp32 master weights),
// every product is exact and accumulated in fp32, bias / activation / pooling act on the fp32 sums.
//
// Why c8.  The B operand of v_mfma_f32_32x32x16_f16 is "8 consecutive reduction indices per lane"; with lane = pixel
// and reduction = input channel that is exactly one 16-byte cell, so
//   * forward / input gradient: the halo tile is COPIED HBM -> LDS (16-byte loads, 16-byte stores, no conversion, no
//     transposition: conv_tile16.hip spent 32 v_cvt + 8 loads per 4 pixels on that), the im2col is a constant added to
//     an LDS address, and with the filters of a 32-row MFMA tile permuted (bits 2 and 3 of the row swapped, done once by
//     the weight-arranging kernel) a lane's accumulators are two complete octets of its pixel: the epilogue stores
//     16-byte cells straight from registers -- no LDS round trip;
//   * a wave's two 32-pixel groups are the two rows of a 2 x 32 patch, so the 2x2 max-pool of a fused block is one
//     in-lane max and one lane-pair exchange;
//   * gradients travel as fp16(gs * g) (gs = GRAD_SCALE, a power of two; |dz| ~ 1e-3/B is fp16-subnormal territory):
//     scaled ONCE where the first fp16 gradient is produced, unscaled in the fp32 epilogues of the weight gradients.
// The weight gradient (reduction = pixels) wants the other orientation: its staging pass transposes 4-pixel x 8-channel
// blocks in registers (16 v_perm_b32 per 64 bytes) into the [channel][pixel] LDS image of conv_tile16.hip's kernel.
#include "conv_tile_common.h"

#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));

struct C8G {
    const _Float16* x;        // gathered tensor, c8 (N, C8, H, W, 8); MODE 3: the POOLED gradient (N, C8, H/2, W/2, 8)
    const _Float16* wt;       // arranged weights [KT][nchunk][tap][2][32*FT][8]
    _Float16* out;            // c8 (N, K8, H, W, 8); MODE 1: pooled (N, K8, H/2, W/2, 8)
    const float* bias;        // forward
    const _Float16* prev_a;   // input gradient: output of the layer below (same shape as out) or NULL
    uint8_t* mask_out;        // MODE 1: pooling mask (N, K8, H/2, W/2, 8) bytes, may be NULL
    const uint8_t* mask_in;   // MODE 3: mask and pooled output of the block whose dz is being gathered
    const _Float16* y_in;
    int N, C8, K8, H, W, act;
    float prm;
    int in_act;
    float in_prm;
    int KT, MT, RT, NI, TH, THi, RS, plane, nchunk, TP, nslots;
};

__host__ __device__ __forceinline__ int c8_swap23(int j) { return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1); }

// wt[kt][chunk][tap][o][j][e] (halfs): MFMA row j of filter tile kt holds filter kt*KBF + (j & ~31) + swap23(j & 31)
// (so that a lane's accumulators 0-7 / 8-15 are whole octets), channel chunk*16 + 8*o + e, correlation tap
__global__ __launch_bounds__(256) void c8_wt_kernel(const float* __restrict__ W, _Float16* __restrict__ wt, int K, int C,
                                                   int KBF, int nchunk, int total, int dgrad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int r = idx;
    const int e = r & 7; r >>= 3;
    const int j = r % KBF; r /= KBF;
    const int o = r & 1; r >>= 1;
    const int tap = r % 9; r /= 9;
    const int chunk = r % nchunk;
    const int kt = r / nchunk;
    const int filt = kt * KBF + (j & ~31) + c8_swap23(j & 31), ch = chunk * 16 + 8 * o + e;
    float v = 0.f;
    if (filt < K && ch < C)
        v = dgrad ? W[((size_t)ch * K + filt) * 9 + tap]          // W[k = ch][c = filt][u][v]
                  : W[((size_t)filt * C + ch) * 9 + (8 - tap)];   // true convolution: flipped taps
    wt[idx] = (_Float16)v;
}

// staging slot e of the halo tile: (octet of the chunk, image, tile row, column) -> 16-byte cell.
// POOLED: the cell is gathered from the pooled tensors: g = the pooled cell, cstride = cells per octet plane there
struct C8Slot { int g, l, o, sh; bool ok; };
template <bool POOLED>
__device__ __forceinline__ C8Slot c8_slot(const C8G& g, int e, int n0, int r0) {
    C8Slot s;
    const bool in = e < g.nslots;
    int rr = min(e, g.nslots - 1);
    const int col = rr % g.W; rr /= g.W;
    const int r = rr % g.THi; rr /= g.THi;
    const int ni = rr % g.NI;
    const int o = rr / g.NI;
    const int row = r0 - 1 + r, n = n0 + ni;
    s.ok = in && (unsigned)row < (unsigned)g.H && n < g.N;
    const int nn = min(n, g.N - 1), rw = min(max(row, 0), g.H - 1);
    s.o = o;
    s.sh = ((rw & 1) << 1) | (col & 1);
    if (POOLED) s.g = ((nn * g.C8 + o) * (g.H >> 1) + (rw >> 1)) * (g.W >> 1) + (col >> 1);
    else s.g = ((nn * g.C8 + o) * g.H + rw) * g.W + col;      // in cells; + 2*chunk*(cells per plane) per chunk
    s.l = (o * g.plane + (ni * g.THi + r) * g.RS + 1 + col) * 16;
    return s;
}

// dz cell of a pooled block: 8 channels at full-resolution pixel (row, col) from the pooled gradient cell, the mask
// bytes and (activations other than leaky-ReLU) the pooled output:  bit (2*(row&1) + (col&1)) of the mask says whether
// this window element attained the maximum; bits 4 / 5 = sign of the pooled value
__device__ __forceinline__ uint4 c8_pool_cell(const uint4 g8, const uint2 m8, const uint4 y8, int sh, int act, float prm) {
    const half8 gh = __builtin_bit_cast(half8, g8), yh = __builtin_bit_cast(half8, y8);
    half8 o;
    const float tie = prm > 0.f ? 1.f + prm : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned m = ((e < 4 ? m8.x : m8.y) >> (8 * (e & 3))) & 0xffu;
        float d;
        if (act == TN_ACT_LEAKY) d = (m & 16u) ? 1.f : ((m & 32u) ? prm : tie);
        else d = tn_act_grad_from_out((float)yh[e], act, prm);
        const float v = ((m >> sh) & 1u) ? (float)gh[e] * d : 0.f;
        o[e] = (_Float16)v;
    }
    return __builtin_bit_cast(uint4, o);
}

// MODE 0: forward (bias + act); 1: forward + 2x2 max-pool + mask; 2: input gradient (x act' of the layer below);
// 3: input gradient of a pooled block, dz gathered from (g, mask, y)
template <int FT, int MODE, int NS>
__global__ __launch_bounds__(256, 2) void c8_conv_kernel(C8G g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr bool DGRAD = MODE >= 2;
    constexpr int KBF = 32 * FT;
    constexpr int WB = 9 * 2 * KBF * 16;              // bytes of one weight chunk
    constexpr int WS = (WB / 16 + 255) / 256;         // 16-byte staging slots per thread (3 or 5)
    const int XB = 2 * g.plane * 16;                  // bytes of one input chunk (two octet planes)
    char* const Xs = reinterpret_cast<char*>(ct_smem);            // [2][XB]
    char* const Ws = Xs + 2 * XB;                                 // [2][WB]
    // XCD-aware decode: the filter tiles of one pixel tile share an L2
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int mt = (idx / g.KT) * 8 + xcd, kt = idx % g.KT;
    if (mt >= g.MT) return;
    const int grp = mt / g.RT, rt = mt - grp * g.RT;
    const int n0 = grp * g.NI, r0 = rt * g.TH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int HW = g.H * g.W;

    for (int i = t * 16; i < 2 * XB; i += 4096) *reinterpret_cast<float4*>(Xs + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    C8Slot sl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sl[s] = c8_slot<MODE == 3>(g, t + 256 * s, n0, r0);
    const char* wsrc = reinterpret_cast<const char*>(g.wt) + (size_t)kt * g.nchunk * WB + 16 * t;
    const uint4* xg = reinterpret_cast<const uint4*>(g.x);

    // this lane's two pixels: the two rows of a (2 x W') patch, same column (so that pooling is in-lane)
    int boff[2], prow[2], pni[2];
    bool pok[2];
    const int L = wave * 32 + l31, pair = L / g.W, pcol = L - pair * g.W;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int R = 2 * pair + pt;
        const bool in = R < g.NI * g.TH;
        const int Rc = in ? R : 0;
        const int ni = Rc / g.TH, r = Rc - ni * g.TH;
        pni[pt] = ni; prow[pt] = r0 + r;
        pok[pt] = in && n0 + ni < g.N && r0 + r < g.H;
        boff[pt] = (hi * g.plane + (ni * g.THi + r) * g.RS + pcol) * 16;
    }
    const int aoff = (hi * KBF + l31) * 16;

    f32x16 acc[FT][2];
#pragma unroll
    for (int a = 0; a < FT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // input cells two chunks ahead (two register sets alternating by chunk parity), weights (L2) one chunk ahead
    uint4 xr[2][NS];
    uint2 mr[2][NS];
    uint4 yr[2][NS];
    uint4 wr0, wr1, wr2, wr3, wr4;
#define C8_WL(J, R) if (WS > J) R = *reinterpret_cast<const uint4*>(w_ + min(4096 * J, WB - 16 - 16 * t))
#define C8_WST(J, R) if (WS > J && (4096 * (J + 1) <= WB || 16 * t + 4096 * J < WB)) *reinterpret_cast<uint4*>(wb + 4096 * J) = R
    auto gloadx = [&](int chunk, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const int ch = min(chunk, g.nchunk - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // octets beyond C8 meet zero weights: any finite value will do (clamped re-read)
            const int oc = min(2 * ch + sl[s].o, g.C8 - 1) - sl[s].o;
            if (MODE == 3) {
                const int pc = sl[s].g + oc * (HW >> 2);
                xr[P][s] = xg[pc];
                mr[P][s] = reinterpret_cast<const uint2*>(g.mask_in)[pc];
                if (g.in_act != TN_ACT_LEAKY) yr[P][s] = reinterpret_cast<const uint4*>(g.y_in)[pc];
            } else {
                xr[P][s] = xg[sl[s].g + oc * HW];
            }
        }
    };
    auto gloadw = [&](int chunk) __attribute__((always_inline)) {
        const int ch = min(chunk, g.nchunk - 1);
        const char* w_ = wsrc + (size_t)ch * WB;
        C8_WL(0, wr0); C8_WL(1, wr1); C8_WL(2, wr2); C8_WL(3, wr3); C8_WL(4, wr4);
    };
    auto lstore = [&](int buf, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        char* xb = Xs + buf * XB;
        char* wb = Ws + buf * WB + 16 * t;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (sl[s].ok) {
                uint4 v = xr[P][s];
                if (MODE == 3) v = c8_pool_cell(v, mr[P][s], yr[P][s], sl[s].sh, g.in_act, g.in_prm);
                *reinterpret_cast<uint4*>(xb + sl[s].l) = v;
            }
        }
        C8_WST(0, wr0); C8_WST(1, wr1); C8_WST(2, wr2); C8_WST(3, wr3); C8_WST(4, wr4);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    gloadx(0, P0{});
    gloadw(0);
    gloadx(1, P1{});
    __syncthreads();                 // the clearing is done
    lstore(0, P0{});
    __syncthreads();
    const int RS16 = g.RS * 16;
    auto body = [&](int chunk, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        gloadw(chunk + 1);
        gloadx(chunk + 2, Pc);
        const char* x0 = Xs + P * XB + boff[0];
        const char* x1 = Xs + P * XB + boff[1];
        const char* Wb = Ws + P * WB + aoff;
        // nine taps: the LDS operands of tap s+1 (FT A vectors, 2 B vectors of 8 halfs) are requested
        // before the 2*FT MFMAs of tap s are issued
        half8 a[2][FT], b[2][2];
#pragma unroll
        for (int f = 0; f < FT; ++f) a[0][f] = *reinterpret_cast<const half8*>(Wb + f * 512);
        b[0][0] = *reinterpret_cast<const half8*>(x0);
        b[0][1] = *reinterpret_cast<const half8*>(x1);
        __builtin_amdgcn_sched_group_barrier(0x100, FT + 2, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1, nx = cur ^ 1;
            if (tap + 1 < 9) {
                const int u = (tap + 1) / 3, v = (tap + 1) % 3;
#pragma unroll
                for (int f = 0; f < FT; ++f)
                    a[nx][f] = *reinterpret_cast<const half8*>(Wb + (tap + 1) * (2 * KBF * 16) + f * 512);
                b[nx][0] = *reinterpret_cast<const half8*>(x0 + u * RS16 + v * 16);
                b[nx][1] = *reinterpret_cast<const half8*>(x1 + u * RS16 + v * 16);
            }
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][f], b[cur][0], acc[f][0], 0, 0, 0);
                acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][f], b[cur][1], acc[f][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, FT + 2, 0);       // DS reads of the next tap
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * FT, 0);       // then this tap's MFMAs
        }
        // chunk + 1 (fetched a chunk ago into the other register set) goes into the other LDS buffer
        if (chunk + 1 < g.nchunk) lstore(P ^ 1, std::integral_constant<int, P ^ 1>{});
        __syncthreads();
    };
    for (int chunk = 0; chunk < g.nchunk; chunk += 2) {
        body(chunk, P0{});
        if (chunk + 1 < g.nchunk) body(chunk + 1, P1{});
    }
#undef C8_WL
#undef C8_WST

    // ---- epilogue: accumulators 0-7 / 8-15 of a lane are octets (4f + hi) / (4f + 2 + hi) of its pixel ----
    const int Ho = g.H, Wo = g.W;
    if (MODE == 1) {
        // conv + act + 2x2 max-pool: vertical max in-lane (the lane's two pixels), horizontal with lane ^ 1
        const int Hp = Ho >> 1, Wp = Wo >> 1;
        const bool ok = pok[0];
        const int dj = l31 & 1;
        const size_t pbase = ((size_t)(n0 + pni[0]) * g.K8 * Hp + (prow[0] >> 1)) * Wp + (pcol >> 1);
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int oct = kt * (KBF / 8) + f * 4 + h * 2 + hi;
                const int ob = min(oct, g.K8 - 1) * 8;
                const float4 b0 = *reinterpret_cast<const float4*>(g.bias + ob);
                const float4 b1 = *reinterpret_cast<const float4*>(g.bias + ob + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                half8 o8;
                unsigned mb[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a0 = tn_act_fwd(acc[f][0][h * 8 + e] + bb[e], g.act, g.prm);
                    const float a1 = tn_act_fwd(acc[f][1][h * 8 + e] + bb[e], g.act, g.prm);
                    const float mv = fmaxf(a0, a1);
                    const float m = fmaxf(mv, __shfl_xor(mv, 1, 64));
                    unsigned bits = (a0 == m ? (1u << dj) : 0u) | (a1 == m ? (4u << dj) : 0u);
                    bits |= (unsigned)__shfl_xor((int)bits, 1, 64);
                    bits |= (m > 0.f ? 16u : 0u) | (m < 0.f ? 32u : 0u);
                    o8[e] = (_Float16)m;
                    mb[e] = bits;
                }
                if (ok && dj == 0 && oct < g.K8) {
                    const size_t o = pbase + (size_t)oct * Hp * Wp;
                    reinterpret_cast<half8*>(g.out)[o] = o8;
                    if (g.mask_out) {
                        uint2 m2;
                        m2.x = mb[0] | (mb[1] << 8) | (mb[2] << 16) | (mb[3] << 24);
                        m2.y = mb[4] | (mb[5] << 8) | (mb[6] << 16) | (mb[7] << 24);
                        reinterpret_cast<uint2*>(g.mask_out)[o] = m2;
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const size_t pbase = ((size_t)(n0 + pni[pt]) * g.K8 * Ho + prow[pt]) * Wo + pcol;
        half8 pa[FT][2];
        if (DGRAD && g.prev_a && pok[pt]) {
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int oct = min(kt * (KBF / 8) + f * 4 + h * 2 + hi, g.K8 - 1);
                    pa[f][h] = reinterpret_cast<const half8*>(g.prev_a)[pbase + (size_t)oct * Ho * Wo];
                }
        }
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int oct = kt * (KBF / 8) + f * 4 + h * 2 + hi;
                half8 o8;
                if (DGRAD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v = acc[f][pt][h * 8 + e];
                        if (g.prev_a) v *= tn_act_grad_from_out((float)pa[f][h][e], g.act, g.prm);
                        o8[e] = (_Float16)v;
                    }
                } else {
                    const int ob = min(oct, g.K8 - 1) * 8;
                    const float4 b0 = *reinterpret_cast<const float4*>(g.bias + ob);
                    const float4 b1 = *reinterpret_cast<const float4*>(g.bias + ob + 4);
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (_Float16)tn_act_fwd(acc[f][pt][h * 8 + e] + bb[e], g.act, g.prm);
                }
                if (pok[pt] && oct < g.K8) reinterpret_cast<half8*>(g.out)[pbase + (size_t)oct * Ho * Wo] = o8;
            }
    }
}

// geometry of the pixel tiling; 0 when the shape is outside the kernel's limits
static int c8_geometry(C8G& g, int FT, int K, int C) {
    const int W = g.W, H = g.H;
    if (W != 8 && W != 16 && W != 32 && W != 64 && W != 128) return 0;
    if (H & 1) return 0;
    int TH = 256 / W;
    if (TH >= H) {
        g.TH = H; g.RT = 1;
        g.NI = 256 / (H * W);
        if (g.NI < 1) g.NI = 1;
        if (g.NI > g.N) g.NI = g.N;
    } else {
        if (H % TH) return 0;
        g.RT = H / TH; g.TH = TH; g.NI = 1;
    }
    if (g.TH & 1) return 0;
    g.TP = g.NI * g.TH * W;
    g.THi = g.TH + 2;
    g.RS = W + 2;
    if (W == 16) g.RS = 24;                 // the two row pairs of a half-wave on distinct bank groups
    if (W == 8) g.RS = 12;
    g.plane = g.NI * g.THi * g.RS + 2;      // cells per octet plane (+ the window overhang)
    g.nslots = 2 * g.NI * g.THi * W;
    if (g.nslots > 4 * 256) return 0;
    g.nchunk = cdiv(C, 16);
    g.KT = cdiv(K, 32 * FT);
    g.MT = cdiv(g.N, g.NI) * g.RT;
    return 1;
}

static int c8_pick_ft(int K) { return K > 32 ? 2 : 1; }

static size_t c8_lds_bytes(const C8G& g, int FT) { return (size_t)2 * (2 * g.plane * 16 + 9 * 2 * 32 * FT * 16); }

template <int FT, int MODE>
static int c8_launch(tn_ctx* ctx, C8G& g) {
    const size_t lds = c8_lds_bytes(g, FT);
    const int ns = cdiv(g.nslots, 256);
    const int grid = 8 * cdiv(g.MT, 8) * g.KT;
#define C8_GO(NS)                                                                                             \
    {                                                                                                         \
        static bool attr_set = false;                                                                         \
        if (!attr_set) {                                                                                      \
            TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&c8_conv_kernel<FT, MODE, NS>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));              \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        c8_conv_kernel<FT, MODE, NS><<<grid, 256, lds, ctx->stream>>>(g);                                     \
    }
    if (ns <= 2) C8_GO(2)
    else if (ns == 3) C8_GO(3)
    else C8_GO(4)
#undef C8_GO
    TN_LAUNCH_CHECK();
    return TN_OK;
}

template <int MODE>
static int c8_run(tn_ctx* ctx, C8G& g, const float* W, int K, int C) {
    const int FT = c8_pick_ft(K);
    TN_REQUIRE(c8_geometry(g, FT, K, C) && c8_lds_bytes(g, FT) <= 156 * 1024, "c8 conv: unsupported shape %dx%d", g.H, g.W);
    TN_REQUIRE((long long)g.N * g.C8 * g.H * g.W < (1ll << 28) && (long long)g.N * g.K8 * g.H * g.W < (1ll << 28),
               "c8 conv: tensor too large for 32-bit cell offsets");
    const int KBF = 32 * FT, total = g.KT * g.nchunk * 9 * 2 * KBF * 8;
    float* wt;
    int rc = tn_scratch_get(ctx, (size_t)total * sizeof(_Float16), &wt);
    if (rc) return rc;
    c8_wt_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(W, reinterpret_cast<_Float16*>(wt), K, C, KBF, g.nchunk,
                                                           total, MODE >= 2 ? 1 : 0);
    TN_LAUNCH_CHECK();
    g.wt = reinterpret_cast<const _Float16*>(wt);
    return FT == 2 ? c8_launch<2, MODE>(ctx, g) : c8_launch<1, MODE>(ctx, g);
}


// =================================================================================================
// Weight gradient of a 3x3 'same' convolution on c8 tensors:
//   dW[k][c][2-u][2-v] = (1/gs) * sum_{n,i,j} dz16[n,k,i,j] * x16[n,c,i-1+u,j-1+v]        (dz16 = fp16(gs*dz))
// GEMM rows = filters, columns = input channels at a fixed tap, reduction = pixels, 16 per MFMA; block = 32*NFT
// filters x 32*NCT channels x a range of 128-pixel tiles; a wave = one (filter tile, channel tile) pair (and, when
// NFT*NCT < 4, one of PS interleaved step subsets) with all nine taps.  LDS image, MFMA loop and slab layout are
// conv_tile16.hip's (A = dz[filter][8 consecutive pixels], B = x[channel][the same pixels shifted by the tap]: the
// centre column an aligned ds_read_b128, the +-1 columns v_alignbit funnel shifts); what differs is the staging:
// a slot = (octet, 4 consecutive pixels of a row) = 64 contiguous bytes of the c8 tensor, transposed in registers
// (16 v_perm_b32) into 8 channel rows of 4 pixels.  The bias gradient is the sum of the staged dz16 (v_dot2 with
// ones: exact fp32 accumulation).
// =================================================================================================
#define C8W_DZROW 136

struct C8WG {
    const _Float16* x;     // c8 (N, C8, H, W, 8)
    const _Float16* dz;    // c8 (N, K8, H, W, 8); POOL: pooled gradient (N, K8, H/2, W/2, 8)
    const uint8_t* mask;   // POOL
    const _Float16* y;     // POOL, activations other than leaky-ReLU
    float* ws;             // [S * PS][K*C*9] partial weight gradients, dW layout
    float* dbws;           // [S][K] partial bias gradients
    int N, C, C8, H, Wd, K, K8, act;
    float prm;
    int KG, CG, S, tpb;    // filter groups, channel groups, slabs, tiles per slab
    int NI, TH, THi, RT, NTILES;
    int RS, plane, q4, P, lgW, lgP;
    float oscale;
};

__device__ __forceinline__ uint4 c8_and4(uint4 v, bool ok) {
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
// 4 cells (pixels p0..p3, 8 channels each) -> out[e] = the 4 pixels of channel e (8 bytes)
__device__ __forceinline__ void c8_transpose4(const uint4 (&c)[4], uint2 (&out)[8]) {
    const unsigned* w0 = reinterpret_cast<const unsigned*>(&c[0]);
    const unsigned* w1 = reinterpret_cast<const unsigned*>(&c[1]);
    const unsigned* w2 = reinterpret_cast<const unsigned*>(&c[2]);
    const unsigned* w3 = reinterpret_cast<const unsigned*>(&c[3]);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        out[2 * d].x = __builtin_amdgcn_perm(w1[d], w0[d], 0x05040100u);
        out[2 * d].y = __builtin_amdgcn_perm(w3[d], w2[d], 0x05040100u);
        out[2 * d + 1].x = __builtin_amdgcn_perm(w1[d], w0[d], 0x07060302u);
        out[2 * d + 1].y = __builtin_amdgcn_perm(w3[d], w2[d], 0x07060302u);
    }
}

template <int NFT, int NCT, bool POOL>
__global__ __launch_bounds__(256) void c8_wgrad_kernel(C8WG g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr int KBF = 32 * NFT, CBF = 32 * NCT, PS = 4 / (NFT * NCT), SPW = 8 / PS;
    constexpr int NX = NCT;                           // x staging slots per thread (4 cells each)
    constexpr int DZSZ = KBF * C8W_DZROW * 2;         // bytes
    char* const smem = reinterpret_cast<char*>(ct_smem);
    const int XSZ = CBF * g.plane * 2, BUFSZ = DZSZ + XSZ;
    const int bid = blockIdx.x, per = g.KG * g.CG;
    const int z = ((bid >> 3) / per) * 8 + (bid & 7), rem = (bid >> 3) % per;
    if (z >= g.S) return;
    const int kg = rem / g.CG, cg = rem - kg * g.CG;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int ft = wave % NFT, ct = (wave / NFT) % NCT, ps = wave / (NFT * NCT);
    const int tile_beg = z * g.tpb, tile_end = min(g.NTILES, tile_beg + g.tpb);
    const int Wm = g.Wd - 1, THm = g.TH - 1;
    const int HW = g.H * g.Wd;

    for (int i = t * 16; i < 2 * BUFSZ; i += 4096) *reinterpret_cast<float4*>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- staging geometry of this thread (the same for every tile) ----
    // dz slot: octet t >> 5 of the filter group (NFT == 1: threads 128.. idle), pixels 4*(t & 31) .. +3 of the tile
    const int d_o = t >> 5, dq = t & 31, dp = 4 * dq;
    const bool d_on = d_o < KBF / 8;
    const int d_oct = kg * (KBF / 8) + d_o;
    const bool d_oct_ok = d_on && d_oct < g.K8;
    const int d_octc = min(d_oct, g.K8 - 1);
    const int d_ni = dp >> g.lgP, d_row = (dp >> g.lgW) & THm, d_col = dp & Wm;
    // x slot s: octet (t + 256 s) >> 6 of the channel group, cell (image, tile row, 4-pixel group) = t & 63 of P
    const int xi = t & 63;
    const bool x_on = xi < g.P;
    int xr_ = min(xi, g.P - 1);
    const int x_q = xr_ % g.q4; xr_ /= g.q4;
    const int x_r = xr_ % g.THi, x_ni = xr_ / g.THi;
    const int x_lds0 = DZSZ + ((x_ni * g.THi + x_r) * g.RS + 8 + 4 * x_q) * 2;

    float dbacc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dbacc[e] = 0.f;
    uint4 dv[4], xv[NX][4];
    uint2 dm[2];
    uint4 dy[2];
    bool dok = false;

    auto gload = [&](int tile) {
        const int gi = tile / g.RT, rt = tile - gi * g.RT;
        const int n0 = gi * g.NI, r0 = rt * g.TH;
        {
            const int n = n0 + d_ni, row = r0 + d_row;
            dok = d_oct_ok && n < g.N;
            const int nn = min(n, g.N - 1);
            if (POOL) {
                const int Hp = g.H >> 1, Wp = g.Wd >> 1;
                const int pc = ((nn * g.K8 + d_octc) * Hp + (row >> 1)) * Wp + (d_col >> 1);
                const uint4* gp = reinterpret_cast<const uint4*>(g.dz);
                dv[0] = gp[pc]; dv[1] = gp[pc + 1];
                dm[0] = reinterpret_cast<const uint2*>(g.mask)[pc];
                dm[1] = reinterpret_cast<const uint2*>(g.mask)[pc + 1];
                if (g.act != TN_ACT_LEAKY) {
                    dy[0] = reinterpret_cast<const uint4*>(g.y)[pc];
                    dy[1] = reinterpret_cast<const uint4*>(g.y)[pc + 1];
                }
            } else {
                const uint4* src = reinterpret_cast<const uint4*>(g.dz) + ((size_t)(nn * g.K8 + d_octc) * g.H + row) * g.Wd + d_col;
#pragma unroll
                for (int i = 0; i < 4; ++i) dv[i] = src[i];
            }
        }
        {
            const int n = n0 + x_ni, row = r0 - 1 + x_r;
            const bool okr = x_on && n < g.N && (unsigned)row < (unsigned)g.H;
            const int nn = min(n, g.N - 1), rr = min(max(row, 0), g.H - 1);
#pragma unroll
            for (int s = 0; s < NX; ++s) {
                const int oct = cg * (CBF / 8) + ((t + 256 * s) >> 6);
                const uint4* src = reinterpret_cast<const uint4*>(g.x) + ((size_t)(nn * g.C8 + min(oct, g.C8 - 1)) * g.H + rr) * g.Wd + 4 * x_q;
                const bool ok = okr && oct < g.C8;
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[s][i] = c8_and4(src[i], ok);
            }
        }
    };
    auto lstore = [&](int buf, float dbw, int tile) {
        char* base = smem + buf * BUFSZ;
        if (d_on) {
            uint4 c[4];
            if (POOL) {
                const int rt = tile % g.RT;
                const int shr = ((rt * g.TH + d_row) & 1) << 1;
                c[0] = c8_pool_cell(dv[0], dm[0], dy[0], shr, g.act, g.prm);
                c[1] = c8_pool_cell(dv[0], dm[0], dy[0], shr | 1, g.act, g.prm);
                c[2] = c8_pool_cell(dv[1], dm[1], dy[1], shr, g.act, g.prm);
                c[3] = c8_pool_cell(dv[1], dm[1], dy[1], shr | 1, g.act, g.prm);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = dv[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = c8_and4(c[i], dok);
            uint2 tr[8];
            c8_transpose4(c, tr);
            const half2v one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                *reinterpret_cast<uint2*>(base + ((d_o * 8 + e) * C8W_DZROW + dp) * 2) = tr[e];
                float sum = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, tr[e].x), one, 0.f, false);
                sum = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, tr[e].y), one, sum, false);
                dbacc[e] += dbw * sum;
            }
        }
        if (x_on) {
#pragma unroll
            for (int s = 0; s < NX; ++s) {
                uint2 tr[8];
                c8_transpose4(xv[s], tr);
                const int o = (t + 256 * s) >> 6;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<uint2*>(base + x_lds0 + (o * 8 + e) * g.plane * 2) = tr[e];
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    gload(tile_beg);
    __syncthreads();                 // the clearing is done
    lstore(0, 1.f, tile_beg);
    __syncthreads();

    const int RS2 = g.RS * 2;
    int cur = 0;
    for (int tile = tile_beg; tile < tile_end; ++tile, cur ^= 1) {
        const bool hasnext = tile + 1 < tile_end;
        const int nxt = hasnext ? tile + 1 : tile;   // (the last tile re-stages itself: branch-free body)
        gload(nxt);
        const char* dzb = smem + cur * BUFSZ + (ft * 32 + l31) * (C8W_DZROW * 2) + 16 * hi;
        const char* xb = smem + cur * BUFSZ + DZSZ + (ct * 32 + l31) * g.plane * 2 + 16;
        int4v av[2], xc[2][3];
        int xl[2][3], xr[2][3];
        auto ops = [&](int slot, int sg) {
            const int p = 16 * sg + 8 * hi;
            av[slot] = *reinterpret_cast<const int4v*>(dzb + 32 * sg);
            const char* xp = xb + (((p >> g.lgP) * g.THi + ((p >> g.lgW) & THm)) * g.RS + (p & Wm)) * 2;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                xl[slot][u] = *reinterpret_cast<const int*>(xp + u * RS2 - 4);
                xc[slot][u] = *reinterpret_cast<const int4v*>(xp + u * RS2);
                xr[slot][u] = *reinterpret_cast<const int*>(xp + u * RS2 + 16);
            }
        };
        ops(0, ps);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int c_ = i & 1, nx_ = c_ ^ 1;
            if (i + 1 < SPW) ops(nx_, ps + PS * (i + 1));
            const half8 a = __builtin_bit_cast(half8, av[c_]);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int4v c = xc[c_][u];
                const int e0 = __builtin_amdgcn_alignbit(c[0], xl[c_][u], 16);
                const int e1 = __builtin_amdgcn_alignbit(c[1], c[0], 16);
                const int e2 = __builtin_amdgcn_alignbit(c[2], c[1], 16);
                const int e3 = __builtin_amdgcn_alignbit(c[3], c[2], 16);
                const int e4 = __builtin_amdgcn_alignbit(xr[c_][u], c[3], 16);
                const int4v b0 = {e0, e1, e2, e3}, b2 = {e1, e2, e3, e4};
                acc[u * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, b0), acc[u * 3 + 0], 0, 0, 0);
                acc[u * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, c), acc[u * 3 + 1], 0, 0, 0);
                acc[u * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, b2), acc[u * 3 + 2], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);       // LDS operands of the next step
            __builtin_amdgcn_sched_group_barrier(0x002, 15, 0);       // this step's funnel shifts
            __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);        // then its MFMAs
        }
        lstore(cur ^ 1, hasnext ? 1.f : 0.f, nxt);
        __syncthreads();
    }

    // bias gradient partial of the slab: per-filter sums of the dz16 this block staged (x 1/gs)
    if (cg == 0 && d_on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = dbacc[e];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            const int k = d_oct * 8 + e;
            if (l31 == 0 && k < g.K) g.dbws[(size_t)z * g.K + k] = v * g.oscale;
        }
    }
    // slab (z, ps): dW layout, tap (u,v) of the correlation is element (2-u, 2-v)
    const int c = cg * CBF + ct * 32 + l31;
    if (c < g.C) {
        float* wz = g.ws + (size_t)(z * PS + ps) * g.K * g.C * 9;
        const float os = g.oscale;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kg * KBF + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) {
#pragma unroll
                for (int a = 0; a < 9; ++a) wz[((size_t)k * g.C + c) * 9 + 8 - a] = acc[a][r] * os;
            }
        }
    }
}

static int c8w_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

static void c8w_tiles(int K, int C, int& NFT, int& NCT) {
    NFT = K > 32 ? 2 : 1;
    NCT = C > 32 ? 2 : 1;
}

static int c8w_geometry(C8WG& g, int num_cus) {
    const int lgW = c8w_log2(g.Wd);
    if (lgW < 3 || lgW > 6) return 0;                  // rows of 8..64 pixels
    int TH = 128 / g.Wd;
    g.NI = 1;
    if (TH > g.H) {
        if (TH % g.H) return 0;
        g.NI = TH / g.H;
        TH = g.H;
    } else if (g.H % TH) {
        return 0;
    }
    if (c8w_log2(TH) < 0) return 0;
    g.TH = TH; g.THi = TH + 2; g.RT = g.H / TH;
    g.lgW = lgW; g.lgP = c8w_log2(TH * g.Wd);
    g.RS = g.Wd + 8;
    g.plane = g.NI * g.THi * g.RS;                     // halfs; 16 bytes * odd apart: conflict-free 16-byte reads
    g.plane += ((g.plane >> 3) & 1) ? 16 : 8;
    g.q4 = g.Wd / 4;
    g.P = g.NI * g.THi * g.q4;
    if (g.P > 64) return 0;
    int NFT, NCT;
    c8w_tiles(g.K, g.C, NFT, NCT);
    g.KG = cdiv(g.K, 32 * NFT);
    g.CG = cdiv(g.C, 32 * NCT);
    g.NTILES = cdiv(g.N, g.NI) * g.RT;
    int S = num_cus / (g.KG * g.CG);
    if (S > g.NTILES) S = g.NTILES;
    if (S < 1) S = 1;
    g.tpb = cdiv(g.NTILES, S);
    g.S = cdiv(g.NTILES, g.tpb);
    return 1;
}

static size_t c8w_lds_bytes(const C8WG& g) {
    int NFT, NCT;
    c8w_tiles(g.K, g.C, NFT, NCT);
    return (size_t)2 * (32 * NFT * C8W_DZROW * 2 + 32 * NCT * g.plane * 2);
}

template <int NFT, int NCT, bool POOL>
static int c8w_launch(tn_ctx* ctx, C8WG& g) {
    static bool attr_set = false;
    if (!attr_set) {
        TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&c8_wgrad_kernel<NFT, NCT, POOL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int grid = 8 * cdiv(g.S, 8) * g.KG * g.CG;
    c8_wgrad_kernel<NFT, NCT, POOL><<<grid, 256, c8w_lds_bytes(g), ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static int c8w_run(tn_ctx* ctx, C8WG& g, float* dW, float* db, bool pool) {
    TN_REQUIRE(c8w_geometry(g, ctx->num_cus) && c8w_lds_bytes(g) <= 160 * 1024, "c8 conv wgrad: unsupported shape");
    TN_REQUIRE((long long)g.N * g.C8 * g.H * g.Wd < (1ll << 28) && (long long)g.N * g.K8 * g.H * g.Wd < (1ll << 28),
               "c8 conv wgrad: tensor too large for 32-bit cell offsets");
    int NFT, NCT;
    c8w_tiles(g.K, g.C, NFT, NCT);
    const int PS = 4 / (NFT * NCT);
    const size_t n = (size_t)g.K * g.C * 9;
    int rc = tn_scratch_get(ctx, ((size_t)g.S * PS * n + (size_t)g.S * g.K) * sizeof(float), &g.ws);
    if (rc) return rc;
    g.dbws = g.ws + (size_t)g.S * PS * n;
    g.oscale = 1.f / ctx->grad_scale;
#define C8W_GO(A, B) rc = pool ? c8w_launch<A, B, true>(ctx, g) : c8w_launch<A, B, false>(ctx, g)
    if (NFT == 2 && NCT == 2) C8W_GO(2, 2);
    else if (NFT == 2) C8W_GO(2, 1);
    else if (NCT == 2) C8W_GO(1, 2);
    else C8W_GO(1, 1);
#undef C8W_GO
    if (rc) return rc;
    rc = tn_red_push(ctx, g.ws, dW, (uint32_t)n, (uint32_t)(g.S * PS), (uint32_t)n, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, g.dbws, db, (uint32_t)g.K, (uint32_t)g.S, (uint32_t)g.K, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

// ---- NCHW fp32 <-> c8 fp16 -------------------------------------------------------------------------------
// one thread = one cell (8 channels of a pixel); rows row0.. of the source (a minibatch window of a dataset)
__global__ __launch_bounds__(256) void c8_pack_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int C, int C8,
                                                     int HW, size_t cells, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cells) return;
    const int p = (int)(i % HW);
    const size_t pl = i / HW;
    const int o = (int)(pl % C8);
    const size_t n = pl / C8;
    half8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = o * 8 + e;
        h[e] = (_Float16)(c < C ? scale * x[(n * C + c) * HW + p] : 0.f);
    }
    reinterpret_cast<half8*>(out)[i] = h;
}
// one thread = 4 consecutive pixels of one channel (16-byte store)
__global__ __launch_bounds__(256) void c8_unpack_kernel(const _Float16* __restrict__ x, float* __restrict__ out, int C,
                                                       int C8, int HW, size_t quads, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const int q4 = HW >> 2;
    const int q = (int)(i % q4);
    const size_t pl = i / q4;
    const int c = (int)(pl % C);
    const size_t n = pl / C;
    const _Float16* src = x + ((n * C8 + (c >> 3)) * HW + 4 * q) * 8 + (c & 7);
    *reinterpret_cast<float4*>(out + (n * C + c) * HW + 4 * q) =
        make_float4(scale * (float)src[0], scale * (float)src[8], scale * (float)src[16], scale * (float)src[24]);
}

extern "C" {

// 1 if the c8 kernels take a 3x3 'same' stride-1 layer of this shape (forward, both gradients)
int tn_c8_conv_supported(int N, int C, int H, int W, int K, int f, int stride, int pad) {
    if (f != 3 || stride != 1 || pad != 1 || (K & 7)) return 0;
    C8G g{};
    g.N = N; g.H = H; g.W = W;
    if (!c8_geometry(g, c8_pick_ft(K), K, C) || c8_lds_bytes(g, c8_pick_ft(K)) > 156 * 1024) return 0;
    C8G d{};
    d.N = N; d.H = H; d.W = W;
    if (!c8_geometry(d, c8_pick_ft(C), C, K) || c8_lds_bytes(d, c8_pick_ft(C)) > 156 * 1024) return 0;
    return 1;
}

// y = act(conv(x, W) + b) [pool != 0: followed by a 2x2 max-pool; mask (may be NULL) records the window elements
// that attained each maximum and the sign of the pooled value]; x, y c8 fp16, W (K, C, 3, 3) and b fp32
int tn_c8_conv_fwd(tn_ctx* ctx, const void* x, const float* W, const float* b, void* y, uint8_t* mask, int N, int C,
                   int H, int Wd, int K, int act, float prm, int pool) {
    TN_REQUIRE((K & 7) == 0, "c8 conv: the number of filters must be a multiple of 8 (got %d)", K);
    C8G g{};
    g.x = static_cast<const _Float16*>(x); g.out = static_cast<_Float16*>(y); g.bias = b; g.mask_out = mask;
    g.N = N; g.C8 = (C + 7) / 8; g.K8 = K / 8; g.H = H; g.W = Wd; g.act = act; g.prm = prm;
    return pool ? c8_run<1>(ctx, g, W, K, C) : c8_run<0>(ctx, g, W, K, C);
}

// dx (N, C, H, W) = conv^T(dz, W) * act'(prev_a) of the layer below (prev_a NULL: no activation below).
// pooled != 0: dz is not a tensor: it is gathered from the pooled gradient g (N, K, H/2, W/2), the block's mask and
// (activations other than leaky-ReLU) its pooled output y, with the block's own (act, prm)
int tn_c8_conv_dgrad(tn_ctx* ctx, const void* dz, const float* W, void* dx, int N, int C, int H, int Wd, int K,
                     const void* prev_a, int prev_act, float prev_prm, int pooled, const uint8_t* mask, const void* y,
                     int act, float prm) {
    TN_REQUIRE((K & 7) == 0, "c8 conv: the number of filters must be a multiple of 8 (got %d)", K);
    C8G g{};
    g.x = static_cast<const _Float16*>(dz); g.out = static_cast<_Float16*>(dx);
    g.prev_a = static_cast<const _Float16*>(prev_a);
    g.N = N; g.C8 = K / 8; g.K8 = (C + 7) / 8; g.H = H; g.W = Wd; g.act = prev_act; g.prm = prev_prm;
    g.mask_in = mask; g.y_in = static_cast<const _Float16*>(y); g.in_act = act; g.in_prm = prm;
    // the roles of filters and channels swap: "filters" = the C input channels (rounded up to whole octets: the
    // arranged weights of channels beyond C are zero, so their cells come out zero)
    return pooled ? c8_run<3>(ctx, g, W, C, K) : c8_run<2>(ctx, g, W, C, K);
}

// dW (K, C, 3, 3), db (K) from x and dz (pooled != 0: from the pooled gradient, the block's mask and pooled output y
// as in tn_c8_conv_dgrad); dz carries the gradient scale, the results do not
int tn_c8_conv_wgrad(tn_ctx* ctx, const void* x, const void* dz, float* dW, float* db, int N, int C, int H, int Wd,
                     int K, int pooled, const uint8_t* mask, const void* y, int act, float prm) {
    TN_REQUIRE((K & 7) == 0, "c8 conv: the number of filters must be a multiple of 8 (got %d)", K);
    C8WG g{};
    g.x = static_cast<const _Float16*>(x); g.dz = static_cast<const _Float16*>(dz);
    g.mask = mask; g.y = static_cast<const _Float16*>(y); g.act = act; g.prm = prm;
    g.N = N; g.C = C; g.C8 = (C + 7) / 8; g.H = H; g.Wd = Wd; g.K = K; g.K8 = K / 8;
    return c8w_run(ctx, g, dW, db, pooled != 0);
}

int tn_c8_conv_wgrad_supported(int N, int C, int H, int Wd, int K) {
    C8WG g{};
    g.N = N; g.C = C; g.C8 = (C + 7) / 8; g.H = H; g.Wd = Wd; g.K = K; g.K8 = K / 8;
    if ((K & 7) || !c8w_geometry(g, 256)) return 0;
    return c8w_lds_bytes(g) <= 160 * 1024;
}

// (N, C, H, W) fp32 rows row0.. of x -> c8 fp16 (values times scale); channels beyond C are zero
int tn_c8_pack(tn_ctx* ctx, const float* x, int64_t row0, void* out, int N, int C, int HW, float scale) {
    const int C8 = (C + 7) / 8;
    const size_t cells = (size_t)N * C8 * HW;
    if (!cells) return TN_OK;
    c8_pack_kernel<<<(unsigned)cdiv(cells, 256), 256, 0, ctx->stream>>>(x + (size_t)row0 * C * HW,
                                                                        static_cast<_Float16*>(out), C, C8, HW, cells, scale);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
// c8 fp16 -> (N, C, H, W) fp32 (values times scale)
int tn_c8_unpack(tn_ctx* ctx, const void* x, float* out, int N, int C, int HW, float scale) {
    TN_REQUIRE((HW & 3) == 0, "tn_c8_unpack: maps of %d pixels", HW);
    const size_t quads = (size_t)N * C * (HW >> 2);
    if (!quads) return TN_OK;
    c8_unpack_kernel<<<(unsigned)cdiv(quads, 256), 256, 0, ctx->stream>>>(static_cast<const _Float16*>(x), out, C,
                                                                          (C + 7) / 8, HW, quads, scale);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
