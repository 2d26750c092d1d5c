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
//     transposition: round 2's operand-rounding kernel spent 32 v_cvt + 8 loads per 4 pixels on that), the im2col is a constant added to
//     an LDS address, and with the filters of a 32-row MFMA tile permuted (bits 2 and 3 of the row swapped, done once by
//     the weight-arranging kernel) a lane's accumulators are two complete octets of its pixel: the epilogue stores
//     16-byte cells straight from registers -- no LDS round trip;
//   * a wave's two 32-pixel groups are the two rows of a 2 x 32 patch, so the 2x2 max-pool of a fused block is one
//     in-lane max and one lane-pair exchange;
//   * gradients travel as fp16(gs * g) (gs = GRAD_SCALE, a power of two; |dz| ~ 1e-3/B is fp16-subnormal territory):
//     scaled ONCE where the first fp16 gradient is produced, unscaled in the fp32 epilogues of the weight gradients.
// The weight gradient (reduction = pixels) wants the other orientation: gfx950's transposing LDS read
// (ds_read_b64_tr_b16) supplies it between LDS and the registers, so its LDS images are plain copies of the c8 tensors.
#include "conv_tile_common.h"

#include <type_traits>

// ablation builds of c8_conv_kernel (wrong results, timing only; hipcc -DC8_FEXP=n into a side library, run through
// TN_HIP_LIB): 1 no input staging, 2 no weight staging, 4 no epilogue (the products go with it: dead code), 8 one tap's
// matrix work instead of nine.  Round 5, wide6 conv2 forward (us): 49.9 as is, 37.6 without input staging, 33.7 with one
// tap of nine, 20.6 staging only, 7.1 the bare loop -- the phases of a block add up.
#ifndef C8_FEXP
#define C8_FEXP 0
#endif
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));

struct C8G {
    const _Float16* x;        // gathered tensor, c8 (N, C8, H, W, 8); MODE 3: the POOLED gradient (N, C8, H/2, W/2, 8)
    const _Float16* wt;       // arranged weights [KT][nchunk][tap][2][32*FT][8]
    _Float16* out;            // c8 (N, K8, H, W, 8); MODE 1: pooled (N, K8, H/2, W/2, 8)
    const float* bias;        // forward
    const _Float16* prev_a;   // input gradient: output of the layer below (same shape as out) or NULL
    uint8_t* mask_out;        // MODE 1: pooling mask (N, K8, H/2, W/2, 8) bytes, may be NULL
    const uint8_t* mask_in;   // MODE 3: pooling mask of the block whose dz is being gathered
    int N, C8, K8, H, W, act;
    float prm;
    int KT, MT, RT, NI, TH, THi, RS, plane, nchunk, TP, nslots, nwork;
    unsigned mKT, mRT;        // 2^32 / KT + 1, 2^32 / RT + 1 (0: divisor 1): the work-item decode without integer divisions
    unsigned long long* dbg;  // TN_C8_DBG=1: per block {start, prologue done, loop done, end} (s_memtime) + wall clock
};

__host__ __device__ __forceinline__ int c8_swap23(int j) { return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1); }

// wt[kt][chunk][tap][o][j][e] (halfs): MFMA row j of filter tile kt holds filter kt*KBF + (j & ~31) + swap23(j & 31)
// (so that a lane's accumulators 0-7 / 8-15 are whole octets), channel chunk*16 + 8*o + e, correlation tap.
// Tap-packed form (tk: forward of a first layer, C <= 8 = one octet): wt[kt][tap pair jp < 5][o][j][e] = tap 2 jp + o,
// channel e -- the 16 reduction indices of an MFMA step are two taps x 8 channels instead of 16 channels of one tap.
__device__ __forceinline__ float c8_wt_value(const float* __restrict__ W, int K, int C, int KBF, int nchunk, int idx,
                                             int dgrad, int tk) {
    int r = idx;
    const int e = r & 7; r >>= 3;
    const int j = r % KBF; r /= KBF;
    const int o = r & 1; r >>= 1;
    if (tk) {
        const int jp = r % 5, kt = r / 5, tap = 2 * jp + o;
        const int filt = kt * KBF + (j & ~31) + c8_swap23(j & 31);
        return (filt < K && e < C && tap < 9) ? W[((size_t)filt * C + e) * 9 + (8 - tap)] : 0.f;
    }
    const int tap = r % 9; r /= 9;
    const int chunk = r % nchunk;
    const int kt = r / nchunk;
    const int filt = kt * KBF + (j & ~31) + c8_swap23(j & 31), ch = chunk * 16 + 8 * o + e;
    if (filt < K && ch < C)
        return dgrad ? W[((size_t)ch * K + filt) * 9 + tap]          // W[k = ch][c = filt][u][v]
                     : W[((size_t)filt * C + ch) * 9 + (8 - tap)];   // true convolution: flipped taps
    return 0.f;
}
__global__ __launch_bounds__(256) void c8_wt_kernel(const float* __restrict__ W, _Float16* __restrict__ wt, int K, int C,
                                                   int KBF, int nchunk, int total, int dgrad, int tk) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    wt[idx] = (_Float16)c8_wt_value(W, K, C, KBF, nchunk, idx, dgrad, tk);
}

__device__ __forceinline__ uint4 c8_and4(uint4 v, bool ok) {
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}

// dz cell of a pooled block at window element sh (= 2*(row & 1) + (col & 1)): the pooled gradient cell where bit sh of the
// channel's mask byte is set (that window element attained the maximum; all of them on a tie: Theano's MaxPoolGrad),
// zero elsewhere.  The pooled gradient already carries act'(pooled output): whichever kernel produced it multiplied by
// the derivative taken from the block's stored output (the backward fusion rule of DESIGN.md section 4).
__device__ __forceinline__ uint4 c8_pool_cell(const uint4 g8, const uint2 m8, int sh) {
    const unsigned b0 = (m8.x >> sh) & 0x01010101u, b1 = (m8.y >> sh) & 0x01010101u;    // one byte per channel: 0 / 1
    // two channels per dword: bytes -> 16-bit all-ones masks (v_perm spreads, x 0xffff fills)
    const unsigned k0 = __umul24(__builtin_amdgcn_perm(0u, b0, 0x0c010c00u), 0xffffu);
    const unsigned k1 = __umul24(__builtin_amdgcn_perm(0u, b0, 0x0c030c02u), 0xffffu);
    const unsigned k2 = __umul24(__builtin_amdgcn_perm(0u, b1, 0x0c010c00u), 0xffffu);
    const unsigned k3 = __umul24(__builtin_amdgcn_perm(0u, b1, 0x0c030c02u), 0xffffu);
    return make_uint4(g8.x & k0, g8.y & k1, g8.z & k2, g8.w & k3);
}

// Pooled epilogue of the leaky-ReLU family for FOUR channels of a lane's octet, hand-scheduled (round 4).  z0 / z1: the
// raw sums of the lane's two pixels (rows 2r, 2r + 1 of its column), partner lane (^ 1) = the other column of the window.
// Per channel: window maximum (one in-lane v_max, one v_max_f32_dpp), tie bits of the own column (kA / kB = 1 << dj /
// 4 << dj where z0 / z1 attains it) merged with the partner's (v_or_b32_dpp), ONE activation (the maximum of the
// activations is the activation of the maximum), sign bits 16 / 32, conversion.  hipcc compiled the C++ statement of the
// same thing channel by channel through three temporaries and VCC: 22 vector instructions + 7 s_nop per channel; here two
// channels advance in lockstep (every DPP / SGPR hazard distance is met by the partner channel's instructions): 16.75.
// o01 / o23: the four activated maxima as halfs; mk: the four mask bytes.
#define C8_DPPQ "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define C8_EPI_PAIR(O, ZA0, ZB0, ZA1, ZB1)                                                          \
    "v_max_f32_e32 %3, " ZA0 ", " ZA1 "\n\t"                                                        \
    "v_max_f32_e32 %4, " ZB0 ", " ZB1 "\n\t"                                                        \
    "s_nop 0\n\t"                                                                                   \
    "v_max_f32_dpp %3, %3, %3 " C8_DPPQ "\n\t"                                                      \
    "v_max_f32_dpp %4, %4, %4 " C8_DPPQ "\n\t"                                                      \
    "v_cmp_eq_f32_e64 %11, " ZA0 ", %3\n\t"                                                         \
    "v_cmp_eq_f32_e64 %12, " ZA1 ", %3\n\t"                                                         \
    "v_cmp_eq_f32_e64 %13, " ZB0 ", %4\n\t"                                                         \
    "v_cmp_eq_f32_e64 %14, " ZB1 ", %4\n\t"                                                         \
    "v_mul_f32_e32 %9, %23, %3\n\t"                                                                 \
    "v_mul_f32_e32 %10, %23, %4\n\t"                                                                \
    "v_cndmask_b32_e64 %5, 0, %24, %11\n\t"                                                         \
    "v_cndmask_b32_e64 %7, 0, %25, %12\n\t"                                                         \
    "v_cndmask_b32_e64 %6, 0, %24, %13\n\t"                                                         \
    "v_cndmask_b32_e64 %8, 0, %25, %14\n\t"                                                         \
    "v_max_f32_e32 %9, %3, %9\n\t"                                                                  \
    "v_max_f32_e32 %10, %4, %10\n\t"                                                                \
    "v_or_b32_e32 %5, %5, %7\n\t"                                                                   \
    "v_or_b32_e32 %6, %6, %8\n\t"                                                                   \
    "v_cmp_lt_f32_e64 %11, 0, %9\n\t"                                                               \
    "v_cmp_gt_f32_e64 %12, 0, %9\n\t"                                                               \
    "v_cmp_lt_f32_e64 %13, 0, %10\n\t"                                                              \
    "v_cmp_gt_f32_e64 %14, 0, %10\n\t"                                                              \
    "v_or_b32_dpp %5, %5, %5 " C8_DPPQ "\n\t"                                                       \
    "v_or_b32_dpp %6, %6, %6 " C8_DPPQ "\n\t"                                                       \
    "v_cndmask_b32_e64 %7, 0, 16, %11\n\t"                                                          \
    "v_cndmask_b32_e64 %3, 0, 32, %12\n\t"                                                          \
    "v_cndmask_b32_e64 %8, 0, 16, %13\n\t"                                                          \
    "v_cndmask_b32_e64 %4, 0, 32, %14\n\t"                                                          \
    "v_or3_b32 %5, %5, %7, %3\n\t"                                                                  \
    "v_or3_b32 %6, %6, %8, %4\n\t"                                                                  \
    "v_cvt_pk_f16_f32 " O ", %9, %10\n\t"
__device__ __forceinline__ void c8_pool_epi4(unsigned& o01, unsigned& o23, unsigned& mk, const float (&z0)[4], const float (&z1)[4],
                                             float prm, unsigned kA, unsigned kB) {
    unsigned t3, t4, t5, t6, t7, t8, t9, t10;
    unsigned long long s11, s12, s13, s14;
    asm volatile(C8_EPI_PAIR("%0", "%15", "%16", "%19", "%20")
                 "v_lshl_or_b32 %2, %6, 8, %5\n\t"
                 C8_EPI_PAIR("%1", "%17", "%18", "%21", "%22")
                 "v_lshl_or_b32 %5, %6, 8, %5\n\t"
                 "v_lshl_or_b32 %2, %5, 16, %2"
                 : "=&v"(o01), "=&v"(o23), "=&v"(mk), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8),
                   "=&v"(t9), "=&v"(t10), "=&s"(s11), "=&s"(s12), "=&s"(s13), "=&s"(s14)
                 : "v"(z0[0]), "v"(z0[1]), "v"(z0[2]), "v"(z0[3]), "v"(z1[0]), "v"(z1[1]), "v"(z1[2]), "v"(z1[3]), "s"(prm),
                   "v"(kA), "v"(kB));
}

// MODE 0: forward (bias + act); 1: forward + 2x2 max-pool + mask; 2: input gradient (x act' of the layer below);
// 3: input gradient of a pooled block, dz gathered from (g, mask, y).
// PERSISTENT blocks: a block walks over work items w = blockIdx.x, + gridDim.x, ... (work item = one 256-pixel tile x one
// filter tile, decoded XCD-aware: the filter tiles of a pixel tile share an L2) with ONE software pipeline across
// them -- (work item, chunk) pairs form a flat sequence, input cells are fetched two steps ahead and weights one step
// ahead whatever tile they belong to, so a tile's epilogue stores and the next tile's first loads overlap the matrix
// work instead of bracketing it (cycle stamps of the one-tile-per-block form, conv2 of wide6: prologue 5.4 k + main
// loop 11.5 k + epilogue 5.5 k cycles per block, all blocks of a round in the same phase).
template <int FT, int MODE, int NS, bool LK, bool TK = false>
__global__ __launch_bounds__(256, 2) void c8_conv_kernel(C8G g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr bool DGRAD = MODE >= 2;
    constexpr bool POOLED = MODE == 3;
    constexpr int KBF = 32 * FT;
    // TK ("tap-packed", forward of a first layer: C <= 8 = one octet plane): an MFMA step reduces over 2 taps x 8 channels
    // instead of 16 channels of one tap -- 5 tap steps instead of 9, one input plane staged instead of two
    constexpr int NTS = TK ? 5 : 9;
    constexpr int WB = NTS * 2 * KBF * 16;            // bytes of one weight chunk
    constexpr int WS = (WB / 16 + 255) / 256;         // 16-byte staging slots per thread (3 or 5)
    const int XB = (TK ? 1 : 2) * g.plane * 16;       // bytes of one input chunk (two octet planes)
    char* const Xs = reinterpret_cast<char*>(ct_smem);            // [2][XB]
    char* const Ws = Xs + 2 * XB;                                 // [2][WB]
    const int bid = blockIdx.x, G = gridDim.x;
    const int nwork = (g.nwork - bid + G - 1) / G;    // work items of this block (>= 1: the grid is not larger than nwork)
    const int SEQ = nwork * g.nchunk;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int HW = g.H * g.W;
    unsigned long long* dbg = (g.dbg && t == 0) ? g.dbg + 8 * (size_t)bid : nullptr;
    if (dbg) { dbg[0] = __builtin_readcyclecounter(); dbg[4] = wall_clock64(); }

    // halo columns and the window overhang stay zero for the life of the block; every other cell is rewritten per step
    for (int i = t * 16; i < 2 * XB; i += 4096) *reinterpret_cast<float4*>(Xs + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- work item i of this block -> tile ----
    struct Tile { int n0, r0, kt, base; bool dead; };
    auto decode = [&](int i) __attribute__((always_inline)) {
        const int w = bid + min(i, nwork - 1) * G, xcd = w & 7, idx = w >> 3;
        // (a body of the loop decodes three work items; as divisions that was ~120 of its ~300 scalar instructions, and the
        // loop is bound by instruction issue: 670 instructions per 36 products)
        // x / d = umulhi(x, 2^32 / d + 1), exact while x * d < 2^32 (the host checks the work-item count)
        const int iq = g.mKT ? (int)__umulhi((unsigned)idx, g.mKT) : idx;
        const int mt = iq * 8 + xcd;
        Tile tl;
        tl.kt = idx - iq * g.KT;
        tl.dead = mt >= g.MT;                         // (MT not a multiple of 8: computed, never stored)
        const int mtc = min(mt, g.MT - 1), grp = g.mRT ? (int)__umulhi((unsigned)mtc, g.mRT) : mtc, rt = mtc - grp * g.RT;
        tl.n0 = grp * g.NI; tl.r0 = rt * g.TH;
        tl.base = POOLED ? tl.n0 * g.C8 * (HW >> 2) + (tl.r0 >> 1) * (g.W >> 1) : tl.n0 * g.C8 * HW + tl.r0 * g.W;
        return tl;
    };

    // ---- staging slots of this thread (the same for every tile): slot e = (octet of the chunk, image, tile row, column)
    // -> LDS cell, cell offset relative to the tile's first cell, flags: bit 0 in use, 1 / 2 top / bottom halo row,
    // 3-4 window element (2*(row&1) + (col&1)), 5 octet, 8.. image of the tile
    int sl_l[NS], sl_rel[NS], sl_fl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = t + 256 * s;
        int rr = min(e, g.nslots - 1);
        const int col = rr % g.W; rr /= g.W;
        const int r = rr % g.THi; rr /= g.THi;
        const int ni = rr % g.NI;
        const int o = rr / g.NI;
        sl_l[s] = (o * g.plane + (ni * g.THi + r) * g.RS + 1 + col) * 16;
        sl_rel[s] = POOLED ? (ni * g.C8 + o) * (HW >> 2) + ((r - 1) >> 1) * (g.W >> 1) + (col >> 1)
                           : (ni * g.C8 + o) * HW + (r - 1) * g.W + col;
        sl_fl[s] = (e < g.nslots ? 1 : 0) | (r == 0 ? 2 : 0) | (r == g.THi - 1 ? 4 : 0) | ((((r - 1) & 1) << 1 | (col & 1)) << 3) |
                   (o << 5) | (ni << 8);
    }
    auto slot_ok = [&](int s, const Tile& tl) __attribute__((always_inline)) {
        // bitwise on purpose: the short-circuit form compiled to a chain of exec-mask branches per slot (~25 instructions
        // each, in a loop that is bound by instruction issue)
        const int fl = sl_fl[s];
        const int edge = (tl.r0 == 0 ? 2 : 0) | (tl.r0 + g.TH >= g.H ? 4 : 0);          // wave-uniform
        return ((fl & 1) != 0) & ((fl & edge) == 0) & (tl.n0 + (fl >> 8) < g.N);
    };
    const uint4* xg = reinterpret_cast<const uint4*>(g.x);

    // this lane's two pixels: the two rows of a (2 x W') patch, same column (so that pooling is in-lane)
    int boff[2], prel[2], pni[2];
    bool pin[2];
    const int L = wave * 32 + l31, pair = L / g.W, pcol = L - pair * g.W;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int R = 2 * pair + pt;
        pin[pt] = R < g.NI * g.TH;
        const int Rc = pin[pt] ? R : 0;
        const int ni = Rc / g.TH, r = Rc - ni * g.TH;
        pni[pt] = ni; prel[pt] = r;
        boff[pt] = ((TK ? 0 : hi * g.plane) + (ni * g.THi + r) * g.RS + pcol) * 16;
    }
    const int aoff = (hi * KBF + l31) * 16;

    // accumulators start from the bias (forward) / zero (gradients): row h*8+e of tile f is filter
    // 8*(kt*KBF/8 + 4f + 2h + hi) + e.  The bias of the NEXT tile is fetched under the last chunk of the current one.
    f32x16 acc[FT][2];
    float4 bq[FT][2][2];
    auto bias_load = [&](int kt) __attribute__((always_inline)) {
        if (DGRAD) return;
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ob = min(kt * (KBF / 8) + f * 4 + h * 2 + hi, g.K8 - 1) * 8;
                bq[f][h][0] = *reinterpret_cast<const float4*>(g.bias + ob);
                bq[f][h][1] = *reinterpret_cast<const float4*>(g.bias + ob + 4);
            }
    };
    auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (DGRAD) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[f][b][h * 8 + e] = 0.f;
                    } else {
                        acc[f][b][h * 8 + 0] = bq[f][h][0].x; acc[f][b][h * 8 + 1] = bq[f][h][0].y;
                        acc[f][b][h * 8 + 2] = bq[f][h][0].z; acc[f][b][h * 8 + 3] = bq[f][h][0].w;
                        acc[f][b][h * 8 + 4] = bq[f][h][1].x; acc[f][b][h * 8 + 5] = bq[f][h][1].y;
                        acc[f][b][h * 8 + 6] = bq[f][h][1].z; acc[f][b][h * 8 + 7] = bq[f][h][1].w;
                    }
                }
    };

    // input cells two steps ahead (two register sets alternating by step parity), weights (L2) one step ahead
    uint4 xr[2][NS];
    uint2 mr[2][NS];
    unsigned okm[2] = {0u, 0u};                       // which slots of the register set hold cells inside their image
    uint4 wr0, wr1, wr2, wr3, wr4;
    // cursors over the flat (work item, chunk) sequence: input loads, weights / LDS stores, matrix work
    int xi = 0, xch = 0, si = 0, sch = 0, ci = 0, cch = 0;
    Tile tx = decode(0);
    int skt = tx.kt;
    bias_load(tx.kt);
    acc_init();
#define C8_WL(J, R) if (WS > J) R = *reinterpret_cast<const uint4*>(w_ + min(4096 * J, WB - 16 - 16 * t))
#define C8_WST(J, R) if (WS > J && (4096 * (J + 1) <= WB || 16 * t + 4096 * J < WB)) *reinterpret_cast<uint4*>(wb + 4096 * J) = R
    auto gloadx = [&](auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        unsigned m = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = (sl_fl[s] >> 5) & 1;
            // octets beyond C8 meet zero weights: any finite value will do (clamped re-read)
            const int oc = min(2 * xch + o, g.C8 - 1) - o;
            const bool ok = slot_ok(s, tx);
            m |= ok ? (1u << s) : 0u;
            const int pc = tx.base + (ok ? sl_rel[s] + oc * (POOLED ? (HW >> 2) : HW) : 0);
            xr[P][s] = xg[pc];
            if (POOLED) mr[P][s] = reinterpret_cast<const uint2*>(g.mask_in)[pc];
        }
        okm[P] = m;
        if (++xch == g.nchunk) { xch = 0; tx = decode(++xi); }
    };
    // one filter tile, one chunk (a first layer; any C <= 16, K <= 32 FT): every step of the block meets the SAME weights
    // and biases -- both LDS buffers are filled once, the bias registers loaded once (the loop is bound by the instructions
    // a wave issues, and such a layer runs one body per tile)
    // (only in the tap-packed instantiations, TK: as a run-time flag in every instantiation its branches around the weight
    // loads cost the multi-chunk layers' loop its schedule -- wide6 conv5 input gradient 34.0 -> 38.0 us, round 5 bisection)
    const bool wconst = TK && g.KT == 1 && g.nchunk == 1;
    auto gloadw = [&]() __attribute__((always_inline)) {
        if (wconst) return;
        const char* w_ = reinterpret_cast<const char*>(g.wt) + ((size_t)skt * g.nchunk + sch) * WB + 16 * t;
        C8_WL(0, wr0); C8_WL(1, wr1); C8_WL(2, wr2); C8_WL(3, wr3); C8_WL(4, wr4);
    };
    auto sadv = [&]() __attribute__((always_inline)) {
        if (++sch == g.nchunk) { sch = 0; skt = decode(++si).kt; }
    };
    auto lstore = [&](int buf, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        char* xb = Xs + buf * XB;
        char* wb = Ws + buf * WB + 16 * t;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (sl_fl[s] & 1) {
                uint4 v = xr[P][s];
                if (POOLED) v = c8_pool_cell(v, mr[P][s], (sl_fl[s] >> 3) & 3);
                // (a cell outside its image is written too: the buffer held another tile's rows a step ago)
                *reinterpret_cast<uint4*>(xb + sl_l[s]) = c8_and4(v, (okm[P] >> s) & 1u);
            }
        }
        if (!wconst) { C8_WST(0, wr0); C8_WST(1, wr1); C8_WST(2, wr2); C8_WST(3, wr3); C8_WST(4, wr4); }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // ---- epilogue of the tile the matrix work has just finished: accumulators 0-7 / 8-15 of a lane are octets
    // (4f + hi) / (4f + 2 + hi) of its pixel.  (The leaky-ReLU family -- every default of the reference, convpool.py:19 --
    // as straight-line code: with a run-time kind every element paid the whole switch of tn_act_fwd /
    // tn_act_grad_from_out.)
    const int Ho = g.H, Wo = g.W;
    auto epilogue = [&](const Tile& tl) __attribute__((always_inline)) {
        const float prm = g.prm, tie = prm > 0.f ? 1.f + prm : 0.f;
        auto actf = [&](float z) __attribute__((always_inline)) {
            if (!LK) return tn_act_fwd(z, g.act, prm);
            // max(z, prm*z) = max(0,z) + min(0,z)*prm for 0 <= prm < 1; the instruction itself: fmaxf() costs a second
            // v_max (sNaN canonicalisation of its operands) per value
            float r;
            const float zp = z * prm;
            asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(z), "v"(zp));
            return r;
        };
        auto actg = [&](float a) __attribute__((always_inline)) {
            return LK ? (a > 0.f ? 1.f : (a < 0.f ? prm : tie)) : tn_act_grad_from_out(a, g.act, prm);
        };
        const int kt = tl.kt;
        bool pok[2];
        int prow[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            prow[pt] = tl.r0 + prel[pt];
            pok[pt] = pin[pt] && !tl.dead && tl.n0 + pni[pt] < g.N && prow[pt] < g.H;
        }
        if (MODE == 1) {
            // conv + act + 2x2 max-pool: vertical max in-lane (the lane's two pixels), horizontal with lane ^ 1
            const int Hp = Ho >> 1, Wp = Wo >> 1;
            const bool ok = pok[0];
            const int dj = l31 & 1;
            // (cell indices fit 32 bits: c8_run checks N * K8 * H * W < 2^28 -- 64-bit products here were ~12 instructions per cell)
            const unsigned pbase = ((unsigned)(tl.n0 + pni[0]) * g.K8 * Hp + (prow[0] >> 1)) * Wp + (pcol >> 1);
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int oct = kt * (KBF / 8) + f * 4 + h * 2 + hi;
                    if (LK) {
                        // (leaky-ReLU does not decrease: the maximum of the four activations is the activation of the maximum,
                        // and -- slope > 0 -- the elements that attain one attain the other; with slope 0 a window of negatives
                        // marks its largest element instead of all four: every one of them receives g * act'(0) = 0 either way,
                        // the derivative is taken from the stored output)
                        int4v o4;
                        uint2 m2;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float z0[4] = {acc[f][0][h * 8 + 4 * q], acc[f][0][h * 8 + 4 * q + 1], acc[f][0][h * 8 + 4 * q + 2],
                                                 acc[f][0][h * 8 + 4 * q + 3]};
                            const float z1[4] = {acc[f][1][h * 8 + 4 * q], acc[f][1][h * 8 + 4 * q + 1], acc[f][1][h * 8 + 4 * q + 2],
                                                 acc[f][1][h * 8 + 4 * q + 3]};
                            unsigned a, b, mk;
                            c8_pool_epi4(a, b, mk, z0, z1, prm, 1u << dj, 4u << dj);
                            o4[2 * q] = (int)a; o4[2 * q + 1] = (int)b;
                            if (q == 0) m2.x = mk; else m2.y = mk;
                        }
                        if (ok && dj == 0 && oct < g.K8) {
                            const unsigned o = pbase + (unsigned)oct * Hp * Wp;
                            reinterpret_cast<int4v*>(g.out)[o] = o4;
                            if (g.mask_out) reinterpret_cast<uint2*>(g.mask_out)[o] = m2;
                        }
                        continue;
                    }
                    half8 o8;
                    unsigned mb[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float m;
                        unsigned bits;
                        {
                            const float a0 = actf(acc[f][0][h * 8 + e]);
                            const float a1 = actf(acc[f][1][h * 8 + e]);
                            const float mv = fmaxf(a0, a1);
                            m = fmaxf(mv, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                              0, __builtin_bit_cast(int, mv), 0xB1, 0xF, 0xF, false)));
                            bits = (a0 == m ? (1u << dj) : 0u) | (a1 == m ? (4u << dj) : 0u);
                            bits |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)bits, 0xB1, 0xF, 0xF, false);
                        }
                        bits |= (m > 0.f ? 16u : 0u) | (m < 0.f ? 32u : 0u);
                        o8[e] = (_Float16)m;
                        mb[e] = bits;
                    }
                    if (ok && dj == 0 && oct < g.K8) {
                        const unsigned o = pbase + (unsigned)oct * Hp * Wp;
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
            const unsigned pbase = ((unsigned)(tl.n0 + pni[pt]) * g.K8 * Ho + prow[pt]) * Wo + pcol;
            half8 pa[FT][2];
            if (DGRAD && g.prev_a && pok[pt]) {
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int oct = min(kt * (KBF / 8) + f * 4 + h * 2 + hi, g.K8 - 1);
                        pa[f][h] = reinterpret_cast<const half8*>(g.prev_a)[pbase + (unsigned)oct * Ho * Wo];
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
                            if (g.prev_a) v *= actg((float)pa[f][h][e]);
                            o8[e] = (_Float16)v;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = (_Float16)actf(acc[f][pt][h * 8 + e]);
                    }
                    if (pok[pt] && oct < g.K8) reinterpret_cast<half8*>(g.out)[pbase + (unsigned)oct * Ho * Wo] = o8;
                }
        }
    };

    gloadx(P0{});
    {
        const char* w_ = reinterpret_cast<const char*>(g.wt) + ((size_t)skt * g.nchunk + sch) * WB + 16 * t;
        C8_WL(0, wr0); C8_WL(1, wr1); C8_WL(2, wr2); C8_WL(3, wr3); C8_WL(4, wr4);
    }
    gloadx(P1{});
    __syncthreads();                 // the clearing is done
    lstore(0, P0{});
    {
        char* wb = Ws + 16 * t;
        C8_WST(0, wr0); C8_WST(1, wr1); C8_WST(2, wr2); C8_WST(3, wr3); C8_WST(4, wr4);
        if (wconst) {
            wb = Ws + WB + 16 * t;
            C8_WST(0, wr0); C8_WST(1, wr1); C8_WST(2, wr2); C8_WST(3, wr3); C8_WST(4, wr4);
        }
    }
    __syncthreads();
    if (dbg) dbg[1] = __builtin_readcyclecounter();
    const int RS16 = g.RS * 16;
    int toff[5];                 // TK: tap step t of this lane = tap 2 t + hi (tap 9 meets zero weights: any cell will do)
#pragma unroll
    for (int t_ = 0; t_ < 5; ++t_) {
        const int tp = min(2 * t_ + hi, 8);
        toff[t_] = (tp / 3) * RS16 + (tp % 3) * 16;
    }
    unsigned long long d_ls = 0, d_ep = 0, d_bar = 0;          // TN_C8_DBG: cycles in LDS stores / epilogues / barriers
    auto body = [&](int seq, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        sadv();
        if (!(C8_FEXP & 2)) gloadw();                    // step seq + 1
        if (!(C8_FEXP & 1)) gloadx(Pc);                  // step seq + 2
        if (!DGRAD && !wconst && cch + 1 == g.nchunk) bias_load(decode(ci + 1).kt);
        const char* x0 = Xs + P * XB + boff[0];
        const char* x1 = Xs + P * XB + boff[1];
        const char* Wb = Ws + P * WB + aoff;
        // nine taps (TK: five tap pairs): the LDS operands of tap s+1 (FT A vectors, 2 B vectors of 8 halfs) are requested
        // before the 2*FT MFMAs of tap s are issued
        half8 a[2][FT], b[2][2];
#pragma unroll
        for (int f = 0; f < FT; ++f) a[0][f] = *reinterpret_cast<const half8*>(Wb + f * 512);
        b[0][0] = *reinterpret_cast<const half8*>(x0 + (TK ? toff[0] : 0));
        b[0][1] = *reinterpret_cast<const half8*>(x1 + (TK ? toff[0] : 0));
        __builtin_amdgcn_sched_group_barrier(0x100, FT + 2, 0);
#pragma unroll
        for (int tap = 0; tap < ((C8_FEXP & 8) ? 1 : NTS); ++tap) {
            const int cur = tap & 1, nx = cur ^ 1;
            if (tap + 1 < NTS) {
                const int bo = TK ? toff[(tap + 1) % NTS] : ((tap + 1) / 3) * RS16 + ((tap + 1) % 3) * 16;
#pragma unroll
                for (int f = 0; f < FT; ++f)
                    a[nx][f] = *reinterpret_cast<const half8*>(Wb + (tap + 1) * (2 * KBF * 16) + f * 512);
                b[nx][0] = *reinterpret_cast<const half8*>(x0 + bo);
                b[nx][1] = *reinterpret_cast<const half8*>(x1 + bo);
            }
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][f], b[cur][0], acc[f][0], 0, 0, 0);
                acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][f], b[cur][1], acc[f][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, FT + 2, 0);       // DS reads of the next tap
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * FT, 0);       // then this tap's MFMAs
        }
        // step seq + 1 (fetched a step ago into the other register set) goes into the other LDS buffer
        unsigned long long s0 = 0, s1 = 0, s2 = 0;
        if (g.dbg) s0 = __builtin_readcyclecounter();
        if (seq + 1 < SEQ && !(C8_FEXP & 1)) lstore(P ^ 1, std::integral_constant<int, P ^ 1>{});
        if (g.dbg) s1 = __builtin_readcyclecounter();
        if (++cch == g.nchunk) {
            const Tile tc = decode(ci);
            asm volatile("; c8 epilogue begin");
            if (!(C8_FEXP & 4)) epilogue(tc);
            asm volatile("; c8 epilogue end");
            acc_init();
            cch = 0; ++ci;
        }
        if (g.dbg) s2 = __builtin_readcyclecounter();
        __syncthreads();
        if (g.dbg) { d_ls += s1 - s0; d_ep += s2 - s1; d_bar += __builtin_readcyclecounter() - s2; }
    };
    for (int seq = 0; seq < SEQ; seq += 2) {
        body(seq, P0{});
        if (seq + 1 < SEQ) body(seq + 1, P1{});
    }
#undef C8_WL
#undef C8_WST
    if (dbg) {
        dbg[2] = __builtin_readcyclecounter(); dbg[5] = wall_clock64();
        dbg[3] = d_ls; dbg[6] = d_ep; dbg[7] = d_bar;
    }
}

// geometry of the pixel tiling; 0 when the shape is outside the kernel's limits
static int c8_geometry(C8G& g, int FT, int K, int C, bool tk = false) {
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
    g.nslots = (tk ? 1 : 2) * g.NI * g.THi * W;
    if (g.nslots > 4 * 256) return 0;
    g.nchunk = cdiv(C, 16);
    g.KT = cdiv(K, 32 * FT);
    g.MT = cdiv(g.N, g.NI) * g.RT;
    g.mKT = g.KT == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)g.KT + 1u);
    g.mRT = g.RT == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)g.RT + 1u);
    if ((long long)(g.MT + 8) * g.KT * (g.KT > g.RT ? g.KT : g.RT) >= (1ll << 31)) return 0;      // (magic divisions exact)
    return 1;
}

static int c8_pick_ft(int K) { return K > 32 ? 2 : 1; }

static size_t c8_lds_bytes(const C8G& g, int FT) { return (size_t)2 * (2 * g.plane * 16 + 9 * 2 * 32 * FT * 16); }

static unsigned long long* c8_dbg_buf = nullptr;
extern "C" int tn_c8_dbg_read(tn_ctx* ctx, unsigned long long* host, int nblocks) {
    if (!c8_dbg_buf) return -1;
    (void)ctx;
    return hipMemcpy(host, c8_dbg_buf, (size_t)nblocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

template <int FT, int MODE, bool TK = false>
static int c8_launch(tn_ctx* ctx, C8G& g) {
    const size_t lds = c8_lds_bytes(g, FT);
    const int ns = cdiv(g.nslots, 256);
    g.nwork = 8 * cdiv(g.MT, 8) * g.KT;
    // two resident blocks per CU, each walking over its work items; three for the 32-filter kernels (their registers and
    // LDS allow it): another wave per SIMD under their epilogues
    int grid = 8 * ((FT == 1 && 3 * lds <= 150 * 1024 ? 3 : 2) * ctx->num_cus / 8);
    if (grid > g.nwork) grid = g.nwork;
    static int dbg_on = -1;
    if (dbg_on < 0) {
        const char* e = getenv("TN_C8_DBG");
        dbg_on = e ? atoi(e) : 0;
    }
    if (dbg_on) {
        if (!c8_dbg_buf) TN_HIP(hipMalloc(&c8_dbg_buf, 8 * sizeof(unsigned long long) * 65536));
        TN_HIP(hipMemsetAsync(c8_dbg_buf, 0, 8 * sizeof(unsigned long long) * 65536, ctx->stream));
        g.dbg = grid <= 65536 ? c8_dbg_buf : nullptr;
    }
#define C8_GO(NS, LK)                                                                                         \
    {                                                                                                         \
        static bool attr_set = false;                                                                         \
        if (!attr_set) {                                                                                      \
            TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&c8_conv_kernel<FT, MODE, NS, LK, TK>),  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));              \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        c8_conv_kernel<FT, MODE, NS, LK, TK><<<grid, 256, lds, ctx->stream>>>(g);                             \
    }
    // the epilogue's activation (forward: the layer's own; gradients: that of the layer below) is a compile-time
    // leaky-ReLU for the reference's defaults; other kinds share one generic instantiation per mode
    const bool lk = g.act == TN_ACT_LEAKY && g.prm >= 0.f && g.prm < 1.f;
    if (!lk) C8_GO(4, false)
    else if (ns <= 2) C8_GO(2, true)
    else if (ns == 3) C8_GO(3, true)
    else C8_GO(4, true)
#undef C8_GO
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// halfs of the arranged weights of a (K filters, C channels) product
static size_t c8_wt_elems(int K, int C) {
    const int KBF = 32 * c8_pick_ft(K);
    return (size_t)cdiv(K, KBF) * cdiv(C, 16) * 9 * 2 * KBF * 8;
}

// halfs the arranging kernels write for a (K filters, C channels) product; tk: the tap-packed forward form (C <= 8)
static bool c8_tap_packed(int C, int dgrad) { return !dgrad && C <= 8; }
static size_t c8_wt_total(int K, int C, int dgrad) {
    const int KBF = 32 * c8_pick_ft(K);
    return c8_tap_packed(C, dgrad) ? (size_t)cdiv(K, KBF) * 5 * 2 * KBF * 8 : c8_wt_elems(K, C);
}

template <int MODE>
static int c8_run(tn_ctx* ctx, C8G& g, const float* W, int K, int C, const void* wt_ready) {
    const int FT = c8_pick_ft(K);
    const bool tk = MODE < 2 && c8_tap_packed(C, 0);
    TN_REQUIRE(c8_geometry(g, FT, K, C, tk) && c8_lds_bytes(g, FT) <= 156 * 1024, "c8 conv: unsupported shape %dx%d", g.H, g.W);
    TN_REQUIRE((long long)g.N * g.C8 * g.H * g.W < (1ll << 28) && (long long)g.N * g.K8 * g.H * g.W < (1ll << 28),
               "c8 conv: tensor too large for 32-bit cell offsets");
    if (wt_ready) {
        g.wt = static_cast<const _Float16*>(wt_ready);            // arranged beforehand (tn_c8_arrange_multi)
    } else {
        const int KBF = 32 * FT, total = (int)c8_wt_total(K, C, MODE >= 2 ? 1 : 0);
        float* wt;
        int rc = tn_scratch_get(ctx, (size_t)total * sizeof(_Float16), &wt);
        if (rc) return rc;
        c8_wt_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(W, reinterpret_cast<_Float16*>(wt), K, C, KBF, g.nchunk,
                                                               total, MODE >= 2 ? 1 : 0, tk ? 1 : 0);
        TN_LAUNCH_CHECK();
        g.wt = reinterpret_cast<const _Float16*>(wt);
    }
    if constexpr (MODE < 2) {
        if (tk) return FT == 2 ? c8_launch<2, MODE, true>(ctx, g) : c8_launch<1, MODE, true>(ctx, g);
    }
    return FT == 2 ? c8_launch<2, MODE>(ctx, g) : c8_launch<1, MODE>(ctx, g);
}

// every conv layer's arranged weights of a step in ONE launch (a net made eleven 5 us launches of c8_wt_kernel per step)
struct C8WtBatch {
    struct { const float* W; _Float16* wt; int K, C, KBF, nchunk, total, dgrad, tk; } s[32];
};
__global__ __launch_bounds__(256) void c8_wt_multi_kernel(C8WtBatch b) {
    const auto& q = b.s[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= q.total) return;
    q.wt[idx] = (_Float16)c8_wt_value(q.W, q.K, q.C, q.KBF, q.nchunk, idx, q.dgrad, q.tk);
}

// =================================================================================================
// Weight gradient of a 3x3 'same' convolution on c8 tensors:
//   dW[k][c][2-u][2-v] = (1/gs) * sum_{n,i,j} dz16[n,k,i,j] * x16[n,c,i-1+u,j-1+v]        (dz16 = fp16(gs*dz))
// GEMM rows = filters, columns = input channels at a fixed tap, reduction = pixels, 16 per MFMA.  Both operands are
// stored [pixel][8 channels] and the matrix core wants [channel][8 consecutive pixels]: gfx950's transposing LDS read
// (ds_read_b64_tr_b16: the 16 lanes of a group point at 4 pixels x 16 channels and each receives one channel's 4
// pixels) does that on the way from LDS to the registers, so
//   * the LDS images are plain copies of the c8 tensors -- the x halo tile [octet plane][row][W+2 cells] and the dz tile
//     [octet plane][pixel] -- filled by LDS-DMA (global_load_lds, 16 bytes per lane, 1 KB per wave instruction) with
//     per-lane source addresses: halo columns, rows outside the image and octets beyond the tensor read a zero cell;
//   * a tap is an address offset of the B read: no funnel shifts, no register transposes, no staging registers at all;
//   * three stages of 128 pixels are in flight (two tiles ahead of the matrix work, counted vmcnt + raw s_barrier);
//   * POOL: the block's dz is not a tensor: the pooled gradient and the pooling mask travel to LDS raw and a pass
//     over LDS expands them (window bit ? g : 0) into the dz image.
// block = 32*NFT filters x 32*NCT channels x a slab of tiles; wave = one (filter tile, channel tile) pair with all nine
// taps (144 accumulators) and, when NFT*NCT < 4, one of PS interleaved step subsets.  The bias gradient rides as a
// tenth product against a vector of ones in the waves of channel tile 0.
// =================================================================================================
__device__ uint4 c8_zero_cell_g = {0u, 0u, 0u, 0u};

typedef short c8_short4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
// LDS-DMA of 16 bytes per lane: lane i's bytes from its own source pointer to LDS byte address lds_dst + 16 i (lds_dst
// wave-uniform, travels in M0).  Inline asm on purpose: hipcc orders every LDS read behind ALL outstanding
// __builtin_amdgcn_global_load_lds (s_waitcnt vmcnt(0) in front of each group of ds_reads -- it cannot prove that the
// DMA's stage and the stage being read are different), which serialises the whole pipeline; an asm statement is not in
// its bookkeeping, the kernels count vmcnt themselves.  M0 is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void c8_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ half4v c8_tr16(const char* l) {
    return __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c8_short4*)l));
}

struct C8WG {
    const uint4* x;        // c8 cells (N, C8, H, W)
    const uint4* dz;       // c8 cells (N, K8, H, W); POOL: the pooled gradient (N, K8, H/2, W/2)
    const uint2* mask;     // POOL: 8 mask bytes per pooled cell
    float* ws;             // [S * PS][K*C*9] partial weight gradients, dW layout
    float* dbws;           // [S * PS][K] partial bias gradients
    int N, C, C8, H, Wd, K, K8;
    int KG, CG, S, tpb, NTILES;            // filter groups, channel groups, slabs, tiles per slab
    int NI, TH, THi, RT, RS, lgW, lgP;     // tile = NI images x TH rows (128 pixels)
    int XC, XCH, XPS, DPS, SB;             // x plane: content cells, 1 KB chunks, stride; dz plane stride; stage bytes
    int offD, offG, offM, offDump;         // dz planes / raw pooled gradient / raw mask / the fillers' dump KB inside a stage
    int nQx, nQd, NQ;                      // LDS-DMA chunks per stage: x, dz (POOL: pooled gradient), all
    int nstage;
    int roll, XA;              // ROLL: the x ring (bytes in front of the stages); stages then hold dz (+ raw pooled gradient, mask) only
    float oscale;
    unsigned long long* dbg;   // TN_C8_DBG: per block {life, DMA wait, barrier, matrix steps} cycles
    int exp;                   // TN_C8_EXP (experiments): bit 0 no refills inside the loop, bit 1 no matrix steps
};

// ROLL (round 4; tiles that are bands of TH rows of ONE image, rows of >= 16 pixels): the x rows live in a RING of four
// regions of TH rows per octet plane instead of a halo tile per stage.  A tile fetches only its own TH rows (2 DMA chunks
// per plane, no fillers: its 128 cells are contiguous in HBM and in LDS) -- the row above it is the last row of the
// previous tile's region, the row below it the first row of the next tile's (which therefore lands one tile earlier than
// the tile's dz), rows outside the image are a zero row behind the ring.  No halo columns either: the cell left of
// column 0 / right of column W-1 is whatever the ring holds there, and the one k element of the transposed operand
// that came from it is cleared with a v_and.  At 64-pixel rows a halo tile is 4 x 66 cells for 2 x 64 of content: 5 chunks
// per plane became 2 (32-pixel rows: 4 -> 2, 16-pixel rows: 3 -> 2); the kernel was bound by the LDS-DMA stream
// (DMA alone 58 k of a block's 87 k cycles on conv2 of wide6).
template <int NFT, int NCT, bool POOL, int NGX, int TM = 1, bool ROLL = false>
__global__ __launch_bounds__(512) void c8_wgrad_kernel(C8WG g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    static_assert(!ROLL || (NCT != 0 && TM == 1), "ROLL: channel-tiled layers, 128-pixel tiles");
    constexpr int RGB = 2048, ZRO = 4 * RGB;          // ROLL: bytes of a ring region per plane; offset of the zero row
    // eight waves = two per SIMD: while one waits for its LDS operands or sits in the issue of an LDS-DMA (~100 cycles
    // each, nothing else of that wave moves meanwhile) the other feeds the matrix pipe.  One wave per SIMD ran this loop
    // at DMA issue + address arithmetic + matrix time, the sum (cycle stamps: 1.4 k + 1.4 k + 3.7 k per tile).
    // NCT == 0 ("tap-packed"): a first layer, C <= 8 = ONE octet plane.  The 32 columns of an MFMA tile are then 4 taps x
    // 8 channels instead of 32 channels of one tap (3 of which would be real): three products per step instead of nine,
    // one x plane in LDS instead of four, and the transposing read picks the tap where it picked the plane.
    constexpr bool TAPK = NCT == 0;
    constexpr int NCTe = TAPK ? 1 : NCT, NACC = TAPK ? 3 : 9;
    constexpr int KP = 4 * NFT, CP = TAPK ? 1 : 4 * NCT, KBF = 32 * NFT, CBF = 32 * NCTe, PS = 8 / (NFT * NCTe), SPW = 8 / PS;
    // TM: a tile is 128 * TM pixels (a tile costs ~1.5-2 k cycles of barriers, DMA issue and set-up whatever it holds:
    // layers whose stage is small take four times the pixels per tile), i.e. SPW * TM steps per wave
    constexpr int NSTEP = SPW * TM, PCP = 32 * TM;            // steps per wave and tile; pooled cells per plane and tile
    char* const smem = reinterpret_cast<char*>(ct_smem);
    const int bid = blockIdx.x, per = g.KG * g.CG;
    const int z = ((bid >> 3) / per) * 8 + (bid & 7), rem = (bid >> 3) % per;
    if (z >= g.S) return;
    const int kg = rem / g.CG, cg = rem - kg * g.CG;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, hi = lane >> 5;
    const int ft = wave % NFT, ct = (wave / NFT) % NCTe, ps = wave / (NFT * NCTe);
    const int tile_beg = z * g.tpb, tile_end = min(g.NTILES, tile_beg + g.tpb);
    const int HW = g.H * g.Wd, Wp = g.Wd >> 1, THm = g.TH - 1, Wm = g.Wd - 1;
    const char* const zero_src = reinterpret_cast<const char*>(&c8_zero_cell_g);

    // ---- this wave's LDS-DMA chunks of a stage, by kind (compile-time counts: the waits are counted and the issue code
    // is branch-free): NGX x chunks q = 8 j + wave, then ONE more chunk per wave: a dz chunk (16 of them; POOL: 4 raw
    // pooled-gradient chunks of two planes and 2 mask chunks of four planes -- the dz image is expanded in LDS).  A
    // chunk index beyond its kind's count is a filler (zero cell -> the dump KB).  Per chunk (wave-uniform): destination
    // inside the stage; per lane: the source cell relative to the tile's first cell and flags (bit 0 lane in use,
    // 1 always zero, 2 / 3 top / bottom halo row, 4 mask bytes (8-byte cells), 8.. image of the tile)
    // POOL (round 4): the pooled gradient and the mask bytes of a tile travel through REGISTERS -- every thread loads
    // NPC pooled cells (16 + 8 bytes, asm loads counted by hand like the DMAs) while the tile in front of theirs is being
    // multiplied and expands them into the next stage's dz image (four 16-byte LDS stores) after the last step.  Before,
    // they went to LDS raw (6 DMA chunks) and a pass over LDS behind a second barrier expanded them: 12 k of a block's 97 k
    // cycles on conv2 of wide6.
    constexpr int NGD = POOL ? 0 : (KP * 2 * TM + 7) / 8, NG = NGX + NGD;
    constexpr int NPC = POOL ? (KP * PCP + 511) / 512 : 1;
    int pg_rel[NPC], pg_fl[NPC], pg_dst[NPC];
    if (POOL) {
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            const int idx = t + 512 * c, live = idx < KP * PCP, ii = live ? idx : 0;
            const int plane = ii / PCP, pc = ii % PCP;
            const int ni = pc >> (g.lgP - 2), prow = (pc >> (g.lgW - 1)) & ((g.TH >> 1) - 1), pcol = pc & (Wp - 1);
            pg_rel[c] = (ni * g.K8 + plane) * (HW >> 2) + prow * Wp + pcol;
            pg_fl[c] = (live ? 1 : 0) | (kg * KP + plane >= g.K8 ? 2 : 0) | (ni << 8);
            pg_dst[c] = g.offD + plane * g.DPS + ((ni << g.lgP) + ((2 * prow) << g.lgW) + 2 * pcol) * 16;
        }
    }
    int4v pgv[NPC];                 // (plain vector types: the asm statements below take them as register operands)
    int2v pmv[NPC];
    // the pooled cells of tile `tile` (clamped into the tensor) -> registers; untracked by the compiler: waited for by hand
    auto pool_load = [&](int tile) __attribute__((always_inline)) {
        const int tc = min(max(tile, 0), g.NTILES - 1), gi_ = tc / g.RT, n0 = gi_ * g.NI, r0 = (tc - gi_ * g.RT) * g.TH;
        const long long db = ((long long)(n0 * g.K8 + kg * KP) * (g.H >> 1) + (r0 >> 1)) * Wp;
        const char* const gp = reinterpret_cast<const char*>(g.dz + db);
        const char* const mp = reinterpret_cast<const char*>(g.mask + db);
        const int nlim = g.N - n0;
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            const bool zero = (pg_fl[c] & 3) != 1 || (pg_fl[c] >> 8) >= nlim;
            const char* sg = zero ? zero_src : gp + (long long)pg_rel[c] * 16;
            const char* sm = zero ? zero_src : mp + (long long)pg_rel[c] * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pgv[c]) : "v"(sg));
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pmv[c]) : "v"(sm));
        }
    };
    // ... expanded into the dz image of stage `stage` (a zero gradient cell expands to zeros whatever its mask says)
    auto pool_expand = [&](int stage) __attribute__((always_inline)) {
        char* const sbn = smem + (ROLL ? g.XA : 0) + stage * g.SB;
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            if (pg_fl[c] & 1) {
                char* const d0 = sbn + pg_dst[c];
                const uint4 gq = __builtin_bit_cast(uint4, pgv[c]);
                const uint2 mq = __builtin_bit_cast(uint2, pmv[c]);
                *reinterpret_cast<uint4*>(d0) = c8_pool_cell(gq, mq, 0);
                *reinterpret_cast<uint4*>(d0 + 16) = c8_pool_cell(gq, mq, 1);
                *reinterpret_cast<uint4*>(d0 + 16 * g.Wd) = c8_pool_cell(gq, mq, 2);
                *reinterpret_cast<uint4*>(d0 + 16 * g.Wd + 16) = c8_pool_cell(gq, mq, 3);
            }
        }
    };
    int gl_rel[NG], gl_fl[NG], gl_dst[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        int rel = 0, fl = 1 | 2, dst = g.offDump;              // filler by default
        if (ROLL && j < NGX) {
            const int q = 8 * j + wave;
            if (q < g.nQx) {
                // (a plane beyond the tensor re-reads its last plane: it only feeds columns c >= C, which are never stored --
                // the ring form's DMAs carry no "zero" select at all, ~25 instructions per chunk became ~8)
                const int plane = q >> 1, sub = q & 1, pc = min(cg * CP + plane, g.C8 - 1) - cg * CP;
                rel = pc * HW + sub * 64 + lane;
                fl = 1;
                dst = plane * g.XPS + 16 + sub * 1024;
            }
        } else if (j < NGX) {
            const int q = 8 * j + wave;
            if (q < g.nQx) {
                const int plane = q / g.XCH, c = (q - plane * g.XCH) * 64 + lane;
                const int cc = min(c, g.XC - 1), colp = cc % g.RS, rr = cc / g.RS, rowh = rr % g.THi, ni = rr / g.THi;
                const bool zero = colp == 0 || colp == g.RS - 1 || cg * CP + plane >= g.C8;
                rel = (ni * g.C8 + plane) * HW + (rowh - 1) * g.Wd + (colp - 1);
                fl = (c < g.XC ? 1 : 0) | (zero ? 2 : 0) | (rowh == 0 ? 4 : 0) | (rowh == g.THi - 1 ? 8 : 0) | (ni << 8);
                dst = plane * g.XPS + (q - plane * g.XCH) * 1024;
            }
        } else {
            const int qq = 8 * (j - NGX) + wave;
            if (!POOL) {
                if (qq < 2 * KP * TM) {
                    const int plane = qq / (2 * TM), sub = qq % (2 * TM), pp = sub * 64 + lane;
                    const int ni = pp >> g.lgP, row = (pp >> g.lgW) & THm, col = pp & Wm;
                    // (ROLL: filter planes beyond the tensor likewise re-read the last one -- rows k >= K are never stored)
                    const int pk = ROLL ? min(kg * KP + plane, g.K8 - 1) - kg * KP : plane;
                    rel = (ni * g.K8 + pk) * HW + row * g.Wd + col;
                    fl = ROLL ? 1 : (1 | (kg * KP + plane >= g.K8 ? 2 : 0) | (ni << 8));
                    dst = g.offD + plane * g.DPS + sub * 1024;
                }
            }
        }
        gl_rel[j] = rel; gl_fl[j] = fl;
        gl_dst[j] = __builtin_amdgcn_readfirstlane(dst);
    }
    // tile being refilled: set by tile_setup(), consumed by the issue_range() calls spread over the matrix steps
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ct_smem;     // LDS byte address of the image
    const char* cur_xp = reinterpret_cast<const char*>(g.x);
    const char* cur_dp = reinterpret_cast<const char*>(g.dz);
    const char* cur_mp = reinterpret_cast<const char*>(g.mask);
    int cur_zmask = 0, cur_nlim = 0;
    unsigned cur_sb = lds0, cur_xr = 0;
    int rf_gi = tile_beg / g.RT, rf_rt = tile_beg - rf_gi * g.RT, rf_tile = tile_beg;      // refill cursor
    // ROLL: tile number `tile` (clamped into the tensor) -> x region, dz stage
    auto roll_setup = [&](int tile, int stage, int region) __attribute__((always_inline)) {
        const int tc = min(max(tile, 0), g.NTILES - 1), n0 = tc / g.RT, r0 = (tc - n0 * g.RT) * g.TH;
        cur_zmask = 2;
        cur_nlim = g.N - n0;
        cur_xp = reinterpret_cast<const char*>(g.x + ((long long)(n0 * g.C8 + cg * CP) * g.H + r0) * g.Wd);
        const long long db = POOL ? ((long long)(n0 * g.K8 + kg * KP) * (g.H >> 1) + (r0 >> 1)) * Wp
                                  : ((long long)(n0 * g.K8 + kg * KP) * g.H + r0) * g.Wd;
        cur_dp = reinterpret_cast<const char*>(g.dz + db);
        cur_mp = reinterpret_cast<const char*>(g.mask + db);
        cur_sb = lds0 + g.XA + stage * g.SB;
        cur_xr = lds0 + region * RGB;
    };
    auto tile_setup = [&](int stage) __attribute__((always_inline)) {
        // (tiles beyond the slab's end re-read its last tile: the number of DMAs per stage stays constant)
        const int n0 = rf_gi * g.NI, r0 = rf_rt * g.TH;
        cur_zmask = 2 | (r0 == 0 ? 4 : 0) | (r0 + g.TH >= g.H ? 8 : 0);
        cur_nlim = g.N - n0;                       // images n0 + ni with ni >= cur_nlim do not exist
        cur_xp = reinterpret_cast<const char*>(g.x + ((long long)(n0 * g.C8 + cg * CP) * g.H + r0) * g.Wd);
        const long long db = POOL ? ((long long)(n0 * g.K8 + kg * KP) * (g.H >> 1) + (r0 >> 1)) * Wp
                                  : ((long long)(n0 * g.K8 + kg * KP) * g.H + r0) * g.Wd;
        cur_dp = reinterpret_cast<const char*>(g.dz + db);
        cur_mp = reinterpret_cast<const char*>(g.mask + db);
        cur_sb = lds0 + stage * g.SB;
        if (rf_tile + 1 < tile_end) {
            ++rf_tile;
            if (++rf_rt == g.RT) { rf_rt = 0; ++rf_gi; }
        }
    };
    auto issue_range = [&](auto J0c, auto J1c) __attribute__((always_inline)) {
        constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            if (ROLL) {              // every chunk real, every source inside its tensor (clamped planes, clamped tiles)
                const char* src = (j < NGX ? cur_xp : cur_dp) + (long long)gl_rel[j] * 16;
                c8_glds16(src, __builtin_amdgcn_readfirstlane((j < NGX ? cur_xr : cur_sb) + gl_dst[j]));
                continue;
            }
            const int fl = gl_fl[j];
            const bool zero = (fl & cur_zmask) || (fl >> 8) >= cur_nlim;
            const char* src = j < NGX ? cur_xp + (long long)gl_rel[j] * 16
                            : (POOL && (fl & 16)) ? cur_mp + (long long)gl_rel[j] * 8 : cur_dp + (long long)gl_rel[j] * 16;
            src = zero ? zero_src : src;
            if ((fl & 1) && !(g.exp & 1))
                c8_glds16(src, __builtin_amdgcn_readfirstlane(((ROLL && j < NGX) ? cur_xr : cur_sb) + gl_dst[j]));
        }
    };
    using J_0 = std::integral_constant<int, 0>;
    using J_N = std::integral_constant<int, NG>;

    f32x16 acc[NACC], accb;
#pragma unroll
    for (int a_ = 0; a_ < NACC; ++a_)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a_][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bool want_db = cg == 0 && ct == 0;
    const half8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f,
                        (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};

    // ---- operand addresses of this lane: group of 16 lanes = 4 pixels x 16 channels; lane (r4, q8) supplies pixel r4,
    // channels 4 q8 .. + 3 of the group's pair of octet planes and receives channel (lane & 15)'s four pixels
    const int grp = lane >> 4, r4 = (lane >> 2) & 3, q8 = lane & 3;
    const int a_off = g.offD + (ft * 4 + 2 * (grp & 1) + (q8 >> 1)) * g.DPS + (q8 & 1) * 8;
    const int b_off = TAPK ? (q8 & 1) * 8 : (ct * 4 + 2 * (grp & 1) + (q8 >> 1)) * g.XPS + (q8 & 1) * 8;
    const int RS16 = g.RS * 16;
    int toff[3];               // TAPK: column tile jt, this lane's 8-channel slot = tap 4 jt + 2 (grp & 1) + (q8 >> 1)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
        const int tap = 4 * jt + 2 * (grp & 1) + (q8 >> 1);
        toff[jt] = tap < 9 ? (tap / 3) * RS16 + (tap % 3) * 16 : 0;      // taps 9..11: columns that are never stored
    }

    unsigned long long d_wait = 0, d_bar = 0, d_mm = 0, d_t0 = 0, d_w0 = 0, d_exp = 0;     // (d_exp: shown as "prologue" by tools/dbg_c8.py)
    if (g.dbg) { d_t0 = __builtin_readcyclecounter(); d_w0 = wall_clock64(); }
    int c_rt = tile_beg % g.RT;                     // ROLL: row band of the tile being multiplied
    // ROLL: the band row of step i's 16 pixels does not change from tile to tile: (row - 1) * row bytes and whether the tap
    // row above / below leaves the band, as scalars (computed per step and tap row they were ~36 instructions per step)
    int st_ro[NSTEP];
    bool st_top[NSTEP], st_bot[NSTEP];
    if (ROLL) {
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            const int rl = __builtin_amdgcn_readfirstlane(((16 * (ps + PS * i)) >> g.lgW) & THm);
            st_ro[i] = (rl - 1) * (g.Wd * 16);
            st_top[i] = rl == 0;
            st_bot[i] = rl == THm;
        }
    }
    if ((g.exp & 4) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if ((g.exp & 8) && wave < 4) __builtin_amdgcn_s_setprio(1);
    if (ROLL) {
        // the zero row of every plane; then: the tile above the slab's first (its last row is that tile's upper halo; unused
        // when the slab starts at the top of an image) into region 3, tiles 0 and 1 of the slab into regions / stages 0, 1
        for (int i = t; i < CP * g.Wd; i += 512)
            *reinterpret_cast<uint4*>(smem + (i >> g.lgW) * g.XPS + 16 + ZRO + (i & Wm) * 16) = make_uint4(0u, 0u, 0u, 0u);
        roll_setup(tile_beg - 1, 0, 3);
        issue_range(J_0{}, std::integral_constant<int, NGX>{});
        roll_setup(tile_beg, 0, 0);
        issue_range(J_0{}, J_N{});
        roll_setup(tile_beg + 1, 1, 1);
        issue_range(J_0{}, J_N{});
    } else {
        tile_setup(0);
        issue_range(J_0{}, J_N{});
        if (g.nstage > 2) {
            tile_setup(1);
            issue_range(J_0{}, J_N{});
        }
    }
    if (POOL) {                                     // the first tile's dz image (the loop's first barrier publishes it)
        pool_load(tile_beg);
#pragma unroll
        for (int c = 0; c < NPC; ++c) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pgv[c]), "+v"(pmv[c]));
        pool_expand(0);
    }
    for (int tile = tile_beg, it = 0; tile < tile_end; ++tile, ++it) {
        const int stage = it % g.nstage;
        unsigned long long s0 = 0, s1 = 0, s2 = 0;
        if (g.dbg) s0 = __builtin_readcyclecounter();
        // this wave's DMAs of the stage have landed (the following stage's may still be in flight) ...
        // (ROLL: the x rows of the NEXT tile too -- a tile's x chunks are issued in front of its dz chunks)
        if (ROLL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG - NGX) : "memory");
        else if (g.nstage > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (g.dbg) s1 = __builtin_readcyclecounter();
        // ... and everybody's; all waves are also done with the stage that is refilled next
        if (POOL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (this wave's stores into the tile's dz image)
        __builtin_amdgcn_s_barrier();
        if (g.dbg) { s2 = __builtin_readcyclecounter(); d_wait += s1 - s0; d_bar += s2 - s1; }
        char* const sb = smem + (ROLL ? g.XA : 0) + stage * g.SB;
        // the refill of the stage two tiles ahead is spread over this tile's matrix steps
        // ROLL: rows of this tile's band at reg_off, the row above / below it at top_off / bot_off (ring neighbours or zeros)
        int reg_off = 0, top_off = 0, bot_off = 0;
        if (ROLL) {
            roll_setup(tile + 2, (it + 2) % 3, (it + 2) & 3);
            const int region = it & 3;
            reg_off = region * RGB;
            top_off = c_rt == 0 ? ZRO : ((region + 3) & 3) * RGB + THm * (g.Wd * 16);
            bot_off = c_rt == g.RT - 1 ? ZRO : ((region + 1) & 3) * RGB;
            if (++c_rt == g.RT) c_rt = 0;
        } else {
            tile_setup((it + g.nstage - 1) % g.nstage);
        }
        if (POOL) pool_load(tile + 1);               // lands under this tile's steps, expanded behind them
        const char* const ab = sb + a_off;
        const char* const bb = sb + b_off;
        auto step = [&](auto Ic) __attribute__((always_inline)) {
            constexpr int i = decltype(Ic)::value;
            constexpr int J0 = i * NG / NSTEP, J1 = (i + 1) * NG / NSTEP;
            if (g.exp & 2) {
                issue_range(std::integral_constant<int, J0>{}, std::integral_constant<int, J1>{});
                return;
            }
            const int p = 16 * (ps + PS * i) + 8 * (grp >> 1) + r4;
            const char* ap = ab + p * 16;
            const half4v a0 = c8_tr16(ap), a1 = c8_tr16(ap + 64);
            const char* xp = bb + (((p >> g.lgP) * g.THi + ((p >> g.lgW) & THm)) * g.RS + (p & Wm)) * 16;
            half4v bv[NACC][2];
            if (ROLL) {
                const int g16 = 16 * (ps + PS * i), W16 = g.Wd * 16;     // wave-uniform
                const char* const xc = smem + b_off + (p & Wm) * 16;
                const int rbase = reg_off + st_ro[i];                    // (row - 1) of the band, in the ring
                // the k element whose cell lies left of column 0 (tap column 0, first pixel of the lower 8) / right of column
                // W - 1 (tap column 2, last pixel of the upper 8)
                const unsigned mL = ((g16 & Wm) == 0 && (grp >> 1) == 0) ? 0xffff0000u : 0xffffffffu;
                const unsigned mR = (((g16 + 16) & Wm) == 0 && (grp >> 1) == 1) ? 0x0000ffffu : 0xffffffffu;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int ro = u == 0 ? (st_top[i] ? top_off : rbase) : u == 1 ? rbase + W16 : (st_bot[i] ? bot_off : rbase + 2 * W16);
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        bv[(u * 3 + v) % NACC][0] = c8_tr16(xc + ro + v * 16);
                        bv[(u * 3 + v) % NACC][1] = c8_tr16(xc + ro + v * 16 + 64);
                    }
                    uint2 e0 = __builtin_bit_cast(uint2, bv[(u * 3) % NACC][0]);
                    e0.x &= mL;
                    bv[(u * 3) % NACC][0] = __builtin_bit_cast(half4v, e0);
                    uint2 e2 = __builtin_bit_cast(uint2, bv[(u * 3 + 2) % NACC][1]);
                    e2.y &= mR;
                    bv[(u * 3 + 2) % NACC][1] = __builtin_bit_cast(half4v, e2);
                }
            } else if (TAPK) {
#pragma unroll
                for (int jt = 0; jt < 3; ++jt) {
                    bv[jt][0] = c8_tr16(xp + toff[jt]);
                    bv[jt][1] = c8_tr16(xp + toff[jt] + 64);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        bv[(u * 3 + v) % NACC][0] = c8_tr16(xp + u * RS16 + v * 16);
                        bv[(u * 3 + v) % NACC][1] = c8_tr16(xp + u * RS16 + v * 16 + 64);
                    }
            }
            const half8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
            for (int tp = 0; tp < NACC; ++tp) {
                const half8 b = {bv[tp][0][0], bv[tp][0][1], bv[tp][0][2], bv[tp][0][3],
                                 bv[tp][1][0], bv[tp][1][1], bv[tp][1][2], bv[tp][1][3]};
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[tp], 0, 0, 0);
            }
            // the bias product of a step is taken by ONE of the channel-tile waves that share its pixels, alternating by
            // step: with it always on channel tile 0 those waves ran 10 MFMAs per step against 9 and the others waited
            // for them at every tile's barrier
            if (cg == 0 && (NCTe == 1 || (i & 1) == ct)) accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ones, accb, 0, 0, 0);
            issue_range(std::integral_constant<int, J0>{}, std::integral_constant<int, J1>{});
        };
#define C8W_ST(I) if (NSTEP > I) step(std::integral_constant<int, (I) % NSTEP>{});
        C8W_ST(0) C8W_ST(1) C8W_ST(2) C8W_ST(3) C8W_ST(4) C8W_ST(5) C8W_ST(6) C8W_ST(7)
        C8W_ST(8) C8W_ST(9) C8W_ST(10) C8W_ST(11) C8W_ST(12) C8W_ST(13) C8W_ST(14) C8W_ST(15)
        C8W_ST(16) C8W_ST(17) C8W_ST(18) C8W_ST(19) C8W_ST(20) C8W_ST(21) C8W_ST(22) C8W_ST(23)
        C8W_ST(24) C8W_ST(25) C8W_ST(26) C8W_ST(27) C8W_ST(28) C8W_ST(29) C8W_ST(30) C8W_ST(31)
#undef C8W_ST
        unsigned long long s3 = 0;
        if (g.dbg) { s3 = __builtin_readcyclecounter(); d_mm += s3 - s2; }
        if (POOL) {
            // the NGX x chunks of this tile's refill were issued behind the loads: they may still be in flight
#pragma unroll
            for (int c = 0; c < NPC; ++c) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(pgv[c]), "+v"(pmv[c]) : "n"(NGX));
            pool_expand((it + 1) % g.nstage);
            if (g.dbg) d_exp += __builtin_readcyclecounter() - s3;
        }
    }
    if (g.dbg && t == 0) {
        unsigned long long* d = g.dbg + 8 * (size_t)bid;
        d[0] = d_t0; d[1] = d_t0 + d_exp; d[2] = __builtin_readcyclecounter(); d[3] = d_wait; d[6] = d_bar; d[7] = d_mm;
        d[4] = d_w0; d[5] = wall_clock64();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing refills (clamped tiles) land before LDS is reused

    // ---- the PS step subsets of a (filter tile, channel tile) pair are added up through LDS, upper half onto lower
    // half, in a fixed order; subset 0 then holds the block's sums.  At most four waves write in a round:
    // [slot][reg][lane] floats, 36 KB per slot (the bias products take a second, small pass over the same memory)
    constexpr int NPR = NFT * NCTe;
    const int pr = wave % NPR;
    if (NCTe == 2) {            // the two halves of the bias products meet first: channel tile 1 onto channel tile 0
        float* const bslot = ct_smem + (size_t)((ps * NFT + ft) * 16) * 64 + lane;
        __syncthreads();
        if (ct == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bslot[r * 64] = accb[r];
        }
        __syncthreads();
        if (ct == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accb[r] += bslot[r * 64];
        }
    }
#pragma unroll
    for (int h = PS / 2; h >= 1; h >>= 1) {
        const bool writer = ps >= h && ps < 2 * h, reader = ps < h;
        float* const slot = ct_smem + (size_t)(((writer ? ps - h : ps) * NPR + pr) * (NACC * 16)) * 64 + lane;
        __syncthreads();
        if (writer) {
#pragma unroll
            for (int a_ = 0; a_ < NACC; ++a_)
#pragma unroll
                for (int r = 0; r < 16; ++r) slot[(a_ * 16 + r) * 64] = acc[a_][r];
        }
        __syncthreads();
        if (reader) {
#pragma unroll
            for (int a_ = 0; a_ < NACC; ++a_)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a_][r] += slot[(a_ * 16 + r) * 64];
        }
        __syncthreads();
        if (writer) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slot[r * 64] = accb[r];
        }
        __syncthreads();
        if (reader) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accb[r] += slot[r * 64];
        }
    }
    // bias gradient partial of the slab: column 0 of the product against ones
    const float os = g.oscale;
    if (ps == 0 && want_db && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kg * KBF + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) g.dbws[(size_t)z * g.K + k] = accb[r] * os;
        }
    }
    // slab z: dW layout, tap (u,v) of the correlation is element (2-u, 2-v)
    if (TAPK) {                                 // column l31 = tap 4 jt + (l31 >> 3), channel l31 & 7
        if (ps != 0) return;
        const int e = l31 & 7;
        if (e < g.C) {
            float* wz = g.ws + (size_t)z * g.K * g.C * 9;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kg * KBF + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (k < g.K) {
#pragma unroll
                    for (int jt = 0; jt < 3; ++jt) {
                        const int tap = 4 * jt + (l31 >> 3);
                        if (tap < 9) wz[((size_t)k * g.C + e) * 9 + 8 - tap] = acc[jt % NACC][r] * os;
                    }
                }
            }
        }
        return;
    }
    if (g.K % KBF == 0 && g.C % CBF == 0) {
        // whole tiles: the block's KBF x (CBF x 9) piece of the slab is KBF runs of CBF * 9 floats.  Straight from the
        // accumulators a store instruction wrote 4 bytes per lane 36 bytes apart (144 of them per lane, 37.7 MB per
        // launch as 4-byte pieces: ~15 us of a 50-65 us kernel); through LDS ([k][c * 9 + tap], conflict-free: 9 is odd)
        // every thread stores 16 bytes next to its neighbour's.
        constexpr int ROWF = CBF * 9;
        __syncthreads();                        // the reduction's slots are free
        if (ps == 0) {
            float* const T = ct_smem + (size_t)(ft * 32 + 4 * hi) * ROWF + (ct * 32 + l31) * 9;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int a_ = 0; a_ < NACC; ++a_) T[((r & 3) + 8 * (r >> 2)) * ROWF + 8 - a_] = acc[a_][r] * os;
        }
        __syncthreads();
        float* const wz = g.ws + (size_t)z * g.K * g.C * 9 + ((size_t)kg * KBF * g.C + (size_t)cg * CBF) * 9;
        constexpr int Q = ROWF / 4;             // 16-byte pieces per run
        for (int idx = t; idx < KBF * Q; idx += 512) {
            const int row = idx / Q, q = idx - row * Q;
            *reinterpret_cast<float4*>(wz + (size_t)row * g.C * 9 + 4 * q) =
                *reinterpret_cast<const float4*>(ct_smem + (size_t)row * ROWF + 4 * q);
        }
        return;
    }
    if (ps != 0) return;
    const int c = cg * CBF + ct * 32 + l31;
    if (c < g.C) {
        float* wz = g.ws + (size_t)z * g.K * g.C * 9;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kg * KBF + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) {
#pragma unroll
                for (int a_ = 0; a_ < NACC; ++a_) wz[((size_t)k * g.C + c) * 9 + 8 - a_] = acc[a_][r] * os;
            }
        }
    }
}

// =================================================================================================
// Round 6: the same weight gradient with a block's work split BY ROLE over sixteen waves (c8_wgrad_tr_kernel), for the
// 64-filter x 64-channel block tile (K, C >= 64: every layer of wide6 from conv2 on, conv3 of cifar_like).
// Stamps of the eight-wave kernel above (wide6 conv2, profiles/r06_wgrad_stamps.txt): a wave's 16-pixel step is a chain
// of nine LDS round trips (two transposing reads - s_waitcnt - product) and takes ~930 cycles for 320 cycles of matrix
// work; two such waves per SIMD = 69 % of the pipe inside the steps; every wave also issues four LDS-DMAs per tile
// (~100 cycles of its issue each), and 15 % of a block's life are barrier waits.  Here
//   * TWELVE compute waves = (filter tile, channel tile) x TAP ROW: 48 accumulators instead of 144, <= 128 registers,
//     three compute waves per SIMD -- three independent chains of round trips feed each matrix pipe instead of two;
//     a wave takes all eight steps of a 128-pixel tile, so there are no step subsets to add up at the end;
//   * FOUR loader waves (one per SIMD) own every LDS-DMA issue, the ring / tile bookkeeping, the counted vmcnt waits
//     and the pooled-gradient expansion: a compute wave's loop holds ds_read_b64_tr_b16, v_mfma and one s_barrier per
//     tile, nothing else;
//   * same LDS images, same DMA chunk lists, same slab layout as the eight-wave kernel (c8w_geometry serves both).
// The bias product of a step is taken by ONE of the six waves that share its dz operand, by step number.
// =================================================================================================
// ablation builds of c8_wgrad_tr_kernel (wrong results, timing only; hipcc -DC8W_ABL=n into a side library, run through
// TN_HIP_LIB; tools/abl_wgrad.sh): 1 no products, 2 no operand reads (the products run on stale registers), 4 no LDS-DMA
// (the loaders only keep the barriers), 8 no edge masks
#ifndef C8W_ABL
#define C8W_ABL 0
#endif
template <int NCT, int NGX, bool POOL, bool ROLL>
__global__ __launch_bounds__(1024) void c8_wgrad_tr_kernel(C8WG g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr int RGB = 2048, ZRO = 4 * RGB;
    // NCT = 2: 64 channels, wave = (filter tile, channel tile, tap row), all eight steps of a tile.  NCT = 1: 32 channels
    // (conv2 of cifar_like), wave = (filter tile, tap row, step subset): PS = 2 interleaved subsets of four steps, added up
    // through LDS at the end
    constexpr int KP = 8, CP = 4 * NCT, KBF = 64, CBF = 32 * NCT, NCW = 12, NLW = 4, PS = 2 / NCT, NSTEP = 8 / PS;
    constexpr int NGD = POOL ? 0 : 2 * KP / NLW, NG = NGX + NGD;
    char* const smem = reinterpret_cast<char*>(ct_smem);
    const int bid = blockIdx.x, per = g.KG * g.CG;
    const int z = ((bid >> 3) / per) * 8 + (bid & 7), rem = (bid >> 3) % per;
    if (z >= g.S) return;
    const int kg = rem / g.CG, cg = rem - kg * g.CG;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, hi = lane >> 5;
    const bool loader = wave >= NCW;
    const int tile_beg = z * g.tpb, tile_end = min(g.NTILES, tile_beg + g.tpb);
    const int HW = g.H * g.Wd, Wp = g.Wd >> 1, THm = g.TH - 1, Wm = g.Wd - 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ct_smem;
    unsigned long long d_bar = 0, d_mm = 0, d_t0 = 0, d_w0 = 0;
    if (g.dbg) { d_t0 = __builtin_readcyclecounter(); d_w0 = wall_clock64(); }

    if (ROLL) {          // the zero row of every plane (read as the halo of an image's first / last band)
        for (int i = t; i < CP * g.Wd; i += 1024)
            *reinterpret_cast<uint4*>(smem + (i >> g.lgW) * g.XPS + 16 + ZRO + (i & Wm) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }

    f32x16 acc[3], accb;
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a_][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    // compute roles (waves 0..11); xr = the wave's second coordinate: channel tile (NCT = 2) / step subset (NCT = 1)
    const int ft = wave & 1, xr = NCT == 2 ? (wave >> 1) & 1 : wave / 6, u = NCT == 2 ? wave >> 2 : (wave >> 1) % 3;
    const int ct = NCT == 2 ? xr : 0;

    if (loader) {
        // ================================ loader waves: every DMA of the block ================================
        const int lw = wave - NCW, lt = t - 64 * NCW;
        const char* const zero_src = reinterpret_cast<const char*>(&c8_zero_cell_g);
        // POOL: one pooled cell (16 + 8 bytes) of the next tile per loader thread, expanded into four dz cells
        int pg_rel = 0, pg_fl = 0, pg_dst = 0;
        if (POOL) {
            const int plane = lt >> 5, pc = lt & 31;
            const int ni = pc >> (g.lgP - 2), prow = (pc >> (g.lgW - 1)) & ((g.TH >> 1) - 1), pcol = pc & (Wp - 1);
            pg_rel = (ni * g.K8 + plane) * (HW >> 2) + prow * Wp + pcol;
            pg_fl = 1 | (kg * KP + plane >= g.K8 ? 2 : 0) | (ni << 8);
            pg_dst = g.offD + plane * g.DPS + ((ni << g.lgP) + ((2 * prow) << g.lgW) + 2 * pcol) * 16;
        }
        int4v pgv;
        int2v pmv;
        auto pool_load = [&](int tile) __attribute__((always_inline)) {
            const int tc = min(max(tile, 0), g.NTILES - 1), gi_ = tc / g.RT, n0 = gi_ * g.NI, r0 = (tc - gi_ * g.RT) * g.TH;
            const long long db = ((long long)(n0 * g.K8 + kg * KP) * (g.H >> 1) + (r0 >> 1)) * Wp;
            const bool zero = (pg_fl & 3) != 1 || (pg_fl >> 8) >= g.N - n0;
            const char* sg = zero ? zero_src : reinterpret_cast<const char*>(g.dz + db) + (long long)pg_rel * 16;
            const char* sm = zero ? zero_src : reinterpret_cast<const char*>(g.mask + db) + (long long)pg_rel * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pgv) : "v"(sg));
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pmv) : "v"(sm));
        };
        auto pool_expand = [&](int stage) __attribute__((always_inline)) {
            char* const d0 = smem + (ROLL ? g.XA : 0) + stage * g.SB + pg_dst;
            const uint4 gq = __builtin_bit_cast(uint4, pgv);
            const uint2 mq = __builtin_bit_cast(uint2, pmv);
            *reinterpret_cast<uint4*>(d0) = c8_pool_cell(gq, mq, 0);
            *reinterpret_cast<uint4*>(d0 + 16) = c8_pool_cell(gq, mq, 1);
            *reinterpret_cast<uint4*>(d0 + 16 * g.Wd) = c8_pool_cell(gq, mq, 2);
            *reinterpret_cast<uint4*>(d0 + 16 * g.Wd + 16) = c8_pool_cell(gq, mq, 3);
        };
        // this wave's chunks of a stage (the eight-wave kernel's lists, dealt to four waves): NGX x chunks q = 4 j + lw,
        // then (not POOL) NGD dz chunks; beyond a kind's count: a filler (zero cell -> the dump KB)
        int gl_rel[NG], gl_fl[NG], gl_dst[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            int rel = 0, fl = 1 | 2, dst = g.offDump;
            if (ROLL && j < NGX) {
                const int q = NLW * j + lw;
                if (q < g.nQx) {
                    const int plane = q >> 1, sub = q & 1, pc = min(cg * CP + plane, g.C8 - 1) - cg * CP;
                    rel = pc * HW + sub * 64 + lane;
                    fl = 1;
                    dst = plane * g.XPS + 16 + sub * 1024;
                }
            } else if (j < NGX) {
                const int q = NLW * j + lw;
                if (q < g.nQx) {
                    const int plane = q / g.XCH, c = (q - plane * g.XCH) * 64 + lane;
                    const int cc = min(c, g.XC - 1), colp = cc % g.RS, rr = cc / g.RS, rowh = rr % g.THi, ni = rr / g.THi;
                    const bool zero = colp == 0 || colp == g.RS - 1 || cg * CP + plane >= g.C8;
                    rel = (ni * g.C8 + plane) * HW + (rowh - 1) * g.Wd + (colp - 1);
                    fl = (c < g.XC ? 1 : 0) | (zero ? 2 : 0) | (rowh == 0 ? 4 : 0) | (rowh == g.THi - 1 ? 8 : 0) | (ni << 8);
                    dst = plane * g.XPS + (q - plane * g.XCH) * 1024;
                }
            } else if (!POOL) {
                const int qq = NLW * (j - NGX) + lw;
                if (qq < 2 * KP) {
                    const int plane = qq >> 1, sub = qq & 1, pp = sub * 64 + lane;
                    const int ni = pp >> g.lgP, row = (pp >> g.lgW) & THm, col = pp & Wm;
                    const int pk = ROLL ? min(kg * KP + plane, g.K8 - 1) - kg * KP : plane;
                    rel = (ni * g.K8 + pk) * HW + row * g.Wd + col;
                    fl = ROLL ? 1 : (1 | (kg * KP + plane >= g.K8 ? 2 : 0) | (ni << 8));
                    dst = g.offD + plane * g.DPS + sub * 1024;
                }
            }
            gl_rel[j] = rel; gl_fl[j] = fl;
            gl_dst[j] = __builtin_amdgcn_readfirstlane(dst);
        }
        const char* cur_xp = reinterpret_cast<const char*>(g.x);
        const char* cur_dp = reinterpret_cast<const char*>(g.dz);
        int cur_zmask = 0, cur_nlim = 0;
        unsigned cur_sb = lds0, cur_xr = 0;
        int rf_gi = tile_beg / g.RT, rf_rt = tile_beg - rf_gi * g.RT, rf_tile = tile_beg;
        auto roll_setup = [&](int tile, int stage, int region) __attribute__((always_inline)) {
            const int tc = min(max(tile, 0), g.NTILES - 1), n0 = tc / g.RT, r0 = (tc - n0 * g.RT) * g.TH;
            cur_xp = reinterpret_cast<const char*>(g.x + ((long long)(n0 * g.C8 + cg * CP) * g.H + r0) * g.Wd);
            cur_dp = reinterpret_cast<const char*>(g.dz + ((long long)(n0 * g.K8 + kg * KP) * g.H + r0) * g.Wd);
            cur_sb = lds0 + g.XA + stage * g.SB;
            cur_xr = lds0 + region * RGB;
        };
        auto tile_setup = [&](int stage) __attribute__((always_inline)) {
            const int n0 = rf_gi * g.NI, r0 = rf_rt * g.TH;
            cur_zmask = 2 | (r0 == 0 ? 4 : 0) | (r0 + g.TH >= g.H ? 8 : 0);
            cur_nlim = g.N - n0;
            cur_xp = reinterpret_cast<const char*>(g.x + ((long long)(n0 * g.C8 + cg * CP) * g.H + r0) * g.Wd);
            cur_dp = reinterpret_cast<const char*>(g.dz + ((long long)(n0 * g.K8 + kg * KP) * g.H + r0) * g.Wd);
            cur_sb = lds0 + stage * g.SB;
            if (rf_tile + 1 < tile_end) {
                ++rf_tile;
                if (++rf_rt == g.RT) { rf_rt = 0; ++rf_gi; }
            }
        };
        auto issue_range = [&](auto J0c, auto J1c) __attribute__((always_inline)) {
            constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
#pragma unroll
            for (int j = J0; j < J1; ++j) {
                if (C8W_ABL & 4) continue;
                if (ROLL) {
                    const char* src = (j < NGX ? cur_xp : cur_dp) + (long long)gl_rel[j] * 16;
                    c8_glds16(src, __builtin_amdgcn_readfirstlane((j < NGX ? cur_xr : cur_sb) + gl_dst[j]));
                    continue;
                }
                const int fl = gl_fl[j];
                const bool zero = (fl & cur_zmask) || (fl >> 8) >= cur_nlim;
                const char* src = (j < NGX ? cur_xp : cur_dp) + (long long)gl_rel[j] * 16;
                src = zero ? zero_src : src;
                if (fl & 1) c8_glds16(src, __builtin_amdgcn_readfirstlane(cur_sb + gl_dst[j]));
            }
        };
        using J_0 = std::integral_constant<int, 0>;
        using J_X = std::integral_constant<int, NGX>;
        using J_N = std::integral_constant<int, NG>;
        if (ROLL) {
            roll_setup(tile_beg - 1, 0, 3);
            issue_range(J_0{}, J_X{});
            roll_setup(tile_beg, 0, 0);
            issue_range(J_0{}, J_N{});
            roll_setup(tile_beg + 1, 1, 1);
            issue_range(J_0{}, J_N{});
        } else {            // (three stages: the host only picks this kernel then)
            tile_setup(0);
            issue_range(J_0{}, J_N{});
            tile_setup(1);
            issue_range(J_0{}, J_N{});
        }
        if (POOL) {         // the dz images of the slab's first two tiles; the third tile's pooled cells stay in flight
            pool_load(tile_beg);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pgv), "+v"(pmv));
            pool_expand(0);
            pool_load(tile_beg + 1);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pgv), "+v"(pmv));
            pool_expand(1);
            pool_load(tile_beg + 2);
        }
        unsigned long long l_wait = 0, l_bar = 0, l_iss = 0;
        for (int tile = tile_beg, it = 0; tile < tile_end; ++tile, ++it) {
            unsigned long long q0 = 0, q1 = 0, q2 = 0;
            if (g.dbg) q0 = __builtin_readcyclecounter();
            // the DMAs of this tile's stage have landed (ROLL: and the x rows of the next tile, its lower halo) ...
            // (POOL: the two loads of pooled cells issued behind the last DMAs may stay in flight)
            if (ROLL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG - NGX + (POOL ? 2 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG + (POOL ? 2 : 0)) : "memory");
            if (POOL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the stores into the dz images)
            if (g.dbg) q1 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();       // ... everybody's; the compute waves are done with the stage refilled next
            if (g.dbg) q2 = __builtin_readcyclecounter();
            if (ROLL) roll_setup(tile + 2, (it + 2) % 3, (it + 2) & 3);
            else tile_setup((it + 2) % 3);
            if (POOL) {
                // the pooled cells of tile + 2 have been travelling for a whole tile: expanded into the stage two tiles
                // ahead (as far ahead as a DMA'd dz image), then the cells of tile + 3 set out
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(pgv), "+v"(pmv));
                pool_expand((it + 2) % 3);
            }
            issue_range(J_0{}, J_N{});
            if (POOL) pool_load(tile + 3);
            if (g.dbg) { l_wait += q1 - q0; l_bar += q2 - q1; l_iss += __builtin_readcyclecounter() - q2; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing refills land before LDS is reused
        if (g.dbg && lt == 0) {      // (first loader wave: DMA wait, its own barrier wait, issue + set-up time)
            unsigned long long* d = g.dbg + 8 * (size_t)bid;
            d[3] = l_wait; d[1] = l_iss; (void)l_bar;
        }
    } else {
        // ================================ compute waves: tap row u of a 32 x 32 tile ================================
        const bool want_b = cg == 0;
        const half8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f,
                            (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
        const int grp = lane >> 4, r4 = (lane >> 2) & 3, q8 = lane & 3;
        const int a_off = g.offD + (ft * 4 + 2 * (grp & 1) + (q8 >> 1)) * g.DPS + (q8 & 1) * 8;
        const int b_off = (ct * 4 + 2 * (grp & 1) + (q8 >> 1)) * g.XPS + (q8 & 1) * 8;
        const int RS16 = g.RS * 16, W16 = g.Wd * 16;
        const int l16 = 8 * (grp >> 1) + r4;                    // this lane's pixel inside a step's sixteen
        const char* const x_lane = smem + b_off + l16 * 16;     // ROLL: + (column of the step, row in the ring) = the cell of tap column 0
        const unsigned lmL = (grp >> 1) == 0 ? 0xffff0000u : 0xffffffffu, lmR = (grp >> 1) == 1 ? 0x0000ffffu : 0xffffffffu;
        int c_rt = tile_beg % g.RT;
        // ROLL: the band row of step i's 16 pixels, tile after tile the same: (row - 1) * row bytes; first / last band row
        int st_ro[NSTEP];
        bool st_top[NSTEP], st_bot[NSTEP];
        if (ROLL) {
#pragma unroll
            for (int i = 0; i < NSTEP; ++i) {
                const int rl = ((16 * (NCT == 2 ? i : xr + PS * i)) >> g.lgW) & THm;
                st_ro[i] = (rl - 1) * W16;
                st_top[i] = rl == 0;
                st_bot[i] = rl == THm;
            }
        }
        // The tile loop exists SIX times, once per (tap row, channel tile) of the wave: with the tap row a compile-time
        // constant the row of a step's operands is one scalar select (as a run-time value it compiled to four scalar
        // branches per step) and the bias product of step i -- taken by the wave whose (tap row, channel tile) is
        // (i % 3, (i / 3) & 1) -- is there or not at compile time: no branch inside a tile's steps at all.
        auto run = [&](auto Uc, auto Xc) __attribute__((always_inline)) {
            constexpr int U = decltype(Uc)::value, X = decltype(Xc)::value;        // X: channel tile (NCT = 2) / step subset (NCT = 1)
            for (int tile = tile_beg, it = 0; tile < tile_end; ++tile, ++it) {
                const int stage = it % 3;
                unsigned long long s1 = 0, s2 = 0;
                if (g.dbg) s1 = __builtin_readcyclecounter();
                __builtin_amdgcn_s_barrier();
                if (g.dbg) { s2 = __builtin_readcyclecounter(); d_bar += s2 - s1; }
                const char* const sb = smem + (ROLL ? g.XA : 0) + stage * g.SB;
                int reg_off = 0, edge_off = 0;       // ROLL: this tile's band in the ring; the row above (U = 0) / below (U = 2) it
                if (ROLL) {
                    const int region = it & 3;
                    reg_off = region * RGB;
                    if (U == 0) edge_off = c_rt == 0 ? ZRO : ((region + 3) & 3) * RGB + THm * W16;
                    if (U == 2) edge_off = c_rt == g.RT - 1 ? ZRO : ((region + 1) & 3) * RGB;
                    if (++c_rt == g.RT) c_rt = 0;
                }
                const char* const ab = sb + a_off;
                // a step's 16 pixels start at a multiple of 16 and this lane's pixel is 8 (grp >> 1) + r4 < 16 further: its operand
                // addresses are ONE lane register per operand + a compile-time constant (dz: 256 bytes per step, an immediate of
                // the read) / a scalar (x, ring form: the step's column and row in the ring).  Left to itself hipcc kept sixteen
                // per-step lane addresses and sixteen per-step edge masks in registers across the tile loop and spilled
                const char* const a_lane = ab + l16 * 16;
                // one software pipeline over the tile's eight steps: the eight transposing reads of step i + 1 are issued in
                // front of the products of step i (two operand sets; LDS returns in order, the compiler counts lgkmcnt)
                half4v av[2][2], bv[2][3][2];
                if (C8W_ABL & 2) {
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_) {
                        av[b_][0] = av[b_][1] = half4v{(_Float16)lane, (_Float16)1.f, (_Float16)2.f, (_Float16)3.f};
#pragma unroll
                        for (int v = 0; v < 3; ++v) bv[b_][v][0] = bv[b_][v][1] = av[b_][0];
                    }
                }
                auto load = [&](auto Ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(Ic)::value, B_ = i & 1, gi = NCT == 2 ? i : X + PS * i;     // gi: step of the tile
                    const int p = 16 * gi + l16;
                    if (C8W_ABL & 2) return;
                    av[B_][0] = c8_tr16(a_lane + 256 * gi);
                    av[B_][1] = c8_tr16(a_lane + 256 * gi + 64);
                    const char* xp;
                    if (ROLL) {
                        const int rbase = reg_off + st_ro[i];
                        const int ro = U == 0 ? (st_top[i] ? edge_off : rbase) : U == 1 ? rbase + W16 : (st_bot[i] ? edge_off : rbase + 2 * W16);
                        xp = x_lane + __builtin_amdgcn_readfirstlane(((16 * gi) & Wm) * 16 + ro);
                    } else {
                        xp = sb + b_off + (((p >> g.lgP) * g.THi + ((p >> g.lgW) & THm)) * g.RS + (p & Wm)) * 16 + U * RS16;
                    }
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        bv[B_][v][0] = c8_tr16(xp + v * 16);
                        bv[B_][v][1] = c8_tr16(xp + v * 16 + 64);
                    }
                };
                auto mult = [&](auto Ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(Ic)::value, B_ = i & 1, gi = NCT == 2 ? i : X + PS * i;
                    if (ROLL && !(C8W_ABL & 8)) {
                        // the k element whose cell lies left of column 0 (tap column 0) / right of column W - 1 (tap column 2):
                        // the lane's mask (lmL / lmR) where the step touches the edge (a scalar), all ones elsewhere
                        const unsigned mL = lmL | (unsigned)__builtin_amdgcn_readfirstlane(((16 * gi) & Wm) == 0 ? 0 : -1);
                        const unsigned mR = lmR | (unsigned)__builtin_amdgcn_readfirstlane(((16 * gi + 16) & Wm) == 0 ? 0 : -1);
                        uint2 e0 = __builtin_bit_cast(uint2, bv[B_][0][0]);
                        e0.x &= mL;
                        bv[B_][0][0] = __builtin_bit_cast(half4v, e0);
                        uint2 e2 = __builtin_bit_cast(uint2, bv[B_][2][1]);
                        e2.y &= mR;
                        bv[B_][2][1] = __builtin_bit_cast(half4v, e2);
                    }
                    const half8 a = {av[B_][0][0], av[B_][0][1], av[B_][0][2], av[B_][0][3], av[B_][1][0], av[B_][1][1], av[B_][1][2], av[B_][1][3]};
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        const half8 b = {bv[B_][v][0][0], bv[B_][v][0][1], bv[B_][v][0][2], bv[B_][v][0][3],
                                         bv[B_][v][1][0], bv[B_][v][1][1], bv[B_][v][1][2], bv[B_][v][1][3]};
                        if (C8W_ABL & 1) { asm volatile("" :: "v"(a), "v"(b)); continue; }      // (the reads stay: their values are "used")
                        acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[v], 0, 0, 0);
                    }
                    // the bias product of a step: ONE of the waves that share its dz operand, known at compile time
                    if ((NCT == 2 ? (i % 3 == U && ((i / 3) & 1) == X) : (i % 3 == U)) && want_b && !(C8W_ABL & 1))
                        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ones, accb, 0, 0, 0);
                };
                load(std::integral_constant<int, 0>{});
                // (forcing the next step's reads between this step's products with sched_group_barrier -- product, three
                // reads, product, three reads, product, two reads -- made every wave slower: conv5 of wide6 65.8 k -> 71.7 k
                // cycles per block; hipcc's own order stays)
                // What round 6 tried on this loop and dropped, each a same-box A/B of the wide6 / cifar_like float16 STEP
                // (tools/ab.py, A/A resolution 0.2-0.4 %; docs/EXPERIMENTS.md):
                //  * sched_barrier(0) around load / mult (a true one-step-ahead pipeline; hipcc sinks most reads of step i + 1
                //    next to their own products): +0.9 % -- the waves do not wait for the round trip;
                //  * the three taps cut out of FOUR aligned reads with five v_alignbit (6 reads per step instead of 8): +0.4 ...
                //    +0.8 % -- nor for the LDS' read rate;
                //  * two LDS counters instead of the per-tile s_barrier (a wave starts tile t when the loaders have signalled
                //    it, a loader refills a stage when the twelve compute waves have left it), static or rotating s_setprio:
                //    +0 ... +0.6 % -- nor for each other: the sum of the three waves' step time per SIMD did not move.
#define C8T_ST(I) if (NSTEP > I + 1) { load(std::integral_constant<int, (I + 1) % NSTEP>{}); mult(std::integral_constant<int, I % NSTEP>{}); }
                C8T_ST(0) C8T_ST(1) C8T_ST(2) C8T_ST(3) C8T_ST(4) C8T_ST(5) C8T_ST(6)
#undef C8T_ST
                mult(std::integral_constant<int, NSTEP - 1>{});
                if (g.dbg) d_mm += __builtin_readcyclecounter() - s2;
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (u == 0) { if (xr == 0) run(I0{}, I0{}); else run(I0{}, I1{}); }
        else if (u == 1) { if (xr == 0) run(I1{}, I0{}); else run(I1{}, I1{}); }
        else { if (xr == 0) run(I2{}, I0{}); else run(I2{}, I1{}); }
    }
    if (g.dbg && t == 0) {
        unsigned long long* d = g.dbg + 8 * (size_t)bid;
        d[0] = d_t0; d[2] = __builtin_readcyclecounter(); d[6] = d_bar; d[7] = d_mm;       // (d[1], d[3]: the first loader wave)
        d[4] = d_w0; d[5] = wall_clock64();
    }
    if (g.dbg && lane == 0 && !loader && bid < 256) {      // per compute wave: records 4096 + 16 bid + wave (tools/dbg_c8.py WAVES=1)
        unsigned long long* d = g.dbg + 8 * (size_t)(4096 + 16 * bid + wave);
        d[0] = d_t0; d[2] = __builtin_readcyclecounter(); d[6] = d_bar; d[7] = d_mm;
    }
    __syncthreads();                            // every stage is dead: LDS becomes the epilogue's

    const float os = g.oscale;
    if (NCT == 1) {             // the two step subsets of a (filter tile, tap row) pair: subset 1 onto subset 0 through LDS
        float* const slot = ct_smem + (size_t)(wave % 6) * (3 * 16 * 64) + lane;
        if (!loader && xr == 1) {
#pragma unroll
            for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
                for (int r = 0; r < 16; ++r) slot[(a_ * 16 + r) * 64] = acc[a_][r];
        }
        __syncthreads();
        if (!loader && xr == 0) {
#pragma unroll
            for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a_][r] += slot[(a_ * 16 + r) * 64];
        }
        __syncthreads();
    }
    // ---- bias gradient partial of the slab: the six waves that shared a filter tile's dz (waves ft, ft + 2, ... ft + 10)
    // add up in wave order through LDS; column 0 of the product against ones
    if (cg == 0) {
        if (!loader) {
            float* const bslot = ct_smem + (size_t)wave * 16 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) bslot[r * 64] = accb[r];
        }
        __syncthreads();
        if (wave < 2 && l31 == 0) {                 // wave = ft, (ct, u) = (0, 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sum = 0.f;
#pragma unroll
                for (int role = 0; role < 6; ++role)
                    sum += ct_smem[(size_t)(wave + 2 * role) * 16 * 64 + r * 64 + lane];
                const int k = kg * KBF + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (k < g.K) g.dbws[(size_t)z * g.K + k] = sum * os;
            }
        }
        __syncthreads();
    }
    // ---- slab z: dW layout, tap (u, v) of the correlation is element (2 - u, 2 - v) = 8 - (3 u + v)
    if (g.K % KBF == 0 && g.C % CBF == 0) {
        constexpr int ROWF = CBF * 9;               // through LDS [k][c * 9 + tap]: every thread stores 16 bytes beside its neighbour's
        if (!loader && (NCT == 2 || xr == 0)) {
            float* const T = ct_smem + (size_t)(ft * 32 + 4 * hi) * ROWF + (ct * 32 + l31) * 9;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int v = 0; v < 3; ++v) T[((r & 3) + 8 * (r >> 2)) * ROWF + 8 - (3 * u + v)] = acc[v][r] * os;
        }
        __syncthreads();
        float* const wz = g.ws + (size_t)z * g.K * g.C * 9 + ((size_t)kg * KBF * g.C + (size_t)cg * CBF) * 9;
        constexpr int Q = ROWF / 4;
        for (int idx = t; idx < KBF * Q; idx += 1024) {
            const int row = idx / Q, q = idx - row * Q;
            *reinterpret_cast<float4*>(wz + (size_t)row * g.C * 9 + 4 * q) =
                *reinterpret_cast<const float4*>(ct_smem + (size_t)row * ROWF + 4 * q);
        }
        return;
    }
    if (loader || (NCT == 1 && xr != 0)) return;
    const int c = cg * CBF + ct * 32 + l31;
    if (c < g.C) {
        float* wz = g.ws + (size_t)z * g.K * g.C * 9;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kg * KBF + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) {
#pragma unroll
                for (int v = 0; v < 3; ++v) wz[((size_t)k * g.C + c) * 9 + 8 - (3 * u + v)] = acc[v][r] * os;
            }
        }
    }
}

static int c8w_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

static void c8w_tiles(int K, int C, int& NFT, int& NCT) {
    NFT = K > 32 ? 2 : 1;
    NCT = C > 32 ? 2 : (C > 8 ? 1 : 0);          // 0: one octet, taps packed into the columns (c8_wgrad_kernel)
}

static int c8w_tr_on() {             // TN_C8_WTR=0: the eight-wave kernel everywhere (A/B)
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("TN_C8_WTR");
        on = e ? atoi(e) : 1;
    }
    return on;
}
static int c8w_roll_on() {           // TN_C8_ROLL=0: the halo-tile form everywhere (A/B)
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("TN_C8_ROLL");
        on = e ? atoi(e) : 1;
    }
    return on;
}
static int c8w_geometry(C8WG& g, int num_cus, bool pool, int tm = 1) {
    const int lgW = c8w_log2(g.Wd);
    if (lgW < 3 || lgW > 6) return 0;                  // rows of 8..64 pixels
    const int TP = 128 * tm;                           // pixels per tile
    int TH = TP / g.Wd;
    g.NI = 1;
    if (TH > g.H) {
        if (TH % g.H) return 0;
        g.NI = TH / g.H;
        TH = g.H;
    } else if (g.H % TH) {
        return 0;
    }
    if (c8w_log2(TH) < 0 || (TH & 1)) return 0;
    g.TH = TH; g.THi = TH + 2; g.RT = g.H / TH;
    g.lgW = lgW; g.lgP = c8w_log2(TH * g.Wd);
    g.RS = g.Wd + 2;
    int NFT, NCT;
    c8w_tiles(g.K, g.C, NFT, NCT);
    const int KP = 4 * NFT, CP = NCT ? 4 * NCT : 1;
    g.XC = g.NI * g.THi * g.RS;
    g.XCH = cdiv(g.XC, 64);
    // plane strides = 64 (mod 256) bytes: the four octet planes a transposing read touches sit on disjoint banks
    g.XPS = (g.XC * 16 + 255) / 256 * 256 + 64;
    g.DPS = TP * 16 + 64;
    g.offD = CP * g.XPS;
    g.offG = g.offD + KP * g.DPS;
    g.offM = g.offG;                                   // (round 3's raw pooled-gradient / mask areas are gone: registers)
    g.offDump = (g.offM + 255) / 256 * 256;            // each stage ends with the fillers' dump KB
    g.SB = g.offDump + 1024;
    g.nQx = CP * g.XCH;
    g.nQd = pool ? 0 : KP * 2 * tm;
    g.NQ = g.nQx + g.nQd;
    g.nstage = 3 * g.SB <= 160 * 1024 ? 3 : 2;
    g.roll = 0; g.XA = 0;
    // (16-pixel rows: the ring only under the sixteen-wave kernel -- its four loader waves are bound by the NUMBER of LDS-DMAs
    // they issue, 8 per tile with the ring against 10 with halo tiles; under the eight-wave kernel the ring lost there, round 4)
    const bool tr_shape = NFT == 2 && NCT >= 1 && tm == 1 && c8w_tr_on();
    if (NCT && tm == 1 && g.NI == 1 && g.RT >= 2 && lgW >= ((c8w_roll_on() == 2 || tr_shape) ? 4 : 5) && c8w_roll_on()) {
        // ROLL (c8_wgrad_kernel): x ring of four TH-row regions + a zero row per plane, no halo columns; three dz stages
        const int xps = (16 + 4 * 2048 + g.Wd * 16 + 16 + 255) / 256 * 256 + 64;
        const int offG = KP * g.DPS, offM = offG;
        const int offDump = (offM + 255) / 256 * 256, sb = offDump + 1024;
        if (CP * xps + 3 * sb <= 160 * 1024) {
            g.roll = 1;
            g.XPS = xps; g.XA = CP * xps;
            g.offD = 0; g.offG = offG; g.offM = offM; g.offDump = offDump; g.SB = sb;
            g.XCH = 2; g.nQx = CP * 2; g.NQ = g.nQx + g.nQd;
            g.nstage = 3;
        }
    }
    g.KG = cdiv(g.K, 32 * NFT);
    g.CG = NCT ? cdiv(g.C, 32 * NCT) : 1;
    g.NTILES = cdiv(g.N, g.NI) * g.RT;
    // Sample slabs for HALF the CUs.  A block owns its CU (512 threads, up to 160 KB of LDS) and is bound by latencies and
    // its fixed costs (prologue, 147 KB of slab through LDS), not by the matrix pipe: with two steps in flight the other
    // stream's launches fill the CUs left free, every block amortises its fixed costs over twice the pixels, and the slabs
    // written here and read back by the update halve (wide6: 189 MB per step).  Same count under every schedule (the
    // slab order is part of the result's bits); one step at a time pays ~10 % for it.  TN_C8_WSLAB_DIV=1: every CU (A/B).
    static int div_ = -1;
    if (div_ < 0) {
        const char* e = getenv("TN_C8_WSLAB_DIV");
        div_ = e && atoi(e) > 0 ? atoi(e) : 2;
    }
    // (first layers -- one octet plane, taps packed, a slab of a few KB -- keep every CU: cifar_like float16 0.3130 ->
    // 0.3031 ms same-box against half)
    int S = num_cus / (NCT == 0 ? 1 : div_) / (g.KG * g.CG);
    if (S > g.NTILES) S = g.NTILES;
    if (S < 1) S = 1;
    g.tpb = cdiv(g.NTILES, S);
    g.S = cdiv(g.NTILES, g.tpb);
    return 1;
}

static size_t c8w_lds_bytes(const C8WG& g) {
    const size_t a = (size_t)g.XA + (size_t)g.nstage * g.SB, red = (size_t)4 * 144 * 64 * sizeof(float);     // (ring +) stages; the final reduction
    return a > red ? a : red;
}

template <int NFT, int NCT, bool POOL, int NGX, int TM = 1, bool ROLL = false>
static int c8w_launch(tn_ctx* ctx, C8WG& g) {
    static bool attr_set = false;
    if (!attr_set) {
        TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&c8_wgrad_kernel<NFT, NCT, POOL, NGX, TM, ROLL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int grid = 8 * cdiv(g.S, 8) * g.KG * g.CG;
    static int dbg_on = -1;
    if (dbg_on < 0) {
        const char* e = getenv("TN_C8_DBG");
        dbg_on = e ? atoi(e) : 0;
    }
    if (dbg_on) {
        if (!c8_dbg_buf) TN_HIP(hipMalloc(&c8_dbg_buf, 8 * sizeof(unsigned long long) * 65536));
        TN_HIP(hipMemsetAsync(c8_dbg_buf, 0, 8 * sizeof(unsigned long long) * 65536, ctx->stream));
        g.dbg = grid <= 65536 ? c8_dbg_buf : nullptr;
    }
    {
        static int exp_ = -1;
        if (exp_ < 0) {
            const char* e = getenv("TN_C8_EXP");
            exp_ = e ? atoi(e) : 0;
        }
        g.exp = exp_;
    }
    c8_wgrad_kernel<NFT, NCT, POOL, NGX, TM, ROLL><<<grid, 512, c8w_lds_bytes(g), ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// the sixteen-wave form (c8_wgrad_tr_kernel): x chunks per LOADER wave and stage
template <int NCT, int NGX, bool POOL, bool ROLL>
static int c8w_tr_launch(tn_ctx* ctx, C8WG& g) {
    static bool attr_set = false;
    if (!attr_set) {
        TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&c8_wgrad_tr_kernel<NCT, NGX, POOL, ROLL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int grid = 8 * cdiv(g.S, 8) * g.KG * g.CG;
    static int dbg_on = -1;
    if (dbg_on < 0) {
        const char* e = getenv("TN_C8_DBG");
        dbg_on = e ? atoi(e) : 0;
    }
    if (dbg_on) {
        if (!c8_dbg_buf) TN_HIP(hipMalloc(&c8_dbg_buf, 8 * sizeof(unsigned long long) * 65536));
        TN_HIP(hipMemsetAsync(c8_dbg_buf, 0, 8 * sizeof(unsigned long long) * 65536, ctx->stream));
        g.dbg = grid <= 65536 ? c8_dbg_buf : nullptr;
    }
    c8_wgrad_tr_kernel<NCT, NGX, POOL, ROLL><<<grid, 1024, c8w_lds_bytes(g), ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
// (NCT planes x 4) x XCH chunks of a halo-tile stage over four loader waves: NCT * XCH each; the ring: 2 chunks per plane
static bool c8w_tr_shape_ok(const C8WG& g, int NCT) {
    if (g.nstage != 3) return false;
    if (g.roll) return true;
    const int xch = g.nQx / (4 * NCT);
    return g.nQx == 4 * NCT * xch && xch >= 2 && xch <= 5;
}
template <int NCT, bool POOL>
static int c8w_tr_go(tn_ctx* ctx, C8WG& g) {
    if (g.roll) return c8w_tr_launch<NCT, 2 * NCT, POOL, true>(ctx, g);
    switch (g.nQx / (4 * NCT)) {
        case 2: return c8w_tr_launch<NCT, 2 * NCT, POOL, false>(ctx, g);
        case 3: return c8w_tr_launch<NCT, 3 * NCT, POOL, false>(ctx, g);
        case 4: return c8w_tr_launch<NCT, 4 * NCT, POOL, false>(ctx, g);
        case 5: return c8w_tr_launch<NCT, 5 * NCT, POOL, false>(ctx, g);
    }
    return tn_fail(ctx, TN_E_ARG, "c8 conv wgrad: %d LDS-DMA chunks per x stage", g.nQx);
}

// x chunks per wave and stage: a compile-time count (the waits are counted); shapes land in one of four buckets
static int c8w_ngx(const C8WG& g) { return cdiv(g.nQx, 8); }
template <int NFT, int NCT, bool POOL>
static int c8w_launch_ng(tn_ctx* ctx, C8WG& g, int tm) {
    const int ngx = c8w_ngx(g);
    if constexpr (NCT == 0) {
        TN_REQUIRE(ngx <= 2, "c8 conv wgrad: %d LDS-DMA chunks per x stage", g.nQx);
        if (tm == 4) return c8w_launch<NFT, 0, POOL, 2, 4>(ctx, g);
        if (tm == 2) return c8w_launch<NFT, 0, POOL, 1, 2>(ctx, g);
        return ngx <= 1 ? c8w_launch<NFT, 0, POOL, 1>(ctx, g) : c8w_launch<NFT, 0, POOL, 2>(ctx, g);
    } else {
        if (g.roll) return c8w_launch<NFT, NCT, POOL, NCT, 1, true>(ctx, g);      // 2 chunks x 4 NCT planes over 8 waves
        if (ngx <= 2) return c8w_launch<NFT, NCT, POOL, 2>(ctx, g);
        if (ngx <= 3) return c8w_launch<NFT, NCT, POOL, 3>(ctx, g);
        if (ngx <= 4) return c8w_launch<NFT, NCT, POOL, 4>(ctx, g);
        TN_REQUIRE(ngx <= 5, "c8 conv wgrad: %d LDS-DMA chunks per x stage", g.nQx);
        return c8w_launch<NFT, NCT, POOL, 5>(ctx, g);
    }
}

static int c8w_run(tn_ctx* ctx, C8WG& g, float* dW, float* db, bool pool) {
    // first layers (one octet plane, small stages): 512- or 256-pixel tiles when every block still gets four of them
    int tm = 1;
    if (g.C <= 8) {
        C8WG g4 = g;
        if (c8w_geometry(g4, ctx->num_cus, pool, 4) && g4.nstage == 3 && g4.tpb >= 4 && c8w_ngx(g4) <= 2) tm = 4;
        if (tm == 1) {          // (the pooled form carries the raw gradient and the mask bytes beside the dz image: 256 pixels)
            C8WG g2 = g;
            if (c8w_geometry(g2, ctx->num_cus, pool, 2) && g2.nstage == 3 && g2.tpb >= 4 && c8w_ngx(g2) <= 1) tm = 2;
        }
    }
    TN_REQUIRE(c8w_geometry(g, ctx->num_cus, pool, tm) && c8w_lds_bytes(g) <= 160 * 1024, "c8 conv wgrad: unsupported shape");
    TN_REQUIRE((long long)g.N * g.C8 * g.H * g.Wd < (1ll << 28) && (long long)g.N * g.K8 * g.H * g.Wd < (1ll << 28),
               "c8 conv wgrad: tensor too large for 32-bit cell offsets");
    int NFT, NCT;
    c8w_tiles(g.K, g.C, NFT, NCT);
    const size_t n = (size_t)g.K * g.C * 9;
    int rc = tn_scratch_get(ctx, ((size_t)g.S * n + (size_t)g.S * g.K) * sizeof(float), &g.ws);
    if (rc) return rc;
    g.dbws = g.ws + (size_t)g.S * n;
    g.oscale = 1.f / ctx->grad_scale;
#define C8W_GO(A, B) rc = pool ? c8w_launch_ng<A, B, true>(ctx, g, tm) : c8w_launch_ng<A, B, false>(ctx, g, tm)
    if (NFT == 2 && NCT >= 1 && tm == 1 && c8w_tr_on() && c8w_tr_shape_ok(g, NCT)) {
        if (NCT == 2) rc = pool ? c8w_tr_go<2, true>(ctx, g) : c8w_tr_go<2, false>(ctx, g);
        else rc = pool ? c8w_tr_go<1, true>(ctx, g) : c8w_tr_go<1, false>(ctx, g);
    } else if (NCT == 0 && NFT == 2) C8W_GO(2, 0);
    else if (NCT == 0) C8W_GO(1, 0);
    else if (NFT == 2 && NCT == 2) C8W_GO(2, 2);
    else if (NFT == 2) C8W_GO(2, 1);
    else if (NCT == 2) C8W_GO(1, 2);
    else C8W_GO(1, 1);
#undef C8W_GO
    if (rc) return rc;
    rc = tn_red_push(ctx, g.ws, dW, (uint32_t)n, (uint32_t)g.S, (uint32_t)n, 0);
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
// halfs of the arranged-weight buffer of a layer's forward (dgrad == 0) or input-gradient (dgrad != 0) product
size_t tn_c8_wt_elems(int K, int C, int dgrad) { return dgrad ? c8_wt_elems(C, K) : c8_wt_elems(K, C); }

// arranged weights of up to 32 products in one launch; segs: host array of tn_c8_wt_seg
int tn_c8_arrange_multi(tn_ctx* ctx, const tn_c8_wt_seg* segs, int nseg) {
    TN_REQUIRE(nseg >= 0 && nseg <= 32 && (segs != nullptr || nseg == 0), "tn_c8_arrange_multi: 0..32 segments");
    if (!nseg) return TN_OK;
    C8WtBatch b;
    int mx = 0;
    for (int i = 0; i < nseg; ++i) {
        // input gradient: the roles of filters and channels swap ("filters" = the layer's input channels)
        const int K = segs[i].dgrad ? segs[i].C : segs[i].K, C = segs[i].dgrad ? segs[i].K : segs[i].C;
        const int KBF = 32 * c8_pick_ft(K);
        b.s[i].W = segs[i].W; b.s[i].wt = static_cast<_Float16*>(segs[i].wt);
        b.s[i].K = K; b.s[i].C = C; b.s[i].KBF = KBF; b.s[i].nchunk = cdiv(C, 16);
        b.s[i].total = (int)c8_wt_total(K, C, segs[i].dgrad); b.s[i].dgrad = segs[i].dgrad;
        b.s[i].tk = c8_tap_packed(C, segs[i].dgrad) ? 1 : 0;
        if (b.s[i].total > mx) mx = b.s[i].total;
    }
    c8_wt_multi_kernel<<<dim3(cdiv(mx, 256), nseg), 256, 0, ctx->stream>>>(b);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_c8_conv_fwd(tn_ctx* ctx, const void* x, const float* W, const float* b, void* y, uint8_t* mask, int N, int C,
                   int H, int Wd, int K, int act, float prm, int pool, const void* wt) {
    TN_REQUIRE((K & 7) == 0, "c8 conv: the number of filters must be a multiple of 8 (got %d)", K);
    C8G g{};
    g.x = static_cast<const _Float16*>(x); g.out = static_cast<_Float16*>(y); g.bias = b; g.mask_out = mask;
    g.N = N; g.C8 = (C + 7) / 8; g.K8 = K / 8; g.H = H; g.W = Wd; g.act = act; g.prm = prm;
    return pool ? c8_run<1>(ctx, g, W, K, C, wt) : c8_run<0>(ctx, g, W, K, C, wt);
}

// dx (N, C, H, W) = conv^T(dz, W) * act'(prev_a) of the layer below (prev_a NULL: no activation below; for a pooled
// block below prev_a is its POOLED output and dx has its shape -- the gradient a pooled block receives always carries
// act'(pooled output)).  pooled != 0: dz is not a tensor: the `dz` argument is the pooled gradient (N, K, H/2, W/2) and
// dz = (bit of the window element in the block's mask) ? pooled gradient : 0, gathered while staging
int tn_c8_conv_dgrad(tn_ctx* ctx, const void* dz, const float* W, void* dx, int N, int C, int H, int Wd, int K,
                     const void* prev_a, int prev_act, float prev_prm, int pooled, const uint8_t* mask, const void* wt) {
    TN_REQUIRE((K & 7) == 0, "c8 conv: the number of filters must be a multiple of 8 (got %d)", K);
    TN_REQUIRE(!pooled || mask != nullptr, "c8 conv dgrad: a pooled block needs its mask");
    C8G g{};
    g.x = static_cast<const _Float16*>(dz); g.out = static_cast<_Float16*>(dx);
    g.prev_a = static_cast<const _Float16*>(prev_a);
    g.N = N; g.C8 = K / 8; g.K8 = (C + 7) / 8; g.H = H; g.W = Wd; g.act = prev_act; g.prm = prev_prm;
    g.mask_in = mask;
    // the roles of filters and channels swap: "filters" = the C input channels (rounded up to whole octets: the
    // arranged weights of channels beyond C are zero, so their cells come out zero)
    return pooled ? c8_run<3>(ctx, g, W, C, K, wt) : c8_run<2>(ctx, g, W, C, K, wt);
}

// dW (K, C, 3, 3), db (K) from x and dz (pooled != 0: dz is gathered from the pooled gradient and the block's mask as in
// tn_c8_conv_dgrad); dz carries the gradient scale, the results do not
int tn_c8_conv_wgrad(tn_ctx* ctx, const void* x, const void* dz, float* dW, float* db, int N, int C, int H, int Wd,
                     int K, int pooled, const uint8_t* mask) {
    TN_REQUIRE((K & 7) == 0, "c8 conv: the number of filters must be a multiple of 8 (got %d)", K);
    TN_REQUIRE(!pooled || mask != nullptr, "c8 conv wgrad: a pooled block needs its mask");
    C8WG g{};
    g.x = static_cast<const uint4*>(x); g.dz = static_cast<const uint4*>(dz);
    g.mask = reinterpret_cast<const uint2*>(mask);
    g.N = N; g.C = C; g.C8 = (C + 7) / 8; g.H = H; g.Wd = Wd; g.K = K; g.K8 = K / 8;
    return c8w_run(ctx, g, dW, db, pooled != 0);
}

// (the answer does not depend on the device's CU count: in c8w_geometry it only sets the number of slabs S and
// the tiles per slab, never the tile shape, the LDS stage or the DMA chunk counts the limits below are about -- so the
// construction-time query and c8w_run, which passes ctx->num_cus, always agree)
// The query has no context argument; it asks the current device for its CU count (what tn_ctx_create stores in
// ctx->num_cus and c8w_run passes), 256 = MI355X where no device answers (GPU-less construction checks).
static int c8_current_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return 256;
    }
    return n;
}
int tn_c8_conv_wgrad_supported(int N, int C, int H, int Wd, int K) {
    const int cus = c8_current_cus();       // (asked every time: the answer must follow the device that is current NOW;
                                            //  what the check reads -- LDS bytes, DMA chunks per stage -- does not depend on it)
    C8WG g{};
    g.N = N; g.C = C; g.C8 = (C + 7) / 8; g.H = H; g.Wd = Wd; g.K = K; g.K8 = K / 8;
    if ((K & 7) || !c8w_geometry(g, cus, true) || c8w_ngx(g) > 5) return 0;
    if (c8w_lds_bytes(g) > 160 * 1024) return 0;
    if (!c8w_geometry(g, cus, false) || c8w_ngx(g) > 5) return 0;
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
