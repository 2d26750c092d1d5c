// 3x3 stride-1 convolution with fp16 OPERANDS and fp32 ACCUMULATION on the matrix cores
// (v_mfma_f32_32x32x16_f16), input tile resident in LDS -- BASELINE.json configs[4] ("fp16 inputs /
// fp32 accum MFMA"); same products as theanet/layer/convpool.py:54-72 and their Theano gradients
// (CorrMM_gradInputs / CorrMM_gradWeights), with both operands of every product rounded to fp16
// (round-to-nearest-even) first.  Tensors stay fp32 in HBM (fp32 master weights, fp32 activations):
// the rounding happens while a tile is staged into LDS, so nothing else in the net changes.
//
// Gradients are small (|dz| ~ 1e-3/B and below: fp16 subnormal territory), so every product that has
// dz as an operand rounds gs*dz (gs = a power of two: exact) and multiplies the fp32 result by 1/gs.
//
// forward / dgrad (conv_tile16_kernel): block = 256 output pixels x 32*FT filters, chunk = 16 input
//   channels.  LDS holds the halo tile as two "octet planes" [o][pixel][8 channels] of halfs (16 bytes
//   per pixel and octet: lane = pixel, so the B operand of a tap is ONE conflict-free ds_read_b128 at
//   lane base + constant) and the weights as [tap][o][filter][8 channels] (A operand: one
//   ds_read_b128).  Per tap 2 + FT reads feed 2*FT MFMAs.  The NCHW -> [pixel][channel] transposition
//   is done by the staging thread: 8 channel loads of 4 pixels (16-byte coalesced) -> 4 x 8 halfs.
// wgrad (conv_tile16_wgrad_kernel): GEMM rows = filters, columns = input channels at a fixed tap,
//   reduction = pixels, 16 per MFMA: A = dz[filter][8 consecutive pixels] (one ds_read_b128), B =
//   x[channel][the same 8 pixels shifted by the tap]: the centre column is an aligned ds_read_b128,
//   the +-1 column shifts are v_alignbit funnel shifts with the neighbouring dwords.  A wave owns one
//   (32 filters x 32 channels) pair and keeps all nine taps (9 x 16 accumulators).
#include "conv_tile_common.h"

#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

int tn_red_push(tn_ctx* ctx, const float* src, float* out, uint32_t n, uint32_t S, uint32_t stride, uint32_t flip);
int tn_red_commit(tn_ctx* ctx);

// wt16[kt][chunk][tap][o][j][e] (halfs): filter kt*KBF + j, channel chunk*16 + 8*o + e, correlation tap
__global__ __launch_bounds__(256) void conv_tile16_wt_kernel(const float* __restrict__ W, _Float16* __restrict__ wt,
                                                            int K, int C, int KBF, int nchunk, int total,
                                                            int dgrad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int r = idx;
    const int e = r & 7; r >>= 3;
    const int j = r % KBF; r /= KBF;
    const int o = r & 1; r >>= 1;
    const int tap = r % 9; r /= 9;
    const int chunk = r % nchunk;
    const int kt = r / nchunk;
    const int filt = kt * KBF + j, ch = chunk * 16 + 8 * o + e;
    float v = 0.f;
    if (filt < K && ch < C)
        v = dgrad ? W[((size_t)ch * K + filt) * 9 + tap]          // W[k = ch][c = filt][u][v]
                  : W[((size_t)filt * C + ch) * 9 + (8 - tap)];   // true convolution: flipped taps
    wt[idx] = (_Float16)v;
}

struct C16Slot { int g, l, n, row, col, o; bool ok; };
// staging slot s of thread t: (channel octet, image, tile row, 4-pixel column group)
__device__ __forceinline__ C16Slot c16_slot(const ConvTG& g, int t, int s, int n0, int r0) {
    C16Slot o;
    const int e = t + 256 * s;
    const bool ok = e < g.nx4;
    int rr = min(e, g.nx4 - 1);
    const int q = rr % g.q4; rr /= g.q4;
    const int r = rr % g.THi; rr /= g.THi;
    const int ni = rr % g.NI;
    o.o = rr / g.NI;
    const int in_row = r0 - g.pad + r, n = n0 + ni;
    o.ok = ok && (unsigned)in_row < (unsigned)g.H && n < g.N;
    o.n = min(n, g.N - 1); o.row = min(max(in_row, 0), g.H - 1); o.col = 4 * q;
    o.g = (o.n * g.C * g.H + o.row) * g.Wd + 4 * q;
    o.l = (o.o * g.plane + (ni * g.THi + r) * g.RS + g.pad + 4 * q) * 16;
    return o;
}

__device__ __forceinline__ half8 c16_pack(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                          float a7) {
    half8 h = {(_Float16)a0, (_Float16)a1, (_Float16)a2, (_Float16)a3,
               (_Float16)a4, (_Float16)a5, (_Float16)a6, (_Float16)a7};
    return h;
}

// NS: staging slots per thread for the input tile (1 or 2)
template <int FT, bool DGRAD, bool POOL, int NS>
__global__ __launch_bounds__(256, (FT <= 2 && NS == 1) ? 2 : 1) void conv_tile16_kernel(ConvTG g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr int KBF = 32 * FT;
    constexpr int WB = 9 * 2 * KBF * 16;              // bytes of one weight chunk
    constexpr int WS = (WB / 16 + 255) / 256;         // 16-byte staging slots per thread (3, 5 or 9)
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
    const int HW = g.H * g.Wd;
    unsigned long long* dbg = (g.dbg && t == 0) ? g.dbg + 8 * (size_t)bid : nullptr;
    if (dbg) { dbg[0] = __builtin_readcyclecounter(); dbg[4] = wall_clock64(); }

    for (int i = t * 16; i < 2 * XB; i += 4096) *reinterpret_cast<float4*>(Xs + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    C16Slot sl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sl[s] = c16_slot(g, t, s, n0, r0);
    const char* wsrc = reinterpret_cast<const char*>(g.wt) + (size_t)kt * g.nchunk * WB + 16 * t;

    // this lane's two pixels (B operand): byte offset of the window's top-left cell in its octet plane
    int boff[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int p = wave * 64 + pt * 32 + l31;
        const int pp = p < g.TP ? p : 0;
        const int per = g.TH * g.Wo;
        const int ni = pp / per, rem = pp - ni * per;
        const int r = rem / g.Wo, col = rem - r * g.Wo;
        boff[pt] = (hi * g.plane + (ni * g.THi + r) * g.RS + col) * 16;
    }
    const int aoff = (hi * KBF + l31) * 16;

    f32x16 acc[FT][2];
#pragma unroll
    for (int a = 0; a < FT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // The input tile is fetched TWO chunks ahead (two register sets, alternating by chunk parity): a chunk's
    // MFMAs last ~1.2 k cycles, less than an HBM round trip under load, so one chunk of look-ahead left every
    // block waiting for its loads (cycle stamps: 4 k cycles per chunk).  Weights come from L2: one chunk ahead.
    // (the pooled-gradient variant forms dz from three loads per value: two sets would halve its occupancy)
    constexpr bool LA2 = !(DGRAD && POOL);
    float4 xr[LA2 ? 2 : 1][NS][8];
    // weight staging slots live in named registers (an indexed array ended up in scratch memory)
    uint4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;
#define C16_WL(J, R) if (WS > J) R = *reinterpret_cast<const uint4*>(w_ + min(4096 * J, WB - 16 - 16 * t))
#define C16_WST(J, R) if (WS > J && (4096 * (J + 1) <= WB || 16 * t + 4096 * J < WB)) *reinterpret_cast<uint4*>(wb + 4096 * J) = R
    auto gloadx = [&](int chunk, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const int ch = min(chunk, g.nchunk - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // channels beyond C meet zero weights: any finite value will do (clamped re-read)
                const int cc = min(ch * 16 + 8 * sl[s].o + e, g.C - 1);
                if (DGRAD && POOL) xr[P][s][e] = pool_expand4(g.ps, sl[s].n * g.C + cc, sl[s].row, sl[s].col);
                else xr[P][s][e] = *reinterpret_cast<const float4*>(g.x + sl[s].g + cc * HW);
            }
    };
    auto gloadw = [&](int chunk) __attribute__((always_inline)) {
        const int ch = min(chunk, g.nchunk - 1);
        const char* w_ = wsrc + (size_t)ch * WB;
        C16_WL(0, wr0); C16_WL(1, wr1); C16_WL(2, wr2); C16_WL(3, wr3); C16_WL(4, wr4);
        C16_WL(5, wr5); C16_WL(6, wr6); C16_WL(7, wr7); C16_WL(8, wr8);
    };
    auto lstore = [&](int buf, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        char* xb = Xs + buf * XB;
        char* wb = Ws + buf * WB + 16 * t;
        const float sc = DGRAD ? g.iscale : 1.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (sl[s].ok) {
                half8* dst = reinterpret_cast<half8*>(xb + sl[s].l);
#define q xr[P][s]
                if (DGRAD) {
                    dst[0] = c16_pack(sc * q[0].x, sc * q[1].x, sc * q[2].x, sc * q[3].x, sc * q[4].x, sc * q[5].x, sc * q[6].x, sc * q[7].x);
                    dst[1] = c16_pack(sc * q[0].y, sc * q[1].y, sc * q[2].y, sc * q[3].y, sc * q[4].y, sc * q[5].y, sc * q[6].y, sc * q[7].y);
                    dst[2] = c16_pack(sc * q[0].z, sc * q[1].z, sc * q[2].z, sc * q[3].z, sc * q[4].z, sc * q[5].z, sc * q[6].z, sc * q[7].z);
                    dst[3] = c16_pack(sc * q[0].w, sc * q[1].w, sc * q[2].w, sc * q[3].w, sc * q[4].w, sc * q[5].w, sc * q[6].w, sc * q[7].w);
                } else {
                    dst[0] = c16_pack(q[0].x, q[1].x, q[2].x, q[3].x, q[4].x, q[5].x, q[6].x, q[7].x);
                    dst[1] = c16_pack(q[0].y, q[1].y, q[2].y, q[3].y, q[4].y, q[5].y, q[6].y, q[7].y);
                    dst[2] = c16_pack(q[0].z, q[1].z, q[2].z, q[3].z, q[4].z, q[5].z, q[6].z, q[7].z);
                    dst[3] = c16_pack(q[0].w, q[1].w, q[2].w, q[3].w, q[4].w, q[5].w, q[6].w, q[7].w);
                }
#undef q
            }
        }
        C16_WST(0, wr0); C16_WST(1, wr1); C16_WST(2, wr2); C16_WST(3, wr3); C16_WST(4, wr4);
        C16_WST(5, wr5); C16_WST(6, wr6); C16_WST(7, wr7); C16_WST(8, wr8);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    gloadx(0, P0{});
    gloadw(0);
    if constexpr (LA2) gloadx(1, P1{});
    __syncthreads();                 // the clearing is done
    lstore(0, P0{});
    __syncthreads();
    if (dbg) dbg[1] = __builtin_readcyclecounter();
    const int RS16 = g.RS * 16;
    auto body = [&](int chunk, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        gloadw(chunk + 1);
        if constexpr (LA2) gloadx(chunk + 2, Pc);
        else gloadx(chunk + 1, P0{});
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
        if (chunk + 1 < g.nchunk) lstore(P ^ 1, std::integral_constant<int, LA2 ? (P ^ 1) : 0>{});
        __syncthreads();
    };
    for (int chunk = 0; chunk < g.nchunk; chunk += 2) {
        body(chunk, P0{});
        if (chunk + 1 < g.nchunk) body(chunk + 1, P1{});
    }
    if (dbg) dbg[2] = __builtin_readcyclecounter();
    if (DGRAD) {
        const float os = g.oscale;
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][pt][r] *= os;
    }
    ct_epilogue<FT, DGRAD, POOL>(g, acc, ct_smem, kt, n0, r0, lane, wave, l31, hi);
    if (dbg) { dbg[3] = __builtin_readcyclecounter(); dbg[5] = wall_clock64(); }
#undef C16_WL
#undef C16_WST
}

static bool c16_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("TN_CONV_TILE16");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

static int c16_pick_ft(int K) {
    return K > 32 ? 2 : 1;
}

// geometry of the pixel tiling; returns 0 when the shape is outside the kernel's limits
static int c16_geometry(ConvTG& g, int FT) {
    if (g.Wo > 256 || (g.Wd & 3) || g.pad < 0 || g.pad > 2) return 0;
    int TH = 256 / g.Wo;
    if (TH >= g.Ho) {
        g.TH = g.Ho; g.RT = 1;
        g.NI = 256 / (g.Ho * g.Wo);
        if (g.NI < 1) g.NI = 1;
        if (g.NI > g.N) g.NI = g.N;
    } else {
        g.RT = cdiv(g.Ho, TH);
        g.TH = cdiv(g.Ho, g.RT);
        g.NI = 1;
    }
    g.TP = g.NI * g.TH * g.Wo;
    g.THi = g.TH + 2;
    g.LP = 0;
    int need = g.Wo + 2;
    if (need < g.pad + g.Wd) need = g.pad + g.Wd;
    g.RS = need;
    if (g.Wo < 32)          // a wave's 32 pixels span several rows: keep their 16-byte cells on distinct banks
        while ((g.RS - g.Wo) & 15) ++g.RS;
    g.plane = g.NI * g.THi * g.RS + 2;                    // pixels per octet plane (+ the window overhang)
    g.q4 = g.Wd / 4;
    g.nx4 = 2 * g.NI * g.THi * g.q4;
    if (g.nx4 > 2 * 256) return 0;
    g.nchunk = cdiv(g.C, 16);
    g.KT = cdiv(g.K, 32 * FT);
    g.MT = cdiv(g.N, g.NI) * g.RT;
    return 1;
}

static size_t c16_lds_bytes(const ConvTG& g, int FT) {
    const size_t loop = (size_t)2 * (2 * g.plane * 16 + 9 * 2 * 32 * FT * 16);
    const size_t epi = (size_t)32 * FT * 256 * sizeof(float);        // the transposed output tile
    return loop > epi ? loop : epi;
}

static unsigned long long* c16_dbg_buf = nullptr;
// debugging aid (not part of the C-ABI): the cycle stamps of the last stamped launch (TN_CT_DBG=1)
extern "C" int tn_conv_tile16_dbg_read(tn_ctx* ctx, unsigned long long* host, int nblocks) {
    if (!c16_dbg_buf) return -1;
    hipStreamSynchronize(ctx->stream);
    return hipMemcpy(host, c16_dbg_buf, (size_t)nblocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

template <int FT, bool DGRAD, bool POOL>
static int c16_launch(tn_ctx* ctx, ConvTG& g) {
    static bool attr_set[2] = {false, false};
    size_t lds = c16_lds_bytes(g, FT);
    const int ns = g.nx4 > 256 ? 2 : 1;
    const int grid = 8 * cdiv(g.MT, 8) * g.KT;
    if (getenv("TN_CT_DBG")) {
        if (!c16_dbg_buf) TN_HIP(hipMalloc(&c16_dbg_buf, 8 * sizeof(unsigned long long) * 65536));
        g.dbg = grid <= 65536 ? c16_dbg_buf : nullptr;
    }
    if (ns == 1) {
        if (!attr_set[0]) {
            TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile16_kernel<FT, DGRAD, POOL, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[0] = true;
        }
        conv_tile16_kernel<FT, DGRAD, POOL, 1><<<grid, 256, lds, ctx->stream>>>(g);
    } else {
        if (!attr_set[1]) {
            TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile16_kernel<FT, DGRAD, POOL, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[1] = true;
        }
        conv_tile16_kernel<FT, DGRAD, POOL, 2><<<grid, 256, lds, ctx->stream>>>(g);
    }
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// 1 if conv_tile16_kernel handles the (gathered tensor N,C,H,Wd; K filters; pad; output Ho,Wo) problem
int tn_conv_tile16_ok(const float* x, int N, int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo) {
    if (!c16_enabled() || f != 3 || pad < 0 || pad > 2) return 0;
    if (reinterpret_cast<uintptr_t>(x) & 15) return 0;
    ConvTG g{};
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.pad = pad; g.Ho = Ho; g.Wo = Wo;
    const int FT = c16_pick_ft(K);
    if (!c16_geometry(g, FT)) return 0;
    return c16_lds_bytes(g, FT) <= 156 * 1024;
}

static int c16_run(tn_ctx* ctx, ConvTG& g, const float* W, bool dgrad, bool pool = false) {
    const int FT = c16_pick_ft(g.K);
    TN_REQUIRE(c16_geometry(g, FT) && c16_lds_bytes(g, FT) <= 156 * 1024, "conv_tile16: unsupported shape");
    TN_REQUIRE((long long)g.N * g.C * g.H * g.Wd < (1ll << 31) && (long long)g.N * g.K * g.Ho * g.Wo < (1ll << 31),
               "conv_tile16: tensor too large for 32-bit offsets");
    const int KBF = 32 * FT, total = g.KT * g.nchunk * 9 * 2 * KBF * 8;
    float* wt;
    int rc = tn_scratch_get(ctx, (size_t)total * sizeof(_Float16), &wt);
    if (rc) return rc;
    conv_tile16_wt_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(W, reinterpret_cast<_Float16*>(wt), g.K, g.C,
                                                                    KBF, g.nchunk, total, dgrad ? 1 : 0);
    TN_LAUNCH_CHECK();
    g.wt = wt;
    g.vec_out = (g.Wo % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.out) | reinterpret_cast<uintptr_t>(g.prev_a)) & 15) == 0;
    if (pool) {
        if (dgrad) {
            if (FT == 4) return c16_launch<4, true, true>(ctx, g);
            return FT == 2 ? c16_launch<2, true, true>(ctx, g) : c16_launch<1, true, true>(ctx, g);
        }
        if (FT == 4) return c16_launch<4, false, true>(ctx, g);
        return FT == 2 ? c16_launch<2, false, true>(ctx, g) : c16_launch<1, false, true>(ctx, g);
    }
    if (dgrad) {
        if (FT == 4) return c16_launch<4, true, false>(ctx, g);
        return FT == 2 ? c16_launch<2, true, false>(ctx, g) : c16_launch<1, true, false>(ctx, g);
    }
    if (FT == 4) return c16_launch<4, false, false>(ctx, g);
    return FT == 2 ? c16_launch<2, false, false>(ctx, g) : c16_launch<1, false, false>(ctx, g);
}

int tn_conv_tile16_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N, int C,
                       int H, int Wd, int K, int pad, int Ho, int Wo, int act, float prm) {
    ConvTG g{};
    g.x = x; g.out = a; g.bias = b;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.pad = pad; g.Ho = Ho; g.Wo = Wo;
    g.act = act; g.prm = prm; g.iscale = 1.f; g.oscale = 1.f;
    return c16_run(ctx, g, W, false);
}

// dx (N,C,H,Wd) from dz (N,K,Ho,Wo): the forward kernel with (channels, filters) = (K, C), padding 2 - pad
int tn_conv_tile16_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int N, int C, int H,
                         int Wd, int K, int pad, int Ho, int Wo, const float* prev_a, int act, float prm) {
    ConvTG g{};
    g.x = dz; g.out = dx; g.prev_a = prev_a;
    g.N = N; g.C = K; g.H = Ho; g.Wd = Wo; g.K = C; g.pad = 2 - pad; g.Ho = H; g.Wo = Wd;
    g.act = act; g.prm = prm; g.iscale = ctx->grad_scale; g.oscale = 1.f / ctx->grad_scale;
    return c16_run(ctx, g, W, true);
}

// =================================================================================================
// Weight gradient of a 3x3 'same' convolution, fp16 operands:
//   dW[k][c][2-u][2-v] = sum_{n,i,j} h(gs*dz[n,k,i,j]) * h(x[n,c,i-1+u,j-1+v]) / gs
// Block = 32*NFT filters x 32*NCT channels x a range of 128-pixel tiles; wave = one (filter tile,
// channel tile) pair (and, when NFT*NCT < 4, one of PS interleaved step subsets) with all nine taps.
// LDS per buffer: dz [filter][128 pixels + 8] halfs, x [channel][(TH+2) rows][8 + Wd] halfs (the 8
// leading cells of a row are zero: the left halo of this row and the right halo of the previous one).
// =================================================================================================
#define CW16_DZROW 136

struct ConvWG16 {
    const float* x;        // (N, C, H, Wd)
    const float* dz;       // (N, K, H, Wd)
    float* ws;             // [S * PS][K*C*9] partial weight gradients, dW layout
    float* dbws;           // [S][K] partial bias gradients
    int N, C, H, Wd, K;
    int KG, CG, S, tpb;    // filter groups, channel groups, slabs, tiles per slab
    int NI, TH, THi, RT, NTILES;
    int RS, plane, q4, P, lgW, lgP;
    float gscale, oscale;
    PoolSrc ps;            // POOL: dz is formed from (g, mask, y) while it is staged
};

__device__ __forceinline__ float4 cw16_mask4(float4 v, bool ok) {
    const int m = ok ? -1 : 0;
    return make_float4(__int_as_float(__float_as_int(v.x) & m), __int_as_float(__float_as_int(v.y) & m),
                       __int_as_float(__float_as_int(v.z) & m), __int_as_float(__float_as_int(v.w) & m));
}

template <int NFT, int NCT, bool POOL>
__global__ __launch_bounds__(256) void conv_tile16_wgrad_kernel(ConvWG16 g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr int KBF = 32 * NFT, CBF = 32 * NCT, PS = 4 / (NFT * NCT), SPW = 8 / PS;
    constexpr int NDZ = NFT * 4;                      // dz staging slots per thread (float4)
    constexpr int NX = 8 * NCT;                       // x staging slots per thread (float4): 4 channels per slot
    constexpr int DZSZ = KBF * CW16_DZROW * 2;        // bytes
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

    for (int i = t * 16; i < 2 * BUFSZ; i += 4096) *reinterpret_cast<float4*>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- staging geometry of this thread (the same for every tile) ----
    // dz slot s: filter 8*s + (t >> 5), pixels 4*(t & 31) .. +3 of the tile
    const int dq = t & 31, dp = 4 * dq;
    const int d_ni = dp >> g.lgP, d_row = (dp >> g.lgW) & THm, d_col = dp & Wm;
    // x slot s: channel 4*s + (t >> 6), cell (image, tile row, 4-pixel group) = t & 63 of P
    const int xi = t & 63;
    const bool x_on = xi < g.P;
    int xr_ = min(xi, g.P - 1);
    const int x_q = xr_ % g.q4; xr_ /= g.q4;
    const int x_r = xr_ % g.THi, x_ni = xr_ / g.THi;
    const int x_lds = DZSZ + (t >> 6) * g.plane * 2 + ((x_ni * g.THi + x_r) * g.RS + 8 + 4 * x_q) * 2;

    float dbacc[NDZ];
#pragma unroll
    for (int s = 0; s < NDZ; ++s) dbacc[s] = 0.f;
    float4 dv[NDZ], xv[NX];

    auto gload = [&](int tile) {
        const int gi = tile / g.RT, rt = tile - gi * g.RT;
        const int n0 = gi * g.NI, r0 = rt * g.TH;
        {
            const int n = n0 + d_ni, row = r0 + d_row;
            const bool okn = n < g.N;
            const int nn = min(n, g.N - 1);
#pragma unroll
            for (int s = 0; s < NDZ; ++s) {
                const int k = kg * KBF + 8 * s + (t >> 5);
                const int pl = nn * g.K + min(k, g.K - 1);
                float4 v;
                if (POOL) v = pool_expand4(g.ps, pl, row, d_col);
                else v = *reinterpret_cast<const float4*>(g.dz + ((size_t)pl * g.H + row) * g.Wd + d_col);
                dv[s] = cw16_mask4(v, okn && k < g.K);
            }
        }
        {
            const int n = n0 + x_ni, row = r0 - 1 + x_r;
            const bool okr = x_on && n < g.N && (unsigned)row < (unsigned)g.H;
            const int nn = min(n, g.N - 1), rr = min(max(row, 0), g.H - 1);
#pragma unroll
            for (int s = 0; s < NX; ++s) {
                const int c = cg * CBF + 4 * s + (t >> 6);
                const float4 v = *reinterpret_cast<const float4*>(
                    g.x + ((size_t)(nn * g.C + min(c, g.C - 1)) * g.H + rr) * g.Wd + 4 * x_q);
                xv[s] = cw16_mask4(v, okr && c < g.C);
            }
        }
    };
    auto lstore = [&](int buf, float dbw) {
        char* base = smem + buf * BUFSZ;
        const float gs = g.gscale;
#pragma unroll
        for (int s = 0; s < NDZ; ++s) {
            half4v h = {(_Float16)(gs * dv[s].x), (_Float16)(gs * dv[s].y), (_Float16)(gs * dv[s].z),
                        (_Float16)(gs * dv[s].w)};
            *reinterpret_cast<half4v*>(base + ((8 * s + (t >> 5)) * CW16_DZROW + dp) * 2) = h;
            dbacc[s] += dbw * ((dv[s].x + dv[s].y) + (dv[s].z + dv[s].w));
        }
        if (x_on) {
#pragma unroll
            for (int s = 0; s < NX; ++s) {
                half4v h = {(_Float16)xv[s].x, (_Float16)xv[s].y, (_Float16)xv[s].z, (_Float16)xv[s].w};
                *reinterpret_cast<half4v*>(base + x_lds + 4 * s * g.plane * 2) = h;
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
    lstore(0, 1.f);
    __syncthreads();

    const int RS2 = g.RS * 2;
    int cur = 0;
    for (int tile = tile_beg; tile < tile_end; ++tile, cur ^= 1) {
        const bool hasnext = tile + 1 < tile_end;
        gload(hasnext ? tile + 1 : tile);            // (the last tile re-stages itself: branch-free body)
        const char* dzb = smem + cur * BUFSZ + (ft * 32 + l31) * (CW16_DZROW * 2) + 16 * hi;
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
        lstore(cur ^ 1, hasnext ? 1.f : 0.f);
        __syncthreads();
    }

    // bias gradient partial of the slab: per-filter sums of the (unrounded) dz this block staged
    if (cg == 0) {
#pragma unroll
        for (int s = 0; s < NDZ; ++s) {
            float v = dbacc[s];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            const int k = kg * KBF + (t >> 5) + 8 * s;
            if (l31 == 0 && k < g.K) g.dbws[(size_t)z * g.K + k] = v;
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

static int cw16_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

static void cw16_tiles(int K, int C, int& NFT, int& NCT) {
    NFT = K > 32 ? 2 : 1;
    NCT = C > 32 ? 2 : 1;
}

static int cw16_geometry(ConvWG16& g, int num_cus) {
    const int lgW = cw16_log2(g.Wd);
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
    if (cw16_log2(TH) < 0) return 0;
    g.TH = TH; g.THi = TH + 2; g.RT = g.H / TH;
    g.lgW = lgW; g.lgP = cw16_log2(TH * g.Wd);
    g.RS = g.Wd + 8;
    g.plane = g.NI * g.THi * g.RS;                     // halfs; 16 bytes * odd apart: conflict-free 16-byte reads
    g.plane += ((g.plane >> 3) & 1) ? 16 : 8;
    g.q4 = g.Wd / 4;
    g.P = g.NI * g.THi * g.q4;
    if (g.P > 64) return 0;
    int NFT, NCT;
    cw16_tiles(g.K, g.C, NFT, NCT);
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

static size_t cw16_lds_bytes(const ConvWG16& g) {
    int NFT, NCT;
    cw16_tiles(g.K, g.C, NFT, NCT);
    return (size_t)2 * (32 * NFT * CW16_DZROW * 2 + 32 * NCT * g.plane * 2);
}

int tn_conv_tile16_wgrad_ok(tn_ctx* ctx, const float* x, const float* dz, int N, int C, int H, int Wd, int K,
                            int f, int pad, int Ho, int Wo) {
    if (!c16_enabled() || f != 3 || pad != 1 || Ho != H || Wo != Wd || C * 9 <= 32) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz)) & 15) return 0;
    ConvWG16 g{};
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    if (!cw16_geometry(g, ctx ? ctx->num_cus : 256)) return 0;
    return cw16_lds_bytes(g) <= 160 * 1024;
}

template <int NFT, int NCT, bool POOL>
static int cw16_launch(tn_ctx* ctx, ConvWG16& g) {
    static bool attr_set = false;
    if (!attr_set) {
        TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile16_wgrad_kernel<NFT, NCT, POOL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int grid = 8 * cdiv(g.S, 8) * g.KG * g.CG;
    conv_tile16_wgrad_kernel<NFT, NCT, POOL><<<grid, 256, cw16_lds_bytes(g), ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static int cw16_run(tn_ctx* ctx, ConvWG16& g, float* dW, float* db, bool pool) {
    TN_REQUIRE(cw16_geometry(g, ctx->num_cus) && cw16_lds_bytes(g) <= 160 * 1024,
               "conv_tile16_wgrad: unsupported shape");
    TN_REQUIRE((long long)g.N * g.C * g.H * g.Wd < (1ll << 31) && (long long)g.N * g.K * g.H * g.Wd < (1ll << 31),
               "conv_tile16_wgrad: tensor too large for 32-bit offsets");
    int NFT, NCT;
    cw16_tiles(g.K, g.C, NFT, NCT);
    const int PS = 4 / (NFT * NCT);
    const size_t n = (size_t)g.K * g.C * 9;
    int rc = tn_scratch_get(ctx, ((size_t)g.S * PS * n + (size_t)g.S * g.K) * sizeof(float), &g.ws);
    if (rc) return rc;
    g.dbws = g.ws + (size_t)g.S * PS * n;
    g.gscale = ctx->grad_scale; g.oscale = 1.f / ctx->grad_scale;
#define CW16_GO(A, B)                                                                            \
    rc = pool ? cw16_launch<A, B, true>(ctx, g) : cw16_launch<A, B, false>(ctx, g)
    if (NFT == 2 && NCT == 2) CW16_GO(2, 2);
    else if (NFT == 2) CW16_GO(2, 1);
    else if (NCT == 2) CW16_GO(1, 2);
    else CW16_GO(1, 1);
#undef CW16_GO
    if (rc) return rc;
    rc = tn_red_push(ctx, g.ws, dW, (uint32_t)n, (uint32_t)(g.S * PS), (uint32_t)n, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, g.dbws, db, (uint32_t)g.K, (uint32_t)g.S, (uint32_t)g.K, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

int tn_conv_tile16_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N, int C,
                         int H, int Wd, int K) {
    ConvWG16 g{};
    g.x = x; g.dz = dz;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    return cw16_run(ctx, g, dW, db, false);
}

// ---- conv + act + 2x2 max-pool blocks (3x3 'same', even maps), fp16 operands ------------------------
int tn_convpool_tile16_ok(int N, int C, int H, int Wd, int K, int f, int stride, int pad, int Ho, int Wo,
                          int p, int Hp, int Wp) {
    if (!c16_enabled() || f != 3 || stride != 1 || p != 2 || pad != 1 || Ho != H || Wo != Wd) return 0;
    if ((Ho & 1) || (Wo & 3) || Hp * 2 != Ho || Wp * 2 != Wo) return 0;
    ConvTG a{};                                   // forward
    a.N = N; a.C = C; a.H = H; a.Wd = Wd; a.K = K; a.pad = 1; a.Ho = Ho; a.Wo = Wo;
    if (!c16_geometry(a, c16_pick_ft(K)) || (a.TH & 1) || c16_lds_bytes(a, c16_pick_ft(K)) > 156 * 1024) return 0;
    ConvTG d{};                                   // input gradient: gathers dz (N,K,Ho,Wo)
    d.N = N; d.C = K; d.H = Ho; d.Wd = Wo; d.K = C; d.pad = 1; d.Ho = H; d.Wo = Wd;
    if (!c16_geometry(d, c16_pick_ft(C)) || c16_lds_bytes(d, c16_pick_ft(C)) > 156 * 1024) return 0;
    if (C * 9 > 32) {
        ConvWG16 w{};                             // weight gradient (first layers use the small-C kernel)
        w.N = N; w.C = C; w.H = H; w.Wd = Wd; w.K = K;
        if (!cw16_geometry(w, 256) || cw16_lds_bytes(w) > 160 * 1024) return 0;
    }
    return 1;
}

int tn_conv_tile16_pool_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y,
                            uint8_t* mask, int N, int C, int H, int Wd, int K, int act, float prm) {
    TN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "conv_tile16_pool_fwd: x must be 16-byte aligned");
    ConvTG g{};
    g.x = x; g.out = y; g.bias = b; g.mask_out = mask;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.pad = 1; g.Ho = H; g.Wo = Wd;
    g.act = act; g.prm = prm; g.iscale = 1.f; g.oscale = 1.f;
    return c16_run(ctx, g, W, false, true);
}

// dW, db (dW != NULL) and the input gradient (dx != NULL) of the fused block from the pooled gradient g_,
// the pooled output y and the pooling mask
// dz = MaxPoolGrad(g) * act'(y) of a pooled block as a tensor of its own (4 pixels of a row per thread)
__global__ __launch_bounds__(256) void pool_expand_kernel(PoolSrc ps, float* __restrict__ dz, int H, int q4,
                                                         size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % q4);
    const size_t pr = i / q4;
    const int row = (int)(pr % H), plane = (int)(pr / H);
    reinterpret_cast<float4*>(dz)[i] = pool_expand4(ps, plane, row, 4 * c4);
}

int tn_conv_tile16_pool_bwd(tn_ctx* ctx, const float* x, const float* W, const float* g_, const float* y,
                            const uint8_t* mask, float* dx, float* dW, float* db, int N, int C, int H, int Wd,
                            int K, int act, float prm, const float* prev_a, int prev_act, float prev_prm) {
    PoolSrc ps{};
    ps.g = g_; ps.y = y; ps.mask = mask; ps.Hp = H / 2; ps.Wp = Wd / 2; ps.act = act; ps.prm = prm;
    TN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g_) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(mask) & 3) == 0, "conv_tile16_pool_bwd: misaligned operand");
    // With fp16 operands the two products are bound by the traffic and the latency of their loads, and forming
    // dz inside them (three narrow loads and the mask arithmetic per 4 pixels, twice) costs more than writing
    // it once: one streaming launch materialises dz (a buffer that lives until the two launches behind it have
    // run), the plain kernels consume it.  TN_C16_UNPOOL=0: the fused forms.
    static int unpool = -1;
    if (unpool < 0) {
        const char* e = getenv("TN_C16_UNPOOL");
        unpool = e ? atoi(e) : 1;
    }
    if (unpool && (dW || dx)) {
        float* dz;
        const size_t total4 = (size_t)N * K * H * (Wd / 4);
        int rc = tn_tmp_get(ctx, total4 * 16, &dz);
        if (rc) return rc;
        pool_expand_kernel<<<(unsigned)cdiv(total4, 256), 256, 0, ctx->stream>>>(ps, dz, H, Wd / 4, total4);
        TN_LAUNCH_CHECK();
        if (dW) {
            rc = tn_conv_tile16_wgrad(ctx, x, dz, dW, db, N, C, H, Wd, K);
            if (rc) return rc;
        }
        if (dx) return tn_conv_tile16_dgrad(ctx, dz, W, dx, N, C, H, Wd, K, 1, H, Wd, prev_a, prev_act, prev_prm);
        return TN_OK;
    }
    if (dW) {
        ConvWG16 w{};
        w.x = x; w.ps = ps;
        w.N = N; w.C = C; w.H = H; w.Wd = Wd; w.K = K;
        int rc = cw16_run(ctx, w, dW, db, true);
        if (rc) return rc;
    }
    if (dx) {
        ConvTG d{};
        d.out = dx; d.prev_a = prev_a; d.ps = ps;
        d.N = N; d.C = K; d.H = H; d.Wd = Wd; d.K = C; d.pad = 1; d.Ho = H; d.Wo = Wd;
        d.act = prev_act; d.prm = prev_prm; d.iscale = ctx->grad_scale; d.oscale = 1.f / ctx->grad_scale;
        return c16_run(ctx, d, W, true, true);
    }
    return TN_OK;
}
