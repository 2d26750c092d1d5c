// 3x3 stride-1 convolution on the fp32 matrix cores with the input tile RESIDENT IN LDS
// (the wide layers of cifar_like / wide6; theanet/layer/convpool.py:54-72).
//
// conv_mfma.hip builds the im2col operand element by element (one clamped gather + a select per
// value, integer divisions per K-tile); here the im2col never exists, not even as a gather:
//   * a block owns 256 output pixels (TH full rows of one image, or NI whole small images) and
//     32*FT filters; per chunk of 8 input channels it copies the (TH+2) x (W+2) halo tile into LDS
//     with 16-byte coalesced loads (zero padding = cells that are cleared once and never written);
//   * the reduction runs tap-major: for a fixed tap (u,v) one v_mfma_f32_32x32x2_f32 consumes two
//     channels, A = W[filter][c+hi][tap] (pre-arranged by a tiny kernel so a chunk's weights are one
//     straight copy into LDS), B = tile[c+hi][row+u][col+v] -- a plain ds_read_b32 of 32
//     consecutive pixels at (lane base + constant): the "im2col" is an LDS address offset;
//   * a wave keeps FT x 2 accumulators (two pixel tiles share every A value, the FT filter tiles
//     share every B value), so 4 MFMAs are fed by 2 + FT ds_read_b32.
// dgrad is the same kernel on dz with W[k][c][u][v] read as (filter = c, channel = k, no flip) and
// padding 2 - pad, with act'(prev_a) of the layer below in the epilogue.
#include "common.h"

#include "conv_tile_common.h"

// wt[kt][chunk][cp][tap][hi][j]: filter kt*KBF + j, channel chunk*8 + 2*cp + hi, correlation tap (u,v)
__global__ __launch_bounds__(256) void conv_tile_wt_kernel(const float* __restrict__ W, float* __restrict__ wt,
                                                          int K, int C, int KBF, int nchunk, int total,
                                                          int dgrad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int r = idx;
    const int j = r % KBF; r /= KBF;
    const int hi = r & 1; r >>= 1;
    const int tap = r % 9; r /= 9;
    const int cp = r & 3; r >>= 2;
    const int chunk = r % nchunk;
    const int kt = r / nchunk;
    const int filt = kt * KBF + j, ch = chunk * CT_CH + 2 * cp + hi;
    float v = 0.f;
    if (filt < K && ch < C)
        v = dgrad ? W[((size_t)ch * K + filt) * 9 + tap]          // W[k = ch][c = filt][u][v]
                  : W[((size_t)filt * C + ch) * 9 + (8 - tap)];   // true convolution: flipped taps
    wt[idx] = v;
}

struct CtSlot { int g, l, c, n, row, col; bool ok; };
// input staging slot s of thread t: (channel-in-chunk, image, tile row, 16-byte column group)
__device__ __forceinline__ CtSlot ct_slot(const ConvTG& g, int t, int s, int n0, int r0) {
    CtSlot o;
    const int e = t + 256 * s;
    bool ok = e < g.nx4;
    int rr = min(e, g.nx4 - 1);
    const int q = rr % g.q4; rr /= g.q4;
    const int r = rr % g.THi; rr /= g.THi;
    const int ni = rr % g.NI;
    o.c = rr / g.NI;
    const int in_row = r0 - g.pad + r, n = n0 + ni;
    o.ok = ok && (unsigned)in_row < (unsigned)g.H && n < g.N;
    o.n = min(n, g.N - 1); o.row = min(max(in_row, 0), g.H - 1); o.col = 4 * q;
    o.g = (o.n * g.C * g.H + o.row) * g.Wd + 4 * q;
    o.l = o.c * g.plane + (ni * g.THi + r) * g.RS + g.pad + g.LP + 4 * q;
    return o;
}

// POOL: forward -> bias + act + 2x2 max-pool + pooling mask in the epilogue (the conv activation never
// reaches HBM); dgrad -> the gathered tensor dz is expanded from the pooled gradient while it is staged.
// NCP: channel pairs of a chunk that can be non-zero (4; 2 for nets' first layers with C <= 4, which
// fit one chunk: the steps of the all-zero pairs are not issued).
template <int FT, bool DGRAD, bool POOL, int NCP = 4>
__global__ __launch_bounds__(256) void conv_tile_kernel(ConvTG g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr int KBF = 32 * FT;
    constexpr int WSZ = 4 * 9 * 2 * KBF;              // floats of one weight chunk
    constexpr int WS4 = (WSZ / 4 + 255) / 256;        // 16-byte staging slots per thread (3 or 5)
    const int XSZ = CT_CH * g.plane;
    float* Xs = ct_smem;                              // [2][XSZ]
    float* Ws = ct_smem + 2 * XSZ;                    // [2][WSZ]
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

    for (int i = t * 4; i < 2 * XSZ; i += 1024) *reinterpret_cast<float4*>(Xs + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging slots live in named registers (arrays of float4 ended up in scratch memory)
    const CtSlot s0 = ct_slot(g, t, 0, n0, r0), s1 = ct_slot(g, t, 1, n0, r0), s2 = ct_slot(g, t, 2, n0, r0),
                 s3 = ct_slot(g, t, 3, n0, r0);
    const float* wsrc = g.wt + (size_t)kt * g.nchunk * WSZ + 4 * t;
    const int wo3 = min(3072, WSZ - 4 - 4 * t), wo4 = min(4096, WSZ - 4 - 4 * t), wo2 = min(2048, WSZ - 4 - 4 * t);

    // this lane's two pixels (B operand)
    int pixb[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int p = wave * 64 + pt * 32 + l31;
        const int pp = p < g.TP ? p : 0;
        const int per = g.TH * g.Wo;
        const int ni = pp / per, rem = pp - ni * per;
        const int r = rem / g.Wo, col = rem - r * g.Wo;
        pixb[pt] = hi * g.plane + (ni * g.THi + r) * g.RS + col + g.LP;
    }

    f32x16 acc[FT][2];
#pragma unroll
    for (int a = 0; a < FT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 xr0, xr1, xr2, xr3, wr0, wr1, wr2, wr3, wr4;
#define CT_XL(S, R)                                                                              \
    {                                                                                            \
        const int cc_ = min(ch_ * CT_CH + S.c, g.C - 1);                                         \
        if (DGRAD && POOL) R = pool_expand4(g.ps, S.n * g.C + cc_, S.row, S.col);                \
        else R = *reinterpret_cast<const float4*>(g.x + S.g + cc_ * HW);                         \
    }
#define CT_GLOAD(CHUNK)                                                                          \
    {                                                                                            \
        const int ch_ = min((CHUNK), g.nchunk - 1);                                              \
        CT_XL(s0, xr0); CT_XL(s1, xr1); CT_XL(s2, xr2); CT_XL(s3, xr3);                          \
        const float* w_ = wsrc + (size_t)ch_ * WSZ;                                              \
        wr0 = *reinterpret_cast<const float4*>(w_);                                              \
        wr1 = *reinterpret_cast<const float4*>(w_ + 1024);                                       \
        wr2 = *reinterpret_cast<const float4*>(w_ + wo2);                                        \
        if (WS4 > 3) {                                                                           \
            wr3 = *reinterpret_cast<const float4*>(w_ + wo3);                                    \
            wr4 = *reinterpret_cast<const float4*>(w_ + wo4);                                    \
        }                                                                                        \
    }
#define CT_XS(S, R) if (S.ok) *reinterpret_cast<float4*>(xb_ + S.l) = R
#define CT_LSTORE(BUF)                                                                           \
    {                                                                                            \
        float* xb_ = Xs + (BUF) * XSZ;                                                           \
        float* wb_ = Ws + (BUF) * WSZ + 4 * t;                                                   \
        CT_XS(s0, xr0); CT_XS(s1, xr1); CT_XS(s2, xr2); CT_XS(s3, xr3);                          \
        *reinterpret_cast<float4*>(wb_) = wr0;                                                   \
        *reinterpret_cast<float4*>(wb_ + 1024) = wr1;                                            \
        if (4 * t + 2048 < WSZ) *reinterpret_cast<float4*>(wb_ + 2048) = wr2;                    \
        if (WS4 > 3) {                                                                           \
            *reinterpret_cast<float4*>(wb_ + 3072) = wr3;                                        \
            if (4 * t + 4096 < WSZ) *reinterpret_cast<float4*>(wb_ + 4096) = wr4;                \
        }                                                                                        \
    }
    CT_GLOAD(0);
    __syncthreads();                 // the clearing is done
    CT_LSTORE(0);
    __syncthreads();
    if (dbg) dbg[1] = __builtin_readcyclecounter();
    const int RS = g.RS, plane2 = 2 * g.plane;
    for (int chunk = 0; chunk < g.nchunk; ++chunk) {
        CT_GLOAD(chunk + 1);
        const float* x0 = Xs + (chunk & 1) * XSZ + pixb[0];
        const float* x1 = Xs + (chunk & 1) * XSZ + pixb[1];
        const float* Wb = Ws + (chunk & 1) * WSZ + hi * KBF + l31;
        // 12 steps (channel pair, tap row): the LDS operands of step s+1 (3 taps: 3 A pairs, 2 x 3 B
        // values) are requested before the 6*FT MFMAs of step s are issued
        float a[2][3][FT], b[2][3][2];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
#pragma unroll
            for (int f = 0; f < FT; ++f) a[0][v][f] = Wb[(v * 2) * KBF + 32 * f];
            b[0][v][0] = x0[v]; b[0][v][1] = x1[v];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);            // step 0's operands
#pragma unroll
        for (int st = 0; st < 3 * NCP; ++st) {
            const int cur = st & 1, nx = cur ^ 1;
            if (st + 1 < 3 * NCP) {
                const int cp = (st + 1) / 3, u = (st + 1) % 3;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
#pragma unroll
                    for (int f = 0; f < FT; ++f) a[nx][v][f] = Wb[((cp * 9 + u * 3 + v) * 2) * KBF + 32 * f];
                    b[nx][v][0] = x0[cp * plane2 + u * RS + v];
                    b[nx][v][1] = x1[cp * plane2 + u * RS + v];
                }
            }
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][v][f], b[cur][v][0], acc[f][0], 0, 0, 0);
                    acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][v][f], b[cur][v][1], acc[f][1], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);        // DS reads of the next step
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * FT, 0);   // then this step's MFMAs
        }
        if (chunk + 1 < g.nchunk) CT_LSTORE((chunk + 1) & 1);
        __syncthreads();
    }
#undef CT_GLOAD
#undef CT_LSTORE
#undef CT_XL
#undef CT_XS
    if (dbg) dbg[2] = __builtin_readcyclecounter();

    ct_epilogue<FT, DGRAD, POOL>(g, acc, ct_smem, kt, n0, r0, lane, wave, l31, hi);
    if (dbg) { dbg[3] = __builtin_readcyclecounter(); dbg[5] = wall_clock64(); }
}

// geometry of the pixel tiling; returns 0 when the shape is outside the kernel's limits
static int ct_geometry(ConvTG& g, int FT) {
    if (g.Wo > 256 || (g.Wd & 3)) return 0;
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
    g.LP = (4 - (g.pad & 3)) & 3;
    g.RS = (g.LP + g.Wo + 2 + 3) & ~3;
    if (g.RS < g.pad + g.LP + g.Wd) g.RS = (g.pad + g.LP + g.Wd + 3) & ~3;
    g.plane = g.NI * g.THi * g.RS;
    g.q4 = g.Wd / 4;
    g.nx4 = CT_CH * g.NI * g.THi * g.q4;
    if (g.nx4 > CT_SX * 256) return 0;
    g.nchunk = cdiv(g.C, CT_CH);
    g.KT = cdiv(g.K, 32 * FT);
    g.MT = cdiv(g.N, g.NI) * g.RT;
    return 1;
}

static size_t ct_lds_bytes(const ConvTG& g, int FT) {
    const size_t loop = (size_t)(2 * CT_CH * g.plane + 2 * 4 * 9 * 2 * 32 * FT) * sizeof(float);
    const size_t epi = (size_t)32 * FT * 256 * sizeof(float);        // the transposed output tile
    return loop > epi ? loop : epi;
}

static int ct_pick_ft(int K) { return K > 32 ? 2 : 1; }

static bool ct_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("TN_CONV_TILE");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

// 1 if conv_tile_kernel handles the (gathered tensor N,C,H,Wd; K filters; pad; output Ho,Wo) problem
int tn_conv_tile_ok(const float* x, int N, int C, int H, int Wd, int K, int f, int pad, int Ho, int Wo) {
    if (!ct_enabled() || f != 3 || pad < 0 || pad > 2) return 0;
    if (reinterpret_cast<uintptr_t>(x) & 15) return 0;
    ConvTG g{};
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.pad = pad; g.Ho = Ho; g.Wo = Wo;
    const int FT = ct_pick_ft(K);
    if (!ct_geometry(g, FT)) return 0;
    return ct_lds_bytes(g, FT) <= 150 * 1024;
}

static unsigned long long* ct_dbg_buf = nullptr;
// debugging aid (not part of the C-ABI): copies the cycle stamps of the last stamped launch
extern "C" int tn_conv_tile_dbg_read(tn_ctx* ctx, unsigned long long* host, int nblocks) {
    if (!ct_dbg_buf) return -1;
    hipStreamSynchronize(ctx->stream);
    return hipMemcpy(host, ct_dbg_buf, (size_t)nblocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

template <int FT, bool DGRAD, bool POOL, int NCP = 4>
static int ct_launch(tn_ctx* ctx, ConvTG& g) {
    static bool attr_set = false;
    size_t lds = ct_lds_bytes(g, FT);
    if (!attr_set) {
        TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile_kernel<FT, DGRAD, POOL, NCP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int grid = 8 * cdiv(g.MT, 8) * g.KT;
    static unsigned long long* dbgbuf = nullptr;
    if (getenv("TN_CT_DBG")) {
        if (!dbgbuf) TN_HIP(hipMalloc(&dbgbuf, 8 * sizeof(unsigned long long) * 65536));
        ct_dbg_buf = dbgbuf;
        g.dbg = grid <= 65536 ? dbgbuf : nullptr;
    }
    conv_tile_kernel<FT, DGRAD, POOL, NCP><<<grid, 256, lds, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static int ct_run(tn_ctx* ctx, ConvTG& g, const float* W, bool dgrad, bool pool = false) {
    const int FT = ct_pick_ft(g.K);
    TN_REQUIRE(ct_geometry(g, FT), "conv_tile: unsupported shape");
    TN_REQUIRE((long long)g.N * g.C * g.H * g.Wd < (1ll << 31) && (long long)g.N * g.K * g.Ho * g.Wo < (1ll << 31),
               "conv_tile: tensor too large for 32-bit offsets");
    const int KBF = 32 * FT, total = g.KT * g.nchunk * 4 * 9 * 2 * KBF;
    float* wt;
    int rc = tn_scratch_get(ctx, (size_t)total * sizeof(float), &wt);
    if (rc) return rc;
    conv_tile_wt_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(W, wt, g.K, g.C, KBF, g.nchunk, total,
                                                                  dgrad ? 1 : 0);
    TN_LAUNCH_CHECK();
    g.wt = wt;
    g.vec_out = (g.Wo % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.out) | reinterpret_cast<uintptr_t>(g.prev_a)) & 15) == 0;
    if (!dgrad && g.C <= 4) {        // first layers: one chunk, two channel pairs
        if (pool) return FT == 2 ? ct_launch<2, false, true, 2>(ctx, g) : ct_launch<1, false, true, 2>(ctx, g);
        return FT == 2 ? ct_launch<2, false, false, 2>(ctx, g) : ct_launch<1, false, false, 2>(ctx, g);
    }
    if (pool) {
        if (dgrad) return FT == 2 ? ct_launch<2, true, true>(ctx, g) : ct_launch<1, true, true>(ctx, g);
        return FT == 2 ? ct_launch<2, false, true>(ctx, g) : ct_launch<1, false, true>(ctx, g);
    }
    if (dgrad) return FT == 2 ? ct_launch<2, true, false>(ctx, g) : ct_launch<1, true, false>(ctx, g);
    return FT == 2 ? ct_launch<2, false, false>(ctx, g) : ct_launch<1, false, false>(ctx, g);
}

int tn_conv_tile_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int N, int C,
                     int H, int Wd, int K, int pad, int Ho, int Wo, int act, float prm) {
    ConvTG g{};
    g.x = x; g.out = a; g.bias = b;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.pad = pad; g.Ho = Ho; g.Wo = Wo;
    g.act = act; g.prm = prm;
    return ct_run(ctx, g, W, false);
}

// dx (N,C,H,Wd) from dz (N,K,Ho,Wo): the forward kernel with (channels, filters) = (K, C), padding 2 - pad
int tn_conv_tile_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx, int N, int C, int H,
                       int Wd, int K, int pad, int Ho, int Wo, const float* prev_a, int act, float prm) {
    ConvTG g{};
    g.x = dz; g.out = dx; g.prev_a = prev_a;
    g.N = N; g.C = K; g.H = Ho; g.Wd = Wo; g.K = C; g.pad = 2 - pad; g.Ho = H; g.Wo = Wd;
    g.act = act; g.prm = prm;
    return ct_run(ctx, g, W, true);
}

// =================================================================================================
// Weight gradient of a 3x3 'same' convolution (CorrMM_gradWeights of convpool.py:54-56) with both
// operands resident in LDS:  dW[k][c][2-u][2-v] = sum_{n,i,j} dz[n,k,i,j] * x[n,c,i-1+u,j-1+v].
//   * GEMM rows = 32 filters, columns = 32 input channels at a FIXED tap, reduction = pixels: one
//     v_mfma_f32_32x32x2_f32 reduces two pixels, A = dz[filter][pixel], B = x[channel][pixel + tap].
//     A wave owns one (filter tile, channel tile) pair and keeps all nine taps: 9 accumulators.
//   * per step of 8 pixels a lane reads ONE 16-byte dz vector and, per tap row, 6 consecutive x
//     values (4-byte + 16-byte + 4-byte, all aligned): 10 LDS reads feed 36 MFMAs, the nine taps
//     are register selections of those 6 values.  Channel planes / dz rows are 4*odd floats apart:
//     conflict-free 16-byte reads.
//   * a block = 32*NFT filters x 32 channels x one slab of images, ONE wave per SIMD (LDS holds two
//     128-pixel tiles of both operands); the 4/NFT waves that share a filter tile take alternate
//     8-pixel steps and write separate slabs.  The next tile is copied global -> registers -> LDS
//     two steps behind its loads, in between the MFMAs; one barrier per tile.
// =================================================================================================
#define CW_DZS 132

struct ConvWG {
    const float* x;        // (N, C, H, Wd)
    const float* dz;       // (N, K, H, Wd)
    float* ws;             // [S * PS][K*C*9] partial weight gradients, dW layout
    float* dbws;           // [S][K] partial bias gradients
    int N, C, H, Wd, K;
    int KG, CG, S, ipb;    // filter groups, channel groups, image slabs, images per slab
    int NI, TH, THi, RT, NT;
    int RS, plane, q4, nx4, lgW, lgP;
    PoolSrc ps;            // POOL: dz is formed from (g, mask, y) while it is staged
};

// v where ok, +0 elsewhere -- as bit masks, so that the compiler cannot turn it into a branch
__device__ __forceinline__ float4 cw_mask4(float4 v, bool ok) {
    const int m = ok ? -1 : 0;
    return make_float4(__int_as_float(__float_as_int(v.x) & m), __int_as_float(__float_as_int(v.y) & m),
                       __int_as_float(__float_as_int(v.z) & m), __int_as_float(__float_as_int(v.w) & m));
}
__device__ __forceinline__ const float4* cw_f4(const float* p) {
    return reinterpret_cast<const float4*>(__builtin_assume_aligned(p, 16));
}

template <int NFT, bool POOL>
__global__ __launch_bounds__(256) void conv_tile_wgrad_kernel(ConvWG g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr int KBF = 32 * NFT, PS = 4 / NFT, SPW = 16 / PS;
    constexpr int NDZ = NFT * 4, NSL = NDZ + 8, SPS = (NSL + SPW - 1) / SPW;
    constexpr int DZSZ = KBF * CW_DZS;
    const int XSZ = 32 * g.plane, BUFSZ = DZSZ + XSZ;
    const int bid = blockIdx.x, per = g.KG * g.CG;
    const int z = ((bid >> 3) / per) * 8 + (bid & 7), rem = (bid >> 3) % per;
    if (z >= g.S) return;
    const int kg = rem / g.CG, cg = rem - kg * g.CG;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int ft = wave % NFT, ps = wave / NFT;
    const int n_beg = z * g.ipb, n_end = min(g.N, n_beg + g.ipb);
    const int HW = g.H * g.Wd, Wm = g.Wd - 1, THm = g.TH - 1;

    for (int i = t * 4; i < 2 * BUFSZ; i += 1024) *reinterpret_cast<float4*>(ct_smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    float dbacc[NDZ];
#pragma unroll
    for (int s = 0; s < NDZ; ++s) dbacc[s] = 0.f;

    // staging slot SL of the tile (first image n0, first row r0): issue the 16-byte load, keep the
    // value and its LDS offset (-1: nothing to write) for the store a couple of steps later
    float4 sv[NSL];
    int so[NSL];
#define CW_SLOAD(SL, N0, R0)                                                                     \
    {                                                                                            \
        if ((SL) < NDZ) {                                                                        \
            const int e_ = t + 256 * (SL);                                                       \
            const int q_ = e_ & 31, f_ = e_ >> 5, p_ = 4 * q_;                                   \
            const int n_ = (N0) + (p_ >> g.lgP), row_ = (R0) + ((p_ >> g.lgW) & THm), k_ = kg * KBF + f_; \
            const bool ok_ = n_ < n_end && k_ < g.K;                                             \
            const int go_ = ((min(n_, g.N - 1) * g.K + min(k_, g.K - 1)) * g.H + row_) * g.Wd + (p_ & Wm); \
            if (POOL)                                                                            \
                sv[SL] = cw_mask4(pool_expand4(g.ps, min(n_, g.N - 1) * g.K + min(k_, g.K - 1), row_, p_ & Wm), ok_); \
            else                                                                                 \
                sv[SL] = cw_mask4(*reinterpret_cast<const float4*>(g.dz + go_), ok_);            \
            so[SL] = f_ * CW_DZS + p_;                                                           \
        } else {                                                                                 \
            const int e_ = t + 256 * ((SL) - NDZ);                                               \
            int rr_ = min(e_, g.nx4 - 1);                                                        \
            const int q_ = rr_ % g.q4; rr_ /= g.q4;                                              \
            const int r_ = rr_ % g.THi; rr_ /= g.THi;                                            \
            const int ni_ = rr_ % g.NI, c_ = rr_ / g.NI;                                         \
            const int n_ = (N0) + ni_, row_ = (R0) - 1 + r_, cc_ = cg * 32 + c_;                 \
            const bool ok_ = n_ < n_end && (unsigned)row_ < (unsigned)g.H && cc_ < g.C;          \
            const int go_ = ((min(n_, g.N - 1) * g.C + min(cc_, g.C - 1)) * g.H + min(max(row_, 0), g.H - 1)) * g.Wd + 4 * q_; \
            sv[SL] = cw_mask4(*reinterpret_cast<const float4*>(g.x + go_), ok_);                 \
            so[SL] = e_ < g.nx4 ? DZSZ + c_ * g.plane + (ni_ * g.THi + r_) * g.RS + 4 + 4 * q_ : -1; \
        }                                                                                        \
    }
#define CW_SSTORE(SL, BUF)                                                                       \
    {                                                                                            \
        if (so[SL] >= 0)                                                                         \
            *reinterpret_cast<float4*>(__builtin_assume_aligned(ct_smem + (BUF) * BUFSZ + so[SL], 16)) = sv[SL]; \
        if ((SL) < NDZ) dbacc[(SL) < NDZ ? (SL) : 0] += dbw_ * ((sv[SL].x + sv[SL].y) + (sv[SL].z + sv[SL].w)); \
    }

    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // tile 0
    float dbw_ = 1.f;
#pragma unroll
    for (int s = 0; s < NSL; ++s) CW_SLOAD(s, n_beg, 0);
    __syncthreads();                 // the clearing is done
#pragma unroll
    for (int s = 0; s < NSL; ++s) CW_SSTORE(s, 0);
    __syncthreads();

    int gi = 0, rt = 0;              // tile = (image group, row tile) inside the slab
    for (int tl = 0; tl < g.NT; ++tl) {
        const int cur = tl & 1;
        int rt1 = rt + 1, gi1 = gi;
        if (rt1 == g.RT) { rt1 = 0; ++gi1; }
        // the last tile re-stages itself into the idle buffer (keeps the loop body branch-free)
        const bool hasnext = tl + 1 < g.NT;
        if (!hasnext) { rt1 = rt; gi1 = gi; }
        dbw_ = hasnext ? 1.f : 0.f;
        const int n1 = n_beg + gi1 * g.NI, r1 = rt1 * g.TH;
        const float* dzb = ct_smem + cur * BUFSZ + (ft * 32 + l31) * CW_DZS + 4 * hi;
        const float* xb = ct_smem + cur * BUFSZ + DZSZ + l31 * g.plane + 3;
        float4 av[2], xm[2][3];
        float xl[2][3], xr[2][3];
#define CW_OPS(SLOT, SG)                                                                         \
        {                                                                                        \
            const int p_ = 8 * (SG) + 4 * hi;                                                    \
            av[SLOT] = *cw_f4(dzb + 8 * (SG));                                                   \
            const float* xp_ = xb + ((p_ >> g.lgP) * g.THi + ((p_ >> g.lgW) & THm)) * g.RS + (p_ & Wm); \
            _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                      \
                xl[SLOT][u] = xp_[u * g.RS];                                                     \
                xm[SLOT][u] = *cw_f4(xp_ + u * g.RS + 1);                                        \
                xr[SLOT][u] = xp_[u * g.RS + 5];                                                 \
            }                                                                                    \
        }
        CW_OPS(0, ps);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int c_ = i & 1, nx_ = c_ ^ 1;
            if (i >= 2) {
#pragma unroll
                for (int s = (i - 2) * SPS; s < (i - 1) * SPS && s < NSL; ++s) CW_SSTORE(s, cur ^ 1);
            }
#pragma unroll
            for (int s = i * SPS; s < (i + 1) * SPS && s < NSL; ++s) CW_SLOAD(s, n1, r1);
            if (i + 1 < SPW) CW_OPS(nx_, ps + PS * (i + 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a_ = j == 0 ? av[c_].x : j == 1 ? av[c_].y : j == 2 ? av[c_].z : av[c_].w;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const float x6[6] = {xl[c_][u], xm[c_][u].x, xm[c_][u].y, xm[c_][u].z, xm[c_][u].w, xr[c_][u]};
#pragma unroll
                    for (int v = 0; v < 3; ++v)
                        acc[u * 3 + v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, x6[j + v], acc[u * 3 + v], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);       // LDS operands of the next step
            __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);       // then this step's MFMAs
        }
#pragma unroll
        for (int s = (SPW - 2) * SPS; s < NSL; ++s) CW_SSTORE(s, cur ^ 1);
        __syncthreads();
        rt = rt1; gi = gi1;
    }
#undef CW_OPS
#undef CW_SLOAD
#undef CW_SSTORE

    // bias gradient partial of the slab: per-filter sums of the dz this block staged
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
    const int c = cg * 32 + l31;
    if (c < g.C) {
        float* wz = g.ws + (size_t)(z * PS + ps) * g.K * g.C * 9;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kg * KBF + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) {
#pragma unroll
                for (int a = 0; a < 9; ++a) wz[((size_t)k * g.C + c) * 9 + 8 - a] = acc[a][r];
            }
        }
    }
}

static int cw_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

static int cw_geometry(ConvWG& g, int num_cus) {
    const int lgW = cw_log2(g.Wd);
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
    if (cw_log2(TH) < 0) return 0;
    g.TH = TH; g.THi = TH + 2; g.RT = g.H / TH;
    g.lgW = lgW; g.lgP = cw_log2(TH * g.Wd);
    g.RS = g.Wd + 8;
    g.plane = g.NI * g.THi * g.RS;
    if (((g.plane >> 2) & 1) == 0) g.plane += 4;       // 4 * odd: conflict-free 16-byte column reads
    g.q4 = g.Wd / 4;
    g.nx4 = 32 * g.NI * g.THi * g.q4;
    if (g.nx4 > 8 * 256) return 0;
    const int NFT = g.K > 32 ? 2 : 1;
    g.KG = cdiv(g.K, 32 * NFT);
    g.CG = cdiv(g.C, 32);
    int S = num_cus / (g.KG * g.CG);
    const int groups = cdiv(g.N, g.NI);
    if (S > groups) S = groups;
    if (S < 1) S = 1;
    g.ipb = cdiv(groups, S) * g.NI;
    g.S = cdiv(g.N, g.ipb);
    g.NT = (g.ipb / g.NI) * g.RT;
    return 1;
}

static size_t cw_lds_bytes(const ConvWG& g) {
    const int NFT = g.K > 32 ? 2 : 1;
    return (size_t)2 * (32 * NFT * CW_DZS + 32 * g.plane) * sizeof(float);
}

int tn_conv_tile_wgrad_ok(tn_ctx* ctx, const float* x, const float* dz, int N, int C, int H, int Wd, int K,
                          int f, int pad, int Ho, int Wo) {
    if (!ct_enabled() || f != 3 || pad != 1 || Ho != H || Wo != Wd) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz)) & 15) return 0;
    if (const char* e = getenv("TN_CONV_TILE_WGRAD")) if (e[0] == '0') return 0;
    ConvWG g{};
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    if (!cw_geometry(g, ctx->num_cus)) return 0;
    return cw_lds_bytes(g) <= 160 * 1024;
}

template <int NFT, bool POOL>
static int cw_launch(tn_ctx* ctx, ConvWG& g) {
    static bool attr_set = false;
    if (!attr_set) {
        TN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile_wgrad_kernel<NFT, POOL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int grid = 8 * cdiv(g.S, 8) * g.KG * g.CG;
    conv_tile_wgrad_kernel<NFT, POOL><<<grid, 256, cw_lds_bytes(g), ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_red_push(tn_ctx* ctx, const float* src, float* out, uint32_t n, uint32_t S, uint32_t stride, uint32_t flip);
int tn_red_commit(tn_ctx* ctx);

static int cw_run(tn_ctx* ctx, ConvWG& g, float* dW, float* db, bool pool) {
    TN_REQUIRE(cw_geometry(g, ctx->num_cus), "conv_tile_wgrad: unsupported shape");
    TN_REQUIRE((long long)g.N * g.C * g.H * g.Wd < (1ll << 31) && (long long)g.N * g.K * g.H * g.Wd < (1ll << 31),
               "conv_tile_wgrad: tensor too large for 32-bit offsets");
    const int NFT = g.K > 32 ? 2 : 1, PS = 4 / NFT;
    const size_t n = (size_t)g.K * g.C * 9;
    int rc = tn_scratch_get(ctx, ((size_t)g.S * PS * n + (size_t)g.S * g.K) * sizeof(float), &g.ws);
    if (rc) return rc;
    g.dbws = g.ws + (size_t)g.S * PS * n;
    if (pool) rc = NFT == 2 ? cw_launch<2, true>(ctx, g) : cw_launch<1, true>(ctx, g);
    else rc = NFT == 2 ? cw_launch<2, false>(ctx, g) : cw_launch<1, false>(ctx, g);
    if (rc) return rc;
    rc = tn_red_push(ctx, g.ws, dW, (uint32_t)n, (uint32_t)(g.S * PS), (uint32_t)n, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, g.dbws, db, (uint32_t)g.K, (uint32_t)g.S, (uint32_t)g.K, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

int tn_conv_tile_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N, int C,
                       int H, int Wd, int K) {
    ConvWG g{};
    g.x = x; g.dz = dz;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    return cw_run(ctx, g, dW, db, false);
}

// ---- conv + act + 2x2 max-pool blocks on the tile kernels (3x3 'same', even maps) ------------------
extern "C" {

// 1 if the block (N,C,H,Wd) -> K maps (Ho,Wo) -> pooled (Hp,Wp) runs fused on the tile kernels: the
// forward pools in its epilogue and records the pooling mask, the backward forms dz from that mask.
int tn_convpool_tile_supported(int N, int C, int H, int Wd, int K, int f, int stride, int pad, int Ho, int Wo,
                               int p, int Hp, int Wp) {
    if (!ct_enabled() || f != 3 || stride != 1 || p != 2 || pad != 1 || Ho != H || Wo != Wd) return 0;
    if ((Ho & 1) || (Wo & 3) || Hp * 2 != Ho || Wp * 2 != Wo) return 0;
    if (C * 9 < 32 || K < 16) return 0;
    if (const char* e = getenv("TN_CONV_TILE_POOL")) if (e[0] == '0') return 0;
    ConvTG a{};                                   // forward
    a.N = N; a.C = C; a.H = H; a.Wd = Wd; a.K = K; a.pad = 1; a.Ho = Ho; a.Wo = Wo;
    if (!ct_geometry(a, ct_pick_ft(K)) || (a.TH & 1) || ct_lds_bytes(a, ct_pick_ft(K)) > 150 * 1024) return 0;
    ConvTG d{};                                   // input gradient: gathers dz (N,K,Ho,Wo)
    d.N = N; d.C = K; d.H = Ho; d.Wd = Wo; d.K = C; d.pad = 1; d.Ho = H; d.Wo = Wd;
    if (C < 16 || !ct_geometry(d, ct_pick_ft(C)) || ct_lds_bytes(d, ct_pick_ft(C)) > 150 * 1024) return 0;
    ConvWG w{};                                   // weight gradient
    w.N = N; w.C = C; w.H = H; w.Wd = Wd; w.K = K;
    if (!cw_geometry(w, 256) || cw_lds_bytes(w) > 160 * 1024) return 0;
    return 1;
}

}  // extern "C"

int tn_conv_tile_pool_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y,
                          uint8_t* mask, int N, int C, int H, int Wd, int K, int act, float prm) {
    TN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "conv_tile_pool_fwd: x must be 16-byte aligned");
    ConvTG g{};
    g.x = x; g.out = y; g.bias = b; g.mask_out = mask;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K; g.pad = 1; g.Ho = H; g.Wo = Wd;
    g.act = act; g.prm = prm;
    return ct_run(ctx, g, W, false, true);
}

// dW, db and (dx != NULL) the input gradient of the fused block from the pooled gradient g, the pooled
// output y and the pooling mask; prev_a: output of the layer below, whose activation gradient is
// applied to dx in the epilogue (NULL: none).
int tn_conv_tile_pool_bwd(tn_ctx* ctx, const float* x, const float* W, const float* g_, const float* y,
                          const uint8_t* mask, float* dx, float* dW, float* db, int N, int C, int H, int Wd,
                          int K, int act, float prm, const float* prev_a, int prev_act, float prev_prm) {
    PoolSrc ps{};
    ps.g = g_; ps.y = y; ps.mask = mask; ps.Hp = H / 2; ps.Wp = Wd / 2; ps.act = act; ps.prm = prm;
    TN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g_) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(mask) & 3) == 0, "conv_tile_pool_bwd: misaligned operand");
    if (dW) {
        ConvWG w{};
        w.x = x; w.ps = ps;
        w.N = N; w.C = C; w.H = H; w.Wd = Wd; w.K = K;
        int rc = cw_run(ctx, w, dW, db, true);
        if (rc) return rc;
    }
    if (dx) {
        ConvTG d{};
        d.out = dx; d.prev_a = prev_a; d.ps = ps;
        d.N = N; d.C = K; d.H = H; d.Wd = Wd; d.K = C; d.pad = 1; d.Ho = H; d.Wo = Wd;
        d.act = prev_act; d.prm = prev_prm;
        return ct_run(ctx, d, W, true, true);
    }
    return TN_OK;
}

// =================================================================================================
// Weight gradient of a fused 3x3 'same' conv + act + 2x2 max-pool block with FEW input channels
// (C*9 <= 32: the first layer of cifar_like, C = 3), from the pooling mask.  The channel tile of the
// kernel above would be 3/32 full; here the 32 GEMM columns are the (channel, tap) pairs themselves:
//   dWf[k][(c,u,v)] = sum_pix dz[k][pix] * x[c][pix + (u-1, v-1)],
// lane n = (c,u,v) reads its own shifted x value (4 consecutive pixels = two ds_read2_b32 at a
// per-lane constant offset), A = one 16-byte dz vector as above, one accumulator per wave.  The four
// waves of a block take every fourth 8-pixel step of a 128-pixel tile; tiles are double-buffered in
// LDS (20 KB per buffer, several blocks per CU).  dz = mask bit ? g*act'(y) : 0 is formed while staging.
// =================================================================================================
struct ConvSG {
    const float* x;
    const float* dz;       // POOL == false: the plain dz tensor (N, K, H, Wd)
    float* ws;             // [S*4][K*C*9]
    float* dbws;           // [S][K]
    int N, C, H, Wd, K;
    int KG, S, ipb;
    int NI, TH, THi, RT, NT;
    int RS, plane, q4, nx4, lgW, lgP;
    PoolSrc ps;
};

template <bool POOL>
__global__ __launch_bounds__(256) void conv_tile_wgrad_smallc_kernel(ConvSG g) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    constexpr int DZSZ = 32 * CW_DZS;
    const int XSZ = (g.C * g.plane + 3) & ~3, BUFSZ = DZSZ + XSZ;      // + one zero cell region below
    const int bid = blockIdx.x;
    const int z = ((bid >> 3) / g.KG) * 8 + (bid & 7), kg = (bid >> 3) % g.KG;
    if (z >= g.S) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int n_beg = z * g.ipb, n_end = min(g.N, n_beg + g.ipb);
    const int Wm = g.Wd - 1, THm = g.TH - 1, CT = g.C * 9;

    for (int i = t * 4; i < 2 * BUFSZ + 16; i += 1024) *reinterpret_cast<float4*>(ct_smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    // this lane's GEMM column (c,u,v): offset of its shifted pixel inside an x tile; columns beyond
    // C*9 read a spare cell that stays zero
    const bool colok = l31 < CT;
    const int cc = colok ? l31 / 9 : 0, tap = colok ? l31 - cc * 9 : 0, tu = tap / 3, tv = tap - tu * 3;
    const int lanex = cc * g.plane + tu * g.RS + tv + 3;

    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 sv[5];
    int so[5];
#define CS_SLOAD(SL, N0, R0)                                                                     \
    {                                                                                            \
        if ((SL) < 4) {                                                                          \
            const int e_ = t + 256 * (SL);                                                       \
            const int q_ = e_ & 31, f_ = e_ >> 5, p_ = 4 * q_;                                   \
            const int n_ = (N0) + (p_ >> g.lgP), row_ = (R0) + ((p_ >> g.lgW) & THm), k_ = kg * 32 + f_; \
            const bool ok_ = n_ < n_end && k_ < g.K;                                             \
            const int pl_ = min(n_, g.N - 1) * g.K + min(k_, g.K - 1);                            \
            if (POOL) sv[SL] = cw_mask4(pool_expand4(g.ps, pl_, row_, p_ & Wm), ok_);            \
            else sv[SL] = cw_mask4(*reinterpret_cast<const float4*>(g.dz + ((size_t)pl_ * g.H + row_) * g.Wd + (p_ & Wm)), ok_); \
            so[SL] = f_ * CW_DZS + p_;                                                           \
        } else {                                                                                 \
            const int e_ = t;                                                                    \
            int rr_ = min(e_, g.nx4 - 1);                                                        \
            const int q_ = rr_ % g.q4; rr_ /= g.q4;                                              \
            const int r_ = rr_ % g.THi; rr_ /= g.THi;                                            \
            const int ni_ = rr_ % g.NI, c_ = rr_ / g.NI;                                         \
            const int n_ = (N0) + ni_, row_ = (R0) - 1 + r_;                                     \
            const bool ok_ = n_ < n_end && (unsigned)row_ < (unsigned)g.H;                       \
            const int go_ = ((min(n_, g.N - 1) * g.C + c_) * g.H + min(max(row_, 0), g.H - 1)) * g.Wd + 4 * q_; \
            sv[SL] = cw_mask4(*reinterpret_cast<const float4*>(g.x + go_), ok_);                 \
            so[SL] = e_ < g.nx4 ? DZSZ + c_ * g.plane + (ni_ * g.THi + r_) * g.RS + 4 + 4 * q_ : -1; \
        }                                                                                        \
    }
#define CS_SSTORE(SL, BUF)                                                                       \
    {                                                                                            \
        *reinterpret_cast<float4*>(__builtin_assume_aligned(                                     \
            ct_smem + (so[SL] >= 0 ? (BUF) * BUFSZ + so[SL] : 2 * BUFSZ + 8), 16)) =             \
            sv[SL];                                                                              \
        if ((SL) < 4) dbacc[(SL) < 4 ? (SL) : 0] += dbw_ * ((sv[SL].x + sv[SL].y) + (sv[SL].z + sv[SL].w)); \
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float dbw_ = 1.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) CS_SLOAD(s, n_beg, 0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 5; ++s) CS_SSTORE(s, 0);
    __syncthreads();
    // the spare zero region [2*BUFSZ, 2*BUFSZ + 8) serves the columns beyond C*9; the dummy store
    // target of absent slots is 2*BUFSZ + 8
    int gi = 0, rt = 0;
    for (int tl = 0; tl < g.NT; ++tl) {
        const int cur = tl & 1;
        int rt1 = rt + 1, gi1 = gi;
        if (rt1 == g.RT) { rt1 = 0; ++gi1; }
        const bool hasnext = tl + 1 < g.NT;
        if (!hasnext) { rt1 = rt; gi1 = gi; }
        dbw_ = hasnext ? 1.f : 0.f;
        const int n1 = n_beg + gi1 * g.NI, r1 = rt1 * g.TH;
#pragma unroll
        for (int s = 0; s < 5; ++s) CS_SLOAD(s, n1, r1);
        const float* dzb = ct_smem + cur * BUFSZ + l31 * CW_DZS + 4 * hi;
        const float* xb = colok ? ct_smem + cur * BUFSZ + DZSZ + lanex : ct_smem + 2 * BUFSZ;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sg = wave + 4 * i, p_ = 8 * sg + 4 * hi;
            const float4 a = *cw_f4(dzb + 8 * sg);
            const float* xp = colok ? xb + ((p_ >> g.lgP) * g.THi + ((p_ >> g.lgW) & THm)) * g.RS + (p_ & Wm) : xb;
            const float x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, x1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, x2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, x3, acc, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 5; ++s) CS_SSTORE(s, cur ^ 1);
        __syncthreads();
        rt = rt1; gi = gi1;
    }
#undef CS_SLOAD
#undef CS_SSTORE
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v = dbacc[s];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
        const int k = kg * 32 + (t >> 5) + 8 * s;
        if (l31 == 0 && k < g.K) g.dbws[(size_t)z * g.K + k] = v;
    }
    if (colok) {
        float* wz = g.ws + (size_t)(z * 4 + wave) * g.K * CT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kg * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k < g.K) wz[(size_t)k * CT + cc * 9 + 8 - tap] = acc[r];
        }
    }
}

static int cs_geometry(ConvSG& g, int num_cus) {
    ConvWG w{};
    w.N = g.N; w.C = 32; w.H = g.H; w.Wd = g.Wd; w.K = 32;
    if (!cw_geometry(w, num_cus)) return 0;
    g.NI = w.NI; g.TH = w.TH; g.THi = w.THi; g.RT = w.RT; g.lgW = w.lgW; g.lgP = w.lgP; g.RS = w.RS;
    g.plane = g.NI * g.THi * g.RS;
    if ((g.plane & 15) == 0) g.plane += 4;          // keep the channel planes off one bank
    g.q4 = g.Wd / 4;
    g.nx4 = g.C * g.NI * g.THi * g.q4;
    if (g.nx4 > 256) return 0;
    g.KG = cdiv(g.K, 32);
    const int groups = cdiv(g.N, g.NI);
    int S = 4 * num_cus / g.KG;
    if (S > groups) S = groups;
    if (S < 1) S = 1;
    g.ipb = cdiv(groups, S) * g.NI;
    g.S = cdiv(g.N, g.ipb);
    g.NT = (g.ipb / g.NI) * g.RT;
    return 1;
}

extern "C" int tn_convpool_smallc_supported(int N, int C, int H, int Wd, int K, int f, int pad, int Ho,
                                            int Wo, int p, int Hp, int Wp) {
    if (!ct_enabled() || f != 3 || p != 2 || pad != 1 || Ho != H || Wo != Wd || C * 9 > 32 || K < 16) return 0;
    if ((Ho & 1) || (Wo & 3) || Hp * 2 != Ho || Wp * 2 != Wo) return 0;
    if (const char* e = getenv("TN_CONV_TILE_SMALLC")) if (e[0] == '0') return 0;
    ConvSG g{};
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    return cs_geometry(g, 256);
}

static int cs_run(tn_ctx* ctx, ConvSG& g, float* dW, float* db, bool pool) {
    TN_REQUIRE(cs_geometry(g, ctx->num_cus), "conv_tile_smallc: unsupported shape");
    TN_REQUIRE((long long)g.N * g.K * g.H * g.Wd < (1ll << 31), "conv_tile_smallc: tensor too large for 32-bit offsets");
    const size_t n = (size_t)g.K * g.C * 9;
    int rc = tn_scratch_get(ctx, ((size_t)g.S * 4 * n + (size_t)g.S * g.K) * sizeof(float), &g.ws);
    if (rc) return rc;
    g.dbws = g.ws + (size_t)g.S * 4 * n;
    const int XSZ = (g.C * g.plane + 3) & ~3;
    const size_t lds = (size_t)(2 * (32 * CW_DZS + XSZ) + 16) * sizeof(float);
    const int grid = 8 * cdiv(g.S, 8) * g.KG;
    if (pool) conv_tile_wgrad_smallc_kernel<true><<<grid, 256, lds, ctx->stream>>>(g);
    else conv_tile_wgrad_smallc_kernel<false><<<grid, 256, lds, ctx->stream>>>(g);
    TN_LAUNCH_CHECK();
    rc = tn_red_push(ctx, g.ws, dW, (uint32_t)n, (uint32_t)(g.S * 4), (uint32_t)n, 0);
    if (rc) return rc;
    rc = tn_red_push(ctx, g.dbws, db, (uint32_t)g.K, (uint32_t)g.S, (uint32_t)g.K, 0);
    if (rc) return rc;
    return tn_red_commit(ctx);
}

int tn_conv_tile_smallc_bwd(tn_ctx* ctx, const float* x, const float* g_, const float* y, const uint8_t* mask,
                            float* dW, float* db, int N, int C, int H, int Wd, int K, int act, float prm) {
    ConvSG g{};
    g.x = x;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    g.ps.g = g_; g.ps.y = y; g.ps.mask = mask; g.ps.Hp = H / 2; g.ps.Wp = Wd / 2; g.ps.act = act; g.ps.prm = prm;
    TN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g_) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(mask) & 3) == 0, "conv_tile_smallc_bwd: misaligned operand");
    return cs_run(ctx, g, dW, db, true);
}

// unfused first layers: dW, db from the plain dz tensor
int tn_conv_tile_smallc_ok(const float* x, const float* dz, int N, int C, int H, int Wd, int K, int f, int pad,
                           int Ho, int Wo) {
    if (!ct_enabled() || f != 3 || pad != 1 || Ho != H || Wo != Wd || C * 9 > 32 || K < 16) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz)) & 15) return 0;
    if (const char* e = getenv("TN_CONV_TILE_SMALLC")) if (e[0] == '0') return 0;
    ConvSG g{};
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    return cs_geometry(g, 256);
}

int tn_conv_tile_smallc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db, int N, int C,
                              int H, int Wd, int K) {
    ConvSG g{};
    g.x = x; g.dz = dz;
    g.N = N; g.C = C; g.H = H; g.Wd = Wd; g.K = K;
    return cs_run(ctx, g, dW, db, false);
}
