/*
 * theanet_hip.h -- C-ABI of libtheanet_hip.so: the MI355X (gfx950) compute backend
 * that replaces Theano underneath theanet's convolutional training hot path.
 *
 * The reference (rakeshvar/theanet) has no FFI: its boundary to the compute
 * backend is "build a Theano graph, call theano.function".  Every entry point
 * below therefore cites the *reference call site into Theano* it replaces
 * (paths relative to the reference root).  The Python host package
 * (theanet_amd/) binds these with ctypes; see INTEGRATION.md for the stub.
 *
 * Conventions
 *  - plain C symbols, plain pointers and sizes; no C++/torch types.
 *  - every function returns int: 0 = ok, <0 = error (TN_E_*); the message is
 *    available from tn_last_error(ctx) (ctx may be NULL for creation errors).
 *  - one ctx <-> one device <-> one HIP stream.  All ops ENQUEUE on that stream
 *    and return immediately; tn_sync / tn_d2h wait.  A ctx is not thread-safe;
 *    distinct ctxs are independent.
 *  - all tensors float32, NCHW, row-major, device pointers from tn_alloc.
 *    Labels int32.  Masks uint8 (0/1).
 *  - "d_*" scalar-pointer arguments are DEVICE pointers (may be NULL -> the
 *    by-value sibling is used) so a captured HIP graph can be replayed with
 *    per-step values changed on the device.
 */
#ifndef THEANET_HIP_H
#define THEANET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TN_OK 0
#define TN_E_HIP (-1)     /* a HIP runtime call failed            */
#define TN_E_ARG (-2)     /* invalid argument / unsupported shape */
#define TN_E_COMM (-3)    /* RCCL failure                         */
#define TN_E_NOMEM (-4)

/* activation kinds -- theanet/layer/layer.py:27-39 (activation_list) */
enum tn_act {
    TN_ACT_LINEAR = 0,
    TN_ACT_LEAKY = 1,       /* relu / relu00..relu99: max(0,z)+min(0,z)*slope   */
    TN_ACT_TANH = 2,
    TN_ACT_SIGMOID = 3,
    TN_ACT_SOFTPLUS = 4,
    TN_ACT_SCALED_TANH = 5  /* 1.7*tanh(2z/3)                                   */
};

typedef struct tn_ctx tn_ctx;

/* ---- lifecycle (replaces: theano device init; neuralnet.py:236 theano.function) ---- */
int tn_version(void);
int tn_device_count(int* count);
int tn_ctx_create(int device, tn_ctx** ctx);
int tn_ctx_destroy(tn_ctx* ctx);
const char* tn_last_error(tn_ctx* ctx);
int tn_sync(tn_ctx* ctx);
/* Two streams per context: 0 = main (the dependent chain), 1 = side (leaf work such as weight
 * gradients, which only the update needs).  tn_stream_select picks the stream the following
 * ops are enqueued on; tn_stream_wait(w, s) makes stream w wait for everything enqueued on s
 * so far.  Both calls are legal inside tn_graph_begin/_end (cross-stream capture).           */
int tn_stream_select(tn_ctx* ctx, int idx);
int tn_stream_wait(tn_ctx* ctx, int waiter, int signaler);
/* name_len bytes are written to name; cus = compute units; hbm_bytes = total memory */
int tn_device_info(tn_ctx* ctx, char* name, int name_len, int* cus, size_t* hbm_bytes);

/* ---- memory (replaces: theano.shared / get_value; train.py:18-19, weights.py:18-22,78-79) ---- */
int tn_alloc(tn_ctx* ctx, size_t bytes, void** dptr);
int tn_free(tn_ctx* ctx, void* dptr);
int tn_h2d(tn_ctx* ctx, void* dst, const void* src, size_t bytes);   /* returns after the copy */
int tn_d2h(tn_ctx* ctx, void* dst, const void* src, size_t bytes);   /* returns after the copy */
/* Outputs a caller reads every step (the [cost, features, logprob] of neuralnet.py:236-240's function) leave
 * as soon as they exist: tn_d2h_early orders a copy into page-locked host memory (tn_host_alloc) behind the
 * work enqueued so far on the current stream and runs it on a copy stream, under the kernels enqueued
 * afterwards; tn_copy_sync waits for every such copy.                                                   */
int tn_host_alloc(tn_ctx* ctx, size_t bytes, void** out);
int tn_host_free(tn_ctx* ctx, void* p);
int tn_d2h_early(tn_ctx* ctx, void* host_dst, const void* src, size_t bytes);
int tn_copy_sync(tn_ctx* ctx);
/* the same copy with an event (tn_event_create) recorded behind it on the copy stream: the host waits for THAT copy
 * (tn_event_sync) instead of for all of them -- a step's cost picked up a few steps later by train.py's loop
 * (theanet_amd/trainfn.py _CostRing; polling the destination instead is not an option: a 4-byte copy was observed
 * half-written from the host) */
int tn_d2h_early_ev(tn_ctx* ctx, void* host_dst, const void* src, size_t bytes, void* done_event);
int tn_d2d(tn_ctx* ctx, void* dst, const void* src, size_t bytes);   /* enqueued                */
int tn_memset(tn_ctx* ctx, void* dst, int byte_value, size_t bytes); /* enqueued                */
int tn_set_u32(tn_ctx* ctx, uint32_t* d_dst, uint32_t value);        /* enqueued scalar store   */
int tn_set_i64(tn_ctx* ctx, int64_t* d_dst, int64_t value);
int tn_set_f32(tn_ctx* ctx, float* d_dst, float value);
int tn_add_u32(tn_ctx* ctx, uint32_t* d_dst, uint32_t inc);          /* *d_dst += inc (step counter) */

/* ---- a whole step as ONE call (the coarse entry point for launch-bound steps) ----
 * A plan is the flat list of the C-ABI calls a step makes -- entry point + argument block -- built once by the host
 * (theanet_amd/plan.py records the calls of a few ordinary steps of a NeuralNet function, checks that they repeat and
 * which integer arguments follow the minibatch index) and replayed by tn_net_step without the interpreter: what the
 * reference's compiled theano.function does for fn(i) (neuralnet.py:236-241).  tn_net_plan_add: name = an entry point
 * of this header taking the context first (the context is not in the list); kinds[k] = 0 pointer / integer (vals[k] =
 * the value as 64 bits, strides[k] added per unit of tn_net_step's index), 1 float (its 32 bits), 2 double (its 64
 * bits); at most 48 integer and 8 floating-point arguments.  tn_net_step issues the calls in order and stops at the
 * first error.  The plan holds raw pointers: it lives no longer than the buffers of the net it was recorded from. */
int tn_net_plan_create(tn_ctx* ctx, void** plan);
int tn_net_plan_add(tn_ctx* ctx, void* plan, const char* name, int nargs, const uint8_t* kinds, const uint64_t* vals,
                    const int64_t* strides);
int tn_net_step(tn_ctx* ctx, void* plan, int64_t index);
int tn_net_plan_size(tn_ctx* ctx, void* plan);
int tn_net_plan_destroy(tn_ctx* ctx, void* plan);

/* ---- HIP graph capture of an op sequence issued through this ABI ---- */
int tn_graph_begin(tn_ctx* ctx);                   /* start stream capture            */
int tn_graph_end(tn_ctx* ctx, void** graph_exec);  /* stop, instantiate               */
int tn_graph_launch(tn_ctx* ctx, void* graph_exec);
int tn_graph_destroy(tn_ctx* ctx, void* graph_exec);

/* ---- timing with HIP events on the ctx stream (bench.py roofline leg) ---- */
int tn_event_create(tn_ctx* ctx, void** ev);
int tn_event_record(tn_ctx* ctx, void* ev);
int tn_event_wait(tn_ctx* ctx, void* ev);            /* the current stream waits for the event */
int tn_event_elapsed_ms(tn_ctx* ctx, void* ev_start, void* ev_stop, float* ms); /* syncs on stop */
int tn_event_destroy(tn_ctx* ctx, void* ev);
/* non-blocking: *done = 1 once everything recorded in front of the event has finished, else 0 (watchdogs
 * that must raise instead of hanging: the communicator self-test of theanet_amd/comm.py)              */
int tn_event_query(tn_ctx* ctx, void* ev, int* done);
int tn_event_sync(tn_ctx* ctx, void* ev);            /* the HOST waits for the event */

/* ---- conv (replaces nnconv.conv2d + tt.grad through it; convpool.py:54-72, layer.py:83) ----
 * True convolution (kernel flipped), W layout (K,C,f,f):
 *   z[n,k,i,j] = b[k] + sum_{c,u,v} xpad[n,c,i*s+u,j*s+v] * W[k,c,f-1-u,f-1-v]
 * pad_lo zeros are virtually prepended to rows/cols (0 for 'valid', f-1-(f-1)/2 for
 * 'same'); Ho/Wo are the output sizes the caller computed (convpool.py:57-70).
 * fwd fuses bias + activation:  a = act(z).                                         */
int tn_conv2d_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a,
                  int N, int C, int H, int Wd, int K, int f, int stride, int pad_lo,
                  int Ho, int Wo, int act, float act_param);
/* dW (K,C,f,f) and db (K) from dz = dcost/dz (activation gradient already applied by
 * the kernel that produced dz).  OVERWRITES dW/db.                                   */
int tn_conv2d_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db,
                    int N, int C, int H, int Wd, int K, int f, int stride, int pad_lo,
                    int Ho, int Wo);
/* dx (N,C,H,W) = full correlation of dz with W.  If prev_a != NULL the gradient of the
 * producing layer's activation is fused:  dx *= act'(prev_a)  (prev_a = that layer's
 * OUTPUT, same shape as dx) -- i.e. the result is already dcost/dz of the layer below. */
int tn_conv2d_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx,
                    int N, int C, int H, int Wd, int K, int f, int stride, int pad_lo,
                    int Ho, int Wo, const float* prev_a, int prev_act, float prev_act_param);

/* ---- DTYPE 'float16' (BASELINE configs[4]; the reference is float32-only, weights.py:8 `floatX`: an opt-in
 * training param of this build, default off) ----
 * dtype 0: every tensor fp32 (reference arithmetic).  dtype 1: the conv stack of the net lives in HBM as halfs and is
 * served by the tn_c8_* / tn_fc8_* entry points below; the fp32-tensor conv entry points (tn_conv2d_*, tn_convpool_*)
 * REFUSE to run in this mode (TN_E_ARG: never a silent fp32 run inside a float16 net).  grad_scale (a power of two,
 * e.g. 4096): gradient tensors of the stack hold fp16(grad_scale * g) so that small gradients do not fall into
 * fp16's subnormal range; the fp32 epilogues of the weight gradients remove the factor.                          */
int tn_set_matmul_dtype(tn_ctx* ctx, int dtype, float grad_scale);
int tn_get_matmul_dtype(tn_ctx* ctx);
/* MATMUL 'bf16x3' (opt-in, this build's extension; the reference is float32, weights.py:8): mode 1 runs the products of
 * tn_fc_fwd / tn_fc_wgrad / tn_fc_dgrad / tn_fc_bwd of layers with more than 16 outputs as six bf16 MFMA products of
 * exactly split operands (x = x0 + x1 + x2, 8 mantissa bits each) with fp32 accumulation -- fp32-grade accuracy
 * (the same tolerances hold), not the same bits as mode 0's exact fp32 MFMA.  theanet_amd/csrc/gemm_b3.hip.        */
int tn_set_fc_matmul(tn_ctx* ctx, int mode);

/* ---- DTYPE 'float16' on fp16-RESIDENT tensors (theanet_amd/csrc/conv_c8.hip, fc_c8.hip) --------------------
 * BASELINE.json configs[4]: "fp16 inputs / fp32 accum MFMA".  The reference is float32-only (weights.py:8); these
 * entry points replace the same Theano call sites as tn_conv2d_* / tn_fc_* (convpool.py:54-72,106-107; hidden.py:30;
 * layer.py:83) for nets built with training param DTYPE = 'float16'.  Activations and the gradients flowing down
 * the net live in HBM as IEEE halfs in the "c8" layout: logical (N, C, H, W) stored [N][ceil(C/8)][H][W][8]
 * (16-byte cell = the 8 channels of an octet at one pixel; channels beyond C are zero).  Master weights, biases,
 * weight gradients, velocities stay fp32.  Gradient tensors hold grad_scale * g (tn_set_matmul_dtype's grad_scale, a
 * power of two; scaled where the first fp16 gradient is produced, removed in the fp32 epilogues of the weight
 * gradients).  Arithmetic: operands rounded to half (nearest-even), exact products, fp32 accumulation; bias,
 * activation and pooling on the fp32 sums; one rounding when a tensor is stored.  Pooling masks: one byte per pooled
 * value in the same [N][K/8][H/2][W/2][8] order, bits 0-3 = window elements (2*di+dj) equal to the maximum (all of
 * them on a tie, Theano's MaxPoolGrad), bit 4 / 5 = pooled value > 0 / < 0.                                      */
int tn_c8_conv_supported(int N, int C, int H, int W, int K, int f, int stride, int pad);
int tn_c8_conv_wgrad_supported(int N, int C, int H, int W, int K);
/* wt (tn_c8_conv_fwd / _dgrad): the layer's weights as fp16 MFMA operand tiles, prepared beforehand by
 * tn_c8_arrange_multi into a buffer of tn_c8_wt_elems halfs -- every conv product of a step in ONE launch -- or NULL:
 * the op arranges them itself (one small launch per call).  The arrangement is a function of W only: redo it after
 * every update.                                                                                                    */
typedef struct tn_c8_wt_seg {
    const float* W;  /* (K, C, 3, 3) master weights */
    void* wt;        /* tn_c8_wt_elems(K, C, dgrad) halfs */
    int K, C, dgrad; /* dgrad != 0: the operand tiles of the input-gradient product */
    int pad_;
} tn_c8_wt_seg;
size_t tn_c8_wt_elems(int K, int C, int dgrad);
int tn_c8_arrange_multi(tn_ctx* ctx, const tn_c8_wt_seg* segs, int nseg);
int tn_c8_conv_fwd(tn_ctx* ctx, const void* x, const float* W, const float* b, void* y, uint8_t* mask, int N, int C,
                   int H, int Wd, int K, int act, float prm, int pool, const void* wt);
/* dx = conv^T(dz, W) * act'(prev_a) (prev_a NULL: none; for a pooled block below, prev_a is its POOLED output and dx
 * has that shape: the gradient a pooled block receives always carries act'(pooled output)).  pooled != 0: dz is not a
 * tensor: the `dz` argument is the pooled gradient (N, K, H/2, W/2) and dz = (bit of the window element in the block's
 * mask) ? pooled gradient : 0 is formed while staging: the conv activation, MaxPoolGrad's output and dz never exist
 * in HBM                                                                                                            */
int tn_c8_conv_dgrad(tn_ctx* ctx, const void* dz, const float* W, void* dx, int N, int C, int H, int Wd, int K,
                     const void* prev_a, int prev_act, float prev_prm, int pooled, const uint8_t* mask, const void* wt);
int tn_c8_conv_wgrad(tn_ctx* ctx, const void* x, const void* dz, float* dW, float* db, int N, int C, int H, int Wd,
                     int K, int pooled, const uint8_t* mask);
/* fully-connected products on an fp16-resident input (replaces hidden.py:30 and its gradients, layer.py:83, for the
 * first dense layer above a c8 conv stack; theanet_amd/csrc/fc_c8.hip): x16 (B, ceil(C/8)*HW*8) halfs is the flattened
 * c8 tensor of C maps of HW pixels (HW = 1: a plain half matrix), W (C*HW, n_out) fp32 keeps the reference's
 * NCHW-flattened row order (neuralnet.py:168-173) and is walked through the row map; the layer output a and dz are
 * fp32.  dgrad writes fp16(grad_scale * dz . W^T * act'(y16)) in x's order (y16 = output of the layer below, NULL:
 * none); wgrad removes the scale.                                                                                */
int tn_c8_fc_supported(int B, int C, int HW, int n_out);
int tn_c8_fc_fwd(tn_ctx* ctx, const void* x, const float* W, const float* b, float* a, int B, int C, int HW, int n_out,
                 int act, float act_param, const uint8_t* mask);
/* ... with the layer's dropout mask drawn in the same launch (dropout.py:10-13; the numbers of tn_dropout_mask with the
 * same seed / step / elem0) and written to mask_out for the backward pass.                                          */
int tn_c8_fc_fwd_dropout(tn_ctx* ctx, const void* x, const float* W, const float* b, float* a, int B, int C, int HW,
                         int n_out, int act, float act_param, uint8_t* mask_out, float pdrop, uint64_t seed, uint32_t step,
                         const uint32_t* d_step, uint64_t elem0);
int tn_c8_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, void* dx, int B, int C, int HW, int n_out, const void* y,
                   int act, float act_param);
int tn_c8_fc_wgrad(tn_ctx* ctx, const void* x, const float* dz, float* dW, float* db, int B, int C, int HW, int n_out);
/* NCHW fp32 (rows row0.. of a dataset) -> c8 fp16 and back; values are multiplied by scale                        */
int tn_c8_pack(tn_ctx* ctx, const float* x, int64_t row0, void* out, int N, int C, int HW, float scale);
/* tn_elastic_apply (below; inlayers.py:126-142) whose output is the c8 tensor the first conv layer consumes: the same
 * values, rounded to halfs when stored -- the resampled fp32 minibatch is neither written nor re-read by tn_c8_pack.  */
int tn_c8_elastic_apply(tn_ctx* ctx, const float* x, int64_t x_row0, const int64_t* d_row0, void* out16,
                        int N, int C, int h, int w, int invert, int nearest, const int32_t* map_idx,
                        const float* map_fy, const float* map_fx, float pflip, const uint8_t* flipmask,
                        uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0);
int tn_c8_unpack(tn_ctx* ctx, const void* x, float* out, int N, int C, int HW, float scale);

/* 1 if tn_conv2d_* run this shape on the implicit-im2col fp32-MFMA kernels (stride 1, reduction
 * C*f*f >= 32, >= 16 output maps); otherwise the direct VALU kernels are used.              */
int tn_conv_mfma_supported(int C, int K, int f, int stride);

/* ---- fused conv + bias + act + max-pool for small feature maps (same reference call sites as
 * tn_conv2d_* followed by tn_pool_*: convpool.py:54-72 then :106-107).  The conv activation
 * never reaches HBM: fwd writes only the pooled map y (N,K,Hp,Wp); bwd takes g = dcost/dy,
 * recomputes the window, routes g to every maximal element (MaxPoolGrad tie rule) times
 * act'(a), and produces dW/db (OVERWRITE) and -- only if dz != NULL -- dz (N,K,Ho,Wo) for a
 * following tn_conv2d_dgrad.  Supported: stride 1, p == 2, f == 3 with C <= 4, f == 5 with
 * C <= 2 (tn_convpool_supported); anything else uses the unfused ops.                      */
int tn_convpool_supported(int C, int f, int stride, int p);
int tn_convpool_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y,
                    int N, int C, int H, int Wd, int K, int f, int pad_lo, int Ho, int Wo,
                    int p, int Hp, int Wp, int act, float act_param);
int tn_convpool_bwd(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                    float* dz, float* dW, float* db, int N, int C, int H, int Wd, int K, int f,
                    int pad_lo, int Ho, int Wo, int p, int Hp, int Wp, int act, float act_param);
/* tn_convpool_fwd that also records WHERE each pooled value came from: bit 2*di+dj of
 * mask[n,k,i,j] (uint8, shape of y) is set iff window element (di,dj) exists and attains the
 * maximum -- all of them on a tie, the elements Theano's MaxPoolGrad routes the gradient to
 * (convpool.py:106-107); bits 4 / 5 hold y > 0 / y < 0.  mask == NULL is plain tn_convpool_fwd.
 * tn_convpool_bwd_mask is tn_convpool_bwd driven by that record (f == 3 only): dz = mask bit ?
 * g * act'(y) : 0 with no conv recompute (y is read only when act is not leaky-relu).        */
int tn_convpool_fwd_mask(tn_ctx* ctx, const float* x, const float* W, const float* b, float* y,
                         uint8_t* mask, int N, int C, int H, int Wd, int K, int f, int pad_lo,
                         int Ho, int Wo, int p, int Hp, int Wp, int act, float act_param);
int tn_convpool_bwd_mask(tn_ctx* ctx, const float* x, const float* g, const float* y,
                         const uint8_t* mask, float* dz, float* dW, float* db, int N, int C, int H,
                         int Wd, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp, int Wp,
                         int act, float act_param);

/* LDS-resident backward of the same fused block for MANY filter elements (K*C*f*f in the
 * hundreds, e.g. mnist.prms conv2): a block keeps G whole images' x and dz in LDS, so dz never
 * reaches HBM and dx (= dcost/dx, may be NULL) comes out of the same kernel as dW/db.
 * tn_convblock_supported returns the group size G (0 = use tn_convpool_bwd + tn_conv2d_dgrad). */
int tn_convblock_supported(int C, int K, int f, int stride, int p, int Ho, int Wo);
int tn_convblock_bwd(tn_ctx* ctx, const float* x, const float* W, const float* b, const float* g,
                     float* dx, float* dW, float* db, int N, int C, int H, int Wd, int K, int f,
                     int pad_lo, int Ho, int Wo, int p, int Hp, int Wp, int act, float act_param);
/* Wide conv + act + 2x2 max-pool blocks (3x3 'same', C*9 >= 32, K >= 16, even maps, rows % 4 == 0)
 * on the LDS-tile matrix-core kernels: tn_convpool_fwd_mask pools in the conv kernel's epilogue (the
 * conv activation never reaches HBM) and tn_convpool_bwd_mask_dx forms dz = mask bit ? g*act'(y) : 0
 * while it stages the operands of the weight- and input-gradient products -- MaxPoolGrad, the
 * activation gradient, CorrMM_gradWeights and CorrMM_gradInputs (convpool.py:54-72,106 under
 * theano.grad, layer.py:83) without dz ever existing.  prev_a / prev_act: output and activation of
 * the layer below, whose gradient is applied to dx in the epilogue (NULL: none); dx NULL: weight
 * gradients only.                                                                               */
int tn_convpool_tile_supported(int N, int C, int H, int Wd, int K, int f, int stride, int pad_lo, int Ho,
                               int Wo, int p, int Hp, int Wp);
int tn_convpool_bwd_mask_dx(tn_ctx* ctx, const float* x, const float* W, const float* g, const float* y,
                            const uint8_t* mask, float* dx, float* dW, float* db, int N, int C, int H,
                            int Wd, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp, int Wp, int act,
                            float act_param, const float* prev_a, int prev_act, float prev_act_param);

/* The same backward driven by the forward's record instead of a conv recompute: y and mask are
 * tn_convpool_fwd_mask's outputs, so dz = mask bit ? g * act'(y) : 0.  One wave per image on
 * the fp32 matrix cores (wgrad and dgrad products), dz lives in LDS only.  dx may be NULL;
 * dW/db are OVERWRITTEN.  Shapes: f == 3, stride 1, p == 2 keeping the border, C <= 4, K <= 32,
 * Wo <= 14, image and pooled map <= 768 elements (tn_convblock_mask_supported).             */
int tn_convblock_mask_supported(int C, int K, int f, int stride, int p, int H, int Wd, int pad_lo,
                                int Ho, int Wo, int Hp, int Wp);
int tn_convblock_bwd_mask(tn_ctx* ctx, const float* x, const float* W, const float* g,
                          const float* y, const uint8_t* mask, float* dx, float* dW, float* db,
                          int N, int C, int H, int Wd, int K, int f, int pad_lo, int Ho, int Wo,
                          int p, int Hp, int Wp, int act, float act_param);

/* ---- pool / mean (replaces pool.pool_2d + MaxPoolGrad, tt.mean; convpool.py:106-107,131) ----
 * max over p x p, stride p, no padding; Ho = ceil(H/p) unless ignore_border (floor).   */
int tn_pool_fwd(tn_ctx* ctx, const float* x, float* y, int NC, int H, int Wd, int p,
                int Ho, int Wo);
/* dx = (x == y[window]) ? dy[window] : 0  (every tie gets the full gradient), then the
 * producing layer's activation gradient is fused exactly as in tn_conv2d_dgrad
 * (x IS that layer's output).  prev_act = TN_ACT_LINEAR disables it.                    */
int tn_pool_bwd(tn_ctx* ctx, const float* x, const float* y, const float* dy, float* dx,
                int NC, int H, int Wd, int p, int Ho, int Wo, int prev_act, float prev_act_param);
int tn_mean_fwd(tn_ctx* ctx, const float* x, float* y, int NC, int HW);
int tn_mean_bwd(tn_ctx* ctx, const float* dy, float* dx, int NC, int HW,
                const float* prev_a, int prev_act, float prev_act_param);

/* ---- fully connected (replaces tt.dot + bias + act + drop_output; hidden.py:30-32) ----
 * a = act(x(B,n_in) . W(n_in,n_out) + b) ; if mask != NULL: a *= mask (uint8 B x n_out,
 * non-inverted dropout, dropout.py:12-13).  fp32 MFMA (v_mfma_f32_32x32x2_f32).         */
int tn_fc_fwd(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a,
              int B, int n_in, int n_out, int act, float act_param, const uint8_t* mask);
/* dW = x^T . dz, db = sum_rows dz (OVERWRITE).  ws: workspace of tn_fc_wgrad_ws_bytes.   */
size_t tn_fc_wgrad_ws_bytes(int B, int n_in, int n_out);
int tn_fc_wgrad(tn_ctx* ctx, const float* x, const float* dz, float* dW, float* db,
                int B, int n_in, int n_out, void* ws);
/* dx = dz . W^T, with the layer-below's activation gradient and dropout mask fused:
 * dx *= act'(prev_a) * prev_mask   (either may be NULL).                                */
int tn_fc_dgrad(tn_ctx* ctx, const float* dz, const float* W, float* dx,
                int B, int n_in, int n_out,
                const float* prev_a, int prev_act, float prev_act_param, const uint8_t* prev_mask);

/* tn_fc_wgrad + tn_fc_dgrad of one layer as ONE op.  The two products only share dz, so their
 * blocks are interleaved in a single launch (a kernel boundary less, and the prologue / epilogue
 * of one product overlaps the main loop of the other).  Same arguments and results as the two
 * ops; shapes the paired kernel cannot take run them one after the other.                    */
int tn_fc_bwd(tn_ctx* ctx, const float* x, const float* dz, const float* W, float* dW, float* db,
              float* dx, int B, int n_in, int n_out, void* ws, const float* prev_a, int prev_act,
              float prev_act_param, const uint8_t* prev_mask);

/* The whole SoftmaxLayer forward (outlayers.py:87-95 + :50-51) as one op: logits = x W + b
 * followed by tn_softmax_nll on them.  With at most 16 classes it is one launch (the class
 * logits of a row sit in one 16-lane DPP row of the matrix-core epilogue); wider heads run
 * tn_fc_fwd + tn_softmax_nll.  Argument meaning as in those two.                            */
int tn_fc_softmax_nll(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits, int B,
                      int n_in, int n_out, const int32_t* y, int64_t y_row0, const int64_t* d_row0,
                      float* logprob, float* rowloss, int32_t* pred, float* rowp, float* dz,
                      float inv_batch);
/* ... and the training step of that layer as ONE op: tn_fc_softmax_nll followed by tn_fc_bwd with
 * dz = dlogits (dW, db OVERWRITTEN, dx = d cost / d (input) times act'(prev_a) * prev_mask of the
 * layer below; prev_a, if given, is x itself).  With at most 16 classes the whole thing is one launch
 * per 32 rows: logits, log-softmax / NLL, the input gradient and the rows' weight-gradient slab
 * (ws >= tn_fc_wgrad_ws_bytes).                                                               */
int tn_fc_softmax_train(tn_ctx* ctx, const float* x, const float* W, const float* b, float* logits, int B,
                        int n_in, int n_out, const int32_t* y, int64_t y_row0, const int64_t* d_row0,
                        float* logprob, float* rowloss, int32_t* pred, float* rowp, float* dz,
                        float inv_batch, float* dW, float* db, float* dx, void* ws, const float* prev_a,
                        int prev_act, float prev_act_param, const uint8_t* prev_mask);
/* tn_fc_fwd with the dropout mask DRAWN in the same launch: mask_out[i] is exactly what
 * tn_dropout_mask(seed, step, d_step, elem0) would produce for a (B, n_out) tensor, the output is
 * multiplied by it, and mask_out stays behind for the backward pass (hidden.py:40-43 + dropout.py:
 * 9-13 as one op).  Shapes the fused epilogue cannot take run tn_dropout_mask + tn_fc_fwd.   */
int tn_fc_fwd_dropout(tn_ctx* ctx, const float* x, const float* W, const float* b, float* a, int B,
                      int n_in, int n_out, int act, float act_param, uint8_t* mask_out, float pdrop,
                      uint64_t seed, uint32_t step, const uint32_t* d_step, uint64_t elem0);

/* ---- dropout (replaces RandomStreams.binomial; dropout.py:9-31) ----
 * mask[i] = uniform(seed, *d_step + step, elem0 + i) >= pdrop  (Philox4x32-10), i < n.
 * elem0 = global index of element 0 so masks do not depend on how a batch is sharded.  */
int tn_dropout_mask(tn_ctx* ctx, uint8_t* mask, size_t n, float pdrop, uint64_t seed,
                    uint32_t step, const uint32_t* d_step, uint64_t elem0);
/* stand-alone DropOutLayer: y = x * mask (fwd);  dx = dy * mask * act'(prev_a) (bwd).   */
int tn_scale_mask(tn_ctx* ctx, const float* x, const uint8_t* mask, float scale, float* y,
                  size_t n, const float* prev_a, int prev_act, float prev_act_param);

/* ---- softmax + NLL (replaces tt.nnet.softmax, log, argmax, -mean; outlayers.py:50-51,69-95) ----
 * logprob = log_softmax(z); rowloss[n] = -logprob[n,y[n]]; pred[n] = argmax (first max);
 * rowp[n] = exp(logprob[n,y[n]]); dz = (softmax - onehot) * inv_batch (NULL to skip).
 * y is read at y[y_row0 + (d_row0 ? *d_row0 : 0) + n].                                  */
int tn_softmax_nll(tn_ctx* ctx, const float* z, const int32_t* y, int64_t y_row0,
                   const int64_t* d_row0, float* logprob, float* rowloss, int32_t* pred,
                   float* rowp, float* dz, int B, int n_out, float inv_batch);
/* Same, plus cost[0] = cost_scale * sum(rowloss) produced by the LAST block to finish (fixed
 * summation order: deterministic), which saves a separate reduction launch.  ws: scratch of
 * tn_softmax_cost_ws_bytes(B) bytes that must be zero before the first call (the kernel leaves
 * it ready for the next one).                                                               */
size_t tn_softmax_cost_ws_bytes(int B);
int tn_softmax_nll_cost(tn_ctx* ctx, const float* z, const int32_t* y, int64_t y_row0,
                        const int64_t* d_row0, float* logprob, float* rowloss, int32_t* pred,
                        float* rowp, float* dz, int B, int n_out, float inv_batch,
                        float cost_scale, float* cost, void* ws);
/* ---- the other output heads and losses (replaces outlayers.py:38-64 losses, :105-126 ExpLossLayer, :129-147
 * HingeLayer, :153-224 CenteredOutLayer) as one row kernel.  head: 0 SOFTMAX (a = logits), 1 EXPLOSS (a = linear
 * output; feat receives a - mean(a)), 2 HINGE (a = linear output), 3 LOGIT / 4 RBF (a = the hidden layer's
 * activated features (B, n), centers (ncls, n); logprob has ncls / ncls+1 columns).  loss (head SOFTMAX only;
 * the other heads have their own): 0 nll, 1 nllsq, 2 nll truncated (loss_param = log threshold), 3 hinge and
 * 4 exp on the softmax output.  rowloss[n]: cost = mean(rowloss); pred = argmax (first maximum); rowstat[n]: the
 * reference's second error statistic per row (P(label); raw output for HINGE; share of wrong bits for LOGIT);
 * da = d cost / d a scaled by inv_batch -- for the centered heads already multiplied by act'(a) (act, act_param =
 * the hidden activation: sigmoid / scaled_tanh), i.e. d cost / d z; dcenters (RBF, may be NULL) accumulates
 * d cost / d centers into a buffer the caller zeroed.  Outputs other than logprob may be NULL.            */
int tn_head_rows(tn_ctx* ctx, int head, int loss, float loss_param, const float* a, const float* centers,
                 int ncls, const int32_t* y, int64_t y_row0, const int64_t* d_row0, float* feat,
                 float* logprob, float* rowloss, int32_t* pred, float* rowstat, float* da, float* dcenters,
                 int B, int n, float inv_batch, float junk_dist, int act, float act_param);
/* out[0] = scale * sum(v[0..n))  (+ out[0] if accumulate) -- cost and error-rate scalars */
int tn_reduce_sum(tn_ctx* ctx, const float* v, size_t n, float scale, float* out, int accumulate);
/* out[0] (+)= L1*sum|p| + L2*sum p^2  (layer.py:109-117)                                 */
int tn_wtcost(tn_ctx* ctx, const float* p, size_t n, float L1, float L2, float* out, int accumulate);
/* out[0] = mean(pred != y) ; out[1] = mean(rowp)   (outlayers.py:69-80)                  */
int tn_error_stats(tn_ctx* ctx, const int32_t* pred, const int32_t* y, int64_t y_row0,
                   const float* rowp, int B, float* out2);

/* ---- finishing reductions of the weight-gradient ops ----
 * tn_conv2d_wgrad, tn_convpool_bwd*, tn_convblock_bwd* and tn_fc_wgrad end with a fixed-order sum
 * of partial slabs.  Between tn_defer_reductions(ctx, 1) and tn_defer_reductions(ctx, 0) those
 * sums are only recorded; the closing call runs all of them as ONE launch (dW/db are valid after
 * it, in stream order).  Outside such a window every op finishes its own gradient.          */
int tn_defer_reductions(tn_ctx* ctx, int on);
/* tn_defer_reductions(ctx, 0) that also advances a device counter (the RNG step counter) in the same
 * launch, so that everything enqueued afterwards already sees the next step's value.            */
int tn_defer_flush_step(tn_ctx* ctx, uint32_t* d_step);
/* A pipelined step may leave its window open (tn_sgd_update_net in TN_UPD_PIPE mode closes it a step later; the window
 * travels with its stream across tn_stream_select).  tn_defer_discard forgets the recorded sums of both
 * streams without running them -- for windows whose owner (its gradient buffers) is gone.               */
int tn_defer_discard(tn_ctx* ctx);

/* ---- momentum SGD + maxnorm (replaces Layer.get_updates; layer.py:70-107) ----
 * g' = g*gscale + L1*sign(p) + 2*L2*p ; v_new = m*v + (1-m)*g' ; p_new = p - rate*lr*v_OLD
 * (simultaneous Theano update: the OLD velocity moves p).  lr is read from *d_lr.
 * Then maxnorm (0 = off): ndim 1 -> clip ; ndim 2 (rows x cols) -> per-column L2 ;
 * ndim 4 (d0 x rest) -> per-d0 L2, scale (1e-7+clip(n,0,M))/(1e-7+n).                    */
int tn_sgd_update(tn_ctx* ctx, float* p, float* v, const float* g, size_t n,
                  float momentum, float rate, const float* d_lr, float L1, float L2, float gscale);
int tn_maxnorm(tn_ctx* ctx, float* p, int ndim, int d0, int rest, float maxnorm);
/* The same projection for EVERY tensor of the net in one call (layer.py:88-103 runs once per parameter): the 1-D and
 * 4-D tensors -- biases and conv kernels, a few hundred to a few thousand values each -- share ONE launch instead
 * of one launch per tensor (13 launch-bound kernels per step of wide6.prms), 2-D tensors take tn_maxnorm's two-pass
 * path.  h_segs: HOST array of nseg <= 32 descriptors.                                                     */
typedef struct tn_mn_seg {
    float* p;
    int32_t ndim, d0, rest;
    float maxnorm;
} tn_mn_seg;
int tn_maxnorm_multi(tn_ctx* ctx, const tn_mn_seg* h_segs, int nseg);
/* The update of EVERY parameter tensor of the net in one launch, in the form the step's schedule needs -- ONE entry
 * point, `mode` selects the form (all of them: Layer.get_updates, layer.py:70-107, bit-identical weight trajectories):
 *
 *   TN_UPD_PLAIN    d_segs = device array of nseg tn_sgd_seg; max_n = largest n among them (sizes the grid).
 *                   g' = g*gscale + L1*sign(p) + 2*L2*p ; v_new = m*v + (1-m)*g' ; p_new = p - rate*lr*v_OLD.
 *   TN_UPD_LAZY     PLAIN that also ENDS a tn_defer_reductions window: a segment whose gradient is still a stack of
 *                   deferred partial slabs sums them on the fly (same order as the reduction launch), stores the
 *                   gradient and applies the update -- one launch less per step.  h_segs = HOST copy of d_segs (matches
 *                   pending sums to segments; the others are finished by the ordinary reduction launch first).
 *                   nseg <= 32.
 *   TN_UPD_DELAYED  data-parallel "delayed" schedule: layer.py:82-86 applies the OLD velocity, so the weights of step
 *                   t+1 do not depend on the gradient of step t and its all-reduce may overlap the whole next step.
 *                   seg.g = the REDUCED gradient of the PREVIOUS step; flags 1: v = m v + (1-m) g, then p -= rate*lr*v
 *                   (steady state); 2: p only (first delayed step); 3: v only (leaving the schedule).  No L1 / L2
 *                   terms (they need the weights the gradient was taken at), no cost rider.
 *   TN_UPD_PIPE     pipelined single-GPU schedule: two steps in flight on the context's two streams, each with its own
 *                   weights / activations / gradients.  d_segs / h_segs = tn_pipe_seg (device array / host copy or
 *                   NULL): p = the stepping stream's own copy, psrc = the other stream's copy (p_{t-1}, read-only
 *                   there), g = the stepping stream's gradient of two steps ago, v shared; flags bit 0 = update v (0
 *                   for the first two steps: no gradient yet).  Also CLOSES the stream's parked tn_defer_reductions
 *                   window like TN_UPD_LAZY (h_segs != NULL, nseg <= 32).  gscale is not used.
 *
 * d_step != NULL: *d_step += step_inc in the same launch (the RNG step counter; step_inc must be 1 outside
 * TN_UPD_PIPE).  rowloss != NULL: one more block computes *d_cost = cost_scale * sum(rowloss[0:nrow]) in a fixed order
 * -- the minibatch cost tt.mean(nll) of outlayers.py:50-51 (TN_UPD_PIPE: of the stream's PREVIOUS step) -- instead
 * of a reduction kernel of its own; with nseg == 0 the launch is that block alone.                                 */
typedef struct tn_sgd_seg {
    float* p;
    float* v;
    const float* g;
    uint64_t n;
    float momentum, rate, L1, L2;
} tn_sgd_seg;
typedef struct tn_pipe_seg {
    float* p;
    const float* psrc;
    float* v;
    const float* g;
    uint64_t n;
    float momentum, rate;
} tn_pipe_seg;
#define TN_UPD_PLAIN 0
#define TN_UPD_LAZY 1
#define TN_UPD_DELAYED 2
#define TN_UPD_PIPE 3
int tn_sgd_update_net(tn_ctx* ctx, int mode, const void* d_segs, const void* h_segs, int nseg, size_t max_n,
                      const float* d_lr, float gscale, uint32_t* d_step, uint32_t step_inc, int flags,
                      const float* rowloss, int nrow, float cost_scale, float* d_cost);
/* tn_sgd_update_net followed by tn_maxnorm_multi(h_mn, nmn) -- the whole update expression of layer.py:82-103 -- as one
 * call: in the TN_UPD_LAZY / TN_UPD_PIPE forms the update launch walks a 2-D max-norm tensor in tn_maxnorm's tiles and
 * leaves its column sums of squares (same partial sums, same order: same bits), so the matrix is not read a second
 * time; only the rescaling pass (which touches nothing while every column is within the bound) follows.  Tensors the
 * walk cannot take, and the other modes, run the two calls back to back.  nmn <= 32.                               */
int tn_sgd_update_net_maxnorm(tn_ctx* ctx, int mode, const void* d_segs, const void* h_segs, int nseg, size_t max_n,
                              const float* d_lr, float gscale, uint32_t* d_step, uint32_t step_inc, int flags,
                              const float* rowloss, int nrow, float cost_scale, float* d_cost, const tn_mn_seg* h_mn,
                              int nmn);

/* ---- elastic input stage (replaces ElasticLayer's graph; inlayers.py:63-144) ----
 * draws layout (float32, device): [0:2] translation u(-1,1) ; [2:4] origin u(.25,.75) ;
 * [4:6] zoom u(-1,1) ; [6] theta u(-1,1) ; [7] pad ; [8 : 8+2hw] N(0,1) noise planes.
 * tn_elastic_draws fills it from Philox (seed, step); parity runs upload it instead.    */
size_t tn_elastic_draws_count(int h, int w);
int tn_elastic_draws(tn_ctx* ctx, float* draws, int h, int w, uint64_t seed,
                     uint32_t step, const uint32_t* d_step);
/* Field -> sample map (one per call, shared by the whole batch; coordinates in float64
 * like the Theano CPU path).  map_idx[h*w] int32 = top*w+left (nearest: the rounded
 * pixel), map_fy/map_fx[h*w] = bilinear fractions (unused for nearest).  target (2hw
 * float64, may be NULL) receives the un-clipped coordinates (debugout, inlayers.py:146). */
int tn_elastic_field(tn_ctx* ctx, const float* draws, int h, int w,
                     double translation, double zoom, double magnitude, int sigma, double angle,
                     int nearest, int32_t* map_idx, float* map_fy, float* map_fx, double* target);
/* out[n,c,:] = resample(invert ? 1-x : x) then flip noise: with flipmask (uint8, N*C*h*w)
 * if given, else Philox(seed, step, global element index) < pflip, else none (pflip=0).
 * x rows are read at x_row0 + (d_row0 ? *d_row0 : 0) + n ; row_global0 is the global
 * index of the first row (for sharding-independent noise).  map_idx == NULL -> identity. */
/* tn_elastic_draws + tn_elastic_field in ONE launch: every block regenerates the (tiny) draws in
 * LDS from Philox (seed, step [+ *d_step]); draws_out (may be NULL for h*w small enough to fit
 * LDS) receives the same values tn_elastic_draws would have written.                         */
int tn_elastic_field_gen(tn_ctx* ctx, float* draws_out, uint64_t seed, uint32_t step,
                         const uint32_t* d_step, int h, int w, double translation, double zoom,
                         double magnitude, int sigma, double angle, int nearest, int32_t* map_idx,
                         float* map_fy, float* map_fx, double* target);
/* tn_elastic_apply fused into the forward of the conv block that consumes it (single-channel
 * images, f == 3, p == 2, at most 16 filters): out (N,1,h,w) is still written (the backward pass
 * reads it), y / mask are tn_convpool_fwd_mask's outputs.  Arguments as in the two ops.        */
int tn_elastic_convpool_supported(int h, int w, int K, int f, int pad_lo, int Ho, int Wo, int p, int Hp,
                                  int Wp);
int tn_elastic_convpool_fwd_mask(tn_ctx* ctx, const float* x, int64_t x_row0, const int64_t* d_row0,
                                 float* out, int N, int h, int w, int invert, int nearest,
                                 const int32_t* map_idx, const float* map_fy, const float* map_fx,
                                 float pflip, const uint8_t* flipmask, uint64_t seed, uint32_t step,
                                 const uint32_t* d_step, int64_t row_global0, const float* W,
                                 const float* b, float* y, uint8_t* mask, int K, int f, int pad_lo, int Ho,
                                 int Wo, int p, int Hp, int Wp, int act, float act_param);

/* Rider: the same field computation, not launched but left with the context; the next paired
 * GEMM launch (tn_fc_bwd) carries it as extra blocks, so it costs no kernel boundary of its own.
 * step is the offset added to *d_step (1 = the minibatch after the one in flight).  tn_rider_pending
 * tells whether it is still waiting, tn_rider_cancel drops it (the caller then runs tn_step_tail or
 * tn_elastic_field_gen itself).                                                              */
int tn_rider_elastic_field(tn_ctx* ctx, float* draws_out, uint64_t seed, uint32_t step,
                           const uint32_t* d_step, int h, int w, double translation, double zoom,
                           double magnitude, int sigma, double angle, int nearest, int32_t* map_idx,
                           float* map_fy, float* map_fx, double* target);
int tn_rider_pending(tn_ctx* ctx);
int tn_rider_cancel(tn_ctx* ctx);
/* The closing launch of a training step: tn_sgd_update_net in TN_UPD_PLAIN mode (without the counter increment)
 * and tn_elastic_field_gen for the NEXT minibatch side by side in one kernel -- the field depends only
 * on *d_step, which the caller has already advanced (tn_defer_flush_step).  Arguments as in the two. */
int tn_step_tail(tn_ctx* ctx, const tn_sgd_seg* d_segs, int nseg, size_t max_n, const float* d_lr,
                 float gscale, const float* rowloss, int nrow, float cost_scale, float* d_cost,
                 float* draws_out, uint64_t seed, const uint32_t* d_step, int h, int w,
                 double translation, double zoom, double magnitude, int sigma, double angle, int nearest,
                 int32_t* map_idx, float* map_fy, float* map_fx, double* target);
int tn_elastic_apply(tn_ctx* ctx, const float* x, int64_t x_row0, const int64_t* d_row0,
                     float* out, int N, int C, int h, int w, int invert, int nearest,
                     const int32_t* map_idx, const float* map_fy, const float* map_fx,
                     float pflip, const uint8_t* flipmask, uint64_t seed, uint32_t step,
                     const uint32_t* d_step, int64_t row_global0);
/* Backward of tn_elastic_apply for an ElasticLayer in the MIDDLE of a net (neuralnet.py:132-142; Theano
 * differentiates through the gather, the inversion and the flip): dx[n,c,src] = sum_p g[n,c,p] * w(p,src)
 * * (invert ? -1 : 1) * (flipped(p) ? -1 : 1), then * act'(prev_a) of the layer below (NULL: none).  The flip
 * noise is regenerated from the same (seed, step, global index) as in the forward.                     */
int tn_elastic_apply_bwd(tn_ctx* ctx, const float* g, float* dx, int N, int C, int h, int w, int invert, int nearest,
                         const int32_t* map_idx, const float* map_fy, const float* map_fx, float pflip,
                         const uint8_t* flipmask, uint64_t seed, uint32_t step, const uint32_t* d_step,
                         int64_t row_global0, const float* prev_a, int prev_act, float prev_act_param);

/* ---- ColorLayer (replaces color.py:9-52) ----
 * fac[(n*C+c)*3 + k] = exp(ln(balance|gamma|gamma) * u_k), u_k ~ U(-1,1): three random variables of shape
 * (N, C).  draws (float32, [3][N][C]) injects the uniforms (parity runs); NULL -> Philox(seed, step + *d_step,
 * global image index * C + c).  tn_color_apply: out = x/maxval*b -> clip(0,1) -> **g1 -> 1-(1-.)**g2 -> *maxval,
 * rows read from x_row0.  tn_color_apply_bwd: dx = g * d out/d x (Clip's gradient is inclusive) * act'(prev_a). */
int tn_color_factors(tn_ctx* ctx, float* fac, int N, int C, double balance, double gamma, const float* draws,
                     uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0);
int tn_color_apply(tn_ctx* ctx, const float* x, int64_t x_row0, const float* fac, float* out,
                   int N, int C, int hw, float maxval);
int tn_color_apply_bwd(tn_ctx* ctx, const float* x, int64_t x_row0, const float* fac, const float* g, float* dx, int N,
                       int C, int hw, float maxval, const float* prev_a, int prev_act, float prev_act_param);

/* ---- aux-input layers (replaces auxiliary.py:14-160; their two small dense maps run on tn_fc_*) ----
 * tn_aux_mix: LocationInfo's input (:27-36).  aux (N, 2, d) float32, rows from row0: train: out[n,:] =
 * boost * (aux[n,0,:]*u_n + aux[n,1,:]*(1-u_n)), u_n ~ U(0,1) (u_inj (B) injected, or Philox(seed, step + *d_step,
 * row_global0 + n)); test: boost * mean of the two.  tn_copy_cols: dst[n, col_dst+j] = src[n, col_src+j] for j <
 * ncols (* act'(prev_a[n, col_dst+j])): the concatenation of AuxConcatLayer and the split of its gradient.   */
int tn_aux_mix(tn_ctx* ctx, const float* aux, int64_t row0, float* out, int B, int d, float boost, int train,
               const float* u_inj, uint64_t seed, uint32_t step, const uint32_t* d_step, int64_t row_global0);
int tn_copy_cols(tn_ctx* ctx, const float* src, int ld_src, int col_src, float* dst, int ld_dst, int col_dst, int ncols,
                 int B, const float* prev_a, int prev_act, float prev_act_param);

/* extras/deformer.py:7-18 -- per-IMAGE deformation, in place semantics of Deformer:
 * trans = indices + scale*noise ; each plane gaussian_filter(sigma, truncate 2, nearest) ;
 * bilinear map_coordinates(mode constant, cval).  noise (N,2,h,w) float32 U(-1,1) given,
 * or NULL -> Philox(seed, image index).  imgs float32 (N,h,w) -> out.                    */
int tn_deformer_transform(tn_ctx* ctx, const float* imgs, float* out, int N, int h, int w,
                          double scale, double sigma, double cval, const float* noise,
                          uint64_t seed, int64_t img_global0);

/* ---- minibatch gather (replaces x_data[indx]; neuralnet.py:228-234) ---- */
int tn_gather_rows(tn_ctx* ctx, const void* src, const int32_t* d_index, void* dst,
                   int nrows, size_t row_bytes);

/* ---- data-parallel gradient exchange over RCCL/xGMI (new) ---- */
#define TN_UNIQUE_ID_BYTES 128
int tn_comm_unique_id(tn_ctx* ctx, void* id128);                 /* rank 0 */
int tn_comm_init(tn_ctx* ctx, const void* id128, int rank, int world);
int tn_comm_destroy(tn_ctx* ctx);
int tn_allreduce_sum(tn_ctx* ctx, float* buf, size_t n);         /* in place, on the ctx stream */
int tn_allreduce_max(tn_ctx* ctx, float* buf, size_t n);
/* The same sum on the context's COMMUNICATION stream (a third stream, so that a bucket of gradients can travel while
 * the compute stream carries on with the backward pass; SURVEY.md 8e "bucket by layer ... to overlap with conv
 * backward"): ordered behind everything enqueued so far on the current compute stream and behind every earlier
 * collective of this entry point (one stream = one order of collectives on the communicator, the same on every rank).
 * done_event (tn_event_create; may be NULL) is recorded on the communication stream behind the collective: the
 * consumer -- the update that opens the stream's next step -- waits for it with tn_event_wait.  tn_sync also waits
 * for the communication stream.                                                                                  */
int tn_allreduce_sum_async(tn_ctx* ctx, float* buf, size_t n, void* done_event);
/* The same in-place sum as a DIRECT reduce-scatter + all-gather (ncclReduceScatter then ncclAllGather, in place: rank r
 * owns elements [r q, (r+1) q), q = n / world; the n % world elements behind them travel in a small all-reduce).
 * MI355X's xGMI is fully connected, so for the large buckets (wide6: 67 MB of dense-layer gradients)
 * the two half-collectives move 2 (S / world) per link pair against a ring's 2 (world - 1) / world * S through every
 * link in turn.  Every element is summed once, at its owner, and broadcast: all ranks hold the same bits.
 * on_comm_stream != 0: ordered and signalled like tn_allreduce_sum_async (done_event may be NULL); 0: on the ctx
 * stream like tn_allreduce_sum (done_event ignored).  theanet_amd/comm.py picks the form per bucket (TN_DP_ALGO,
 * TN_DP_RSAG_MIN_BYTES); the reference has no counterpart (single process; the batch mean of outlayers.py:50-51 is what
 * makes the sum of shard gradients the gradient).                                                                   */
int tn_allreduce_sum_rsag(tn_ctx* ctx, float* buf, size_t n, int on_comm_stream, void* done_event);
int tn_axpby(tn_ctx* ctx, float* y, const float* x, size_t n, float a, float b); /* y = a*x + b*y */

#ifdef __cplusplus
}
#endif
#endif /* THEANET_HIP_H */
