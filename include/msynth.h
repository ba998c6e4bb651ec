/*
 * msynth.h -- C ABI of libmsynth_hip.so: the MI355X (gfx950) kernels under the
 * featuresynth stage-2 mel->waveform GAN hot path.
 *
 * The reference (JohnVinyard/music-synthesis) has no native layer: its hot path
 * is a sequence of stock PyTorch nn.Module / functional calls.  Each entry point
 * below therefore replaces the ATen op that the cited reference line invokes;
 * the Python side (music-synthesis_amd/featuresynth/_ops) binds them with ctypes
 * and is the only caller.  INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - all tensors are fp32, contiguous, (B, C, L) exactly as PyTorch lays them out;
 *   - every pointer is a DEVICE pointer owned by the caller (torch's allocator);
 *     the library never allocates or frees device memory: scratch comes from the
 *     caller-provided workspace (query its size with the *_workspace_bytes calls);
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); no call
 *     synchronises the device, so every call is hipGraph-capture safe;
 *   - return value: MS_OK (0) or a negative ms_status; nothing is thrown;
 *   - re-entrant; no global state.
 */
#ifndef MSYNTH_H
#define MSYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSYNTH_VERSION 200 /* 0.2.0 */

typedef void* ms_stream_t; /* hipStream_t */

enum ms_status {
    MS_OK = 0,
    MS_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, inconsistent shapes */
    MS_ERR_UNSUPPORTED = -2, /* shape/mode outside what the kernels implement */
    MS_ERR_WORKSPACE = -3,   /* workspace missing or too small */
    MS_ERR_LAUNCH = -4,      /* HIP launch failure (hipGetLastError) */
    MS_ERR_COMM = -5         /* RCCL missing, or an RCCL call failed (see ms_comm_last_error) */
};

enum ms_act { MS_ACT_NONE = 0, MS_ACT_LRELU = 1, MS_ACT_TANH = 2 };
enum ms_pad_mode { MS_PAD_ZERO = 0, MS_PAD_REFLECT = 1 };

int ms_version(void);
const char* ms_status_string(int status);

/* nn.Conv1d geometry.  w is (Cout, Cin/groups, K).  Lout = (Lin + 2*pad - dil*(K-1) - 1)/stride + 1 */
typedef struct ms_conv1d_desc {
    int32_t B, Cin, Lin, Cout, K, stride, pad, dil, groups;
    int32_t pad_mode; /* ms_pad_mode; REFLECT = nn.ReflectionPad1d(pad) fused in front */
    int32_t act;      /* ms_act fused behind the conv */
    float slope;      /* LeakyReLU negative slope */
    int32_t in_act;   /* ms_act applied to x on load, i.e. an activation layer IN FRONT of the conv
                         (NONE or LRELU; the pre-activation ResnetBlock / LeakyReLU -> ConvTranspose1d
                         of experiment/realmelgan.py:35-37,63-66) */
} ms_conv1d_desc;

int ms_conv1d_out_len(const ms_conv1d_desc* d);

/*
 * y = residual + act(conv1d(x, w) + bias)
 * Replaces nn.Conv1d (+ F.leaky_relu / nn.LeakyReLU / nn.Tanh, + the ResidualAtom skip add):
 *   generator/full.py:23-25,43-44 ; util/modules.py:358-365,384-388 ;
 *   discriminator/full.py:14-22,36-39.
 * bias, residual, y_act may be NULL.  If y_act != NULL it receives act(conv+bias) WITHOUT the
 * residual (what the backward pass needs to rebuild the LeakyReLU mask of a ResidualAtom).
 */
int ms_conv1d_fwd(const ms_conv1d_desc* d, const float* x, const float* w, const float* bias,
                  const float* residual, float* y, float* y_act, void* workspace,
                  size_t workspace_bytes, ms_stream_t stream);

/*
 * gx = gx_add + conv1d_backward_input(gy * act'(y_act), w)
 * (autograd of the op above w.r.t. x).  y_act NULL => gy is already w.r.t. the pre-activation.
 * gx_add NULL => no add.  pad_mode must be ZERO.
 */
int ms_conv1d_bwd_data(const ms_conv1d_desc* d, const float* gy, const float* y_act,
                       const float* w, const float* gx_add, float* gx, void* workspace,
                       size_t workspace_bytes, ms_stream_t stream);

/*
 * gw = beta*gw + conv1d_backward_weight(x, gy * act'(y_act)); gb likewise (gb may be NULL).
 * beta is 0 (overwrite) or 1 (accumulate: the shared FullDiscriminator is applied at 3 scales,
 * discriminator/melgan.py:16-24, so its weight grads sum over scales).
 */
int ms_conv1d_bwd_weight(const ms_conv1d_desc* d, const float* x, const float* gy,
                         const float* y_act, float* gw, float* gb, float beta, void* workspace,
                         size_t workspace_bytes, ms_stream_t stream);

/*
 * One conv layer applied, with the SAME weights, to up to MS_CONV_PARTS_MAX inputs of different batch size / length -- the
 * reference's multi-scale discriminator runs one FullDiscriminator on x, pool(x), pool(pool(x)) (discriminator/melgan.py:13-27,
 * full.py:13-22).  Three calls of ms_conv1d_* give the same results; here the parts share ONE launch where a kernel takes them
 * (the grouped k41 / stride-4 layers, the 1024 -> 1024 k5 layer on its weight image): the pooled scales are too small to fill
 * the chip on their own, and the weight gradient of the shared parameters is summed over the parts by the launch's one
 * reduction instead of by extra adds.  Where no parts kernel applies the entry points run the parts one after the other on
 * `stream` (ms_conv1d_parts_launches tells which).  d->B and d->Lin are ignored: they are taken per part.
 *   fwd         y[i] = act(conv(x[i], w) + bias);   image: the layer's forward weight image (ms_conv1d_img_pack) or NULL
 *   bwd_data    gx[i] = conv_backward_input(gy[i] * act'(y_act[i]), w) (+ gx_add[i]);   image_bwd likewise
 *   bwd_weight  gw = beta gw + sum_i conv_backward_weight(x[i], gy[i] * act'(y_act[i])), gb likewise
 * Pointers of unused roles / parts may be NULL.
 */
#define MS_CONV_PARTS_MAX 3
typedef struct ms_conv1d_parts {
    int32_t count;                               /* 1 .. MS_CONV_PARTS_MAX */
    int32_t reserved;
    int32_t B[MS_CONV_PARTS_MAX];                /* batch rows of part i */
    int32_t Lin[MS_CONV_PARTS_MAX];              /* input length of part i */
    const float* x[MS_CONV_PARTS_MAX];           /* fwd, bwd_weight: (B[i], Cin, Lin[i]) */
    float* y[MS_CONV_PARTS_MAX];                 /* fwd: (B[i], Cout, Lout[i]) */
    const float* gy[MS_CONV_PARTS_MAX];          /* bwd_data, bwd_weight: gradient w.r.t. the output */
    const float* y_act[MS_CONV_PARTS_MAX];       /* bwd_data, bwd_weight: the saved output (ignored when d->act is NONE) */
    const float* gx_add[MS_CONV_PARTS_MAX];      /* bwd_data: optional addend */
    float* gx[MS_CONV_PARTS_MAX];                /* bwd_data: (B[i], Cin, Lin[i]) */
} ms_conv1d_parts;
/* which: 0 fwd, 1 bwd_data, 2 bwd_weight.  -> number of kernel launch groups the call issues: 1 = one launch takes all
 * parts, count = part by part; < 0: an ms_status error (invalid description). */
int ms_conv1d_parts_launches(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, int which, int with_image);
size_t ms_conv1d_parts_workspace_bytes(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, int which, int with_image);
int ms_conv1d_parts_fwd(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, const float* w, const float* bias,
                        const void* image, void* workspace, size_t workspace_bytes, ms_stream_t stream);
int ms_conv1d_parts_bwd_data(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, const float* w, const void* image_bwd,
                             void* workspace, size_t workspace_bytes, ms_stream_t stream);
int ms_conv1d_parts_bwd_weight(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, float* gw, float* gb, float beta,
                               void* workspace, size_t workspace_bytes, ms_stream_t stream);

/*
 * `count` independent ms_conv1d_bwd_weight calls in one entry.  The six k3 convs of a ResidualStack
 * (util/modules.py:391-405: same channels and length, dilations 1/3/9) are issued as ONE launch pair when
 * their geometry agrees -- a single layer's weight gradient leaves every workgroup a contraction too short
 * to amortise its prologue, slab write and reduce launch.  Any other combination runs entry by entry.
 * Results are identical to the single calls up to fp32 summation order of the split-K slices.
 */
#define MS_WGRAD_MULTI_MAX 8
typedef struct ms_wgrad_multi_desc {
    int32_t count;
    int32_t reserved;
    ms_conv1d_desc conv[MS_WGRAD_MULTI_MAX];
    const float* x[MS_WGRAD_MULTI_MAX];
    const float* gy[MS_WGRAD_MULTI_MAX];
    const float* y_act[MS_WGRAD_MULTI_MAX];     /* may be NULL per entry */
    float* gw[MS_WGRAD_MULTI_MAX];
    float* gb[MS_WGRAD_MULTI_MAX];              /* may be NULL per entry */
    float beta[MS_WGRAD_MULTI_MAX];             /* 0 or 1 per entry */
    /* optional per entry (both or neither; all entries or none): MS_ATOM_AMAX_N upper bounds each whose maximum bounds
     * |x| / |gy| over the whole tensor (ms_residual_atom_fwd / _bwd_data publish them).  With them the batched launch runs
     * on block-scaled two-piece fp16 operands (three MFMA products per multiply), without on exact three-piece bf16 (six). */
    const float* xmax[MS_WGRAD_MULTI_MAX];
    const float* gmax[MS_WGRAD_MULTI_MAX];
    /* optional (all entries or none): SIGN WORDS of the activation in place of y_act (ms_residual_atom_fwd_signs): the
     * LeakyReLU derivative only needs "y_act > 0".  Only the batched kernels read them: MS_ERR_UNSUPPORTED otherwise (ask
     * ms_residual_stack_signs_supported first).  y_act[i] is then ignored. */
    const uint16_t* y_signs[MS_WGRAD_MULTI_MAX];
} ms_wgrad_multi_desc;
size_t ms_conv1d_bwd_weight_multi_workspace_bytes(const ms_wgrad_multi_desc* d);
int ms_conv1d_bwd_weight_multi(const ms_wgrad_multi_desc* d, void* workspace, size_t workspace_bytes,
                               ms_stream_t stream);

/*
 * Fused ResidualAtom forward (util/modules.py:350-388):
 *     y = x + lrelu(conv1d(lrelu(conv1d(x, w0, b0, padding = dil, dilation = dil)), w1, b1, padding = 1))
 * both convs k = 3, C -> C channels, zero padding, in ONE launch: the intermediate activation stays on chip.
 * fp32 in / fp32 accumulate / fp32 out; the operands reach the 16-bit matrix pipe as block-scaled two-piece fp16 (three
 * products per multiply, an fp32 FMA chain's accuracy; MSYNTH_ATOM_NP=3: exact three-piece bf16, bitwise equal to two
 * ms_conv1d_fwd calls).
 *   ms_residual_atom_pack_multi  splits the fp32 weights of up to MS_ATOM_PACK_MAX atoms once into the kernel's
 *                                fragment-ordered piece images (ms_residual_atom_image_bytes(C) bytes each, caller-
 *                                owned, 16-byte aligned); call it again whenever the weights changed (once per step)
 *   ms_residual_atom_fwd         t / y_act (both or neither): training additionally stores t = lrelu(conv_d + b0) and
 *                                y_act = lrelu(conv1 + b1), the activations the backward pass of the atom needs
 *   ms_residual_atom_supported   1 when the fused kernel takes this geometry (C in {32, 64, 128, 256}, L % 4 == 0,
 *                                dil <= 9), else 0: the caller then issues the two convs
 * b0, b1 and image must be 16-byte aligned.
 */
typedef struct ms_atom_desc {
    int32_t B, C, L, dil;
    float slope;      /* LeakyReLU negative slope */
} ms_atom_desc;
#define MS_ATOM_PACK_MAX 16
typedef struct ms_atom_pack_desc {
    int32_t count;
    int32_t reserved;
    int32_t C[MS_ATOM_PACK_MAX];
    const float* w0[MS_ATOM_PACK_MAX];     /* (C, C, 3): the dilated conv */
    const float* w1[MS_ATOM_PACK_MAX];     /* (C, C, 3): the dilation-1 conv */
    void* image[MS_ATOM_PACK_MAX];
    int32_t backward[MS_ATOM_PACK_MAX];    /* 0: image for ms_residual_atom_fwd; 1: for ms_residual_atom_bwd_data */
} ms_atom_pack_desc;
size_t ms_residual_atom_image_bytes(int32_t C);
int ms_residual_atom_supported(const ms_atom_desc* d);
int ms_residual_atom_pack_multi(const ms_atom_pack_desc* d, ms_stream_t stream);
/* amax (optional, 2 * MS_ATOM_AMAX_N floats): per-workgroup largest magnitudes of the launch's two GEMM operands --
 * forward: [0][.] of x, [1][.] of t; backward: [0][.] of gy, [1][.] of gt * lrelu'(t); entries behind the launch's grid are
 * zero -- for a consumer that block-scales these tensors (ms_conv1d_bwd_weight_multi: xmax / gmax).  Only written when
 * ms_residual_atom_publishes_amax() is 1. */
#define MS_ATOM_AMAX_N 1024
int ms_residual_atom_publishes_amax(void);
int ms_residual_atom_fwd(const ms_atom_desc* d, const float* x, const void* image, const float* b0, const float* b1,
                         float* y, float* t, float* y_act, float* amax, ms_stream_t stream);
/*
 * Backward data of the atom in one launch (autograd of the op above w.r.t. x), from the two saved activations:
 *     gt = conv1d_backward_input(gy * lrelu'(y_act), w1)                     (raw: the weight gradient of the dilated conv
 *                                                                            applies lrelu'(t) itself, ms_conv1d_bwd_weight)
 *     gx = gy + conv1d_backward_input(gt * lrelu'(t), w0)
 * image_bwd: packed with backward[i] = 1.  Same arithmetic as the two ms_conv1d_bwd_data calls it replaces.
 * ms_residual_atom_bwd_supported: 1 when the fused kernel takes (and pays for) this geometry.
 */
int ms_residual_atom_bwd_supported(const ms_atom_desc* d);
int ms_residual_atom_bwd_data(const ms_atom_desc* d, const float* gy, const float* y_act, const float* t,
                              const void* image_bwd, float* gt, float* gx, float* amax, ms_stream_t stream);
/*
 * The same pair with SIGN WORDS (r04): the backward pass uses y_act -- and, in its data path, t -- only through
 * "value > 0" (the LeakyReLU derivatives).  ms_residual_atom_fwd_signs stores t (fp32: the second conv's weight gradient
 * multiplies by it) and, instead of y_act, one bit per element of y_act and of t:
 *     signs[((b * (C / 32) + blk) * 2 + h) * L + l],  bit 15 - r  =  value[b, 32 blk + (r & 3) + 8 (r >> 2) + 4 h, l] > 0
 * (16-bit words; ms_residual_atom_sign_words(d) = B * (C / 32) * 2 * L of them per tensor, 0 = not available: the
 * three-piece scheme).  Per atom the forward writes 3 1/16 instead of 4 tensors, the backward reads 1 1/16 instead of 3,
 * ms_conv1d_bwd_weight_multi (y_signs) 4 1/16 instead of 6.  16-byte aligned arrays.
 * ms_residual_stack_signs_supported: 1 when forward, backward data AND the batched weight gradients of a stack of such
 * atoms all take sign words (else save the fp32 activations).
 */
size_t ms_residual_atom_sign_words(const ms_atom_desc* d);
int ms_residual_atom_fwd_signs(const ms_atom_desc* d, const float* x, const void* image, const float* b0, const float* b1,
                               float* y, float* t, uint16_t* t_signs, uint16_t* y_signs, float* amax, ms_stream_t stream);
int ms_residual_atom_bwd_data_signs(const ms_atom_desc* d, const float* gy, const uint16_t* y_signs, const uint16_t* t_signs,
                                    const void* image_bwd, float* gt, float* gx, float* amax, ms_stream_t stream);

/*
 * Fused ResidualStack forward, inference (util/modules.py:391-405: `count` ResidualAtoms back to back, the generator's
 * stacks with dilations 1, 3, 9): y = atom[count-1](... atom[0](x)) in ONE launch, nothing saved -- the values between
 * the atoms never travel to memory (csrc/stack_fused.hip).  images[i]: atom i's forward image (ms_residual_atom_pack_multi,
 * backward = 0; the two-piece scheme, i.e. not under MSYNTH_ATOM_NP=3), b0[i] / b1[i]: its biases; all 16-byte aligned.
 * x and y must not alias.  ms_residual_stack_supported: 1 when the kernel takes this geometry (C in {32, 64},
 * L % 4 == 0, sum(dil + 1) <= 16, dil[0] <= 4), else 0: the caller issues ms_residual_atom_fwd per atom.
 */
#define MS_STACK_MAX 3
typedef struct ms_stack_desc {
    int32_t B, C, L, count;
    int32_t dil[MS_STACK_MAX];
    float slope;      /* LeakyReLU negative slope */
} ms_stack_desc;
int ms_residual_stack_supported(const ms_stack_desc* d);
int ms_residual_stack_signs_supported(const ms_stack_desc* d);
int ms_residual_stack_fwd(const ms_stack_desc* d, const float* x, const void* const* images, const float* const* b0,
                          const float* const* b1, float* y, ms_stream_t stream);

/*
 * Dense k = 5 / stride 1 / padding 2 conv on short rows (L <= 64) with PRE-SPLIT weight images: the discriminator's
 * 1024 -> 1024 layer (discriminator/full.py:19) at its three scales, forward and backward data (csrc/conv5_img.hip).
 * Same operation and accuracy as ms_conv1d_fwd / ms_conv1d_bwd_data on that geometry (other summation order: ~1e-7).
 *   ms_conv1d_img_bytes            bytes of one image (forward and backward images have the same size); 0 = this geometry is
 *                                  not taken (the caller uses ms_conv1d_fwd / ms_conv1d_bwd_data)
 *   ms_conv1d_img_pack             w (Cout, Cin, 5) -> image (caller-owned, 16-byte aligned); backward = 1: the image of the
 *                                  backward-data pass.  Once per weight update; one image serves every batch size / row length
 *   ms_conv1d_img_fwd              y = act(conv1d(x, w) + bias)
 *   ms_conv1d_img_bwd_data         gx = gx_add + conv1d_backward_input(gy * act'(y_act), w)    (y_act, gx_add may be NULL)
 *   ms_conv1d_img_workspace_bytes  which: 0 fwd, 1 bwd_data (split-K slabs)
 */
size_t ms_conv1d_img_bytes(const ms_conv1d_desc* d);
size_t ms_conv1d_img_workspace_bytes(const ms_conv1d_desc* d, int which);
int ms_conv1d_img_pack(const ms_conv1d_desc* d, const float* w, int backward, void* image, ms_stream_t stream);
/* Both images of a layer (forward, backward data) from ONE pass over the weights for their common scale: what a train step
 * needs, one launch fewer than two ms_conv1d_img_pack calls. */
int ms_conv1d_img_pack2(const ms_conv1d_desc* d, const float* w, void* image_fwd, void* image_bwd, ms_stream_t stream);
int ms_conv1d_img_fwd(const ms_conv1d_desc* d, const float* x, const void* image, const float* bias, float* y,
                      void* workspace, size_t workspace_bytes, ms_stream_t stream);
int ms_conv1d_img_bwd_data(const ms_conv1d_desc* d, const float* gy, const float* y_act, const void* image_bwd,
                           const float* gx_add, float* gx, void* workspace, size_t workspace_bytes, ms_stream_t stream);

/* which: 0 fwd, 1 bwd_data, 2 bwd_weight */
size_t ms_conv1d_workspace_bytes(const ms_conv1d_desc* d, int which);

/* Name of the device kernel the dispatch selects for this geometry (which: 0 fwd, 1 bwd_data,
 * 2 bwd_weight) -- lets a profiler line be matched to a layer.  Static string, never NULL. */
const char* ms_conv1d_kernel_name(const ms_conv1d_desc* d, int which);

/* Profiling aid (off by default; never needed for normal operation).  ms_profile_kernels(1) opens a profile session of the
 * calling THREAD: every kernel the library launches from that thread is then bracketed by the dispatch's own begin / end
 * timestamps (hipExtLaunchKernelGGL start / stop events: what rocprofv3 reports as the kernel's duration) and waited for, and
 * the launchers of the templated dense families note what they dispatched.  ms_profile_take() copies the record of the calls
 * made since the previous take into *out and resets it; ms_profile_kernels(0) closes the session.  The session is the only
 * state involved, it is thread-local and owned by the thread that opened it; outside a session launches record nothing.  Not
 * capturable into a hipGraph while on. */
#define MS_PROFILE_NAME_MAX 160
typedef struct ms_profile_record {
    int kernels;                      /* kernel launches since the previous take */
    int products;                     /* matrix-pipe products per fp32 multiply of the kernel noted last: 6 = exact three-piece
                                         bf16 split, 3 = block-scaled two-piece fp16 split, 0 = fp32-input MFMA / vector FMA */
    double device_us;                 /* summed device time of those launches */
    char kernel[MS_PROFILE_NAME_MAX]; /* instantiation noted last, in rocprofv3's spelling without the namespace; "" when
                                         the launcher notes none (then ms_conv1d_kernel_name applies) */
} ms_profile_record;
void ms_profile_kernels(int on);
int ms_profile_take(ms_profile_record* out);

/* Debug aid for test harnesses (no reference counterpart, never called by the product path): installs handlers for SIGSEGV /
 * SIGBUS / SIGABRT / SIGFPE / SIGILL that write the NATIVE backtrace of the faulting thread to stderr and re-raise with the
 * default action -- a fault inside the HIP runtime (e.g. in hipGraphLaunch) otherwise leaves only interpreter frames.
 * Process-wide by nature (signal dispositions); the handler that was installed before (e.g. Python's faulthandler) runs next. */
int ms_debug_install_crash_handler(void);

/* nn.ConvTranspose1d geometry.  w is (Cin, Cout, K).  Lout = (Lin-1)*stride - 2*pad + K */
typedef struct ms_convt1d_desc {
    int32_t B, Cin, Lin, Cout, K, stride, pad;
    int32_t act;
    float slope;
    int32_t in_act;   /* activation in front of the transposed conv (NONE or LRELU) */
} ms_convt1d_desc;

int ms_convt1d_out_len(const ms_convt1d_desc* d);

/* y = act(conv_transpose1d(x, w) + bias).  Replaces generator/full.py:27-28,31-32,35-36,39-40. */
int ms_convt1d_fwd(const ms_convt1d_desc* d, const float* x, const float* w, const float* bias,
                   float* y, void* workspace, size_t workspace_bytes, ms_stream_t stream);
/* gx = conv_transpose1d_backward_input(gy * act'(y_act), w) */
int ms_convt1d_bwd_data(const ms_convt1d_desc* d, const float* gy, const float* y_act,
                        const float* w, float* gx, void* workspace, size_t workspace_bytes,
                        ms_stream_t stream);
/* gw = beta*gw + backward_weight(x, gy * act'(y_act)); gb likewise */
int ms_convt1d_bwd_weight(const ms_convt1d_desc* d, const float* x, const float* gy,
                          const float* y_act, float* gw, float* gb, float beta, void* workspace,
                          size_t workspace_bytes, ms_stream_t stream);
size_t ms_convt1d_workspace_bytes(const ms_convt1d_desc* d, int which);
/*
 * The generator's four upsampling layers (generator/full.py:27-40: 512->256 and 256->128 with k16 / s8 / p4, 128->64 and
 * 64->32 with k4 / s2 / p1) forward on pre-split weight images (csrc/convt_img.hip): same operation as ms_convt1d_fwd.
 *   ms_convt1d_img_bytes            bytes of the image; 0 = geometry not taken (use ms_convt1d_fwd)
 *   ms_convt1d_img_workspace_bytes  split-K slabs of the short-row form (0 for the generator's layers); 16-byte aligned
 *   ms_convt1d_img_pack             w (Cin, Cout, K) -> image (caller-owned, 16-byte aligned); once per weight update
 *   ms_convt1d_img_fwd              y = act(conv_transpose1d(x, w) + bias)
 * Also taken (csrc/convt_fwd_short.hip, same image): kernel 4 / stride 2 / padding 1 on rows of 4 .. 16 positions with >= 256
 * input channels -- the first line convolutions of the stage-1 generator (featuregenerator/upscale.py:85-91).
 */
size_t ms_convt1d_img_bytes(const ms_convt1d_desc* d);
size_t ms_convt1d_img_workspace_bytes(const ms_convt1d_desc* d);
int ms_convt1d_img_pack(const ms_convt1d_desc* d, const float* w, void* image, ms_stream_t stream);
int ms_convt1d_img_fwd(const ms_convt1d_desc* d, const float* x, const void* image, const float* bias, float* y,
                       void* workspace, size_t workspace_bytes, ms_stream_t stream);
/*
 * Transposed-conv BACKWARD DATA on pre-split weight images (csrc/convt_bwd_img.hip): same operation as ms_convt1d_bwd_data for
 * kernel 2S / stride S / padding S/2 (S = 2, 8) on rows of 4 .. 256 input positions (a power of two) -- the generator's two
 * stride-8 layers (generator/full.py:27-32) and the stage-1 generator's line convolutions (featuregenerator/upscale.py:85-97).
 *   ms_convt1d_bwd_img_bytes            bytes of the image; 0 = geometry not taken (use ms_convt1d_bwd_data)
 *   ms_convt1d_bwd_img_workspace_bytes  split-K slabs (16-byte aligned workspace)
 *   ms_convt1d_bwd_img_pack             w (Cin, Cout, K) -> image (caller-owned, 16-byte aligned); once per weight update
 *   ms_convt1d_bwd_img_data             gx = conv(gy * act'(y_act), w) with the mirrored geometry
 */
size_t ms_convt1d_bwd_img_bytes(const ms_convt1d_desc* d);
size_t ms_convt1d_bwd_img_workspace_bytes(const ms_convt1d_desc* d);
int ms_convt1d_bwd_img_pack(const ms_convt1d_desc* d, const float* w, void* image, ms_stream_t stream);
int ms_convt1d_bwd_img_data(const ms_convt1d_desc* d, const float* gy, const float* y_act, const void* image, float* gx,
                            void* workspace, size_t workspace_bytes, ms_stream_t stream);
const char* ms_convt1d_kernel_name(const ms_convt1d_desc* d, int which);

/*
 * F.avg_pool1d(x, kernel_size=4, stride=2, padding=2) with count_include_pad=True
 * (discriminator/melgan.py:22).  x is (rows, Lin) with rows = B*C; Lout = (Lin + 2*2 - 4)/2 + 1.
 */
int ms_avg_pool1d_4_2_2_fwd(const float* x, float* y, int64_t rows, int32_t Lin, ms_stream_t stream);
int ms_avg_pool1d_4_2_2_bwd(const float* gy, const float* gx_add, float* gx, int64_t rows,
                            int32_t Lin, ms_stream_t stream);

/*
 * nn.AvgPool1d(4, stride=2, padding=1, count_include_pad=False) (experiment/realmelgan.py:168-169):
 * the divisor is the number of in-range samples of each window.  Lout = (Lin + 2 - 4)/2 + 1.
 */
int ms_avg_pool1d_4_2_1_fwd(const float* x, float* y, int64_t rows, int32_t Lin, ms_stream_t stream);
int ms_avg_pool1d_4_2_1_bwd(const float* gy, const float* gx_add, float* gx, int64_t rows,
                            int32_t Lin, ms_stream_t stream);
/* F.avg_pool1d(x, k): window = stride = k, no padding (the conditioning branch of the weight-normed MelGAN's
 * discriminators, experiment/realmelgan.py:150-151).  x (rows, Lin) -> y (rows, Lin / k). */
int ms_avg_pool1d_k_fwd(const float* x, float* y, int64_t rows, int32_t Lin, int32_t k, ms_stream_t stream);
int ms_avg_pool1d_k_bwd(const float* gy, float* gx, int64_t rows, int32_t Lin, int32_t k, ms_stream_t stream);

/*
 * torch.nn.utils.weight_norm (experiment/realmelgan.py:24-29): w[r, :] = g[r] * v[r, :] / ||v[r, :]||_2
 * for a (rows, cols) view of the parameter (rows = dim 0 of the weight).  Backward:
 * gv = (g/||v||) * (gw - (gw . v^) v^),  gg = gw . v^   with v^ = v/||v||.
 */
int ms_weight_norm_fwd(const float* v, const float* g, float* w, int32_t rows, int32_t cols,
                       ms_stream_t stream);
int ms_weight_norm_bwd(const float* v, const float* g, const float* gw, float* gv, float* gg,
                       int32_t rows, int32_t cols, float beta, ms_stream_t stream);

/*
 * The same for every weight-normed layer of a network in ONE launch (the reference re-derives all
 * weights on each forward through the weight_norm pre-hooks, realmelgan.py:24-29: 63 tiny launches
 * per forward otherwise).  Tensor i is a (rows[i], cols[i]) view; the descriptor is passed by value.
 * Forward writes out[i]; backward reads out[i] as gw and writes gv[i], gg[i] (beta as above).
 */
#define MS_WN_MULTI_MAX 64
typedef struct ms_wn_multi_desc {
    int32_t count;
    int32_t reserved;
    const float* v[MS_WN_MULTI_MAX];
    const float* g[MS_WN_MULTI_MAX];
    float* out[MS_WN_MULTI_MAX];       /* fwd: w (written);  bwd: gw (read) */
    float* gv[MS_WN_MULTI_MAX];        /* bwd only */
    float* gg[MS_WN_MULTI_MAX];        /* bwd only */
    int32_t rows[MS_WN_MULTI_MAX];
    int32_t cols[MS_WN_MULTI_MAX];
} ms_wn_multi_desc;
int ms_weight_norm_multi_fwd(const ms_wn_multi_desc* d, ms_stream_t stream);
int ms_weight_norm_multi_bwd(const ms_wn_multi_desc* d, float beta, ms_stream_t stream);

/* gpre = gy * act'(y_act), elementwise (stand-alone form of the fused modifier above) */
int ms_act_bwd(const float* y_act, const float* gy, float* gpre, int64_t n, int32_t act,
               float slope, ms_stream_t stream);
/* out = a + b */
int ms_add(const float* a, const float* b, float* out, int64_t n, ms_stream_t stream);
/* out = act(a + b): the residual layer of DilatedStack -- the activation is applied OVER the skip sum
 * (util/modules.py:131-134: x = activation(z + x)); its backward is ms_act_bwd on the saved output */
int ms_add_act(const float* a, const float* b, float* out, int64_t n, int32_t act, float slope,
               ms_stream_t stream);

/*
 * Line movement of the stage-1 generator's ConvTranspose2d layers (reference featuregenerator/upscale.py:85-99; the host
 * side is util/modules.py:HipConvTranspose2d).  Activations are lines (B, H, C, W); output row sH*q + phase of the 2-D
 * transposed conv is a 1-D transposed conv over the channels of the `taps` input rows q + dy[phase][tap]:
 *   ms_lines_stack       out (phases, B*H, taps*C, W): out[ph][(b,q)][j*C + c][w] = x[b][q + dy[ph][j]][c][w], 0 outside
 *   ms_lines_fold        gx (B, H, C, W) = its transpose applied to gstack (phases, B*H, taps*C, W); terms are summed
 *                        phase-major, then by tap (deterministic)
 *   ms_lines_interleave  inverse = 0: src (phases, rows, n) -> dst (rows, phases, n)  (phase outputs -> image row order)
 *                        inverse = 1: src (rows, phases, n) -> dst (phases, rows, n)
 * 16-byte aligned buffers, C*W and n multiples of 4.
 */
#define MS_LINES_MAX_PHASES 2
#define MS_LINES_MAX_TAPS 3
typedef struct ms_lines_desc {
    int32_t B, H, C, W;
    int32_t phases, taps;
    int32_t dy[MS_LINES_MAX_PHASES * MS_LINES_MAX_TAPS];   /* dy[phase * MS_LINES_MAX_TAPS + tap] */
} ms_lines_desc;
int ms_lines_stack(const ms_lines_desc* d, const float* x, float* out, ms_stream_t stream);
int ms_lines_fold(const ms_lines_desc* d, const float* gstack, float* gx, ms_stream_t stream);
int ms_lines_interleave(const float* src, float* dst, int64_t rows, int32_t phases, int64_t n, int32_t inverse,
                        ms_stream_t stream);

/*
 * Losses (loss/loss.py).  Every *_fwd writes ONE float to `out` (device); every *_bwd reads
 * the upstream scalar gradient from the device pointer `gout` (so no host sync is needed) and
 * multiplies it by the host constant `scale`.
 *   hinge_d : mean(relu(1 - r) + relu(1 + f))          loss.py:17-18
 *   neg_mean: mean(-f)      (hinge generator loss)     loss.py:9-10
 *   l1_mean : mean(|r - f|) (F.l1_loss)                loss.py:62
 *   ls_d / ls_g: least-squares variants                loss.py:5-6,13-14
 * l1_mean_fwd needs workspace (ms_reduce_workspace_bytes(n)).
 */
size_t ms_reduce_workspace_bytes(int64_t n);
int ms_hinge_d_fwd(const float* r, const float* f, int64_t n, float* out, void* workspace,
                   size_t workspace_bytes, ms_stream_t stream);
int ms_hinge_d_bwd(const float* r, const float* f, int64_t n, const float* gout, float scale,
                   float* gr, float* gf, ms_stream_t stream);
int ms_neg_mean_fwd(const float* f, int64_t n, float* out, void* workspace,
                    size_t workspace_bytes, ms_stream_t stream);
int ms_neg_mean_bwd(int64_t n, const float* gout, float scale, float* gf, ms_stream_t stream);
int ms_l1_mean_fwd(const float* r, const float* f, int64_t n, float* out, void* workspace,
                   size_t workspace_bytes, ms_stream_t stream);
/* gf = (accumulate ? gf : 0) + sign(f - r) * scale * (*gout) / n */
int ms_l1_mean_bwd(const float* r, const float* f, int64_t n, const float* gout, float scale,
                   float* gf, int32_t accumulate, ms_stream_t stream);
int ms_ls_g_fwd(const float* j, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                ms_stream_t stream);
int ms_ls_g_bwd(const float* j, int64_t n, const float* gout, float scale, float* gj,
                ms_stream_t stream);
int ms_ls_d_fwd(const float* r, const float* f, int64_t n, float* out, void* workspace,
                size_t workspace_bytes, ms_stream_t stream);
int ms_ls_d_bwd(const float* r, const float* f, int64_t n, const float* gout, float scale,
                float* gr, float* gf, ms_stream_t stream);
/*
 * Feature-matching loss over up to MS_L1_MULTI_MAX tensor pairs in ONE launch pair
 * (mel_gan_feature_loss, loss/loss.py:28-65: 18 L1 means per generator step):
 *   fwd: out[0] = sum_i w[i] * mean(|f_i - r_i|)
 *   bwd: gf_i = sign(f_i - r_i) * w[i] * scale * (*gout) / n[i]     (gf[i] == NULL: skipped)
 * The descriptor is passed by value to the kernels: pointers are device pointers, n[i] > 0.
 */
#define MS_L1_MULTI_MAX 24
typedef struct ms_l1_multi_desc {
    int32_t count;
    int32_t reserved;
    const float* r[MS_L1_MULTI_MAX];
    const float* f[MS_L1_MULTI_MAX];
    float* gf[MS_L1_MULTI_MAX];
    int64_t n[MS_L1_MULTI_MAX];
    float w[MS_L1_MULTI_MAX];
} ms_l1_multi_desc;
size_t ms_l1_mean_multi_workspace_bytes(const ms_l1_multi_desc* d);
int ms_l1_mean_multi_fwd(const ms_l1_multi_desc* d, float* out, void* workspace,
                         size_t workspace_bytes, ms_stream_t stream);
int ms_l1_mean_multi_bwd(const ms_l1_multi_desc* d, const float* gout, float scale,
                         ms_stream_t stream);
/*
 * The same sum AND its gradients in one pass over the maps, for a caller that knows the upstream gradient as a host
 * constant (the loss is the root of a train step's backward pass: train/train.py:36, loss.backward()):
 *     out[0] = sum_i w[i] * mean(|f_i - r_i|),     gf[i] = gconst * w[i] / n[i] * sign(f_i - r_i)   (gf[i] may be NULL)
 * Every map must hold a multiple of 4 elements and be 16-byte aligned (workspace query returns 0 otherwise, the call
 * MS_ERR_UNSUPPORTED: use the two calls above).
 */
size_t ms_l1_mean_multi_fwd_bwd_workspace_bytes(const ms_l1_multi_desc* d);
int ms_l1_mean_multi_fwd_bwd(const ms_l1_multi_desc* d, float* out, float gconst, void* workspace,
                             size_t workspace_bytes, ms_stream_t stream);
/*
 * The GAN terms over the (small) judgement tensors of all discriminator scales in ONE launch each way:
 *   kind MS_JUDGE_HINGE_D:  out[0] = sum_i mean(relu(1 - r_i) + relu(1 + f_i))      (loss/loss.py:17-25)
 *   kind MS_JUDGE_NEG_MEAN: out[0] = sum_i mean(-f_i)                                (loss/loss.py:9, 68-78)
 * n[i] <= MS_JUDGE_MULTI_NMAX (one workgroup folds every tensor in a fixed order: deterministic).
 * Backward writes gr[i] / gf[i] (either may be NULL) = d(out)/d r_i, f_i * gout[0] * scale.
 */
#define MS_JUDGE_MULTI_MAX 8
#define MS_JUDGE_MULTI_NMAX (1 << 20)
#define MS_JUDGE_HINGE_D 0
#define MS_JUDGE_NEG_MEAN 1
typedef struct ms_judge_multi_desc {
    int32_t count;
    int32_t kind;
    const float* r[MS_JUDGE_MULTI_MAX];        /* unused for NEG_MEAN */
    const float* f[MS_JUDGE_MULTI_MAX];
    float* gr[MS_JUDGE_MULTI_MAX];
    float* gf[MS_JUDGE_MULTI_MAX];
    int64_t n[MS_JUDGE_MULTI_MAX];
} ms_judge_multi_desc;
int ms_judge_loss_multi_fwd(const ms_judge_multi_desc* d, float* out, ms_stream_t stream);
int ms_judge_loss_multi_bwd(const ms_judge_multi_desc* d, const float* gout, float scale,
                            ms_stream_t stream);
/* out[0] = sum_i coef[i] * (*terms[i]) for n device scalars laid out contiguously in `terms` */
int ms_weighted_sum(const float* terms, const float* coef, int32_t n, float* out,
                    ms_stream_t stream);

/*
 * torch.optim.Adam step (no weight decay / amsgrad) over one flat fp32 bucket, as configured at
 * experiment/experiment.py:111-117.  `step` is a DEVICE int32 counter: the call increments it and
 * uses the new value for the bias corrections (keeps the call graph-capturable).  Gradients are
 * multiplied by grad_scale first (1/world_size after a summing all-reduce).
 */
int ms_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, float grad_scale, int32_t* step, ms_stream_t stream);

/*
 * Audio2Mel.forward (feature/feature.py:39-59): right-pad (n_fft-hop)/2 zeros, windowed STFT
 * (center=False), magnitude, mel_basis @ magnitude, log10(clamp(., 1e-5)).
 * audio (B, N); window (n_fft); mel_basis (n_mel, n_fft/2+1); out (B, n_mel, frames).
 * n_fft must be a power of two in [64, 4096].
 */
int ms_audio2mel_frames(int32_t N, int32_t n_fft, int32_t hop);
int ms_audio2mel_fwd(const float* audio, int32_t B, int32_t N, const float* window, int32_t n_fft,
                     int32_t hop, const float* mel_basis, int32_t n_mel, float* out,
                     ms_stream_t stream);

/*
 * audio() front-end of the dataset pass (feature/feature.py:64-71): librosa.resample (default 'kaiser_best':
 * resampy's interpolated Kaiser-windowed sinc) and librosa.util.normalize(x) * 0.95.
 * x (rows, n_in) -> y (rows, n_out) with n_out = ceil(n_in * ratio), ratio = target_sr / orig_sr.
 * interp_win / interp_delta: the half window (nwin entries, num_table per zero crossing, already multiplied by
 * min(1, ratio)) and its first differences, built on the host as resampy does.
 * ms_peak_normalize scales every row in place by scale / max|row| (rows of all zeros stay); workspace = rows floats.
 */
int ms_resample_sinc_fwd(const float* x, int32_t rows, int32_t n_in, float* y, int32_t n_out, double ratio,
                         const float* interp_win, const float* interp_delta, int32_t nwin, int32_t num_table,
                         ms_stream_t stream);
int ms_peak_normalize(float* x, int32_t rows, int32_t n, float scale, float* workspace, ms_stream_t stream);

/*
 * Data-parallel gradient exchange (the reference is single-process, so these
 * replace nothing in it: they are what its training loop would call between loss.backward() and
 * optim.step(), train/train.py:37-38,72-73, once the batch is sharded over ranks).
 *
 * One RCCL communicator per process (one process per GPU).  The library binds the RCCL that is already
 * loaded into the process (torch's "nccl" backend IS RCCL on ROCm), else librccl.so.1 from the loader
 * path, at the first ms_comm_* call; nothing links against it at build time.
 *   ms_comm_unique_id   rank 0 fills a MS_COMM_ID_BYTES buffer; the caller ships it to the other ranks
 *                       (file, TCP store, torch.distributed broadcast ... -- bootstrap is the host's job)
 *   ms_comm_init        collective over all ranks; the current HIP device is the rank's GPU
 *   ms_allreduce_f32    in-place SUM over ranks of n floats, enqueued on `stream` (no host sync; the
 *                       flat FlatAdam gradient bucket or a 16-byte aligned slice of it)
 *   ms_comm_destroy     releases the communicator
 * The only global state is the bound RCCL entry points and the text of the last RCCL error.
 */
#define MS_COMM_ID_BYTES 128
typedef struct ms_comm* ms_comm_t;
int ms_comm_unique_id(void* id_out /* MS_COMM_ID_BYTES, host */);
int ms_comm_init(const void* id /* MS_COMM_ID_BYTES, host */, int32_t world, int32_t rank, ms_comm_t* out);
int ms_comm_world(ms_comm_t comm);
int ms_comm_rank(ms_comm_t comm);
int ms_allreduce_f32(ms_comm_t comm, float* buf, int64_t n, ms_stream_t stream);
int ms_comm_destroy(ms_comm_t comm);
const char* ms_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MSYNTH_H */
