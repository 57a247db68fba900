/*
 * gdn_hip.h -- C ABI of libgdn_hip.so, the MI355X (gfx950) kernels behind the
 * GDN-Pytorch hot path.
 *
 * The reference (tjqansthd/GDN-Pytorch) has no FFI: its boundary is Python
 * (nn.Module.forward + loss helpers) and every device op is a PyTorch/cuDNN
 * call.  Each entry point below therefore replaces one *PyTorch call site* of
 * the reference; the file:line cited is that call site (paths relative to
 * /root/reference/src).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - return 0 on success, negative gdn_status on error (gdn_strerror());
 *   - never allocate, never synchronise, never read device memory on the host:
 *     every call is stream-ordered on `stream` (a hipStream_t passed as void*)
 *     and is hipGraph-capture safe; the caller owns every buffer, including
 *     workspaces sized by the matching *_workspace_bytes() query;
 *   - activations are fp32 NHWC: pixel p of an image batch [B,H,W] with C
 *     channels lives at base + p*ld (ld >= C floats, lets a tensor be a channel
 *     slice of a wider one); weights are fp32 in "tap-major" layout
 *     [kh*kw][Cout][Cin] (Cin contiguous) for both Conv2d and ConvTranspose2d;
 *   - stateless and thread-safe: the library owns no stream, event, buffer or cache.
 *     Work that profits from a second stream (gdn_fftconv_bwd's two chains) is split
 *     into phases the CALLER places on streams of its own.
 */
#ifndef GDN_HIP_H
#define GDN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GDN_OK = 0,
    GDN_ERR_BAD_ARG = -1,
    GDN_ERR_UNSUPPORTED = -2,
    GDN_ERR_WORKSPACE = -3,
    GDN_ERR_LAUNCH = -4
} gdn_status;

/* Revision of this header (argument lists, struct layouts).  222: gdn_fftconv_bwd bnb_*, gdn_fftconv_bnb_slots.  221: gdn_clock_probe_*.  220: gdn_conv_dgrad dx_up2x; gdn_bn_apply_up2x; bf16 tile id 12 (conv_ring2_bf16).  219: gdn_conv_wgrad_bf16 cfg 4 (wgrad_ring_bf16); gdn_fftconv_cgemm* measurement hooks; plan overrides in gdn_conv_geom.hints; gdn_gemm_x3_nt_packed / gdn_gemm_x3_ring_workspace_bytes removed (the measured-and-not-wired kernel now lives under tests/diag/).  218: gdn_gemm_x3_tn_splits.  217: gdn_conv_dgrad bnb_*.  216: gdn_conv_c1_fwd Cin.  215: GDN_HINT_NO_WINO_F4.  214: gdn_gemm_x3_nt_packed.  213: gdn_conv_c1_fwd dtypes / gdn_conv_c1_wgrad gw_bf16.  212: GDN_HINT_NO_X3 (replaces the GDN_X3 environment read).  211: gdn_gemm_x3_*.  210: gdn_conv_geom.hints, in_up2x / dx_up2x.  A binding checks it
 * for equality at load time (gdn_amd/_lib.py: ABI_VERSION). */
int gdn_version(void);
const char* gdn_strerror(int status);
/* Fills name[] with the HIP device name of the current device; returns CU count (<0 on error). */
int gdn_device_info(char* name, int name_len);

/* ------------------------------------------------------------------------
 * Convolution geometry, in torch.nn semantics.
 *   transposed == 0 : nn.Conv2d(Cin, Cout, k, stride, pad), optionally preceded
 *                     by nn.ReflectionPad2d(pad) (pad_mode 1, conv pad 0) --
 *                     ConvBlock AE_model_unet.py:65-69, ResidualBlock :49-54.
 *   transposed == 1 : nn.ConvTranspose2d(Cin, Cout, k, stride, pad) --
 *                     ConvTBlock AE_model_unet.py:84-87, upconv4 :521.
 * H, W are the spatial dims of the layer INPUT.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t B, H, W;
    int32_t Cin, Cout;
    int32_t k, stride, pad;
    int32_t pad_mode;   /* 0 zero padding, 1 reflection padding (Conv2d only) */
    int32_t transposed; /* 0 Conv2d, 1 ConvTranspose2d */
    int32_t hints;      /* GDN_HINT_* bits; 0 = none.  A hint never changes results beyond rounding, only which internal
                         * plan a transform-domain path picks; every call of one layer instance (workspace / state queries,
                         * forward, backward) must carry the same hints. */
} gdn_conv_geom;

/* GDN_HINT_TRAIN: the layer runs forward AND backward with a weight gradient (a trained layer in train mode).  The
 * frequency-domain path then tiles for the sum of both passes: 40-point tiles (32 valid outputs of a 9x9 window: 128 x 416
 * is exactly 4 x 13 of them, 26 % fewer transformed points) shorten the three per-bin GEMM chains of a training step by a
 * quarter but make the forward's single-pass transforms slower, so inference / frozen layers keep the 32-point tiles. */
enum { GDN_HINT_TRAIN = 1, GDN_HINT_NO_X3 = 2, GDN_HINT_NO_WINO_F4 = 4, GDN_HINT_FFT_NP32 = 8, GDN_HINT_FFT_NP40 = 16 };
/* Plan overrides (tests, measurements; 0 = none), fields of `hints`: the library itself reads no environment variable.
 *   GDN_HINT_PLAN_BATCH(n), bits 8..15: every plan that depends on the batch size is made as if the batch were n (1..255), so
 *                           a batch-1 call takes the plans of a batch-n call and an image compares bitwise across batch sizes;
 *   GDN_HINT_PLAN_CUS(n),   bits 16..23: the persistent kernels (conv_ring_bf16, wgrad_ring_bf16) are planned as for a chip
 *                           with n CUs (a multiple of 8, <= 2040): small test shapes reach the multi-round / split paths;
 *   GDN_HINT_FFT_NP32 / _NP40: gdn_fftconv_* plans 32- / 40-point tiles where it would choose the other. */
#define GDN_HINT_PLAN_BATCH(n) (((n) & 0xff) << 8)
#define GDN_HINT_PLAN_CUS(n) ((((n) / 8) & 0xff) << 16)
/* GDN_HINT_NO_X3: the Winograd paths (gdn_winoconv_*, gdn_wino2conv_*) run their per-bin GEMMs on the fp32 matrix
 * instruction instead of as bf16 x 3 split products (gdn_gemm_x3_*).  Same results to rounding.  The library reads no
 * environment variable for this: the caller decides once (gdn_amd/ops.py: set_x3; ranks that share one GPU switch it off,
 * DESIGN.md 2.10) and the form of a forward's saved weight set follows the geometry its backward is called with.
 * GDN_HINT_NO_WINO_F4: gdn_winoconv_* plans F(2x2,3x3) where it would plan F(4x4,3x3) (zero-padded 3x3 layers whose GEMMs are
 * bf16 x 3 eligible); results differ by rounding only, the layout of the saved state follows the plan. */

enum { GDN_ACT_NONE = 0, GDN_ACT_TANH = 1, GDN_ACT_RELU = 2 };   /* bit flags */

/* tile_cfg flag bits shared by the conv entry points (low byte: tile id, 0 = automatic).
 * GDN_CFG_BF16: x/x2/w/y/addsrc (dy/wt/dx for dgrad) hold bfloat16 instead of float
 * (BASELINE configs[2]); accumulation, BatchNorm statistics and split-K partials stay fp32.
 * Needs Cin, C1 (dgrad: Cout) multiples of 64 and pixel pitches multiples of 8.
 * Exception: the 1-channel 9x9 heads (Cout == 1) take a bf16 x but fp32 weights and write an
 * fp32 depth map; their backward runs through the fp32 entry points.
 * The workspace/slot queries must be given the same tile_cfg as the launch.
 * bf16 tile ids (tests, A/B measurements): 1-3 tap-major tiles, 8/9 the round-1 row-patch kernel, 10/11 conv_ring_bf16 (256 x 64 /
 * 256 x 128: the LDS-DMA ring kernel of the stride-1 layers with a 3/5/7/9 window, the automatic choice for them; its persistent
 * workgroups may cut the last round of tiles into stage ranges, whose fp32 slabs live in `workspace`: the forward and the
 * data-gradient workspace queries include them; bit 0x800, "single stage", keeps every unit whole, as it keeps the other
 * kernels from splitting over the filter taps), 12 conv_ring2_bf16 (the same kernel on 512 x 64 tiles and 32-channel slabs:
 * the automatic choice for 64-output-channel layers with a 7 / 9 window that fill the chip twice).  For ids 10..12 bits 12..15
 * are measurement knobs of that kernel (timing-only variants that drop an operand's LDS-DMA traffic or skip the tap loop; 0 in
 * production). */
enum { GDN_CFG_KC64 = 0x200, GDN_CFG_NO_SPLITK = 0x800, GDN_CFG_BF16 = 0x10000 };

/* Output spatial dims of the layer. */
int gdn_conv_out_dims(const gdn_conv_geom* g, int32_t* Ho, int32_t* Wo);

/* Number of per-block partial-statistics slots gdn_conv_fwd writes when
 * `stats` is non-NULL, for the same tile_cfg; the stats buffer must hold
 * slots*2*Cout floats laid out [slot][2][Cout]. */
int64_t gdn_conv_stats_slots(const gdn_conv_geom* g, int32_t tile_cfg);

/* Forward convolution.  Replaces nn.Conv2d / nn.ConvTranspose2d forward
 * (AE_model_unet.py:50,53,67,85,300,521) and, through x2, the
 * torch.cat(...)+1x1 ConvBlock pair (:338-339,:345-346,:351-352,:357-358)
 * without materialising the concat: reduction channels [0,C1) come from x,
 * [C1,Cin) from x2 (x2 may be NULL when C1 == Cin).
 *   y[pixel][c] = tanh?( relu?( conv(x)[c] * ep_scale[c] + ep_shift[c] ) + addsrc[pixel][c] )
 *                 ep_scale/ep_shift (both or neither; NULL = identity) fold an eval-mode BatchNorm
 *                 (gdn_bn_eval_coeffs) into the epilogue: conv + BN + ReLU (+ residual) in one pass --
 *                 the frozen guide network and the inference path (depth_extract.py); act is a bit mask
 *                 of GDN_ACT_RELU (before the add) and GDN_ACT_TANH (after it)
 *   stats[slot][0][c], [1][c] = per-block sum / sum of squares of the raw conv
 *                               output (feeds train-mode BatchNorm, K7).
 * Launches that cannot fill the chip are split over the filter taps (split-K);
 * their partial sums live in `workspace` (gdn_conv_fwd_workspace_bytes, 0 when
 * no split is planned) and are combined -- with the same fusions -- by a second
 * kernel inside the call.
 * tile_cfg: 0 = automatic; low byte >0 forces a tile configuration (tuning/testing);
 * flag bits above.  x, x2, w, y, addsrc: float (default) or bfloat16 (GDN_CFG_BF16). */
size_t gdn_conv_fwd_workspace_bytes(const gdn_conv_geom* g, int32_t tile_cfg);
int gdn_conv_fwd(const gdn_conv_geom* g,
                 const void* x, int32_t ldx, const void* x2, int32_t ldx2, int32_t C1,
                 const void* w, void* y, int32_t ldy,
                 const void* addsrc, int32_t ld_add,
                 float* stats, const float* ep_scale, const float* ep_shift, int32_t act, int32_t tile_cfg,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Data gradient.  Replaces autograd's conv backward-data for the call sites
 * above (loss.backward(), trainer.py:467,767).  wt is the tap-major TRANSPOSED
 * weight [kh*kw][Cin][Cout] (see gdn_transpose_taps).  dx = dgrad(dy) (+ addsrc).
 * Reflection-padded layers need a workspace for the gradient on the padded
 * domain, folded back onto dx inside the call. */
size_t gdn_conv_dgrad_workspace_bytes(const gdn_conv_geom* g, int32_t tile_cfg);
/* bnb_y != NULL: dx (+ addsrc) is the FINAL gradient of z = [relu](BN_train(bnb_y)) -- the layer's input is the output of a
 * train-mode BatchNorm (+ReLU) and this call comes from its first consumer -- and bnb_partial [slots][2][Cin] receives that
 * BatchNorm's backward partial sums (sum dz, sum dz * xhat per slot; bnb_co = [scale, shift, mean, invstd][Cin]) computed from
 * the values as they are stored, so gdn_bn_bwd can skip its reduce pass.  slots = gdn_conv_dgrad_bnb_slots(g, tile_cfg);
 * 0 = not available for this layer (today: the bf16 stride-1 layers of the LDS-DMA ring kernel, zero padding).
 * dx_up2x (0 none, 1 align_corners False, 2 True): the layer's input was F.interpolate(t, scale_factor=2, mode='bilinear') and
 * nothing else reads the upsampled tensor (AE_model_unet.py:336-359: every upconvN); dx and addsrc are then dL/dt,
 * [B, H/2, W/2, Cin] -- the fold pass of a reflection-padded layer applies the adjoint of the interpolation while it folds, so
 * the full-resolution gradient is neither written nor read back.  Reflection-padded layers with Cin % 4 == 0 and even H, W
 * only (GDN_ERR_UNSUPPORTED otherwise); fp32 and bf16. */
int64_t gdn_conv_dgrad_bnb_slots(const gdn_conv_geom* g, int32_t tile_cfg);
int gdn_conv_dgrad(const gdn_conv_geom* g, const void* dy, int32_t ldy,
                   const void* wt, void* dx, int32_t ldx,
                   const void* addsrc, int32_t ld_add,
                   const void* bnb_y, int32_t ld_bnb, const float* bnb_co, int32_t bnb_relu, float* bnb_partial,
                   int32_t dx_up2x, void* workspace, size_t workspace_bytes, int32_t tile_cfg, void* stream);

/* Weight gradient (same call sites).  x is the layer input ([B,H,W], channel
 * slice [0,Cx) of width ldx), dy the output gradient.  Writes
 * dw[tap][co][ci_off + ci] for ci in [0,Cx), with row stride ld_dw (= total
 * Cin), so the two halves of a concat 1x1 conv are two calls.  Deterministic:
 * split-K partial slabs in `workspace` are summed in a fixed order.
 * dtypes (bit0 x, bit1 dy): 0 = fp32.  On the mixed-precision path the layers with a 1-3 channel
 * tensor (image-input conv, 64->1 heads) keep that tensor in fp32 and may pass the wide one as bf16. */
size_t gdn_conv_wgrad_workspace_bytes(const gdn_conv_geom* g, int32_t Cx);
int gdn_conv_wgrad(const gdn_conv_geom* g, const void* x, int32_t ldx, int32_t Cx,
                   const void* dy, int32_t ldy,
                   float* dw, int32_t ld_dw, int32_t ci_off,
                   void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream);

/* FFT-domain convolution for the large-window stride-1 layers (AE_model_unet.py ResidualBlock: 9x9 on
 * 64 channels, 7x7 on 128, 5x5 on 256; ConvBlock with ReflectionPad2d in R's decoder; the legacy
 * network's stride-1 ConvTranspose2d): stride 1, pad k/2 (zeros or reflection), odd k in 3..9, Cin and
 * Cout multiples of 64 up to 256, fp32.  Overlap-save on 32x32 tiles: real FFT of the input patches,
 * one complex (real-embedded, MFMA) GEMM per frequency bin, inverse FFT of the valid outputs.
 * Same contract as gdn_conv_fwd for y / addsrc / stats / ep_scale / ep_shift / act (slots:
 * gdn_fftconv_stats_slots).
 * in_scale / in_shift (nullable pair, Cin floats) + in_relu: x is the RAW output of the producer
 * convolution and the layer input is [relu](x*in_scale[c] + in_shift[c]) -- the producer's train-mode
 * BatchNorm (+ReLU) applied while the patch is loaded, so `a = relu(bn1(conv1 x))` of a ResidualBlock
 * (AE_model_unet.py:49-54) is never written to memory.  Padding stays zero.
 * in_up2x (0 = off, 1 = align_corners False, 2 = True; not together with in_scale): x is the LOW-resolution
 * tensor [B][H/2][W/2][Cin] and the layer input is its x2 bilinear upsampling (F.interpolate(scale_factor=2,
 * mode='bilinear') + ConvBlock, AE_model_unet.py:336-359; the legacy decoder :214-230), interpolated from the four
 * neighbours of every element while the patch is loaded -- border rule (zeros / reflection) first, on the upsampled
 * coordinates -- so the upsampled tensor is never written to memory.  g->H, g->W stay the convolution's (upsampled, even)
 * input extent.
 * xf_out (nullable, gdn_fftconv_spectrum_bytes) receives the input and weight spectra, which
 * gdn_fftconv_bwd reuses (weight gradient; data gradient without a second weight transform).
 * GDN_ERR_UNSUPPORTED for other geometries; transposed (stride-1) layers are forward-only
 * (gdn_fftconv_bwd_workspace_bytes == 0). */
size_t gdn_fftconv_fwd_workspace_bytes(const gdn_conv_geom* g);
size_t gdn_fftconv_spectrum_bytes(const gdn_conv_geom* g);
int64_t gdn_fftconv_stats_slots(const gdn_conv_geom* g);
int gdn_fftconv_fwd(const gdn_conv_geom* g, const float* x, int32_t ldx, const float* w,
                    float* y, int32_t ldy, const float* addsrc, int32_t ld_add, float* stats,
                    const float* ep_scale, const float* ep_shift, int32_t act,
                    const float* in_scale, const float* in_shift, int32_t in_relu, int32_t in_up2x,
                    void* xf_out, void* workspace, size_t workspace_bytes, void* stream);
/* Backward of the same layer from one transform of dy: dx = dgrad (+ addsrc) when dx != NULL,
 * dw[tap][Cout][Cin] = wgrad when dw != NULL (needs xf, the state saved by the forward; with
 * xf == NULL the data gradient transforms w itself).
 * phases: 0 / GDN_FFT_BWD_ALL = everything on `stream`.  The weight-gradient chain and the
 * data-gradient chain only share the spectrum of dy and use disjoint parts of the workspace, so a
 * caller may issue GDN_FFT_BWD_TRANSFORM on stream A, let stream B wait for it, issue GDN_FFT_BWD_DW
 * on B and GDN_FFT_BWD_DX on A (same arguments, same workspace) and join B into A: the two
 * latency-bound chains then overlap.  The library creates no stream or event itself.
 * dyb_* (nullable; replaces pass 3 of gdn_bn_bwd for THIS layer's own BatchNorm): `dy` is dout, the
 * gradient of z = [relu](BN_train(dyb_y)) with dyb_y this layer's raw conv output, dyb_co = {scale, shift,
 * mean, invstd}[Cout] and dyb_kk = {k1, k2}[Cout] from gdn_bn_bwd_coeffs; the dy transform computes
 * dy = scale*(dz - k1 - xhat*k2) while loading, so dy itself is never written to memory.
 * bnb_* (nullable; zero-padded layers, gdn_fftconv_bnb_slots(g) > 0 slots): dx is the final gradient of this layer's INPUT
 * z = [relu](BN_train(bnb_y)) (bnb_y: the producer's raw conv output [B][H][W][Cin], pitch ld_bnb; bnb_co = {scale, shift,
 * mean, invstd}[Cin]); the pass that writes dx also writes that BatchNorm's backward partial sums (sum dz, sum dz * xhat) to
 * bnb_partial[slots][2][Cin] -- gdn_bn_bwd / gdn_bn_bwd_coeffs take them as ext_partial, and the stand-alone reduce pass
 * over (dx, y) disappears (AE_model_unet.py:51,54 as consumed by the next conv of :50,53).
 * dx_up2x (as in_up2x of the forward; reflection-padded layers only, else GDN_ERR_UNSUPPORTED): dx, addsrc are
 * [B][H/2][W/2][Cin], the gradient of the LOW-resolution tensor the forward upsampled on load -- the pass that folds the
 * padded-domain gradient back onto the image also applies the adjoint of the interpolation (gather form, deterministic). */
enum { GDN_FFT_BWD_TRANSFORM = 1, GDN_FFT_BWD_DW = 2, GDN_FFT_BWD_DX = 4, GDN_FFT_BWD_ALL = 7 };
size_t gdn_fftconv_bwd_workspace_bytes(const gdn_conv_geom* g);
int gdn_fftconv_bwd(const gdn_conv_geom* g, const float* dy, int32_t ldy, const float* w,
                    const void* xf, float* dx, int32_t ldx, const float* addsrc, int32_t ld_add,
                    float* dw, const float* dyb_y, int32_t ld_dyb, const float* dyb_co,
                    const float* dyb_kk, int32_t dyb_relu,
                    const float* bnb_y, int32_t ld_bnb, const float* bnb_co, int32_t bnb_relu, float* bnb_partial,
                    int32_t dx_up2x, int32_t phases,
                    void* workspace, size_t workspace_bytes, void* stream);
int64_t gdn_fftconv_bnb_slots(const gdn_conv_geom* g);
/* Measurement hooks (bench.py roofline_cgemm): launch ONLY the per-bin complex GEMMs of the frequency-domain layer `g` on
 * whatever the workspace holds -- which: 0 forward (Y = X W), 1 data gradient, 2 the weight gradient's reduction over the
 * tiles; 1 and 2 use disjoint outputs and may run on two streams like gdn_fftconv_bwd's chains.  gdn_fftconv_cgemm_shape:
 * host query of {frequency bins, tiles M, transform points}: per bin the GEMM is [M x Cin] x [Cin x Cout] complex, executed
 * as three real products (Gauss) on v_mfma_f32_32x32x2_f32.  Replace nothing of the reference: the product path reaches these
 * kernels through gdn_fftconv_fwd / _bwd (AE_model_unet.py:50,53: the 5x5 ... 9x9 ResidualBlock convolutions). */
size_t gdn_fftconv_cgemm_workspace_bytes(const gdn_conv_geom* g);
int gdn_fftconv_cgemm(const gdn_conv_geom* g, int32_t which, void* workspace, size_t workspace_bytes, void* stream);
int gdn_fftconv_cgemm_shape(const gdn_conv_geom* g, int32_t* bins, int32_t* M, int32_t* np);

/* Winograd F(2x2,3x3) convolution for the 3x3 stride-1 layers (zero or reflection padding 1) on
 * 64..512 channels (the 512-channel ResidualBlocks of levels 3 and 4, AE_model_unet.py:45-57; R's
 * decoder ConvBlocks upconv0 / upconv1, :60-77), fp32: 2.25x fewer
 * multiplies than the direct kernel, transforms that only add and halve.  Same contract as
 * gdn_conv_fwd for y / addsrc / stats / ep_scale / ep_shift / act (slots: gdn_winoconv_stats_slots).
 * in_scale / in_shift / in_relu: as for gdn_fftconv_fwd (the producer's train-mode BatchNorm + ReLU
 * applied while the 4x4 patches are loaded).  in_up2x: as for gdn_fftconv_fwd (x2 bilinear upsampling of a
 * low-resolution x while the 4x4 patches are loaded: R's upconv0 / upconv1).
 * state_out (nullable, gdn_winoconv_state_bytes) receives the transformed input, which
 * gdn_winoconv_bwd needs for the weight gradient, followed by the data gradient's transformed weights (written by the same
 * weight-transform launch: the backward of that step then runs no weight transform).  Cin/64 and Cout/64 must be powers of two. */
size_t gdn_winoconv_fwd_workspace_bytes(const gdn_conv_geom* g);
size_t gdn_winoconv_state_bytes(const gdn_conv_geom* g);
int64_t gdn_winoconv_stats_slots(const gdn_conv_geom* g);
int gdn_winoconv_fwd(const gdn_conv_geom* g, const float* x, int32_t ldx, const float* w,
                     float* y, int32_t ldy, const float* addsrc, int32_t ld_add, float* stats,
                     const float* ep_scale, const float* ep_shift, int32_t act,
                     const float* in_scale, const float* in_shift, int32_t in_relu, int32_t in_up2x,
                     void* state_out, void* workspace, size_t workspace_bytes, void* stream);
/* dx = dgrad (+ addsrc) when dx != NULL (needs state, or w when state is NULL), dw[tap][Cout][Cin] = wgrad when
 * dw != NULL (needs state).
 * bnb_* (nullable; replaces the reduce pass of gdn_bn_bwd, AE_model_unet.py:51,54,68): dx is the gradient of
 * z = [relu](BN_train(bnb_y)), bnb_y [B,H,W,Cin] with pitch ld_bnb being that BatchNorm's input and
 * bnb_co = {scale, shift, mean, invstd}[Cin]; the output transform that writes dx (+ addsrc) also writes
 * bnb_partial[slot][2][Cin] = per-slot sum(dz), sum(dz*xhat), dz = dx*[z>0 if bnb_relu], slots =
 * gdn_winoconv_bnb_slots(g) (0: not available -- reflection-padded layers).  (The frequency-domain layers do
 * not offer this: their inverse transforms are VALU-bound and the fused sums measured slower than the reduce pass.)
 * dx_up2x: as for gdn_fftconv_bwd (reflection-padded layers: dx is the gradient of the low-resolution tensor). */
size_t gdn_winoconv_bwd_workspace_bytes(const gdn_conv_geom* g);
int64_t gdn_winoconv_bnb_slots(const gdn_conv_geom* g);
/* Measurement hook: only the 16 per-bin MFMA GEMMs of one forward, V [16][tiles][Cin] x U [16][Cout][Cin]
 * -> Mo [16][tiles][Cout] (tiles = B * ceil(H/2) * ceil(W/2)). */
int gdn_winoconv_gemm(const gdn_conv_geom* g, const float* V, const float* U, float* Mo, void* stream);
int gdn_winoconv_bwd(const gdn_conv_geom* g, const float* dy, int32_t ldy, const float* w,
                     const void* state, float* dx, int32_t ldx, const float* addsrc, int32_t ld_add,
                     float* dw, const float* bnb_y, int32_t ld_bnb, const float* bnb_co, int32_t bnb_relu,
                     float* bnb_partial, int32_t dx_up2x, void* workspace, size_t workspace_bytes, void* stream);

/* Winograd F(3x3,2x2) for the 4x4 stride-2 pad-1 layers of G (ConvBlock(k4,s2,p1) AE_model_unet.py:497-500 with
 * ReflectionPad2d(1) or zero padding; ConvTBlock = ConvTranspose2d(k4,s2,p1) :517-520), fp32: the layer is a 2x2
 * stride-1 convolution over the four polyphase images of its input, computed with 16 instead of 36 multiplies per 3x3
 * output tile (2.25x fewer than the direct kernel; same interpolation points as F(2x2,3x3)).  Cin, Cout in
 * {64,128,256,512}; Conv2d needs even H, W.  y / addsrc / stats / ep_scale / ep_shift / act as for gdn_conv_fwd (no tanh;
 * slots: gdn_wino2conv_stats_slots).  state_out (Conv2d only, gdn_wino2conv_state_bytes) receives the transformed input
 * for the weight gradient.
 * Backward: dx = dgrad (+ addsrc) when dx != NULL (needs w); dw[tap][Cout][Cin] when dw != NULL -- a Conv2d needs `state`,
 * a ConvTranspose2d needs x (its forward input, pitch ldx_in): its weight gradient shares the transform of dy with the
 * data gradient. */
size_t gdn_wino2conv_fwd_workspace_bytes(const gdn_conv_geom* g);
size_t gdn_wino2conv_state_bytes(const gdn_conv_geom* g);
int64_t gdn_wino2conv_stats_slots(const gdn_conv_geom* g);
int gdn_wino2conv_fwd(const gdn_conv_geom* g, const float* x, int32_t ldx, const float* w,
                      float* y, int32_t ldy, const float* addsrc, int32_t ld_add, float* stats,
                      const float* ep_scale, const float* ep_shift, int32_t act,
                      void* state_out, void* workspace, size_t workspace_bytes, void* stream);
size_t gdn_wino2conv_bwd_workspace_bytes(const gdn_conv_geom* g);
int gdn_wino2conv_bwd(const gdn_conv_geom* g, const float* dy, int32_t ldy, const float* w,
                      const float* x, int32_t ldx_in, const void* state,
                      float* dx, int32_t ldx, const float* addsrc, int32_t ld_add, float* dw,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The 1 <-> 64 channel 9x9 stride-1 pad-4 layers on the matrix cores (G's first convolution AE_model_unet.py:496, the
 * data and weight gradients of the 64 -> 1 heads :362 / :521), fp32.  x1 is a dense single-channel image [B,H,W].
 *   gdn_conv_c1_fwd:   y[p][n] = sum_tap x1[p + tap - 4] * w[tap][n]   (N = 64; taps flipped when flip != 0; zero or reflection
 *                      padding of x1); y / addsrc / stats / ep_scale / ep_shift / act as for gdn_conv_fwd (no tanh), slots =
 *                      gdn_conv_c1_stats_slots.  The first convolution forward, and -- with x1 = d(pre-tanh) -- a head's data gradient.
 *   gdn_conv_c1_wgrad: dw[tap][n] = sum_p gw[p][n] * x1[p + tap - 4]   (gw [B,H,W,64], pitch ldg; result written at the flipped
 *                      tap when flip != 0).  The first convolution's weight gradient (x1 = input, gw = dy) and a head's
 *                      (x1 = d(pre-tanh), gw = the head's input; flip for a Conv2d head, none for a ConvTranspose2d one).
 *   gdn_conv_c1_fwd also takes Cin = 3 (x1 [B,H,W,3], w [tap][64][3]): R's first convolution Conv2d(3, 64, 9) after
 *   ReflectionPad2d(4), AE_model_unet.py:273 (its weight gradient stays on gdn_conv_wgrad).
 *   x1, w, stats, dw stay fp32.  dtypes of gdn_conv_c1_fwd: bit 0 = y is bf16, bit 1 = addsrc is bf16 (a bf16 model's head
 *   data gradient, rounded once after the fp32 add); gw_bf16 != 0: gw is bf16 (the head's bf16 input), ldg % 4 == 0 either way. */
int64_t gdn_conv_c1_stats_slots(int32_t B, int32_t H, int32_t W);
int gdn_conv_c1_fwd(const float* x1, int32_t Cin, int32_t B, int32_t H, int32_t W, int32_t N, int32_t k, int32_t pad, int32_t reflect,
                    int32_t flip, const float* w, void* y, int32_t ldy, const void* addsrc, int32_t ld_add,
                    float* stats, const float* ep_scale, const float* ep_shift, int32_t act, int32_t dtypes, void* stream);
size_t gdn_conv_c1_wgrad_workspace_bytes(void);
int gdn_conv_c1_wgrad(const float* x1, const void* gw, int32_t ldg, int32_t gw_bf16, int32_t B, int32_t H, int32_t W, int32_t N,
                      int32_t k, int32_t pad, int32_t reflect, int32_t flip, float* dw,
                      void* workspace, size_t workspace_bytes, void* stream);

/* fp32 per-bin GEMMs on the bf16 matrix pipe ("bf16 x 3": an fp32 operand is exactly the sum of three bf16 terms; six bf16
 * products, accumulated in fp32, reproduce the fp32 product to its own rounding level at 6/16 of the matrix-pipe cycles) --
 * the core behind the Winograd layers' per-bin GEMMs (torch / cuDNN conv2d of the 512-channel ResidualBlocks,
 * AE_model_unet.py:45-57).  Measurement / test hooks: the product path reaches the same kernels through gdn_winoconv_*.
 *   gdn_gemm_x3_pack : B fp32 row-major [bins][rows][K] -> packed bf16 panels (gdn_gemm_x3_packed_bytes), split once
 *   gdn_gemm_x3_nt   : C[bin][m][n] = sum_k A[bin][m][k] * B[bin][n][k];  A fp32 row-major [bins][M][K] (split while it is
 *                      staged), Bp the packed panels of B [bins][N][K], C fp32 row-major [bins][M][N].
 *   gdn_gemm_x3_tn   : P[split][bin][i][j] = sum_t A[bin][t][i] * B[bin][t][j] over the split's chunk of the T rows (the weight
 *                      gradients' reduction over tiles); A [bins][T][NI], B [bins][T][NJ] fp32 row-major, both split while staged;
 *                      P fp32 [nsplit][bins][NI][NJ], partial sets to be summed by the caller.  NI, NJ multiples of 128.
 * N a multiple of 128, K a multiple of 32 (GDN_ERR_UNSUPPORTED otherwise). */
size_t gdn_gemm_x3_packed_bytes(int32_t bins, int32_t rows, int32_t K);
int gdn_gemm_x3_pack(const float* src, void* dst, int32_t bins, int32_t rows, int32_t K, void* stream);
int gdn_gemm_x3_nt(const float* A, const void* Bp, float* C, int32_t bins, int32_t M, int32_t N, int32_t K, void* stream);
int gdn_gemm_x3_tn(const float* A, const float* B, float* P, int32_t bins, int32_t T, int32_t NI, int32_t NJ, int32_t nsplit,
                   void* stream);
/* host query: the number of reduction splits gdn_winoconv_bwd / gdn_wino2conv_bwd give gdn_gemm_x3_tn for a shape (0: shape not
 * eligible): rounds of the chip per split plus the partial-product sets read back, DESIGN.md 2.10 */
int64_t gdn_gemm_x3_tn_splits(int32_t bins, int32_t T, int32_t NI, int32_t NJ);

/* bf16 weight gradient (BASELINE configs[2]): x and dy hold bfloat16, dw is fp32 (the master
 * gradient arena).  Same contract as gdn_conv_wgrad otherwise.  Needs Cx and Cout multiples of
 * 64, pixel pitches multiples of 8 and 16-byte aligned bases.  v_mfma_f32_32x32x16_bf16 fed by
 * ds_read_b64_tr_b16.  cfg (low 3 bits): 0 automatic; 1 / 3 / 2 force the small / medium / large
 * staging class of the round-1 kernel (tuning, tests); 4 forces wgrad_ring_bf16 (csrc/wgrad_ring.h,
 * round 5: persistent workgroups, LDS-DMA staged row images, one filter-row pair per workgroup,
 * the taps of a row as register shifts of one transposed window -- stride-1 Conv2d with a 3 / 5 / 7 / 9
 * window; GDN_ERR_UNSUPPORTED otherwise), the automatic choice for 5x5 ... 9x9 windows.  Bits 12..15:
 * measurement knobs of that kernel (timing-only variants that drop its DMA traffic or its MFMA
 * loop; 0 in production).  The workspace query takes the same cfg.  Results are bitwise reproducible
 * for a given geometry and cfg (fixed-order split-K). */
size_t gdn_conv_wgrad_bf16_workspace_bytes(const gdn_conv_geom* g, int32_t Cx, int32_t cfg);
int gdn_conv_wgrad_bf16(const gdn_conv_geom* g, const void* x, int32_t ldx, int32_t Cx,
                        const void* dy, int32_t ldy,
                        float* dw, int32_t ld_dw, int32_t ci_off,
                        void* workspace, size_t workspace_bytes, int32_t cfg, void* stream);

/* `dtypes` of the element-wise entry points below: bit i set = the i-th ACTIVATION-type tensor argument
 * (in argument order, NULL-able ones included) holds bfloat16 instead of float; 0 = all fp32.
 * Per-channel coefficient vectors, statistics, parameter gradients and scalars are always fp32. */

/* [ntaps][R][C] -> [ntaps][C][R].  dtypes: bit0 w, bit1 wt (fp32 -> bf16 transposed copy in one pass). */
int gdn_transpose_taps(const void* w, void* wt, int32_t ntaps, int32_t R, int32_t C, int32_t dtypes, void* stream);
/* dst[i] = src[i] with dtype conversion.  dtypes: bit0 src, bit1 dst. */
int gdn_cast(const void* src, void* dst, int64_t n, int32_t dtypes, void* stream);
/* torch layout [A][Bc][ntaps] (Conv2d weight [Cout][Cin][kh][kw], or ConvTranspose2d
 * weight [Cin][Cout][kh][kw]) <-> tap-major [ntaps][Cout][Cin].  a_is_cout selects
 * which of the two leading torch dims is Cout. */
int gdn_weight_to_tapmajor(const float* w_torch, float* w_tap, int32_t Cout, int32_t Cin, int32_t ntaps,
                           int32_t a_is_cout, void* stream);
int gdn_weight_from_tapmajor(const float* w_tap, float* w_torch, int32_t Cout, int32_t Cin, int32_t ntaps,
                             int32_t a_is_cout, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm2d (AE_model_unet.py:51,54,68,86), ReLU (:52,69,87), residual add (:57).
 * ---------------------------------------------------------------------- */
/* Train mode: reduce the conv kernel's partial statistics (fp64), update the
 * running stats (momentum, unbiased variance), and emit the per-channel
 * affine  scale = gamma*invstd,  shift = beta - mean*scale  plus mean/invstd
 * for backward.  num_batches_tracked (nullable, device int64): += 1, nn.BatchNorm2d's counter. */
int gdn_bn_finalize_train(const float* stats, int64_t slots, int32_t C, int64_t count,
                          const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps,
                          float* scale, float* shift, float* mean, float* invstd,
                          int64_t* num_batches_tracked, void* stream);
/* Eval mode: scale/shift from the running statistics. */
int gdn_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int32_t C,
                       float* scale, float* shift, void* stream);
/* out = [relu](y*scale[c] + shift[c]) (+ residual).  npix pixels of C channels.
 * dtypes: bit0 y, bit1 residual, bit2 out. */
int gdn_bn_apply(const void* y, int32_t ldy, const float* scale, const float* shift,
                 const void* residual, int32_t ld_res, void* out, int32_t ld_out,
                 int64_t npix, int32_t C, int32_t relu, int32_t dtypes, void* stream);
/* low = [relu](y*scale[c] + shift[c]) (+ residual), up = F.interpolate(low, scale_factor=2, mode='bilinear', align_corners) in ONE
 * pass over dense NHWC tensors [B,H,W,C] -> [B,2H,2W,C]: a ResidualBlock's output followed by the decoder's upsampling
 * (AE_model_unet.py:55-57 -> :336-359, north_star "bilinear-interp + residual-add fused").  low may be NULL (the block output has
 * no other reader).  Bit-identical to gdn_bn_apply followed by gdn_upsample2x_fwd: the interpolated values are rounded to low's
 * storage type first.  dtypes: bit0 y, bit1 residual, bit2 low, bit3 up (1 = bfloat16).  C % 4 == 0, B * 2H <= 65535. */
int gdn_bn_apply_up2x(const void* y, const float* scale, const float* shift, const void* residual, void* low, void* up,
                      int32_t B, int32_t H, int32_t W, int32_t C, int32_t relu, int32_t align_corners, int32_t dtypes,
                      void* stream);
/* Backward of out = [relu](BN_train(y)):
 *   pass 1 (reduce): per-channel sum(dz), sum(dz*xhat) with dz = dout*[z>0];
 *   pass 2 (apply) : dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)),
 *                    dgamma = sum(dz*xhat), dbeta = sum(dz).
 * Both passes inside one call; workspace from gdn_bn_bwd_workspace_bytes.
 * ext_partial (nullable, [ext_slots][2][C]): pass 1 was already done by the epilogue of the kernel that wrote
 * dout (bnb_partial of gdn_winoconv_bwd / gdn_fftconv_bwd) and is skipped.
 * dtypes: bit0 dout, bit1 y, bit2 dy. */
size_t gdn_bn_bwd_workspace_bytes(int64_t npix, int32_t C);
int gdn_bn_bwd(const void* dout, int32_t ld_dout, const void* y, int32_t ldy,
               const float* gamma, const float* scale, const float* shift,
               const float* mean, const float* invstd,
               void* dy, int32_t ld_dy, float* dgamma, float* dbeta,
               int64_t npix, int32_t C, int32_t relu,
               const float* ext_partial, int64_t ext_slots,
               void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream);

/* Passes 1 + 2 of gdn_bn_bwd only: dgamma, dbeta (nullable) and kk = {mean(dz), mean(dz*xhat)}[C] for a consumer that
 * applies pass 3 while loading (gdn_fftconv_bwd dyb_*).  With ext_partial, dout / y / workspace may be NULL. */
int gdn_bn_bwd_coeffs(const void* dout, int32_t ld_dout, const void* y, int32_t ldy,
                      const float* scale, const float* shift, const float* mean, const float* invstd,
                      float* dgamma, float* dbeta, float* kk, int64_t npix, int32_t C, int32_t relu,
                      const float* ext_partial, int64_t ext_slots,
                      void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream);

/* Backward of out = [relu](y*scale + shift) through an EVAL-mode BatchNorm (fixed coefficients from
 * gdn_bn_eval_coeffs): dy = scale*dout*[z>0].  Used when a gradient crosses the frozen guide network
 * (--latent_grad, the guided training of trainer.py:699-703 without the no_grad).  relu: 0 none, 1 y is the raw conv
 * output (mask = y*scale+shift > 0), 2 y is the activated output of a fused conv+BN+ReLU epilogue (mask = y > 0).
 * dtypes: bit0 dout, bit1 y, bit2 dy. */
int gdn_bn_eval_bwd(const void* dout, int32_t ld_dout, const void* y, int32_t ldy,
                    const float* scale, const float* shift, void* dy, int32_t ld_dy,
                    int64_t npix, int32_t C, int32_t relu, int32_t dtypes, void* stream);

/* ------------------------------------------------------------------------
 * x2 bilinear up-sampling: F.interpolate(align_corners=False) AE_model_unet.py:336,343,349,355
 * and nn.Upsample(align_corners=True) :135,203,215,227.  NHWC, C channels.
 * ---------------------------------------------------------------------- */
/* dtypes: bit0 input, bit1 output. */
int gdn_upsample2x_fwd(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                       int32_t align_corners, int32_t dtypes, void* stream);
int gdn_upsample2x_bwd(const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                       int32_t align_corners, int32_t dtypes, void* stream);

/* Layout and small element-wise helpers.  dtypes: bit0 x, bit1 y (gdn_add: bit0 a, bit1 b, bit2 out). */
int gdn_nchw_to_nhwc(const void* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtypes, void* stream);
int gdn_nhwc_to_nchw(const void* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t dtypes, void* stream);
int gdn_add(const void* a, const void* b, void* out, int64_t n, int32_t dtypes, void* stream);
/* out[p][c] = a[p][c] (+ b[p][c], b nullable): npix pixels of C channels, each tensor with its own pixel pitch -- the
 * gradient of a torch.cat half (AE_model_unet.py:340,346,352,358) is a channel slice of the 1x1 conv's data gradient.
 * dtypes: bit0 a, bit1 b, bit2 out. */
int gdn_add_pitched(const void* a, int32_t lda, const void* b, int32_t ldb, void* out, int32_t ld_out,
                    int64_t npix, int32_t C, int32_t dtypes, void* stream);
/* out[i] = x[i] * (*s), s a device scalar: d(loss)/d(pred) times the upstream gradient of the loss node. */
int gdn_scale_dev(const float* x, const float* s, float* out, int64_t n, void* stream);
/* dpre = dout * (1 - out^2): backward of x15.tanh() (AE_model_unet.py:363,571). */
int gdn_tanh_bwd(const float* dout, const float* out, float* dpre, int64_t n, void* stream);
int gdn_fill(float* p, float value, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * Training losses, forward value and gradient in one call, sync-free (F6).
 * Each writes its scalar to *loss (device) and ADDS its gradient into dout
 * (caller zero-fills dout once per step); `w` scales both.
 * ---------------------------------------------------------------------- */
/* Masked BerHu, trainer.py:433-448 == :705-720.  out/gt [B,1,H,W];
 * sparse [B,Cs,H,W] NCHW (channel 0 used) or NULL; box = {y1,y2,x1,x2}.
 * loss = 3*mean(w*rho).  workspace >= gdn_loss_workspace_bytes().
 * ext_max: NULL = the threshold c = 0.2*max|out-gt| is taken over THIS batch (one reference run, and what
 * each rank of the data-parallel build does); a device float = use that maximum instead -- the value of
 * gdn_absdiff_max after a 4-byte all-reduce(MAX), which reproduces nn.DataParallel's threshold over the
 * gathered batch (--global_berhu). */
size_t gdn_loss_workspace_bytes(int64_t npix);
int gdn_absdiff_max(const float* a, const float* b, int64_t n, float* max_out, void* stream);
int gdn_berhu_masked(const float* out, const float* gt, const float* sparse, int32_t Cs,
                     int32_t B, int32_t H, int32_t W, const int32_t box[4], const float* ext_max,
                     float* loss, float* dout, void* workspace, size_t workspace_bytes, void* stream);
/* Sobel L1, utils.py:105-131 with the factor 3 of trainer.py:453 passed as `weight`.
 * total (nullable): *total = *loss + *plus (plus nullable) -- the step's loss sum, trainer.py:456. */
int gdn_sobel_l1(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W, float weight,
                 float* loss, float* dpred, const float* plus, float* total,
                 void* workspace, size_t workspace_bytes, void* stream);
/* Edge-aware smoothness, utils.py:139-178 + trainer.py:753-754.
 * depth [B,1,H,W], img [B,Ci,H,W] NCHW.  loss = mean|0.1*smooth|.
 * total (nullable): *total = (*loss + *plus) + *plus2 (both nullable) -- trainer.py:757. */
int gdn_smoothness(const float* depth, const float* img, int32_t Ci, int32_t B, int32_t H, int32_t W,
                   float* loss, float* ddepth, const float* plus, const float* plus2, float* total,
                   void* workspace, size_t workspace_bytes, void* stream);
/* loss_accum (+)= weight * mean((a-b)^2): the latent MSE terms, trainer.py:728-733
 * (value only, F3).  accumulate != 0 adds to the existing *loss.  dtypes: bit0 a, bit1 b. */
int gdn_mse(const void* a, const void* b, int64_t n, float weight, int32_t accumulate,
            float* loss, void* workspace, size_t workspace_bytes, int32_t dtypes, void* stream);
/* Gradient of weight*mean((a-b)^2) w.r.t. a, times the device scalar *gscale (NULL = 1):
 * da = gscale * 2*weight/n * (a-b).  dtypes: bit0 a, bit1 b, bit2 da.  (--latent_grad) */
int gdn_mse_grad(const void* a, const void* b, int64_t n, float weight, const float* gscale,
                 void* da, int32_t dtypes, void* stream);

/* ------------------------------------------------------------------------
 * Depth metrics, calculate_error.py:10-103: per image min-max -> x80, Godard
 * crop, validity mask, median scaling (lower median), clamp, 8 reductions.
 * gt_sparse/gt/pred are [B,1,H,W]; errors[8] = batch means, device memory.
 * ---------------------------------------------------------------------- */
size_t gdn_depth_metrics_workspace_bytes(int32_t B, int32_t H, int32_t W);
int gdn_depth_metrics(const float* gt_sparse, const float* gt, const float* pred,
                      int32_t B, int32_t H, int32_t W, int32_t crop,
                      float* errors, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * KITTI training-time augmentation on the device.  Replaces the host pipeline
 * datasets_list.py:82-101 -> transform_list.py RandomHorizontalFlip :158-166, RandomScaleCrop :185-199
 * (scipy.misc.imresize = bytescale + Pillow bilinear), ArrayToTensor :100-118, Normalize :84-92,
 * bit-exactly (uint8 resampling in Pillow's 22-bit fixed point, IEEE float32 normalisation).
 *   src    [B][H][W][C] as decoded from the image files: uint8, or float32 when src_is_f32 (then the
 *          per-image min/max stretch of scipy's bytescale is applied first); C <= 4
 *   params device int32 [B][5] = {flip, scaled_h, scaled_w, off_y, off_x}: the sample is flipped, resized to
 *          scaled_h x scaled_w (>= H x W) and cropped back to H x W at (off_y, off_x); ignored when train == 0
 *   train  0: validation transform (GDN_main.py:49-52): ArrayToTensor + Normalize only
 *   dst    [B][C][H][W] float32 = (v/255 - 0.5)/0.5
 * ---------------------------------------------------------------------- */
size_t gdn_kitti_augment_workspace_bytes(int32_t B);
int gdn_kitti_augment(const void* src, int32_t src_is_f32, int32_t B, int32_t H, int32_t W, int32_t C,
                      const int32_t* params, int32_t train, float* dst,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Fused Adam with coupled L2 weight decay over one flat arena
 * (torch.optim.Adam(..., weight_decay=5e-4), GDN_main.py:157,173; step at
 * trainer.py:468,768).  grad_scale multiplies the gradient first (1/world
 * after an RCCL sum all-reduce).
 * ---------------------------------------------------------------------- */
int gdn_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  int32_t step, float grad_scale, void* stream);
/* The same update with every step-dependent scalar in DEVICE memory, so the launch can be captured in a hipGraph and
 * replayed: hyper = float[6] {lr, beta1, beta2, eps, weight_decay, grad_scale} (the host rewrites it when the schedule
 * changes the learning rate); state = 32 bytes {double beta1^t, double beta2^t, int32 t, float bc1, float bc2s},
 * initialised to {1.0, 1.0, 0, 0, 0} and advanced by the call itself. */
int gdn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n,
                      const float* hyper, void* state, void* stream);

/* ------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): the shader clock the chip holds WHILE a window of launches runs, so a
 * roofline fraction can be read against the clock of the box it was measured on (bench.py `clock_ghz`, `frac_at_clock`).
 *   buf    HOST-PINNED memory mapped into the device (hipHostMalloc; not device memory: the setter and the watcher run on
 *          different XCDs, whose L2s are not coherent inside a kernel), uint64[4]:
 *          {flag, shader cycles (s_memtime), 100 MHz ticks (s_memrealtime), ended_by_flag}
 *   gdn_clock_probe_arm(buf, stream)            clears buf, stream-ordered BEFORE the measured launches;
 *   gdn_clock_probe_watch(buf, max_ticks, side) one wave on a SECOND stream of the caller's (which must wait for the arm):
 *                                               reads both counters, sleeps until the flag is set or max_ticks (<= 1e9 =
 *                                               10 s) have passed, reads them again: clock = cycles / ticks x 100 MHz;
 *   gdn_clock_probe_stop(buf, stream)           sets the flag, stream-ordered AFTER the measured launches.
 * One sleeping wave: no LDS, 16 registers -- it fits beside any kernel of this library.
 * ---------------------------------------------------------------------- */
int gdn_clock_probe_arm(uint64_t* buf, void* stream);
int gdn_clock_probe_watch(uint64_t* buf, uint64_t max_ticks, void* side_stream);
int gdn_clock_probe_stop(uint64_t* buf, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDN_HIP_H */
