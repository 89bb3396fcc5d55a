/* cine_hip.h -- C ABI of libcine_hip.so, the MI355X (gfx950) cine-MRI reconstruction kernels.
 *
 * The reference (f78bono/deep-cine-cardiac-mri) is pure Python on PyTorch and has no
 * FFI of its own: the boundary this library sits under is the ATen operator layer that
 * `reconstruction.models.*` / `reconstruction.utils.*` dispatch to.  Each entry point
 * below names the reference code it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative CINE_E* code; it never throws,
 *     aborts, allocates device memory or synchronises.  cine_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - all pointers are DEVICE pointers owned by the caller (except where noted), fp32,
 *     contiguous, in the reference's layouts: complex = trailing pair (re, im);
 *     k-space (b, t, coil, h, w, 2); image (b, t, h, w, 2); mask uint8 (b, t, 1, h, 1, 1).
 *   - `stream` is the caller's hipStream_t passed as void* (NULL = default stream);
 *     every launch goes onto it, so the calls are hipGraph-capturable.
 *   - scratch comes from the caller: `ws`/`ws_bytes` with a matching *_ws_bytes() query.
 *   - re-entrant: what a call computes depends on its arguments only (activation slopes and ReLU switches are arguments, never
 *     library state).  State that outlives a call is limited to: the optional launch profiler (cine_profile_begin/end: a
 *     mutex-guarded event list, off by default) and per-(kernel, device) once-flags for the > 64 KB LDS opt-in -- process-wide,
 *     neither changes a result; and three settings of the CALLING THREAD, invisible to every other thread: its error string
 *     (cine_last_error), its optional side stream (cine_set_side_stream) and its diagnostic kernel-selection mask
 *     (cine_set_conv_plane: which of two bit-identical kernels a launch uses).  The library reads no environment variable.
 *     tests/test_hip_parity.py::test_two_threads_two_streams_are_independent runs two models on two threads and streams while
 *     one of them flips the diagnostic mask and uses another slope.
 *   - the library creates no streams.  The two backward entry points that overlap weight gradients with the input-gradient
 *     chain (cine_unet2d_backward, cine_unet3d_backward, cine_mwcnn_backward) create and destroy hipEvents (no timing) to order the two streams.
 */
#ifndef CINE_HIP_H
#define CINE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CINE_OK 0
#define CINE_EINVAL (-1)      /* bad argument (null pointer, non-positive size, bad enum)      */
#define CINE_EUNSUPPORTED (-2)/* size/shape outside what the kernels implement                 */
#define CINE_EWORKSPACE (-3)  /* workspace too small                                           */
#define CINE_EHIP (-4)        /* a HIP launch failed; see cine_last_error()                    */

/* library / build identification */
int cine_version(void);                 /* ABI version, currently 3 */
const char* cine_last_error(void);      /* thread-local, never NULL */
const char* cine_build_arch(void);      /* "gfx950" */

/* ------------------------------------------------------------------------------------------
 * Centered ortho FFTs                          reference: reconstruction/utils/fftc.py
 * ------------------------------------------------------------------------------------------ */

/* fft2c (fftc.py:59-83) / ifft2c (fftc.py:86-110) over the last two spatial dims of
 * `nimg` images of h x w complex.  inverse = 0 forward, 1 inverse.  in == out allowed.
 * Line lengths (every transform of this header): 200 (the 10 x 20 engine), any 2^a 3^b 5^c <= 512 (mixed-radix Stockham engine),
 * any other length <= 400 (direct DFT in LDS); CINE_EUNSUPPORTED beyond. */
int cine_fft2c(const float* in, float* out, int nimg, int h, int w, int inverse, void* stream);

/* fft1c (fftc.py:5-29) / ifft1c (fftc.py:32-56): `nlines` contiguous lines of n complex.
 * variant 0 = fftc.py shift order (ifftshift, transform, fftshift);
 * variant 1 = XPDNet's order (xpdnet.py:466 / :500), which differs for odd n. */
int cine_fft1c(const float* in, float* out, long nlines, int n, int inverse, int variant, void* stream);

/* ------------------------------------------------------------------------------------------
 * Coil operators                               reference: reconstruction/models/varnet.py
 * ------------------------------------------------------------------------------------------ */

/* VarNetBlock.sens_reduce (varnet.py:187-194): out[b,t,h,w] = sum_c conj(S[b,c]) * ifft2c(k[b,t,c]).
 * sens (b, 1, c, h, w, 2).  out (b, t, 1, h, w, 2) complex, or with magnitude != 0 the
 * real (b, t, h, w) of VarNet.forward's final line (varnet.py:150-151, math.py:48-62).
 * `tmp` holds b*t*c*h*w complex (8 bytes each); tmp == k transforms k in place (k is destroyed). */
int cine_sens_reduce(const float* k, const float* sens, float* out, float* tmp,
                     int b, int t, int c, int h, int w, int magnitude, void* stream);

/* VarNetBlock.sens_expand (varnet.py:181-185) fused with the soft data-consistency of
 * VarNetBlock.forward (varnet.py:281-282):
 *   kth = fft2c(S * img);  out = mask ? (kth + v * kref) / (1 + v) : kth,  v = softplus(*lambda_dev)
 * img (b, t, 1, h, w, 2); kref (b, t, c, h, w, 2); mask uint8 (b, t, 1, h, 1, 1);
 * lambda_dev points to ONE float in device memory (cascades.N.lambda_reg).
 * kref == NULL  => plain sens_expand (no blend; mask / lambda_dev ignored).
 * hard_mask == 1 => out = mask ? kth : 0  (CineNet HOperator, cinenet.py:121-133; kref ignored).
 * hard_mask == 2 => out = mask ? kth - kref : 0  (XPDNet measurement residual of the masked forward
 *                   operator, xpdnet.py:128-131, 295-298). */
int cine_sens_expand_dc(const float* img, const float* sens, const float* kref, const uint8_t* mask,
                        const float* lambda_dev, float* out, int b, int t, int c, int h, int w,
                        int hard_mask, void* stream);

/* The same two operators split at "hybrid space" (image along h, k-space along w), so that a
 * cascade chain never writes k-space to HBM: the next cascade's sens_reduce (varnet.py:253) starts
 * with the column IFFT of exactly the tile the DC (varnet.py:281-282) just produced.
 *   cine_kspace_to_hybrid : centered column IFFT of nimg k-space images (first half of ifft2c)
 *   cine_hybrid_reduce    : row IFFT + conj(S) multiply + coil sum (second half of sens_reduce)
 *   cine_expand_dc_hybrid : sens_expand + DC as cine_sens_expand_dc, then the column IFFT, in one pass
 * cine_sens_reduce(k) == cine_hybrid_reduce(cine_kspace_to_hybrid(k)); and
 * cine_expand_dc_hybrid(...) == cine_kspace_to_hybrid(cine_sens_expand_dc(...)). hyb may alias k. */
int cine_kspace_to_hybrid(const float* k, float* hyb, long nimg, int h, int w, void* stream);
int cine_hybrid_reduce(const float* hyb, const float* sens, float* out,
                       int b, int t, int c, int h, int w, int magnitude, void* stream);
int cine_expand_dc_hybrid(const float* img, const float* sens, const float* kref, const uint8_t* mask,
                          const float* lambda_dev, float* hyb, int b, int t, int c, int h, int w,
                          int hard_mask, void* stream);

/* Image-space data consistency: one cascade's  sens_expand -> FFT2 -> DC -> [next cascade] IFFT2 -> sens_reduce
 * (models/varnet.py:181-194, 253, 281-282) as ONE operator on the coil-combined image, for Cartesian ROW masks
 * (mask (b, t, 1, h, 1, 1), the only layout the reference produces: data/transforms.py:341-343, subsample.py:146-150).
 * Mask and blend weights depend on the k-space row only, so they commute with the transform along w and
 *     out = sum_c conj(S_c) IFFT_h[ wgt(ky) * FFT_h(S_c img) ] + beta * zf,
 *     wgt = mask ? w_sampled : w_unsampled          (centered ortho 1-D transforms along h)
 * equals sens_reduce(DC(sens_expand(img))) with  zf = sens_reduce(mask * k_ref)  (constant over the cascades).
 *   lambda_dev != NULL : soft DC, v = softplus(*lambda_dev): w_sampled = 1/(1+v), w_unsampled = 1, beta = v/(1+v)
 *                        (the float arguments are ignored)
 *   lambda_dev == NULL : the float arguments are used as given, e.g. (1, 0, 0) = CineNet's normal operator A^H M A
 *                        (models/cinenet.py:121-133, 255-257), (1, 0, -1) = XPDNet's backward image A^H M (A x - k_ref)
 *                        (models/xpdnet.py:128-131, 161-167, 295-298).
 * img, zf (may be NULL), out: (b, t, h, w, 2); sens (b, 1, c, h, w, 2); mask uint8 (b, t, h).  magnitude != 0: out is
 * (b, t, h, w) = |result| (varnet.py:150-151).  out must not alias img.  ws: cine_image_dc_ws_bytes() of scratch
 * (per-coil-group partial sums of the h == 200 kernel; 0 bytes / NULL for other sizes or <= 4 coils). */
size_t cine_image_dc_ws_bytes(int b, int t, int c, int h, int w);
int cine_image_dc(const float* img, const float* sens, const float* zf, const uint8_t* mask,
                  const float* lambda_dev, float w_sampled, float w_unsampled, float beta,
                  float* out, int b, int t, int c, int h, int w, int magnitude,
                  void* ws, size_t ws_bytes, void* stream);
/* CineNet's normal operator with its regulariser weight (models/cinenet.py:121-133): out = A^H M A img + softplus(*lambda_dev) img
 * for a row mask -- cine_image_dc with weights (1, 0), zf = img and beta = softplus(lambda) read on the device.  Same
 * shapes / workspace as cine_image_dc. */
int cine_normal_op(const float* img, const float* sens, const uint8_t* mask, const float* lambda_dev,
                   float* out, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream);
/* The sensitivities in column-tile-major order [b][c][ceil(w / 5)][200][5] (complex pairs; zero past the last column), for the h == 200
 * image-space operators: a workgroup of imgdc200_kernel then reads each coil's 200 rows as one contiguous run instead of 40 bytes out of
 * every 128-byte line (38 -> 34 us per launch at cfg 2 / cfg 4).  The maps are constant over a forward pass (varnet.py:144, cinenet.py
 * forward's argument), so one pack serves its 6 - 42 operator applications.  cine_sens_tile_floats() == 0 where no kernel reads them
 * (h != 200); the *_t entry points take the tiled copy beside the plain one (NULL = read the plain maps: identical results either way). */
size_t cine_sens_tile_floats(int b, int c, int h, int w);
int cine_sens_tile_pack(const float* sens, float* tiled, int b, int c, int h, int w, void* stream);
int cine_image_dc_t(const float* img, const float* sens, const float* sens_tiled, const float* zf, const uint8_t* mask,
                    const float* lambda_dev, float w_sampled, float w_unsampled, float beta,
                    float* out, int b, int t, int c, int h, int w, int magnitude,
                    void* ws, size_t ws_bytes, void* stream);
int cine_normal_op_t(const float* img, const float* sens, const float* sens_tiled, const uint8_t* mask, const float* lambda_dev,
                     float* out, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream);
/* cine_normal_op that also leaves 256 partial sums of <img, out> in pd_part (device floats): the p.d of the conjugate-gradient step that
 * follows (models/cinenet.py:155-159), produced where `out` is produced instead of by a separate pass over both vectors; feed them to
 * cine_cg_step_pd (pd_part = the first 256 floats of its workspace).  CINE_EUNSUPPORTED where cine_image_dc_ws_bytes() is 0. */
int cine_normal_op_pd(const float* img, const float* sens, const uint8_t* mask, const float* lambda_dev,
                      float* out, float* pd_part, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream);

/* cine_kspace_to_hybrid of (mask * k) without reading the rows the mask drops: the hybrid-space image of the
 * measured lines only (the zero-filled term zf above: cine_hybrid_reduce of it).  k, hyb (bt, c, h, w, 2); mask (bt, h). */
int cine_masked_kspace_to_hybrid(const float* k, const uint8_t* mask, float* hyb, int bt, int c, int h, int w,
                                 void* stream);

/* ---- the steps either side of the path ------------------------------------------------------------ */

/* out = kspace * mask + 0.0 for a Cartesian row mask (reference data/transforms.py:66-92 `apply_mask`, :91).
 * kspace, out (bt, c, h, w, 2) (may alias); mask uint8 (bt, h). */
int cine_apply_mask(const float* kspace, const uint8_t* mask, float* out, long bt, int c, int h, int w, void* stream);

/* x[i] *= s, in place: the "backward" / "forward" normalisations of torch.fft on top of the ortho kernels
 * (reference utils/fftc.py:59-110 with norm=None, traintest_scripts/run_inference.py:66). */
int cine_scale(float* x, long n, float s, void* stream);

/* Zero-filled reconstruction of traintest_scripts/run_inference.py:64-67:
 * rss_complex(ifft2c(k, norm=None) * sqrt(h w), dim=coil) (utils/coil_combine.py:21-34) -> out (b, t, h, w).
 * k (b, t, c, h, w, 2); tmp: scratch of k's size (tmp == k destroys k). */
int cine_zero_filled_rss(const float* k, float* out, float* tmp, int b, int t, int c, int h, int w, void* stream);

/* SSIM / NMSE / PSNR / MSE of a reconstruction against its target, on the device (reference utils/evaluate.py:6-50 --
 * skimage structural_similarity / peak_signal_noise_ratio defaults -- after data/transforms.py:161-183
 * center_crop_to_smallest; utils/losses.py:25-58 for SSIMLoss's per-frame data range).  gt (t, hg, wg), pred (t, hp, wp)
 * float32; both are center-cropped to (min h, min w).  SSIM: win x win uniform window (odd, <= 11; reference 7), sample
 * covariance, K1, K2, mean over the window-valid region, then over frames; float64 arithmetic like skimage.
 * range_mode 0: data range = max of the cropped gt volume (evaluate.py:31); 1: max of each gt frame (losses.py:34);
 * 2: `maxval`.  out (device, 4 + t doubles) = {ssim, nmse, psnr, mse, ssim of frame 0..t-1}. */
size_t cine_image_metrics_ws_bytes(int t, int hg, int wg, int hp, int wp, int win);
int cine_image_metrics(const float* gt, const float* pred, int t, int hg, int wg, int hp, int wp,
                       int win, double k1, double k2, int range_mode, double maxval,
                       double* out, void* ws, size_t ws_bytes, void* stream);

/* SensitivityModel prologue (varnet.py:62-74): mean over frames, keep rows [row_lo, row_hi) of
 * dim h (transforms.mask_center, data/transforms.py:95-108), ifft2c.  k (b,t,c,h,w,2) -> out (b,c,h,w,2). */
int cine_sens_prologue(const float* k, float* out, int b, int t, int c, int h, int w,
                       int row_lo, int row_hi, void* stream);
/* The same without a host read-back of the mask (the reference finds the window on the host, varnet.py:64-68): cine_acs_window finds the fully
 * sampled centre rows of a row mask on the device -- mask_rows: the 1-D pattern of frame 0, n >= h float32 entries, 0 = not sampled;
 * window[0..1] = {pad, pad + n_low} with left = the last unsampled row below h / 2 (-1: none), right = the first one at or above it (n: none),
 * n_low = right - left, pad = (h - n_low + 1) / 2 -- and cine_sens_prologue_win reads {row_lo, row_hi} from that device buffer. */
int cine_acs_window(const float* mask_rows, int n, int h, int* window, void* stream);
int cine_sens_prologue_win(const float* k, float* out, int b, int t, int c, int h, int w, const int* window, void* stream);

/* SensitivityModel.divide_root_sum_of_squares (varnet.py:58-59, coil_combine.py:21-34), in place
 * on x (b, c, h, w, 2). */
int cine_rss_normalise(float* x, int b, int c, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------
 * NormUnet pre/post and the XT/XF rotations    reference: denoisers/norm_unet.py, varnet.py:196-241
 * ------------------------------------------------------------------------------------------ */

/* padded size of NormUnet.pad (norm_unet.py:76-86): ((n - 1) | 15) + 1 */
int cine_pad16(int n);

/* NormUnet front half on `n` complex images (n, h, w, 2): complex_to_chan_dim (:48-51), group
 * norm with unbiased std (:59-69), zero pad to x16 (:76-86).
 * planes (n, 2, hp, wp) fp32; stats (n, 2, 2) = {mean, std} per (sample, re|im).
 * norm == 0: just the (re, im) -> 2-channel repack with no normalisation and no padding, planes
 * (n, 2, h, w) -- the layout CineNet feeds its bare Unet (cinenet.py:242). */
int cine_normunet_pack(const float* x, float* planes, float* stats, int n, int h, int w, int norm, void* stream);

/* NormUnet back half (:88-96 unpad, :71-74 unnorm, :53-57 chan_complex_to_last_dim).
 * stats == NULL: inverse of the norm == 0 repack (cinenet.py:243-244). */
int cine_normunet_unpack(const float* planes, const float* stats, float* y, int n, int h, int w, void* stream);

/* VarNetBlock.xfyf_transform front half (varnet.py:202-217) + both NormUnet front halves:
 * temporal mean subtract, (xf != 0) centered temporal DFT, rotation into x-f / y-f planes,
 * group norm, pad.  img (b, t, h, w, 2).
 *   planes_xf (b*h, 2, pad16(w), pad16(t)), planes_yf (b*w, 2, pad16(h), pad16(t)),
 *   stats_xf (b*h, 2, 2), stats_yf (b*w, 2, 2), mean_img (b, h, w, 2).
 * ws: cine_xfyf_ws_bytes(b, t, h, w). */
size_t cine_xfyf_ws_bytes(int b, int t, int h, int w);
int cine_xfyf_pack(const float* img, float* planes_xf, float* planes_yf, float* stats_xf, float* stats_yf,
                   float* mean_img, int b, int t, int h, int w, int xf, int norm, void* ws, size_t ws_bytes, void* stream);

/* norm == 0 gives CineNet's variant (cinenet.py:181-195): same rotation, planes (b*h, 2, w, t) /
 * (b*w, 2, h, t) unnormalised and unpadded; stats may then be NULL (also in cine_xfyf_unpack).
 * back half (varnet.py:229-241 / cinenet.py:206-219): unpad + unnorm both planes, un-rotate, average,
 * inverse temporal DFT, add the temporal mean.  out (b, t, 1, h, w, 2). */
int cine_xfyf_unpack(const float* planes_xf, const float* planes_yf, const float* stats_xf,
                     const float* stats_yf, const float* mean_img, float* out,
                     int b, int t, int h, int w, int xf, void* stream);

/* ------------------------------------------------------------------------------------------
 * U-Net regulariser                            reference: denoisers/unet.py
 * ------------------------------------------------------------------------------------------ */

/* Weight repacking into the MFMA staging layout [chunk][tap][ck][rows padded to 16]:
 *   conv3x3 (cout, cin, 3, 3) (unet.py:160,164); tconv (cin, cout, 2, 2) (unet.py:213-215) as a
 *   1x1 GEMM with 4*cout rows; conv1x1 (cout, cin) (unet.py:69).  *_packed_floats sizes `packed`. */
size_t cine_conv3x3_packed_floats(int cout, int cin);
size_t cine_tconv2x2_packed_floats(int cin, int cout);
size_t cine_conv1x1_packed_floats(int cout, int cin);
int cine_pack_conv3x3(const float* w, float* packed, int cout, int cin, void* stream);
int cine_pack_tconv2x2(const float* w, float* packed, int cin, int cout, void* stream);
int cine_pack_conv1x1(const float* w, float* packed, int cout, int cin, void* stream);

/* InstanceNorm statistics travel as PARTIAL records {count, mean, M2}: a tensor (n, c, h, w) carries
 * part (n, c, np, 3).  The conv / tconv kernels emit one record per (sample, channel, output tile)
 * (np = cine_conv_stat_partials(cout, h, w, is_tconv), h/w = the kernel's INPUT grid for tconv);
 * consumers merge them (Chan) into mean and rstd = 1/sqrt(M2/count + eps): biased variance, as
 * nn.InstanceNorm2d (unet.py:161,165,216). */
int cine_conv_stat_partials(int cout, int h, int w, int is_tconv);

/* One ConvBlock half (unet.py:159-162): y = conv3x3(X, pad 1, no bias) and the partial statistics of y.
 * The normalise + LeakyReLU(slope) of y is applied by whichever kernel consumes (y, part_y) next; 0 <= slope <= 1
 * (leaky_relu is evaluated as max(v, v * slope); other slopes are rejected with CINE_EINVAL).
 * X = channel concat (torch.cat, unet.py:122) of up to two sources; source s has c_s channels,
 * extent (h_s, w_s), np_s partial records and mode_s:
 *   0 = use as is;  1 = InstanceNorm + LeakyReLU on load;
 *   2 = mode 1 followed by 2x2 average pool (unet.py:97; the source extent is then ~(2h, 2w)).
 * A source smaller than (h, w) reads as zero outside its extent (up-path zero pad, unet.py:106-120).
 * wpacked2 != NULL: samples >= set_split use wpacked2 (two weight sets in one launch).
 * part_y may be NULL (no statistics). */
int cine_conv3x3_in(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                    const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1,
                    const float* wpacked, const float* wpacked2, int set_split,
                    float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream);

/* Extended form used by the wavelet CNN (denoisers/mwcnn.py): source modes additionally take
 *   3 = Haar DWT of the source (mwcnn.py:224-236): 4 c_s channels [LL, HL, LH, HH] at (h_s/2, w_s/2)
 *   4 = Haar IWT of the source (mwcnn.py:252-261): c_s/4 channels at (2 h_s, 2 w_s)
 * with bit 3 (| 8) set when the source is raw and must be InstanceNorm + LeakyReLU'd first;
 * add_src1 != 0 ADDS source 1 to source 0 channel-wise (the MWCNN skips, :164,172) instead of
 * concatenating; bias (cout) may be NULL.
 * Epilogue for the convolutional-RNN cells (models/recurrent_varnet.py:153-200): y = conv + bias + addend,
 * then ReLU when relu != 0; addend (n, cout, h, w) may be NULL.  A sum of convolutions of different inputs
 * (i2h + h2h + ih2ih, :172-178) is one convolution over the concatenated inputs with concatenated weights. */
int cine_conv3x3_ex(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                    const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1, int add_src1,
                    const float* wpacked, const float* bias, const float* addend, int relu,
                    float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream);
/* cine_conv3x3_ex with two weight / bias sets in one launch: samples [0, set_split) use (wpacked, bias), the rest (wpacked2, bias2)
 * (two networks of one topology on planes of equal shape: XPDNet's x-t / y-t MWCNNs, xpdnet.py:424-446). */
int cine_conv3x3_ex2(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                     const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1, int add_src1,
                     const float* wpacked, const float* bias, const float* wpacked2, const float* bias2, int set_split,
                     const float* addend, int relu,
                     float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream);

/* One step of a convolutional-RNN time sweep (reference models/recurrent_varnet.py:241-254, CRNNcell :172-178 with the
 * input terms precomputed): y = ReLU(conv3x3(x; wpacked) + addend), and accum += y when accum != NULL (the backward sweep
 * adding onto the forward sweep's outputs, :254).  x, addend, y, accum (n, c, h, w); wpacked = cine_pack_conv3x3 of the
 * (c, c, 3, 3) hidden-to-hidden weight (its bias belongs in addend).  y / accum must not alias x or each other. */
int cine_crnn_step(const float* x, const float* wpacked, const float* addend, float* y, float* accum,
                   int n, int c, int h, int w, int relu, void* stream);
/* Both directions of the BCRNN time sweep in one launch (recurrent_varnet.py:241-252: two independent chains, summed at :254):
 * set f and set b are each a cine_crnn_step on their own tensors; store_* != 0 writes accum_* = y_* instead of adding (the
 * first direction to reach a frame).  x_b == NULL runs set f alone.  The sets must not write what the other reads, and must
 * not accumulate into the same tensor. */
int cine_crnn_step2(const float* x_f, const float* addend_f, float* y_f, float* accum_f, int store_f,
                    const float* x_b, const float* addend_b, float* y_b, float* accum_b, int store_b,
                    const float* wpacked, int n, int c, int h, int w, int relu, void* stream);
/* `relu`: 1 = the cell's nn.ReLU (recurrent_varnet.py:178), 0 = no activation (the identity-activation gradient fixtures of
 * tests/test_hip_grad.py, made by the reference with F.relu patched out). */
/* BOTH time sweeps of a BCRNN layer (BCRNNlayer.forward, recurrent_varnet.py:236-254, batch 1) in one call -- the reference's two Python loops
 * over the frames: h_t = [ReLU](conv3x3(h_prev; W_h2h) + P_t) forward in time and backward in time, hid_init = `zero` ((c, h, w) zeros, read
 * only), one cine_crnn_step2 pair launch per step (T launches; T + 1 for odd T, where both chains reach the middle frame in the same step),
 * out = hidden_f + hidden_b (the first chain to reach a frame stores, the second adds: no zero fill, no extra pass).  P (T, c, h, w) = the
 * input terms i2h(x_t) + ih2ih(hid_iter_t) + the three biases for ALL frames (:172-176, one batched launch by the caller), out (T, c, h, w).
 * keep != 0: hf / hb (T, c, h, w) receive every hidden state of the chain (training); keep == 0: hf / hb are (2, c, h, w) ping-pong buffers. */
int cine_bcrnn_sweep(const float* P, const float* wpacked_hh, const float* zero, float* hf, float* hb, float* out,
                     int T, int c, int h, int w, int relu, int keep, void* stream);
/* Back-propagation through time of cine_bcrnn_sweep (keep != 0) -- what torch.autograd unrolls over the reference's two loops: from gout =
 * d loss / d out,  gf[t] = [hf[t] > 0] (conv(gf[t + 1]; Wd) + gout[t]) for t = T-1 .. 0 and gb[t] = [hb[t] > 0] (conv(gb[t - 1]; Wd) + gout[t]) for
 * t = 0 .. T-1, Wd = cine_pack_conv3x3_dgrad of W_h2h (the forward conv kernel; gout rides in as the addend, the ReLU mask is the epilogue's
 * gate; relu == 0: no mask), and gP = gf + gb = d loss / d P.  One pair launch per step; gf, gb, gP (T, c, h, w).  The weight gradients are the
 * caller's: dW_h2h from (hf[t - 1], gf[t]) and (hb[t + 1], gb[t]) with cine_conv3x3_wgrad, everything upstream of P from gP. */
int cine_bcrnn_sweep_bwd(const float* gout, const float* wpacked_hh_dgrad, const float* zero, const float* hf, const float* hb,
                         float* gf, float* gb, float* gP, int T, int c, int h, int w, int relu, void* stream);

/* TransposeConvBlock (unet.py:212-217): y (n, cout, 2h, 2w) = conv_transpose2d(act(x), k 2, s 2, no bias)
 * and the partial statistics of y.  x mode 0|1 as above. */
int cine_tconv2x2_in(const float* x, const float* part_x, int np_x, int mode,
                     const float* wpacked, const float* wpacked2, int set_split,
                     float* y, float* part_y, int n, int cin, int cout, int h, int w,
                     float eps, float slope, void* stream);

/* final 1x1 conv with bias (unet.py:69).  x mode 0|1 as above. */
int cine_conv1x1_bias(const float* x, const float* part_x, int np_x, int mode,
                      const float* wpacked, const float* bias, const float* wpacked2, const float* bias2,
                      int set_split, float* y, int n, int cin, int cout, int h, int w,
                      float eps, float slope, void* stream);

/* Stand-alone statistics of `planes` planes of plane_elems floats: part (planes, 1, 3). */
int cine_instnorm_partials(const float* x, float* part, long planes, long plane_elems, void* stream);
/* merge np partials per plane into stats (planes, 2) = {mean, rstd}. */
int cine_instnorm_finalize(const float* part, float* stats, long planes, int np, float eps, void* stream);
/* materialise act(x) = LeakyReLU((x - mean) * rstd) (block-level parity / debugging). */
int cine_instnorm_lrelu_apply(const float* x, const float* part, int np, float* y, long planes, long plane_elems,
                              float eps, float slope, void* stream);

/* Whole 2-D U-Net (unet.py:73-125) on n planes (n, in_ch, h, w) -> (n, out_ch, h, w).
 * `weights` is a HOST array of device pointers in this order:
 *   for d in 0..pools-1: down[d].conv1 (packed), down[d].conv2 (packed)
 *   bottleneck conv1 (packed), conv2 (packed)
 *   for u in 0..pools-1: tconv[u] (packed), up[u].conv1 (packed), up[u].conv2 (packed)
 *   final 1x1 weight (packed), final bias (out_ch)
 * i.e. 2*pools + 2 + 3*pools + 2 pointers.
 * `nsets` == 2 runs two weight sets over the two halves of the n planes in the SAME launches
 * (the xf and yf U-Nets of one cascade, varnet.py:224-226); `weights` then holds two such arrays
 * back to back. */
size_t cine_unet2d_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools);
int cine_unet2d_forward(const float* x, float* y, const void* const* weights, int nsets,
                        int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                        void* ws, size_t ws_bytes, void* stream);
/* `slope`: the LeakyReLU slope of every ConvBlock (unet.py:162 hard-wires 0.2; 0 <= slope <= 1, 1 = identity -- how
 * tests/test_hip_grad.py compares full-size gradients with reference fixtures made without activation kinks). */

/* ---- 3-D variants (reference unet.py with dims = 3, norm_unet.py:117-219; VarNet / CineNet dynamic_type '3D') ----
 * Volumes are (n, c, d, h, w).  Conv3d 3x3x3 pad 1 (unet.py:48-49), ConvTranspose3d k2 s2 as a GEMM with 8*cout rows,
 * 1x1x1 conv + bias; on-load modes 0 / 1 / 2 (2 = avg_pool3d 2x2x2), concat of two sources, up-path zero pad, bias /
 * addend / ReLU epilogue as in 2-D.  A volume emits cine_conv_stat_partials3d() statistics records per (sample,
 * channel); cine_instnorm_merge folds np records into one. */
size_t cine_conv3d_packed_floats(int cout, int cin);
size_t cine_tconv3d_packed_floats(int cin, int cout);
int cine_pack_conv3d(const float* w, float* packed, int cout, int cin, void* stream);     /* (cout, cin, 3, 3, 3) */
int cine_pack_tconv3d(const float* w, float* packed, int cin, int cout, void* stream);    /* (cin, cout, 2, 2, 2) */
int cine_conv_stat_partials3d(int cout, int d, int h, int w, int is_tconv);
int cine_conv3d_in(const float* x0, const float* part0, int np0, int c0, int mode0, int d0, int h0, int w0,
                   const float* x1, const float* part1, int np1, int c1, int mode1, int d1, int h1, int w1,
                   const float* wpacked, const float* bias, const float* addend, int relu,
                   float* y, float* part_y, int n, int cout, int d, int h, int w, float eps, float slope, void* stream);
int cine_tconv3d_in(const float* x, const float* part_x, int np_x, int mode, const float* wpacked,
                    float* y, float* part_y, int n, int cin, int cout, int d, int h, int w,
                    float eps, float slope, void* stream);
int cine_conv1x1x1_bias(const float* x, const float* part_x, int np_x, int mode, const float* wpacked,
                        const float* bias, float* y, int n, int cin, int cout, int d, int h, int w,
                        float eps, float slope, void* stream);
int cine_instnorm_merge(const float* part, float* out, long planes, int np, void* stream);
/* y (planes, d/2, h/2, w/2) = avg_pool3d(LeakyReLU(InstanceNorm3d(x)), 2) (unet.py:88,97 with dims = 3; part: np statistics records per
 * plane), and the planning query that goes with it: cine_conv3d_pools_on_load() == 1 when cine_conv3d_in serves a mode-2 source of this
 * layer shape inside an efficient kernel (the coarse levels), 0 when the caller does better to pool once with cine_pool3d_act and pass
 * the result as a plain source (what cine_unet3d_forward does for its level 1; same arithmetic, same summation order). */
int cine_pool3d_act(const float* x, const float* part, int np, float* y, long planes, int d, int h, int w,
                    float eps, float slope, void* stream);
int cine_conv3d_pools_on_load(int cout, int d, int h, int w);
/* Whole 3-D U-Net (unet.py:73-125, dims = 3); weights ordered as for cine_unet2d_forward (one set), packed with
 * cine_pack_conv3d / cine_pack_tconv3d / cine_pack_conv1x1. */
size_t cine_unet3d_ws_bytes(int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools);
int cine_unet3d_forward(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                        int in_ch, int out_ch, int chans, int pools, float slope, void* ws, size_t ws_bytes, void* stream);
/* NormUnet3D front / back halves (norm_unet.py:149-219): x (n, t, h, w, 2) -> planes (n, 2, pad16(t), pad16(h),
 * pad16(w)) with group norm (unbiased std) and stats (n, 2, 2); norm == 0 / stats == NULL: plain unpadded repack
 * (CineNet's bare 3-D Unet, cinenet.py:251-253). */
int cine_normunet3d_pack(const float* x, float* planes, float* stats, int n, int t, int h, int w, int norm, void* stream);
int cine_normunet3d_unpack(const float* planes, const float* stats, float* y, int n, int t, int h, int w, void* stream);

/* Whole MWCNN (denoisers/mwcnn.py:135-179) on n planes (n, in_ch, h, w) -> (n, out_ch, h, w); h, w
 * multiples of 2^n_scales (utils/padding.py pads before the call).  Handles the topology XPDNet builds
 * (n_first_convs = 1, res = False, xpdnet.py:251-262); others return CINE_EUNSUPPORTED.
 * `weights` (host array of device pointers, all 3x3 packed with cine_pack_conv3x3):
 *   first_convs[0]; conv_blocks_per_scale[s][i] for s, i in module order; first_convs[1] weight, bias. */
size_t cine_mwcnn_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters,
                           const int* n_convs, int first_filters);
int cine_mwcnn_forward(const float* x, float* y, const void* const* weights, int n, int h, int w,
                       int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                       int n_first_convs, int first_filters, int res, float slope, void* ws, size_t ws_bytes, void* stream);
/* `slope`: the LeakyReLU slope of every conv block (mwcnn.py:204 hard-wires 0.2; 0 <= slope <= 1, 1 = identity). */
/* Two MWCNNs of one topology in ONE launch sequence: samples [0, set_split) through `weights`, the rest through `weights2`
 * (same pointer-array order).  XPDNet's image networks for the x-t and the y-t planes (xpdnet.py:424-446) when both plane sets
 * have the same shape. */
int cine_mwcnn_forward2(const float* x, float* y, const void* const* weights, const void* const* weights2, int set_split,
                        int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                        int n_first_convs, int first_filters, int res, float slope, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * XPDNet primal-buffer plumbing                reference: models/xpdnet.py:301-326, 406-509
 * ------------------------------------------------------------------------------------------ */
/* utils/padding.py:26-47 for one dimension: returns the padded size (multiple of 2^n_scales); odd sizes
 * get the extra element on the left. */
int cine_mwcnn_pad(int size, int n_scales, int* left, int* right);
/* I-step front half (:424-471): buf (b,t,1,h,w,2n) + backward-op image `extra` (b,t,1,h,w,2) as complex
 * channel n, temporal mean subtract, (xf) ifftshift(fft(fftshift(.))) over t, rotation into
 * planes_xf (b*h, 2(n+1), pad(w), pad(t)) / planes_yf (b*w, 2(n+1), pad(h), pad(t)); mean (b,h,w,n+1,2). */
size_t cine_xpd_ws_bytes(int b, int t, int h, int w, int n_primal);
int cine_xpd_pack(const float* buf, const float* extra, float* planes_xf, float* planes_yf, float* mean,
                  int b, int t, int h, int w, int n_primal, int n_scales, int xf, void* ws, size_t ws_bytes, void* stream);
/* I-step back half (:485-509) on the MWCNN outputs (2n channels): unpad, un-rotate, average, inverse
 * temporal transform, add the temporal mean of channels 0..n-1.  out (b,t,1,h,w,2n). */
int cine_xpd_unpack(const float* planes_xf, const float* planes_yf, const float* mean, float* out,
                    int b, int t, int h, int w, int n_primal, int n_scales, int xf, void* stream);
/* 2-D mode repacks (:442-444): channel-last (n, h, w, c) <-> zero-padded planes (n, c, pad(h), pad(w)). */
int cine_chanlast_to_planes(const float* x, float* planes, int n, int c, int h, int w, int n_scales, void* stream);
int cine_planes_to_chanlast(const float* planes, float* y, int n, int c, int h, int w, int n_scales, void* stream);
/* complex image from channels (c_re, c_im) of a channel-last buffer (:128, :161, :321-326); and
 * repeat_interleave of a complex image into an n-fold buffer (:306-307). */
int cine_extract_complex(const float* buf, float* out, long npix, int c, int c_re, int c_im, void* stream);
int cine_repeat_complex(const float* img, float* buf, long npix, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Conjugate-gradient vector ops                reference: reconstruction/models/cinenet.py:136-171
 * ------------------------------------------------------------------------------------------ */
/* *out_dev = sum_i a[i] * b[i] over n floats (torch.dot on the flattened (re, im) pairs, :148,155,163);
 * deterministic two-stage reduction; ws holds cine_dot_ws_bytes(). */
size_t cine_dot_ws_bytes(void);
int cine_dot(const float* a, const float* b, long n, float* out_dev, void* ws, void* stream);
/* out = a + sign * s * b with s read from DEVICE memory: s = *num_dev / *den_dev (den_dev NULL -> 1), or
 * s = softplus(*lambda_dev) when num_dev is NULL.  Covers x + alpha p, r - alpha d, r + beta p (:159-169),
 * the rhs x_ref + v x_reg (:255-257) and H's "+ v x" (:133) without the reference's .item() host syncs. */
int cine_axpby_dev(float* out, const float* a, const float* b, long n, const float* num_dev, const float* den_dev,
                   const float* lambda_dev, float sign, void* stream);
/* One conjugate-gradient iteration after d = H p (:155-169): alpha = rr_old / (p.d); x += alpha p; r -= alpha d; rr_new = r.r;
 * p = r + (rr_new / rr_old) p -- three launches, scalars in device memory, bit-identical to the cine_dot / cine_axpby_dev
 * sequence.  x, r, p updated in place; rr_old_dev != rr_new_dev; ws holds cine_cg_ws_bytes(). */
size_t cine_cg_ws_bytes(void);
int cine_cg_step(float* x, float* r, float* p, const float* d, long n, const float* rr_old_dev, float* rr_new_dev,
                 void* ws, void* stream);
/* cine_cg_step without its p.d pass: ws[0..256) floats already hold the partial sums (cine_normal_op_pd).  Two launches. */
int cine_cg_step_pd(float* x, float* r, float* p, const float* d, long n, const float* rr_old_dev, float* rr_new_dev,
                    void* ws, void* stream);
/* One whole conjugate-gradient iteration of models/cinenet.py:153-169 for a row mask in three launches (h == 200, more than 5 coils;
 * cine_cg_fused_ws_bytes() == 0 otherwise): the operator kernel leaves its coil groups' partial sums in ws_dc (cine_image_dc_ws_bytes) and
 * one partial sum of <p, H p> per workgroup in ws_cg; the update kernel adds the groups + softplus(lambda) p on the fly (H p is never
 * written), alpha, x += alpha p, r -= alpha H p, r.r; the direction kernel beta and p.  x, r, p in place; *pd_out_dev (may be NULL)
 * receives p.Hp.  Same values as cine_normal_op_pd + cine_cg_step_pd up to the summation order of p.Hp. */
size_t cine_cg_fused_ws_bytes(int b, int t, int c, int h, int w);
int cine_normal_op_cg_fused(float* x, float* r, float* p, const float* sens, const uint8_t* mask, const float* lambda_dev,
                            const float* rr_old_dev, float* rr_new_dev, float* pd_out_dev, int b, int t, int c, int h, int w,
                            void* ws_dc, size_t ws_dc_bytes, void* ws_cg, size_t ws_cg_bytes, void* stream);
int cine_normal_op_cg_fused_t(float* x, float* r, float* p, const float* sens, const float* sens_tiled, const uint8_t* mask,
                              const float* lambda_dev, const float* rr_old_dev, float* rr_new_dev, float* pd_out_dev,
                              int b, int t, int c, int h, int w, void* ws_dc, size_t ws_dc_bytes, void* ws_cg, size_t ws_cg_bytes, void* stream);
/* The WHOLE conjugate-gradient solve of CineNet's data-consistency block (models/cinenet.py:136-171: H x = b with exactly `iters`
 * iterations, H = A^H M A + softplus(lambda) I, row mask) in 2 + 2 * iters launches: x (b, t, 1, h, w, 2) holds the start value and
 * receives the solution, rhs = b -- or, with rhs_is_ref != 0, `rhs` holds x_ref and b = x_ref + softplus(lambda) x is formed inside
 * (cinenet.py:106-107: x is then the regulariser's output, start value and regularisation target at once).  Per iteration one operator kernel -- which forms the new direction p = r + beta p on load and
 * needs no direction pass -- and one update kernel (alpha, x, r, the partial sums of r.r).  Same arithmetic as cine_normal_op_cg_fused
 * iterated, except for the summation order of the first r.r.  h == 200 and more than 5 coils (else CINE_EUNSUPPORTED: iterate
 * cine_normal_op / cine_cg_step); sens_tiled: cine_sens_tile_pack's copy or NULL.  ws: cine_conj_grad_ws_bytes(). */
size_t cine_conj_grad_ws_bytes(int b, int t, int c, int h, int w);
int cine_conj_grad(float* x, const float* rhs, int rhs_is_ref, const float* sens, const float* sens_tiled, const uint8_t* mask,
                   const float* lambda_dev, int iters, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream);
/* cine_conj_grad for TRAINING: the same 2 + 2 * iters launches (+ one 1-block sum), recording what the adjoint recurrence of the iteration needs --
 * the reference detaches its step sizes (cinenet.py:159-169: alpha.item(), beta.item()), so what torch.autograd differentiates is linear in
 * (x0, b) with these constants: p_rec (iters, b, t, 1, h, w, 2) = every direction p_k, rr_rec (iters + 1) = r_k . r_k, pd_rec (iters) = p_k . H p_k. */
int cine_conj_grad_rec(float* x, const float* rhs, int rhs_is_ref, const float* sens, const float* sens_tiled, const uint8_t* mask,
                       const float* lambda_dev, int iters, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes,
                       float* p_rec, float* rr_rec, float* pd_rec, void* stream);
/* cine_cg_step_pd that also stores p.d into *pd_out_dev (training: the adjoint recurrence needs alpha_k = rr_k / pd_k). */
int cine_cg_step_pd2(float* x, float* r, float* p, const float* d, long n, const float* rr_old_dev, float* rr_new_dev,
                     float* pd_out_dev, void* ws, void* stream);
/* Adjoint of ONE conjugate-gradient iteration with recorded step sizes (the reference detaches alpha / beta, cinenet.py:159-169), run from
 * the last iteration to the first: with q = gr_{k+1} + gp_{k+1} and hg = H(q) (cine_normal_op),
 *     gp <- (rr_new / rr) gp + (rr / pd) gx - (rr / pd) hg;   part[0..256) <- partial sums of <q, p_k>;   q <- q + gp
 * in one launch, bit-identical to the cine_axpby_dev / cine_dot / add sequence.  cine_cg_adjoint_finish: gv = - sum_k (rr[k] / pd[k]) <q_k, p_k>
 * from `iters` consecutive partial-sum blocks of cine_cg_adjoint_part_floats() floats (d loss / d softplus(lambda) of the solve, without the
 * - <gb, x0> term). */
size_t cine_cg_adjoint_part_floats(void);
int cine_cg_adjoint_step(float* gp, float* q, const float* gx, const float* hg, const float* pk, long n, const float* rr_dev,
                         const float* pd_dev, const float* rr_new_dev, float* part, void* stream);
int cine_cg_adjoint_finish(const float* part, const float* rr_dev, const float* pd_dev, int iters, float* gv_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * data front-end + sensitivity calibration (the step BEFORE the path)
 *                                              reference: reconstruction/data/mri_data.py:283-303,
 *                                                         reconstruction/data/transforms.py:186-220, 425-432
 * ------------------------------------------------------------------------------------------ */
/* data[:t_out, :, y0:y0+hout, x0:x0+wout] with y0 = (hin - hout) / 2, x0 = (win - wout) / 2 (transforms.py:209-214):
 * in (t_in, c, hin, win, 2) -> out (t_out, c, hout, wout, 2). */
int cine_crop_select(const float* in, float* out, int t_in, int c, int hin, int win, int t_out, int hout, int wout, void* stream);
/* One axis pass of scipy.ndimage.gaussian_filter as transforms.py:216-217 calls it (mode 'reflect', truncate 4.0, weights and
 * accumulation in float64, float32 result) over a (outer, n, inner) array of complex pairs: the real and imaginary parts
 * are filtered alike.  The caller runs one pass per axis with sigma > 0, in axis order (mri_data.py:279: [0.7, 0, 0.3, 0.3]). */
int cine_gauss_axis(const float* in, float* out, long outer, int n, long inner, double sigma, void* stream);
/* target = center_crop(|sum_c img * conj(sens)|, (ch, cw))  (mri_data.py:302-303, transforms.py:136-158):
 * img (t, c, h, w, 2), sens (c, h, w, 2) -> out (t, ch, cw). */
int cine_combine_target(const float* img, const float* sens, float* out, int t, int c, int h, int w, int ch, int cw, void* stream);
/* ESPIRiT calibration in place of `bart ecalib` (mri_data.py:296 `-r 200`, transforms.py:429 `-r 15`; BART is a third-party
 * program the reference shells out to).  proj = V V^H (kk*kk*c square, complex pairs, index (py, px, coil)) is the
 * projector onto the row space of the calibration matrix; cine_espirit_lag_kernels writes the (2 kk - 1)^2 lag kernels of
 * every coil pair, centered and scaled, into kpad (c*c, ny, nx, 2) so that cine_fft2c(kpad, inverse) is the c x c
 * image-space operator M(r) of every pixel.  cine_espirit_eig: dominant eigenpair of M(r) by `iters` power iterations;
 * maps (c, npix, 2) unit norm with coil 0 real and non-negative, zero where the eigenvalue < crop; lam (npix). */
int cine_espirit_lag_kernels(const float* proj, float* kpad, int c, int kk, int ny, int nx, void* stream);
int cine_espirit_eig(const float* m, float* maps, float* lam, int c, long npix, int iters, float crop, void* stream);

/* ------------------------------------------------------------------------------------------
 * small element-wise helpers                   reference: reconstruction/utils/math.py
 * ------------------------------------------------------------------------------------------ */
int cine_complex_abs(const float* x, float* y, long n, void* stream);          /* math.py:48-62 */

/* ------------------------------------------------------------------------------------------
 * Training: gradients through the path
 *   reference: what torch.autograd differentiates in reconstruction/pl_modules/varnet_module.py:97-113 (training_step:
 *   forward + SSIMLoss) -- models/varnet.py:143-151, 181-282; denoisers/norm_unet.py:59-114; denoisers/unet.py:73-125, 159-218
 * Conventions: the gradient of a complex tensor is the pair (d loss / d re, d loss / d im), i.e. d loss = Re(conj(g) dx);
 * weight gradients are ACCUMULATED (+=) into caller-zeroed tensors in the parameters' own layouts; every reduction runs in a
 * fixed order (no atomics): two backward passes over the same data give bit-identical gradients.
 * ------------------------------------------------------------------------------------------ */

/* Input-gradient packings of the U-Net weights: the gradient of a convolution with respect to its input is a convolution of the
 * output gradient -- conv3x3 with (cout, cin) swapped and the taps flipped (unet.py:160,164); the k2 s2 transpose conv with its
 * (cin, 4 cout) weight matrix over the space-to-depth view of the output gradient (unet.py:213-215); the 1x1 conv with the
 * transposed matrix (unet.py:69).  `w` are the parameters in their own layouts. */
size_t cine_conv3x3_dgrad_packed_floats(int cout, int cin);
size_t cine_tconv2x2_dgrad_packed_floats(int cin, int cout);
size_t cine_conv1x1_dgrad_packed_floats(int cout, int cin);
int cine_pack_conv3x3_dgrad(const float* w, float* packed, int cout, int cin, void* stream);
int cine_pack_tconv2x2_dgrad(const float* w, float* packed, int cin, int cout, void* stream);
int cine_pack_conv1x1_dgrad(const float* w, float* packed, int cout, int cin, void* stream);
/* Batched packing (training re-packs every weight after every optimiser step: hundreds of 4-us launches per step otherwise).  cine_pack_desc writes
 * into HOST memory (cine_pack_desc_bytes() each) the descriptor of one tensor: op 0 = cine_pack_conv3x3 (n1 = cout, n2 = cin), 1 = cine_pack_tconv2x2
 * (cin, cout), 2 = cine_pack_conv1x1 (cout, cin), 3 = cine_pack_conv3x3_dgrad (cout, cin), 4 = cine_pack_tconv2x2_dgrad (cin, cout),
 * 5 = cine_pack_conv1x1_dgrad (cout, cin): the same arguments, layout and size as that entry point.  The caller copies the array to the device once
 * (parameters and packed buffers keep their addresses across steps); cine_pack_batch re-packs all n of them in ONE launch (max_total: the largest
 * packed size in floats, for the grid). */
size_t cine_pack_desc_bytes(void);
int cine_pack_desc(void* desc_host, int op, const float* w, float* packed, int n1, int n2);
int cine_pack_batch(const void* descs_dev, int n, long max_total, void* stream);
/* gx = d loss / d (conv input) from gy = d loss / d (raw conv output); wpacked* from the packings above (two sets: samples
 * >= set_split use wpacked2, NULL = one set).  conv3x3: gy (n, cout, h, w) -> gx (n, cin, h, w); tconv: gy (n, cout, 2h, 2w) ->
 * gx (n, cin, h, w); conv1x1: gy (n, cout, h, w) -> gx (n, cin, h, w). */
int cine_conv3x3_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                       float* gx, int n, int cout, int cin, int h, int w, void* stream);
int cine_tconv2x2_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                        float* gx, int n, int cin, int cout, int h, int w, void* stream);
/* cine_conv3x3_dgrad with the chain rule of a ReLU'd input in the epilogue: gx = gate > 0 ? (conv(gy) + addend) : 0, where gate (n, cin, h, w)
 * is the stored ReLU output that was the conv's input and addend (n, cin, h, w) the gradient it receives from its other consumers (either
 * may be NULL).  The conv blocks of the CRNN body (reference recurrent_varnet.py:122-134): x_k feeds conv_{k+1}_x AND the next cascade's
 * conv_k_h, torch.autograd adds the two gradients and applies the ReLU mask in separate kernels. */
int cine_conv3x3_dgrad_gated(const float* gy, const float* wpacked, const float* addend, const float* gate,
                             float* gx, int n, int cout, int cin, int h, int w, void* stream);
int cine_conv1x1_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                       float* gx, int n, int cout, int cin, int h, int w, void* stream);

/* cine_unet2d_forward that KEEPS every layer's raw output and statistics in `ws` (cine_unet2d_train_ws_bytes) for the backward
 * pass; same arguments and results as cine_unet2d_forward. */
size_t cine_unet2d_train_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools);
int cine_unet2d_forward_train(const float* x, float* y, const void* const* weights, int nsets,
                              int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                              void* ws, size_t ws_bytes, void* stream);
/* cine_unet2d_forward / cine_unet2d_forward_train (`train` != 0) with the n planes cut into nside + 1 contiguous runs that go through the
 * SAME launch sequence concurrently: run 0 on `stream`, run k on side[k - 1] (HOST array of caller-owned streams, none of them `stream`),
 * forked from `stream` and joined back into it with events before the call returns -- on return everything is ordered on `stream`
 * again.  The planes of a U-Net pass are independent (the x-f and y-f networks of a cascade, reference varnet.py:216-232, meet only in
 * the sum behind them; the coils of the sensitivity network, varnet.py:76-86, only in the RSS), but on one stream every layer boundary
 * drains the chip (a cfg-2 level-1 layer is 800 workgroups on 768 resident slots): with two runs each one's next layer fills the slots
 * the other's last round leaves empty.  Same kernels, tiles and per-plane statistics records: outputs are BIT-IDENTICAL to the
 * one-stream call.  nsets == 2: nside + 1 must be even (a run never spans both weight sets).  Workspace: cine_unet2d_branch_ws_bytes
 * (inference: one private workspace per run; training: the layout of cine_unet2d_train_ws_bytes, which cine_unet2d_backward reads).
 * Capturable: inside a stream capture the side streams join the capture through the fork event.  The reference runs the two networks
 * one after the other on torch's current stream.
 * nside > 0 (inference) also tells the dispatcher that this slice runs beside nothing but its own branches: the 128-channel bottleneck
 * then launches two 64-row workgroups per plane instead of one of 128 rows (same pixel tiling, same statistics records, same bits; faster
 * for a slice alone, slower with many slices in flight -- which run one stream each and pass nside == 0). */
size_t cine_unet2d_branch_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools, int nsets, int nbranch, int train);
int cine_unet2d_forward_branches(const float* x, float* y, const void* const* weights, int nsets,
                                 int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                                 void* ws, size_t ws_bytes, void* stream, void* const* side, int nside, int train, const float* drop);
/* Dropout (unet.py:22,40,159-168: Dropout2d behind every LeakyReLU of a ConvBlock, active in training mode with drop_prob > 0).  `drop` of
 * cine_unet2d_forward_branches (train != 0 only; NULL = none) holds the multiplier d of every (3x3 conv, sample, channel) plane -- 0 for a
 * dropped channel, 1 / (1 - p) for a kept one -- drawn by the CALLER (the binding uses torch's generator, like nn.Dropout2d): one (n, ch) block
 * per 3x3 conv in launch order (down path and bottleneck level d: two blocks of chans << d channels; up path from the coarsest level: two blocks
 * each; the transpose-conv blocks have no dropout), cine_unet2d_drop_floats(n, chans, pools) floats in all.  d >= 0 commutes with LeakyReLU, so
 * the forward folds it into the plane's InstanceNorm statistics records and no activation is ever rewritten; cine_unet2d_backward_drop is
 * cine_unet2d_backward with the same array. */
size_t cine_unet2d_drop_floats(int n, int chans, int pools);
/* The same for the 3-D U-Net (Dropout3d: whole (sample, channel) volumes): cine_unet3d_forward_train / cine_unet3d_backward with `drop` in the layout above. */
int cine_unet3d_forward_train_drop(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                                   int in_ch, int out_ch, int chans, int pools, float slope, void* ws, size_t ws_bytes, const float* drop, void* stream);
int cine_unet3d_backward_drop(const float* x, const float* gy, const void* const* wdgrad, void* const* grads,
                              int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                              const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, const float* drop, void* stream);
int cine_unet2d_backward_drop(const float* x, const float* gy, const void* const* wdgrad, void* const* grads, int nsets,
                              int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                              const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, const float* drop, void* stream);
/* The reference's small tensor helpers as device kernels, for user code written against its utils (the fused path never calls
 * them).  utils/math.py:20-44: complex_mul with broadcasting -- `shape` = the broadcast result's dimensions WITHOUT the trailing
 * complex pair (at most 6), `xstride` / `ystride` the operands' strides in complex elements, 0 on broadcast dimensions; out is dense.
 * complex_conj, complex_abs_sq: n complex elements.  utils/coil_combine.py: out (outer, inner) = sqrt(sum_k v(x[outer][k][inner])),
 * v = x^2 or, with is_complex, re^2 + im^2 of (outer, k, inner, 2).  utils/fftc.py:141-213 roll along one dimension:
 * out[o][(j + shift) mod n][i] = x[o][j][i] (out of place; fftshift = shift n / 2, ifftshift = (n + 1) / 2).
 * utils/padding.py:22-47: zero padding of `planes` (h, w) planes to (hp, wp) with the data at (top, left). */
int cine_complex_mul(const float* x, const float* y, float* out, int ndim, const int* shape, const long* xstride, const long* ystride, void* stream);
int cine_complex_conj(const float* x, float* out, long n, void* stream);
int cine_complex_abs_sq(const float* x, float* out, long n, void* stream);
int cine_rss(const float* x, float* out, long outer, int k, long inner, int is_complex, void* stream);
int cine_roll(const float* x, float* out, long outer, int n, long inner, int shift, void* stream);
int cine_pad2d(const float* x, float* out, long planes, int h, int w, int top, int left, int hp, int wp, void* stream);

/* Diagnostics.  3x3 convolutions whose tile spans the plane's width (the x-f / y-f planes of the cascade U-Nets, reference
 * denoisers/unet.py:159-168) and the k2 s2 transpose convs between them (unet.py:212-218) run on lean kernels (csrc/conv_plane.hip)
 * that are BIT-IDENTICAL to the general one, and so do the 3x3 (x3) convolutions of wider planes and volumes in 16-wide column
 * tiles (sensitivity network, CRNN cells, 3-D U-Net); `on` is a mask -- bit 0 the plane-wide 3x3 convolutions, bit 1 the
 * transpose convolutions, bit 2 the wide planes / volumes; a cleared bit routes that kind through the general kernel (the
 * bit-identity tests, A/B timing).  Bit 4 SET routes the plane-wide weight gradients of training (grad_kernels.hip:
 * wgrad_plane_kernel, also bit-identical) through the general weight-gradient kernel.  A setting of the CALLING THREAD (default 7
 * in every thread): it selects kernels for the launches that thread enqueues afterwards and never changes a result. */
int cine_set_conv_plane(int on);

/* A second stream of the CALLING THREAD for the weight-gradient launches of cine_unet2d_backward / cine_unet3d_backward / cine_mwcnn_backward (they
 * depend only on a layer's output gradient, not on the input-gradient chain behind it): the calls fork onto it with events
 * and join before returning, so on return everything is ordered on `stream` again and the results do not depend on timing.
 * NULL (the default) or the same stream as `stream`: every launch stays on `stream`.  Replaces nothing in the reference
 * (torch.autograd runs its backward nodes on one stream); it is how `loss.backward()` (pl_modules/varnet_module.py:97-113) fills
 * the chip. */
int cine_set_side_stream(void* side_stream);
/* Backward pass of the U-Net (the autograd graph of unet.py:73-125): from gy = d loss / d y, the forward's input x and its
 * filled workspace `fwd_ws`.  `wdgrad`: HOST array of device pointers ordered like `weights` of the forward, holding the
 * input-gradient packings (the bias slot is ignored).  `grads`: HOST array in the same order of device pointers to the weight
 * gradients ((cout, cin, 3, 3) / (cin, cout, 2, 2) / (out_ch, chans) / (out_ch)), accumulated into.  nsets == 2: both arrays
 * hold two such lists back to back.  gx (n, in_ch, h, w) may be NULL (the sensitivity network's input needs no gradient).
 * ws: cine_unet2d_backward_ws_bytes() of scratch. */
size_t cine_unet2d_backward_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools);
int cine_unet2d_backward(const float* x, const float* gy, const void* const* wdgrad, void* const* grads, int nsets,
                         int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                         const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, void* stream);

/* The same for the 3-D U-Net (unet.py:73-125 with dims = 3; CineNet's regulariser, cinenet.py:98-116): cine_unet3d_forward_train is
 * cine_unet3d_forward with every raw layer output and its merged statistics record kept in `ws` (cine_unet3d_train_ws_bytes), and
 * cine_unet3d_backward walks it in reverse.  `wdgrad` (order of `weights`, bias slot ignored): cine_pack_conv3d of each 3x3x3 weight with its taps
 * flipped and (cout, cin) transposed; cine_pack_conv1x1 of a transpose conv's weight (cin, cout, 2, 2, 2) read as the (cin, 8 cout) matrix;
 * cine_pack_conv1x1 of the final conv's matrix transposed (chans, out_ch).  `grads`: the parameters' own layouts ((cout, cin, 3, 3, 3) /
 * (cin, cout, 2, 2, 2) / (out_ch, chans) / (out_ch)), accumulated into.  One weight set.  gx (n, in_ch, d, h, w) may be NULL.
 * Weight gradients run on the calling thread's side stream (cine_set_side_stream) when it has one. */
size_t cine_unet3d_train_ws_bytes(int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools);
int cine_unet3d_forward_train(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                              int in_ch, int out_ch, int chans, int pools, float slope, void* ws, size_t ws_bytes, void* stream);
size_t cine_unet3d_backward_ws_bytes(int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools);
int cine_unet3d_backward(const float* x, const float* gy, const void* const* wdgrad, void* const* grads,
                         int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                         const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, void* stream);

/* The same for the wavelet CNN (denoisers/mwcnn.py:135-179): a forward that keeps every feature map, and the backward pass -- InstanceNorm +
 * LeakyReLU backward with the Haar DWT / IWT adjoints (the transforms are orthogonal: each one's adjoint is the other) and the additive
 * skips gathered on load, input gradients on the forward conv kernel, weight gradients with the wavelet / summed sources re-staged.
 * weights / weights2, set_split as cine_mwcnn_forward2 (weights2 NULL: one network); wdgrad*: the cine_pack_conv3x3_dgrad packings in the
 * order of `weights` (bias slot ignored); grads*: the parameters' own layouts, accumulated into. */
size_t cine_mwcnn_train_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters,
                                 const int* n_convs, int first_filters);
int cine_mwcnn_forward_train(const float* x, float* y, const void* const* weights, const void* const* weights2, int set_split,
                             int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                             int n_first_convs, int first_filters, int res, float slope, void* ws, size_t ws_bytes, void* stream);
size_t cine_mwcnn_backward_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters,
                                    const int* n_convs, int first_filters);
int cine_mwcnn_backward(const float* x, const float* gy, const void* const* wdgrad, const void* const* wdgrad2, void* const* grads,
                        void* const* grads2, int set_split, int n, int h, int w, int in_ch, int out_ch, int n_scales,
                        const int* n_filters, const int* n_convs, int first_filters, float slope, const void* fwd_ws, size_t fwd_ws_bytes,
                        void* ws, size_t ws_bytes, float* gx, void* stream);

/* Adjoints of cine_normunet_unpack / cine_normunet_pack (norm_unet.py:59-96).
 *   unpack_bwd: gout (n, h, w, 2), the U-Net output planes_q -> gq (planes, zero on the pad frame) and dstats (n, 2, 2) =
 *               {d/d mean, d/d std} per (sample, re|im) from the un-normalisation x * std + mean.
 *   pack_bwd  : gp (gradient of the U-Net input planes), the planes themselves, stats, dstats -> gz (n, h, w, 2), through
 *               (x - mean) / std with the unbiased std.  stats == NULL: the plain repack (cinenet.py:242-244). */
int cine_normunet_unpack_bwd(const float* gout, const float* planes_q, const float* stats, float* gq, float* dstats,
                             int n, int h, int w, void* stream);
int cine_normunet_pack_bwd(const float* gp, const float* planes_p, const float* stats, const float* dstats, float* gz,
                           int n, int h, int w, void* stream);
/* Adjoints of cine_xfyf_unpack / cine_xfyf_pack (varnet.py:196-241): gout (b, t, 1, h, w, 2) -> gradients of the two U-Nets'
 * output planes (+ dstats, and gmean (b, h, w, 2) = the gradient of the temporal mean image); then from the gradients of the
 * U-Nets' input planes -> gimg (b, t, h, w, 2).  The temporal DFT is unitary: its adjoint is the inverse transform.
 * ws: cine_xfyf_bwd_ws_bytes(). */
size_t cine_xfyf_bwd_ws_bytes(int b, int t, int h, int w);
int cine_xfyf_unpack_bwd(const float* gout, const float* q_xf, const float* q_yf, const float* stats_xf, const float* stats_yf,
                         float* gq_xf, float* gq_yf, float* dstats_xf, float* dstats_yf, float* gmean,
                         int b, int t, int h, int w, int xf, void* ws, size_t ws_bytes, void* stream);
int cine_xfyf_pack_bwd(const float* gp_xf, const float* gp_yf, const float* p_xf, const float* p_yf,
                       const float* stats_xf, const float* stats_yf, const float* dstats_xf, const float* dstats_yf,
                       const float* gmean, float* gimg, int b, int t, int h, int w, int xf,
                       void* ws, size_t ws_bytes, void* stream);

/* Backward pieces of the convolutional-RNN cells (models/recurrent_varnet.py:153-259: sums of plain 3x3 convolutions + bias, ReLU).
 * cine_relu_mask: g *= (y > 0) in place, y = the stored ReLU output.  cine_conv3x3_wgrad: gw (cout, c0 + c1, 3, 3) += the weight
 * gradient of y = conv3x3(cat(x0, x1); W) from g = d loss / d y (x1 NULL / c1 0: one input), gb (cout) += the bias gradient when
 * not NULL; deterministic (partial sums in `ws`, fixed-order reduction).  The input gradient is cine_conv3x3_dgrad, or
 * cine_conv3x3_ex on the cine_pack_conv3x3_dgrad packing when an addend rides along (the time sweep's chain rule). */
int cine_relu_mask(float* g, const float* y, long n, void* stream);
size_t cine_conv3x3_wgrad_ws_bytes(int cout, int cin, int n);
int cine_conv3x3_wgrad(const float* x0, int c0, const float* x1, int c1, const float* g, float* gw, float* gb,
                       int n, int cout, int h, int w, void* ws, size_t ws_bytes, void* stream);

/* Layer-level backward pieces (the 3-D U-Net's backward pass is composed from these, cine_hip/autograd.py: Unet3dFn).
 * cine_conv1x1_wgrad: gw (cout, cin) += weight gradient of a 1x1(x1) convolution, gb (cout) += bias gradient when not NULL.
 * cine_in_lrelu_bwd: d loss / d raw from d loss / d LeakyReLU(InstanceNorm(raw)) (unet.py:159-168) for planes (n, c) of h * w
 * elements with statistics records part (n, c, np, 3).  Volumes pass (d h, w). */
size_t cine_conv1x1_wgrad_ws_bytes(int cout, int cin, int n);
int cine_conv1x1_wgrad(const float* x, int cin, const float* g, float* gw, float* gb, int n, int cout, int h, int w,
                       void* ws, size_t ws_bytes, void* stream);
size_t cine_in_lrelu_bwd_ws_bytes(int n, int c, int h, int w);      /* 0 for small planes; large ones are cut into chunks (ws may be NULL then: one workgroup per plane) */
int cine_in_lrelu_bwd(const float* r, const float* part, int np, const float* g, float* gr, int n, int c, int h, int w,
                      float eps, float slope, void* ws, size_t ws_bytes, void* stream);

/* Adjoints of cine_xpd_unpack / cine_xpd_pack (models/xpdnet.py:424-509): gout (b, t, 1, h, w, 2n) -> the gradients of the two MWCNNs' output
 * planes (2n channels, zero on the pad frames) and gmean (b, h, w, n + 1, 2) (the temporal mean of channels < n is added back, :504-509); then
 * from the gradients of the MWCNNs' input planes (2 (n + 1) channels) -> gbuf (b, t, 1, h, w, 2n) and gextra (b, t, 1, h, w, 2) (the
 * backward-operator image).  XPDNet's temporal transforms (:466, :500) are unitary: each adjoint is the inverse with the same shifts. */
int cine_xpd_unpack_bwd(const float* gout, float* gplanes_xf, float* gplanes_yf, float* gmean,
                        int b, int t, int h, int w, int n_primal, int n_scales, int xf, void* stream);
int cine_xpd_pack_bwd(const float* gplanes_xf, const float* gplanes_yf, const float* gmean, float* gbuf, float* gextra,
                      int b, int t, int h, int w, int n_primal, int n_scales, int xf, void* stream);

/* cine_image_dc is self-adjoint in the image (T = IFFT_h W FFT_h is Hermitian): its image gradient is cine_image_dc of the output
 * gradient (zf = NULL).  With respect to the maps it gives, per frame, part (b, t, c, h, w) = conj(g) T(S_c img) + T(S_c g) conj(img)
 * (varnet.py:181-194, 281-282); add the frames with cine_coil_accum(NULL, part, ...).  Weights as cine_image_dc. */
int cine_image_dc_sens_grad(const float* img, const float* gout, const float* sens, const uint8_t* mask,
                            const float* lambda_dev, float w_sampled, float w_unsampled,
                            float* part, int b, int t, int c, int h, int w, void* stream);
/* gs (b, c, h, w, 2) (+)= sum_t conj(g[b, t]) z[b, t, c]: gradient of sens_reduce's coil sum sum_c conj(S_c) z_c (varnet.py:187-194)
 * with respect to S; z (b, t, c, h, w, 2) are the coil images.  g == NULL: gs (+)= sum_t z.  accumulate == 0 overwrites gs. */
int cine_coil_accum(const float* g, const float* z, float* gs, int b, int t, int c, int h, int w, int accumulate, void* stream);
/* gx of y = x / rss(x, coil) (varnet.py:58-59) from gy and the UN-normalised x (b, c, h, w, 2). */
int cine_rss_normalise_bwd(const float* gy, const float* x, float* gx, int b, int c, int h, int w, void* stream);
/* gx (n, 2) of y = |x| (math.py:48-62). */
int cine_complex_abs_bwd(const float* gy, const float* x, float* gx, long n, void* stream);
/* out = a + sign * f(v) * b with v = softplus(*lambda_dev) (varnet.py:281-282): kind 0 f = v; 1 f = v / (1 + v); 2 f = 1 / (1 + v)^2;
 * 3 f = 1 / (1 + v).  a == NULL: out = sign * f(v) * b. */
int cine_axpby_lam(float* out, const float* a, const float* b, long n, const float* lambda_dev, int kind, float sign, void* stream);

/* SSIMLoss (utils/losses.py:25-58): loss = mean over frames of (1 - mean SSIM over the win x win windows of the valid region), sample
 * covariance, K1 / K2, the data range of every frame = the maximum of that TARGET frame (losses.py:34).  x = reconstruction, y = target,
 * both (t, h, w) float32; *loss_dev one float.  The forward keeps the per-window derivatives in `ws` (cine_ssim_loss_ws_bytes) for
 * cine_ssim_loss_bwd: gx (t, h, w) = d loss / d x scaled by the upstream gradient *gloss_dev.  Window sums in float64. */
size_t cine_ssim_loss_ws_bytes(int t, int h, int w, int win);
int cine_ssim_loss(const float* x, const float* y, int t, int h, int w, int win, double k1, double k2,
                   float* loss_dev, void* ws, size_t ws_bytes, void* stream);
int cine_ssim_loss_bwd(const float* x, const float* y, int t, int h, int w, int win, const float* gloss_dev,
                       const void* ws, size_t ws_bytes, float* gx, void* stream);

/* ------------------------------------------------------------------------------------------
 * measurement aid (no reference counterpart; the reference only wraps time.time() around the
 * model call, traintest_scripts/run_inference.py:53-61)
 * ------------------------------------------------------------------------------------------ */
/* Between cine_profile_begin() and cine_profile_end() every kernel launch made through this
 * library is bracketed by a hipEvent pair on the stream it is launched on.  cine_profile_end()
 * waits for them and returns, per kernel family i < nfam, the summed device time in ms and the
 * launch count.  Not for use during graph capture. */
int cine_profile_begin(void);
/* Diagnostics: process-wide counts of launches that took one of two interchangeable routes, so that a test can prove WHICH ran
 * (`which`: 0 plane-wide weight gradients on the lean kernel, 1 on the general kernel, 2 U-Net passes run as concurrent branches,
 * 3 BCRNN layers run by the C time-sweep entry points).  reset != 0 returns the count and zeroes it; -1 for an unknown counter. */
long cine_diag_counter(int which, int reset);
/* Diagnostics: a one-workgroup kernel that runs for `microseconds` (1 .. 100 000; clock-bounded AND iteration-bounded: it always ends).  The
 * binding times two of them on two streams to learn whether the streams share a hardware queue (GPU_MAX_HW_QUEUES): streams on one queue run
 * one after the other, and a side stream chosen that way would serialise with its main stream. */
int cine_spin(int microseconds, void* stream);
int cine_profile_end(double* ms, long* launches, int nfam);
int cine_profile_families(void);
const char* cine_profile_family_name(int family);

#ifdef __cplusplus
}
#endif
#endif /* CINE_HIP_H */
