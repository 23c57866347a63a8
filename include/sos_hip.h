/* sos_hip.h -- C ABI of libsos_hip.so, the MI355X (gfx950) hot path of the
 * Listening-to-Sound-of-Silence speech-denoising pipeline.
 *
 * The reference is pure Python/PyTorch: it has no FFI or operator registry, so
 * the drop-in seam is its Python module API.  Each entry point
 * below is what the thin Python shims in
 * listening-to-sound-of-silence-for-speech-denoising_amd/ bind through ctypes;
 * the comment on each cites the reference function it replaces
 * (paths relative to /root/reference, M1 = model_1_silent_interval_detection/
 * audioonly_model, M2 = model_2_audio_denoising/audio_denoising_model).
 *
 * Conventions: every pointer is a DEVICE pointer unless said otherwise; no
 * allocation and no synchronisation inside; work is enqueued on `stream`
 * (a hipStream_t); return 0 on success, negative on error (message via
 * sos_last_error()).  No C++ exceptions cross this boundary.
 */
#ifndef SOS_HIP_H
#define SOS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sos_stream_t; /* hipStream_t */

enum { SOS_OK = 0, SOS_EINVAL = -22, SOS_ELAUNCH = -5, SOS_ENOSPC = -28 };

/* activation / padding / dtype codes */
enum { SOS_ACT_NONE = 0, SOS_ACT_RELU = 1, SOS_ACT_PRELU = 2, SOS_ACT_SIGMOID = 3 };
enum { SOS_PAD_ZERO = 0, SOS_PAD_REFLECT = 1 };
enum { SOS_DT_BF16 = 0, SOS_DT_BF16X3 = 1, SOS_DT_F32 = 2 };

int sos_abi_version(void);
/* size in bytes of the descriptor structs as compiled into the library; which = 0: sos_view, 1: sos_conv_desc,
 * 2: sos_wgrad_desc; anything else: -1 */
int sos_struct_size(int which);
const char* sos_last_error(void);
/* The package builds this ABI twice from the same sources: libsos_hip.so computes on bfloat16 storage ("bf16"),
 * libsos_hip_f16.so on IEEE half ("fp16", same MFMA rate, 11 instead of 8 significand bits).  Wherever this header
 * says "bf16" for an activation / packed-weight buffer it means "the library's 16-bit storage type". */
const char* sos_storage_dtype(void);

/* ---- a1  fast_stft: M1/transform.py:188-193 (librosa.stft(data,510,158,400), hann-periodic window centred in
 * n_fft, center=True, reflect pad) fused with real_imag_expand (:10-17) and the caller's transpose to [2,F,T]
 * (M2/dataset.py:255).  The transform is a windowed-DFT GEMM on the matrix cores (csrc/stft_mfma.hip): the constant
 * matrix is packed ONCE on the host into MFMA fragment order, as hi + lo half-precision parts (the product is taken as
 * hi*hi + hi*lo + lo*hi with fp32 accumulation, ~22 significand bits), and uploaded by the caller:
 *   sos_stft_matrix_bytes(): bytes of ONE part; sos_stft_pack_matrix(): fills two HOST buffers of that size.
 * wave f32 [B][wave_stride], out f32 [B][2][n_fft/2+1][T], T = 1 + n_samples/hop.  Supported geometry: (n_fft+2) % 32
 * == 0, win_length % 16 == 0, even hop, ceil(win/hop) <= 3 (the reference's 510/158/400); anything else is EINVAL. */
int64_t sos_stft_matrix_bytes(int n_fft, int hop, int win_length);
int sos_stft_pack_matrix(int n_fft, int hop, int win_length, void* hi /* host */, void* lo /* host */);
int sos_stft_f32(const float* wave, int64_t batch, int64_t n_samples, int64_t wave_stride,
                 const void* mat_hi, const void* mat_lo /* device copies of the packed matrix */, int n_fft, int hop,
                 int win_length, float* out, int64_t n_frames,
                 const int32_t* clip_samples /* optional device [B], ragged batch: clip b has clip_samples[b] <= n_samples
                    samples (reflected at ITS end) and 1 + clip_samples[b]/hop frames; rows keep the pitch n_frames */,
                 sos_stream_t stream);

/* ---- a2  fast_istft: M1/transform.py:196-202 (librosa.istft(S,158,400)): inverse real DFT, synthesis window,
 * overlap-add, division by the window-sum-square (where > tiny), trim n_fft/2 both ends -- the synthesis GEMM
 * (window and irfft weights folded into the packed matrix, same hi/lo scheme), then a fixed-order overlap-add of the
 * <= 3 frames covering a sample with the window-sum-square taken over those same frames.
 * sos_istft_pack_matrix also fills win_sq f32 [win_length] (HOST), the squared window.
 * spec f32 [B][2][F][T]; out f32 [B][out_stride], hop*(T-1) samples written. */
int64_t sos_istft_matrix_bytes(int n_fft, int hop, int win_length);
int sos_istft_pack_matrix(int n_fft, int hop, int win_length, void* hi /* host */, void* lo /* host */, float* win_sq /* host */);
int sos_istft_f32(const float* spec, int64_t batch, int64_t n_frames, const void* mat_hi, const void* mat_lo,
                  const float* win_sq /* device */, int n_fft, int hop, int win_length, float* out, int64_t out_stride,
                  const int32_t* clip_frames /* optional device [B], ragged batch: clip b has clip_frames[b] <= n_frames
                     frames -> hop*(clip_frames[b]-1) samples */,
                  sos_stream_t stream);

/* power_law: M1/transform.py:178-185, out = sign(x) |x|^power -- the optional companding of fast_stft / fast_istft
 * (`power=True`: 0.3 before the STFT, 1/0.3 after the ISTFT). */
int sos_power_law_f32(const float* x, int64_t n, float power, float* out, sos_stream_t stream);

/* ---- a4/a5  batch_fast_icRM_sigmoid: M1/transform.py:156-169 (and the numpy
 * twin fast_icRM_sigmoid :141-153): M = (1/a)(log(c/(1-c+1e-8)+1e-10)+b), rec = M*Y
 * (complex).  Y, crm, rec: f32 [B][2][plane]. */
int sos_crm_apply_f32(const float* Y, const float* crm, float* rec, int64_t batch, int64_t plane,
                      float a, float b, sos_stream_t stream);
/* backward of the above w.r.t. crm (grad flows to the mask, M2/agent.py:186-189). */
int sos_crm_apply_bwd_f32(const float* Y, const float* crm, const float* grad_rec, float* grad_crm,
                          int64_t batch, int64_t plane, float a, sos_stream_t stream);
/* ---- a3  fast_cRM_sigmoid: M1/transform.py:130-138 (generate_cRM :36-54 +
 * cRM_sigmoid_compress :92-94).  clean, mix, out: f32 [B][2][plane]. */
int sos_crm_target_f32(const float* clean, const float* mix, float* out, int64_t batch, int64_t plane,
                       float a, float b, sos_stream_t stream);

/* ---- a13/a15  convert_bitstreammask_to_audiomask: M2/tools.py:340-362 (=
 * M1/tools.py:770-792, M2/predict.py:232-252) + `noise_sig = mixed*mask`
 * (M2/predict.py:317).  bits u8 [B][n_frames] (1 = non-silent); mask f32
 * [B][n_samples] (1 on silent samples); if sig != NULL, masked = sig*mask.
 * Integer index rule evaluated in IEEE double exactly like the Python. */
int sos_bits_to_mask(const uint8_t* bits, int64_t batch, int64_t n_frames, double ratio,
                     int64_t n_samples, float* mask, const float* sig, float* masked,
                     const int32_t* clip_frames, const int32_t* clip_samples /* optional device [B] each, ragged batch:
                        clip b has clip_frames[b] bits and clip_samples[b] samples; row pitches stay n_frames / n_samples */,
                     sos_stream_t stream);

/* ---- a16  add_signals / add_noise_to_audio: M2/tools.py:217-303.  Per clip b: every noise k is rescaled so that
 * energy(signal) / energy(noise) = 10^(snr_db[b]/10) (left alone when signal or noise is silent), mixed = signal +
 * sum_k noise_k, then mixed, signal and the noises are divided by max|mixed| / norm (norm == 0: no normalisation).
 * signal f32 [B][n], noises f32 [B][n_noises][n] (n_noises <= 8), snr_db f32 [B] (device); outputs same shapes. */
int sos_add_signals_f32(const float* signal, const float* noises, const float* snr_db, int64_t batch, int n_noises,
                        int64_t n, float norm, float* mixed, float* signal_out, float* noises_out, sos_stream_t stream);

/* ---- a14  detector post-processing: M1/predict.py:117-119,
 * bit = sigmoid(logit) >= 0.5 (1 = non-silent).  conf (optional) = sigmoid. */
int sos_threshold_bits(const float* logits, int64_t n, float threshold, uint8_t* bits, float* conf,
                       sos_stream_t stream);
/* Two-pass detector of the 'mixed' pipeline (the threshold of M1/predict.py:117-119 only depends on the SIGN of a logit):
 * logits f32 [B][n] of a 1x-cost 16-bit detector pass; clip b is MARKED (mark[b] = 1) when one of its first n_valid[b] (NULL: n)
 * frames has |logit| < band_rel * max(1, max_t |logit[b][t]|, max_t scale[b][t]), i.e. lies inside the 16-bit pass's error band
 * around the threshold.  scale (ABI 10; optional f32 [B][n]): the magnitude that error is relative to -- the last layer's
 * sum_i |w_i| |a_i| + |bias| per frame (>= |logit|), so that logits that are small by cancellation do not shrink the band.  tabs_in / tabs_out int32 [ntab][B]: per-clip width tables of the ragged geometry (sos_conv_desc.wl_tab / wo_tab,
 * the BiLSTM's lengths, ...): tabs_out = tabs_in for marked clips, 0 for the others, so that a second detector pass in the
 * parity precision computes ONLY the marked clips (tiles of a zero-width clip exit at once) without a host round trip.
 * count (optional int32 [2]) += {marked clips, clips}. */
int sos_logit_band_mark(const float* logits, const float* scale, int64_t batch, int64_t n, const int32_t* n_valid,
                        float band_rel, const int32_t* tabs_in, int32_t* tabs_out, int ntab, int32_t* mark, int32_t* count,
                        sos_stream_t stream);

/* ---- layout glue at the module boundary: f32 NCHW [B][C][H][W] -> bf16 NHWC
 * [B][H][W][cs] (channels >= C zero filled; SOS_DT_BF16X3 writes hi|hi|lo thirds
 * of width cs/3). */
int sos_pack_nchw_to_nhwc(const float* in, int64_t B, int C, int64_t H, int64_t W, void* out, int cs,
                          int dtype, const float* mul /* optional device scalar multiplied in (loss scale) */,
                          sos_stream_t stream);
/* The same boundary pack with the kw HORIZONTAL TAPS of the first conv layer folded into the channel axis: stored channel
 * t*C + c of pixel (h, w) = in[b][c][h][w + t - pad_left] (t < kw; outside the clip: zero for SOS_PAD_ZERO, mirrored for
 * SOS_PAD_REFLECT; channels >= kw*C zero).  The first block of every encoder (Conv2d(2, nf, (1,7)), M1/networks.py:120-128,
 * M2/networks.py:72-80) and of the U-Net (DownConvBlock(2, 64, 5, 1), M2/networks.py:158,165) then runs as a kh x 1 layer over
 * kw*C real channels with the weight w'[o][t*C + c][a][0] = w[o][c][a][t] -- 14 or 10 of the 16 stored channels real instead
 * of 2, at no extra byte.  clip_w: optional device [B] (ragged batch: clip b has clip_w[b] <= W columns, its border is taken
 * at ITS end, columns past it are written as zeros).  cs (per third) % 8 == 0 and >= kw*C. */
int sos_pack_nchw_wtaps(const float* in, int64_t B, int C, int64_t H, int64_t W, int kw, int pad_left, int pad_mode,
                        const int32_t* clip_w, void* out, int cs, int dtype, const float* mul, sos_stream_t stream);

/* ---- a6,a8,a9 (+ a7/a10/a11 heads): Conv2d (zero or reflect pad, stride,
 * dilation) / ConvTranspose2d phases / Linear, fused with folded BatchNorm or bias
 * and ReLU / PReLU / Sigmoid -- M1/networks.py:28-51, M2/networks.py:28-51,97-149.
 * Implicit GEMM on bf16 MFMA, fp32 accumulate. */
typedef struct sos_conv_desc {
    /* input activation, bf16 NHWC: element (b,h,w,c) at ((b*H + h)*W + w)*in_cs + c */
    const void* in;
    int32_t B, H, W;        /* physical input dims                                     */
    int32_t in_cs;          /* channels stored per pixel (multiple of 8)               */
    int32_t cin_off, cin;   /* contracted channel range [cin_off, cin_off+cin), cin%16==0 */
    int32_t in_nseg;        /* number of such ranges (1, or 3 for the hi|hi|lo thirds)   */
    int32_t in_seg_stride;  /* channel distance between consecutive ranges               */
    const int32_t* w_gather;/* optional [Wl]: physical column of logical column (nearest resize) */
    int32_t Wl;             /* logical input width (== W when w_gather == NULL)        */
    /* weights, bf16 [kh*kw][cout_pad][in_nseg*cin], cout_pad % 32 == 0                */
    const void* wgt;
    int32_t kh, kw, cout, cout_pad;
    int32_t stride, dil_h, dil_w, pad_top, pad_left, pad_mode;
    /* output: pixel (b,ho,wo), channel co at out + b*sb + ho*sh + wo*sw + (c_off+co)*sc  */
    int32_t Ho, Wo;
    void* out;
    int32_t out_dtype;      /* SOS_DT_*                                                */
    int64_t out_sb, out_sh, out_sw, out_sc;
    int32_t out_c_off;
    int32_t cout_store;     /* channels [cout, cout_store) are written as zero          */
    int64_t out_third;      /* SOS_DT_BF16X3: element distance between hi|hi|lo thirds  */
    /* epilogue: y = act(acc*scale[co] + shift[co]); scale == shift == NULL: y = act(acc)  */
    const float* scale;     /* [cout_pad]                                               */
    const float* shift;     /* [cout_pad]                                               */
    int32_t act;            /* SOS_ACT_*                                                */
    const float* act_param; /* device scalar (PReLU slope) or NULL                      */
    int32_t accumulate;     /* 1: add to the existing bf16 output (dense NHWC outputs only): gradient
                               fan-in of skip connections in the backward pass            */
    /* optional fused BatchNorm statistics of the (bf16-rounded) output, dense bf16 NHWC outputs only:
     * stats[0][c][tile] = sum, stats[1][c][tile] = sum of squares over the tile's valid pixels, c < stats_c,
     * tile < sos_conv2d_tile_count(desc) (2 * stats_c * tiles floats; a channel's tile sums are contiguous for the
     * finalize); feed to sos_bn_finalize(partial = stats, nblk = tile count). */
    float* stats;
    int32_t stats_c;
    /* optional RAGGED batch (BASELINE configs[3]: clips of different lengths in one launch; the reference runs each
     * file at its own length, M2/predict.py:405-447): the buffers keep the geometry above with W / Wl / Wo = the
     * batch's maxima, image b uses only its first wl_tab[b] logical input columns (zero / reflect borders are taken
     * at ITS end) and produces wo_tab[b] output columns; the rest of its output columns are left untouched.  Device
     * int32 [B] each, both or neither.  w_gather_stride: distance in ints between the images' w_gather tables
     * (0: one table for all). */
    const int32_t* wl_tab;
    const int32_t* wo_tab;
    int32_t w_gather_stride;
    /* optional TEMPORAL taps (Conv3d with temporal stride 1 and padding (kt-1)/2, M1/networks.py:54-77: the audio-visual
     * variant's video branch): the B images are clips of t_frames consecutive frames and the contraction also runs over
     * t_taps neighbouring frames, image b reading frames b + dt - t_pad (dt < t_taps) of ITS clip, zeros outside the
     * clip.  The contraction index is (range s < in_nseg, dt, channel): weights [kh*kw][cout_pad][in_nseg*t_taps*cin].
     * t_taps <= 1: plain 2-D convolution.  Replaces a materialised time-stacked input (sos_time_stack). */
    int32_t t_frames, t_taps, t_pad;
    /* optional REFLECTION-PAD FOLD of the output (backward of ReflectionPad2d in DownConvBlock, M2/networks.py:105: the data
     * gradient of a reflect-padded conv is a full correlation onto the PADDED domain [H+2p][W+2p] whose border cells add to the
     * interior cell they mirror).  fold_pad = p > 0: output pixel (ho, wo) is cell (hp, wp) = (ho*fold_sy + fold_oy,
     * wo*fold_sx + fold_ox) of the padded domain (sy = sx = 1, oy = ox = 0 for a stride-1 layer; 2 and the phase for the four
     * phase convolutions of a stride-2 layer).  Interior cells (p <= hp < p + fold_H, p <= wp < p + fold_W) are written -- with
     * `accumulate`: added -- straight to `out` as pixel (hp - p, wp - p) of the dense [B][fold_H][fold_W] NHWC tensor that
     * out_sb / out_sw / out_c_off / out_third describe (out_sh is ignored); border cells are stored to the padded scratch tensor
     * fold_pad_out, dense [B][fold_H+2p][fold_W+2p][fold_row] with the channels from 0 (thirds fold_third apart), which
     * sos_reflect_fold_border then adds onto `out`.  Only the border of the scratch tensor is ever touched.  16-bit NHWC outputs
     * (out_sc == 1) without fused statistics only. */
    void* fold_pad_out;
    int32_t fold_pad, fold_H, fold_W, fold_sy, fold_oy, fold_sx, fold_ox, fold_row;
    int64_t fold_third;
    /* optional FUSED INPUT BatchNorm + ReLU (round 5; Conv2dBlock, M2/networks.py:28-51, training mode): `in` holds the RAW conv
     * output of the producing block and in_scale / in_shift (f32 [cin], channel c of the contraction = input channel cin_off + c)
     * its batch-statistics scale / shift: every value read becomes max(x * in_scale[c] + in_shift[c], 0), rounded to the storage
     * type exactly as sos_bn_act_apply would have stored it, while the patch is staged; zero padding stays zero.  The separate
     * apply pass and the activated tensor are then not needed by this consumer.  One 16-bit channel segment, no temporal taps;
     * built for the 96-channel context layers' tilings (EINVAL otherwise: the caller falls back to the materialised tensor). */
    const float* in_scale;
    const float* in_shift;
} sos_conv_desc;

int sos_conv2d_fwd(const sos_conv_desc* desc /* host pointer */, sos_stream_t stream);
/* number of pixel tiles (workgroup columns) the NEXT sos_conv2d_fwd of this shape will use: the tuned tiling if
 * sos_conv2d_tune has run for the shape, the default otherwise */
int64_t sos_conv2d_tile_count(const sos_conv_desc* desc);

/* One-time autotune for the SHAPE of `desc` (not its pointers): runs the `max_candidates` most
 * promising tilings `iters` times each, timed with HIP events on `stream` (this call
 * SYNCHRONISES -- keep it out of graph capture and timed regions), and caches the winner for
 * every later sos_conv2d_fwd of the same shape.  *best_ms (optional) = winning time, or -1 if
 * the shape was already tuned. */
int sos_conv2d_tune(const sos_conv_desc* desc, int max_candidates, int iters, float* best_ms,
                    sos_stream_t stream);
/* Save / load the tuned-tiling cache (host text file).  load returns the number of entries read. */
int sos_conv2d_tune_save(const char* path);
int sos_conv2d_tune_load(const char* path);

/* ---- a7/a11 recurrent part of nn.LSTM(bidirectional=True), gate order i,f,g,o
 * (M1/networks.py:95,143-148; M2/networks.py:64,88).  The input projection
 * x@W_ih^T + b_ih + b_hh is a sos_conv2d_fwd (1x1) producing xproj.
 * W_hh (f32 [2][4H][H], the layout torch stores: weight_hh_l0, weight_hh_l0_reverse) is packed once per
 * weight version into MFMA fragment order: sos_lstm_pack_bytes(H, 0 / 1) = bytes of ONE forward /
 * backward array; the lo arrays (both or neither) hold the bf16 remainders for the three-pass
 * hi*hi + hi*lo + lo*hi product of the bf16x3 precision mode.  H % 4 == 0, H <= 256.
 * xproj f32 [B][T][2][4H] (dir 0 fwd, 1 reverse) in GATE-INTERLEAVED order: channel dir*4H + 4*j + q for
 * hidden unit j and gate q in i,f,g,o (permute the rows of W_ih and of the bias; torch's order is q*H + j);
 * out_bf16 bf16 [B][T][out_cs] (+ thirds when dtype == SOS_DT_BF16X3). */
int64_t sos_lstm_pack_bytes(int H, int backward);
int sos_lstm_pack_whh(const float* whh, int H, void* fwd_hi, void* fwd_lo, void* bwd_hi, void* bwd_lo,
                      sos_stream_t stream);
int sos_lstm_bidir_fwd(const float* xproj, const void* wpk_hi, const void* wpk_lo /* optional */, int64_t B, int64_t T,
                       int H, void* out_bf16, int out_cs, int out_dtype, int64_t out_third,
                       float* save_gates /* optional f32, ceil16(B)*T*2*4H: post-activation i,f,g,o */,
                       float* save_c /* optional f32, ceil16(B)*T*2*H: cell state */,
                       const int32_t* lengths /* optional device [B]: clip b has lengths[b] <= T frames (ragged batch:
                          its reverse pass starts at frame lengths[b]-1; rows past it are neither read nor written) */,
                       sos_stream_t stream);
/* save_gates / save_c are consumed only by sos_lstm_bidir_bwd; their layout is the forward kernel's MFMA lane order
 * [16-clip group][t][dir][4-unit tile][clip][unit][i,f,g,o] (one coalesced store per tile and step). */

/* ---------------------------------------------------------------- training-mode kernels
 * A `sos_view` describes a channel slice of a bf16 NHWC activation: element (pix, c) lives at
 * ptr[pix*row + c_off + c]; with x3 != 0 the value is hi + lo, hi at that address (and again at
 * +third), lo at +2*third.  C <= 256. */
typedef struct sos_view {
    void* ptr;
    int64_t npix;
    int32_t row, c_off, C, x3;
    int64_t third;
} sos_view;

/* ---- BatchNorm2d(train) of Conv2dBlock/ConvBlock/DownConvBlock/UpConvBlock (M1/networks.py:38-39,
 * M2/networks.py:38-39,107-108,137-138): batch statistics over all pixels of the raw conv output.
 * Stage 1 writes deterministic per-workgroup partial sums (no atomics): partial f32 [2][C][nblk] (a channel's sums contiguous),
 * nblk = sos_bn_stats_blocks(npix). */
int sos_bn_stats_blocks(int64_t npix);
int sos_bn_stats(const sos_view* x, float* partial, sos_stream_t stream);
/* Stage 2: mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale (for the apply
 * pass), save_mean / save_invstd (for backward); running stats updated with momentum and the
 * UNBIASED variance, num_batches_tracked += 1 (torch semantics).  gamma/beta may be NULL (=1/0). */
int sos_bn_finalize(const float* partial, int nblk, int C, int64_t count, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* scale, float* shift, float* save_mean, float* save_invstd, sos_stream_t stream);
/* y = act(x*scale + shift).  Dense form: y is a sos_view with the same npix.  Feature form
 * (feat_H > 0): the reference's view(B,-1,T).permute(2,0,1) (+ optional nearest-resize column
 * gather): pixel (b,h,w'), channel c is written to y.ptr[(b*feat_Wo + w')*y.row + (y.c_off + c)*feat_H + h]
 * reading source column w = gather ? gather[w'] : w' of a [B][feat_H][feat_W] pixel grid. */
int sos_bn_act_apply(const sos_view* x, const float* scale, const float* shift, int act, const float* slope,
                     const sos_view* y, int feat_H, int feat_W, int feat_Wo, const int32_t* gather,
                     sos_stream_t stream);

/* ---- weight gradient of Conv2d / ConvTranspose2d / Linear (autograd of F.conv2d etc. behind
 * loss.backward(), M1/agent.py:106-111, M2/agent.py:101-106):
 *   dw[m][n][a][b] (+)= scale * sum_p G[p][m] * X[p*stride + (a,b)*dil - pad][n]
 * G ("tile side"): bf16 NHWC [B][Hg][Wg][g_cs], channels [g_off, g_off+M); X ("patch side"): bf16
 * NHWC [B][Hx][Wx][x_cs], channels [x_off, x_off+N).  Conv: G = grad of the raw conv output,
 * X = layer input.  ConvTranspose2d(k3,s2,p1): G = layer input, X = output grad, stride 2 ->
 * result is in the (Cin, Cout, kh, kw) layout.  partial: fp32 workspace of
 * sos_wgrad_workspace_bytes(); dw fp32 [M][N][kh][kw]. */
typedef struct sos_wgrad_desc {
    const void* g;
    int32_t B, Hg, Wg, g_cs, g_off;
    const void* x;
    int32_t Hx, Wx, x_cs, x_off;
    int32_t M, N, kh, kw, stride, dil_h, dil_w, pad_top, pad_left, pad_mode;
    int32_t ksplit;         /* pixel-range split (parallelism), <= 0: automatic (one workgroup per CU); partial
                             * sums are reduced deterministically.  Size `partial` with sos_wgrad_workspace_bytes */
    float* partial;
    float* dw;
    int32_t accumulate;     /* 0: dw = result, 1: dw += result */
    float scale;
    const float* scale_dev; /* optional device scalar multiplied into `scale` (1 / loss scale of the fp16 mode) */
    /* optional TEMPORAL taps (see sos_conv_desc): N = t_taps * t_cin columns, column n = dt * t_cin + c pairs image b of
     * G with channel c of frame b + dt - t_pad of X (same clip of t_frames frames, zeros outside); t_cin % 128 == 0.
     * t_taps <= 1: off. */
    int32_t t_frames, t_taps, t_pad, t_cin;
} sos_wgrad_desc;
int64_t sos_wgrad_workspace_bytes(const sos_wgrad_desc* desc);
int sos_conv2d_wgrad(const sos_wgrad_desc* desc, sos_stream_t stream);
/* ABI 8: the two halves of sos_conv2d_wgrad as separate entry points -- _partial launches the MFMA kernel (per-split partial sums
 * into desc->partial), _reduce the deterministic reduction of those sums into desc->dw (scale, scale_dev, accumulate).  Both derive
 * the same launch plan from the descriptor, so the reduce may run on ANOTHER stream behind an event (the gradient is needed only by
 * the optimizer; engine.wgrad(defer=True) keeps a ring of workspaces and joins before the gradients are used). */
int sos_conv2d_wgrad_partial(const sos_wgrad_desc* desc, sos_stream_t stream);
int sos_conv2d_wgrad_reduce(const sos_wgrad_desc* desc, sos_stream_t stream);
/* ABI 7: measured launch plans of sos_conv2d_wgrad (workgroup channel tile, pixel tile, pixel order, workgroups per CU), the
 * counterpart of sos_conv2d_tune: times the candidate plans for the SHAPE of `desc` (`iters` launches each, the fastest few
 * again over 8x as many; HIP events on `stream`, SYNCHRONISES, overwrites desc->dw / desc->partial: pass accumulate = 0 and
 * scratch outputs) and caches the winner for every later sos_conv2d_wgrad of that shape.  *best_ms (optional) = the winning time,
 * -1 if the shape already had a plan (or takes the GEMM path).  save / load: host text file; load returns the entries accepted
 * (0: no file).  The package ships wgrad_table_gfx950.txt so that every process and rank runs the same plans. */
int sos_wgrad_tune(const sos_wgrad_desc* desc, int iters, float* best_ms, sos_stream_t stream);
int sos_wgrad_tune_save(const char* path);
int sos_wgrad_tune_load(const char* path);

/* ---- backward of the BatchNorm(+activation) / bias(+activation) tail of a conv block (autograd of
 * nn.BatchNorm2d + ReLU/PReLU in train mode).  dy: grad of the block output; x: raw conv output;
 * scale/shift/mean/invstd: from sos_bn_finalize (mean == invstd == NULL: no BatchNorm);
 * partial: f32 scratch of 3 * C * sos_bn_stats_blocks(npix) floats; coef: f32 [4][C] scratch.  Writes dgamma, dbeta
 * (or the bias gradient), dslope[0] (PReLU) and dx = grad of the raw conv output. */
int sos_bn_bwd(const sos_view* dy, const sos_view* x, const float* scale, const float* shift, const float* mean,
               const float* invstd, const float* gamma, int act, const float* slope, float* partial, float* coef,
               float* dgamma, float* dbeta, float* dslope, const sos_view* dx,
               const float* out_scale /* optional device scalar: dgamma, dbeta, dslope are multiplied by it */,
               sos_stream_t stream);
/* dz = dy * act'(y) from the stored OUTPUT y (Linear+ReLU / Linear+Sigmoid heads). */
int sos_act_bwd_from_y(const sos_view* dy, const sos_view* y, int act, const sos_view* dz, sos_stream_t stream);
/* f32 strided gradient (x sigmoid'(y) if act == SOS_ACT_SIGMOID) -> bf16 rows: element (o,t,c) read at
 * g[o*so + t*st + c*sc], written to row o*inner+t, channel c of `out`. */
int sos_pack_grad_f32(const float* g, const float* y, int act, int64_t outer, int64_t inner, int C, int64_t so,
                      int64_t st, int64_t sc, const sos_view* out, const float* mul /* optional device scalar */,
                      sos_stream_t stream);
/* gradient of the LSTM feature matrix back to NHWC (inverse of the feature form of sos_bn_act_apply):
 * out[b][h][w][c] = sum_{i in [lo[w],hi[w])} feat[b][i][(feat.c_off+c)*H + h]  (lo/hi NULL: i == w). */
int sos_feat_to_nhwc(const sos_view* feat, int B, int H, int W, int Wo, const int32_t* lo, const int32_t* hi,
                     const sos_view* out, sos_stream_t stream);
/* ---- BPTT of the recurrent part of nn.LSTM (autograd of M1/networks.py:148, M2/networks.py:88).
 * dh_out: bf16 grad of the LSTM output [B][T][dh_cs] (dh_cs % 4 == 0); gates/csave from the forward;
 * wtk_hi / wtk_lo: the backward arrays of sos_lstm_pack_whh; dgates f32 [B][T][2][4H] (gate
 * pre-activation grads, gate-interleaved like xproj; dW_ih, dW_hh, bias and input grads are GEMMs over it). */
int sos_lstm_bidir_bwd(const void* dh_out, int dh_cs, int dh_dtype, int64_t dh_third, const float* gates,
                       const float* csave, const void* wtk_hi, const void* wtk_lo /* optional */, int64_t B, int64_t T,
                       int H, float* dgates, sos_stream_t stream);
/* ---- losses (M2/agent.py:174,188-189; M1/agent.py:187,202) and optimizer (M1/agent.py:177). */
int sos_mse_loss(const float* a, const float* b, int64_t n, float upstream, float* loss, float* grad, float* partial,
                 sos_stream_t stream);
int sos_bce_logits_loss(const float* x, const float* y, int64_t n, float upstream, float* loss, float* grad,
                        float* partial, sos_stream_t stream);
/* ---- loss scale of the fp16 storage mode (the reference trains in fp32, M2/agent.py:101-106; half precision needs
 * the gradients of the activations moved into its exponent range).  The scale is a power of two chosen ON THE DEVICE
 * from the gradient entering the hand-written backward, so no host synchronisation: sos_amax_f32 folds max|g| into
 * *amax (caller zeroes it; several tensors may be folded), sos_loss_scale writes scale2 = {S, 1/S} with
 * S = 2^floor(log2(target / amax)) clamped to [2^-40, 2^40] (S = 1 when amax is 0 or not finite).  The backward multiplies S in
 * where f32 gradients become 16-bit (sos_pack_grad_f32 / sos_pack_nchw_to_nhwc `mul`) and 1/S where parameter
 * gradients leave (sos_wgrad_desc.scale_dev, sos_bn_bwd out_scale, sos_scale_f32): exact, the pass is linear. */
int sos_amax_f32(const float* x, int64_t n, float* amax, sos_stream_t stream);
int sos_loss_scale(const float* amax, float target, float* scale2, const float* guard /* optional: SOS_GUARD state, the
                   target is multiplied by its back-off factor */, sos_stream_t stream);
int sos_scale_f32(float* x, int64_t n, const float* s /* device scalar */, sos_stream_t stream);
/* ---- overflow guard of a training step (no counterpart in the reference, which trains in fp32: M2/agent.py:101-106;
 * the analogue of torch.cuda.amp.GradScaler's found_inf / skipped step, decided ON THE DEVICE: no host sync).
 * guard: device f32 [SOS_GUARD_FLOATS], zero-initialised by the caller once per model:
 *   [0] found  : 1 if the gradients checked last were not all finite, else 0 -- the optimizer kernels skip their whole
 *                update (parameters, exp_avg, exp_avg_sq untouched) while it is 1
 *   [1] backoff: factor in (0, 1] applied to sos_loss_scale's target (0 reads as 1): halved by every overflow (floor 2^-20),
 *                doubled again after SOS_GUARD_GROWTH consecutive finite steps
 *   [2] finite steps since the last change of [1];  [3] steps skipped so far;  [4] scratch (raw flag bits)
 * sos_grad_guard scans g[0..n) (a flat gradient buffer of the model, after the all-reduce) into the scratch flag and, when
 * `finalize` is non-zero, updates the state from it (several buffers: finalize with the last one). */
#define SOS_GUARD_FLOATS 5
#define SOS_GUARD_GROWTH 200
int sos_grad_guard(const float* g, int64_t n, float* guard, int finalize, sos_stream_t stream);
/* skip: optional device pointer to the guard state: a non-zero [0] turns the launch into a no-op, and once [3] (steps skipped so
 * far) is non-zero the bias corrections are taken at step - [3], the number of updates actually APPLIED (`step` counts attempts;
 * torch.cuda.amp.GradScaler semantics: a skipped update does not advance the optimizer's step) */
int sos_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, float grad_scale, const float* skip, sos_stream_t stream);

/* multi-tensor form: one launch for all parameter tensors of a model.  tensors: DEVICE array; chunks: DEVICE int32
 * [n_chunks][2] = (tensor index, chunk index), SOS_ADAM_CHUNK elements per chunk, covering every tensor. */
#define SOS_ADAM_CHUNK 16384
typedef struct sos_adam_tensor { float* p; const float* g; float* m; float* v; int64_t n; } sos_adam_tensor;
int sos_adam_multi_step(const sos_adam_tensor* tensors, int n_tensors, const int32_t* chunks, int64_t n_chunks, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                        const float* skip /* optional, see sos_adam_step */, sos_stream_t stream);

/* Re-pack every 16-bit weight tensor of a model from its fp32 parameters in ONE launch (after an optimizer step).
 * entries: DEVICE array; entry e fills out[0..n) (16-bit storage type) from idx[0..n) (device int64): idx > 0 = absolute
 * address of the fp32 source element, idx < 0 = the low part of the value at -idx (hi|lo|hi thirds of the three-pass
 * mode), 0 = zero padding.  chunks as for sos_adam_multi_step (SOS_ADAM_CHUNK elements each). */
typedef struct sos_pack_entry { const int64_t* idx; void* out; int64_t n; } sos_pack_entry;
int sos_gather_pack_multi(const sos_pack_entry* entries, int n_entries, const int32_t* chunks, int64_t n_chunks,
                          sos_stream_t stream);

/* backward of ReflectionPad2d(pad) (DownConvBlock, M2/networks.py:105): out (+)= fold of the padded-domain
 * gradient `padded` ([B][H+2pad][W+2pad]) onto the [B][H][W] interior. */
int sos_reflect_fold(const sos_view* padded, int H, int W, int pad, const sos_view* out, int accumulate,
                     sos_stream_t stream);
/* second half of a convolution launched with sos_conv_desc.fold_pad: out += the BORDER cells of `padded` folded onto the
 * interior cells they mirror (the interior cells were written by the convolution itself).  Touches only the pixels of `out`
 * within `pad` of an edge and the border of `padded`. */
int sos_reflect_fold_border(const sos_view* padded, int H, int W, int pad, const sos_view* out, sos_stream_t stream);
/* copy a channel slice between NHWC grids [B][Hs][Ws] -> [B][Hd][Wd]: overlap copied, rest of dst zeroed
 * (the crop that stands in for F.interpolate(out, skip.size()) at M2/networks.py:199-203, and its backward). */
int sos_copy_crop(const sos_view* src, int Hs, int Ws, const sos_view* dst, int Hd, int Wd, sos_stream_t stream);

/* ---- 8f-2  the wave front door: `librosa.load(path, sr=14000)` at M1/dataset.py:226, M2/predict.py:288,297,303
 * (soundfile decode -> to_mono -> resampy 'kaiser_best' -> fix_length) and the samples handed to
 * `librosa.output.write_wav` (M2/predict.py:515-528).  The RIFF container is parsed on the host; these take the
 * decoded, still interleaved samples. */
enum { SOS_PCM_S16 = 0, SOS_PCM_S32 = 1, SOS_PCM_F32 = 2, SOS_PCM_U8 = 3 };
/* pcm: interleaved [n_frames][channels] device samples -> out f32 [n_frames] = mean over channels of
 * sample / 2^(bits-1) (u8: (v-128)/128; f32: as is). */
int sos_pcm_to_mono_f32(const void* pcm, int format, int channels, int64_t n_frames, float* out, sos_stream_t stream);
/* Band-limited resampling by `ratio` = sr_new / sr_orig (resampy 0.2.2 `resample_f`): win = the filter's half
 * window (nwin floats, num_table samples per zero crossing), already multiplied by min(1, ratio).  Writes
 * out[0 .. n_out): the first floor(n_in * ratio) entries are resampled, the rest zero (librosa pads to
 * ceil(n_in * ratio)).  (nwin + 1) * 4 bytes must fit the 160 KB LDS. */
int sos_resample_f32(const float* x, int64_t n_in, double ratio, const float* win, int nwin, int num_table,
                     float* out, int64_t n_out, sos_stream_t stream);
/* Host-only helper (no GPU work): the piecewise-linear description of resampy's running f64 time register
 * `time_register += 1/ratio` that sos_resample_f32 hands its kernel: time(t) = fma(t - k0[i], d[i], s0[i]) for
 * the last i with k0[i] <= t.  Returns the number of segments (<= capacity) or a negative error. */
int sos_resample_time_segments(double ratio, int64_t n_out, int64_t* k0, double* s0, double* d, int capacity);

/* ---- 8f-1  audio-visual variant, video branch: Conv3dBlock M1/networks.py:54-77, make_video_branch :110-118
 * (configuration :87-89), fusion :135-142.  A Conv3d with temporal stride 1 runs on sos_conv2d_fwd over a
 * time-stacked input (temporal taps on the contraction axis).
 * sos_time_stack: in bf16 [B*T][HW][nseg*in_cs] (C real channels per third) -> out [B*T][HW][nseg*out_cs] with
 *   out channel dt*C + c = frame t + dt - (kt-1)/2, zeros outside the clip and in the padding (out_cs >= kt*C).
 * sos_spatial_mean: torch.mean(f_v, dim=(-2,-1)) of N images into out[n*out_row + third*out_third + out_c_off + c]. */
int sos_time_stack(const void* in, int64_t B, int T, int64_t HW, int C, int in_cs, int nseg, int kt, void* out,
                   int out_cs, sos_stream_t stream);
int sos_spatial_mean(const void* in, int64_t N, int64_t HW, int C, int in_cs, int nseg, void* out, int64_t out_row,
                     int out_third, int out_c_off, sos_stream_t stream);
/* their transposes (autograd of the variant): gradient of the stacked tensor folded back onto the frames, and the
 * gradient of the spatial mean broadcast over the HW pixels (hi|hi|lo thirds re-split in bf16x3 mode). */
int sos_time_unstack(const void* d_stacked, int64_t B, int T, int64_t HW, int C, int st_cs, int nseg, int kt,
                     void* d_in, int in_cs, sos_stream_t stream);
int sos_spatial_mean_bwd(const void* dfeat, int64_t N, int64_t HW, int C, int64_t f_row, int f_third, int f_c_off,
                         int nseg, void* dy, int cs, sos_stream_t stream);

/* ---- 8f-4  objective measures of M2/metrics.py, the per-sample / per-frame work (finalisation = host code in
 * sos_amd/metrics.py): f32 device signals, frames start = f*skip of `winlength` samples under `window` (f64[winlength]).
 * sos_metric_totals      : out3 = {sum ref^2, sum (ref-deg)^2, max |ref|}            (overall SNR :97, silence threshold :192)
 * sos_metric_frame_energy: out[f] = {sum (w c)^2, sum (w c - w p)^2}                (metrics_ssnr* :119-127)
 * sos_metric_compact     : keeps, in order, the samples with |clean| >= thr         (metrics_ssnr_exclude_silence :189-199)
 * sos_metric_llr         : out[f] = log(a_p R_c a_p' / a_c R_c a_c'), order-P LPC   (llr :561-623, lpcoeff :626-681)
 * sos_metric_wss         : out[f] = weighted spectral slope distance, crit_filter f32 [25][n_fft/2]   (wss :404-558)
 * sos_metric_l1          : mean |lerp(output)(linspace(0, n_out-1, n_t)) - target|  (metrics_L1 :40-45) */
int sos_metric_totals(const float* ref, const float* deg, int64_t n, double* out3, sos_stream_t stream);
int sos_metric_frame_energy(const float* ref, const float* deg, int64_t n, int winlength, int skip, int64_t num_frames,
                            const double* window, double* out, sos_stream_t stream);
int sos_metric_compact(const float* clean, const float* proc, int64_t n, float thr, float* out_clean, float* out_proc,
                       int64_t* count, sos_stream_t stream);
int sos_metric_llr(const float* ref, const float* deg, int64_t n, int winlength, int skip, int64_t num_frames,
                   const double* window, int P, float* out, sos_stream_t stream);
int sos_metric_wss(const float* ref, const float* deg, int64_t n, int winlength, int skip, int64_t num_frames,
                   const double* window, int n_fft, const float* crit_filter, double eps, float* out, sos_stream_t stream);
int sos_metric_l1(const float* output, int64_t n_out, const float* target, int64_t n_t, double* result, sos_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SOS_HIP_H */
