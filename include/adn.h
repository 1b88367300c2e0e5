/*
 * adn.h — C ABI of libadn.so, the MI355X (gfx950) implementation of the AudioDenoiser hot path.
 *
 * The reference (jimonld2000/AudioDenoiser) is pure Python and has no FFI of its own; its boundary for this
 * path is the Python API of code/model.py, code/data_loader.py and the two STFT helpers.  Each entry point
 * below replaces the arithmetic behind one of those Python call sites (cited per function); the Python
 * mirror in audiodenoiser_amd/{model,stft,data_loader}.py binds them with ctypes (see INTEGRATION.md for the
 * stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C types only: device pointers are raw `float*`, the stream is a `hipStream_t` passed as `void*`
 *     (NULL = the null stream).  No torch / C++ types cross this boundary.
 *   - every function returns an int status (ADN_OK = 0); on failure adn_last_error() returns a thread-local
 *     message.  Nothing throws or aborts across the ABI.
 *   - work is enqueued on the caller's stream.  The only calls that block or allocate: adn_unet_create / adn_unet_destroy
 *     (one-time weight upload / free), adn_prepare, and the FIRST call per (device, n_fft) of an STFT-family entry point
 *     (adn_stft_mag, adn_stft_mag_fit, adn_stft_complex, adn_istft, adn_griffin_lim) or per device of adn_perceptual_loss, which
 *     builds a few KB of constant tables (window, twiddles, mel filters) with a blocking upload -- unless adn_prepare did so
 *     before.  Such a cold call on a stream that is being captured enqueues nothing and returns ADN_ERR_INVALID (never a HIP
 *     error): call adn_prepare(device, n_fft) before capturing.  adn_unet_forward never blocks or allocates.
 *   - ownership: the caller owns every buffer it passes (x, y, audio, out, workspace); a handle owns only
 *     its packed (BatchNorm-folded, re-laid-out) weights.
 *   - a handle is bound to one device and is not re-entrant: one forward at a time per handle.  Entry points that
 *     take a handle run on the handle's device and restore the caller's current device before returning; the
 *     handle-free entry points (STFT, loader, losses, Griffin-Lim) run on the caller's current device.
 *   - alignment: workspaces 16 bytes; complex spectrograms (float pairs) 8 bytes; everything else 4 bytes
 *     (ADN_ERR_INVALID otherwise).  Any hipMalloc / PyTorch allocation satisfies this.
 *   - there is NO CPU implementation in this library.
 */
#ifndef ADN_H
#define ADN_H

#include <stddef.h>

/* The 28 functions below are the ONLY symbols libadn.so exports: the library is built with -fvisibility=hidden and linked with
 * a version script (audiodenoiser_amd/csrc/libadn.map: `adn_*` global, everything else local). */
#if defined(__GNUC__)
#define ADN_API __attribute__((visibility("default")))
#else
#define ADN_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ADN_OK 0
#define ADN_ERR_INVALID 1      /* bad argument (shape, null pointer, unsupported size) */
#define ADN_ERR_HIP 2          /* a HIP runtime call failed; message carries hipGetErrorString */
#define ADN_ERR_WORKSPACE 3    /* workspace too small */
#define ADN_ERR_NO_DEVICE 4    /* no gfx950 device visible */

#define ADN_N_WEIGHT_TENSORS 118   /* state_dict float tensors (136 entries minus 18 num_batches_tracked) */
#define ADN_N_TAPS 10              /* down1..down4, bottleneck, up1..up4, out */
#define ADN_N_LAUNCHES 23          /* timing slots per forward: first conv, 17 MFMA 3x3 convs, 4 convT, 1x1 out; a slot
                                      whose layer runs fused into its neighbour (1x1 out: only the tail that adds the
                                      partial planes remains; first conv on the F(2x2,3x3) path) stays ~0 */

typedef struct adn_unet adn_unet;

/* Library / diagnostics --------------------------------------------------------------------------------- */
ADN_API int adn_version(void);
ADN_API const char *adn_last_error(void);
ADN_API int adn_device_count(int *count);
/* Builds the constant tables the handle-free entry points need on `device`: the window / twiddle tables of `n_fft` (a power
 * of two in [64, 4096]; 0 = none) and the mel filterbank of adn_perceptual_loss.  Synchronous (blocking upload), idempotent,
 * thread-safe.  After it, every adn_stft_* / adn_istft / adn_griffin_lim call with that n_fft and adn_perceptual_loss only
 * enqueue on the caller's stream, so they can be captured into a HIP graph (e.g. adn_stft_mag_fit + adn_unet_forward, the
 * wav -> network path of the reference's test.py:94-113). */
ADN_API int adn_prepare(int device, int n_fft);

/* U-Net forward: replaces UNet.forward (reference code/model.py:70-94) and everything it calls —
 * DoubleConvLayer (model.py:7-20), DownSampleLayer (model.py:23-32), UpSampleLayer (model.py:35-50). ----- */

/* Build a handle from the 118 fp32 HOST tensors of `UNet(1,1).state_dict()` in state_dict order with the
 * num_batches_tracked entries skipped (per conv+BN pair: conv.weight, conv.bias, bn.weight, bn.bias,
 * bn.running_mean, bn.running_var; per UpSampleLayer first up.weight, up.bias; finally out.weight, out.bias).
 * This is the weight format of the reference checkpoint (train.py:142, test.py:65).  BatchNorm (eval mode,
 * eps 1e-5) is folded into the preceding convolution here.  Synchronous. */
ADN_API int adn_unet_create(adn_unet **handle, int device, const float *const *host_tensors, int n_tensors);
/* Same with an explicit arithmetic type.  ADN_DTYPE_F32 (= adn_unet_create): exact-fp32 matrix cores.
 * ADN_DTYPE_F16 (BASELINE configs[4]): activations and weights stored in fp16 inside the library, fp16 MFMA
 * with fp32 accumulation; x and y stay fp32 at the boundary; outputs within 1e-2 of the fp32 path. */
#define ADN_DTYPE_F32 0
#define ADN_DTYPE_F16 1
ADN_API int adn_unet_create_ex(adn_unet **handle, int device, const float *const *host_tensors, int n_tensors, int dtype);
/* UNet(in_channels, num_classes) as the reference declares it (code/model.py:54,56,68; its own callers use (1, 1), test.py:63):
 * the same 118 tensors with downconv1.conv.double_conv.0.weight (64, in_channels, 3, 3), out.weight (num_classes, 64, 1, 1) and
 * out.bias (num_classes).  x is then (N, in_channels, F, T) and y (N, num_classes, F, T), both NCHW fp32.  1 <= in_channels <= 64,
 * 1 <= num_classes <= 64.  With more than one input plane / class the first / last
 * convolution run as their own launches (the fused forms are for one plane / one class). */
ADN_API int adn_unet_create_general(adn_unet **handle, int device, const float *const *host_tensors, int n_tensors, int dtype,
                            int in_channels, int num_classes);
ADN_API int adn_unet_channels(const adn_unet *handle, int *in_channels, int *num_classes);
/* Kernel choice by the launch's grid.  Default (on = 0): fp32 3x3 layers run F(4x4,3x3) where its 32x32-pixel tiles fill the chip,
 * F(2x2,3x3) otherwise; layers whose grid is still too small (one clip ... a dozen, at the deep levels) have their K loop cut over
 * up to 8 workgroups + a reduce launch -- fp32 3x3 layers (either Winograd form), fp32 transposed convolutions, fp16 3x3 layers.  Fastest at every batch
 * size, but the same clip computed alone and inside a large batch then differs in the last bits (fp32: both within 1e-4 of the
 * reference; fp16: within 5e-3 of each other, both within 1e-2 of the reference).
 * on = 1: one kernel per layer chosen by the layer's geometry alone -- a clip's result is bit-identical whatever batch it is
 * computed in (evaluation / regression runs that compare across batch sizes), fp32 and fp16 handles alike.  May be changed
 * between forwards.  Initial value: 0, or the environment's ADN_BATCH_INVARIANT when the handle is created. */
ADN_API int adn_unet_set_batch_invariant(adn_unet *handle, int on);
ADN_API int adn_unet_destroy(adn_unet *handle);

/* Bytes of device scratch adn_unet_forward needs for an (N,1,F,T) batch (half as much for an fp16 handle;
 * handle may be NULL = fp32). */
ADN_API int adn_unet_workspace_bytes(const adn_unet *handle, int N, int F, int T, size_t *bytes);

/* y(N,K,F,T) = UNet(x(N,C,F,T)) (C = K = 1 unless the handle came from adn_unet_create_general), eval-mode semantics
 * (BatchNorm uses running statistics), fp32.
 * x, y, workspace: device memory on the handle's device.  F,T >= 16 (four 2x poolings). */
/* Shape limits: N >= 1, F >= 16, T >= 16, F*T < 2^27 (ADN_ERR_INVALID otherwise) -- e.g. 513 bins x 261 000 frames; the workspace
 * (adn_unet_workspace_bytes: ~1.1 KB per pixel in fp32, half in fp16) is the practical bound. */
ADN_API int adn_unet_forward(adn_unet *handle, const float *x, float *y, int N, int F, int T,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Same, additionally exporting block outputs as NCHW fp32 device tensors for parity tests: taps[i] may be
 * NULL (skipped) or a buffer of the block's size, i = 0..3 skip tensors down1..4 (DownSampleLayer's first
 * return value), 4 bottleneck, 5..8 up1..4, 9 out. */
ADN_API int adn_unet_forward_taps(adn_unet *handle, const float *x, float *y, int N, int F, int T,
                          void *workspace, size_t workspace_bytes, float *const *taps, void *stream);

/* Measurement hook (bench.py's roofline object): with timing enabled every kernel launch of adn_unet_forward
 * is bracketed by hipEventRecord on the caller's stream (events are created here, never in the forward).
 * max_forwards = 0 disables.  adn_unet_get_timing synchronises on the events of forward number `index`
 * (0-based since the last adn_unet_set_timing) and returns the ADN_N_LAUNCHES kernel durations in
 * milliseconds, in launch order: conv_first; conv3x3+pool (down1); [conv3x3, conv3x3+pool] x3 (down2..4);
 * conv3x3 x2 (bottleneck); [convT, conv3x3(cat), conv3x3] x4 (up1..4); conv1x1 out. */
ADN_API int adn_unet_set_timing(adn_unet *handle, int max_forwards);
ADN_API int adn_unet_get_timing(adn_unet *handle, int index, float *ms);

/* STFT magnitude: replaces audio_to_magnitude_spectrogram (code/create_train_dataset.py:162-174,
 * center=0) and audio_to_spectrogram (code/create_test_dataset.py:35-41, center=1), i.e.
 * librosa.stft(y, n_fft, hop_length, center, window="hann", pad_mode="constant") + librosa.magphase. ------- */
ADN_API int adn_stft_n_frames(long length, int n_fft, int hop, int center, long *n_frames);
/* audio (n_clips, length) fp32 -> out (n_clips, n_fft/2+1, n_frames) fp32, frame index fastest.
 * n_fft: power of two in [64, 4096]; hop >= 1. */
ADN_API int adn_stft_mag(const float *audio, int n_clips, long length, int n_fft, int hop, int center,
                 float *out, void *stream);

/* STFT magnitude fused with the loader rule that follows it on the wav -> network path: out (n_clips, H, W) =
 * fp32(fp16(|STFT|)) cropped / zero padded at the bottom and right to (H, W), i.e. adn_quantize_pad(adn_stft_mag(...))
 * (code/data_loader.py:41-42,54-72 applied to code/create_test_dataset.py:35-41) in one kernel: only the min(n_frames, W)
 * frames inside the window are computed, nothing is written outside it, rows are W floats apart.  Bit-identical to the
 * two-step form. */
ADN_API int adn_stft_mag_fit(const float *audio, int n_clips, long length, int n_fft, int hop, int center,
                     float *out, int H, int W, void *stream);

/* Loader arithmetic: replaces SpectrogramDataset.__getitem__/_pad_or_truncate (code/data_loader.py:41-42,
 * 54-72) for a batch already on the device: out(n,H,W) = fp32(fp16(in(n,h,w))) cropped / zero padded at the
 * bottom and right. */
ADN_API int adn_quantize_pad(const float *in, int n, int h, int w, float *out, int H, int W, void *stream);

/* Per-clip mean absolute error, the payload of the multi-GPU all-gather (F.l1_loss per clip, cf.
 * code/loss.py:86):  out[i] = mean_j |a[i,j] - b[i,j]|. */
ADN_API int adn_per_clip_l1(const float *a, const float *b, int n_clips, long elems_per_clip, float *out, void *stream);

/* Per-clip CombinedPerceptualLoss: replaces, clip by clip, CombinedPerceptualLoss.forward and the two losses it
 * calls (code/loss.py:6-95; caller code/test.py:118-122).  pred, target: (n_clips,1,F,T) fp32 device tensors;
 * out: (n_clips,4) = {total, stft, mel, l1}.  The reference's batch values are the means over clips (equal clip
 * sizes).  Constants are the reference's: scales (63,16),(32,8),(16,4); mel: sr 8000, n_fft 63, hop 16, 64 mels;
 * weights 0.4/0.4/0.2.  Needs T >= 32 (the mel term's reflect padding; the reference raises below that too) and
 * adn_perceptual_loss_workspace_bytes of device scratch.  Up to 6784 frames a clip's frequency-mean series and mel spectra are held
 * in the LDS of one CU; longer clips (no limit in the reference) keep the series in the workspace and walk the mel frames in blocks. */
ADN_API int adn_perceptual_loss_workspace_bytes(int n_clips, int F, int T, size_t *bytes);
ADN_API int adn_perceptual_loss(const float *pred, const float *target, int n_clips, int F, int T, void *workspace,
                        size_t workspace_bytes, float *out, void *stream);

/* ---- inverse STFT and Griffin-Lim ----------------------------------------------------------------------------
 * Replaces griffin_lim_reconstruction (/root/reference/code/test.py:29-48): librosa.istft + librosa.stft iterated
 * from a random-phase start (librosa 0.10 defaults: n_fft = 2*(n_bins-1), periodic Hann, center=True, zero pad,
 * istft length = hop*(n_frames-1), window sum-of-squares normalisation).  As in the reference the target magnitude is
 * NOT re-imposed inside the loop.  `rnd` holds what np.random.rand(F, T) returned (uniform [0,1), as float32).
 * magnitude, rnd: device (n_clips, n_bins, n_frames) fp32 [the reference's (F, T) layout]; audio_out: device
 * (n_clips, hop*(n_frames-1)) fp32.  Complex spectrograms of the building blocks below are FRAME-major:
 * (n_clips, n_frames, n_bins, 2) fp32. */
ADN_API int adn_istft_length(int n_frames, int hop, long *length);
ADN_API int adn_griffin_lim_workspace_bytes(int n_clips, int n_bins, int n_frames, size_t *bytes);
ADN_API int adn_griffin_lim(const float *magnitude, const float *rnd, int n_clips, int n_bins, int n_frames, int n_fft, int hop,
                    int iterations, void *workspace, size_t workspace_bytes, float *audio_out, void *stream);
/* librosa.stft(audio, n_fft, hop) complex result, centred: n_frames = 1 + length/hop (test.py:41-43). */
ADN_API int adn_stft_complex(const float *audio, int n_clips, long length, int n_fft, int hop, float *spec_out, void *stream);
/* librosa.istft(spec, hop_length=hop) (test.py:40,48); workspace = n_clips*n_frames*n_fft floats. */
ADN_API int adn_istft_workspace_bytes(int n_clips, int n_frames, int n_fft, size_t *bytes);
ADN_API int adn_istft(const float *spec, int n_clips, int n_frames, int n_fft, int hop, void *workspace, size_t workspace_bytes,
              float *audio_out, void *stream);

/* ---- environment switches -------------------------------------------------------------------------------------------------
 * Read ONCE, when a U-Net handle is created (never per call).  None is needed in production: the defaults are the measured best
 * and every family below is covered by the parity tests (tests/test_gpu_variants.py::MODES).  They select between kernel
 * families that all compute the reference's forward within the stated tolerance:
 *
 *   ADN_CONV_ALGO=direct     fp32 3x3 layers on the direct implicit-GEMM kernels (conv_mfma) instead of Winograd
 *   ADN_WINO_TILE=2 | 4      fp32: F(2x2,3x3) for every 3x3 layer | F(4x4,3x3) for every plain / pooled 3x3 layer whatever its size
 *   ADN_WINO_SPLITK=1        fp32: split-K wherever the F(2x2,3x3) grid cannot fill the chip (default: only by the small-grid rule)
 *   ADN_BATCH_INVARIANT=1    initial value of adn_unet_set_batch_invariant (above)
 *   ADN_AUTO_GRID=n, ADN_AUTO_GRID64=n   thresholds of the small-grid rule in F(4x4,3x3) workgroups (192 / 512; tools/small_grid_probe.py)
 *   ADN_CONVT_SPLIT=0        fp32 transposed convolutions on the exact-fp32 MFMA instead of the three-term bf16 split
 *   ADN_F16_CONV=32          fp16 3x3 layers on conv_dma<_Float16> (32x32x16 MFMA) instead of conv16_f16 (16x16x32)
 *   ADN_F16_FIRST=0          fp16: Conv2d(1 -> 64) as its own launch instead of fused into down1's second convolution
 *   ADN_F16_CONVT=dma        fp16 transposed convolutions on conv_dma<_Float16> (32x32x16 MFMA, LDS-staged stores) instead of convt16_f16
 *
 * The library reads no other environment variable; its sources contain no timing-experiment code. */

#ifdef __cplusplus
}
#endif
#endif /* ADN_H */
