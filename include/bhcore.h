/*
 * bhcore.h — C-ABI of the MI355X-native volumetric reconstruction core.
 *
 * This is the drop-in boundary for biahub's per-position hot path.  The
 * reference is pure Python, so what binds here is a ctypes stub; the
 * entry points are one flat, re-entrant call per operator of the reference's L1 layer:
 *
 *   bh_deskew_shape        <- biahub/deskew.py:213-274  get_deskewed_data_shape
 *   bh_deskew              <- biahub/deskew.py:456-542  fast_deskew_zyx  (+ :99-154)
 *   bh_flat_field, bh_median_z
 *                          <- biahub/flat_field.py:101-120 flat_field_zyx, :56-99 _median_tiled (np.median, axis 0)
 *   bh_bin_reduce, bh_bin_finish
 *                          <- biahub/process_data.py:29-105 binning_czyx
 *   bh_valid_mask, bh_bits_and, bh_bits_unpack
 *                          <- biahub/estimate_crop.py:58-94 estimate_crop_one_position (masks, counts, their AND)
 *   bh_blosc_unfilter, bh_blosc_filter
 *                          <- (upstream of the reference) the chunk compressor of the OME-Zarr stores iohub 0.3.11 reads and
 *                             writes for biahub/deskew.py:608-640,738-749: numcodecs 0.15.1 Blosc (uv.lock:3160-3161) =
 *                             c-blosc 1.21 shuffle.c / bitshuffle-generic.c; the byte permutation half of it
 *   bh_overhang_fill       <- biahub/deskew.py:339-368  _fill_overhang_torch
 *   bh_transfer_function   <- biahub/deconvolve.py:30-43 compute_tranfser_function
 *   bh_tikhonov            <- biahub/deconvolve.py:46-66 deconvolve (waveorder Tikhonov)
 *   bh_inverse_filter      <- biahub/apply_inverse_transfer_function.py:158-170 (the per-position job: waveorder's
 *                             apply_inverse_transfer_function_single_position -> models.phase_thick_3d /
 *                             isotropic_fluorescent_thick_3d .apply_inverse_transfer_function, Tikhonov)
 *   bh_phase_transfer_function_3d, bh_fluorescence_transfer_function_3d
 *                          <- biahub/compute_transfer_function.py:16-38, reconstruct.py:59-64 (waveorder's
 *                             compute_transfer_function_cli -> models.*.calculate_transfer_function)
 *   bh_richardson_lucy     <- (north-star extension; no reference function)
 *   bh_phase_cross_corr    <- biahub/estimate_stabilization.py:199-256 phase_cross_corr
 *   bh_image_stats, bh_smooth_shrink, bh_mattes_mi
 *                          <- biahub/registration/ants.py:55-122 estimate (the data-parallel pieces of the
 *                             ants.registration call at :104-109: pyramid, Mattes MI value + derivative)
 *   bh_block_peaks         <- biahub/characterize_psf.py:562-711 detect_peaks (its pooling part, :622-650)
 *   bh_patch_peaks, bh_average_patches
 *                          <- biahub/estimate_psf.py:84-112 (extract_beads + normalised mean),
 *                             vendor/napari_psf_analysis/psf_analysis/extract/BeadExtractor.py:36-78
 *   bh_affine              <- biahub/register.py:202-281 apply_affine_transform,
 *                             biahub/stabilize.py:32-90 apply_stabilization_transform,
 *                             biahub/core/transform.py:374-396 Transform._apply_scipy
 *   bh_crop_flip           <- biahub/utils/array_ops.py:9-59 copy_n_paste[_czyx],
 *                             biahub/flip.py:22-32 flip_cli body
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer (hipMalloc'd, or a torch CUDA tensor's
 *     data_ptr()).  bh_malloc/bh_free/bh_memcpy_* are exported so a host without torch
 *     can stage buffers itself.
 *   - Volumes are C-contiguous (Z, Y, X), X fastest, exactly like the numpy arrays the
 *     reference hands to its operators.
 *   - All work is enqueued on the context's stream (hipStream_t passed as void*; NULL =
 *     the default stream) and is asynchronous unless stated; bh_ctx_synchronize waits.
 *   - Every function returns BH_OK (0) or an error code; the message for the calling
 *     thread's last error is returned by bh_last_error().  BH_ERR_INVALID corresponds
 *     to the reference's ValueError.
 *   - A context owns cached hipFFT plans and a grow-only device workspace; it is not
 *     thread-safe (use one per thread / per process, as the reference uses one process
 *     per worker: biahub/deskew.py:38-40).
 */
#ifndef BHCORE_H
#define BHCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_ABI_VERSION 1

/* status codes */
#define BH_OK 0
#define BH_ERR_INVALID 1     /* bad argument / geometry (reference: ValueError)   */
#define BH_ERR_HIP 2         /* HIP runtime or hipFFT failure                     */
#define BH_ERR_NOMEM 3       /* device allocation failed                          */
#define BH_ERR_UNSUPPORTED 4 /* valid request this build does not implement       */

/* element types of input volumes */
#define BH_DT_U8 0
#define BH_DT_U16 1
#define BH_DT_F32 2
#define BH_DT_I16 3

/* overhang fill modes (biahub/deskew.py:538-540) */
#define BH_FILL_NONE 0
#define BH_FILL_CONSTANT 1
#define BH_FILL_MEAN 2

/* affine interpolation (register.py:207 "linear" | "nearestneighbor") */
#define BH_INTERP_NEAREST 0
#define BH_INTERP_LINEAR 1
#define BH_INTERP_CUBIC 3 /* cubic B-spline with SciPy's prefilter (scipy order=3); boundary BH_BOUNDARY_SCIPY_CONSTANT only */

/* affine boundary rule */
#define BH_BOUNDARY_ITK 0            /* inside iff -0.5 <= c < N-0.5, neighbours clamped (ANTs/ITK) */
#define BH_BOUNDARY_SCIPY_CONSTANT 1 /* inside iff 0 <= c <= N-1 (scipy mode="constant")            */
#define BH_BOUNDARY_ZEROS 2          /* out-of-range neighbours read cval (grid-constant)           */

typedef struct bh_ctx bh_ctx;

/* ---- library / context ------------------------------------------------------------ */
int bh_abi_version(void);
const char* bh_last_error(void);
int bh_device_count(int* count);
/* Device memory laid out like the library's own workspace (blocks of 2 GiB and more: 2-MiB physical chunks mapped in a
 * shuffled order through the HIP virtual-memory API; smaller ones: hipMalloc).  The signatures are the ones
 * torch.cuda.memory.CUDAPluggableAllocator binds (biahub_amd/device.py: volume_pool); any host may call them directly. */
int bh_alloc_layout(int* chunk_kib, int* shuffled, uint64_t* live_blocks, uint64_t* live_bytes);
/* Address ranges of released virtual-memory blocks are retained, not returned (a range reserved again right after its release
 * was seen to deliver the old mapping's pages): how many ranges and bytes of address space that is.  Past
 * BH_ALLOC_VMM_VA_CAP_GB (default 16 TiB) further gigabyte blocks come from hipMalloc. */
int bh_alloc_retained(uint64_t* ranges, uint64_t* bytes); /* diagnostics; any pointer may be NULL */
void* bh_torch_alloc(size_t size, int device, void* hip_stream);
void bh_torch_free(void* ptr, size_t size, int device, void* hip_stream);

int bh_ctx_create(int device, void* hip_stream, bh_ctx** out);
int bh_ctx_destroy(bh_ctx* ctx);
int bh_ctx_set_stream(bh_ctx* ctx, void* hip_stream);
int bh_ctx_synchronize(bh_ctx* ctx);
int bh_ctx_release_workspace(bh_ctx* ctx); /* frees cached plans + scratch             */
int bh_ctx_workspace_bytes(bh_ctx* ctx, uint64_t* bytes);
/* Every hipFFT plan runs a round trip on pseudo-random data when it is created (rocFFT 1.0.36 can return 3-D real plans
 * that compute garbage, depending on the plans created before); a 3-D plan that fails is rebuilt as batched 1-D + strided
 * 2-D transforms, which are checked too (BH_ERR_HIP if those fail as well).  count = plans rebuilt so far in this context. */
int bh_ctx_fft_plans_replaced(bh_ctx* ctx, int* count);

/* device-memory helpers for hosts that do not bring their own allocator */
int bh_malloc(void** dptr, uint64_t bytes);
int bh_free(void* dptr);
int bh_memcpy_h2d(bh_ctx* ctx, void* dst, const void* src, uint64_t bytes); /* synchronous */
int bh_memcpy_d2h(bh_ctx* ctx, void* dst, const void* src, uint64_t bytes); /* synchronous */

/* ---- flat field (the step before deskew in the mantis pipeline) -------------------------------------------- */
/* pattern[y*X + x] (device, float64) = np.median(in[:, y, x]) exactly: the middle element, or for an even Z the mean of
 * the two middle elements (float32 mean for float32 input, like numpy).  BH_ERR_UNSUPPORTED when Z samples of 4 pixels
 * do not fit the LDS staging (Z > 18432 for 16-bit input, 9216 for float32).  NaN input is not supported. */
int bh_median_z(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, double* pattern);

/* out = in / pattern * mean(pattern), pattern = median along Z (biahub/flat_field.py:101-120), written as float32 like
 * _flat_field_czyx (:144-155): float64 arithmetic for integer input, float32 arithmetic for float32 input.
 * pattern (device, Y*X float64) may be NULL (context scratch is used); mean_out (host) may be NULL (no sync). */
int bh_flat_field(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, float* out,
                  double* pattern, double* mean_out);

/* ---- binning (biahub/process_data.py:29-105 binning_czyx) ---------------------------------------------------------- */
/* out (device float32, (Z/fz, Y/fy, X/fx)) = window sums (mean == 0) or sums / count (mean != 0) of a volume whose shape is
 * divisible by the factors (BH_ERR_INVALID otherwise, like numpy's reshape); minmax (host) = min and max of out. Sync. */
int bh_bin_reduce(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, const int factor[3], int mean,
                  float* out, float minmax[2]);
/* out[i] = cast(apply ? ((v[i] - sub) * mul) / div : v[i]) in float32 in that operation order, truncating like
 * ndarray.astype; out_dtype is one of the BH_DT_* codes. */
int bh_bin_finish(bh_ctx* ctx, const float* v, int64_t n, int apply, float sub, float mul, float div, int out_dtype,
                  void* out);

/* ---- validity masks (crop estimation, biahub/estimate_crop.py:58-94) -------------------- */
/* bits: ceil(n / 64) * 2 words on the device; voxel i -> bit i % 32 of word i / 32, set when the voxel is neither 0 nor
 * NaN; count = number of set bits (synchronises the stream). */
int bh_valid_mask(bh_ctx* ctx, const void* vol, int dtype, int64_t n, uint32_t* bits, uint64_t* count);
/* acc &= src, word-wise */
int bh_bits_and(bh_ctx* ctx, uint32_t* acc, const uint32_t* src, int64_t nwords);
/* out[i] = bit i of bits (one byte per voxel) */
int bh_bits_unpack(bh_ctx* ctx, const uint32_t* bits, int64_t n, uint8_t* out);

/* ---- chunk codec: the byte permutations of the Blosc-1 container ---------------------- */
#define BH_BLOSC_NOSHUFFLE 0
#define BH_BLOSC_SHUFFLE 1
#define BH_BLOSC_BITSHUFFLE 2
/* src, dst: nbytes device bytes, distinct.  The buffer is ceil(nbytes / blocksize) blocks (the last one shorter), each
 * permuted on its own exactly as c-blosc 1.x does for elements of `typesize` bytes: byte shuffle (mode 1; a
 * size % typesize tail stays in place), bit shuffle (mode 2; only blocks holding a multiple of 8 elements, the others are
 * stored unpermuted), or none (mode 0: copy).  bh_blosc_unfilter undoes what bh_blosc_filter (or a Blosc writer) did. */
int bh_blosc_unfilter(bh_ctx* ctx, const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize,
                      int mode);
int bh_blosc_filter(bh_ctx* ctx, const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize,
                    int mode);
/* ---- chunk codec: LZ4 blocks inside the Blosc-1 container, on the device ---------------- */
/* Blosc frames of `nframes` chunks: src = nframes * cbytes bytes ALREADY permuted block by block (bh_blosc_filter with this
 * blocksize), chunk after chunk.  Every block is LZ4-compressed by one wavefront (stored raw when it does not shrink), and the
 * frames — 16-byte header, block start table, per block an int32 size and its payload; blocks never split (flag 0x10), inner
 * codec lz4 — are packed into `out` at 16-byte aligned offsets: foff (device, nframes + 1 entries) receives the offset of every
 * frame and, last, the total.  `out` must hold bh_blosc_lz4_bound() bytes.  c-blosc >= 1.15 / numcodecs read the frames.  Does
 * not synchronise. */
uint64_t bh_blosc_lz4_bound(uint32_t nframes, uint32_t cbytes, uint32_t blocksize);
int bh_blosc_lz4_compress(bh_ctx* ctx, const void* src, uint32_t nframes, uint32_t cbytes, uint32_t blocksize, uint32_t typesize,
                          int shuffle_mode, void* out, uint64_t* foff);
/* LZ4 streams back to bytes on the device: stream i is csize[i] bytes at src + soff[i] and decodes to dlen[i] bytes at
 * dst + doff[i] (device arrays of nstreams entries; csize == dlen marks a stored stream).  The caller reads the Blosc frame's
 * block table on the host — a block is one stream, or `typesize` streams when its writer split it (c-blosc 1.21 splits lz4
 * blocks, this library's writer does not) — and the result is still permuted: bh_blosc_unfilter finishes.  Synchronises;
 * BH_ERR_INVALID on a corrupt stream. */
int bh_lz4_decompress_streams(bh_ctx* ctx, const void* src, const uint64_t* soff, const uint32_t* csize, const uint64_t* doff,
                              const uint32_t* dlen, uint32_t nstreams, void* dst);

/* The same two permutations on host memory, on the calling thread (no context, no GPU): for volumes that stay on the
 * host.  Re-entrant; callers parallelise over chunks. */
int bh_host_blosc_unfilter(const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode);
int bh_host_blosc_filter(const void* src, void* dst, uint64_t nbytes, uint32_t blocksize, uint32_t typesize, int mode);

/* ---- deskew ------------------------------------------------------------------------ */
/* Host-only geometry. out_shape = (ceil(Y/n), X, Xp); voxel = (n*sin(t)*px, px, px).
 * BH_ERR_INVALID when keep_overhang==0 and Xp<=0, with the reference's message. */
int bh_deskew_shape(int64_t Z, int64_t Y, int64_t X, double ls_angle_deg, double px_to_scan_ratio,
                    int keep_overhang, int average_n_slices, double pixel_size_um,
                    int64_t out_shape[3], double voxel_size[3]);

/* Fused permute/flip + scan-axis shear interpolation + N-slice mean.
 * in : (Z,Y,X) of in_dtype ;  out : float32 (out_shape) as given by bh_deskew_shape.
 * fill_mode/fill_value apply only when keep_overhang!=0 (reference :538).
 * mean_out (host pointer, may be NULL) receives the fill value used (sync if non-NULL). */
int bh_deskew(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X,
              double ls_angle_deg, double px_to_scan_ratio, int keep_overhang, int average_n_slices,
              int fill_mode, float fill_value, float* out, float* mean_out);

/* bh_deskew with the row sums of the input handed in: row_sums (device, may be NULL) = float64 [Z * Y], row_sums[z * Y + y] =
 * sum over x of in[z, y, x].  With a "mean" fill of a float32 volume the fill value is derived from them before the resampling
 * kernel starts (which then writes whole rows, fill included, in one pass); an operator that has just produced `in` can
 * reduce them on the way (bh_richardson_lucy_apply_rows), otherwise bh_deskew reduces them itself in one read of `in`. */
int bh_deskew_rows(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X,
                   double ls_angle_deg, double px_to_scan_ratio, int keep_overhang, int average_n_slices,
                   int fill_mode, float fill_value, float* out, float* mean_out, const double* row_sums);
/* Diagnostic (synchronises): how the last bh_deskew of this context filled the overhang — 0 mask pipeline (or no fill),
 * 1 one pass, 2 one pass followed by the mask pipeline because the data held exact zeros that geometry does not explain. */
int bh_deskew_fill_path(bh_ctx* ctx, int* path);

/* The same operator on HOST memory, on the calling process's CPU threads (no context, no GPU): what `--cluster debug` with
 * the reference's default `device: cpu` needs (biahub/settings.py:348-383, biahub/deskew.py:762-766; BASELINE config 1).
 * in / out / mean_out are host pointers; nthreads <= 0 uses every hardware thread.  Same float32 sample positions and the
 * same per-output operation order as bh_deskew: bit-identical results (the mean of a "mean" fill to float32 rounding). */
int bh_host_deskew(const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, double ls_angle_deg,
                   double px_to_scan_ratio, int keep_overhang, int average_n_slices, int fill_mode, float fill_value,
                   float* out, float* mean_out, int nthreads);

/* Stand-alone overhang fill on an already deskewed float32 volume, in place. */
int bh_overhang_fill(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode,
                     float fill_value, int dilation_iterations, float* mean_out);
/* Same with the structuring element chosen: connectivity 26 (the production path above, torch max_pool3d) or 6 (SciPy's
 * default cross, what the legacy biahub/deskew.py:277-336 _fill_overhang_with_mean dilates with: `iterations` steps = the
 * L1 ball of that radius). */
int bh_overhang_fill_connectivity(bh_ctx* ctx, float* data, int64_t Z, int64_t Y, int64_t X, int fill_mode,
                                  float fill_value, int dilation_iterations, int connectivity, float* mean_out);

/* ---- deconvolution ----------------------------------------------------------------- */
/* tf_full: float32 (Z,Y,X) = |FFT(zero-padded psf)| / max, full spectrum like the reference. */
int bh_transfer_function(bh_ctx* ctx, const float* psf, int64_t pz, int64_t py, int64_t px, int64_t Z,
                         int64_t Y, int64_t X, float* tf_full);

/* out = real(ifftn(fftn(in) * conj(H) / (|H|^2 + reg))) with real H = tf_full. in/out float32,
 * may alias. */
int bh_tikhonov(bh_ctx* ctx, const float* in, const float* tf_full, int64_t Z, int64_t Y, int64_t X,
                double regularization_strength, float* out);

/* Inverse transfer function (BASELINE config 5).  waveorder 3.0.5 is not in the reference tree: the arithmetic is restated
 * from its published algorithm — parity unpinned.
 *   out = crop_z( Re ifftn( fftn( pad_z( n(in) ) ) * conj(H) / (|H|^2 + reg) ) )
 * in / out: float32 (Z, Y, X) (may alias); tf: H in natural FFT order over (Z + 2 z_padding, Y, X), complex64
 * (tf_is_complex = 1: a phase transfer function) or float32 (an optical transfer function); normalize = 1 applies
 * n(x) = x / mean(x) - 1 first (waveorder inten_normalization_3D, phase) and the z padding planes hold 0.
 * filter_storage: BH_FILTER_F32, or BH_FILTER_BF16 — the staged filter is kept as bfloat16 pairs (half the filter bytes in
 * the Z pass, products in float32); only for shapes the fused FFT engine takes. */
enum { BH_FILTER_F32 = 0, BH_FILTER_BF16 = 1 };
int bh_inverse_filter(bh_ctx* ctx, const float* in, const void* tf, int tf_is_complex, int64_t Z, int64_t Y, int64_t X,
                      int64_t z_padding, double regularization_strength, int normalize, int filter_storage, float* out);
/* The same in two steps, for the time points of a position / the positions of a plate that share one transfer function
 * (what waveorder's per-position job does with the array it reads once from the transfer-function store): `create` stages
 * the inverse filter (a quarter of a one-shot call's time) into device memory the handle owns, `apply` runs one volume
 * through it (in / out float32 (Z, Y, X), may alias), `destroy` releases it.  A handle belongs to the device of the context
 * that created it and may be used from any context of that device. */
typedef struct bh_filter bh_filter;
int bh_inverse_filter_create(bh_ctx* ctx, const void* tf, int tf_is_complex, int64_t Z, int64_t Y, int64_t X, int64_t z_padding,
                             double regularization_strength, int filter_storage, bh_filter** out);
int bh_inverse_filter_apply(bh_ctx* ctx, const bh_filter* filter, const float* in, int normalize, float* out);
int bh_inverse_filter_destroy(bh_filter* filter);
/* Released filter blocks are kept per (device, size) for the next handle of that size (allocating gigabytes costs more than
 * reconstructing a position): this hands them back to the driver. */
int bh_inverse_filter_trim(void);

/* Transfer functions from the optical parameters, complex64 (Z + 2 z_padding, Y, X) in natural FFT order.
 * Phase: weak-object transfer functions of the real and imaginary scattering potential (waveorder
 * phase_thick_3d.calculate_transfer_function at its native sampling; the caller refuses pixel sizes above Nyquist, where
 * waveorder oversamples).  Fluorescence: fftn(|ifft2(pupil * propagation)|^2) / max. */
int bh_phase_transfer_function_3d(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, double yx_pixel_size, double z_pixel_size,
                                  double wavelength_illumination, int64_t z_padding, double index_of_refraction_media,
                                  double numerical_aperture_illumination, double numerical_aperture_detection,
                                  int invert_phase_contrast, void* real_potential_tf, void* imag_potential_tf);
int bh_fluorescence_transfer_function_3d(bh_ctx* ctx, int64_t Z, int64_t Y, int64_t X, double yx_pixel_size,
                                         double z_pixel_size, double wavelength_emission, int64_t z_padding,
                                         double index_of_refraction_media, double numerical_aperture_detection,
                                         void* optical_transfer_function);

/* dst (nz, ny, nx) = the low-frequency corners of the complex64 spectrum src (NZ, NY, NX), both in FFT order (waveorder
 * sampling.nd_fourier_central_cuboid: how a transfer function computed on a finer grid comes back to the data's grid). */
int bh_fourier_central_cuboid(bh_ctx* ctx, const void* src, int64_t NZ, int64_t NY, int64_t NX, void* dst, int64_t nz,
                              int64_t ny, int64_t nx);

/* Host-only: the transform box and back-end bh_richardson_lucy picks for a shape.  BH_RL_ENGINE: the fused FFT engine at
 * the volume's own (power-of-two) shape; BH_RL_ENGINE_PADDED: the engine at a larger box (axes of 2^k, 3 * 2^k or
 * 5 * 2^k) that carries the volume wrap-extended by K - 1; BH_RL_LIBRARY: hipFFT at the shape itself or, for axes with a
 * prime factor above 7, at a 7-smooth pad-and-fold box.  All three compute the circular Richardson-Lucy at size (Z,Y,X). */
#define BH_RL_ENGINE 0
#define BH_RL_ENGINE_PADDED 1
#define BH_RL_LIBRARY 2
int bh_richardson_lucy_plan(int64_t pz, int64_t py, int64_t px, int64_t Z, int64_t Y, int64_t X, int64_t box[3],
                            int* backend);
/* Richardson-Lucy with circular (FFT) boundary; psf is normalised to unit sum on device.
 * e0 = max(in,0); e <- max(e * corr(d / max(conv(e), eps)), 0).  in/out float32, may alias. */
int bh_richardson_lucy(bh_ctx* ctx, const float* in, const float* psf, int64_t pz, int64_t py,
                       int64_t px, int64_t Z, int64_t Y, int64_t X, int iterations, float eps,
                       float* out);
/* The same in two steps, for the volumes of a plate that share one PSF — the shape of the reference's deconvolve: the
 * transfer function is computed once (biahub/deconvolve.py:140-149) and handed to every (position, t, c) unit
 * (:183-191, :52-66).  `create` builds the transfer function at the box bh_richardson_lucy_plan picks (it may synchronise
 * the stream once); `apply` runs `iterations` on one volume (in / out float32 (Z, Y, X), may alias) and only enqueues work
 * on the context's stream: no read-back and no host synchronisation, so uploads and downloads of neighbouring units on
 * other streams overlap it (the one-shot entry above validates its cached transfer function against the PSF's bytes on
 * every call, which is a host stall).  The handle owns the transfer function (the size bh_richardson_lucy_info reports;
 * released blocks are pooled like staged inverse filters, bh_inverse_filter_trim returns them), belongs to the device of
 * the context that created it and may be used from any context of that device. */
typedef struct bh_rl bh_rl;
int bh_richardson_lucy_create(bh_ctx* ctx, const float* psf, int64_t pz, int64_t py, int64_t px, int64_t Z, int64_t Y,
                              int64_t X, bh_rl** out);
int bh_richardson_lucy_apply(bh_ctx* ctx, const bh_rl* handle, const float* in, int iterations, float eps, float* out);
/* bh_richardson_lucy_apply that also reduces the row sums of its result: row_sums (device, float64 [Z * Y], may be NULL) receives
 * sum over x of out[z, y, x] from the last update pass when the handle's back-end can do so (*produced = 1; the fused engine on
 * rows of 512 / 1024 / 2048 voxels), and is left untouched otherwise (*produced = 0).  bh_deskew_rows takes them. */
int bh_richardson_lucy_apply_rows(bh_ctx* ctx, const bh_rl* h, const float* in, int iterations, float eps, float* out,
                                  double* row_sums, int* produced);
int bh_richardson_lucy_destroy(bh_rl* handle);
/* box / backend as bh_richardson_lucy_plan; otf_is_real: the PSF is point-symmetric, one float per bin is kept; any
 * output pointer may be NULL. */
int bh_richardson_lucy_info(const bh_rl* handle, int64_t box[3], int* backend, int* otf_is_real, uint64_t* otf_bytes);

/* Phase cross-correlation of two equally shaped float32 volumes (biahub/estimate_stabilization.py:199-256
 * phase_cross_corr): corr = irfftn(F1 conj(F2) / norm), norm = 1 | max(|F1 conj F2|, eps) | |F1||F2|.
 * shift (host, 3 floats) = argmax |corr| folded to the signed range like the reference; corr_shifted (device,
 * Z*Y*Xc floats with Xc = X - (X & 1), may be NULL) = fftshift(|corr|): the reference inverts without a shape, so
 * an odd last axis comes back one shorter.  Synchronises (the shift is returned to the host). */
#define BH_PCC_NORM_NONE 0
#define BH_PCC_NORM_MAGNITUDE 1
#define BH_PCC_NORM_CLASSIC 2
int bh_phase_cross_corr(bh_ctx* ctx, const float* ref, const float* mov, int64_t Z, int64_t Y, int64_t X,
                        int normalization, float shift[3], float* corr_shifted);

/* Prepared form for the stabilisation estimate's loop (biahub/estimate_stabilization.py:505-520: every timepoint against the
 * first one, or against its predecessor): `fixed`'s spectrum is computed once and kept in the handle; a call transforms the
 * other image only (5 passes over the volume instead of 8).  fixed_is_second: the stored image is phase_cross_corr's second
 * argument (the conjugated factor) rather than its first.  roll != 0: after the call `img` is the stored image (its spectrum
 * falls out of the call's own Z pass) — the "previous" reference.  Results equal bh_phase_cross_corr's on the same pair.
 * apply synchronises (the shift is returned to the host); corr_shifted may be NULL, and then no correlation volume is written
 * at all when the rows are ones the wave-private X passes take (X = 512, 1024, 2048). */
typedef struct bh_pcc bh_pcc;
int bh_phase_cross_corr_create(bh_ctx* ctx, const float* fixed, int64_t Z, int64_t Y, int64_t X, int fixed_is_second,
                               bh_pcc** handle);
int bh_phase_cross_corr_apply(bh_ctx* ctx, bh_pcc* handle, const float* img, int normalization, int roll, float shift[3],
                              float* corr_shifted);
int bh_phase_cross_corr_destroy(bh_pcc* handle);

/* ---- intensity-registration building blocks (biahub/registration/ants.py:55-122 estimate) ---------- */
/* The reference delegates to ants.registration(type_of_transform="Similarity", aff_shrink_factors (6,3,1),
 * aff_smoothing_sigmas (2,1,0), aff_iterations (2100,1200,50)) (:93-98,104-109): ITK's multi-resolution gradient descent
 * on the Mattes mutual-information metric.  These are its data-parallel pieces; the optimiser itself is host code.
 *
 * out[6] (host) = min, max, sum, sum*z, sum*y, sum*x of a float32 volume (Parzen intensity range; centre of mass for
 * the moments initialisation).  Synchronises. */
int bh_image_stats(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, double out[6]);

/* One pyramid level: per axis a separable Gaussian of sigma[a] input voxels (edge-clamped, radius ceil(4 sigma), 0 = none)
 * sampled at input index factor[a]*i + offset[a], i < out_shape[a] = max(1, N/factor), offset centring the kept samples.
 * out == NULL only fills out_shape / offset (host-only geometry query). */
int bh_smooth_shrink(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, const double sigma[3],
                     const int factor[3], float* out, int64_t out_shape[3], int64_t offset[3]);

/* Mattes mutual information between fixed(p) and moving(P p) over the fixed voxels offset, offset+stride, ... (raster
 * order), P = 3x4 row-major pull matrix in index space; trilinear moving interpolation, samples mapped outside
 * [0, N-1]^3 are dropped.  range = {fixed min, fixed max, moving min, moving max}; `bins` Parzen bins incl. 2 padding bins
 * per end, box window on fixed, cubic B-spline on moving.  value = MI (to be maximised), grad[12] = dMI/dP, nvalid = samples
 * used.  Bit-reproducible (integer histogram, fixed-order reductions).  Synchronises. */
int bh_mattes_mi(bh_ctx* ctx, const float* fixed, int64_t Zf, int64_t Yf, int64_t Xf, const float* moving, int64_t Zm,
                 int64_t Ym, int64_t Xm, const double P[12], const double range[4], int bins, int64_t stride,
                 int64_t offset, double* value, double grad[12], double* nvalid);

/* skimage.filters.sobel of a 3-D float32 volume (registration/ants.py:272-275 preprocessing option): gradient magnitude
 * sqrt((gz^2+gy^2+gx^2)/3) of the [1,0,-1] x [1,2,1]/4 x [1,2,1]/4 stencils, edges reflected.  out must not alias in. */
int bh_sobel(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, float* out);

/* ---- bead detection / PSF averaging (biahub/characterize_psf.py:562-711, biahub/estimate_psf.py:58-121) ------------- */
/* One peak candidate per pooling block: values[b] / indices[b] = max and flat input index (first maximum in z, y, x scan
 * order) of the k x k x k box mean (divisor = in-volume voxels: count_include_pad=False) over block b of
 * max_pool3d(kernel=block, stride=block, padding=block/2); nblocks[a] = (N + 2*(block/2) - block) / block + 1 and
 * b = (oz * nblocks[1] + oy) * nblocks[2] + ox.  values == indices == NULL only fills nblocks (host-only).
 * Bit-identical to torch's avg_pool3d + max_pool3d on CPU. */
int bh_block_peaks(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, int blur_kernel_size,
                   const int block[3], float* values, int64_t* indices, int64_t nblocks[3]);

/* For each of n patches [starts[3b..], starts + patch) (host int array, all fully inside the volume): peaks[b] (host) =
 * flat index inside the patch of the first maximum of the patch smoothed by a Gaussian of `sigma` voxels (zero outside
 * the patch, radius int(4 sigma + 0.5)) — BeadExtractor._compute_peak_offset.  Synchronises. */
int bh_patch_peaks(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, const int* starts, int n,
                   const int patch[3], double sigma, int64_t* peaks);

/* out (device, patch voxels) = mean over the n patches of patch / max(patch); with normalise != 0 followed by
 * out -= min(out); out /= max(out) (estimate_psf.py:104-112).  Synchronises. */
int bh_average_patches(bh_ctx* ctx, const float* in, int64_t Z, int64_t Y, int64_t X, const int* starts, int n,
                       const int patch[3], int normalise, float* out);

/* ---- affine warp ------------------------------------------------------------------- */
/* out(p) = in(M p), M = 3x4 row-major pull matrix (rows z,y,x; last column translation) in ZYX
 * index space; NaN inputs read as 0 (register.py:254).  The output is the sub-box
 * [crop_lo, crop_lo + out_shape) of the full target grid (register.py:278-279); pass crop_lo =
 * {0,0,0} for no crop. */
int bh_affine(bh_ctx* ctx, const void* in, int in_dtype, int64_t Zi, int64_t Yi, int64_t Xi,
              const double matrix[12], int interpolation, int boundary, float cval, float* out,
              int64_t Zo, int64_t Yo, int64_t Xo, const int64_t crop_lo[3]);

/* Cubic B-spline coefficients of a volume (the prefilter of scipy.ndimage.affine_transform(order=3, mode="constant"):
 * biahub/core/transform.py:374-396, biahub/register.py:271-272): per axis the recursive inverse of the sampled cubic
 * B-spline with mirror boundaries; axes of length 1 are left alone; NaN inputs read as 0.  in: (Z,Y,X) of in_dtype;
 * coef: float32 (Z,Y,X), distinct from in.  bh_affine(BH_INTERP_CUBIC) runs this into the context's scratch itself. */
int bh_spline_prefilter(bh_ctx* ctx, const void* in, int in_dtype, int64_t Z, int64_t Y, int64_t X, float* coef);

/* ---- bit-exact crop / flip --------------------------------------------------------- */
/* (C, Zi, Yi, Xi) of itemsize bytes -> (C, Zo, Yo, Xo) starting at lo, optionally flipped along
 * Y and/or X *after* the crop.  nan_to_zero (float32/float64 only) restates copy_n_paste's
 * np.nan_to_num.  Pure data movement: outputs are bit-identical to numpy slicing. */
int bh_crop_flip(bh_ctx* ctx, const void* in, int itemsize, int64_t C, int64_t Zi, int64_t Yi,
                 int64_t Xi, const int64_t lo[3], int64_t Zo, int64_t Yo, int64_t Xo, int flip_y,
                 int flip_x, int nan_to_zero, void* out);

/* ---- measurement support ----------------------------------------------------------- */
/* Elapsed milliseconds of the last call of each kind, measured with HIP events on the
 * context's stream (what: 0 deskew kernel, 1 fill passes, 2 RL total, 3 tikhonov, 4 affine,
 * 5 crop_flip, 6 one RL iteration (mean), 7 transfer function, 8 flat field). Synchronises. */
int bh_last_elapsed_ms(bh_ctx* ctx, int what, float* ms);
/* Enable/disable event timing (off by default: events cost a few us per call). */
int bh_ctx_set_timing(bh_ctx* ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* BHCORE_H */
