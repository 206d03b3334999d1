/* xvec_hip.h - C ABI of the MI355X-native x-vector / c-vector embedding extractor.
 *
 * The reference (mycrazycracy/speaker-embedding-with-phonetic-information) has no plugin, operator or
 * FFI interface for this path: its boundary is the command line of Kaldi's `nnet3-xvector-compute`
 * (call sites egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:86-87,92-93) plus Kaldi's file
 * formats.  The drop-in for that boundary is the `nnet3-xvector-compute` executable built from
 * csrc/nnet3_xvector_compute_main.cc; this header is the C ABI *underneath* it, i.e.
 * what a maintainer who wants to call the extractor from a host language binds.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; the caller owns every host buffer;
 * the library owns device memory; no exception crosses the boundary (every entry returns xv_status and
 * the message is available from xv_last_error(), thread-local); an xv_ctx may be used by one thread at
 * a time, different xv_ctx objects are independent.  There is NO CPU compute path behind this ABI:
 * creating a context without a gfx950 device fails with XV_ERR_DEVICE.
 */
#ifndef XVEC_HIP_H_
#define XVEC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  XV_OK = 0,
  XV_ERR_IO = 1,       /* unreadable rxfilename / malformed Kaldi object */
  XV_ERR_MODEL = 2,    /* graph outside the supported TDNN grammar */
  XV_ERR_DEVICE = 3,   /* no gfx950 device, HIP failure */
  XV_ERR_ARG = 4,      /* invalid argument (e.g. a chunk shorter than the network context) */
  XV_ERR_INTERNAL = 5
} xv_status;

typedef enum {
  XV_PREC_DEFAULT = -1, /* the policy of the command-line tools, and what every entry point of this library defaults to:
                          the context is PACKED as XV_PREC_FP16MX2 for a pooled (x-vector) output whose layers can all
                          run it, XV_PREC_FP16X3 otherwise (a layer the 4-bit walk cannot cover, frame-level outputs);
                          xv_ctx_info reports that mode.  On its own this meets the parity bar on every model tried,
                          heavy-tailed ones included.  A context packed this way can be
                          CALIBRATED (xv_ctx_calibrate / xv_calibrate_table; the command-line tools and dist_extract.py
                          do it by default, xv_forward_batch users only when they call it): it then runs the lighter,
                          model-dependent XV_PREC_FP16MX where - and only where - that was measured within the
                          tolerance on a sample of the job's own data; xv_ctx_fast_mode reports what it runs */
  XV_PREC_BF16X3 = 0,  /* split-bf16 MFMA (3 products, fp32 accumulate): fp32-grade, the first version's parity mode */
  XV_PREC_BF16 = 1,    /* single-pass bf16 MFMA */
  XV_PREC_FP16 = 2,    /* single-pass fp16 MFMA */
  XV_PREC_FP16X3 = 3,  /* split-fp16 MFMA (3 products of 11+11-bit operands, fp32 accumulate) */
  XV_PREC_FP16X2 = 4,  /* fp16 activations x split-fp16 weights (2 products): removes the weight rounding error, which
                          is coherent over the frames of a chunk; the activation rounding error averages out in the
                          statistics pooling.  Layers after the pooling run XV_PREC_FP16X3.  Not for frame-level
                          outputs (they run XV_PREC_FP16X3) */
  XV_PREC_AUTO = 5,    /* XV_PREC_FP16MX for chunks that pool >= 300 frames (XVEC_FAST_MIN_POOLED), XV_PREC_FP16X3 for
                          shorter ones; the choice depends on the chunk's own length only */
  XV_PREC_FP16MX = 6,  /* XV_PREC_FP16X2 with the second product (activations x weight residual) done on block-scaled
                          4-bit operands at four times the fp16 MFMA rate (v_mfma_scale_f32_16x16x128_f8f6f4): 1.25
                          passes per product.  The residual term is 2^-11 of the product and only has to be good to
                          ~4 bits.  Layers whose K length is not a multiple of 128 run XV_PREC_FP16X2 */
  XV_PREC_FP16MX2 = 7, /* XV_PREC_FP16MX plus a block-scaled 4-bit product for what the fp16 rounding of the ACTIVATIONS
                          dropped (the producing layer writes that residual as a 4-bit plane): 1.5 passes per product,
                          error independent of how well the pooling averages the activation rounding.  Layers whose K
                          length is not a multiple of 128 run XV_PREC_FP16X3E */
  XV_PREC_FP16X3E = 8, /* kernel mode of XV_PREC_FP16MX2 passes: XV_PREC_FP16X3 whose planes epilogue writes the fp16
                          plane + its 4-bit residual instead of two fp16 planes */
  XV_PREC_FP16MXE = 9  /* kernel mode of XV_PREC_FP16MX2 passes: XV_PREC_FP16MX whose planes epilogue writes the fp16 plane +
                          its 4-bit residual (a "lite" layer, xv_calibration.lite_mask, in front of a 1.5-pass consumer) */
} xv_precision;

typedef struct xv_model xv_model; /* host side: parsed nnet3 model lowered to a TDNN program */
typedef struct xv_ctx xv_ctx;     /* device side: weights resident on one GPU + workspaces + stream */

typedef struct {
  int32_t input_dim;         /* feature dimension (23 in egs/sre, conf/mfcc.conf:5) */
  int32_t output_dim;        /* embedding dimension (512) */
  int32_t left_context;      /* frames not computable at the left edge of a chunk (7 / 13) */
  int32_t right_context;     /* ... at the right edge (7) */
  int32_t min_frames;        /* smallest chunk with >= 1 pooled frame (15 / 21) */
  int32_t num_layers;        /* affine layers in the dependency cone of the output */
  int32_t output_is_segment; /* 1: one vector per chunk (x-vector); 0: one row per frame */
  int32_t reserved;
} xv_model_info_t;

/* Message of the last failure on the calling thread ("" if none). */
const char* xv_last_error(void);
const char* xv_version(void);

/* ---- model (replaces ReadKaldiObject + SetBatchnormTestMode + CollapseModel + Compile) -------------
 * raw/n: the bytes of an nnet3 "raw" model, binary or text (what `$srcdir/final.raw` holds,
 *        extract_xvectors_new.sh:53).
 * nnet_config: optional text applied like `nnet3-copy --nnet-config=<file>` (extract_xvectors_new.sh:58-59
 *        writes "output-node name=output input=tdnn6.affine" into extract.config); may be NULL.
 * output_node: name of the output-node to compute; NULL means "output". */
xv_status xv_model_load(const void* raw, size_t n, const char* nnet_config, const char* output_node,
                        xv_model** out);
/* Same, reading a Kaldi rxfilename: "file", "file:offset", "-", or "command |" (the form every recipe
 * call uses: "nnet3-copy --nnet-config=... final.raw - |"). */
xv_status xv_model_load_rxfilename(const char* rxfilename, const char* nnet_config, const char* output_node,
                                   xv_model** out);
void xv_model_free(xv_model* m);
xv_status xv_model_info(const xv_model* m, xv_model_info_t* info);
/* Algorithmic multiply-accumulates of one chunk of `frames` frames (unpadded dims, only the frames
 * nnet3 would compute). */
double xv_model_macs(const xv_model* m, int32_t frames);
/* Human-readable layer table; returns the number of bytes needed (including the NUL). */
size_t xv_model_describe(const xv_model* m, char* buf, size_t n);
/* Packed, padded, precision-split weight image: what one rank broadcasts to the others over RCCL
 * (SURVEY.md §8(e)).  Call with blob == NULL to get the size. */
xv_status xv_model_pack(const xv_model* m, int precision, void* blob, size_t* nbytes);

/* ---- device context ---------------------------------------------------------------------------- */
xv_status xv_ctx_create(const xv_model* m, int device, int precision, xv_ctx** out);
xv_status xv_ctx_create_from_blob(const void* blob, size_t nbytes, int device, xv_ctx** out);
/* Same, the packed image being DEVICE memory on `device` (what a rank holds after the RCCL broadcast of the weights,
 * SURVEY.md section 8(e)): the weights go device to device, only the header and layer table are read back. */
xv_status xv_ctx_create_from_device_blob(const void* blob_dev, size_t nbytes, int device, xv_ctx** out);
void xv_ctx_free(xv_ctx* c);
xv_status xv_ctx_info(const xv_ctx* c, xv_model_info_t* info, int32_t* precision, int32_t* device);

/* One forward pass over a batch of B chunks (replaces RunNnetComputation for B chunks at once).
 * feats: packed rows [row_offsets[B]][input_dim] fp32; chunk b = rows row_offsets[b] .. row_offsets[b+1]-1;
 * out: [B][output_dim] fp32.  Every chunk must have >= min_frames rows (XV_ERR_ARG otherwise; nnet3
 * would refuse to compile the same request).  Host-buffer flavour: blocking.
 * Frame-level models (xv_model_info_t.output_is_segment == 0: the output node does not follow the statistics pooling,
 * e.g. senone log-posteriors `output.log-softmax` or bottleneck features `tdnn5.batchnorm`; what the reference gets
 * from `nnet3-compute`, sid/nnet3_cvector/cvector/extract_log_post.sh:77-84, sid/nnet3_cvector/am/extract_bn.sh:68):
 * the output has ONE ROW PER INPUT FRAME - chunk b occupies rows row_offsets[b]..row_offsets[b+1]-1 of out, exactly
 * like its features - the chunk being extended by replicating its first / last frame over the network's left / right
 * context (nnet3-compute's edge behaviour); any chunk length >= 1 is accepted. */
xv_status xv_forward_batch(xv_ctx* c, const float* feats, const int32_t* row_offsets, int32_t B, float* out);
/* Device-buffer flavour: feats/out are device pointers on the context's GPU, row_offsets is a HOST array;
 * asynchronous on hip_stream (a hipStream_t, NULL = the context's own stream). out rows are out_ld apart. */
xv_status xv_forward_batch_device(xv_ctx* c, const float* feats_dev, const int32_t* row_offsets, int32_t B,
                                  float* out_dev, int32_t out_ld, void* hip_stream);
/* Waits for everything the context launched, on its own streams and on callers' streams, and reports what only shows
 * once the kernels ran: XV_ERR_DEVICE if a persistent GEMM launch gave up waiting for another workgroup's partial tile
 * (results of that launch are invalid; xv_last_error() says how to select the per-tile kernels).  A caller of
 * xv_forward_batch_device on its own stream calls this before trusting the results. */
xv_status xv_ctx_synchronize(xv_ctx* c);
/* Per-kernel timing with HIP events recorded on the stream the kernels are launched on (used by bench.py for
 * the roofline figure).  xv_ctx_profile_report synchronises, writes "label<TAB>launches<TAB>total_ms" lines for
 * everything recorded since the previous report into buf (NUL terminated), resets the record, and returns
 * the number of bytes needed; call it often enough - every recorded forward keeps a handful of events alive. */
/* ---- calibrated arithmetic -----------------------------------------------------------------------------------------
 * A context packed as XV_PREC_FP16MX2 (what XV_PREC_DEFAULT resolves to for a pooled output) can also run the lighter
 * XV_PREC_FP16MX arithmetic (1.25 instead of 1.5 MFMA passes, +30 % throughput) and the three-pass XV_PREC_FP16X3 on the
 * same packed weights.  XV_PREC_FP16MX meets the parity bar on some models only, so it is never
 * assumed: xv_ctx_calibrate runs the caller's own chunks in all three, compares the embeddings of the two fast modes with
 * the three-pass ones (worst max|d| / max|ref|; XV_PREC_FP16MX over the chunks that pool >= 300 frames - the others run
 * three-pass in that mode -, XV_PREC_FP16MX2 over every chunk it runs fast, from 160 pooled frames) and switches the context to
 * XV_PREC_FP16MX only when at least 16 such chunks were compared, its worst chunk stays within tol (the tools use 7.5e-5: three
 * quarters of the 1e-4 bar) AND the tail its errors project (mean + 6 standard deviations over those chunks, xv_calibration.tail)
 * stays within tol x 1.10 - over 32 768 chunks the worst one was measured 4.6-5.9 standard deviations above the mean
 * (profiles/r05_tail_error.md) -, else leaves XV_PREC_FP16MX2
 * (or drops to XV_PREC_FP16X3 should even that exceed 1e-4 on a sampled chunk, or project a tail beyond tol x 1.20).  Contexts that cannot switch report their precision with
 * checked = 0.  (No reference counterpart: Kaldi computes in fp32 throughout.)  A measured choice depends on the sample it was
 * measured on, so the command-line tools never make one per job by default - `--precision=default` is plain XV_PREC_FP16MX2, a
 * function of the model alone - and use a measured choice only when it is SHARED between the jobs of a recipe through a
 * calibration file (xv_ctx_share_calibration below; `--calibration=<file>` / $XVEC_CALIBRATION).  xv_ctx_set_fast_mode applies a choice made elsewhere
 * (the other ranks of a multi-GPU job); xv_calibrate_table calibrates on max_utts utterances of a table: spread evenly
 * over the whole list where its objects can be addressed (archive file, script file - the reference's lists are sorted
 * by speaker, utils/data/split_data.sh:18-21, so the head of a list is one or two speakers), the head of a stream. */
typedef struct {
  int32_t chosen;   /* xv_precision the context now runs its fast chunks in */
  int32_t checked;  /* chunks compared */
  float err_mx;     /* XV_PREC_FP16MX against XV_PREC_FP16X3 */
  float err_mx2;    /* XV_PREC_FP16MX2 against XV_PREC_FP16X3 */
  int32_t checked_mx;  /* of them, chunks XV_PREC_FP16MX would run fast (what err_mx was measured on; fewer than 16: not chosen) */
  float err_lite;      /* chosen == XV_PREC_FP16MX2 with lite_mask != 0: error of that mixture over ALL checked_mx chunks */
  uint64_t lite_mask;  /* chosen == XV_PREC_FP16MX2: the layers (bit = index in xv_model_describe's table) that compute in 1.25 passes
                          inside the 1.5-pass context - where XV_PREC_FP16MX as a whole misses tol, the calibration keeps the
                          second walk (the correction of the activations' fp16 rounding) only on the layers that need it and
                          takes it off the most expensive ones the tolerance allows; 0: none.  The mixture is SELECTED on the
                          chunks at even positions of the sample and CONFIRMED on those at odd positions, which took no part
                          in the selection: layers are dropped again, last added first, until the held-out half is within tol */
  float err_holdout;        /* error of the adopted mixture over the held-out half (what confirmed it) */
  int32_t checked_holdout;  /* chunks of the held-out half XV_PREC_FP16MX would run fast */
  int32_t lite_dropped;     /* layers the selection half admitted and the held-out half threw out again */
  float tail;               /* projected tail of the per-chunk error of what was adopted: mean + 6 standard deviations over the
                               chunks that confirmed it (the whole sample; a mixture: its held-out half), never below their worst.
                               A configuration is adopted only while this is within tol x 1.10 (7.5e-5 -> 8.25e-5: the projection has
                               been measured up to 10 % below the worst of 32 768 chunks, profiles/r05_tail_error.md) */
} xv_calibration;
xv_status xv_ctx_calibrate(xv_ctx* c, const float* feats, const int32_t* row_offsets, int32_t B, float tol, xv_calibration* out);
xv_status xv_ctx_set_fast_mode(xv_ctx* c, int32_t precision);   /* also clears the lite layers */
xv_status xv_ctx_fast_mode(const xv_ctx* c, int32_t* precision);
/* The lite layers of a context running XV_PREC_FP16MX2 (xv_calibration.lite_mask): set applies a choice made elsewhere
 * (bits of layers that cannot run the 1.25-pass arithmetic are dropped; XV_ERR_ARG in any other fast mode), get reports it.
 * The mask has 64 bits: a layer with index >= 64 in xv_model_describe's table is never lite (the reference's deepest graph on
 * this path has 13 layers). */
xv_status xv_ctx_set_lite_layers(xv_ctx* c, uint64_t mask);
xv_status xv_ctx_lite_layers(const xv_ctx* c, uint64_t* mask);
xv_status xv_calibrate_table(xv_ctx* c, const char* feature_rspecifier, int32_t chunk_size, int32_t min_chunk_size,
                             int32_t pad_input, int32_t max_utts, float tol, xv_calibration* out);
/* xv_extract_table calibrates on a sample of its own table first (as xv_calibrate_table) when this is enabled (default: off) */
xv_status xv_ctx_set_calibration(xv_ctx* c, int32_t enable, float tol);
/* ---- the shared choice of a recipe (csrc/calib_file.h) -------------------------------------------------------------------
 * The reference splits a list over `nj` independent processes and concatenates their outputs
 * (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:72,91-99); an utterance gets the same vector in whatever shard it lands.
 * For that to hold here, a MEASURED choice of arithmetic must be one choice for all jobs: a small text file beside the model
 * ("xvec-calibration 1", the fingerprint of the packed model image, the arithmetic, the mask of lite layers, provenance).
 * xv_ctx_model_fingerprint: the fingerprint of the context's packed image (part of the image header, so a context built from a
 * broadcast image knows it too).  xv_ctx_share_calibration: the file exists -> its choice is applied to the context
 * (*outcome = 0; XV_ERR_IO when it names another model image or cannot be parsed); it does not -> the context's CURRENT choice
 * (after xv_ctx_calibrate / xv_calibrate_table, or the packed default) is published atomically (temporary file + link(2): the
 * first of several concurrent publishers wins) and what the file then holds is applied: *outcome = 1 when this context's
 * choice was published, 2 when another job's was adopted.  note: one line of provenance written into the file (may be NULL).
 * xv_ctx_set_calibration_file: xv_extract_table does the same before its first batch (file present: applied, nothing measured;
 * absent: measured on the job's own sample, published, adopted).  NULL / "" turns it off.
 * xv_calibration_file_read / _publish: the file itself, without a context (no device call): read reports *found = 0 for a
 * missing file; publish writes (model, precision, lite_mask) unless the file exists and reports what the file then holds. */
xv_status xv_calibration_file_read(const char* path, int32_t* found, uint64_t* model, int32_t* precision, uint64_t* lite_mask);
xv_status xv_calibration_file_publish(const char* path, uint64_t model, int32_t precision, uint64_t lite_mask, float tol,
                                      const char* note, int32_t* published, uint64_t* adopted_model, int32_t* adopted_precision,
                                      uint64_t* adopted_lite_mask);
xv_status xv_ctx_model_fingerprint(const xv_ctx* c, uint64_t* fingerprint);
xv_status xv_ctx_share_calibration(xv_ctx* c, const char* path, float tol, const char* note, int32_t* outcome);
xv_status xv_ctx_set_calibration_file(xv_ctx* c, const char* path);

xv_status xv_ctx_set_profiling(xv_ctx* c, int32_t enable);
size_t xv_ctx_profile_report(xv_ctx* c, char* buf, size_t n);

/* The per-utterance loop of nnet3-xvector-compute (chunking + length-weighted average):
 * utterance u = rows row_offsets[u] .. row_offsets[u+1]-1 of feats (host).  chunk_size <= 0 means the whole
 * utterance; pad_input != 0 replicates edge frames of chunks shorter than min_chunk_size, pad_input == 0
 * skips them.  ok[u] = 1 if an embedding was written to out[u][:], 0 if the utterance counts as failed
 * (zero frames, too short).  Blocking. */
xv_status xv_extract_utterances(xv_ctx* c, const float* feats, const int32_t* row_offsets, int32_t n_utts,
                                int32_t chunk_size, int32_t min_chunk_size, int32_t pad_input, float* out,
                                int32_t* ok);

/* The whole job of one nnet3-xvector-compute process on an existing context: read a Kaldi feature table
 * (rspecifier, e.g. "scp:feats.scp" or "ark:apply-cmvn-sliding ... |"), extract, write a vector table
 * (wspecifier, e.g. "ark,scp:xvector.1.ark,xvector.1.scp").  Per-utterance problems are warnings on stderr and
 * are counted in *num_failed; the call fails only for fatal I/O or device errors.  batch_frames <= 0: default.
 * This is what a multi-GPU launcher calls per rank after the weights were broadcast. */
xv_status xv_extract_table(xv_ctx* c, const char* feature_rspecifier, const char* vector_wspecifier, int32_t chunk_size,
                           int32_t min_chunk_size, int32_t pad_input, int32_t batch_frames, int64_t* num_done,
                           int64_t* num_failed);

/* Feature front-end on the device (the two pipe stages of extract_xvectors_new.sh:79): sliding-window cepstral mean
 * subtraction (`apply-cmvn-sliding --norm-vars=false --center=<center> --cmn-window=<cmn_window>`; cmn_window <= 0: none)
 * over each whole utterance, then `select-voiced-frames` with vad[r] != 0 (vad: one float per raw row, NULL = keep
 * all).  raw: packed rows, utterance u = rows raw_off[u]..raw_off[u+1]-1; out receives the kept rows (capacity >= the
 * number of raw rows), out_off[n_utts+1] their offsets.  Host buffers, blocking. */
/* The feature pipeline every extraction script of the reference builds - "ark:apply-cmvn-sliding --norm-vars=false
 * --center=true --cmn-window=300 scp:feats.scp ark:- | select-voiced-frames ark:- scp,s,cs:vad.scp ark:- |"
 * (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:79 and its siblings) - recognised as text (csrc/fuse_pipe.h): *found = 1
 * and the inner feature table, the VAD table ("" when there is no selection stage) and the sliding-CMN parameters when the string
 * is exactly that pipeline with options the device front-end implements; *found = 0 for anything else.  nnet3-xvector-compute
 * uses it to run the two stages on the device instead of as two CPU tools and two pipes. */
xv_status xv_recognize_feature_pipeline(const char* rspecifier, int32_t* found, char* feats, size_t feats_cap, char* vad, size_t vad_cap,
                                        int32_t* cmn_window, int32_t* min_cmn_window, int32_t* center);
xv_status xv_frontend_cmvn_select(xv_ctx* c, const float* raw, const int32_t* raw_off, int32_t n_utts, const float* vad,
                                  int32_t cmn_window, int32_t center, float* out, int32_t* out_off);

/* Host-only: the chunk list nnet3-xvector-compute would build for one utterance of num_rows frames
 * (SURVEY.md App. B.5).  Arrays of capacity `cap` receive per chunk: first source row, frames taken (= averaging
 * weight), copies of the first / last frame added by --pad-input.  *n_chunks = number of chunks; returns XV_ERR_ARG
 * with the reason in xv_last_error() when the utterance counts as failed (0 frames, too short). */
xv_status xv_plan_chunks(int32_t num_rows, int32_t chunk_size, int32_t min_chunk_size, int32_t pad_input,
                         int32_t min_net_frames, int32_t cap, int32_t* start, int32_t* len, int32_t* left_pad,
                         int32_t* right_pad, int32_t* n_chunks);

/* ---- multi-GPU: weights read once, broadcast over RCCL/xGMI --------------------------
 * Single-process form: creates one context per device in devices[0..n) from the model, reading/packing once
 * on the host, uploading to devices[0] and broadcasting device-to-device with one ncclBroadcast (n == 1 included:
 * a one-rank communicator; every context is built from the bytes its device received, without a host round trip). */
xv_status xv_ctx_create_broadcast(const xv_model* m, const int* devices, int n, int precision, xv_ctx** out);

/* ---- speaker-level back-end ----------------------------------------------
 * What the reference does to the vectors right after extraction, as device kernels behind host buffers
 * (no Kaldi FFI exists for these either; they replace the processes of egs/sre/v2/run_sre10.sh:238-241 and
 * egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:106-107):
 *   xv_backend_apply   ivector-subtract-global-mean (mean != NULL) -> transform-vec (transform != NULL; t_cols == dim
 *                      linear, dim + 1 affine, anything else XV_ERR_ARG "Dimension mismatch") ->
 *                      ivector-normalize-length (normalize != 0; scaleup as Kaldi's --scaleup).
 *                      out[n][out_dim], out_dim = transform ? t_rows : dim; ratio[n] (optional) = |y|/sqrt(out_dim)
 *                      (|y| without scaleup) before normalisation; a zero vector is left unchanged (ratio 0).
 *   xv_segment_mean    ivector-mean: out[s] = mean of rows idx[seg_off[s] .. seg_off[s+1]) of x, added in list order,
 *                      fp32 accumulator (speaker means) or fp64 (acc64 != 0, the global mean); empty segment -> zeros.
 * Both fail with XV_ERR_DEVICE when no gfx950 device is usable. */
xv_status xv_backend_apply(int device, const float* x, int32_t n, int32_t dim, const float* mean, const float* transform,
                           int32_t t_rows, int32_t t_cols, int32_t normalize, int32_t scaleup, float* out, float* ratio);
xv_status xv_segment_mean(int device, const float* x, int32_t n, int32_t dim, const int32_t* seg_off, const int32_t* idx,
                          int32_t n_seg, int32_t acc64, float* out);

/* ---- kernel-level entry (unit tests of the HIP GEMM against a plain fp32 reference) ------------------- */
typedef struct {
  const void* hi;   /* device plane (bf16 / fp16) at logical row 0 */
  const void* lo;   /* residual plane (XV_PREC_BF16X3 / XV_PREC_FP16X3 only) */
  int32_t ld;       /* leading dimension in elements */
  int32_t row_shift;
  int32_t k_len;    /* multiple of 32 */
  const void* gmax; /* XV_PREC_FP16MX: device uint32 [rows/16], float bits of max |x| per 16-row group of this plane */
  /* XV_PREC_FP16MX2: 4-bit image of (activation - hi), [rows][ld / 2 bytes], and its E8M0 scales [rows][ld / 64 rounded up to a multiple of 4] (what
   * epilogue 0 of XV_PREC_FP16MX2 / XV_PREC_FP16X3E / XV_PREC_FP16MXE wrote to out_lo4 / out_lo4_scale), both at logical row 0 */
  const void* lo4; const void* lo4_scale;
} xv_seg_desc;
typedef struct {
  int32_t precision, epilogue; /* epilogue: 0 planes out, 1 fp32 out, 2 per-16-row (sum, sumsq) partials */
  int32_t nseg;
  xv_seg_desc seg[8];
  const void* w_hi; const void* w_lo; int32_t ldw;
  int32_t rows;     /* multiple of 128 */
  int32_t n_pad;    /* multiple of 128 */
  const float* bias; const float* scale; const float* offset;
  int32_t relu, bn;
  void* out_hi; void* out_lo; int32_t ldo;
  float* out_f32; int32_t ldf; int32_t m_valid;
  float* partial; int32_t ldp; const int8_t* grp_range;
  void* hip_stream;
  /* XV_PREC_FP16MX: e2m1 residual plane [n_pad][ldw4 bytes] in K-walk order + its E8M0 scales, one per row, block of four K
   * steps and lane group, in the staging order of this epilogue (xv_tile_mx_scales; see kernels.h);
   * gmax_out (epilogue 0, any precision): device uint32 [rows/16], receives the group maxima of the output plane */
  const void* w4; int32_t ldw4; const void* w4_scale;
  void* gmax_out;
  /* XV_PREC_FP16MX2: 4-bit image of the weights for the second K walk (xv_pack_mx_weights) + scales in staging order;
   * out_lo4 / out_lo4_scale (epilogue 0 of XV_PREC_FP16MX2 / XV_PREC_FP16X3E / XV_PREC_FP16MXE): see xv_seg_desc */
  const void* w4b; int32_t ldw4b; const void* w4b_scale;
  void* out_lo4; void* out_lo4_scale;
  /* != 0: run tdnn_gemm_kernel_p8 (256 x 256 tiles, K tiles of 64 columns; XV_PREC_FP16, XV_PREC_FP16MX and XV_PREC_FP16MX2,
   * epilogues 0 and 2; rows and n_pad multiples of 256).  Its K walk is group -> 64-column chunk -> offset: w4 / w4_scale must
   * come from xv_pack_mx_residual64 (and, XV_PREC_FP16MX2, w4b / w4b_scale from xv_pack_mx_weights64 with ldw4b = 2 * ldw).  A layer runs this kernel for every launch of a mode or for none (its sums are formed in
   * another order than the 32-column kernels'). */
  int32_t p8;
} xv_gemm_desc;
/* Host helper for the test above: packs the e2m1 residual plane of one weight matrix exactly like xv_model_pack does
 * (w, w_hi_f16: [n_pad][k_len] row-major, k_len = sum of the segments' k_len; seg_src[j] equal = same source plane).
 * w4 receives n_pad * (k_len / 128 * 64) bytes, w4_scale n_pad * (k_len / 32) bytes.  XV_ERR_ARG when the walk has a group that is
 * not a multiple of four steps. */
xv_status xv_pack_mx_residual(const float* w, const uint16_t* w_hi_f16, int32_t n_pad, int32_t nseg,
                              const int32_t* seg_src, const int32_t* seg_shift, const int32_t* seg_klen, uint8_t* w4,
                              uint8_t* w4_scale);
/* the same in the K-walk order of tdnn_gemm_kernel_p8 (xv_gemm_desc.p8) */
xv_status xv_pack_mx_residual64(const float* w, const uint16_t* w_hi_f16, int32_t n_pad, int32_t nseg, const int32_t* seg_src,
                                const int32_t* seg_shift, const int32_t* seg_klen, uint8_t* w4, uint8_t* w4_scale);
/* XV_PREC_FP16MX2: the 4-bit image of the weights for the second K walk (w: [n_pad][k_len], the values of the fp16 planes'
 * domain; every k_len a multiple of 128): w4b receives n_pad * (k_len / 128 * 64) bytes, w4b_scale n_pad * (k_len / 32)
 * bytes in natural order (tile them with xv_tile_mx_scales). */
xv_status xv_pack_mx_weights(const float* w, int32_t n_pad, int32_t nseg, const int32_t* seg_src, const int32_t* seg_shift,
                             const int32_t* seg_klen, uint8_t* w4b, uint8_t* w4b_scale);
/* the same in the order of tdnn_gemm_kernel_p8's second walk (tiles of 256 columns: every k_len a multiple of 256), rows of
 * k_len * 2 bytes - the pitch of the fp16 plane, which is what xv_gemm_desc.ldw4b must then say: w4b receives n_pad * k_len * 2 bytes */
xv_status xv_pack_mx_weights64(const float* w, int32_t n_pad, int32_t nseg, const int32_t* seg_src, const int32_t* seg_shift,
                               const int32_t* seg_klen, uint8_t* w4b, uint8_t* w4b_scale);
/* natural[n_pad][k_len / 32] (what xv_pack_mx_residual / xv_pack_mx_weights wrote) -> the order the kernels stage the scales in for the given
 * epilogue (xv_gemm_desc.epilogue): n_pad * (k_len / 32) bytes.  n_pad must be a multiple of 128. */
xv_status xv_tile_mx_scales(const uint8_t* natural, int32_t n_pad, int32_t k_len, int32_t epilogue, uint8_t* tiled);
xv_status xv_kernel_tdnn_gemm(const xv_gemm_desc* d);

#ifdef __cplusplus
}
#endif
#endif /* XVEC_HIP_H_ */
