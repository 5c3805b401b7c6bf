/*
 * volpick_hip.h — C ABI of libvolpick_hip.so, the MI355X (gfx950) implementation of
 * volpick's sliding-window phase-picking inference path.
 *
 * The reference reaches this path through a Python object API, not an FFI
 * (SURVEY.md §8b): seisbench.models.{PhaseNet,EQTransformer} instances created with
 * from_pretrained("volpick") and driven by classify()/annotate()/__call__
 * (/root/reference README.md:46-66, Final_models/demo.ipynb:300-301,359-360,397-398,
 * volpick/model/eval_taks0.py:58-89).  Each entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no torch types.
 *
 * Conventions: every function returns 0 on success or a negative vp_status;
 * vp_last_error() returns a thread-local message.  Handles are opaque; one handle is
 * bound to one HIP device and one HIP stream and is used from one thread at a time.
 * "dev" pointers are HIP device pointers on the handle's device, "host" pointers are
 * ordinary host memory.  All floats are IEEE fp32.
 */
#ifndef VOLPICK_HIP_H
#define VOLPICK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp_handle vp_handle;

typedef enum {
  VP_OK = 0,
  VP_ERR_INVALID = -1,   /* bad argument */
  VP_ERR_HIP = -2,       /* a HIP runtime call failed */
  VP_ERR_NOMEM = -3,     /* workspace too small / allocation failed */
  VP_ERR_UNSUPPORTED = -4
} vp_status;

enum { VP_MODEL_PHASENET = 0, VP_MODEL_EQTRANSFORMER = 1 };
enum { VP_NORM_PEAK = 0, VP_NORM_STD = 1 };
enum { VP_STACK_AVG = 0, VP_STACK_MAX = 1 };
enum { VP_MEM_HOST = 0, VP_MEM_DEVICE = 1 };

/* Constants the reference does not pin; defaults are the
 * published SeisBench values, filled in by vp_default_config(). */
typedef struct {
  int32_t norm;               /* VP_NORM_*  — model_args.norm, Final_models/ ** /volpick.json.v1:6 */
  int32_t norm_amp_per_comp;  /* EQT only: 1 = per-channel PEAK amplitude (whatever `norm` says, as SeisBench's
                                 annotate_batch_pre), 0 = one peak / std amplitude over all channels */
  int32_t max_batch;          /* windows the workspace is sized for per forward pass */
  float bn_eps;               /* BatchNorm eps, folded into the conv weights at load time */
  float attention_eps;        /* EQT additive attention denominator eps */
  float layernorm_eps;        /* EQT LayerNormalization eps */
  float norm_eps;             /* x / (amp + norm_eps) */
  int32_t taper_samples;      /* EQT half-cosine taper length (0 for PhaseNet) */
  int32_t plan_flags[8];      /* plan selectors (debug plans, A/B timing; all 0 = the default plan):
                                 [0]: 1 = PhaseNet layer-by-layer plan instead of the fused kernels (debug / A-B);
                                 [1]: bit0 = fused kernels also dump their LDS intermediates to the debug tensors
                                      (selects the three-launch plan), bit1 = per-layer clock stamps;
                                 [2]: PhaseNet: (1 = hand-pipelined K loop in the MFMA layers: removed in round 6, rejected);
                                      EQTransformer: 1 = the three
                                      BiLSTM blocks, two transformer blocks and the pick branches as six launches instead
                                      of the one-launch middle kernel, 2 = eqt_mid_kernel with one window per 512-thread
                                      workgroup, 3 = with two windows per 1024-thread workgroup (teams of eight waves: the
                                      default of rounds 3-5); default: eqt_mid4_kernel, FOUR windows per 1024-thread
                                      workgroup in teams of four waves, so that a batch of 256 holds 64 CUs and leaves the
                                      rest to the kernels of the other device contexts (all three bit-identical);
                                 [3]: 1 = the tiled level-0 up kernel of the three-launch plans with one workgroup per tile
                                      (A/B; 2, the persistent down kernel, was removed in round 6); 64 = PhaseNet's one-launch plan
                                      WITHOUT the first-come-first-served gate between the device contexts' forward
                                      launches (A/B: csrc/api.hip ForwardGate);
                                 [4]: 1 = no L2 warm-up of the weight streams;
                                 [5]: PhaseNet plan: 0 = the whole network in one launch, its five deepest layers on the
                                      bf16 matrix cores with exact three-piece operands (default), 1 = three launches,
                                      all MFMA (bit-identical to the layer plan), 2 = three launches, level-0
                                      stride-1 convs on the VALU, 3 = one launch, every core layer on the fp32 MFMA
                                      (the reference of the bf16-piece layers), 8 = one launch with inc and down0.same
                                      as packed-FMA direct convolutions on the vector ALUs and up3.convT / up3.same /
                                      head in round 4's forms (the rounding reference of the tiled level-0 layers);
                                      4, 5, 6, 7, 9 -- intermediate forms of rounds 2-5 -- were removed in round 6 and
                                      are rejected (default: all of these on the bf16 matrix cores too -- up1.same /
                                      up2.same in two K halves over one piece image that is refilled in between, inc /
                                      down0.same time-tiled in six tiles of 512 samples, up3.convT / up3.same / head in
                                      twelve tiles of 256 with producer and consumer waves);
                                 [6]: PhaseNet: 1 = the one-launch plan reads the input tensor filled by gather_normalize
                                      instead of cutting and normalising its windows itself; EQTransformer: 2 = the fused
                                      encoder front cuts and normalises its windows itself (A/B: slower end to end);
                                 [7]: EQTransformer: bit0 = decoder stages 4-6 + heads as three launches instead of the
                                      time-tiled fused kernel, bit1 = decoder stages 0-3 as five launches instead of one
                                      per-row fused kernel, bit2 = encoder stages 0-2 as three launches instead of the
                                      time-tiled fused kernel, bit3 = encoder stages 3-6 as four launches instead of one
                                      per-window fused kernel (all bit-identical; layer tests, A/B timing),
                                      bit4 = the fused ResCNN kernel, bit5 = stages 1 and 2 of the fused decoder 0-3
                                      kernel, bit6 = the fused decoder tail (stages 4-6 + heads), bit7 = the fused encoder 3-6 kernel, bit8 = stages 1 and 2 of the fused encoder 0-2 kernel on the fp32 MFMA
                                      instead of the bf16 matrix cores with exact three-piece operands (the two
                                      forms agree to fp32 rounding, not bitwise), bit9 = the bf16-piece ResCNN kernel with
                                      four waves per window and ONE window per 256-thread workgroup,
                                      bit10 = the decoder tail computes every tile of a row even where annotate / classify
                                      blind the output (default: only the tiles that hold kept samples),
                                      bit11 = stage 3 of the fused decoder 0-3 kernel shares its n-tiles evenly between
                                      the two waves of a SIMD (default: 14 + 10),
                                      bit13 = the bf16-piece ResCNN kernel with four waves per window and TWO windows per
                                      512-thread workgroup (the default of rounds 5-6; default now: eqt_res3t_kernel, eight
                                      waves per THREE windows, a wave taking 16 output channels of five or four of the nine
                                      n-tiles; the three forms are bit-identical),
                                      (bit12, the bf16-piece ResCNN kernel with eight waves per window and K split over
                                      wave pairs, was removed in round 6 and is rejected) */
  int32_t reserved[4];        /* must be 0 */
} vp_config;

/* Fills cfg with the defaults for model_kind. */
int vp_default_config(int model_kind, vp_config* cfg);

/* Number of fp32 values vp_create expects in `weights` for model_kind, and the canonical
 * tensor order: the state-dict order of Final_models/ ** / *.pt.v1 without the
 * num_batches_tracked entries.  vp_param_name/vp_param_size
 * enumerate it (index 0..vp_param_count-1). */
size_t vp_weight_count(int model_kind);
int vp_param_count(int model_kind);
const char* vp_param_name(int model_kind, int index);
size_t vp_param_size(int model_kind, int index);

/* Replaces sbm.<Model>.from_pretrained(name) + model.cuda()  (README.md:46-47,
 * demo.ipynb:224-227): builds the device plan (BN folding, MFMA fragment packing),
 * uploads it and allocates the workspace.  weights_mem says where `weights` lives
 * (VP_MEM_DEVICE after vp_bcast-style distribution). */
int vp_create(int device_id, int model_kind, const float* weights, size_t n_floats, int weights_mem,
              const vp_config* cfg, vp_handle** out);
int vp_destroy(vp_handle* h);

/* Geometry of the model behind the handle. */
int vp_in_samples(const vp_handle* h);   /* 3001 (PhaseNet) / 6000 (EQTransformer) */
int vp_n_outputs(const vp_handle* h);    /* 3: PhaseNet (P,S,N in `phases` order) / EQT (Detection,P,S) */

/* Replaces model(x) under torch.no_grad()/model.eval() on a (B,3,T) float32 tensor
 * (volpick/model/eval_taks0.py:58-61,69,86).  x: B*3*T, y: B*n_out*T, both in the memory
 * space x_mem/y_mem.  preprocess != 0 additionally applies annotate_batch_pre
 * (demean + amplitude normalisation [+ taper]) to each window first.
 * B may exceed max_batch; it is processed in chunks. */
int vp_forward(vp_handle* h, const float* x, int x_mem, int B, int preprocess, float* y, int y_mem);

/* Replaces WaveformModel.annotate on one station block (call
 * stack §3.1): windows of in_samples at stride in_samples-overlap plus one tail window,
 * annotate_batch_pre, forward, NaN blinding, overlap stacking (avg / max).
 * stream: 3*N samples, rows in component_order.  out: n_out*N, NaN where no un-blinded
 * window covers a sample.  first_valid/last_valid (may be NULL): the NaN-trimmed range
 * [first_valid, last_valid] of output row 0; first_valid = -1 if nothing is valid.
 * Returns the number of windows in *n_windows (may be NULL).  N < in_samples yields an
 * all-NaN output and 0 windows (the reference only warns). */
int vp_annotate(vp_handle* h, const float* stream, int stream_mem, int64_t N, int overlap, int blind_l,
                int blind_r, int stacking, int batch, float* out, int out_mem, int64_t* first_valid,
                int64_t* last_valid, int64_t* n_windows);

/* Replaces obspy trigger_onset + per-trigger max/argmax as used by
 * picks_from_annotations / detections_from_annotations and by the reference's own
 * get_picks_from_prob (volpick/model/eval_taks0.py:46-56).  trace: n samples (may hold
 * NaN).  A trigger opens at the first sample > thr_on and closes at the last sample of
 * the run of samples > thr_off.  Writes the earliest (by onset) min(cap, total) triggers in onset order;
 * *n_found is the total. */
int vp_pick(vp_handle* h, const float* trace, int trace_mem, int64_t n, float thr_on, float thr_off,
            int64_t* on, int64_t* off, int64_t* peak, float* value, int cap, int* n_found);

/* Replaces model.classify(stream, ...) on one station block (README.md:54-66): vp_annotate
 * followed by the trigger/peak scan of the output rows named in `specs`, with a single
 * host synchronisation and a single device->host copy of the triggers.  `out` may be NULL
 * (annotations not wanted), a device buffer or a host buffer of n_out*N floats.  Triggers
 * are indices into the N-sample output rows (not the NaN-trimmed traces), grouped by spec in
 * spec order and sorted by onset inside each group; spec_of[i] (may be NULL) is the spec
 * index of trigger i.  Requires thr_off <= thr_on. */
typedef struct {
  int32_t row;     /* output row: PhaseNet 0..2 in `phases` order; EQT 0 Detection, 1 P, 2 S */
  float thr_on;    /* trigger opens at the first sample > thr_on */
  float thr_off;   /* ... and closes at the last sample of the run of samples > thr_off */
} vp_trigger_spec;
int vp_classify(vp_handle* h, const float* stream, int stream_mem, int64_t N, int overlap, int blind_l,
                int blind_r, int stacking, int batch, const vp_trigger_spec* specs, int n_specs, float* out,
                int out_mem, int64_t* first_valid, int64_t* last_valid, int64_t* n_windows, int64_t* on,
                int64_t* off, int64_t* peak, float* value, int32_t* spec_of, int cap, int* n_found);

/* The trigger scan of vp_classify alone: `rows_dev` is a device (n_out, N) array of stacked rows (what vp_annotate left
 * there); every spec's row is scanned in one launch, one synchronisation, one result copy.  Same result layout and cap
 * rule as vp_classify. */
int vp_pick_rows(vp_handle* h, const float* rows_dev, int64_t N, const vp_trigger_spec* specs, int n_specs, int64_t* on,
                 int64_t* off, int64_t* peak, float* value, int32_t* spec_of, int cap, int* n_found);

/* Asynchronous form of vp_classify for callers that keep the GPU fed (several station blocks
 * or bench steps in flight): vp_classify_submit enqueues the whole path on the handle's stream
 * and returns without synchronising; vp_classify_collect(slot) waits for that submit and returns
 * its triggers.  Up to VP_MAX_INFLIGHT submits may be outstanding, each in its own slot; work is
 * executed in submit order.  Buffers passed to submit (stream, out) must stay valid until the
 * matching collect; a device `out` shared by several in-flight submits is overwritten in order. */
#define VP_MAX_INFLIGHT 4
int vp_classify_submit(vp_handle* h, int slot, const float* stream, int stream_mem, int64_t N, int overlap,
                       int blind_l, int blind_r, int stacking, int batch, const vp_trigger_spec* specs, int n_specs,
                       float* out, int out_mem, int cap);
int vp_classify_collect(vp_handle* h, int slot, int64_t* first_valid, int64_t* last_valid, int64_t* n_windows,
                        int64_t* on, int64_t* off, int64_t* peak, float* value, int32_t* spec_of, int cap,
                        int* n_found);

/* Replaces the per-sample loop of the reference's own evaluation protocol
 * (volpick/model/eval_taks0.py:46-56,96-142: get_picks_from_prob on window_borders slices).
 * prob: (B, n_rows, T) model outputs; for window b the samples [lo[b], hi[b]) of channel `row`
 * are scanned (lo/hi NULL = whole window).  Per window up to K triggers are returned, unordered:
 * peak[b*K + i] = argmax index relative to lo[b], value[b*K + i] = the maximum; count[b] is the
 * number found (may exceed K).  The reference calls this with thr_off = thr_on / 2. */
int vp_pick_windows(vp_handle* h, const float* prob, int prob_mem, int B, int n_rows, int row, const int32_t* lo,
                    const int32_t* hi, float thr_on, float thr_off, int K, int32_t* count, int32_t* peak,
                    float* value);

/* vp_classify over K stream blocks (stations / contiguous segments) in ONE call: the windows of all blocks
 * share the forward batches -- SeisBench's batch_size spans every fragment of the stream it is given
 * (/root/reference README.md:54-66) -- and stacking and the trigger scan are one launch each.
 * `streams` (host or device) holds block k as 3 rows of lengths[k] samples at float offset offsets[k];
 * `out` (may be NULL) receives the stacked rows in the same layout.  first_valid / last_valid / n_windows
 * are arrays of K.  Triggers come back grouped by block, then spec, sorted by onset, with block_of /
 * spec_of; every (block, spec) row has room for cap_per_row triggers on the device.  *n_found > cap (or a row
 * with more than cap_per_row triggers) means: call again with more room. */
int vp_classify_multi(vp_handle* h, const float* streams, int stream_mem, const int64_t* offsets,
                      const int64_t* lengths, int K, int overlap, int blind_l, int blind_r, int stacking, int batch,
                      const vp_trigger_spec* specs, int n_specs, float* out, int out_mem, int64_t* first_valid,
                      int64_t* last_valid, int64_t* n_windows, int64_t* on, int64_t* off, int64_t* peak, float* value,
                      int32_t* spec_of, int32_t* block_of, int cap_per_row, int cap, int* n_found);

/* Host-only variant of vp_pick for traces already in host memory (no handle, no GPU). */
int vp_pick_host(const float* trace, int64_t n, float thr_on, float thr_off, int64_t* on, int64_t* off,
                 int64_t* peak, float* value, int cap, int* n_found);

/* Window start rule of the reference.  Writes up to cap starts,
 * returns the count (0 if N < in_samples), negative on bad arguments. */
int64_t vp_window_starts(int64_t N, int in_samples, int overlap, int64_t* starts, int64_t cap);

/* Timing of the last vp_forward / vp_annotate on this handle, measured with HIP events on
 * the handle's stream (only while vp_set_timing(h, 1) is in effect): total milliseconds and the per-stage split
 * (0 preprocess, 1 model forward, 2 blinding+stacking, 3 pick scan). */
int vp_last_timing(const vp_handle* h, float* total_ms, float stage_ms[4]);
/* Stage events are OFF by default (each hipEventRecord opens a ~5 us bubble on the stream);
 * enable them for the calls whose vp_last_timing split is wanted. */
int vp_set_timing(vp_handle* h, int enable);

/* The handle's HIP stream (hipStream_t) for callers that want to order their own work. */
void* vp_stream(const vp_handle* h);
int vp_synchronize(vp_handle* h);

/* Introspection for bench.py: the forward pass is a fixed list of kernel launches
 * ("steps").  vp_step_info gives a step's name and its ALGORITHMIC flop count per window
 * (2*MAC of the layer it implements, not the padded MFMA work); vp_flops_per_window is
 * their sum.  vp_profile_steps runs each launch `iters` times on B windows and reports its
 * mean duration in ms, measured with HIP events on the handle's stream. */
int vp_step_count(const vp_handle* h);
int vp_step_info(const vp_handle* h, int index, const char** name, double* flops_per_window);
double vp_flops_per_window(const vp_handle* h);
/* The work a launch actually issues per window (whole 16-column tiles, channels padded to 4, recomputed halos, the
 * folded taps of the polyphase forms), as fp32-equivalent FLOP. */
int vp_step_issued_flops(const vp_handle* h, int index, double* issued_flops_per_window);
/* The same work by the pipe it is issued to -- what a roofline of the launch has to be priced with: FLOP per window as
 * fp32 MFMAs (v_mfma_f32_16x16x4_f32), as bf16 MFMAs (v_mfma_f32_16x16x32_bf16; an exact three-piece product is SIX of
 * them, all counted) and on the vector ALUs (direct convolutions, recurrences, attention scores).  Filled for every
 * launch of every plan; vp_step_issued_flops = mfma_f32 + mfma_bf16 / 6 + valu (its fp32 equivalent). */
typedef struct {
  double mfma_f32_flop, mfma_bf16_flop, valu_flop;
} vp_issued_work;
int vp_step_issued_work(const vp_handle* h, int index, vp_issued_work* out);
/* A launch whose tiling follows the output range the caller keeps (eqt_tail3_kernel: only the tiles that hold un-blinded
 * samples) issues different work for different ranges.  vp_step_issued_work answers for the range the vp_profile_* calls
 * time (the one the handle's latest preprocessing batch kept); this one for any [out_lo, out_hi) (out_hi <= 0: the whole
 * row, what model(x) computes).  Pure: no launch changes what either reports. */
int vp_step_issued_work_for_range(const vp_handle* h, int index, int out_lo, int out_hi, vp_issued_work* out);
int vp_profile_steps(vp_handle* h, int B, int iters, float* step_ms, int cap);
/* One step timed IN the pipeline: the whole list runs in order `iters` times, only step `index` is bracketed by
 * events (its inputs come from the preceding kernel, as under rocprofv3). */
int vp_profile_step_in_pipeline(vp_handle* h, int B, int iters, int index, float* ms);
/* launch `index` alone, iters times back to back (two handles on two host threads: do two kernels share the chip?) */
int vp_profile_one_step(vp_handle* h, int B, int iters, int index, float* ms);

/* Host-only (no GPU): plans the model and returns conv layer `conv_index`'s geometry
 * (13 ints: cin1,cin2,cout,P,taps,sn,in_off,out_off,waves_m,waves_n,nw,relu,epi), packed
 * MFMA A-fragments and folded bias.  Returns the fragment float count, or -(1000 + number
 * of conv layers) when conv_index is out of range.  Used by the CPU tests of BN folding and
 * fragment packing. */
int vp_debug_plan_conv(int model_kind, const float* weights, size_t n_floats, const vp_config* cfg,
                       int conv_index, int* geom13, float* afrag, size_t afrag_cap, float* bias, size_t bias_cap,
                       const char** name, int* cols, int* l_out);

/* Debug: number of intermediate activation tensors of the last forward pass and a copy of
 * one of them as a dense (B, C, L) host array (used by the layer-by-layer parity tests). */
int vp_debug_tensor_count(const vp_handle* h);
int vp_debug_tensor_info(const vp_handle* h, int index, const char** name, int* channels, int* length);
int vp_debug_tensor_read(vp_handle* h, int index, int B, float* host_out);  /* VP_ERR_UNSUPPORTED: the plan keeps it in LDS */

/* Debug guard of the data-layout invariant the conv loaders rely on: the margins of every
 * activation row -- [0, 8) and [8 + L, row stride) -- are the convolution padding, zeroed once at vp_create and never
 * written again.  Scans every row of every tensor; *n_bad = margin words that are not +-0.0, *first_bad_tensor (may be
 * NULL) the name of the first offending tensor.  Meant to be run after calls on ragged sizes.  self_test = k > 0 plants
 * one stray word in a margin of tensor k - 1 for the duration of the scan (the checker checking itself: expect 1). */
int vp_debug_check_halos(vp_handle* h, int self_test, int64_t* n_bad, const char** first_bad_tensor);

/* Debug (handles created with plan_flags[1] & 2): B x 32 words per window of the fused PhaseNet core
 * kernel: [0..14] shader-clock stamps (kernel start, input loaded, after each of the 13 layers),
 * [16],[17] the 100 MHz wall clock at kernel start / end, [18..22] phase stamps of pn_up3p_kernel. */
int vp_debug_core_clock(vp_handle* h, int B, unsigned long long* out32);
/* Same flag: 8 stamps per conv_mfma_kernel launch (in plan order of the conv layers) of one probe
 * workgroup: start, input staged, MFMA loop done, output staged, stored.  Returns the layer count. */
int vp_debug_conv_clock(vp_handle* h, unsigned long long* out, int max_layers);
/* Same flag, EQTransformer: B x 32 words of eqt_tail_kernel, one row per workgroup: six stamps for each of its first
 * four tiles (tile start, image parked, after stage 4 / 5 / 6, heads done); [24], [25] the shader clock and [30], [31] the
 * 100 MHz wall clock at kernel start / end. */
int vp_debug_tail_clock(vp_handle* h, int B, unsigned long long* out32);

/* ---------------------------------------------------------------------------------------------
 * Waveform-file ingestion: miniSEED 2 and miniSEED 3 records -> sample arrays,
 * the step the reference performs with obspy.read() (libmseed) before stream_to_array
 * (/root/reference volpick/data/convert.py:7,26-70).
 *
 * vp_mseed_scan (host only, no GPU) walks a miniSEED byte string and fills one vp_mseed_record per
 * data record; the two formats may be mixed in one buffer.  miniSEED 2: fixed header, blockette 1000
 * (encoding, word order, record length) and blockette 1001 (microseconds); the header time
 * correction is applied unless activity flag bit 1 says it already was.  miniSEED 3 (FDSN 2020:
 * "MS" 3, 40-byte little-endian fixed header, source identifier "FDSN:NET_STA_LOC_B_S_SS", extra
 * headers, payload): the CRC-32C of every record is verified (VP_ERR_INVALID on a mismatch), start
 * times are truncated from nanoseconds to microseconds, a negative rate field is a period in
 * seconds, the channel is BAND SOURCE SUBSOURCE joined when each is one character; identifiers whose
 * codes do not fit the fields below are VP_ERR_UNSUPPORTED.  Writes at most `cap` records and always
 * reports the total in *n_found.
 *
 * vp_mseed_decode (HIP, one wavefront per record) decodes records into `out`: record r's samples go
 * to out[out_index[r] .. out_index[r] + min(nsamples, out_count[r])), clipped to [0, out_len);
 * out_index[r] < 0 skips the record, out_count may be NULL.  Encodings: 0 text (one sample per byte,
 * its unsigned value), 1 int16, 2 int24, 3 int32, 4 float32, 5 float64, 10 Steim-1, 11 Steim-2,
 * either byte order.  out_kind VP_SAMPLES_INT32 (integer
 * encodings only, exact) or VP_SAMPLES_FLOAT32.  zero_fill != 0 clears `out` first (the zero fill of
 * gaps in stream_to_array).  `status` (host, may be NULL) receives per record 0 = ok,
 * 1 = Steim reverse-integration constant mismatch, 2 = payload shorter than the header's count.
 * buf / out may be host or device memory (buf_mem / out_mem); recs, out_index, out_count, status
 * are host arrays.  buf itself must be 4-byte aligned; payloads may start at any byte (miniSEED 3
 * headers have no fixed length). */
enum { VP_SAMPLES_INT32 = 0, VP_SAMPLES_FLOAT32 = 1 };
typedef struct {
  int64_t offset;      /* byte offset of the record */
  int64_t start_us;    /* first sample, microseconds since 1970-01-01T00:00:00 UTC */
  double sample_rate;  /* Hz */
  int32_t reclen, data_offset, nsamples, encoding;
  int32_t big_endian;  /* word order of the payload (and of a miniSEED 2 header) */
  int32_t quality;     /* miniSEED 2: data quality indicator character; miniSEED 3: 0x300 | publication version */
  char network[4], station[8], location[4], channel[4]; /* NUL terminated, blanks stripped */
} vp_mseed_record;
int vp_mseed_scan(const uint8_t* buf, size_t nbytes, vp_mseed_record* recs, int64_t cap, int64_t* n_found);
int vp_mseed_decode(int device_id, const uint8_t* buf, int buf_mem, size_t nbytes, const vp_mseed_record* recs,
                    const int64_t* out_index, const int64_t* out_count, int64_t n_recs, int out_kind, void* out,
                    int out_mem, int64_t out_len, int zero_fill, int32_t* status);
/* Mean kernel time (ms, HIP events) of `iters` decodes of the same device-resident inputs: bench only. */
int vp_mseed_decode_bench(int device_id, const uint8_t* buf_dev, size_t nbytes, const vp_mseed_record* recs,
                          const int64_t* out_index, int64_t n_recs, int out_kind, void* out_dev, int64_t out_len,
                          int iters, float* ms);
/* vp_mseed_decode keeps its device scratch (file image, sample array of host destinations, record table: ~175 MB after one
 * host-to-host station-day) per device, grow-only, for the life of the process, and serialises the calls on one device from
 * their first use of it to their final stream synchronisation.  This frees it (waits for a call in flight; the next decode
 * allocates again).  bytes_freed may be NULL.  No counterpart in the reference (obspy.read holds no device memory). */
int vp_mseed_release_scratch(int device_id, size_t* bytes_freed);

/* ---------------------------------------------------------------------------------------------
 * PhaseNet training step (BASELINE config 5): what one
 * PhaseNetLit.training_step + Adam optimizer.step of the reference computes
 * (/root/reference volpick/model/models.py:34-51 vector_cross_entropy, :160-164 training_step,
 * :177-185 torch.optim.Adam).  vp_train_create: fp32 throughout (what the reference trains in: no autocast).
 * vp_train_create_dtype(..., VP_TRAIN_BF16, ...): the activation and gradient rows of every layer (z, a, gz, ga -- all
 * of the step's HBM traffic but the weights) rest in memory as bfloat16, round-to-nearest-even on store; weights,
 * weight gradients, BatchNorm statistics, the loss and the Adam state stay fp32 and every product accumulates in
 * fp32 (BASELINE config 5's dtype).  Inputs, labels, predictions and everything vp_train_read returns are fp32 in
 * both modes; vp_train_tensor_read widens.
 *
 * vp_train_create uploads the flat weight blob (same order as vp_create; BatchNorm running
 * statistics included) and sizes the workspace for max_batch windows.  vp_train_step runs
 * forward in training mode (batch statistics; running statistics updated with momentum 0.1),
 * the loss -(1/B) sum_b sum_c mean_t y log(p + 1e-5), backward, and -- if update != 0 -- one Adam
 * step with learning rate lr (the caller owns the schedule, e.g. the reference's 500-step
 * warm-up, models.py:168-175).  x, y: (B, 3, 3001) fp32, host or device (mem).  *loss (may be
 * NULL; non-NULL synchronises) receives the batch loss.  vp_train_read copies out 0 = weights,
 * 1 = gradients of the last step, 2/3 = Adam first/second moments, 4 = EMA weights. */
typedef struct vp_trainer vp_trainer;
#define VP_TRAIN_FP32 0
#define VP_TRAIN_BF16 1
int vp_train_create(int device_id, int model_kind, const float* weights, size_t n_floats, int max_batch,
                    vp_trainer** out);
int vp_train_create_dtype(int device_id, int model_kind, const float* weights, size_t n_floats, int max_batch, int dtype,
                          vp_trainer** out);
int vp_train_dtype(const vp_trainer* t); /* VP_TRAIN_FP32 / VP_TRAIN_BF16 */
int vp_train_destroy(vp_trainer* t);
int vp_train_set_hyper(vp_trainer* t, float beta1, float beta2, float adam_eps, float bn_momentum, float loss_eps);
/* Optional exponential moving average of the weights after every update (the reference's EMA callback,
 * volpick/model/train.py:153-176: decay 0.999, every step); starts at the current weights. */
int vp_train_set_ema(vp_trainer* t, float decay);
int vp_train_step(vp_trainer* t, const float* x, const float* y, int mem, int B, float lr, int update, double* loss);
int vp_train_synchronize(vp_trainer* t);
/* The trainer runs on its own non-blocking stream (vp_train_stream).  A caller that hands vp_train_step DEVICE x / y
 * produced on another stream (a) makes the trainer's stream wait for them before the call (an event of its stream that
 * vp_train_stream waits for) and (b) calls this afterwards: `stream` (hipStream_t, NULL = legacy default stream) then
 * waits, on the device, for the point behind the step's last read of x / y -- so refilling or freeing them on `stream`
 * cannot overtake a queued step.  Host x / y are staged inside vp_train_step and need neither. */
int vp_train_wait_inputs_consumed(vp_trainer* t, void* stream);
/* The number (0 = this trainer's first vp_train_step) of the latest step whose reads of x / y are known to have completed,
 * -1 if none: a caller that keeps device batches alive for queued steps may drop those of steps <= the result.  No host
 * wait, nothing enters the trainer's stream (an event recorded there per step cost ~30 us of the step). */
long long vp_train_inputs_consumed_upto(vp_trainer* t);
long long vp_train_steps_enqueued(const vp_trainer* t);  /* the step a vp_train_step call just queued has the number (this - 1) */
int vp_train_read(vp_trainer* t, int which, float* out, size_t n_floats);
int vp_train_write_weights(vp_trainer* t, const float* weights, size_t n_floats);
int vp_train_predictions(vp_trainer* t, float* out, int B);
/* Debug / parity tests: the z, a, gz, ga tensors of the last step as dense (B, C, L) arrays. */
int vp_train_tensor_count(const vp_trainer* t);
int vp_train_tensor_info(const vp_trainer* t, int index, const char** name, int* channels, int* length);
int vp_train_tensor_read(vp_trainer* t, int index, int B, float* out);
void* vp_train_stream(const vp_trainer* t);
int vp_train_launch_count(const vp_trainer* t); /* kernel launches the latest vp_train_step enqueued */

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU bring-up.  The reference is single-GPU; windows are independent given the
 * weights, so the one exchange is the start-up broadcast of the flat weight blob from the root rank: a single
 * ncclBroadcast over RCCL (xGMI inside a node), after which vp_create(..., VP_MEM_DEVICE, ...) builds the plan from the
 * received buffer.  One process per GPU.  RCCL is bound at run time (dlopen librccl.so.1).
 *
 * vp_rccl_unique_id: the root fills 128 bytes that every rank must receive out of band (a file, a socket,
 *   torch.distributed's store ...).  vp_rccl_comm_init: collective over the n_ranks processes, binds the communicator
 *   to device_id.  vp_bcast_weights: in place -- the root sends weights_dev, every other rank receives into it;
 *   returns when the data has arrived.  vp_rccl_comm_destroy releases the communicator.
 *
 * vp_rccl_available: 1 if RCCL could be bound in this process, 0 otherwise (vp_last_error says why).  A binder checks it
 *   on EVERY rank and agrees on the outcome (e.g. an all-reduce MIN over its own transport) before anyone enters the
 *   collective vp_rccl_comm_init: a rank that cannot take part must not leave the others waiting inside it. */
#define VP_RCCL_UNIQUE_ID_BYTES 128
int vp_rccl_available(void);
int vp_rccl_unique_id(void* id128);
int vp_rccl_comm_init(int device_id, int n_ranks, const void* id128, int rank, void** comm);
int vp_rccl_comm_info(void* comm, int* n_ranks, int* rank); /* ncclCommCount / ncclCommUserRank of the communicator */
int vp_rccl_comm_destroy(void* comm);
int vp_bcast_weights(void* rccl_comm, float* weights_dev, size_t n_floats, int root);
/* Absolute path of the shared object the RCCL entry points above were bound from (dladdr of ncclBroadcast), so that a
 * multi-GPU run can show that the process carries ONE RCCL: PyTorch-ROCm bundles a librccl with the same SONAME
 * (librccl.so.1), and dlopen by SONAME hands back the copy that is already mapped.  Writes a NUL-terminated string of at
 * most cap bytes; VP_ERR_UNSUPPORTED if RCCL could not be bound. */
int vp_rccl_library_path(char* buf, size_t cap);

const char* vp_last_error(void);
const char* vp_version(void);

/* ABI revision of this header.  It changes whenever a struct a caller fills (vp_config, vp_trigger_spec,
 * vp_mseed_record) changes size or layout, or an entry point changes its signature; added entry points do not bump it.
 * A binder compiled against this header checks vp_abi_version() == VP_ABI_VERSION and vp_config_size() ==
 * sizeof(vp_config) once, before vp_create reads its struct.  History: 1 = round 1 (vp_config with reserved[8]),
 * 2 = round 3 (plan_flags[8] + reserved[4]: the struct grew by 16 bytes), 3 = this header (layout of 2, the two
 * checks added). */
#define VP_ABI_VERSION 3
int vp_abi_version(void);
size_t vp_config_size(void);

#ifdef __cplusplus
}
#endif
#endif /* VOLPICK_HIP_H */
