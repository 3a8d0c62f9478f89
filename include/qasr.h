/* qasr.h -- C ABI of libqasr.so: MI355X-native (gfx950) Qwen3-ASR transcribe() hot path.
 *
 * Drop-in boundary for ivan-digital/qwen3-asr-swift.  Every entry point names the reference
 * interface it replaces (file:line under the reference tree).  Plain pointers and sizes only.
 *
 *   Qwen3ASRModel.fromPretrained(modelId:cacheDir:...)   Sources/Qwen3ASR/Qwen3ASR.swift:608-668
 *        -> qasr_default_config + qasr_create (+ qasr_set_tensor / qasr_finalize for in-memory weights)
 *   Qwen3ASRModel.transcribe(audio:sampleRate:language:maxTokens:context:)   Qwen3ASR.swift:131-164
 *   SpeechRecognitionModel.transcribe / inputSampleRate   Sources/AudioCommon/Protocols.swift:151-165,
 *                                                         Sources/Qwen3ASR/Qwen3ASR+Protocols.swift:5-11
 *        -> qasr_transcribe, qasr_input_sample_rate
 *   sc_stt_vtable_t.transcribe callback                   Sources/SpeechCore/VoicePipeline.swift:374-410
 *        -> qasr_stt_vtable (same field order / ownership rules)
 *   ModelMemoryManageable isLoaded/unload/memoryFootprint Sources/Qwen3ASR/Qwen3ASR+Memory.swift:3-18
 *        -> qasr_is_loaded, qasr_unload, qasr_memory_footprint
 *   TranscribeBatchCommand (sequential loop in the reference; batched here)
 *                                                         Sources/AudioCLILib/TranscribeBatchCommand.swift:69-133
 *        -> qasr_batch_begin / qasr_batch_run / qasr_batch_tokens, qasr_transcribe_batch
 *   Parakeet-TDT / Nemotron / Parakeet-EOU (BASELINE configs[4]; networks are opaque CoreML there): MelPreprocessor.extract,
 *   StreamingMelPreprocessor.extract / extractRaw / extractStreaming, TDTGreedyDecoder.decode, RNNTGreedyDecoder.decode,
 *   ParakeetVocabulary / NemotronVocabulary, StreamingSession.pushAudio   (file:line at each declaration)
 *        -> qasr_nemo_mel_*, qasr_tdt_greedy_decode, qasr_rnnt_greedy_decode, qasr_sp_vocab_*, qasr_stream_chunker_*
 *   Stage entry points (no reference counterpart; they expose R1-R8 of SURVEY.md section 8a so
 *   each kernel can be diffed against the oracle in isolation): qasr_mel, qasr_encode,
 *   qasr_prefill_logits, qasr_decode_forced.
 *
 * Threading: like the reference ("not thread-safe", Qwen3ASR.swift:67) one engine = one HIP
 * device + one stream, not re-entrant; use one engine per GPU.
 * Errors: integer status (0 = ok) + qasr_last_error(engine).  transcribe() in the reference never
 * throws (Qwen3ASR.swift:151-154); shims convert a non-zero status to "[qasr error: ...]".
 */
#ifndef QASR_H
#define QASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QASR_OK 0
#define QASR_ERR_INVALID 1      /* bad argument / shape */
#define QASR_ERR_HIP 2          /* HIP runtime failure (message in qasr_last_error) */
#define QASR_ERR_NOT_LOADED 3   /* weights missing / engine unloaded */
#define QASR_ERR_IO 4           /* checkpoint directory / file problem */
#define QASR_ERR_CAPACITY 5     /* batch or clip exceeds the engine's configured capacity */
#define QASR_ERR_EMPTY_AUDIO 6  /* zero-length clip (the reference traps on it) */
#define QASR_ERR_UNSUPPORTED 7  /* input the reference handles with a closed-source dependency (see qasr_split_words) */

#define QASR_DTYPE_F32 0
#define QASR_DTYPE_BF16 1
#define QASR_DTYPE_F16 2
#define QASR_DTYPE_U32 3

typedef struct qasr_engine qasr_engine;

/* Compile-time presets of the reference (Sources/Qwen3ASR/AudioEncoder.swift:28-88,
 * Configuration.swift:47-108) plus engine capacity knobs. */
typedef struct qasr_config {
    /* audio encoder */
    int32_t enc_d_model, enc_heads, enc_ffn, enc_layers, n_mels, enc_out_dim, conv_channels;
    int32_t n_window, n_window_infer;
    float ln_eps;
    /* text decoder */
    int32_t vocab, hidden, dec_layers, heads, kv_heads, head_dim, inter;
    float rms_eps, rope_theta;
    int32_t group_size, bits;          /* MLX affine quantisation of the checkpoint (4 / 8); 16 = bf16 */
    /* special token ids (Qwen3ASR.swift:54-63,181-193) */
    int32_t tok_im_start, tok_im_end, tok_audio_start, tok_audio_end, tok_audio_pad, tok_asr_text;
    int32_t tok_newline, tok_system, tok_user, tok_assistant;
    /* front-end: 2.0 reproduces vDSP_fft_zrip's documented 2x scaling */
    float fft_scale;
    /* engine capacity */
    int32_t device;                    /* HIP device ordinal */
    int32_t max_batch;                 /* clips per batch */
    int32_t max_audio_seconds;         /* longest clip, seconds at 16 kHz (reference cap: 1200) */
    int32_t max_new_tokens;            /* decoder output cap (reference default 448) */
    int32_t max_prompt_extra;          /* room for context / language hint tokens (aligner: the slotted text) in the prompt */
    /* forced aligner head (Qwen3ForcedAligner, ForcedAligner.swift:63-83; Configuration.swift:132-133).
     * classify_num == 0: an ASR engine (tied LM head).  > 0: the checkpoint's `lm_head.{weight,bias}` is a
     * Linear(hidden, classify_num) over timestamp classes and only the qasr_align* entry points run. */
    int32_t classify_num;              /* 5000 for Qwen3-ForcedAligner-0.6B */
    int32_t tok_timestamp;             /* <|timestamp|> = 151705 (Qwen3ASR.swift:62) */
    float timestamp_segment_time;      /* seconds per class, 0.08 */
} qasr_config;

typedef struct qasr_options {
    int32_t max_tokens;                /* <= config.max_new_tokens; 0 -> 448 */
    int32_t ignore_eos;                /* 1: always emit max_tokens tokens (fixed-work benchmarking) */
    const int32_t* context_ids;        /* tokenised context (Qwen3ASR.swift:203-206) or NULL */
    int32_t n_context;
    const int32_t* language_ids;       /* tokenised "language XX" hint (:228-232) or NULL */
    int32_t n_language;
    /* Qwen3DecodingOptions (Qwen3ASR.swift:13-51).  All zero = greedy fast path (isGreedyFastPath, :300-304).
     * Any non-default value selects the reference's slow path (generateSlow): the f32 logits of every row are written each
     * step and pickNextToken (:449-520) runs on them -- on the device by default, on the host (qasr_pick_next_token) with the
     * tuning knob device_sampler = 0; same tokens at temperature 0, same uniform stream (seed) otherwise. */
    float repetition_penalty;          /* 0 or 1.0 = off; HF sign-aware penalty on already generated ids */
    int32_t no_repeat_ngram_size;      /* 0 = off */
    float temperature;                 /* 0 = argmax; > 0 = Gumbel-max sampling */
    uint64_t seed;                     /* sampler seed (the reference uses the system RNG) */
} qasr_options;

typedef struct qasr_result {
    const char* text;                  /* owned by the engine, valid until the next call on it */
    const int32_t* tokens;             /* generated ids incl. the EOS that stopped the loop */
    int32_t n_tokens;
} qasr_result;

/* speech-core STT vtable shapes, field order as built at VoicePipeline.swift:376-409 */
typedef struct sc_transcription_result_t {
    const char* text;
    const char* language;
    float confidence;
    float start_time;
    float end_time;
} sc_transcription_result_t;

typedef struct sc_stt_vtable_t {
    void* context;
    sc_transcription_result_t (*transcribe)(void* ctx, const float* audio, size_t length, int sample_rate);
    int32_t (*input_sample_rate)(void* ctx);
    void* begin_stream;
    void* push_chunk;
    void* flush_stream;
    void* end_stream;
    void* cancel_stream;
} sc_stt_vtable_t;

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* preset: "0.6B" | "1.7B" | a model id such as "aufklarer/Qwen3-ASR-0.6B-MLX-4bit"
 * (size/bits detection of Qwen3ASR.swift:581-601) | "tiny" (test geometry). */
int qasr_default_config(const char* preset, qasr_config* out);
/* model_dir: directory with *.safetensors (+ vocab.json, tokenizer_config.json); NULL = create an
 * empty engine whose tensors arrive through qasr_set_tensor. */
int qasr_create(const char* model_dir, const qasr_config* cfg, qasr_engine** out);
int qasr_set_tensor(qasr_engine* e, const char* name, const void* host_data, int dtype,
                    const int64_t* shape, int ndim);
int qasr_finalize(qasr_engine* e);                      /* checks completeness, builds fused layouts */
int qasr_set_vocab(qasr_engine* e, const int32_t* ids, const char* const* tokens, size_t n);
int qasr_is_loaded(const qasr_engine* e);
int qasr_unload(qasr_engine* e);
/* Weight bytes resident in HBM: the uploaded tensors plus everything qasr_finalize derives from them (fused q|k|v / gate|up,
 * fragment-major decode-step images; for an MLX 4 / 8-bit checkpoint the packed decode images and ONE layer-sized bf16 scratch the
 * prompt pass dequantises into -- no bf16 expansion of the decoder stays resident).  KV caches / workspaces are not counted. */
size_t qasr_memory_footprint(const qasr_engine* e);
void qasr_destroy(qasr_engine* e);
const char* qasr_last_error(const qasr_engine* e);      /* e may be NULL: last create() failure */
int qasr_input_sample_rate(const qasr_engine* e);       /* 16000 */

/* ---- harness input ---------------------------------------------------------------------------- */
/* PCM16 WAV reader (AudioFileLoader.loadWAV, Sources/AudioCommon/AudioFileLoader.swift:70-157, with the bounds
 * checks pinned by Tests/Qwen3ASRTests/SecurityHardeningTests.swift:83-196): first channel, int16 / 32768.
 * *samples is malloc'ed, release with qasr_free.  Pure CPU; malformed files return QASR_ERR_IO. */
int qasr_load_wav(const char* path, float** samples, size_t* n_samples, int* sample_rate);
void qasr_free(void* p);

/* ---- transcribe --------------------------------------------------------------------------- */
int qasr_transcribe(qasr_engine* e, const float* pcm, size_t n, int sample_rate,
                    const qasr_options* opt, qasr_result* out);
/* tokens: [B, max_new_tokens + 1] row-major, row b = lens[b] ids then padding (-1). */
int qasr_transcribe_batch(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B,
                          int sample_rate, const qasr_options* opt, int32_t* tokens, int32_t* lens);
/* detokenise + "<asr_text>" post-strip (Tokenizer.swift:111-142, Qwen3ASR.swift:283-289);
 * returns bytes written (excluding NUL) or -1. */
int qasr_detokenize(qasr_engine* e, const int32_t* tokens, int32_t n, char* buf, size_t cap);
int qasr_stt_vtable(qasr_engine* e, sc_stt_vtable_t* out);
/* BPE encode of a UTF-8 string with the engine's vocab + merges (Qwen3Tokenizer.encode, Tokenizer.swift:195-289):
 * the reference encodes `context` and "language XX" with it (Qwen3ASR.swift:203-206,228-232).  Returns the
 * number of ids written or -1.  qasr_set_merges installs merges.txt content for engines built in memory. */
int qasr_encode_text(qasr_engine* e, const char* utf8, int32_t* ids, int32_t cap);
int qasr_set_merges(qasr_engine* e, const char* merges_txt);
/* pickNextToken (Qwen3ASR.swift:449-520) on a host logits vector; pure CPU function, no engine needed.
 * rng_state: in/out sampler state (only used when temperature > 0), may be NULL. */
int32_t qasr_pick_next_token(const float* logits, int32_t vocab, const int32_t* generated, int32_t n_generated,
                             float repetition_penalty, int32_t no_repeat_ngram_size, float temperature,
                             uint64_t* rng_state);

/* ---- split batch API: H2D / compute / D2H separately timed -------------------------------- */
int qasr_batch_begin(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B,
                     const qasr_options* opt);          /* host -> HBM, builds the batch plan */
int qasr_batch_run(qasr_engine* e);                     /* mel + encoder + prefill + greedy decode, async */
/* Software pipeline over consecutive batches (a serving loop; no reference counterpart -- it loads and transcribes one file at a time):
 * qasr_batch_stage copies the NEXT batch's clips into a second pinned buffer and queues their host -> HBM copy on a copy stream behind
 * the current batch's log-mel (the only reader of the device PCM buffer), so that staging + PCIe run under the current batch's encoder /
 * prompt pass / decode.  Call it after qasr_batch_run of the current batch (QASR_ERR_INVALID before that); qasr_batch_begin_staged then
 * adopts the staged batch (planning only) once the current batch's tokens have been read.  Same results as qasr_batch_begin.  Once its
 * successor is staged a batch cannot be run again (qasr_batch_rewind + qasr_batch_run: QASR_ERR_INVALID): its samples have left the device. */
int qasr_batch_stage(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B);
int qasr_batch_begin_staged(qasr_engine* e, const qasr_options* opt);
/* Re-arm the resident batch (PCM + plans stay in HBM, greedy state is reset on the device) so that
 * qasr_batch_run can be timed repeatedly without host->device traffic.  No reference counterpart. */
int qasr_batch_rewind(qasr_engine* e);
int qasr_batch_sync(qasr_engine* e);                    /* wait for the engine stream */
int qasr_batch_tokens(qasr_engine* e, int32_t* tokens, int32_t* lens);   /* HBM -> host */
/* per-stage device time of the last qasr_batch_run, milliseconds (HIP events on the engine stream):
 * [0] mel [1] encoder [2] prefill [3] decode [4] total; n_steps = decode steps executed. */
int qasr_batch_timings(qasr_engine* e, float ms[5], int32_t* n_steps);
/* Dominant-kernel probe used by bench.py's roofline object: average duration (ms) of kernel
 * `which` over the last run measured with HIP events on the engine stream, plus its launch count
 * and algorithmic bytes per launch.  which: 0 = decode-step weight-streaming GEMV group,
 * 1 = decode attention, 2 = LM head; 3 / 4 = prompt-pass QKV / gate-up GEMM, 5 = prompt attention of one layer (bytes_per_launch
 * then holds FLOPs; causal count for 5).  The probes time what the step launches: where the step runs the q|k|v projection and the attention
 * of a layer as ONE launch (qasr_decode_structure), 1 is that launch (K / V rows + the q|k|v weights) and 0 the three linears left. 
 * 6 / 7: diagnostic phase stamps of the persistent launches (stderr). */
int qasr_kernel_probe(qasr_engine* e, int which, int reps, float* avg_ms, double* bytes_per_launch);
/* How the decode step of the CURRENT batch is launched (csrc/decoder.hip run_decode_step): *fused_qa = 1 when a layer's q|k|v projection and
 * attention are one launch (csrc/dec_qa.hip), *chain = the `chain` knob where it applies, else 0; *launches_per_layer = dependent launches per
 * decoder layer (5 when neither applies).  No reference counterpart. */
int qasr_decode_structure(qasr_engine* e, int* fused_qa, int* chain, int* launches_per_layer);
/* Diagnostic: the MFMA GEMM the encoder / prompt pass / wav2vec2 path are built on, by itself.  out[M][N] (f32, host) =
 * A[M][K] . W[N][K]^T + bias[N] with bf16 operands (host arrays of bf16 bit patterns), f32 accumulation, f32 bias (may be
 * NULL).  form: 0 = 128x128 double-buffered, 1 = 128x128 single LDS buffer, 2 = 256x256 ping-pong (csrc/gemm_p8.h),
 * -1 = whatever the engine would pick for this shape.  K % 8 == 0 and N % 4 == 0 like every caller.  Needs no weights;
 * no reference counterpart (MLX supplies the matmul there) -- used by tests/test_gpu_gemm.py to compare the forms with
 * each other bit for bit and with a float64 product on ragged shapes.  *avg_ms (may be NULL): mean of `reps` launches. */
int qasr_gemm_probe(qasr_engine* e, const uint16_t* A, const uint16_t* W, const float* bias, int M, int N, int K, int form,
                    int reps, float* out, float* avg_ms);

/* ---- utterance-batch data parallelism inside one process ----------------------------------------------------------------------
 * Replaces the sequential file loop of `speech transcribe-batch` (Sources/AudioCLILib/TranscribeBatchCommand.swift:82-93) for a caller
 * that owns several GPUs: one engine (= one HIP device + one stream, weights replicated) and one host thread per listed device; clips
 * are independent (Qwen3ASR.swift:131-164), so device i takes the contiguous block [lo_i, hi_i) of the list (the first B % n devices one
 * clip more) and there is no data-path exchange.  The gather of the decoded token streams is each engine's device -> host copy of its
 * [rows, max_new_tokens + 1] int32 block straight into the caller's buffer.  (The one-process-per-GPU form of the same partition, with
 * the RCCL all_gather of that block over xGMI, is bench.py / qasr/dist.py.)  A device may be listed twice (two engines on one GPU). */
typedef struct qasr_dp qasr_dp;
/* cfg->device is ignored (devices[i] is used); model_dir as for qasr_create (NULL: fill with qasr_dp_set_tensor + qasr_dp_finalize). */
int qasr_dp_create(const char* model_dir, const qasr_config* cfg, const int32_t* devices, int32_t n_devices, qasr_dp** out);
void qasr_dp_destroy(qasr_dp* dp);
int qasr_dp_n_devices(const qasr_dp* dp);
qasr_engine* qasr_dp_engine(qasr_dp* dp, int32_t i);               /* borrowed: engine i (vocabulary, options, stage entry points) */
const char* qasr_dp_last_error(const qasr_dp* dp);                 /* dp may be NULL: last create() failure */
int qasr_dp_set_tensor(qasr_dp* dp, const char* name, const void* host_data, int dtype, const int64_t* shape, int ndim);   /* every engine */
int qasr_dp_finalize(qasr_dp* dp);
/* tokens [B, max_new_tokens + 1], lens [B] as for qasr_transcribe_batch; B may exceed n_devices x max_batch (blocks go through an
 * engine in slices of its capacity).  Results are identical to qasr_transcribe_batch on one engine (batch invariance). */
int qasr_dp_transcribe_batch(qasr_dp* dp, const float* const* pcm, const size_t* n, size_t B, int sample_rate, const qasr_options* opt,
                             int32_t* tokens, int32_t* lens);
int qasr_dp_timings(const qasr_dp* dp, float* ms_per_engine, int32_t cap);    /* wall time of each engine's share of the last call */
/* Whole batches in flight, one per engine: batch k runs on engine k % n_devices as ONE pass (the reference's loop body,
 * TranscribeBatchCommand.swift:82-93, without its "one after the other"); submit returns at once with a ticket, collect waits for that
 * batch and copies tokens [B, max_new_tokens + 1] / lens [B] out.  List a device twice in qasr_dp_create to keep two passes in flight on
 * one GPU: the decode stage is bound by launch latency, so the second pass runs in the first one's gaps (+30 % audio-seconds/sec at
 * 32 x 30 s on one MI355X).  The pcm / n arrays, the samples and the id arrays *opt points to must stay valid until the
 * ticket is collected (*opt itself is copied).  At most one uncollected ticket per engine: a submit whose engine still holds one, a
 * collect of a ticket that is not in flight, and qasr_dp_transcribe_batch while any ticket is in flight return QASR_ERR_INVALID.
 * Tokens are identical to qasr_transcribe_batch's.  submit / collect / destroy are for ONE caller thread (the engines' host threads
 * are the library's own). */
int qasr_dp_submit(qasr_dp* dp, const float* const* pcm, const size_t* n, size_t B, int sample_rate, const qasr_options* opt, int64_t* ticket);
int qasr_dp_collect(qasr_dp* dp, int64_t ticket, int32_t* tokens, int32_t* lens);

/* ---- tuning / diagnostic knobs ----------------------------------------------------------------
 * Process-wide A/B switches between kept kernel variants (csrc/tuning.h lists them with their measurements: "gemm_nbuf",
 * "gemv_splitb", "use_graph", ...).  Defaults are the measured winners; every value passes the parity tests.  Each knob
 * is also seeded once from the environment variable QASR_<KEY IN UPPER CASE>.  No reference counterpart.
 * QASR_ERR_INVALID: unknown key. */
int qasr_set_tuning(const char* key, int value);
/* An engine normally has its GPU to itself while a call runs, and its decode step may then use launches whose workgroups wait for each
 * other inside the launch (csrc/dec_qa.hip: q|k|v projection + attention in one launch; every wait is bounded and ends in QASR_ERR_HIP,
 * never in a hang).  Such a launch needs its whole grid resident at once: an engine that runs concurrently with OTHER engines on the same
 * GPU (qasr_dp_* with a device listed more than once sets this itself) must be marked shared = 1 and then keeps to ordinary launches.
 * No reference counterpart (the reference runs one model instance per process). */
int qasr_set_shared_device(qasr_engine* engine, int shared);
int qasr_get_tuning(const char* key, int* value);

/* ---- stage entry points (oracle diffing) -------------------------------------------------- */
int qasr_num_mel_frames(size_t n_samples);              /* frames handed to the encoder */
int qasr_num_audio_tokens(const qasr_engine* e, int n_frames);
/* out: [n_mels, T] float32 row-major, T = qasr_num_mel_frames(n) (AudioPreprocessing.swift:315-316) */
int qasr_mel(qasr_engine* e, const float* pcm, size_t n, float* out);
/* mel: [n_mels, T] f32 -> out [tokens, enc_out_dim] f32 (bf16 values widened) */
int qasr_encode(qasr_engine* e, const float* mel, int n_frames, float* out);
/* audio_embeds [n_audio, hidden] f32 -> logits [vocab] f32 of the prompt's last position; keeps the
 * KV cache of slot 0 for qasr_decode_forced. */
int qasr_prefill_logits(qasr_engine* e, const float* audio_embeds, int n_audio,
                        const qasr_options* opt, float* logits);
/* teacher-forced steps on slot 0: feeds tokens[i], returns logits [n, vocab]. */
int qasr_decode_forced(qasr_engine* e, const int32_t* tokens, int n, float* logits);

/* ---- forced aligner ----------------------------------------------------------------------
 * Replaces Qwen3ForcedAligner.align / alignLong (Sources/Qwen3ASR/ForcedAligner.swift:226-331, :97-180):
 * mel -> audio encoder -> ONE decoder pass over [chat template + audio + text with <timestamp> slots] (no cache,
 * no autoregression) -> Linear(hidden, classify_num) at the slots -> argmax -> LIS monotonicity fix-up -> seconds.
 * Engines are created from the "aligner-0.6B" preset (encoder = the reference's `.forcedAligner` config); its default
 * capacity max_audio_seconds = 1200 is the reference's own mel cap (AudioPreprocessing.swift:299-313), longer clips
 * return QASR_ERR_CAPACITY. */
typedef struct qasr_aligned_word {     /* AlignedWord (Sources/AudioCommon/Protocols.swift) */
    const char* text;                  /* surface form: the word with its adjacent punctuation */
    float start_time, end_time;        /* seconds */
} qasr_aligned_word;

typedef struct qasr_alignment {        /* owned by the engine, valid until its next align call */
    const qasr_aligned_word* words;
    size_t n_words;
    const int32_t* raw_indices;        /* argmax class per timestamp slot before the fix-up (last pass), 2 per word */
    size_t n_indices;
    int32_t passes;                    /* align passes run (qasr_align: 1; qasr_align_long: 1 + re-alignments) */
} qasr_alignment;

/* TextPreprocessor.splitIntoWordPairs (TextPreprocessing.swift:103-263), default path: whitespace split, one word
 * per Han ideograph, cleaned form keeps Unicode letters / numbers / marks and the ASCII apostrophe.  *surfaces and
 * *cleaned receive '\n'-joined UTF-8 (malloc'ed, release with qasr_free); returns the word count, or
 * -QASR_ERR_UNSUPPORTED for the languages the reference hands to Apple's NLTokenizer (Japanese, Korean, Thai, Lao,
 * Khmer, Burmese, Tibetan: not reproducible) -- pass pre-split words to qasr_align_words for those.  Pure CPU. */
int qasr_split_words(const char* text, const char* language, char** surfaces, char** cleaned);
/* TimestampCorrection.longestIncreasingSubsequencePositions (TimestampCorrection.swift:102-144); returns the count. */
int qasr_lis_positions(const int32_t* values, size_t n, int32_t* positions);
/* TimestampCorrection.enforceMonotonicity (TimestampCorrection.swift:15-99); out has n entries.  Pure CPU. */
int qasr_enforce_monotonicity(const int32_t* raw, size_t n, int32_t* out);
/* Qwen3ForcedAligner.findTrailingPlateauStart (ForcedAligner.swift:196-215) on the words' start times. */
int qasr_find_trailing_plateau(const float* start_times, size_t n, float tolerance, int32_t min_size);
/* TextPreprocessor.prepareForAlignment (TextPreprocessing.swift:48-93) with the engine's tokenizer: token ids with
 * <timestamp> slots around every word, the slot positions inside ids, and the number of words kept.
 * Returns the id count or -1 (capacity / unsupported language / no tokenizer). */
int qasr_align_prepare(qasr_engine* e, const char* text, const char* language, int32_t* ids, int32_t ids_cap,
                       int32_t* ts_positions, int32_t ts_cap, int32_t* n_ts, int32_t* n_words);
/* Stage entry point (oracle diffing): forward for already slotted ids -> raw argmax class per slot; logits
 * (optional) receives [n_ts, classify_num] f32 (bf16 values widened). */
int qasr_align_raw(qasr_engine* e, const float* pcm, size_t n, const int32_t* slotted_ids, int32_t n_ids,
                   const int32_t* ts_positions, int32_t n_ts, int32_t* raw_indices, float* logits);
/* align(audio:text:sampleRate:language:) -- single pass.  16 kHz input (no resampler, as for transcribe). */
int qasr_align(qasr_engine* e, const float* pcm, size_t n, int sample_rate, const char* text, const char* language,
               qasr_alignment* out);
/* the same with caller-split words (surface + cleaned form per word): the NLTokenizer languages. */
int qasr_align_words(qasr_engine* e, const float* pcm, size_t n, int sample_rate, const char* const* surfaces,
                     const char* const* cleaned, size_t n_words, qasr_alignment* out);
/* Batched single-pass align (new: the reference aligns one utterance at a time): B clips with one text each, one
 * mel / encoder / decoder pass over the packed batch; out[b] as for qasr_align (all owned by the engine). */
int qasr_align_batch(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B, int sample_rate,
                     const char* const* texts, const char* language, qasr_alignment* out);
/* alignLong: re-aligns the remainder while a trailing plateau (>= 5 words within 0.1 s) is detected on audio longer
 * than 240 s, at most 10 passes (ForcedAligner.swift:97-180). */
int qasr_align_long(qasr_engine* e, const float* pcm, size_t n, int sample_rate, const char* text,
                    const char* language, qasr_alignment* out);

/* ---- Omnilingual ASR: wav2vec2 encoder + CTC head (BASELINE configs[3]) --------------------------------------------
 * Replaces OmnilingualASRMLXModel (Sources/OmnilingualASR/MLX/OmnilingualMLXModel.swift:20-210) behind the same
 * SpeechRecognitionModel surface (MLX/OmnilingualASRMLXModel+Protocols.swift:3-15): raw 16 kHz samples -> utterance
 * layer-norm -> conv feature extractor -> positional encoder -> N pre-norm transformer layers -> CTC head -> per-frame argmax
 * -> duplicate collapse -> SentencePiece text.  Language-agnostic (the `language` hint is ignored by the reference too);
 * 40 s cap like the reference (:154-159): longer clips return QASR_ERR_CAPACITY.  Batched: clips of a call are independent. */
typedef struct qasr_ctc_engine qasr_ctc_engine;
typedef struct qasr_ctc_config {       /* OmnilingualMLXConfig (MLX/OmnilingualMLXConfig.swift:11-103) + engine capacity */
    int32_t model_dim, layers, heads, ffn_dim;
    int32_t feature_dim;               /* 512: conv feature extractor width; kernels [10,3,3,3,3,2,2], strides [5,2,2,2,2,2,2] */
    int32_t pos_kernel, pos_groups;    /* 128, 16 */
    int32_t vocab;                     /* 10288 */
    int32_t group_size, bits;          /* MLX quantisation of the encoder / head linears: 64, 4 | 8 */
    float ln_eps;                      /* 1e-5 */
    int32_t device, max_batch;
    int32_t max_audio_seconds;         /* <= 40 (the reference's cap) */
} qasr_ctc_config;
/* variant: "300M" | "1B" | "3B" | "7B", a model id such as "aufklarer/Omnilingual-ASR-CTC-1B-MLX-8bit" (detectVariant /
 * detectBits, OmnilingualMLXModel.swift:121-133), or "tiny" (test geometry). */
int qasr_ctc_default_config(const char* variant, qasr_ctc_config* out);
/* model_dir: model.safetensors (+ tokenizer.model) as published (OmnilingualMLXWeightLoader.swift:12-37); NULL = empty
 * engine filled through qasr_ctc_set_tensor (fairseq2 tensor names, PyTorch Conv1d layout, weight_g / weight_v). */
int qasr_ctc_create(const char* model_dir, const qasr_ctc_config* cfg, qasr_ctc_engine** out);
int qasr_ctc_set_tensor(qasr_ctc_engine* e, const char* name, const void* host_data, int dtype, const int64_t* shape, int ndim);
int qasr_ctc_finalize(qasr_ctc_engine* e);
/* SentencePiece vocabulary: piece texts + types (1 normal, 2 unknown, 3 control, 4 user, 5 unused, 6 byte) */
int qasr_ctc_set_pieces(qasr_ctc_engine* e, const char* const* texts, const int32_t* types, size_t n);
int qasr_ctc_is_loaded(const qasr_ctc_engine* e);
int qasr_ctc_unload(qasr_ctc_engine* e);
size_t qasr_ctc_memory_footprint(const qasr_ctc_engine* e);
void qasr_ctc_destroy(qasr_ctc_engine* e);
const char* qasr_ctc_last_error(const qasr_ctc_engine* e);
/* encoder frames for n samples (Wav2Vec2FeatureExtractor.outputLength, Wav2Vec2Frontend.swift:47-54): 160000 -> 499 */
int qasr_ctc_num_frames(size_t n_samples);
/* ids: [B][stride] collapsed token ids of every clip (argmax per frame, consecutive duplicates removed, blank kept:
 * OmnilingualMLXModel.swift:183-188), lens[b] = count; stride >= qasr_ctc_num_frames(longest clip). */
int qasr_ctc_transcribe_batch(qasr_ctc_engine* e, const float* const* pcm, const size_t* n, size_t B, int sample_rate,
                              int32_t* ids, size_t stride, int32_t* lens);
/* transcribe(audio:sampleRate:language:) -> text owned by the engine until its next call; "" for an empty clip */
int qasr_ctc_transcribe(qasr_ctc_engine* e, const float* pcm, size_t n, int sample_rate, const char** text);
/* stage entry point (oracle diffing): logits [frames][vocab] f32 of one clip */
int qasr_ctc_logits(qasr_ctc_engine* e, const float* pcm, size_t n, float* logits);
int qasr_ctc_detokenize(qasr_ctc_engine* e, const int32_t* ids, int32_t n, char* buf, size_t cap);
/* device time of the last call, ms: [0] feature extractor + positional encoder [1] transformer [2] head + argmax [3] total */
int qasr_ctc_timings(qasr_ctc_engine* e, float ms[4]);
/* pure CPU: CTCGreedyDecoder.decode (CTCGreedyDecoder.swift:28-55) -> count; OmnilingualASRModel.layerNormalize
 * (OmnilingualASR.swift:305-325) */
int qasr_ctc_greedy(const float* logits, int32_t T, int32_t V, int32_t valid_frames, int32_t* out);
int qasr_layer_normalize(const float* x, size_t n, float eps, float* out);

/* ---- Parakeet-TDT / Nemotron streaming / Parakeet-EOU (BASELINE configs[4]): the restatable slice ------------------------------
 * In the reference the FastConformer encoder, the LSTM prediction network and the joint of these models are opaque CoreML bundles
 * (`encoder.mlmodelc`, `decoder.mlmodelc`, `joint.mlmodelc`): their arithmetic is not in its tree and is NOT rebuilt here.  What the
 * reference does in Swift around them is: the log-mel front-end (CPU, Accelerate), the greedy transducer loops, vocabulary decode and
 * the streaming session's chunk cutting.  Those are the entry points below; the networks are caller-supplied callbacks. */

/* log-mel front-ends, batched on the device (csrc/nemo_mel.hip).  One launch serves every stream's chunk (64 concurrent streams x one
 * 160 ms chunk replay one captured hipGraph: H2D, three kernels, D2H) where the reference makes one Accelerate call per chunk. */
#define QASR_NEMO_MEL_TDT 0            /* MelPreprocessor.extract, Sources/ParakeetASR/MelPreprocessor.swift:52-202 (float16 values) */
#define QASR_NEMO_MEL_EOU 1            /* StreamingMelPreprocessor.extract, Sources/ParakeetStreamingASR/StreamingMelPreprocessor.swift:62-186 */
#define QASR_NEMO_MEL_RAW 2            /* extractRaw, Sources/NemotronStreamingASR/StreamingMelPreprocessor.swift:55-129 (= ParakeetStreamingASR :193-273) */
#define QASR_NEMO_MEL_EOU_STREAMING 3  /* extractStreaming (running mean / std per stream), ParakeetStreamingASR/StreamingMelPreprocessor.swift:280-393 */
typedef struct qasr_nemo_mel qasr_nemo_mel;
/* fft_scale: 2.0 = vDSP_fft_zrip's scaling as the reference states it (NemotronStreamingASR/StreamingMelPreprocessor.swift:100); only
 * the normalised variants depend on it (through the 2^-24 log guard), RAW divides it out. */
int qasr_nemo_mel_create(int device, int max_streams, size_t max_samples, float fft_scale, qasr_nemo_mel** out);
void qasr_nemo_mel_destroy(qasr_nemo_mel* m);
const char* qasr_nemo_mel_last_error(const qasr_nemo_mel* m);      /* m may be NULL: last create() failure */
int qasr_nemo_mel_num_frames(size_t n_samples);                     /* n / 160 + 1 */
int qasr_nemo_mel_length(size_t n_samples);                         /* melLength = n / 160 */
/* B clips or chunks -> out [B][128][stride] float32 (host), mel_len[b].  fit > 0: every row is cut / zero-padded to `fit` frames
 * (StreamingSession.truncateMel / padMel, Sources/NemotronStreamingASR/StreamingSession.swift:245-274); fit <= 0: nFrames of the
 * longest row.  A zero-length row gives zeros and mel_len 0 (the streaming variants' `guard !audio.isEmpty`); the reflect-padded
 * variants need more than 256 samples (the Swift code indexes out of bounds below that): QASR_ERR_INVALID.  stream_ids: the running-
 * statistics slot of each row for EOU_STREAMING (NULL = row index), ignored otherwise. */
int qasr_nemo_mel_extract(qasr_nemo_mel* m, int variant, const float* const* pcm, const size_t* n, size_t B, const int32_t* stream_ids,
                          float* out, size_t stride, int32_t* mel_len, int fit);
int qasr_nemo_mel_reset_stats(qasr_nemo_mel* m, int stream);        /* resetRunningStats; stream < 0: every stream */
/* device time of the last extract in ms (HIP events: H2D + kernels + D2H) and whether it replayed the captured graph */
int qasr_nemo_mel_timing(const qasr_nemo_mel* m, float* ms, int* was_graph);

/* transducer greedy loops (pure CPU).  The caller owns the networks and their state:
 *   decoder_step(ctx, token)  advance the prediction network with `token` (the loops prime it with the blank id where the reference does)
 *   joint(ctx, frame, token_logits[vocab_size + 1], duration_logits[n_durations] or NULL)  logits for encoder frame `frame` and the
 *        current prediction-network output (the CoreML outputs are float16; pass them widened)
 * Both return 0 on success; any other value aborts the decode with QASR_ERR_INVALID. */
typedef struct qasr_transducer_config {
    int32_t vocab_size;                /* 8192 Parakeet-TDT | 1024 Nemotron | 1026 Parakeet-EOU (Configuration.swift of each target) */
    int32_t blank_id;                  /* = vocab_size */
    int32_t eou_id;                    /* 1024 for Parakeet-EOU, -1 otherwise */
    int32_t n_durations;               /* TDT: 5; RNNT: 0 */
    int32_t durations[8];              /* TDT: 0 1 2 3 4 */
    int32_t first_text_id;             /* TDT: 274 (ids below are fed to the network but not reported, TDTGreedyDecoder.swift:91-94) */
    int32_t max_symbols;               /* RNNT: 10 symbols per encoder frame (RNNTGreedyDecoder.swift:35) */
} qasr_transducer_config;
typedef struct qasr_transducer_callbacks {
    void* ctx;
    int (*decoder_step)(void* ctx, int32_t token);
    int (*joint)(void* ctx, int32_t frame, float* token_logits, float* duration_logits);
} qasr_transducer_callbacks;
/* model: "parakeet-tdt" | "nemotron-streaming" | "parakeet-eou" (or the reference's model ids containing those families' names) */
int qasr_transducer_default_config(const char* model, qasr_transducer_config* out);
/* TDTGreedyDecoder.decode (Sources/ParakeetASR/TDTGreedyDecoder.swift:45-143) -> number of tokens (<= cap) or -status.
 * log_probs / confidence may be NULL. */
int qasr_tdt_greedy_decode(const qasr_transducer_config* cfg, const qasr_transducer_callbacks* cb, int32_t encoded_length,
                           int32_t* tokens, float* log_probs, int32_t cap, float* confidence);
/* RNNTGreedyDecoder.decode (Sources/NemotronStreamingASR/RNNTGreedyDecoder.swift:38-90; with cfg->eou_id >= 0:
 * Sources/ParakeetStreamingASR/RNNTGreedyDecoder.swift:58-126).  The prediction network is NOT primed here: a session primes it once
 * (StreamingSession.swift:93-99) and its state persists across chunks. */
int qasr_rnnt_greedy_decode(const qasr_transducer_config* cfg, const qasr_transducer_callbacks* cb, int32_t encoded_length,
                            int32_t frame_offset, int32_t* tokens, float* log_probs, int32_t cap, int32_t* eou_detected);
float qasr_log_softmax_at(const float* logits, int32_t n, int32_t id);          /* TDTGreedyDecoder.logSoftmax (:149-172) */
float qasr_transducer_confidence(const float* log_probs, int32_t n);           /* min(1, exp(mean log-prob)), 0 when n == 0 */

/* SentencePiece-style vocabularies of these models (vocab.json: {"0": "\u2581the", ...}).  style 0: ParakeetVocabulary
 * (Sources/ParakeetASR/Vocabulary.swift:42-96); style 1: NemotronVocabulary = ParakeetEOUVocabulary
 * (Sources/NemotronStreamingASR/Vocabulary.swift:31-77).  Pure CPU. */
typedef struct qasr_sp_vocab qasr_sp_vocab;
int qasr_sp_vocab_create(const int32_t* ids, const char* const* pieces, size_t n, int style, qasr_sp_vocab** out);
int qasr_sp_vocab_load(const char* vocab_json_path, int style, qasr_sp_vocab** out);
void qasr_sp_vocab_destroy(qasr_sp_vocab* v);
int qasr_sp_vocab_count(const qasr_sp_vocab* v);
int qasr_sp_vocab_decode(const qasr_sp_vocab* v, const int32_t* ids, int32_t n, char* buf, size_t cap);        /* bytes written or -1 */
/* decodeWords: words '\n'-joined into buf, one confidence per word; returns the word count or -1 (buffer / conf_cap too small) */
int qasr_sp_vocab_decode_words(const qasr_sp_vocab* v, const int32_t* ids, int32_t n_ids, const float* log_probs, int32_t n_log_probs,
                               char* buf, size_t cap, float* confidences, int32_t conf_cap);

/* sample bookkeeping of a streaming session (StreamingSession.pushAudio / finalize, Sources/NemotronStreamingASR/StreamingSession.swift:
 * 110-139): whenever samples_per_chunk samples are buffered one chunk is cut and the buffer advances by `shift` samples
 * (Nemotron 160 ms: 17 x 160 = 2720 and 2 x 8 x 160 = 2560; Parakeet-EOU 320 ms: 33 x 160 = 5280 and 4 x 8 x 160 = 5120).  Pure CPU. */
typedef struct qasr_stream_chunker qasr_stream_chunker;
int qasr_stream_chunker_create(int32_t samples_per_chunk, int32_t shift, qasr_stream_chunker** out);
void qasr_stream_chunker_destroy(qasr_stream_chunker* c);
int qasr_stream_chunker_push(qasr_stream_chunker* c, const float* samples, size_t n);
int qasr_stream_chunker_pop(qasr_stream_chunker* c, float* chunk);       /* 1: chunk[samples_per_chunk] filled; 0: not enough samples */
int qasr_stream_chunker_flush(qasr_stream_chunker* c, float* chunk);     /* finalize: 1: the zero-padded remainder; 0: buffer was empty */
size_t qasr_stream_chunker_buffered(const qasr_stream_chunker* c);

#ifdef __cplusplus
}
#endif
#endif /* QASR_H */
