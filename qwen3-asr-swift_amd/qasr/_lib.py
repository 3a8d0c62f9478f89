"""ctypes binding of libqasr.so (the C ABI in include/qasr.h).  No torch types cross it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libqasr.so")


class QasrConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("enc_d_model", "enc_heads", "enc_ffn", "enc_layers", "n_mels",
                                         "enc_out_dim", "conv_channels", "n_window", "n_window_infer")] + \
               [("ln_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("vocab", "hidden", "dec_layers", "heads", "kv_heads", "head_dim", "inter")] + \
               [("rms_eps", C.c_float), ("rope_theta", C.c_float), ("group_size", C.c_int32), ("bits", C.c_int32)] + \
               [(n, C.c_int32) for n in ("tok_im_start", "tok_im_end", "tok_audio_start", "tok_audio_end",
                                         "tok_audio_pad", "tok_asr_text", "tok_newline", "tok_system",
                                         "tok_user", "tok_assistant")] + \
               [("fft_scale", C.c_float)] + \
               [(n, C.c_int32) for n in ("device", "max_batch", "max_audio_seconds", "max_new_tokens",
                                         "max_prompt_extra", "classify_num", "tok_timestamp")] + \
               [("timestamp_segment_time", C.c_float)]


class QasrCtcConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("model_dim", "layers", "heads", "ffn_dim", "feature_dim", "pos_kernel", "pos_groups",
                                         "vocab", "group_size", "bits")] + \
               [("ln_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("device", "max_batch", "max_audio_seconds")]


class QasrOptions(C.Structure):
    _fields_ = [("max_tokens", C.c_int32), ("ignore_eos", C.c_int32),
                ("context_ids", C.POINTER(C.c_int32)), ("n_context", C.c_int32),
                ("language_ids", C.POINTER(C.c_int32)), ("n_language", C.c_int32),
                ("repetition_penalty", C.c_float), ("no_repeat_ngram_size", C.c_int32),
                ("temperature", C.c_float), ("seed", C.c_uint64)]


class QasrResult(C.Structure):
    _fields_ = [("text", C.c_char_p), ("tokens", C.POINTER(C.c_int32)), ("n_tokens", C.c_int32)]


class ScTranscriptionResult(C.Structure):
    _fields_ = [("text", C.c_char_p), ("language", C.c_char_p), ("confidence", C.c_float),
                ("start_time", C.c_float), ("end_time", C.c_float)]


class QasrAlignedWord(C.Structure):
    _fields_ = [("text", C.c_char_p), ("start_time", C.c_float), ("end_time", C.c_float)]


class QasrAlignment(C.Structure):
    _fields_ = [("words", C.POINTER(QasrAlignedWord)), ("n_words", C.c_size_t),
                ("raw_indices", C.POINTER(C.c_int32)), ("n_indices", C.c_size_t), ("passes", C.c_int32)]


class QasrTransducerConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("blank_id", C.c_int32), ("eou_id", C.c_int32), ("n_durations", C.c_int32),
                ("durations", C.c_int32 * 8), ("first_text_id", C.c_int32), ("max_symbols", C.c_int32)]


TD_DECODER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32)
TD_JOINT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))


class QasrTransducerCallbacks(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("decoder_step", TD_DECODER_FN), ("joint", TD_JOINT_FN)]


SC_TRANSCRIBE_FN = C.CFUNCTYPE(ScTranscriptionResult, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int)
SC_RATE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p)


class ScSttVtable(C.Structure):
    _fields_ = [("context", C.c_void_p), ("transcribe", SC_TRANSCRIBE_FN), ("input_sample_rate", SC_RATE_FN),
                ("begin_stream", C.c_void_p), ("push_chunk", C.c_void_p), ("flush_stream", C.c_void_p),
                ("end_stream", C.c_void_p), ("cancel_stream", C.c_void_p)]


_P = C.POINTER
_F = _P(C.c_float)
_I = _P(C.c_int32)
_E = C.c_void_p

# name -> (restype, argtypes); mirrors include/qasr.h one to one
SIGNATURES = {
    "qasr_default_config": (C.c_int, [C.c_char_p, _P(QasrConfig)]),
    "qasr_create": (C.c_int, [C.c_char_p, _P(QasrConfig), _P(_E)]),
    "qasr_set_tensor": (C.c_int, [_E, C.c_char_p, C.c_void_p, C.c_int, _P(C.c_int64), C.c_int]),
    "qasr_finalize": (C.c_int, [_E]),
    "qasr_set_vocab": (C.c_int, [_E, _I, _P(C.c_char_p), C.c_size_t]),
    "qasr_is_loaded": (C.c_int, [_E]),
    "qasr_unload": (C.c_int, [_E]),
    "qasr_memory_footprint": (C.c_size_t, [_E]),
    "qasr_destroy": (None, [_E]),
    "qasr_last_error": (C.c_char_p, [_E]),
    "qasr_input_sample_rate": (C.c_int, [_E]),
    "qasr_load_wav": (C.c_int, [C.c_char_p, _P(_F), _P(C.c_size_t), _P(C.c_int)]),
    "qasr_free": (None, [C.c_void_p]),
    "qasr_transcribe": (C.c_int, [_E, _F, C.c_size_t, C.c_int, _P(QasrOptions), _P(QasrResult)]),
    "qasr_transcribe_batch": (C.c_int, [_E, _P(_F), _P(C.c_size_t), C.c_size_t, C.c_int, _P(QasrOptions), _I, _I]),
    "qasr_detokenize": (C.c_int, [_E, _I, C.c_int32, C.c_char_p, C.c_size_t]),
    "qasr_stt_vtable": (C.c_int, [_E, _P(ScSttVtable)]),
    "qasr_encode_text": (C.c_int, [_E, C.c_char_p, _I, C.c_int32]),
    "qasr_set_merges": (C.c_int, [_E, C.c_char_p]),
    "qasr_pick_next_token": (C.c_int32, [_F, C.c_int32, _I, C.c_int32, C.c_float, C.c_int32, C.c_float, _P(C.c_uint64)]),
    "qasr_batch_begin": (C.c_int, [_E, _P(_F), _P(C.c_size_t), C.c_size_t, _P(QasrOptions)]),
    "qasr_batch_stage": (C.c_int, [_E, _P(_F), _P(C.c_size_t), C.c_size_t]),
    "qasr_batch_begin_staged": (C.c_int, [_E, _P(QasrOptions)]),
    "qasr_batch_run": (C.c_int, [_E]),
    "qasr_batch_rewind": (C.c_int, [_E]),
    "qasr_batch_sync": (C.c_int, [_E]),
    "qasr_batch_tokens": (C.c_int, [_E, _I, _I]),
    "qasr_batch_timings": (C.c_int, [_E, _F, _I]),
    "qasr_kernel_probe": (C.c_int, [_E, C.c_int, C.c_int, _F, _P(C.c_double)]),
    "qasr_gemm_probe": (C.c_int, [_E, _P(C.c_uint16), _P(C.c_uint16), _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F, _F]),
    "qasr_set_shared_device": (C.c_int, [_E, C.c_int]),
    "qasr_decode_structure": (C.c_int, [_E, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "qasr_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "qasr_get_tuning": (C.c_int, [C.c_char_p, _P(C.c_int)]),
    "qasr_num_mel_frames": (C.c_int, [C.c_size_t]),
    "qasr_num_audio_tokens": (C.c_int, [_E, C.c_int]),
    "qasr_mel": (C.c_int, [_E, _F, C.c_size_t, _F]),
    "qasr_encode": (C.c_int, [_E, _F, C.c_int, _F]),
    "qasr_prefill_logits": (C.c_int, [_E, _F, C.c_int, _P(QasrOptions), _F]),
    "qasr_decode_forced": (C.c_int, [_E, _I, C.c_int, _F]),
    "qasr_split_words": (C.c_int, [C.c_char_p, C.c_char_p, _P(C.c_void_p), _P(C.c_void_p)]),
    "qasr_lis_positions": (C.c_int, [_I, C.c_size_t, _I]),
    "qasr_enforce_monotonicity": (C.c_int, [_I, C.c_size_t, _I]),
    "qasr_find_trailing_plateau": (C.c_int, [_F, C.c_size_t, C.c_float, C.c_int32]),
    "qasr_align_prepare": (C.c_int, [_E, C.c_char_p, C.c_char_p, _I, C.c_int32, _I, C.c_int32, _I, _I]),
    "qasr_align_raw": (C.c_int, [_E, _F, C.c_size_t, _I, C.c_int32, _I, C.c_int32, _I, _F]),
    "qasr_align": (C.c_int, [_E, _F, C.c_size_t, C.c_int, C.c_char_p, C.c_char_p, _P(QasrAlignment)]),
    "qasr_align_words": (C.c_int, [_E, _F, C.c_size_t, C.c_int, _P(C.c_char_p), _P(C.c_char_p), C.c_size_t, _P(QasrAlignment)]),
    "qasr_align_batch": (C.c_int, [_E, _P(_F), _P(C.c_size_t), C.c_size_t, C.c_int, _P(C.c_char_p), C.c_char_p, _P(QasrAlignment)]),
    "qasr_align_long": (C.c_int, [_E, _F, C.c_size_t, C.c_int, C.c_char_p, C.c_char_p, _P(QasrAlignment)]),
    "qasr_ctc_default_config": (C.c_int, [C.c_char_p, _P(QasrCtcConfig)]),
    "qasr_ctc_create": (C.c_int, [C.c_char_p, _P(QasrCtcConfig), _P(_E)]),
    "qasr_ctc_set_tensor": (C.c_int, [_E, C.c_char_p, C.c_void_p, C.c_int, _P(C.c_int64), C.c_int]),
    "qasr_ctc_finalize": (C.c_int, [_E]),
    "qasr_ctc_set_pieces": (C.c_int, [_E, _P(C.c_char_p), _I, C.c_size_t]),
    "qasr_ctc_is_loaded": (C.c_int, [_E]),
    "qasr_ctc_unload": (C.c_int, [_E]),
    "qasr_ctc_memory_footprint": (C.c_size_t, [_E]),
    "qasr_ctc_destroy": (None, [_E]),
    "qasr_ctc_last_error": (C.c_char_p, [_E]),
    "qasr_ctc_num_frames": (C.c_int, [C.c_size_t]),
    "qasr_ctc_transcribe_batch": (C.c_int, [_E, _P(_F), _P(C.c_size_t), C.c_size_t, C.c_int, _I, C.c_size_t, _I]),
    "qasr_ctc_transcribe": (C.c_int, [_E, _F, C.c_size_t, C.c_int, _P(C.c_char_p)]),
    "qasr_ctc_logits": (C.c_int, [_E, _F, C.c_size_t, _F]),
    "qasr_ctc_detokenize": (C.c_int, [_E, _I, C.c_int32, C.c_char_p, C.c_size_t]),
    "qasr_ctc_timings": (C.c_int, [_E, _F]),
    "qasr_ctc_greedy": (C.c_int, [_F, C.c_int32, C.c_int32, C.c_int32, _I]),
    "qasr_layer_normalize": (C.c_int, [_F, C.c_size_t, C.c_float, _F]),
    "qasr_dp_create": (C.c_int, [C.c_char_p, _P(QasrConfig), _I, C.c_int32, _P(_E)]),
    "qasr_dp_destroy": (None, [_E]),
    "qasr_dp_n_devices": (C.c_int, [_E]),
    "qasr_dp_engine": (_E, [_E, C.c_int32]),
    "qasr_dp_last_error": (C.c_char_p, [_E]),
    "qasr_dp_set_tensor": (C.c_int, [_E, C.c_char_p, C.c_void_p, C.c_int, _P(C.c_int64), C.c_int]),
    "qasr_dp_finalize": (C.c_int, [_E]),
    "qasr_dp_transcribe_batch": (C.c_int, [_E, _P(_F), _P(C.c_size_t), C.c_size_t, C.c_int, _P(QasrOptions), _I, _I]),
    "qasr_dp_timings": (C.c_int, [_E, _F, C.c_int32]),
    "qasr_dp_submit": (C.c_int, [_E, _P(_F), _P(C.c_size_t), C.c_size_t, C.c_int, _P(QasrOptions), _P(C.c_int64)]),
    "qasr_dp_collect": (C.c_int, [_E, C.c_int64, _I, _I]),
    "qasr_nemo_mel_create": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_float, _P(_E)]),
    "qasr_nemo_mel_destroy": (None, [_E]),
    "qasr_nemo_mel_last_error": (C.c_char_p, [_E]),
    "qasr_nemo_mel_num_frames": (C.c_int, [C.c_size_t]),
    "qasr_nemo_mel_length": (C.c_int, [C.c_size_t]),
    "qasr_nemo_mel_extract": (C.c_int, [_E, C.c_int, _P(_F), _P(C.c_size_t), C.c_size_t, _I, _F, C.c_size_t, _I, C.c_int]),
    "qasr_nemo_mel_reset_stats": (C.c_int, [_E, C.c_int]),
    "qasr_nemo_mel_timing": (C.c_int, [_E, _F, _P(C.c_int)]),
    "qasr_transducer_default_config": (C.c_int, [C.c_char_p, _P(QasrTransducerConfig)]),
    "qasr_tdt_greedy_decode": (C.c_int, [_P(QasrTransducerConfig), _P(QasrTransducerCallbacks), C.c_int32, _I, _F, C.c_int32, _F]),
    "qasr_rnnt_greedy_decode": (C.c_int, [_P(QasrTransducerConfig), _P(QasrTransducerCallbacks), C.c_int32, C.c_int32, _I, _F, C.c_int32, _I]),
    "qasr_log_softmax_at": (C.c_float, [_F, C.c_int32, C.c_int32]),
    "qasr_transducer_confidence": (C.c_float, [_F, C.c_int32]),
    "qasr_sp_vocab_create": (C.c_int, [_I, _P(C.c_char_p), C.c_size_t, C.c_int, _P(_E)]),
    "qasr_sp_vocab_load": (C.c_int, [C.c_char_p, C.c_int, _P(_E)]),
    "qasr_sp_vocab_destroy": (None, [_E]),
    "qasr_sp_vocab_count": (C.c_int, [_E]),
    "qasr_sp_vocab_decode": (C.c_int, [_E, _I, C.c_int32, C.c_char_p, C.c_size_t]),
    "qasr_sp_vocab_decode_words": (C.c_int, [_E, _I, C.c_int32, _F, C.c_int32, C.c_char_p, C.c_size_t, _F, C.c_int32]),
    "qasr_stream_chunker_create": (C.c_int, [C.c_int32, C.c_int32, _P(_E)]),
    "qasr_stream_chunker_destroy": (None, [_E]),
    "qasr_stream_chunker_push": (C.c_int, [_E, _F, C.c_size_t]),
    "qasr_stream_chunker_pop": (C.c_int, [_E, _F]),
    "qasr_stream_chunker_flush": (C.c_int, [_E, _F]),
    "qasr_stream_chunker_buffered": (C.c_size_t, [_E]),
}

_lib = None


def load(strict=True):
    """dlopen libqasr.so; raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"libqasr.so not built ({LIB_PATH}); run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise RuntimeError(f"libqasr.so lacks symbols declared in include/qasr.h: {missing}")
    _lib = lib
    return lib
