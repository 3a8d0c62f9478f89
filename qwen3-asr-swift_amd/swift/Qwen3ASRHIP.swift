// Qwen3ASRHIP.swift -- reference-side binding of libqasr.so (include/qasr.h).
//
// NOT COMPILED in this repository (no Swift toolchain in the build image; see DESIGN.md section 1).
// Written against include/qasr.h; drop into the reference as Sources/Qwen3ASRHIP/ with a
// `CQasr` system-library target whose module map exposes qasr.h and links -lqasr.
//
// It gives the reference's callers the same surface as `Qwen3ASRModel`
// (Sources/Qwen3ASR/Qwen3ASR.swift:68-164, Qwen3ASR+Protocols.swift:5-11, Qwen3ASR+Memory.swift:3-18),
// so `SpeechRecognitionModel` consumers (CLI, StreamingASR, VoicePipeline) work unchanged.
import Foundation
import AudioCommon
import CQasr

public final class Qwen3ASRHIPModel: SpeechRecognitionModel, ModelMemoryManageable {
    private var engine: OpaquePointer?
    private let tokenizer: Qwen3Tokenizer?            // host-side BPE encode of language / context hints
    public var inputSampleRate: Int { 16000 }

    /// Mirrors `Qwen3ASRModel.fromPretrained(modelId:cacheDir:offlineMode:progressHandler:)`
    /// (Qwen3ASR.swift:608-668): the download step stays in Swift (HuggingFaceDownloader), the engine
    /// is created from the local cache directory.
    public static func fromPretrained(
        modelId: String = "aufklarer/Qwen3-ASR-0.6B-MLX-4bit",
        cacheDir: URL? = nil,
        offlineMode: Bool = false,
        device: Int32 = 0,
        maxBatch: Int32 = 32
    ) async throws -> Qwen3ASRHIPModel {
        let dir = try cacheDir ?? HuggingFaceDownloader.getCacheDirectory(for: modelId)
        try await HuggingFaceDownloader.downloadWeights(
            modelId: modelId, to: dir,
            additionalFiles: ["vocab.json", "merges.txt", "tokenizer_config.json"],
            offlineMode: offlineMode, progressHandler: { _ in })
        var cfg = qasr_config()
        guard qasr_default_config(modelId, &cfg) == QASR_OK else {
            throw AudioModelError.modelLoadFailed(modelId: modelId, reason: "unknown preset")
        }
        cfg.device = device
        cfg.max_batch = maxBatch
        var handle: OpaquePointer?
        let rc = qasr_create(dir.path, &cfg, &handle)
        guard rc == QASR_OK, let h = handle else {
            throw AudioModelError.modelLoadFailed(modelId: modelId, reason: String(cString: qasr_last_error(nil)))
        }
        var tok: Qwen3Tokenizer? = Qwen3Tokenizer()
        do { try tok?.load(from: dir.appendingPathComponent("vocab.json")) } catch { tok = nil }
        return Qwen3ASRHIPModel(engine: h, tokenizer: tok)
    }

    private init(engine: OpaquePointer, tokenizer: Qwen3Tokenizer?) {
        self.engine = engine
        self.tokenizer = tokenizer
    }

    deinit { if let e = engine { qasr_destroy(e) } }

    /// `transcribe(audio:sampleRate:language:maxTokens:context:)` (Qwen3ASR.swift:131-137).
    /// Non-throwing like the reference: failures come back as a bracketed diagnostic string.
    public func transcribe(audio: [Float], sampleRate: Int = 16000, language: String? = nil,
                           maxTokens: Int = 448, context: String? = nil) -> String {
        guard let e = engine else { return "[qasr error: engine destroyed]" }
        var pcm = audio
        if sampleRate != 16000 {            // AudioPreprocessing.swift:327-329: resampling stays on the host
            pcm = AudioFileLoader.resample(audio, from: sampleRate, to: 16000)
        }
        let ctxIds: [Int32] = (context.flatMap { c in c.isEmpty ? nil : tokenizer?.encode(c) } ?? []).map(Int32.init)
        let langIds: [Int32] = (language.flatMap { l in tokenizer?.encode("language \(l)") } ?? []).map(Int32.init)
        var result = qasr_result()
        let rc: Int32 = ctxIds.withUnsafeBufferPointer { c in
            langIds.withUnsafeBufferPointer { l in
                var opt = qasr_options(max_tokens: Int32(maxTokens), ignore_eos: 0,
                                       context_ids: c.baseAddress, n_context: Int32(c.count),
                                       language_ids: l.baseAddress, n_language: Int32(l.count))
                return pcm.withUnsafeBufferPointer { p in
                    qasr_transcribe(e, p.baseAddress, p.count, 16000, &opt, &result)
                }
            }
        }
        guard rc == QASR_OK, let text = result.text else {
            return "[qasr error: \(String(cString: qasr_last_error(e)))]"
        }
        return String(cString: text)        // already detokenised + "<asr_text>" stripped (Qwen3ASR.swift:283-289)
    }

    /// `SpeechRecognitionModel.transcribe(audio:sampleRate:language:)` (Qwen3ASR+Protocols.swift:8-10).
    public func transcribe(audio: [Float], sampleRate: Int, language: String?) -> String {
        transcribe(audio: audio, sampleRate: sampleRate, language: language, maxTokens: 448)
    }

    /// New: batched transcription (the reference's `transcribe-batch` loops, TranscribeBatchCommand.swift:82-93).
    public func transcribeBatch(_ clips: [[Float]], maxTokens: Int = 448) -> [String] {
        guard let e = engine, !clips.isEmpty else { return [] }
        let stride = 449
        var tokens = [Int32](repeating: -1, count: clips.count * stride)
        var lens = [Int32](repeating: 0, count: clips.count)
        var sizes = clips.map { $0.count }
        var opt = qasr_options(max_tokens: Int32(maxTokens), ignore_eos: 0, context_ids: nil, n_context: 0,
                               language_ids: nil, n_language: 0)
        // keep every clip's storage alive for the duration of the call
        let buffers = clips.map { UnsafeMutableBufferPointer<Float>.allocate(capacity: max($0.count, 1)) }
        defer { buffers.forEach { $0.deallocate() } }
        for (b, c) in zip(buffers, clips) { _ = b.initialize(from: c) }
        var ptrs: [UnsafePointer<Float>?] = buffers.map { UnsafePointer($0.baseAddress) }
        let rc = qasr_transcribe_batch(e, &ptrs, &sizes, clips.count, 16000, &opt, &tokens, &lens)
        guard rc == QASR_OK else {
            return clips.map { _ in "[qasr error: \(String(cString: qasr_last_error(e)))]" }
        }
        var buf = [CChar](repeating: 0, count: 16 * stride + 64)
        return (0..<clips.count).map { i in
            let n = tokens.withUnsafeBufferPointer { t in
                qasr_detokenize(e, t.baseAddress! + i * stride, lens[i], &buf, buf.count)
            }
            return n >= 0 ? String(cString: buf) : "[qasr error: detokenize]"
        }
    }

    // MARK: ModelMemoryManageable (Qwen3ASR+Memory.swift:3-18)
    public var isLoaded: Bool { engine.map { qasr_is_loaded($0) != 0 } ?? false }
    public func unload() { if let e = engine { _ = qasr_unload(e) } }
    public var memoryFootprint: Int { engine.map { Int(qasr_memory_footprint($0)) } ?? 0 }

    /// speech-core hand-off: the C vtable comes straight from the library (VoicePipeline.swift:374-410
    /// builds the same struct around a Swift closure).
    public func makeSTTVtable() -> sc_stt_vtable_t {
        var vt = sc_stt_vtable_t()
        if let e = engine { _ = qasr_stt_vtable(e, &vt) }
        return vt
    }
}
