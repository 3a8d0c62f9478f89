// OmnilingualASRHIP.swift -- Swift shim a maintainer of ivan-digital/qwen3-asr-swift would add to put libqasr.so behind
// the OmnilingualASRMLXModel surface (Sources/OmnilingualASR/MLX/OmnilingualMLXModel.swift:20-210,
// MLX/OmnilingualASRMLXModel+Protocols.swift:3-43).  Written against include/qasr.h; UNCOMPILED here (no Swift toolchain in
// the build image).  Module map: see INTEGRATION.md (CQasr).
import Foundation
import AudioCommon
import CQasr

public final class OmnilingualASRHIPModel: SpeechRecognitionModel, ModelMemoryManageable {
    private var engine: OpaquePointer?
    public let sampleRate = 16000
    public var inputSampleRate: Int { sampleRate }
    public static let maxAudioSeconds: Double = 40.0

    /// `directory` = the reference's cache directory of an `aufklarer/Omnilingual-ASR-CTC-*-MLX-*bit` repo
    /// (model.safetensors + tokenizer.model); downloading stays with HuggingFaceDownloader.
    public init(modelId: String, directory: URL, maxBatch: Int32 = 32) throws {
        var cfg = qasr_ctc_config()
        guard qasr_ctc_default_config(modelId, &cfg) == QASR_OK else {
            throw AudioModelError.modelLoadFailed(modelId: modelId, reason: "unknown Omnilingual variant", underlying: nil)
        }
        cfg.max_batch = maxBatch
        var e: OpaquePointer?
        guard qasr_ctc_create(directory.path, &cfg, &e) == QASR_OK, let created = e else {
            throw AudioModelError.modelLoadFailed(modelId: modelId, reason: String(cString: qasr_ctc_last_error(nil)), underlying: nil)
        }
        engine = created
    }

    deinit { if let e = engine { qasr_ctc_destroy(e) } }

    /// Throws like the reference's `transcribeAudio` (40 s cap, unloaded model); "" for empty input.
    public func transcribeAudio(_ audio: [Float], sampleRate: Int, language: String? = nil) throws -> String {
        let samples = sampleRate == 16000 ? audio : AudioFileLoader.resample(audio, from: sampleRate, to: 16000)
        var text: UnsafePointer<CChar>? = nil
        let rc = samples.withUnsafeBufferPointer { qasr_ctc_transcribe(engine, $0.baseAddress, samples.count, 16000, &text) }
        guard rc == QASR_OK, let t = text else {
            throw AudioModelError.inferenceFailed(operation: "transcribe", reason: String(cString: qasr_ctc_last_error(engine)))
        }
        return String(cString: t)
    }

    public func transcribe(audio: [Float], sampleRate: Int, language: String?) -> String {
        (try? transcribeAudio(audio, sampleRate: sampleRate, language: language)) ?? ""
    }

    public var isLoaded: Bool { qasr_ctc_is_loaded(engine) != 0 }
    public func unload() { _ = qasr_ctc_unload(engine) }
    public var memoryFootprint: Int { Int(qasr_ctc_memory_footprint(engine)) }
}
