// Qwen3ForcedAlignerHIP.swift -- reference-side binding of the forced-aligner entry points of libqasr.so.
//
// NOT COMPILED in this repository (no Swift toolchain in the build image).  Written against include/qasr.h
// (qasr_align, qasr_align_long, qasr_align_words); lives next to Qwen3ASRHIP.swift in a `Qwen3ASRHIP` target that
// depends on the `CQasr` system library and on AudioCommon.
//
// Same surface as `Qwen3ForcedAligner` (Sources/Qwen3ASR/ForcedAligner.swift:52-331) and conformance to
// `ForcedAlignmentModel` (Sources/AudioCommon/Protocols.swift:170-173, ForcedAligner+Protocols.swift).
import Foundation
import AudioCommon
import CQasr

public final class Qwen3ForcedAlignerHIP: ForcedAlignmentModel {
    private var engine: OpaquePointer?

    /// `Qwen3ForcedAligner.fromPretrained` (ForcedAligner.swift:385-440): download in Swift, weights + vocab + merges are
    /// read by the engine from the cache directory (`thinker.` keys, PyTorch conv layout, float `lm_head`).
    public static func fromPretrained(
        modelId: String = "aufklarer/Qwen3-ForcedAligner-0.6B-4bit",
        cacheDir: URL? = nil,
        offlineMode: Bool = false,
        device: Int32 = 0,
        maxAudioSeconds: Int32 = 300
    ) async throws -> Qwen3ForcedAlignerHIP {
        let dir = try cacheDir ?? HuggingFaceDownloader.getCacheDirectory(for: modelId)
        try await HuggingFaceDownloader.downloadWeights(
            modelId: modelId, to: dir,
            additionalFiles: ["vocab.json", "merges.txt", "tokenizer_config.json", "quantize_config.json"],
            offlineMode: offlineMode, progressHandler: { _ in })
        var cfg = qasr_config()
        // the preset string carries the variant ("...-4bit" / "-8bit" / "-bf16"), like ForcedAlignerVariant.detect
        guard qasr_default_config(modelId, &cfg) == QASR_OK, cfg.classify_num > 0 else {
            throw AudioModelError.modelLoadFailed(modelId: modelId, reason: "not a forced-aligner preset")
        }
        cfg.device = device
        cfg.max_audio_seconds = maxAudioSeconds
        var handle: OpaquePointer?
        guard qasr_create(dir.path, &cfg, &handle) == QASR_OK, let h = handle else {
            throw AudioModelError.modelLoadFailed(modelId: modelId, reason: String(cString: qasr_last_error(nil)))
        }
        return Qwen3ForcedAlignerHIP(engine: h)
    }

    private init(engine: OpaquePointer) { self.engine = engine }
    deinit { if let e = engine { qasr_destroy(e) } }

    /// The languages `TextPreprocessor` hands to Apple's NLTokenizer (TextPreprocessing.swift:103-129): the split stays in
    /// Swift and the (surface, cleaned) pairs go to `qasr_align_words`; everything else is split inside the library.
    private static func usesNLTokenizer(_ language: String) -> Bool {
        let l = language.lowercased()
        return ["japanese", "korean", "thai", "lao", "khmer", "burmese", "myanmar", "tibetan"].contains { l.contains($0) }
            || ["ja", "ko", "th", "lo", "km", "my", "bo"].contains(l)
    }

    private func collect(_ rc: Int32, _ out: qasr_alignment) -> [AlignedWord] {
        guard rc == QASR_OK, let words = out.words else {       // the reference prints and returns [] on failure (:233-236)
            if let e = engine { print("Error: \(String(cString: qasr_last_error(e)))") }
            return []
        }
        return (0..<out.n_words).map {
            AlignedWord(text: String(cString: words[$0].text), startTime: words[$0].start_time, endTime: words[$0].end_time)
        }
    }

    public func align(audio: [Float], text: String, sampleRate: Int = 16000, language: String = "English") -> [AlignedWord] {
        let pcm = sampleRate == 16000 ? audio : AudioFileLoader.resample(audio, from: sampleRate, to: 16000)
        var out = qasr_alignment()
        if Self.usesNLTokenizer(language) {
            let pairs = TextPreprocessor.splitIntoWordPairs(text, language: language)
            var surf = pairs.map { strdup($0.surface) }
            var clean = pairs.map { strdup($0.cleaned) }
            defer { surf.forEach { free($0) }; clean.forEach { free($0) } }
            let rc = pcm.withUnsafeBufferPointer { p in
                surf.withUnsafeMutableBufferPointer { s in
                    clean.withUnsafeMutableBufferPointer { c in
                        qasr_align_words(engine, p.baseAddress, p.count, 16000,
                                         UnsafePointer(OpaquePointer(s.baseAddress)), UnsafePointer(OpaquePointer(c.baseAddress)),
                                         pairs.count, &out)
                    }
                }
            }
            return collect(rc, out)
        }
        let rc = pcm.withUnsafeBufferPointer { qasr_align(engine, $0.baseAddress, $0.count, 16000, text, language, &out) }
        return collect(rc, out)
    }

    /// `alignLong` (ForcedAligner.swift:97-180): plateau detection and re-alignment run inside the library.
    public func alignLong(audio: [Float], text: String, sampleRate: Int = 16000, language: String = "English") -> [AlignedWord] {
        if Self.usesNLTokenizer(language) { return align(audio: audio, text: text, sampleRate: sampleRate, language: language) }
        let pcm = sampleRate == 16000 ? audio : AudioFileLoader.resample(audio, from: sampleRate, to: 16000)
        var out = qasr_alignment()
        let rc = pcm.withUnsafeBufferPointer { qasr_align_long(engine, $0.baseAddress, $0.count, 16000, text, language, &out) }
        return collect(rc, out)
    }

    // ForcedAlignmentModel
    public func align(audio: [Float], text: String, sampleRate: Int, language: String?) -> [AlignedWord] {
        align(audio: audio, text: text, sampleRate: sampleRate, language: language ?? "English")
    }
}
