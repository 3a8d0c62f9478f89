// NemotronStreamingHIP.swift -- Swift shim a maintainer of ivan-digital/qwen3-asr-swift would add to move the Swift-side work of the
// Nemotron streaming session (Sources/NemotronStreamingASR/StreamingSession.swift:110-231) behind libqasr.so: chunk cutting, the
// extractRaw log-mel (batched over every stream that has a chunk ready) and the RNNT greedy loop.  The encoder / decoder / joint stay the
// reference's CoreML models -- they are opaque bundles and are not rebuilt.  Written against include/qasr.h; UNCOMPILED here (no Swift
// toolchain in the build image).  Module map: see INTEGRATION.md (CQasr).
import CoreML
import Foundation
import AudioCommon
import CQasr

/// One device context shared by all sessions of a process: a call takes one 160 ms chunk of EVERY stream that is ready.
public final class NemoMelDevice {
    let handle: OpaquePointer
    public init(device: Int32 = 0, maxStreams: Int32 = 64, maxSamples: Int = 17 * 160) throws {
        var h: OpaquePointer?
        guard qasr_nemo_mel_create(device, maxStreams, maxSamples, 2.0, &h) == QASR_OK, let created = h else {
            throw AudioModelError.modelLoadFailed(modelId: "nemo-mel", reason: String(cString: qasr_nemo_mel_last_error(nil)), underlying: nil)
        }
        handle = created
    }
    deinit { qasr_nemo_mel_destroy(handle) }

    /// chunks: one 2720-sample buffer per ready stream -> [stream][128][17] float32 (StreamingSession.truncateMel / padMel applied)
    public func extractRaw(chunks: [[Float]]) throws -> [Float] {
        var out = [Float](repeating: 0, count: chunks.count * 128 * 17)
        var lens = [Int32](repeating: 0, count: chunks.count)
        var counts = chunks.map { $0.count }
        let rc = withExtendedLifetime(chunks) { () -> Int32 in
            var ptrs: [UnsafePointer<Float>?] = chunks.map { $0.withUnsafeBufferPointer { $0.baseAddress } }
            return qasr_nemo_mel_extract(handle, QASR_NEMO_MEL_RAW, &ptrs, &counts, chunks.count, nil, &out, 17, &lens, 17)
        }
        guard rc == QASR_OK else {
            throw AudioModelError.inferenceFailed(operation: "nemo mel", reason: String(cString: qasr_nemo_mel_last_error(handle)))
        }
        return out
    }
}

/// The per-stream part: sample bookkeeping + RNNT greedy over the caller's CoreML decoder / joint.
public final class NemotronHIPSession {
    private var chunker: OpaquePointer?
    private var cfg = qasr_transducer_config()
    private let decoder: MLModel, joint: MLModel
    public private(set) var tokens: [Int32] = [], logProbs: [Float] = []

    public init(decoder: MLModel, joint: MLModel) {
        self.decoder = decoder; self.joint = joint
        qasr_transducer_default_config("nemotron-streaming", &cfg)
        qasr_stream_chunker_create(17 * 160, 2 * 8 * 160, &chunker)           // StreamingSession.swift:113-115
    }
    deinit { qasr_stream_chunker_destroy(chunker) }

    public func push(_ samples: [Float]) { samples.withUnsafeBufferPointer { _ = qasr_stream_chunker_push(chunker, $0.baseAddress, samples.count) } }
    public func popChunk() -> [Float]? {
        var chunk = [Float](repeating: 0, count: 17 * 160)
        return qasr_stream_chunker_pop(chunker, &chunk) == 1 ? chunk : nil
    }
    public func flushChunk() -> [Float]? {
        var chunk = [Float](repeating: 0, count: 17 * 160)
        return qasr_stream_chunker_flush(chunker, &chunk) == 1 ? chunk : nil
    }

    /// RNNTGreedyDecoder.decode (RNNTGreedyDecoder.swift:38-90) over `encodedLength` frames of the encoder output the caller just produced.
    /// `step` / `jointLogits` wrap decoder.prediction / joint.prediction exactly as the reference's loop does (token -> h, c, decoder_output;
    /// (encoder frame, decoder_output) -> float16 logits widened to Float).
    public func decode(encodedLength: Int32, step: @escaping (Int32) -> Bool, jointLogits: @escaping (Int32, UnsafeMutablePointer<Float>) -> Bool) {
        final class Box { let s: (Int32) -> Bool; let j: (Int32, UnsafeMutablePointer<Float>) -> Bool
            init(_ s: @escaping (Int32) -> Bool, _ j: @escaping (Int32, UnsafeMutablePointer<Float>) -> Bool) { self.s = s; self.j = j } }
        let box = Box(step, jointLogits)
        var cb = qasr_transducer_callbacks(
            ctx: Unmanaged.passUnretained(box).toOpaque(),
            decoder_step: { ctx, token in Unmanaged<Box>.fromOpaque(ctx!).takeUnretainedValue().s(token) ? 0 : 1 },
            joint: { ctx, frame, logits, _ in Unmanaged<Box>.fromOpaque(ctx!).takeUnretainedValue().j(frame, logits!) ? 0 : 1 })
        var ids = [Int32](repeating: 0, count: 64), lps = [Float](repeating: 0, count: 64)
        let n = withExtendedLifetime(box) { qasr_rnnt_greedy_decode(&cfg, &cb, encodedLength, 0, &ids, &lps, 64, nil) }
        if n > 0 { tokens += ids[0..<Int(n)]; logProbs += lps[0..<Int(n)] }
    }

    public var confidence: Float { logProbs.withUnsafeBufferPointer { qasr_transducer_confidence($0.baseAddress, Int32(logProbs.count)) } }
}
