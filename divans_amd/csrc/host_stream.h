// host_stream.h -- host side of a literal-only .divans stream (product code, C++; no oracle involved).
// The command stream (CMD coder), the two-stream Mux, header and CRC-32C trailer are serial, branchy,
// a few percent of the symbols: they stay on the host, while every literal byte
// goes through the HIP kernels.  Citations are relative to the reference tree.
#ifndef DIVANS_HOST_STREAM_H_
#define DIVANS_HOST_STREAM_H_
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <utility>
#include <memory>
#include <vector>

#include "../../include/divans_gpu.h"

namespace divans_host {

struct StreamOptions {            // DivansCompressorOptions, src/interface.rs:444-484 (fields the literal-only path reads)
    int window_size = 22;
    uint8_t dynamic_context_mixing = 1;
    bool has_prior_depth = false; uint8_t prior_depth = 0;
    bool use_context_map = true;
    uint8_t force_stride = 9;     // StrideSelection: 0..8, 9 = UseBrotliRec
    bool has_literal_adaptation = false; divans_speed literal_adaptation[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    int use_brotli = 1;           // BrotliCompressionSetting (only 0 = internal command selection is implemented)
    int wire = 0;                 // DIVANS_WIRE_* (divans_batch.h): which build's PredictionMode prior rows the CMD model uses; the coders use HEAD
};

struct PredictionModeIn {                 // the PredictionMode command as the encoder receives it (raw_to_cmd/mod.rs:115-143, bin/divans.rs:199-316)
    uint8_t prediction_mode = 0, is_adv = 0;
    std::vector<uint8_t> literal_context_map, distance_context_map, mixing_values;
    bool has_context_speeds = false;
    uint8_t cm_speed[2][2] = {{0, 0}, {0, 0}}, stride_speed[2][2] = {{0, 0}, {0, 0}}, combined_speed[2][2] = {{0, 0}, {0, 0}};   // (inc, lim) as f8
};
// What LiteralBookKeeping holds after `pm` went through the CMD coder under `opt` (null pm: the constructor defaults).
int lit_config_from_prediction_mode(const StreamOptions& opt, const PredictionModeIn* pm, divans_lit_config& cfg);
uint8_t speed_to_f8(int16_t v);           // probability/interface.rs:566-575

// A stream split into the half that needs no literal data (CMD coder bytes, the order in which coder bytes reach the Mux)
// and the half that does (assemble_container): the first runs on host threads while the GPU codes the literals.
struct StreamPlan {
    // CmdAvail: the CMD coder holds `value` bytes in all, drained before its next nibble; LitChunk: chunk `value` of the LIT coder
    // completes and is drained; LitChunkLast: it completes on the LAST byte of its Literal command, where the drain's status is
    // dropped (codec/literal.rs:376-390); LitDrain: a point where the LIT coder is drained again (start of a Literal's content,
    // DivansCodec::flush); NewCall: the application's next call; InputDone: the current encode call has taken all of its input
    enum Kind : uint8_t { CmdAvail, LitChunk, LitChunkLast, LitDrain, NewCall, InputDone };
    struct Step { Kind kind; uint32_t value; };
    size_t n = 0; int window = 22;
    divans_lit_config cfg;                        // what the LIT coder runs under (from the stream's PredictionMode)
    std::vector<uint8_t> cmd;
    std::vector<Step> steps;
    uint32_t lit_chunks = 0;                      // 65 536-symbol chunks of the LIT stream
};
struct PlanPrefix;     // model and CMD coder after the PredictionMode command of the internal compressor under given options
std::shared_ptr<const PlanPrefix> make_plan_prefix(const StreamOptions& opt);   // null when the options cannot be coded
int plan_stream(const StreamOptions& opt, size_t n, const std::vector<size_t>* call_inputs, StreamPlan& plan, const PlanPrefix* prefix = nullptr);
int assemble_container(const StreamPlan& plan, const uint8_t* lit, size_t lit_size, const uint32_t* chunk_bytes, size_t call_buffer,
                       std::vector<uint8_t>& out);

// DivansCompressor with the internal command selection, call by call and with bounded memory (src/divans_compressor.rs:276-426):
// the ring of 2^window bytes, every Literal command coded on the GPU the moment the reference would code it
// (divans_gpu_lit_stream_*), container bytes handed out inside the encode() calls as the reference hands them out.
// encode() / flush() return 0 (NeedsMoreInput resp. Success), 1 (NeedsMoreOutput) or a negative DIVANS_GPU_E* code.
class StreamEncoder {
  public:
    StreamEncoder(const StreamOptions& opt, int device);
    ~StreamEncoder();
    StreamEncoder(const StreamEncoder&) = delete;
    StreamEncoder& operator=(const StreamEncoder&) = delete;
    int encode(const uint8_t* in, size_t n, size_t* in_off, uint8_t* out, size_t cap, size_t* out_off);
    int flush(uint8_t* out, size_t cap, size_t* out_off);
  private:
    struct Impl;
    Impl* p_;
};

// DivansDecompressor for literal-only containers, call by call (src/divans_decompressor.rs:356-397): the container is demultiplexed
// as it arrives, the CMD coder is read as far as its bytes reach (command types, the PredictionMode, literal lengths), and the
// literals are decoded on the GPU one or two 65 536-symbol chunks at a time as soon as the commands read so far cover them and their
// coded bytes are certainly in (divans_gpu_lit_stream_decode); decoded bytes are handed out in the same call.  Neither the container
// nor the output is held as a whole.  decode() returns 0 (done), 1 (NeedsMoreInput), 2 (NeedsMoreOutput) or a negative code:
// a DIVANS_GPU_E* value, -100 corrupt, -101 a stream this decoder does not take (Copy / Dict commands, a second PredictionMode).
class StreamDecoder {
  public:
    StreamDecoder(bool skip_crc, size_t max_output, int device);
    ~StreamDecoder();
    StreamDecoder(const StreamDecoder&) = delete;
    StreamDecoder& operator=(const StreamDecoder&) = delete;
    int decode(const uint8_t* in, size_t n, size_t* in_off, uint8_t* out, size_t cap, size_t* out_off);
  private:
    struct Impl;
    Impl* p_;
};

enum ParseStatus { PARSE_OK = 0, PARSE_NEED_MORE = 1, PARSE_CORRUPT = 2, PARSE_UNSUPPORTED = 3, PARSE_GPU_ERROR = 4 };
struct ParsedStream {
    // the LIT configuration (25 KB): shared with the memo that interned it and with every other stream of that configuration -- a batch
    // of 16 384 containers used to carry 400 MB of identical copies through parsing (round 6); never null after PARSE_OK
    std::shared_ptr<const divans_lit_config> cfg;
    size_t total = 0;
    int cfg_id = -1;                                         // with a ParseMemo: equal ids (>= 0) of one memo <=> identical `*cfg`
    std::vector<uint8_t> lit;                                // the LIT coder's bytes (left empty when parse_container_host is asked for spans)
    std::vector<std::pair<uint32_t, uint32_t>> lit_spans;    // ... or where they lie in the container: (offset, length) of every LIT slice, in order
    size_t lit_size = 0;                                     // their total either way
    void copy_lit(const uint8_t* container, uint8_t* dst) const {   // gathers the spans (the batch interface stages them straight into page-locked memory)
        for (const auto& sp : lit_spans) { std::memcpy(dst, container + sp.first, sp.second); dst += sp.second; }   // <= 64 KiB pieces
    }
};
// What a CMD stream decodes to is a function of its bytes alone, and a batch of equal-length literal-only streams coded under the same
// options carries the same few hundred CMD bytes in every container (PredictionMode + one literal length per ring lap): the memo keeps
// (CMD bytes -> decoded size, LIT configuration) of the streams parsed so far, so that the 8.2 k nibbles of a PredictionMode are walked
// once per distinct CMD stream and not once per container (220 us -> 15 us per 64 KiB container on one core).  Thread-safe; bounded
// (65 536 CMD streams of ~100 bytes, 64 distinct configurations of 25 KB).
class ParseMemo {
  public:
    ParseMemo();
    ~ParseMemo();
    ParseMemo(const ParseMemo&) = delete;
    ParseMemo& operator=(const ParseMemo&) = delete;
    struct Impl;
    Impl* p_;
};
// The host half of parse_container: framing + CRC + CMD coder; `ps` gets the LIT-coder bytes, the decoded size and the LIT configuration.
// spans_only: the whole container is at hand and stays there -- the LIT bytes are not copied out, ps.lit_spans says where they are.
ParseStatus parse_container_host(const uint8_t* in, size_t n, bool skip_crc, size_t max_output, ParsedStream& ps, size_t* consumed,
                                 ParseMemo* memo = nullptr, bool spans_only = false);

// divans_probe_container (divans_batch.h): the same walk, reporting instead of refusing
struct divans_container_probe_fields {
    int status = 2; uint8_t window = 0, crc_ok = 0, have_pm = 0, stopped_at = 0;
    uint32_t cmd_bytes = 0, lit_bytes = 0, commands = 0, cmd_nibbles = 0, first_literal_length = 0; uint64_t literal_bytes = 0;
    divans_lit_config cfg;
};
void probe_container_host(const uint8_t* in, size_t n, int wire, divans_container_probe_fields& pr);

uint32_t crc32c(uint32_t crc, const uint8_t* p, size_t n);   // src/codec/crc32.rs (SSE4.2 crc32 where the CPU has it)
uint32_t crc32c_portable(uint32_t crc, const uint8_t* p, size_t n);   // the table walk, always

}  // namespace divans_host
#endif
