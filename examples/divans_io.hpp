// divans_io.hpp -- the reference's stream adaptors over the C ABI of divans_ffi.h, header only.
//
// The reference wraps its compressor / decompressor in std::io adaptors (src/writer.rs, src/reader.rs); the names its
// users know are kept here:
//   divans::DivansCompressorWriter<Sink>     bytes written in, .divans bytes pushed to `sink`; flush() ends the stream
//                                            (DivansExperimentalCompressorWriter / GenWriter, writer.rs:30-98,192-227)
//   divans::DivansDecompressorReader<Source> .divans bytes pulled from `source`, plain bytes read out
//                                            (DivansDecompressorReader / GenReader, reader.rs:45-126,289-325)
//   divans::DivansDecompressorWriter<Sink>   .divans bytes written in, plain bytes pushed to `sink` (writer.rs:254-296)
// Sink   = anything callable as  sink(const uint8_t* data, size_t n)            (must take all n bytes or throw)
// Source = anything callable as  size_t source(uint8_t* buf, size_t capacity)   (0 = end of input)
// Errors are exceptions (divans::IoError) where the reference returns io::Error: InvalidInput / InvalidData for
// DIVANS_FAILURE, UnexpectedEof for a stream that ends early, TrailingInput for bytes after the end of a stream.
#ifndef DIVANS_IO_HPP_
#define DIVANS_IO_HPP_
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "divans_ffi.h"

namespace divans {

struct IoError : std::runtime_error {
    enum Kind { InvalidInput, InvalidData, UnexpectedEof, TrailingInput } kind;
    IoError(Kind k, const std::string& what) : std::runtime_error(what), kind(k) {}
};

using Options = std::vector<std::pair<DivansOptionSelect, uint32_t>>;

template <class Sink>
class DivansCompressorWriter {
public:
    // buffer_size 0 selects the reference's 4096 (writer.rs:206-208); it is the size of the output chunks handed to the sink
    explicit DivansCompressorWriter(Sink sink, const Options& options = Options(), size_t buffer_size = 0)
        : sink_(std::move(sink)), buffer_(buffer_size ? buffer_size : 4096), state_(divans_new_compressor()) {
        if (!state_) throw IoError(IoError::InvalidInput, "divans_new_compressor failed");
        for (const auto& o : options)
            if (divans_set_option(state_, o.first, o.second) != DIVANS_SUCCESS) {
                divans_free_compressor(state_); state_ = nullptr;
                throw IoError(IoError::InvalidInput, "divans_set_option rejected selector " + std::to_string(o.first));
            }
    }
    DivansCompressorWriter(const DivansCompressorWriter&) = delete;
    DivansCompressorWriter& operator=(const DivansCompressorWriter&) = delete;
    ~DivansCompressorWriter() { if (state_) divans_free_compressor(state_); }

    // Takes all n bytes (or throws): the compressor consumes input until it asks for more, output goes to the sink as it appears.
    size_t write(const uint8_t* data, size_t n) {
        if (flushed_) throw IoError(IoError::InvalidInput, "write after flush");
        size_t in_off = 0;
        for (;;) {
            size_t out_off = 0;
            const DivansResult r = divans_encode(state_, data, n, &in_off, buffer_.data(), buffer_.size(), &out_off);
            if (out_off) sink_(static_cast<const uint8_t*>(buffer_.data()), out_off);
            if (r == DIVANS_FAILURE) throw IoError(IoError::InvalidInput, "divans_encode failed");
            if (r == DIVANS_NEEDS_MORE_OUTPUT) continue;
            if (r == DIVANS_NEEDS_MORE_INPUT && in_off != n) throw IoError(IoError::TrailingInput, "compressor left input unread");
            return n;
        }
    }
    // Ends the stream: remaining commands, trailer.  Further calls do nothing (writer.rs:65-83).
    void flush() {
        while (!flushed_) {
            size_t out_off = 0;
            const DivansResult r = divans_encode_flush(state_, buffer_.data(), buffer_.size(), &out_off);
            if (out_off) sink_(static_cast<const uint8_t*>(buffer_.data()), out_off);
            if (r == DIVANS_FAILURE) throw IoError(IoError::InvalidInput, "divans_encode_flush failed");
            if (r == DIVANS_SUCCESS) flushed_ = true;
        }
    }
    Sink& sink() { return sink_; }

private:
    Sink sink_;
    std::vector<uint8_t> buffer_;
    DivansCompressorState* state_;
    bool flushed_ = false;
};

template <class Source>
class DivansDecompressorReader {
public:
    // buffer_size 0 selects the reference's 4096 (reader.rs:300-302): how much compressed input is pulled per source call
    explicit DivansDecompressorReader(Source source, size_t buffer_size = 0, bool skip_crc = false, bool multithread = false)
        : source_(std::move(source)), buffer_(buffer_size ? buffer_size : 4096) {
        CAllocator libc_alloc = {nullptr, nullptr, nullptr};
        state_ = divans_new_decompressor_with_custom_alloc(libc_alloc, skip_crc ? 1 : 0, multithread ? 1 : 0);
        if (!state_) throw IoError(IoError::InvalidInput, "divans_new_decompressor failed");
    }
    DivansDecompressorReader(const DivansDecompressorReader&) = delete;
    DivansDecompressorReader& operator=(const DivansDecompressorReader&) = delete;
    ~DivansDecompressorReader() { if (state_) divans_free_decompressor(state_); }

    // Up to n plain bytes; 0 only at the verified end of the stream.  Throws UnexpectedEof when the source dries up first.
    size_t read(uint8_t* out, size_t n) {
        if (done_ || n == 0) return 0;
        size_t produced = 0;
        while (produced == 0) {
            if (begin_ == end_ && !source_eof_) {
                begin_ = 0;
                end_ = source_(buffer_.data(), buffer_.size());
                if (end_ == 0) source_eof_ = true;
            }
            size_t in_off = begin_, out_off = 0;
            const DivansResult r = divans_decode(state_, buffer_.data(), end_, &in_off, out, n, &out_off);
            begin_ = in_off;
            produced = out_off;
            if (r == DIVANS_FAILURE) throw IoError(IoError::InvalidData, "divans_decode failed (corrupt stream, bad checksum or unsupported commands)");
            if (r == DIVANS_SUCCESS) { done_ = true; break; }
            if (r == DIVANS_NEEDS_MORE_INPUT && begin_ == end_ && source_eof_ && produced == 0)
                throw IoError(IoError::UnexpectedEof, "compressed stream ends before its trailer");
        }
        return produced;
    }
    // bytes of the source that were fetched but lie behind the end of the stream (0 until read() has returned 0)
    size_t unread_input() const { return end_ - begin_; }

private:
    Source source_;
    std::vector<uint8_t> buffer_;
    DivansDecompressorState* state_ = nullptr;
    size_t begin_ = 0, end_ = 0;
    bool source_eof_ = false, done_ = false;
};

template <class Sink>
class DivansDecompressorWriter {
public:
    explicit DivansDecompressorWriter(Sink sink, size_t buffer_size = 0, bool skip_crc = false, bool multithread = false)
        : sink_(std::move(sink)), buffer_(buffer_size ? buffer_size : 4096) {
        CAllocator libc_alloc = {nullptr, nullptr, nullptr};
        state_ = divans_new_decompressor_with_custom_alloc(libc_alloc, skip_crc ? 1 : 0, multithread ? 1 : 0);
        if (!state_) throw IoError(IoError::InvalidInput, "divans_new_decompressor failed");
    }
    DivansDecompressorWriter(const DivansDecompressorWriter&) = delete;
    DivansDecompressorWriter& operator=(const DivansDecompressorWriter&) = delete;
    ~DivansDecompressorWriter() { if (state_) divans_free_decompressor(state_); }

    size_t write(const uint8_t* data, size_t n) {
        size_t in_off = 0;
        for (;;) {
            if (done_) {
                if (in_off != n) throw IoError(IoError::TrailingInput, "bytes after the end of the compressed stream");
                return n;
            }
            size_t out_off = 0;
            const DivansResult r = divans_decode(state_, data, n, &in_off, buffer_.data(), buffer_.size(), &out_off);
            if (out_off) sink_(static_cast<const uint8_t*>(buffer_.data()), out_off);
            if (r == DIVANS_FAILURE) throw IoError(IoError::InvalidData, "divans_decode failed (corrupt stream, bad checksum or unsupported commands)");
            if (r == DIVANS_SUCCESS) { done_ = true; continue; }
            if (r == DIVANS_NEEDS_MORE_INPUT && in_off == n) return n;
        }
    }
    // true once the trailer has been verified; a caller that has written everything and sees false has a truncated stream
    bool finished() const { return done_; }
    void flush() { if (!done_) throw IoError(IoError::UnexpectedEof, "compressed stream ends before its trailer"); }
    Sink& sink() { return sink_; }

private:
    Sink sink_;
    std::vector<uint8_t> buffer_;
    DivansDecompressorState* state_ = nullptr;
    bool done_ = false;
};

}  // namespace divans
#endif
