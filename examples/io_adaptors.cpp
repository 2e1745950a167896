// Round trips through examples/divans_io.hpp, the C++ form of the reference's writer / reader adaptors
// (src/writer.rs, src/reader.rs; their own tests: writer.rs:298-420, reader.rs:329-470 -- odd chunk sizes in, odd chunk
// sizes out, the compressed stream as an in-memory buffer).  usage: io_adaptors <input file> <out.divans>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "divans_io.hpp"

typedef std::vector<uint8_t> Bytes;

struct VecSink {
    Bytes* v;
    void operator()(const uint8_t* p, size_t n) const { v->insert(v->end(), p, p + n); }
};
struct VecSource {
    const Bytes* v; size_t pos; size_t step;
    size_t operator()(uint8_t* buf, size_t cap) {
        size_t n = v->size() - pos; if (n > cap) n = cap; if (n > step) n = step;
        std::memcpy(buf, v->data() + pos, n); pos += n; return n;
    }
};

static Bytes compress(const Bytes& in, size_t write_step, size_t buffer_size, const divans::Options& opts) {
    Bytes out;
    divans::DivansCompressorWriter<VecSink> w(VecSink{&out}, opts, buffer_size);
    for (size_t i = 0; i < in.size(); i += write_step) {
        const size_t n = in.size() - i < write_step ? in.size() - i : write_step;
        if (w.write(in.data() + i, n) != n) { std::fprintf(stderr, "short write\n"); std::exit(2); }
    }
    w.flush();
    w.flush();    // idempotent
    return out;
}

static Bytes decompress_reader(const Bytes& dv, size_t source_step, size_t read_step, size_t buffer_size) {
    Bytes out; Bytes chunk(read_step);
    divans::DivansDecompressorReader<VecSource> r(VecSource{&dv, 0, source_step}, buffer_size);
    for (;;) {
        const size_t n = r.read(chunk.data(), chunk.size());
        if (n == 0) break;
        out.insert(out.end(), chunk.begin(), chunk.begin() + n);
    }
    return out;
}

static Bytes decompress_writer(const Bytes& dv, size_t write_step, size_t buffer_size) {
    Bytes out;
    divans::DivansDecompressorWriter<VecSink> w(VecSink{&out}, buffer_size);
    for (size_t i = 0; i < dv.size(); i += write_step) w.write(dv.data() + i, dv.size() - i < write_step ? dv.size() - i : write_step);
    w.flush();
    return out;
}

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    std::ifstream f(argv[1], std::ios::binary);
    const Bytes in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    CHECK(!in.empty());
    const divans::Options opts = {{DIVANS_OPTION_USE_BROTLI_COMMAND_SELECTION, 0}, {DIVANS_OPTION_DYNAMIC_CONTEXT_MIXING, 2},
                                  {DIVANS_OPTION_USE_CONTEXT_MAP, 1}, {DIVANS_OPTION_WINDOW_SIZE, 16}};
    // the container depends on the caller's call pattern -- write sizes through the ring buffer, the output chunk size through
    // the Mux: equal patterns give equal bytes, and every pattern must round-trip
    const Bytes a = compress(in, 65536, 0, opts), a2 = compress(in, 65536, 4096, opts), b = compress(in, 65536, 1000, opts), c = compress(in, 7777, 333, opts);
    CHECK(a == a2);
    CHECK(decompress_reader(b, 999, 4097, 0) == in);
    CHECK(a.size() > 24 && a[0] == 0xff && a[1] == 0xe5 && a[2] == 0x8c && a[3] == 0x9f);
    CHECK(decompress_reader(a, 1 << 20, 1 << 16, 0) == in);
    CHECK(decompress_reader(a, 3, 5, 7) == in);
    CHECK(decompress_reader(c, 4096, 100000, 4096) == in);
    CHECK(decompress_writer(a, 1 << 20, 0) == in);
    CHECK(decompress_writer(c, 13, 1) == in);
    CHECK(compress(Bytes(), 1, 0, opts).size() >= 24 && decompress_reader(compress(Bytes(), 1, 0, opts), 100, 100, 0).empty());
    // a stream that stops early, a damaged stream, bytes behind the end
    Bytes cut(a.begin(), a.begin() + a.size() / 2);
    bool threw = false;
    try { decompress_reader(cut, 4096, 4096, 0); } catch (const divans::IoError& e) { threw = e.kind == divans::IoError::UnexpectedEof; }
    CHECK(threw);
    threw = false;
    try { decompress_writer(cut, 4096, 0); } catch (const divans::IoError& e) { threw = e.kind == divans::IoError::UnexpectedEof; }
    CHECK(threw);
    Bytes bad = a; bad[bad.size() / 2] ^= 0x10;
    threw = false;
    try { decompress_reader(bad, 4096, 4096, 0); } catch (const divans::IoError& e) { threw = e.kind == divans::IoError::InvalidData; }
    CHECK(threw);
    threw = false;      // bytes written after the stream has ended
    try {
        Bytes sink_bytes; divans::DivansDecompressorWriter<VecSink> w(VecSink{&sink_bytes});
        w.write(a.data(), a.size());
        CHECK(w.finished() && sink_bytes == in);
        const uint8_t extra = 0x55; w.write(&extra, 1);
    } catch (const divans::IoError& e) { threw = e.kind == divans::IoError::TrailingInput; }
    CHECK(threw);
    threw = false;
    try { divans::Options o = {{200, 1}}; compress(in, 4096, 0, o); } catch (const divans::IoError& e) { threw = e.kind == divans::IoError::InvalidInput; }
    CHECK(threw);
    std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char*>(a.data()), (std::streamsize)a.size());
    std::printf("ok %zu -> %zu\n", in.size(), a.size());
    return 0;
}
