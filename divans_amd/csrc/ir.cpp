// ir.cpp -- host side of general (non literal-only) streams: the reference's textual command IR.
//   parser     : src/bin/divans.rs:191-483 (command_parse), hex literals src/bin/util.rs:251-286
//   expansion  : src/cmd_to_raw/mod.rs:242-330 (Literal / Copy / Dict into the ring buffer)
//   last_8     : src/cmd_to_raw/mod.rs:69-90 + src/codec/mod.rs:771-783 (reloaded after every command that fills the ring)
// Product code (declared in include/divans_ir.h); never touches the CPU oracle.
#include "../../include/divans_ir.h"

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "host_stream.h"

namespace {

struct Cmd {
    int kind = 0;
    uint32_t a = 0, b = 0;            // Copy: num_bytes, distance.  Dict: final_size, word_size.  *type: block type, stride
    uint32_t word_id = 0, transform = 0;
    size_t data_off = 0, data_len = 0;   // Literal bytes / Dict result bytes inside divans_ir::bytes
};

}  // namespace

struct divans_ir {
    std::vector<Cmd> cmds;
    std::vector<uint8_t> bytes;
    bool has_pm = false;
    divans_host::PredictionModeIn pm;
    size_t raw_size = 0, literal_size = 0, n_segments = 0;
    uint32_t n_btypes = 1;
    size_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

extern "C" const char* divans_gpu_last_error(void);
namespace divans_host { int set_last_error(int code, const std::string& msg); }

namespace {

int hexval(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }

std::vector<std::string> split(const std::string& s) {   // str::split(' '): empty fields are kept
    std::vector<std::string> v; size_t p = 0;
    for (;;) { size_t q = s.find(' ', p); v.push_back(s.substr(p, q == std::string::npos ? q : q - p)); if (q == std::string::npos) break; p = q + 1; }
    return v;
}

bool parse_u32(const std::string& s, uint32_t& v) {
    if (s.empty() || s.size() > 10) return false;
    uint64_t x = 0;
    for (char c : s) { if (c < '0' || c > '9') return false; x = x * 10 + (uint64_t)(c - '0'); }
    if (x > 0xffffffffull) return false;
    v = (uint32_t)x; return true;
}

bool hex_to_bytes(const std::string& s, std::vector<uint8_t>& out) {   // util.rs:261-286
    int rem = 0; uint8_t buf = 0;
    for (char c : s) {
        const int h = hexval(c);
        if (h < 0) { if (c == '\n' || c == '\t' || c == '\r') continue; return false; }
        buf = (uint8_t)((buf << 4) | h);
        if (++rem == 2) { rem = 0; out.push_back(buf); }
    }
    return rem == 0;
}

int fail(const std::string& msg) { return divans_host::set_last_error(DIVANS_GPU_EINVAL, "IR: " + msg); }

// numbers after keyword `key` until the first non-number (divans.rs:219-262)
bool list_after(const std::vector<std::string>& v, const char* key, std::vector<uint32_t>& out) {
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == key) {
            for (size_t j = i + 1; j < v.size(); ++j) { uint32_t x; if (!parse_u32(v[j], x)) break; out.push_back(x); }
            return true;
        }
    return false;
}

int parse_line(divans_ir& ir, const std::string& line) {
    const std::vector<std::string> v = split(line);
    const std::string& cmd = v[0];
    if (cmd == "window") return 0;
    if (cmd == "prediction") {
        if (v.size() < 2) return fail("prediction needs 1 argument");
        divans_host::PredictionModeIn pm;
        // brotli's LiteralPredictionModeNibble numbering: LSB6 0, MSB6 1, UTF8 2, SIGNED 3 (RFC 7932 section 7.1)
        if (v[1] == "utf8") pm.prediction_mode = 2; else if (v[1] == "sign") pm.prediction_mode = 3;
        else if (v[1] == "lsb6") pm.prediction_mode = 0; else if (v[1] == "msb6") pm.prediction_mode = 1;
        else return fail("invalid prediction mode; not {utf8,sign,lsb6,msb6}");
        std::vector<uint32_t> vals;
        if (list_after(v, "lcontextmap", vals)) for (uint32_t x : vals) { if (x > 255) return fail("literal context map val must be u8"); pm.literal_context_map.push_back((uint8_t)x); }
        vals.clear();
        if (list_after(v, "dcontextmap", vals)) for (uint32_t x : vals) { if (x > 255) return fail("distance context map val must be u8"); pm.distance_context_map.push_back((uint8_t)x); }
        vals.clear();
        pm.mixing_values.assign(DIVANS_GPU_NUM_MIXING_VALUES, 0);
        if (list_after(v, "mixingvalues", vals)) {
            if (vals.size() > DIVANS_GPU_NUM_MIXING_VALUES) return fail("too many mixing values");
            for (size_t i = 0; i < vals.size(); ++i) { if (vals[i] > 8) return fail("mixing value must be 0..8"); pm.mixing_values[i] = (uint8_t)vals[i]; }
        }
        pm.has_context_speeds = true;
        const char* keys[3][2] = {{"cmspeedinc", "cmspeedmax"}, {"stspeedinc", "stspeedmax"}, {"mxspeedinc", "mxspeedmax"}};
        uint8_t (*dst[3])[2] = {pm.cm_speed, pm.stride_speed, pm.combined_speed};
        for (int t = 0; t < 3; ++t)
            for (int im = 0; im < 2; ++im) {
                vals.clear();
                if (!list_after(v, keys[t][im], vals)) continue;
                for (size_t i = 0; i < vals.size() && i < 2; ++i) {
                    if (vals[i] > 16384) return fail("speed val must be u16 <= 16384");
                    dst[t][i][im] = divans_host::speed_to_f8((int16_t)vals[i]);   // PredictionModeContextMap stores speeds as f8 (set_*_speed)
                }
            }
        ir.pm = pm; ir.has_pm = true;
        Cmd c; c.kind = DIVANS_IR_PREDICTION_MODE; ir.cmds.push_back(c);
        return 0;
    }
    if (cmd == "ctype" || cmd == "ltype" || cmd == "dtype") {
        if (v.size() != 2 && (v.size() != 3 || cmd != "ltype")) return fail("*type needs 1 argument");
        uint32_t bt; if (!parse_u32(v[1], bt)) return fail("bad block type");
        Cmd c; c.a = bt & 0xffu;
        if (cmd[0] == 'c') c.kind = DIVANS_IR_BTYPE_COMMAND; else if (cmd[0] == 'd') c.kind = DIVANS_IR_BTYPE_DISTANCE;
        else {
            c.kind = DIVANS_IR_BTYPE_LITERAL;
            if (v.size() == 3) { uint32_t st; if (!parse_u32(v[2], st)) return fail("bad stride"); if (st > 8) return fail("stride must be <= 8"); c.b = st; }
        }
        ir.cmds.push_back(c);
        return 0;
    }
    if (cmd == "copy") {
        if (v.size() < 4) return fail("copy needs 4 arguments");
        Cmd c; c.kind = DIVANS_IR_COPY;
        if (!parse_u32(v[1], c.a)) return fail("bad copy length");
        if (v[2] != "from") return fail("copy needs a from statement in the 2nd arg");
        if (!parse_u32(v[3], c.b)) return fail("bad copy distance");
        if (c.a == 0) return 0;
        ir.cmds.push_back(c);
        return 0;
    }
    if (cmd == "dict") {
        if (v.size() < 6) return fail("dict needs 6+ arguments");
        Cmd c; c.kind = DIVANS_IR_DICT;
        if (!parse_u32(v[1], c.a)) return fail("bad dict length");
        c.a &= 0xffu;
        if (v[2] != "word") return fail("dict needs a word after the expected len");
        const size_t comma = v[3].find(',');
        uint32_t wl;
        if (comma == std::string::npos || !parse_u32(v[3].substr(0, comma), wl) || !parse_u32(v[3].substr(comma + 1), c.word_id)) return fail("dict needs a comma separated word value");
        c.b = wl & 0xffu;
        for (size_t i = 5; i < v.size(); ++i)
            if (v[i - 1] == "func") {
                if (!parse_u32(v[i], c.transform)) return fail("bad dict transform");
                // the transformed word follows in hex (see divans_ir.h: stands in for the static-dictionary lookup)
                std::vector<uint8_t> w;
                if (i + 1 >= v.size() || !hex_to_bytes(v[i + 1], w) || w.size() != c.a) return fail("dict command without its " + std::to_string(c.a) + " result bytes");
                c.data_off = ir.bytes.size(); c.data_len = w.size();
                ir.bytes.insert(ir.bytes.end(), w.begin(), w.end());
                ir.cmds.push_back(c);
                return 0;
            }
        return fail("dict needs a func");
    }
    if (cmd == "insert" || cmd == "rndins") {
        if (v.size() < 3) { if (v.size() == 2 && v[1] == "0") return 0; return fail("insert needs 3 arguments, not (" + line + ")"); }
        uint32_t n; if (!parse_u32(v[1], n)) return fail("bad insert length");
        if (n == 0) return 0;
        if (cmd == "rndins") return fail("high-entropy literals (rndins) are not supported");
        const std::string rest = line.substr(v[0].size() + v[1].size() + 2);
        if (!rest.empty() && rest[0] == '"') return fail("quoted literals are not supported");
        if (v.size() > 3 && !v[3].empty()) return fail("literals with external probabilities are not supported");
        std::vector<uint8_t> w;
        if (!hex_to_bytes(v[2], w) || w.size() != n) return fail("Length does not match " + line.substr(0, 60));
        Cmd c; c.kind = DIVANS_IR_LITERAL; c.a = n; c.data_off = ir.bytes.size(); c.data_len = n;
        ir.bytes.insert(ir.bytes.end(), w.begin(), w.end());
        ir.cmds.push_back(c);
        return 0;
    }
    return fail("Unknown " + line.substr(0, 60));
}

}  // namespace

extern "C" {

int divans_ir_parse(const char* text, size_t len, divans_ir** out) {
    if (!text || !out) return fail("null argument");
    std::unique_ptr<divans_ir> ir(new divans_ir());
    size_t p = 0;
    while (p < len) {
        size_t q = p;
        while (q < len && text[q] != '\n') ++q;
        std::string line(text + p, q - p);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        p = q + 1;
        if (line.empty()) continue;
        const int rc = parse_line(*ir, line);
        if (rc) return rc;
    }
    // one pass over the commands: sizes, block types, validity of the copies
    uint32_t bt = 0;
    for (const Cmd& c : ir->cmds) {
        ir->counts[c.kind & 7]++;
        if (c.kind == DIVANS_IR_LITERAL) { ir->raw_size += c.a; ir->literal_size += c.a; ir->n_segments++; if (bt + 1 > ir->n_btypes) ir->n_btypes = bt + 1; }
        else if (c.kind == DIVANS_IR_COPY) { if (c.b == 0 || c.b > ir->raw_size) return fail("copy from before the start of the stream"); ir->raw_size += c.a; }
        else if (c.kind == DIVANS_IR_DICT) ir->raw_size += c.data_len;
        else if (c.kind == DIVANS_IR_BTYPE_LITERAL) bt = c.a;
    }
    *out = ir.release();
    return 0;
}

void divans_ir_free(divans_ir* ir) { delete ir; }
size_t divans_ir_num_commands(const divans_ir* ir) { return ir ? ir->cmds.size() : 0; }
size_t divans_ir_count(const divans_ir* ir, int kind) { return ir && kind >= 0 && kind < 8 ? ir->counts[kind] : 0; }
size_t divans_ir_raw_size(const divans_ir* ir) { return ir ? ir->raw_size : 0; }
size_t divans_ir_literal_size(const divans_ir* ir) { return ir ? ir->literal_size : 0; }
size_t divans_ir_num_segments(const divans_ir* ir) { return ir ? ir->n_segments : 0; }
uint32_t divans_ir_num_block_types(const divans_ir* ir) { return ir ? ir->n_btypes : 0; }

// walks the commands once, producing the output bytes and (optionally) the literal coder's view of them
static int walk(const divans_ir* ir, uint8_t* out, uint8_t* lit, divans_lit_segment* segs) {
    size_t pos = 0, lpos = 0, nseg = 0; uint32_t bt = 0;
    for (const Cmd& c : ir->cmds) {
        if (c.kind == DIVANS_IR_LITERAL) {
            if (segs) {
                uint64_t last8 = 0;   // last_8_literals(): oldest byte first, zeros before the start (cmd_to_raw/mod.rs:69-90)
                for (int i = 0; i < 8; ++i) { const size_t back = (size_t)(8 - i); if (pos >= back) last8 |= (uint64_t)out[pos - back] << (8 * i); }
                segs[nseg++] = divans_lit_segment{c.a, bt, last8};
            }
            std::memcpy(out + pos, ir->bytes.data() + c.data_off, c.a);
            if (lit) { std::memcpy(lit + lpos, ir->bytes.data() + c.data_off, c.a); lpos += c.a; }
            pos += c.a;
        } else if (c.kind == DIVANS_IR_COPY) {
            for (uint32_t i = 0; i < c.a; ++i) out[pos + i] = out[pos + i - c.b];   // byte by byte: overlapping copies repeat
            pos += c.a;
        } else if (c.kind == DIVANS_IR_DICT) {
            std::memcpy(out + pos, ir->bytes.data() + c.data_off, c.data_len);
            pos += c.data_len;
        } else if (c.kind == DIVANS_IR_BTYPE_LITERAL) bt = c.a;
    }
    return 0;
}

int divans_ir_expand(const divans_ir* ir, uint8_t* out, size_t cap) {
    if (!ir || (!out && ir->raw_size)) return fail("null argument");
    if (cap < ir->raw_size) return divans_host::set_last_error(DIVANS_GPU_ECAP, "IR: output buffer too small");
    return walk(ir, out, nullptr, nullptr);
}

int divans_ir_literal_segments(const divans_ir* ir, uint8_t* lit, size_t lit_cap, divans_lit_segment* segs, size_t seg_cap) {
    if (!ir || (!lit && ir->literal_size) || (!segs && ir->n_segments)) return fail("null argument");
    if (lit_cap < ir->literal_size || seg_cap < ir->n_segments) return divans_host::set_last_error(DIVANS_GPU_ECAP, "IR: literal / segment buffer too small");
    std::vector<uint8_t> raw(ir->raw_size ? ir->raw_size : 1);
    return walk(ir, raw.data(), lit, segs);
}

void divans_ir_options_default(divans_ir_options* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->dynamic_context_mixing = 1; o->use_context_map = 1; o->force_stride = 9;   // src/interface.rs:463-483
}

int divans_ir_lit_config(const divans_ir* ir, const divans_ir_options* o, divans_lit_config* cfg) {
    if (!ir || !o || !cfg) return fail("null argument");
    divans_host::StreamOptions so;
    so.dynamic_context_mixing = o->dynamic_context_mixing; so.use_context_map = o->use_context_map != 0; so.force_stride = o->force_stride;
    so.has_prior_depth = o->has_prior_depth != 0; so.prior_depth = o->prior_depth;
    so.has_literal_adaptation = o->has_literal_adaptation != 0;
    for (int i = 0; i < 4; ++i) so.literal_adaptation[i] = o->literal_adaptation[i];
    const int rc = divans_host::lit_config_from_prediction_mode(so, ir->has_pm ? &ir->pm : nullptr, *cfg);
    if (rc) return divans_host::set_last_error(rc, "IR: the PredictionMode command cannot be coded (context map index out of range)");
    return 0;
}

}  // extern "C"
