/*
 * divans_ir.h -- the textual command IR of the reference (what `divans -i` reads and the files testdata/<name>.ir hold) on the
 * host: parser (src/bin/divans.rs:191-483 command_parse), expansion to the original bytes (src/cmd_to_raw/mod.rs:
 * Literal / Copy / Dict into the ring buffer; integration_test.rs:76-108 pins recode(ir) == raw file) and the split of a
 * general stream into what the GPU literal coder consumes: all literal bytes in command order plus one
 * divans_lit_segment per Literal command (include/divans_gpu.h).
 *
 * Dict commands: the reference looks the word up in brotli's static dictionary and applies the transform
 * (cmd_to_raw/mod.rs:284-318, crate `brotli` ~3.1, not vendored).  The IR text carries the resulting bytes after
 * `func <transform>`; this parser takes them from there, so no dictionary table is needed to expand an IR.
 */
#ifndef DIVANS_IR_H_
#define DIVANS_IR_H_
#include <stddef.h>
#include <stdint.h>

#include "divans_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct divans_ir divans_ir;

enum { DIVANS_IR_COPY = 0, DIVANS_IR_DICT = 1, DIVANS_IR_LITERAL = 3, DIVANS_IR_BTYPE_LITERAL = 4, DIVANS_IR_BTYPE_COMMAND = 5,
       DIVANS_IR_BTYPE_DISTANCE = 6, DIVANS_IR_PREDICTION_MODE = 7 };   /* command type nibbles, src/codec/mod.rs:143-158 */

/* Parses `len` bytes of IR text (one command per line).  0 on success, DIVANS_GPU_EINVAL with divans_gpu_last_error() otherwise. */
int divans_ir_parse(const char *text, size_t len, divans_ir **out);
void divans_ir_free(divans_ir *ir);

size_t divans_ir_num_commands(const divans_ir *ir);
/* counts per command kind (DIVANS_IR_*), 0 for kinds that do not occur */
size_t divans_ir_count(const divans_ir *ir, int kind);

/* cmd_to_raw: the bytes the command list stands for.  Copy distances reach back into the output (overlap allowed, as
 * in the ring buffer); a distance beyond the bytes produced so far is an error. */
size_t divans_ir_raw_size(const divans_ir *ir);
int divans_ir_expand(const divans_ir *ir, uint8_t *out, size_t cap);

/* What the literal coder sees: literal bytes of every Literal command back to back, and per command
 * (len, literal block type in force, the 8 output bytes before it).  Empty Literal commands do not exist (the parser
 * drops them like the reference's). */
size_t divans_ir_literal_size(const divans_ir *ir);
size_t divans_ir_num_segments(const divans_ir *ir);
uint32_t divans_ir_num_block_types(const divans_ir *ir);   /* highest literal block type + 1 */
int divans_ir_literal_segments(const divans_ir *ir, uint8_t *lit, size_t lit_cap, divans_lit_segment *segs, size_t seg_cap);

/* The compressor options that shape the PredictionMode command (DivansCompressorOptions, src/interface.rs:444-484) */
typedef struct divans_ir_options {
    uint8_t dynamic_context_mixing;   /* default 1 */
    uint8_t use_context_map;          /* default 1 */
    uint8_t force_stride;             /* StrideSelection 0..8, 9 = UseBrotliRec (default) */
    uint8_t has_prior_depth, prior_depth;
    uint8_t has_literal_adaptation;
    divans_speed literal_adaptation[4];
} divans_ir_options;
void divans_ir_options_default(divans_ir_options *o);
/* LiteralBookKeeping after the stream's PredictionMode command has gone through the CMD coder with these options
 * (src/codec/context_map.rs:105-428 + obs_prediction_mode_context_map, codec/interface.rs:293-319); an IR without a
 * prediction command leaves the LiteralBookKeeping::new defaults (codec/interface.rs:244-262).  `cfg->btype` = 0: block
 * types come from the segments. */
int divans_ir_lit_config(const divans_ir *ir, const divans_ir_options *o, divans_lit_config *cfg);

#ifdef __cplusplus
}
#endif
#endif
