/* Round trip through the per-stream C ABI from plain C, with the calling pattern of the reference's c/example.c:25-80
 * (encode everything, flush until SUCCESS with a fixed output buffer, decode until SUCCESS).
 * usage: ffi_roundtrip <file> <out.divans> [selector=value ...]   exit 0 on success */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "divans_ffi.h"

#define BUF_SIZE 65536
static size_t n_alloc, n_free;
static void *count_alloc(void *opaque, size_t n) { (void)opaque; ++n_alloc; return calloc(1, n); }
static void count_free(void *opaque, void *p) { (void)opaque; ++n_free; free(p); }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); size_t len = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *data = malloc(len ? len : 1);
    if (fread(data, 1, len, f) != len) return 2;
    fclose(f);
    struct CAllocator alloc = {count_alloc, count_free, NULL};
    struct DivansCompressorState *cs = divans_new_compressor_with_custom_alloc(alloc);
    for (int i = 3; i < argc; ++i) {
        unsigned sel, val;
        if (sscanf(argv[i], "%u=%u", &sel, &val) == 2 && divans_set_option(cs, (DivansOptionSelect)sel, val) != DIVANS_SUCCESS) return 3;
    }
    unsigned char *coded = malloc(2 * len + 65536); size_t coded_len = 0;
    unsigned char buf[BUF_SIZE];
    const unsigned char *p = data; size_t left = len;
    while (left) {
        size_t roff = 0, woff = 0;
        DivansResult r = divans_encode(cs, p, left, &roff, buf, sizeof(buf), &woff);
        if (r == DIVANS_FAILURE) return 4;
        p += roff; left -= roff;
        memcpy(coded + coded_len, buf, woff); coded_len += woff;
    }
    DivansResult r;
    do {
        size_t woff = 0;
        r = divans_encode_flush(cs, buf, sizeof(buf), &woff);
        if (r == DIVANS_FAILURE) return 5;
        memcpy(coded + coded_len, buf, woff); coded_len += woff;
    } while (r != DIVANS_SUCCESS);
    /* options are refused once encoding has started (OptionStage, src/ffi/compressor.rs:63-66) */
    if (divans_set_option(cs, DIVANS_OPTION_WINDOW_SIZE, 20) != DIVANS_FAILURE) return 6;
    divans_free_compressor(cs);
    f = fopen(argv[2], "wb"); fwrite(coded, 1, coded_len, f); fclose(f);

    struct DivansDecompressorState *ds = divans_new_decompressor_with_custom_alloc(alloc, 0, 0);
    unsigned char *back = malloc(len + 1); size_t back_len = 0;
    p = coded; left = coded_len;
    do {
        size_t roff = 0, woff = 0;
        size_t feed = left < 4099 ? left : 4099;             /* odd-sized input pieces */
        r = divans_decode(ds, p, feed, &roff, buf, sizeof(buf), &woff);
        if (r == DIVANS_FAILURE || (r == DIVANS_NEEDS_MORE_INPUT && left == 0)) return 7;
        p += roff; left -= roff;
        if (back_len + woff > len) return 8;
        memcpy(back + back_len, buf, woff); back_len += woff;
    } while (r != DIVANS_SUCCESS);
    divans_free_decompressor(ds);
    if (back_len != len || memcmp(back, data, len) != 0) return 9;
    if (n_alloc != 2 || n_free != 2) return 10;               /* both states went through the custom allocator */
    /* truncated stream must not succeed (c/example.c:69) */
    ds = divans_new_decompressor_with_custom_alloc(alloc, 0, 0);
    { size_t roff = 0, woff = 0; r = divans_decode(ds, coded, coded_len - 5, &roff, buf, sizeof(buf), &woff); }
    if (r == DIVANS_SUCCESS) return 11;
    divans_free_decompressor(ds);
    /* corrupted payload byte must fail the CRC */
    if (coded_len > 40) {
        coded[30] ^= 0x40;
        ds = divans_new_decompressor_with_custom_alloc(alloc, 0, 0);
        size_t roff = 0, woff = 0; r = divans_decode(ds, coded, coded_len, &roff, buf, sizeof(buf), &woff);
        if (r != DIVANS_FAILURE) return 12;
        divans_free_decompressor(ds);
    }
    printf("File length %zu reduced to %zu\n", len, coded_len);
    free(back); free(coded); free(data);
    return 0;
}
