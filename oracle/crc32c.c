/*
 * crc32c.c -- ORACLE (test infrastructure): CRC-32C (Castagnoli), bitwise-reflected,
 * init 0 / final xor as in /root/reference/src/codec/crc32.rs:29-66 (`!crc` in, `!crc` out).
 * Pinned by the reference KATs crc32.rs:99-115.
 */
#include "divans_oracle.h"

static uint32_t table[256];
static int table_ready;

static void make_table(void) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        table[i] = c;
    }
    table_ready = 1;
}

uint32_t orc_crc32c_update(uint32_t crc, const uint8_t *buf, size_t len) {
    if (!table_ready) make_table();
    crc = ~crc;
    for (size_t i = 0; i < len; ++i) crc = table[(uint8_t)crc ^ buf[i]] ^ (crc >> 8);
    return ~crc;
}
