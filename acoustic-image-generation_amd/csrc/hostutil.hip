// Host-side helpers of libacimg that are not kernels: CRC-32C (Castagnoli), the checksum TensorFlow's
// checkpoint bundles (tensor_bundle.cc) and TFRecord files (record_writer.cc) carry.  Slicing-by-8 tables.
#include "common.hpp"

namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
}  // namespace

extern "C" uint32_t acimg_crc32c(const void* data, size_t n, uint32_t crc) {
    static const Crc32cTables T;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~crc;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
        --n;
    }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = T.t[7][v & 0xFF] ^ T.t[6][(v >> 8) & 0xFF] ^ T.t[5][(v >> 16) & 0xFF] ^ T.t[4][(v >> 24) & 0xFF] ^
            T.t[3][(v >> 32) & 0xFF] ^ T.t[2][(v >> 40) & 0xFF] ^ T.t[1][(v >> 48) & 0xFF] ^ T.t[0][v >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}
