// Legacy Keccak-256 (the pre-standard padding 0x01 ... 0x80 of sha3.NewLegacyKeccak256, not SHA-3's 0x06), one message
// per lane: the hash of pairing/bn254's expand_message_xmd (point.go:289-340).  Rate 136 bytes, digest 32 bytes.
// All lanes of a launch hash inputs of the same length, so control flow is uniform.
#pragma once
#include "hd.h"

namespace kyb {

KYB_HD uint64_t keccak_rol(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

KYB_HD_NOINLINE void keccak_f1600(uint64_t (&a)[25]) {  // lane (x, y) at a[x + 5 y]
    constexpr uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            const uint64_t d = c[(x + 4) % 5] ^ keccak_rol(c[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = keccak_rol(a[x + 5 * y], ROT[x + 5 * y]);
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}

struct Keccak256 {
    uint64_t a[25];
    uint32_t pos;  // bytes absorbed into the current block
    KYB_HD void init() {
        for (int i = 0; i < 25; i++) a[i] = 0;
        pos = 0;
    }
    KYB_HD void put(uint8_t b) {
        a[pos >> 3] ^= (uint64_t)b << (8 * (pos & 7));
        if (++pos == 136) {
            keccak_f1600(a);
            pos = 0;
        }
    }
    KYB_HD void update(const uint8_t* p, size_t n) {
        for (size_t i = 0; i < n; i++) put(p[i]);
    }
    // digest as 32 bytes
    KYB_HD void finish(uint8_t (&out)[32]) {
        a[pos >> 3] ^= (uint64_t)0x01 << (8 * (pos & 7));
        a[16] ^= 0x8000000000000000ull;  // last byte of the 136-byte rate
        keccak_f1600(a);
        for (int i = 0; i < 32; i++) out[i] = (uint8_t)(a[i >> 3] >> (8 * (i & 7)));
    }
};

}  // namespace kyb
