// m3d_mt19937.hpp -- the reference's sampler stream (`rng_() % size_` on a seeded std::mt19937, utils.h:71-97) produced
// a block at a time.  Same recurrence, tempering and seeding as std::mt19937 (the tests compare the tables with the
// oracle's std-library-free restatement); the difference is only in how it is evaluated: the 624-word twist, the
// tempering and the remainder run as three plain loops over arrays that the host compiler vectorises (AVX2 / AVX-512
// clones picked at load time), ~4x the rate of drawing word by word.  Every rank of a sharded fit walks the whole
// stream (distributed.py), so this rate bounds the multi-GPU scaling of short fits.
#pragma once
#include <cstdint>

// host-only multiversioning (the driver sources also pass through the device compiler, which has no ifunc)
#if defined(__HIP_DEVICE_COMPILE__)
#define M3D_HOST_SIMD_CLONES
#else
#define M3D_HOST_SIMD_CLONES __attribute__((target_clones("default", "avx2", "avx512f")))
#endif

namespace m3d {

struct Mt19937Mod {
    static constexpr int kN = 624;
    uint32_t mt[kN];
    uint32_t out[kN];   // tempered outputs of the current block, already reduced mod d
    int pos = kN;       // next unread entry of out (kN: block exhausted)
    uint32_t d = 0, magic = 0, shift = 0;

    void seed(uint32_t s) {   // std::mt19937::seed(value)
        mt[0] = s;
        for (uint32_t i = 1; i < (uint32_t)kN; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
        pos = kN;
    }
    // remainder by an invariant 32-bit divisor without a divide: t = mulhi(magic, y), q = (((y - t) >> 1) + t) >> shift
    // (round-up method; exact for every 32-bit y and 2 <= d < 2^32)
    void set_modulus(uint32_t divisor) {
        d = divisor;
        uint32_t L = 0;
        while ((UINT64_C(1) << L) < d) ++L;
        magic = (uint32_t)(((UINT64_C(1) << 32) * ((UINT64_C(1) << L) - d)) / d + 1);
        shift = L ? L - 1 : 0;
    }
    M3D_HOST_SIMD_CLONES void refill() {
        auto tw = [](uint32_t a, uint32_t b, uint32_t c) {
            const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
            return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        };
        for (int i = 0; i < 227; ++i) mt[i] = tw(mt[i], mt[i + 1], mt[i + 397]);
        for (int i = 227; i < 623; ++i) mt[i] = tw(mt[i], mt[i + 1], mt[i - 227]);
        mt[623] = tw(mt[623], mt[0], mt[396]);
        const uint32_t dd = d, mg = magic, sh = shift;
        for (int i = 0; i < kN; ++i) {
            uint32_t y = mt[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            const uint32_t t = (uint32_t)(((uint64_t)mg * y) >> 32);
            const uint32_t q = (((y - t) >> 1) + t) >> sh;
            out[i] = dd == 1u ? 0u : y - q * dd;
        }
        pos = 0;
    }
    uint32_t next() {   // rng() % d
        if (pos == kN) refill();
        return out[pos++];
    }
    // m distinct indices per hypothesis, duplicates redrawn (utils.h:88-95)
    template <int M>
    void fill(uint32_t* s, size_t n_hyp) {
        for (size_t h = 0; h < n_hyp; ++h, s += M) {
            if (pos + M <= kN) {   // the whole sample sits in the current block: accept it if it has no repeat
                const uint32_t* p = out + pos;
                bool distinct = true;
                for (int k = 1; k < M; ++k)
                    for (int j = 0; j < k; ++j) distinct = distinct && (p[k] != p[j]);
                if (distinct) {
                    for (int k = 0; k < M; ++k) s[k] = p[k];
                    pos += M;
                    continue;
                }
            }
            int valid = 0;
            while (valid < M) {
                const uint32_t idx = next();
                bool dup = false;
                for (int k = 0; k < valid; ++k) dup = dup || (s[k] == idx);
                if (!dup) s[valid++] = idx;
            }
        }
    }
};

}  // namespace m3d
