// Host micro-benchmark behind the evaluator's byte-skeleton match (stream_engine.cpp, EvalSkel): how fast can one core
// decide "this block equals the reference block outside its global ids and table rows" and move the rows out?
//   (a) the chunk list walked with 8-byte compares and one bswap pair per row   (the round-3 form)
//   (b) one masked 32-byte compare over the whole block (AVX2) + rows moved with pshufb
// The block mimics a streamed 32 x 32-bit multiplier: 7-byte gate headers, 30 % of the gates with two 16-byte rows, a global id
// every 60 gates; the stream is 512 MiB of such blocks (rows random), so the block itself comes from DRAM as in a real run.
//   g++ -O2 -o tools/skel_match_ubench tools/skel_match_ubench.cpp && tools/skel_match_ubench
#include <immintrin.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

struct Chunk {
    uint32_t cmp;
    uint16_t skip, nrows;
};

static inline uint64_t be64(const uint8_t *p) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

static bool match_chunks(const std::vector<Chunk> &ch, const uint8_t *ref, const uint8_t *buf) {
    const uint8_t *p = buf, *q = ref;
    for (const Chunk &c : ch) {
        uint64_t acc = 0;
        uint32_t i = 0;
        for (; i + 8 <= c.cmp; i += 8) {
            uint64_t x, y;
            std::memcpy(&x, p + i, 8);
            std::memcpy(&y, q + i, 8);
            acc |= x ^ y;
        }
        if (i < c.cmp) {
            if (c.cmp >= 8) {
                uint64_t x, y;
                std::memcpy(&x, p + c.cmp - 8, 8);
                std::memcpy(&y, q + c.cmp - 8, 8);
                acc |= x ^ y;
            } else
                for (; i < c.cmp; i++) acc |= (uint64_t)(p[i] ^ q[i]);
        }
        if (acc) return false;
        p += c.cmp + c.skip + 16u * c.nrows;
        q += c.cmp + c.skip + 16u * c.nrows;
    }
    return true;
}

static void rows_chunks(const std::vector<Chunk> &ch, const uint8_t *buf, uint64_t *dst) {
    const uint8_t *p = buf;
    for (const Chunk &c : ch) {
        p += c.cmp + c.skip;
        for (uint32_t r = 0; r < c.nrows; r++, p += 16) {
            *dst++ = be64(p);
            *dst++ = be64(p + 8);
        }
    }
}

__attribute__((target("avx2"))) static bool match_masked(const uint8_t *ref, const uint8_t *mask, const uint8_t *buf, size_t n) {
    __m256i acc = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        __m256i a = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)(buf + i)), _mm256_loadu_si256((const __m256i *)(ref + i)));
        __m256i b = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)(buf + i + 32)), _mm256_loadu_si256((const __m256i *)(ref + i + 32)));
        a = _mm256_and_si256(a, _mm256_loadu_si256((const __m256i *)(mask + i)));
        b = _mm256_and_si256(b, _mm256_loadu_si256((const __m256i *)(mask + i + 32)));
        acc = _mm256_or_si256(acc, _mm256_or_si256(a, b));
    }
    uint8_t tail = 0;
    for (; i < n; i++) tail |= (uint8_t)((buf[i] ^ ref[i]) & mask[i]);
    return _mm256_testz_si256(acc, acc) && !tail;
}

static bool match_masked_sse2(const uint8_t *ref, const uint8_t *mask, const uint8_t *buf, size_t n) {
    __m128i acc0 = _mm_setzero_si128(), acc1 = acc0;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        __m128i a = _mm_xor_si128(_mm_loadu_si128((const __m128i *)(buf + i)), _mm_loadu_si128((const __m128i *)(ref + i)));
        __m128i b = _mm_xor_si128(_mm_loadu_si128((const __m128i *)(buf + i + 16)), _mm_loadu_si128((const __m128i *)(ref + i + 16)));
        acc0 = _mm_or_si128(acc0, _mm_and_si128(a, _mm_loadu_si128((const __m128i *)(mask + i))));
        acc1 = _mm_or_si128(acc1, _mm_and_si128(b, _mm_loadu_si128((const __m128i *)(mask + i + 16))));
    }
    uint8_t tail = 0;
    for (; i < n; i++) tail |= (uint8_t)((buf[i] ^ ref[i]) & mask[i]);
    acc0 = _mm_or_si128(acc0, acc1);
    return _mm_movemask_epi8(_mm_cmpeq_epi8(acc0, _mm_setzero_si128())) == 0xffff && !tail;
}

__attribute__((target("ssse3"))) static void rows_pshufb(const uint32_t *off, uint32_t n, const uint8_t *buf, uint64_t *dst) {
    const __m128i sw = _mm_set_epi8(8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7);
    for (uint32_t r = 0; r < n; r++)
        _mm_storeu_si128((__m128i *)(dst + 2 * (size_t)r), _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(buf + off[r])), sw));
}

int main() {
    std::mt19937_64 g(1);
    const uint32_t ngates = 4000;
    std::vector<Chunk> ch;
    std::vector<uint8_t> ref, mask;
    std::vector<uint32_t> row_off;
    uint32_t run = 0;
    auto cut = [&](uint32_t skip, uint32_t rows) {
        ch.push_back(Chunk{run, (uint16_t)skip, (uint16_t)rows});
        run = 0;
        for (uint32_t k = 0; k < skip; k++) ref.push_back(0), mask.push_back(0);
        for (uint32_t r = 0; r < rows; r++) {
            row_off.push_back((uint32_t)ref.size());
            for (int k = 0; k < 16; k++) ref.push_back(0), mask.push_back(0);
        }
    };
    for (uint32_t i = 0; i < ngates; i++) {
        const bool is_and = g() % 100 < 30, glob = i % 60 == 0;
        const uint32_t hdr = glob ? 5 : 7;
        for (uint32_t k = 0; k < hdr; k++) ref.push_back((uint8_t)g()), mask.push_back(0xff);
        run += hdr;
        if (glob || is_and) cut(glob ? 2 : 0, is_and ? 2 : 0);
    }
    if (run) cut(0, 0);
    const size_t nb = ref.size(), nrows = row_off.size();
    const size_t nblocks = ((size_t)512 << 20) / nb;
    std::vector<uint8_t> stream(nblocks * nb);
    for (size_t b = 0; b < nblocks; b++) {
        std::memcpy(&stream[b * nb], ref.data(), nb);
        for (size_t r = 0; r < nrows; r++) {
            uint64_t x = g(), y = g();
            std::memcpy(&stream[b * nb + row_off[r]], &x, 8);
            std::memcpy(&stream[b * nb + row_off[r] + 8], &y, 8);
        }
    }
    std::vector<uint64_t> out(2 * nrows + 2);
    std::printf("block %zu bytes, %zu chunks, %zu rows; %zu blocks (%.0f MiB)\n", nb, ch.size(), nrows, nblocks, (double)stream.size() / (1 << 20));
    const bool avx2 = __builtin_cpu_supports("avx2"), ssse3 = __builtin_cpu_supports("ssse3");
    uint64_t sink = 0;
    for (int form = 0; form < 4; form++) {
        const auto t0 = std::chrono::steady_clock::now();
        size_t ok = 0;
        for (size_t b = 0; b < nblocks; b++) {
            const uint8_t *buf = &stream[b * nb];
            bool same = (form & 1) ? (avx2 ? match_masked(ref.data(), mask.data(), buf, nb) : match_masked_sse2(ref.data(), mask.data(), buf, nb)) : match_chunks(ch, ref.data(), buf);
            ok += same;
            if (form >= 2) {
                if (ssse3) rows_pshufb(row_off.data(), (uint32_t)nrows, buf, out.data());
                else rows_chunks(ch, buf, out.data());
            } else
                rows_chunks(ch, buf, out.data());
            sink += out[b % (2 * nrows)];
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("compare %-8s rows %-8s  %6.2f us per block  %6.2f GB/s   (%zu matched, sink %llx)\n", (form & 1) ? (avx2 ? "mask256" : "mask128") : "chunks",
                    form >= 2 && ssse3 ? "pshufb" : "bswap", s / (double)nblocks * 1e6, (double)stream.size() / s / 1e9, ok, (unsigned long long)sink);
    }
    return 0;
}
