// see skel_match.h
#include "skel_match.h"

#include <immintrin.h>

#include <cstdlib>
#include <cstring>

namespace skel_simd {
namespace {
__attribute__((target("avx2"))) bool same_avx2(const uint8_t *buf, const uint8_t *ref, const uint8_t *mask, size_t n) {
    __m256i acc = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        __m256i x = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)(buf + i)), _mm256_loadu_si256((const __m256i *)(ref + i)));
        __m256i y = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)(buf + i + 32)), _mm256_loadu_si256((const __m256i *)(ref + i + 32)));
        x = _mm256_and_si256(x, _mm256_loadu_si256((const __m256i *)(mask + i)));
        y = _mm256_and_si256(y, _mm256_loadu_si256((const __m256i *)(mask + i + 32)));
        acc = _mm256_or_si256(acc, _mm256_or_si256(x, y));
    }
    uint8_t tail = 0;
    for (; i < n; i++) tail |= (uint8_t)((buf[i] ^ ref[i]) & mask[i]);
    return _mm256_testz_si256(acc, acc) && !tail;
}

bool same_sse2(const uint8_t *buf, const uint8_t *ref, const uint8_t *mask, size_t n) {
    __m128i acc0 = _mm_setzero_si128(), acc1 = acc0;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        __m128i x = _mm_xor_si128(_mm_loadu_si128((const __m128i *)(buf + i)), _mm_loadu_si128((const __m128i *)(ref + i)));
        __m128i y = _mm_xor_si128(_mm_loadu_si128((const __m128i *)(buf + i + 16)), _mm_loadu_si128((const __m128i *)(ref + i + 16)));
        acc0 = _mm_or_si128(acc0, _mm_and_si128(x, _mm_loadu_si128((const __m128i *)(mask + i))));
        acc1 = _mm_or_si128(acc1, _mm_and_si128(y, _mm_loadu_si128((const __m128i *)(mask + i + 16))));
    }
    uint8_t tail = 0;
    for (; i < n; i++) tail |= (uint8_t)((buf[i] ^ ref[i]) & mask[i]);
    acc0 = _mm_or_si128(acc0, acc1);
    return _mm_movemask_epi8(_mm_cmpeq_epi8(acc0, _mm_setzero_si128())) == 0xffff && !tail;
}

__attribute__((target("ssse3"))) void rows_ssse3(const uint8_t *buf, const uint32_t *off, size_t n, gc_label *dst) {
    const __m128i sw = _mm_set_epi8(8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7);
    for (size_t r = 0; r < n; r++)
        _mm_storeu_si128((__m128i *)(dst + r), _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(buf + off[r])), sw));
}

void rows_plain(const uint8_t *buf, const uint32_t *off, size_t n, gc_label *dst) {
    for (size_t r = 0; r < n; r++) {
        uint64_t a, b;
        std::memcpy(&a, buf + off[r], 8);
        std::memcpy(&b, buf + off[r] + 8, 8);
        dst[r] = gc_label{__builtin_bswap64(a), __builtin_bswap64(b)};
    }
}
}  // namespace

int level() {
    static const int v = [] {
        if (std::getenv("GC_STREAM_PLAIN_MATCH")) return 0;
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx2") ? 2 : __builtin_cpu_supports("ssse3") ? 1 : 0;
    }();
    return v;
}

bool same(const uint8_t *buf, const uint8_t *ref, const uint8_t *mask, size_t n) {
    return level() == 2 ? same_avx2(buf, ref, mask, n) : same_sse2(buf, ref, mask, n);
}

void rows(const uint8_t *buf, const uint32_t *off, size_t n, gc_label *dst) {
    if (level() >= 1) rows_ssse3(buf, off, n, dst);
    else rows_plain(buf, off, n, dst);
}
}  // namespace skel_simd
