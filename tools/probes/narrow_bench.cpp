#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <immintrin.h>
template <class SrcT>
void work_scalar(const SrcT* src, int32_t* dst, size_t lo, size_t hi, int* flag) {
    int b = 0;
    for (size_t i = lo; i < hi; ++i) {
        const SrcT v = src[i];
        b |= (v < 0) | ((long long)v > 2147483647LL);
        dst[i] = (int32_t)v;
    }
    if (b) *flag = 1;
}
__attribute__((target("avx2"))) void work_avx2(const int64_t* src, int32_t* dst, size_t lo, size_t hi, int* flag) {
    __m256i acc = _mm256_setzero_si256();
    size_t i = lo;
    const __m256i idx = _mm256_setr_epi32(0, 2, 4, 6, 1, 3, 5, 7);
    for (; i + 8 <= hi; i += 8) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 4));
        // high 32 bits must be zero and bit 31 clear: OR everything, check once
        acc = _mm256_or_si256(acc, _mm256_or_si256(a, c));
        const __m256i pa = _mm256_permutevar8x32_epi32(a, idx), pc = _mm256_permutevar8x32_epi32(c, idx);
        _mm_storeu_si128((__m128i*)(dst + i), _mm256_castsi256_si128(pa));
        _mm_storeu_si128((__m128i*)(dst + i + 4), _mm256_castsi256_si128(pc));
    }
    int b = 0;
    alignas(32) uint64_t t[4];
    _mm256_store_si256((__m256i*)t, acc);
    if ((t[0] | t[1] | t[2] | t[3]) & 0xFFFFFFFF80000000ull) b = 1;
    for (; i < hi; ++i) { const int64_t v = src[i]; b |= (v < 0) | (v > 2147483647LL); dst[i] = (int32_t)v; }
    if (b) *flag = 1;
}
int main(int argc, char** argv) {
    const size_t n = (size_t)8 << 20;  // one chunk
    const int nt = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<int64_t> src(n * 8);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (int64_t)(i % 1000);
    std::vector<int32_t> dst(n);
    for (int mode = 0; mode < 2; ++mode) {
        auto t0 = std::chrono::steady_clock::now();
        for (int c = 0; c < 8; ++c) {
            std::vector<std::thread> th; std::vector<int> flags(nt, 0);
            const size_t per = (n + nt - 1) / nt;
            for (int t = 0; t < nt; ++t) {
                const size_t lo = t * per, hi = lo + per < n ? lo + per : n;
                if (mode == 0) th.emplace_back(work_scalar<int64_t>, src.data() + c * n, dst.data(), lo, hi, &flags[t]);
                else th.emplace_back(work_avx2, src.data() + c * n, dst.data(), lo, hi, &flags[t]);
            }
            for (auto& x : th) x.join();
        }
        auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        printf("mode %d threads %d: %.2f ms for 8 chunks (%.1f GB/s read)\n", mode, nt, ms, 8.0 * n * 8 / ms / 1e6);
    }
}
