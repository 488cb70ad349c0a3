// mono_pack.cpp — is an RGB24 frame a mono camera's (R = G = B everywhere)?  If so, its 8-bit plane.
//
// The reference's camera format is RGB24 (Image<RGB24Pixel>: VideoLib/image.h, customcam.h) and its first step is the sum
// b + g + r per pixel (Image<float>::ConvertRGB2BW, image.h:197-203).  A mono camera — EuRoC, TUM-VI, most VIO rigs — delivers one
// byte per pixel and the application (or DataSetCam, datasetcam.cpp:109-171) triples it to fit the surface.  On a GPU host the tripled
// frame is what crosses PCIe, and at batch-group sizes the link is the bound: the group sends the 8-bit plane
// instead (edgehip_upload_grey8_pinned: the device forms 3 g, the same sum) whenever EVERY frame of a step is mono.  The test and
// the packing are one pass over the frame on the application's thread (releaseCustomCamBuffer), 16 pixels per iteration with SSSE3
// byte shuffles where the CPU has them; a colour frame leaves the loop at its first coloured pixel.
#include <cstddef>
#include <cstdint>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define REBVO_X86 1
#endif

namespace {

// pixels [first, n): scalar
bool pack_scalar(const uint8_t *rgb, size_t first, size_t n, uint8_t *grey) {
    for (size_t i = first; i < n; i++) {
        const uint8_t a = rgb[3 * i], b = rgb[3 * i + 1], c = rgb[3 * i + 2];
        if (a != b || b != c) return false;
        grey[i] = b;
    }
    return true;
}

#ifdef REBVO_X86
// 16 pixels = 48 bytes per iteration.  Byte j of the stream belongs to channel j % 3: three loads v0 v1 v2, and for each channel a
// shuffle of each load that drops its bytes of that channel into their place (zero elsewhere: index 0x80), OR-ed together.
__attribute__((target("ssse3"))) size_t pack_ssse3(const uint8_t *rgb, size_t n, uint8_t *grey, bool &mono) {
    const char Z = (char)0x80;
    // channel 0 (bytes 0, 3, 6, ...): v0 holds pixels 0..5 (bytes 0..15 -> 0,3,6,9,12,15), v1 pixels 6..10 (bytes 16..31 -> 18,21,24,27,30
    // = local 2,5,8,11,14), v2 pixels 11..15 (bytes 32..47 -> 33,36,39,42,45 = local 1,4,7,10,13)
    const __m128i a0 = _mm_setr_epi8(0, 3, 6, 9, 12, 15, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z);
    const __m128i a1 = _mm_setr_epi8(Z, Z, Z, Z, Z, Z, 2, 5, 8, 11, 14, Z, Z, Z, Z, Z);
    const __m128i a2 = _mm_setr_epi8(Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, 1, 4, 7, 10, 13);
    // channel 1 (bytes 1, 4, 7, ...): v0 local 1,4,7,10,13 (pixels 0..4), v1 local 0,3,6,9,12,15 (pixels 5..10), v2 local 2,5,8,11,14 (11..15)
    const __m128i b0 = _mm_setr_epi8(1, 4, 7, 10, 13, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z);
    const __m128i b1 = _mm_setr_epi8(Z, Z, Z, Z, Z, 0, 3, 6, 9, 12, 15, Z, Z, Z, Z, Z);
    const __m128i b2 = _mm_setr_epi8(Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, 2, 5, 8, 11, 14);
    // channel 2 (bytes 2, 5, 8, ...): v0 local 2,5,8,11,14 (pixels 0..4), v1 local 1,4,7,10,13 (5..9), v2 local 0,3,6,9,12,15 (10..15)
    const __m128i c0 = _mm_setr_epi8(2, 5, 8, 11, 14, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z);
    const __m128i c1 = _mm_setr_epi8(Z, Z, Z, Z, Z, 1, 4, 7, 10, 13, Z, Z, Z, Z, Z, Z);
    const __m128i c2 = _mm_setr_epi8(Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, 0, 3, 6, 9, 12, 15);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i v0 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(rgb + 3 * i));
        const __m128i v1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(rgb + 3 * i + 16));
        const __m128i v2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(rgb + 3 * i + 32));
        const __m128i ch0 = _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(v0, a0), _mm_shuffle_epi8(v1, a1)), _mm_shuffle_epi8(v2, a2));
        const __m128i ch1 = _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(v0, b0), _mm_shuffle_epi8(v1, b1)), _mm_shuffle_epi8(v2, b2));
        const __m128i ch2 = _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(v0, c0), _mm_shuffle_epi8(v1, c1)), _mm_shuffle_epi8(v2, c2));
        const __m128i eq = _mm_and_si128(_mm_cmpeq_epi8(ch0, ch1), _mm_cmpeq_epi8(ch1, ch2));
        if (_mm_movemask_epi8(eq) != 0xffff) { mono = false; return i; }
        _mm_storeu_si128(reinterpret_cast<__m128i *>(grey + i), ch1);
    }
    mono = true;
    return i;
}

// 32 pixels = 96 bytes per iteration: the two halves of a 256-bit register take two consecutive 48-byte groups (bytes 0..15 | 48..63,
// 16..31 | 64..79, 32..47 | 80..95), the byte shuffles work inside each half with the masks above, and the channel comes out as 32
// contiguous pixels.  Half the shuffles per pixel: the pass is bound by them, not by memory (90 -> ~50 us per 752x480 frame).
__attribute__((target("avx2"))) size_t pack_avx2(const uint8_t *rgb, size_t n, uint8_t *grey, bool &mono) {
    const char Z = (char)0x80;
#define REBVO_M2(...) _mm256_setr_epi8(__VA_ARGS__, __VA_ARGS__)
    const __m256i a0 = REBVO_M2(0, 3, 6, 9, 12, 15, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z), a1 = REBVO_M2(Z, Z, Z, Z, Z, Z, 2, 5, 8, 11, 14, Z, Z, Z, Z, Z),
                  a2 = REBVO_M2(Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, 1, 4, 7, 10, 13);
    const __m256i b0 = REBVO_M2(1, 4, 7, 10, 13, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z), b1 = REBVO_M2(Z, Z, Z, Z, Z, 0, 3, 6, 9, 12, 15, Z, Z, Z, Z, Z),
                  b2 = REBVO_M2(Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, 2, 5, 8, 11, 14);
    const __m256i c0 = REBVO_M2(2, 5, 8, 11, 14, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, Z), c1 = REBVO_M2(Z, Z, Z, Z, Z, 1, 4, 7, 10, 13, Z, Z, Z, Z, Z, Z),
                  c2 = REBVO_M2(Z, Z, Z, Z, Z, Z, Z, Z, Z, Z, 0, 3, 6, 9, 12, 15);
#undef REBVO_M2
#define REBVO_LOAD2(lo, hi) _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128(reinterpret_cast<const __m128i *>(lo))), \
                                                    _mm_loadu_si128(reinterpret_cast<const __m128i *>(hi)), 1)
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const uint8_t *p = rgb + 3 * i;
        const __m256i v0 = REBVO_LOAD2(p, p + 48), v1 = REBVO_LOAD2(p + 16, p + 64), v2 = REBVO_LOAD2(p + 32, p + 80);
        const __m256i ch0 = _mm256_or_si256(_mm256_or_si256(_mm256_shuffle_epi8(v0, a0), _mm256_shuffle_epi8(v1, a1)), _mm256_shuffle_epi8(v2, a2));
        const __m256i ch1 = _mm256_or_si256(_mm256_or_si256(_mm256_shuffle_epi8(v0, b0), _mm256_shuffle_epi8(v1, b1)), _mm256_shuffle_epi8(v2, b2));
        const __m256i ch2 = _mm256_or_si256(_mm256_or_si256(_mm256_shuffle_epi8(v0, c0), _mm256_shuffle_epi8(v1, c1)), _mm256_shuffle_epi8(v2, c2));
        const __m256i eq = _mm256_and_si256(_mm256_cmpeq_epi8(ch0, ch1), _mm256_cmpeq_epi8(ch1, ch2));
        if (_mm256_movemask_epi8(eq) != -1) { mono = false; return i; }
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(grey + i), ch1);
    }
#undef REBVO_LOAD2
    mono = true;
    return i;
}
#endif

}  // namespace

// 1: every pixel of the npix-pixel RGB24 frame has R = G = B and grey[0..npix) holds that byte; 0: a coloured pixel was found
// (grey is then partly written and means nothing).
extern "C" int rebvo_pack_mono(const uint8_t *rgb, size_t npix, uint8_t *grey) {
    size_t done = 0;
#ifdef REBVO_X86
    if (__builtin_cpu_supports("avx2")) {
        bool mono = true;
        done = pack_avx2(rgb, npix, grey, mono);
        if (!mono) return 0;
    } else if (__builtin_cpu_supports("ssse3")) {
        bool mono = true;
        done = pack_ssse3(rgb, npix, grey, mono);
        if (!mono) return 0;
    }
#endif
    return pack_scalar(rgb, done, npix, grey) ? 1 : 0;
}
