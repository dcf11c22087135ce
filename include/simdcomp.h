/*
 * simdcomp.h -- drop-in declarations of lemire/simdcomp's 128-integer block
 * codec API, implemented by libsimdcomp_hip.so on an AMD MI355X.
 *
 * A program written against the reference's umbrella header
 * (include/simdcomp.h:8-21 there) compiles against this one unchanged and links
 * with -lsimdcomp_hip instead of -lsimdcomp.  Pointers are HOST pointers, as in
 * the reference; every data-touching routine stages its operands to the GPU,
 * runs the HIP kernel and copies the result back before returning, so results
 * are bit-identical to the reference's but a single 128-integer call costs a
 * kernel launch plus two PCIe copies.  Callers with many blocks should use the
 * *_length routines here (pipelined in 64 MiB pieces) or, better, keep data on
 * the device and call the batched entry points of simdcomp_hip.h.
 *
 * Threads: as in the reference (stateless, src/simdbitpacking.c:14379-14399),
 * any number of threads may call at the same time on disjoint buffers; every
 * calling thread works with streams and staging buffers of its own.
 * Devices: a call runs on the HIP device that is current in the calling thread
 * and leaves that selection untouched.  With SIMDCOMP_HIP_DEVICES=0,1,... in
 * the environment the calling threads are spread over the listed GPUs instead
 * (first thread -> first entry, and so on, round robin).
 *
 * Differences a caller can observe, all outside the reference's contract:
 *   - if no usable GPU is present, or a HIP call fails, the routine calls the
 *     handler set with simdcomp_hip_set_error_handler (simdcomp_hip.h) or, by
 *     default, prints the HIP error to stderr and aborts (the reference API
 *     has no error channel); it never computes on the CPU instead;
 *   - simdpackFOR / simdunpackFOR accept unaligned pointers (the reference's
 *     need 16-byte alignment, src/simdfor.c:32,418).
 *
 * Each prototype cites the reference declaration it replaces.
 */
#ifndef SIMDCOMP_H_
#define SIMDCOMP_H_

#include <emmintrin.h> /* __m128i: the compressed stream is addressed in 16-byte vectors */
#include <string.h>    /* the reference's headers bring it in (simdbitpacking.h:13); its callers rely on that */
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SIMDBlockSize = 128 }; /* include/simdcomputil.h:22 */

/* ---- include/simdcomputil.h ------------------------------------------------ */
uint32_t bits(const uint32_t v);                                        /* :14 */
uint32_t maxbits(const uint32_t *begin);                                /* :17 */
uint32_t maxbits_length(const uint32_t *in, uint32_t length);           /* :20 */
uint32_t simdmin(const uint32_t *in);                                   /* :25 */
uint32_t simdmin_length(const uint32_t *in, uint32_t length);           /* :28 */
void simdmaxmin_length(const uint32_t *in, uint32_t length, uint32_t *getmin, uint32_t *getmax); /* :33 */
void simdmaxmin(const uint32_t *in, uint32_t *getmin, uint32_t *getmax);                         /* :37 */
uint32_t simdmaxbitsd1(uint32_t initvalue, const uint32_t *in);                                  /* :43 */
uint32_t simdmaxbitsd1_length(uint32_t initvalue, const uint32_t *in, uint32_t length);          /* :47 */

/* ---- include/simdbitpacking.h ---------------------------------------------- */
void simdpack(const uint32_t *in, __m128i *out, const uint32_t bit);             /* :24 */
void simdpackwithoutmask(const uint32_t *in, __m128i *out, const uint32_t bit);  /* :28 */
void simdunpack(const __m128i *in, uint32_t *out, const uint32_t bit);           /* :31 */
int simdpack_compressedbytes(int length, const uint32_t bit);                    /* :35 */
__m128i *simdpack_length(const uint32_t *in, size_t length, __m128i *out, const uint32_t bit);         /* :43 */
const __m128i *simdunpack_length(const __m128i *in, size_t length, uint32_t *out, const uint32_t bit); /* :51 */
__m128i *simdpack_shortlength(const uint32_t *in, int length, __m128i *out, const uint32_t bit);       /* :60 */
const __m128i *simdunpack_shortlength(const __m128i *in, int length, uint32_t *out,
                                      const uint32_t bit);                                             /* :68 */
void simdfastset(__m128i *in128, uint32_t b, uint32_t value, size_t index);                            /* :73 */

/* ---- include/simdintegratedbitpacking.h ------------------------------------ */
void simdpackd1(uint32_t initvalue, const uint32_t *in, __m128i *out, const uint32_t bit);            /* :24 */
void simdpackwithoutmaskd1(uint32_t initvalue, const uint32_t *in, __m128i *out, const uint32_t bit); /* :30 */
void simdunpackd1(uint32_t initvalue, const __m128i *in, uint32_t *out, const uint32_t bit);          /* :34 */
/* single-block accessors (host routines; batched device forms: simdcomp_hip_select / _search / _scand1) */
int simdsearchd1(__m128i *initOffset, const __m128i *in, uint32_t bit, uint32_t key, uint32_t *presult); /* :45 */
int simdsearchwithlengthd1(uint32_t initvalue, const __m128i *in, uint32_t bit, int length, uint32_t key,
                           uint32_t *presult);                                                         /* :57 */
uint32_t simdselectd1(uint32_t initvalue, const __m128i *in, uint32_t bit, int slot);                  /* :62 */
void simdfastsetd1fromprevious(__m128i *in, uint32_t bit, uint32_t previousvalue, uint32_t value,
                               size_t index);                                                          /* :71 */
void simdfastsetd1(uint32_t initvalue, __m128i *in, uint32_t bit, uint32_t value, size_t index);      /* :81 */
void simdscand1(__m128i *initOffset, const __m128i *in, uint32_t bit);                                 /* :90 */

/* ---- include/simdfor.h ------------------------------------------------------ */
void simdpackFOR(uint32_t initvalue, const uint32_t *in, __m128i *out, const uint32_t bit);   /* :20 */
void simdunpackFOR(uint32_t initvalue, const __m128i *in, uint32_t *out, const uint32_t bit); /* :24 */
int simdpackFOR_compressedbytes(int length, const uint32_t bit);                              /* :29 */
__m128i *simdpackFOR_length(uint32_t initvalue, const uint32_t *in, int length, __m128i *out,
                            const uint32_t bit);                                              /* :35 */
const __m128i *simdunpackFOR_length(uint32_t initvalue, const __m128i *in, int length, uint32_t *out,
                                    const uint32_t bit);                                      /* :42 */
uint32_t simdselectFOR(uint32_t initvalue, const __m128i *in, uint32_t bit, int slot);        /* :48 */
void simdfastsetFOR(uint32_t initvalue, __m128i *in, uint32_t bit, uint32_t value, size_t index); /* :53 */
int simdsearchwithlengthFOR(uint32_t initvalue, const __m128i *in, uint32_t bit, int length, uint32_t key,
                            uint32_t *presult);                                               /* :65 */

/* ---- include/avxbitpacking.h, include/avx512bitpacking.h ---------------------------
 * 256- / 512-integer blocks with 8 / 16 interleaved lanes (data packed by the reference's AVX paths).  The vector
 * types exist only when the including program is compiled for those ISAs, as in the reference's headers; other
 * programs see the same functions with untyped pointers. */
enum { AVXBlockSize = 256 };    /* avxbitpacking.h:19 */
enum { AVX512BlockSize = 512 }; /* avx512bitpacking.h */
#if defined(__AVX2__) && !defined(SIMDCOMP_HIP_BUILDING)
#include <immintrin.h>
uint32_t avxmaxbits(const uint32_t *begin);                                    /* avxbitpacking.h:22 */
void avxpack(const uint32_t *in, __m256i *out, const uint32_t bit);            /* :25 */
void avxpackwithoutmask(const uint32_t *in, __m256i *out, const uint32_t bit); /* :28 */
void avxunpack(const __m256i *in, uint32_t *out, const uint32_t bit);          /* :31 */
#else
uint32_t avxmaxbits(const uint32_t *begin);
void avxpack(const uint32_t *in, void *out, const uint32_t bit);
void avxpackwithoutmask(const uint32_t *in, void *out, const uint32_t bit);
void avxunpack(const void *in, uint32_t *out, const uint32_t bit);
#endif
#if defined(__AVX512F__) && !defined(SIMDCOMP_HIP_BUILDING)
#include <immintrin.h>
uint32_t avx512maxbits(const uint32_t *begin);                                       /* avx512bitpacking.h */
void avx512pack(const uint32_t *in, __m512i *out, const uint32_t bit);
void avx512packwithoutmask(const uint32_t *in, __m512i *out, const uint32_t bit);
void avx512unpack(const __m512i *in, uint32_t *out, const uint32_t bit);
#else
uint32_t avx512maxbits(const uint32_t *begin);
void avx512pack(const uint32_t *in, void *out, const uint32_t bit);
void avx512packwithoutmask(const uint32_t *in, void *out, const uint32_t bit);
void avx512unpack(const void *in, uint32_t *out, const uint32_t bit);
#endif

#ifdef __cplusplus
}
#endif
#endif /* SIMDCOMP_H_ */
