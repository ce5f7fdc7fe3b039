/*
 * xeng_cpu_fast.c -- what the HOST cores can do on the X-engine contraction: the `cpu_baseline` of bench.py (round 5).
 *
 * TEST / BENCH INFRASTRUCTURE ONLY, like xeng_oracle.c: nothing in the product package may call, link or import it.
 *
 * The parity oracle (xeng_oracle.c) is a scalar restatement built for a generic x86-64-v3 target: right for a checker, a strawman as
 * a baseline (1.7e8 cMAC/s per core).  This translation unit is the same arithmetic -- sum_t x[R] conj(x[C]) on 4+4-bit samples
 * into int32, xGPU register-tile order (corr_block.py:27-58, xgpu_test.py:76-131) -- written for the machine it runs on
 * (-march=native, built on the box that times it, never shipped as a binary): 16-bit (re, im) pairs, one `vpdpwssd` (AVX-512
 * VNNI; `vpmaddwd` + `vpaddd` without it; plain C without AVX-512) per 16 baselines and product pair, register tiles of 4 rows x
 * 32 columns over the whole time axis, OpenMP over channels.  It is checked against the scalar oracle word for word before it is
 * timed (tests/test_oracle.py; bench.py).
 *
 * Conventions as in xeng_oracle.c: row input R = i, column input C = j <= i (stand-wise: C stand <= R stand):
 *     re += re_i re_j + im_i im_j        im += im_i re_j - re_i im_j
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__AVX512F__) && defined(__AVX512BW__)
#include <immintrin.h>
#define FAST_AVX512 1
#else
#define FAST_AVX512 0
#endif

int fast_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads of the following calls (bench.py picks the count that is fastest on the host it runs on: a container's CPU quota can be far
 * below the number of cores it sees) */
void fast_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* 2: AVX-512 VNNI, 1: AVX-512 BW, 0: plain C */
int fast_isa(void) {
#if FAST_AVX512 && defined(__AVX512VNNI__)
    return 2;
#elif FAST_AVX512
    return 1;
#else
    return 0;
#endif
}

static inline int nib_hi(uint8_t d) { int v = d >> 4;  return v > 7 ? v - 16 : v; }
static inline int nib_lo(uint8_t d) { int v = d & 0xf; return v > 7 ? v - 16 : v; }
static inline int64_t tri_index(int64_t i, int64_t j) { return (i * (i + 1)) / 2 + j; }

/* word of (row input i, column input j), stand(j) <= stand(i), in a channel's plane: xeng_oracle.c orc_xgpu_correlate */
static inline int64_t word_of(int i, int j, int64_t qs) {
    const int R = i >> 1, pR = i & 1, C = j >> 1, pC = j & 1;
    const int64_t cell = (int64_t)(2 * (C & 1) + (R & 1)) * qs + tri_index(R >> 1, C >> 1);
    return cell * 4 + 2 * pR + pC;
}

#define IB 4      /* rows of a register tile */
#define JB 32     /* columns of a register tile: two vectors of 16 */

int fast_xgpu_correlate(const uint8_t *in, int32_t *acc, int ntime, int nchan, int nstand, int accumulate) {
    const int ninput = 2 * nstand;
    if (nstand % 4 != 0 || nstand <= 0 || ntime <= 0 || nchan <= 0) return -1;
    const int np = (ninput + JB - 1) / JB * JB;                 /* inputs padded to whole column tiles */
    const int nrow = (ninput + IB - 1) / IB * IB;               /* ... and to whole row strips */
    const int64_t per_chan = (int64_t)(nstand / 2 + 1) * (nstand / 4) * 16;
    const int64_t matlen = per_chan * nchan;
    const int64_t qs = ((int64_t)(nstand / 2 + 1) * nstand) / 4;
    /* V[c][t][j] = (re_j, im_j) as int16 pairs (the column operand, 16 per vector); A1[c][i][t] = (re_i, im_i), A2[c][i][t] = (im_i, -re_i) */
    const size_t vsz = (size_t)ntime * np, asz = (size_t)nrow * ntime;
    int32_t *V = NULL, *A1 = NULL, *A2 = NULL;
    if (posix_memalign((void **)&V, 64, vsz * nchan * 4) || posix_memalign((void **)&A1, 64, asz * nchan * 4) ||
        posix_memalign((void **)&A2, 64, asz * nchan * 4)) { free(V); free(A1); free(A2); return -2; }
#pragma omp parallel for schedule(static)
    for (int c = 0; c < nchan; c++) {
        int32_t *v = V + vsz * c, *a1 = A1 + asz * c, *a2 = A2 + asz * c;
        memset(v, 0, vsz * 4);
        memset(a1, 0, asz * 4);
        memset(a2, 0, asz * 4);
        for (int t = 0; t < ntime; t++) {
            const uint8_t *row = in + ((size_t)t * nchan + c) * ninput;
            for (int i = 0; i < ninput; i++) {
                const int re = nib_hi(row[i]), im = nib_lo(row[i]);
                v[(size_t)t * np + i] = (int32_t)(((uint32_t)(uint16_t)(int16_t)im << 16) | (uint16_t)(int16_t)re);
                a1[(size_t)i * ntime + t] = (int32_t)(((uint32_t)(uint16_t)(int16_t)im << 16) | (uint16_t)(int16_t)re);
                a2[(size_t)i * ntime + t] = (int32_t)(((uint32_t)(uint16_t)(int16_t)(-re) << 16) | (uint16_t)(int16_t)im);
            }
        }
    }
    /* one task per (channel, block of 32 columns): its column operand (61 KB at 480 samples) stays in the core's cache while the row
     * strips at and below the diagonal stream past; more tasks than cores, handed out dynamically (the triangle is uneven) */
    const int njb = np / JB;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int c = 0; c < nchan; c++)
        for (int jb = njb - 1; jb >= 0; jb--) {
            const int j0 = jb * JB;
            const int32_t *v = V + vsz * c, *a1 = A1 + asz * c, *a2 = A2 + asz * c;
            int32_t *out_r = acc + (size_t)c * per_chan, *out_i = acc + matlen + (size_t)c * per_chan;
            for (int i0 = j0 & ~(IB - 1); i0 < ninput; i0 += IB) {
                int32_t tr[IB][JB], ti[IB][JB];
#if FAST_AVX512
                __m512i sr[IB][2], si[IB][2];
                for (int a = 0; a < IB; a++)
                    for (int w = 0; w < 2; w++) { sr[a][w] = _mm512_setzero_si512(); si[a][w] = _mm512_setzero_si512(); }
                for (int t = 0; t < ntime; t++) {
                    const __m512i v0 = _mm512_load_si512((const void *)(v + (size_t)t * np + j0));
                    const __m512i v1 = _mm512_load_si512((const void *)(v + (size_t)t * np + j0 + 16));
                    for (int a = 0; a < IB; a++) {
                        const __m512i b1 = _mm512_set1_epi32(a1[(size_t)(i0 + a) * ntime + t]);
                        const __m512i b2 = _mm512_set1_epi32(a2[(size_t)(i0 + a) * ntime + t]);
#ifdef __AVX512VNNI__
                        sr[a][0] = _mm512_dpwssd_epi32(sr[a][0], v0, b1);
                        sr[a][1] = _mm512_dpwssd_epi32(sr[a][1], v1, b1);
                        si[a][0] = _mm512_dpwssd_epi32(si[a][0], v0, b2);
                        si[a][1] = _mm512_dpwssd_epi32(si[a][1], v1, b2);
#else
                        sr[a][0] = _mm512_add_epi32(sr[a][0], _mm512_madd_epi16(v0, b1));
                        sr[a][1] = _mm512_add_epi32(sr[a][1], _mm512_madd_epi16(v1, b1));
                        si[a][0] = _mm512_add_epi32(si[a][0], _mm512_madd_epi16(v0, b2));
                        si[a][1] = _mm512_add_epi32(si[a][1], _mm512_madd_epi16(v1, b2));
#endif
                    }
                }
                for (int a = 0; a < IB; a++)
                    for (int w = 0; w < 2; w++) {
                        _mm512_storeu_si512((void *)&tr[a][16 * w], sr[a][w]);
                        _mm512_storeu_si512((void *)&ti[a][16 * w], si[a][w]);
                    }
#else
                memset(tr, 0, sizeof(tr));
                memset(ti, 0, sizeof(ti));
                for (int t = 0; t < ntime; t++)
                    for (int a = 0; a < IB; a++) {
                        const int32_t p1 = a1[(size_t)(i0 + a) * ntime + t];
                        const int re_i = (int16_t)(p1 & 0xFFFF), im_i = (int16_t)((uint32_t)p1 >> 16);
                        for (int b = 0; b < JB; b++) {
                            const int32_t pv = v[(size_t)t * np + j0 + b];
                            const int re_j = (int16_t)(pv & 0xFFFF), im_j = (int16_t)((uint32_t)pv >> 16);
                            tr[a][b] += re_i * re_j + im_i * im_j;
                            ti[a][b] += im_i * re_j - re_i * im_j;
                        }
                    }
#endif
                /* the tile's live words: stand PAIR of j <= stand pair of i, all four (stand, stand) combinations of a pair of pairs --
                 * xeng_oracle.c's loop (Ch <= Rh, rx, ry), including the diagonal cells' words that regtile_index never addresses */
                for (int a = 0; a < IB; a++) {
                    const int i = i0 + a;
                    if (i >= ninput) break;
                    for (int b = 0; b < JB; b++) {
                        const int j = j0 + b;
                        if (j >= ninput || (j >> 2) > (i >> 2)) continue;
                        const int64_t w = word_of(i, j, qs);
                        if (accumulate) { out_r[w] += tr[a][b]; out_i[w] += ti[a][b]; }
                        else            { out_r[w]  = tr[a][b]; out_i[w]  = ti[a][b]; }
                    }
                }
            }
        }
    free(V); free(A1); free(A2);
    return 0;
}
