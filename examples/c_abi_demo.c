/* Plain-C client of libxeng (include/xeng.h): the same calls a cgo / JNI / ctypes binding would make.
 *
 * Replays the reference's golden input (tests/golden/in_8t_4c_16s_2p_deadbeef.dat, written by
 * verification/make_golden_inputs.py) through the X-engine with the reference's call sequence
 * (xgpu_test.py:76-89: Initialize, Kernel per gulp with doDump on the last, GetOrder, Reorder) and compares
 * every visibility with the golden file, in the x[s0,p0]*conj(x[s1,p1]) convention
 * (corr_output_full_block.py:582-591).  No HIP headers, no Python: cc examples/c_abi_demo.c -lxeng.
 *
 * usage: c_abi_demo <in.dat> <corr.dat>      (files with a one-line JSON header, as the generator writes them)
 */
#include <complex.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/xeng.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != XENG_STATUS_SUCCESS) {                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, xengGetLastError());                   \
            return 2;                                                                            \
        }                                                                                        \
    } while (0)

static void *read_payload(const char *path, size_t *nbytes) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    int c;
    while ((c = fgetc(f)) != EOF && c != '\n') {}          /* skip the JSON header line */
    long start = ftell(f);
    fseek(f, 0, SEEK_END);
    long end = ftell(f);
    fseek(f, start, SEEK_SET);
    *nbytes = (size_t)(end - start);
    void *buf = malloc(*nbytes);
    if (fread(buf, 1, *nbytes, f) != *nbytes) { fclose(f); free(buf); return NULL; }
    fclose(f);
    return buf;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.dat corr.dat\n", argv[0]); return 2; }
    enum { NTIME = 8, NCHAN = 4, NSTAND = 16, NPOL = 2, ACC = 4, GULP = 2 };   /* BASELINE config 1 */
    size_t nin, ngold;
    uint8_t *vin = read_payload(argv[1], &nin);
    double complex *gold = read_payload(argv[2], &ngold);
    if (!vin || !gold) return 2;
    const size_t gulp_bytes = (size_t)GULP * NCHAN * NSTAND * NPOL;
    if (nin != (size_t)NTIME * NCHAN * NSTAND * NPOL ||
        ngold != sizeof(double complex) * (NTIME / ACC) * NCHAN * NSTAND * NSTAND * NPOL * NPOL) {
        fprintf(stderr, "unexpected file sizes %zu %zu\n", nin, ngold);
        return 2;
    }
    printf("%s\n", xengVersion());
    CHECK(xengXgpuConfigure(NSTAND, NPOL, NCHAN, GULP, 0));
    CHECK(xengXgpuInitialize(0));
    int64_t matlen;
    CHECK(xengXgpuGetInfo(NULL, NULL, NULL, NULL, &matlen, NULL));
    void *din, *dout;
    CHECK(xengMalloc(&din, nin, XENG_SPACE_CUDA));
    CHECK(xengMalloc(&dout, (size_t)matlen * 8, XENG_SPACE_CUDA));
    CHECK(xengMemcpy(din, vin, nin));

    int32_t a2i[NSTAND * NPOL];
    for (int k = 0; k < NSTAND * NPOL; k++) a2i[k] = k;
    const size_t nmap = (size_t)NSTAND * NSTAND * NPOL * NPOL;
    int32_t *bl = malloc(nmap * 4), *cj = malloc(nmap * 4);
    CHECK(xengXgpuGetOrder(a2i, bl, cj));
    int32_t *xg = malloc((size_t)matlen * 8), *ro = malloc(nmap * NCHAN * 2 * 4);
    long bad = 0, checked = 0;
    for (int it = 0; it < NTIME / ACC; it++) {
        for (int g = 0; g < ACC / GULP; g++)
            CHECK(xengXgpuKernel((uint8_t *)din + ((size_t)it * (ACC / GULP) + g) * gulp_bytes, dout, g == ACC / GULP - 1));
        CHECK(xengMemcpy(xg, dout, (size_t)matlen * 8));
        CHECK(xengXgpuReorder(xg, ro, bl, cj));          /* -> [s0][s1][p0][p1][chan][2] */
        for (int c = 0; c < NCHAN; c++)
            for (int s0 = 0; s0 < NSTAND; s0++)
                for (int s1 = s0; s1 < NSTAND; s1++)
                    for (int p = 0; p < NPOL * NPOL; p++) {
                        const double complex gv = gold[((((size_t)it * NCHAN + c) * NSTAND + s0) * NSTAND + s1) * NPOL * NPOL + p];
                        const int32_t *v = ro + ((((size_t)s0 * NSTAND + s1) * NPOL * NPOL + p) * NCHAN + c) * 2;
                        checked++;
                        if (v[0] != (int32_t)creal(gv) || v[1] != (int32_t)cimag(gv)) bad++;
                    }
    }
    CHECK(xengXgpuDestroy());
    CHECK(xengFree(din, XENG_SPACE_CUDA));
    CHECK(xengFree(dout, XENG_SPACE_CUDA));
    printf("%ld visibilities checked against the golden file, %ld mismatches: %s\n", checked, bad, bad ? "FAIL" : "PASS");
    return bad ? 1 : 0;
}
