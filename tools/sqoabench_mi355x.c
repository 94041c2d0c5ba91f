/*
 * sqoabench_mi355x.c -- counterpart of the reference's sqoabench.c for libsqoa_mi355x.so.
 *
 * Same protocol and columns as sqoabench.c: BENCHMARK_FN (sqoabench.c:394-406: run 0 is a
 * discarded warm-up, mean of <runs>, the library's malloc and the caller's free inside the timed
 * region), round-trip memcmp verify (sqoabench.c:448-455), table columns
 * "decode ms / encode ms / decode mpps / encode mpps / size kb / rate" (sqoabench.c:331-390), flags
 * --nowarmup --noverify --noencode --nodecode --onlytotals (sqoabench.c:640-665).  The PNG /
 * stb / libpng / qoi.h rows are dropped (un-vendored third-party code); inputs are .sqoa / .qoi
 * files of a directory, or the synthetic classes of SURVEY.md Appendix C (--synth).
 *
 * Two rows are printed: "sqoa" = the legacy per-image calls on host memory (PCIe inclusive, what a
 * relinked sqoabench measures) and "sqoa-batch" = the batch C-ABI on device-resident buffers.
 *
 *   sqoabench_mi355x <runs> <directory>|--synth <class> <w> <h> <channels> <count> [options]
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define SQOA_IMPLEMENTATION /* accepted and ignored: the symbols come from the shared library */
#include "seqoia.h"
#include "sqoa_hip.h"

static uint64_t ns(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec;
}

#define ERROR(...) do { fprintf(stderr, "abort at line %d: ", __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

#define BENCHMARK_FN(NOWARMUP, RUNS, AVG_TIME, ...)             \
    do {                                                        \
        uint64_t time = 0;                                      \
        for (int i_ = NOWARMUP; i_ <= RUNS; i_++) {             \
            uint64_t t0_ = ns();                                \
            __VA_ARGS__                                         \
            uint64_t t1_ = ns();                                \
            if (i_ > 0) time += t1_ - t0_;                      \
        }                                                       \
        AVG_TIME = time / RUNS;                                 \
    } while (0)

static int opt_runs = 1, opt_nowarmup = 0, opt_noverify = 0, opt_noencode = 0, opt_nodecode = 0, opt_onlytotals = 0;

typedef struct { uint64_t size, encode_time, decode_time; } lib_res;
typedef struct { int count; uint64_t px, raw_size; lib_res legacy, batch; } bench_res;

static uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

static void print_res(const char *head, bench_res r) {
    if (!r.count) return;
    double px = (double)r.px / r.count;
    printf("## %s\n          decode ms   encode ms   decode mpps   encode mpps   size kb    rate\n", head);
    const char *names[2] = {"sqoa:      ", "sqoa-batch:"};
    lib_res l[2] = {r.legacy, r.batch};
    for (int k = 0; k < 2; k++) {
        double dt = (double)l[k].decode_time / r.count, et = (double)l[k].encode_time / r.count;
        printf("%s %8.3f  %10.3f      %8.2f      %8.2f %9llu   %4.1f%%\n", names[k], dt / 1e6, et / 1e6, dt > 0 ? px / (dt / 1000.0) : 0,
               et > 0 ? px / (et / 1000.0) : 0, (unsigned long long)(l[k].size / r.count / 1024),
               100.0 * (double)l[k].size / (double)r.raw_size);
    }
    printf("\n");
}

/* one image through the legacy API, sqoabench.c:409-547 minus the third-party rows */
static void bench_legacy(const void *pixels, int w, int h, int channels, bench_res *res) {
    int enc_size = 0;
    void *enc = sqoa_encode(pixels, &(sqoa_desc){.width = w, .height = h, .channels = channels, .colorspace = SQOA_SRGB, .qoi_compat = 0}, &enc_size);
    if (!enc) ERROR("Error encoding %dx%dx%d (is a GPU present?)", w, h, channels);
    if (!opt_noverify) {
        sqoa_desc dc;
        void *back = sqoa_decode(enc, enc_size, &dc, channels);
        if (!back || memcmp(pixels, back, (size_t)w * h * channels) != 0) ERROR("SQOA roundtrip pixel mismatch");
        free(back);
    }
    res->legacy.size += enc_size;
    if (!opt_nodecode) {
        uint64_t t;
        BENCHMARK_FN(opt_nowarmup, opt_runs, t, { sqoa_desc d; void *p = sqoa_decode(enc, enc_size, &d, 4); free(p); });
        res->legacy.decode_time += t;
    }
    if (!opt_noencode) {
        uint64_t t;
        BENCHMARK_FN(opt_nowarmup, opt_runs, t, { int n; void *p = sqoa_encode(pixels, &(sqoa_desc){.width = w, .height = h, .channels = channels, .colorspace = SQOA_SRGB, .qoi_compat = 0}, &n); free(p); });
        res->legacy.encode_time += t;
    }
    free(enc);
}

/* n same-shaped images through the batch API, buffers resident on the device */
static void bench_batch(unsigned long long d_px, int n, int w, int h, int channels, bench_res *res) {
    sqoa_desc desc = {.width = w, .height = h, .channels = channels, .colorspace = SQOA_SRGB, .qoi_compat = 0};
    size_t slot = sqoa_hip_encode_bound(&desc), img = (size_t)w * h * channels, out4 = (size_t)w * h * 4;
    unsigned long long d_enc, d_dec;
    if (sqoa_hip_malloc(0, slot * n, &d_enc) || sqoa_hip_malloc(0, out4 * n, &d_dec)) ERROR("device alloc: %s", sqoa_hip_last_error());
    sqoa_hip_job *jobs = calloc(n, sizeof *jobs);
    sqoa_hip_result *r = calloc(n, sizeof *r);
    for (int i = 0; i < n; i++) jobs[i] = (sqoa_hip_job){.src = d_px + i * img, .dst = d_enc + i * slot, .dst_cap = (unsigned)slot, .desc = desc};
    sqoa_hip_plan *pe, *pd;
    if (sqoa_hip_plan_create(0, SQOA_HIP_ENCODE, jobs, n, &pe) || sqoa_hip_plan_launch(pe, NULL) || sqoa_hip_plan_results(pe, r, n)) ERROR("%s", sqoa_hip_last_error());
    for (int i = 0; i < n; i++) {
        if (r[i].status) ERROR("batch encode status %d", r[i].status);
        res->batch.size += r[i].len;
        jobs[i] = (sqoa_hip_job){.src = d_enc + i * slot, .dst = d_dec + i * out4, .src_size = r[i].len, .dst_cap = (unsigned)out4, .desc = {.width = w, .height = h, .channels = 4}};
    }
    if (sqoa_hip_plan_create(0, SQOA_HIP_DECODE, jobs, n, &pd) || sqoa_hip_plan_launch(pd, NULL) || sqoa_hip_plan_results(pd, r, n)) ERROR("%s", sqoa_hip_last_error());
    for (int i = 0; i < n; i++) if (r[i].status) ERROR("batch decode status %d", r[i].status);
    float ms;
    if (!opt_noencode) { sqoa_hip_plan_time(pe, opt_runs, &ms); res->batch.encode_time += (uint64_t)(ms * 1e6 / opt_runs); }
    if (!opt_nodecode) { sqoa_hip_plan_time(pd, opt_runs, &ms); res->batch.decode_time += (uint64_t)(ms * 1e6 / opt_runs); }
    sqoa_hip_plan_destroy(pe); sqoa_hip_plan_destroy(pd);
    sqoa_hip_free(0, d_enc); sqoa_hip_free(0, d_dec);
    free(jobs); free(r);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        puts("Usage: sqoabench_mi355x <iterations> <directory with .sqoa/.qoi files> [options]");
        puts("       sqoabench_mi355x <iterations> --synth <mixed|icon|photo> <w> <h> <channels> <count> [options]");
        puts("Options:\n    --nowarmup ... don't perform a warmup run\n    --noverify ... don't verify sqoa roundtrip");
        puts("    --noencode ... don't run encoders\n    --nodecode ... don't run decoders\n    --onlytotals . don't print individual image results");
        return 1;
    }
    opt_runs = atoi(argv[1]);
    if (opt_runs <= 0) ERROR("Invalid number of runs %d", opt_runs);
    int synth = strcmp(argv[2], "--synth") == 0, first_opt = synth ? 8 : 3;
    if (synth && argc < 8) ERROR("--synth needs <class> <w> <h> <channels> <count>");
    for (int i = first_opt; i < argc; i++) {
        if (!strcmp(argv[i], "--nowarmup")) opt_nowarmup = 1;
        else if (!strcmp(argv[i], "--noverify")) opt_noverify = 1;
        else if (!strcmp(argv[i], "--noencode")) opt_noencode = 1;
        else if (!strcmp(argv[i], "--nodecode")) opt_nodecode = 1;
        else if (!strcmp(argv[i], "--onlytotals")) opt_onlytotals = 1;
        else ERROR("Unknown option %s", argv[i]);
    }
    bench_res total = {0};
    if (synth) {
        const char *names[3] = {"mixed", "icon", "photo"};
        int cls = -1, w = atoi(argv[4]), h = atoi(argv[5]), ch = atoi(argv[6]), n = atoi(argv[7]);
        for (int k = 0; k < 3; k++) if (!strcmp(argv[3], names[k])) cls = k;
        if (cls < 0 || w <= 0 || h <= 0 || (ch != 3 && ch != 4) || n <= 0) ERROR("bad --synth arguments");
        size_t img = (size_t)w * h * ch;
        unsigned long long d_px;
        if (sqoa_hip_malloc(0, img * n, &d_px) || sqoa_hip_synth(0, d_px, img, w, h, ch, cls, 1, 0, n)) ERROR("synth: %s", sqoa_hip_last_error());
        unsigned char *host = malloc(img);
        int legacy_n = n < 8 ? n : 8; /* the per-image PCIe path is slow: sample it */
        for (int i = 0; i < legacy_n; i++) {
            sqoa_hip_memcpy_d2h(0, host, d_px + i * img, img);
            bench_legacy(host, w, h, ch, &total);
        }
        /* scale the legacy sample to the same count so both rows average per image */
        total.legacy.size = total.legacy.size * n / legacy_n;
        total.legacy.encode_time = total.legacy.encode_time * n / legacy_n;
        total.legacy.decode_time = total.legacy.decode_time * n / legacy_n;
        bench_batch(d_px, n, w, h, ch, &total);
        total.count = n; total.px = (uint64_t)w * h * n; total.raw_size = img * n;
        free(host); sqoa_hip_free(0, d_px);
        (void)mix32;
    } else {
        DIR *dir = opendir(argv[2]);
        if (!dir) ERROR("Couldn't open directory %s", argv[2]);
        struct dirent *f;
        while ((f = readdir(dir)) != NULL) {
            size_t L = strlen(f->d_name);
            if (L < 5 || (strcmp(f->d_name + L - 5, ".sqoa") && strcmp(f->d_name + L - 4, ".qoi"))) continue;
            char path[4096];
            snprintf(path, sizeof path, "%s/%s", argv[2], f->d_name);
            sqoa_desc d;
            void *px = sqoa_read(path, &d, 0);
            if (!px) ERROR("Error decoding %s", path);
            int ch = (d.channels < 3 ? 1 : 3) + ((d.channels & 1) ? 0 : 1);
            bench_res r = {0};
            bench_legacy(px, d.width, d.height, ch, &r);
            unsigned long long d_px;
            size_t img = (size_t)d.width * d.height * ch;
            if (ch >= 3 && !sqoa_hip_malloc(0, img, &d_px)) {
                sqoa_hip_memcpy_h2d(0, d_px, px, img);
                bench_batch(d_px, 1, d.width, d.height, ch, &r);
                sqoa_hip_free(0, d_px);
            }
            r.count = 1; r.px = (uint64_t)d.width * d.height; r.raw_size = img;
            if (!opt_onlytotals) { char head[4200]; snprintf(head, sizeof head, "%s size: %ux%u", path, d.width, d.height); print_res(head, r); }
            total.count++; total.px += r.px; total.raw_size += r.raw_size;
            total.legacy.size += r.legacy.size; total.legacy.encode_time += r.legacy.encode_time; total.legacy.decode_time += r.legacy.decode_time;
            total.batch.size += r.batch.size; total.batch.encode_time += r.batch.encode_time; total.batch.decode_time += r.batch.decode_time;
            free(px);
        }
        closedir(dir);
    }
    char head[256];
    snprintf(head, sizeof head, "Total for %s (%d images, per-image averages)", synth ? argv[3] : argv[2], total.count);
    print_res(head, total);
    return 0;
}
