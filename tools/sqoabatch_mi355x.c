/*
 * sqoabatch_mi355x.c -- directory front-end on the batch C-ABI: what sqoaconv.c
 * (sqoaconv.c:38-100) does for one file, for a whole directory in ONE plan per direction.
 *
 *   sqoabatch_mi355x encode <indir> <outdir> [--qoi]    every *.pam  -> *.sqoa (or *.qoi)
 *   sqoabatch_mi355x decode <indir> <outdir>            every *.sqoa / *.qoi -> *.pam (channels as stored)
 *
 * encode: all pixels go to the device in one buffer, one encode plan, sqoa_hip_pack() compacts the
 * worst-case slots, one D2H of the exact-size blob, then the files are written.  decode mirrors it.
 * The streams written are byte-identical to what sqoa_write (seqoia.h:336) writes file by file.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "seqoia.h"
#include "sqoa_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define CK(e) do { if (e) DIE("%s: %s", #e, sqoa_hip_last_error()); } while (0)

typedef struct { char name[256]; unsigned w, h, ch; size_t size, off; unsigned char *data; } item;

static int ends_with(const char *s, const char *e) { size_t a = strlen(s), b = strlen(e); return a >= b && !strcmp(s + a - b, e); }

static unsigned char *read_file(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *d = malloc(sz > 0 ? (size_t)sz : 1);
    if (d && fread(d, 1, (size_t)sz, f) != (size_t)sz) { free(d); d = NULL; }
    fclose(f);
    *n = (size_t)sz;
    return d;
}

static int parse_pam(item *it, unsigned char *file, size_t n) { /* P7 header, MAXVAL 255 */
    unsigned maxval = 0;
    char *p = (char *)file, *end = (char *)file + n, *hdr_end = memmem(file, n, "ENDHDR\n", 7);
    if (n < 3 || memcmp(file, "P7", 2) || !hdr_end) return 0;
    while (p < hdr_end) {
        sscanf(p, "WIDTH %u", &it->w); sscanf(p, "HEIGHT %u", &it->h); sscanf(p, "DEPTH %u", &it->ch); sscanf(p, "MAXVAL %u", &maxval);
        p = memchr(p, '\n', (size_t)(end - p));
        if (!p) return 0;
        p++;
    }
    it->data = (unsigned char *)hdr_end + 7;
    it->size = (size_t)it->w * it->h * it->ch;
    return maxval == 255 && it->ch >= 1 && it->ch <= 4 && it->w && it->h && it->data + it->size <= file + n;
}

int main(int argc, char **argv) {
    if (argc < 4 || (strcmp(argv[1], "encode") && strcmp(argv[1], "decode")))
        DIE("Usage: sqoabatch_mi355x encode|decode <indir> <outdir> [--qoi]");
    const int encode = !strcmp(argv[1], "encode"), qoi = argc > 4 && !strcmp(argv[4], "--qoi");
    DIR *dir = opendir(argv[2]);
    if (!dir) DIE("cannot open %s", argv[2]);
    item *items = NULL;
    unsigned char **files = NULL;
    int n = 0;
    size_t in_total = 0, out_total = 0;
    struct dirent *e;
    while ((e = readdir(dir)) != NULL) {
        if (encode ? !ends_with(e->d_name, ".pam") : !(ends_with(e->d_name, ".sqoa") || ends_with(e->d_name, ".qoi"))) continue;
        char path[4096];
        snprintf(path, sizeof path, "%s/%s", argv[2], e->d_name);
        size_t fn;
        unsigned char *file = read_file(path, &fn);
        if (!file) DIE("cannot read %s", path);
        items = realloc(items, (n + 1) * sizeof *items);
        files = realloc(files, (n + 1) * sizeof *files);
        item *it = &items[n];
        memset(it, 0, sizeof *it);
        snprintf(it->name, sizeof it->name, "%.*s", (int)(strrchr(e->d_name, '.') - e->d_name), e->d_name);
        if (encode) { if (!parse_pam(it, file, fn)) DIE("%s: not a P7/255 PAM", path); }
        else {
            if (fn < 22) DIE("%s: too short", path);
            it->data = file; it->size = fn;
            it->w = (unsigned)file[4] << 24 | file[5] << 16 | file[6] << 8 | file[7];
            it->h = (unsigned)file[8] << 24 | file[9] << 16 | file[10] << 8 | file[11];
            it->ch = (file[12] < 3 ? 1 : 3) + ((file[12] & 1) ? 0 : 1);
        }
        it->off = in_total;
        in_total += (it->size + 255) & ~(size_t)255;
        files[n++] = file;
    }
    closedir(dir);
    if (!n) DIE("no input files in %s", argv[2]);

    unsigned long long d_in, d_out;
    CK(sqoa_hip_malloc(0, in_total, &d_in));
    sqoa_hip_job *jobs = calloc(n, sizeof *jobs);
    size_t *out_off = calloc(n + 1, sizeof *out_off);
    for (int i = 0; i < n; i++) {
        CK(sqoa_hip_memcpy_h2d(0, d_in + items[i].off, items[i].data, items[i].size));
        sqoa_desc d = {.width = items[i].w, .height = items[i].h, .channels = (unsigned char)items[i].ch, .colorspace = SQOA_SRGB, .qoi_compat = (unsigned char)qoi};
        size_t need = encode ? sqoa_hip_encode_bound(&d) : (((size_t)items[i].w * items[i].h * items[i].ch + 255) & ~(size_t)255);
        out_off[i] = out_total;
        out_total += need;
        jobs[i].src = d_in + items[i].off;
        jobs[i].src_size = (unsigned)items[i].size;
        jobs[i].dst_cap = need > 0xffffffffu ? 0xffffffffu : (unsigned)need;
        jobs[i].desc = d;
        if (!encode) jobs[i].desc.channels = 0; /* sqoa_read(path, &desc, 0): channels as stored (sqoaconv.c:65) */
    }
    CK(sqoa_hip_malloc(0, out_total, &d_out));
    for (int i = 0; i < n; i++) jobs[i].dst = d_out + out_off[i];
    sqoa_hip_plan *plan;
    sqoa_hip_result *res = calloc(n, sizeof *res);
    CK(sqoa_hip_plan_create(0, encode ? SQOA_HIP_ENCODE : SQOA_HIP_DECODE, jobs, n, &plan));
    CK(sqoa_hip_plan_launch(plan, NULL));
    CK(sqoa_hip_plan_results(plan, res, n));

    unsigned char *host = NULL;
    unsigned long long *offs = calloc(n + 1, sizeof *offs);
    if (encode) { /* exact-size blob: one D2H */
        unsigned long long d_blob, d_offs, total = 0;
        size_t cap = 0;
        for (int i = 0; i < n; i++) cap += res[i].status ? 0 : (size_t)res[i].len;
        CK(sqoa_hip_malloc(0, cap ? cap : 1, &d_blob));
        CK(sqoa_hip_malloc(0, (n + 1) * 8, &d_offs));
        CK(sqoa_hip_pack(plan, d_blob, cap, d_offs, &total, NULL));
        host = malloc(total ? total : 1);
        CK(sqoa_hip_memcpy_d2h(0, host, d_blob, total));
        CK(sqoa_hip_memcpy_d2h(0, offs, d_offs, (n + 1) * 8));
    } else {
        host = malloc(out_total);
        CK(sqoa_hip_memcpy_d2h(0, host, d_out, out_total));
        for (int i = 0; i <= n; i++) offs[i] = i < n ? out_off[i] : out_total;
    }
    int failed = 0;
    for (int i = 0; i < n; i++) {
        if (res[i].status) { fprintf(stderr, "%s: refused (status %d)\n", items[i].name, res[i].status); failed++; continue; }
        char path[4096];
        snprintf(path, sizeof path, "%s/%s.%s", argv[3], items[i].name, encode ? (qoi ? "qoi" : "sqoa") : "pam");
        FILE *f = fopen(path, "wb");
        if (!f) DIE("cannot write %s", path);
        if (!encode) {
            static const char *tupl[5] = {"", "GRAYSCALE", "GRAYSCALE_ALPHA", "RGB", "RGB_ALPHA"};
            fprintf(f, "P7\nWIDTH %u\nHEIGHT %u\nDEPTH %u\nMAXVAL 255\nTUPLTYPE %s\nENDHDR\n", items[i].w, items[i].h, items[i].ch, tupl[items[i].ch]);
        }
        fwrite(host + offs[i], 1, (size_t)res[i].len, f);
        fclose(f);
    }
    printf("%s: %d files, %zu bytes in, %d failed\n", argv[1], n, in_total, failed);
    return failed != 0;
}
