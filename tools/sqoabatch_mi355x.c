/*
 * sqoabatch_mi355x.c -- directory front-end on the batch C-ABI: what sqoaconv.c
 * (sqoaconv.c:38-100) does for one file, for a whole directory, with file I/O overlapped with the GPU.
 *
 *   sqoabatch_mi355x encode <indir> <outdir> [--qoi] [--batch-mb N]    every *.pam  -> *.sqoa (or *.qoi)
 *   sqoabatch_mi355x decode <indir> <outdir> [--batch-mb N]            every *.sqoa / *.qoi -> *.pam (channels as stored)
 *
 * Three stages run at once on sub-batches of the directory (about N MB of input each, default 256):
 *
 *   reader thread   fopen / fread of sub-batch k+1 into a PINNED host buffer (two of them, used in turn);
 *   this thread     sub-batch k: asynchronous copy pinned -> device, one plan, launch, results; encode: sqoa_hip_pack()
 *                   compacts the worst-case slots; asynchronous copy device -> pinned output buffer (two of them);
 *   writer thread   fopen / fwrite of sub-batch k-1 out of its pinned buffer.
 *
 * The streams written are byte-identical to what sqoa_write (seqoia.h:336) writes file by file (sqoaconv.c:83-91), the
 * pixels to what sqoa_read(path, &desc, 0) returns (sqoaconv.c:63-69).
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "seqoia.h"
#include "sqoa_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define CK(e) do { if (e) DIE("%s: %s", #e, sqoa_hip_last_error()); } while (0)

typedef struct { char name[256]; size_t fsize; } entry;                                  /* one directory entry */
typedef struct { unsigned w, h, ch; size_t size, off, hdr; int ok; } item;               /* one image of a sub-batch: payload at in + off + hdr */
typedef struct { int first, n; item *it; size_t in_bytes; unsigned char *in; size_t in_cap; } inbuf;
typedef struct { int first, n; item *it; size_t *off; int *len, *status; unsigned char *out; size_t out_cap; } outbuf;

static const char *g_indir, *g_outdir;
static int g_encode, g_qoi, g_nfiles, g_nbatches;
static entry *g_files;
static int *g_bfirst; /* sub-batch k = files g_bfirst[k] .. g_bfirst[k+1]-1 */

/* a one-slot-per-buffer hand-off: `ready[k & 1]` says which sub-batch buffer k & 1 currently holds (-1: free) */
typedef struct { pthread_mutex_t mu; pthread_cond_t cv; int holds[2]; } handoff;
static handoff g_in = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, {-1, -1}}, g_out = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, {-1, -1}};
static inbuf g_inb[2];
static outbuf g_outb[2];

static void wait_for(handoff *h, int slot, int want) {
    pthread_mutex_lock(&h->mu);
    while (h->holds[slot] != want) pthread_cond_wait(&h->cv, &h->mu);
    pthread_mutex_unlock(&h->mu);
}
static void set_to(handoff *h, int slot, int v) {
    pthread_mutex_lock(&h->mu);
    h->holds[slot] = v;
    pthread_cond_broadcast(&h->cv);
    pthread_mutex_unlock(&h->mu);
}

static int ends_with(const char *s, const char *e) { size_t a = strlen(s), b = strlen(e); return a >= b && !strcmp(s + a - b, e); }

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
    it->hdr = (size_t)((unsigned char *)hdr_end + 7 - file);
    it->size = (size_t)it->w * it->h * it->ch;
    return maxval == 255 && it->ch >= 1 && it->ch <= 4 && it->w && it->h && it->hdr + it->size <= n;
}

/* reader: files of sub-batch k straight into pinned buffer k & 1, every file at a 256-byte aligned offset */
static void *reader_main(void *arg) {
    (void)arg;
    for (int k = 0; k < g_nbatches; k++) {
        inbuf *b = &g_inb[k & 1];
        wait_for(&g_in, k & 1, -1);
        b->first = g_bfirst[k];
        b->n = g_bfirst[k + 1] - g_bfirst[k];
        size_t off = 0;
        for (int i = 0; i < b->n; i++) {
            const entry *e = &g_files[b->first + i];
            item *it = &b->it[i];
            memset(it, 0, sizeof *it);
            char path[4096];
            snprintf(path, sizeof path, "%s/%s", g_indir, e->name);
            FILE *f = fopen(path, "rb");
            it->off = off;
            size_t got = f ? fread(b->in + off, 1, e->fsize, f) : 0;
            if (f) fclose(f);
            unsigned char *file = b->in + off;
            if (got != e->fsize) it->ok = 0;
            else if (g_encode) it->ok = parse_pam(it, file, got);
            else if (got < 22) it->ok = 0;
            else {
                it->ok = 1; it->hdr = 0; it->size = got;
                it->w = (unsigned)file[4] << 24 | file[5] << 16 | file[6] << 8 | file[7];
                it->h = (unsigned)file[8] << 24 | file[9] << 16 | file[10] << 8 | file[11];
                it->ch = (file[12] < 3 ? 1 : 3) + ((file[12] & 1) ? 0 : 1);
            }
            if (!it->ok) fprintf(stderr, "%s: unreadable or not a %s\n", path, g_encode ? "P7/255 PAM" : "stream");
            off += (e->fsize + 16 + 255) & ~(size_t)255;
        }
        b->in_bytes = off;
        set_to(&g_in, k & 1, k);
    }
    return NULL;
}

static int g_failed;

static void *writer_main(void *arg) {
    (void)arg;
    for (int k = 0; k < g_nbatches; k++) {
        outbuf *b = &g_outb[k & 1];
        wait_for(&g_out, k & 1, k);
        for (int i = 0; i < b->n; i++) {
            const entry *e = &g_files[b->first + i];
            if (b->status[i]) { fprintf(stderr, "%s: refused (status %d)\n", e->name, b->status[i]); g_failed++; continue; }
            char path[4096];
            snprintf(path, sizeof path, "%s/%.*s.%s", g_outdir, (int)(strrchr(e->name, '.') - e->name), e->name, g_encode ? (g_qoi ? "qoi" : "sqoa") : "pam");
            FILE *f = fopen(path, "wb");
            if (!f) DIE("cannot write %s", path);
            if (!g_encode) {
                static const char *tupl[5] = {"", "GRAYSCALE", "GRAYSCALE_ALPHA", "RGB", "RGB_ALPHA"};
                fprintf(f, "P7\nWIDTH %u\nHEIGHT %u\nDEPTH %u\nMAXVAL 255\nTUPLTYPE %s\nENDHDR\n", b->it[i].w, b->it[i].h, b->it[i].ch, tupl[b->it[i].ch]);
            }
            fwrite(b->out + b->off[i], 1, (size_t)b->len[i], f);
            fclose(f);
        }
        set_to(&g_out, k & 1, -1);
    }
    return NULL;
}

static int by_name(const void *a, const void *b) { return strcmp(((const entry *)a)->name, ((const entry *)b)->name); }

int main(int argc, char **argv) {
    if (argc < 4 || (strcmp(argv[1], "encode") && strcmp(argv[1], "decode")))
        DIE("Usage: sqoabatch_mi355x encode|decode <indir> <outdir> [--qoi] [--batch-mb N]");
    g_encode = !strcmp(argv[1], "encode");
    g_indir = argv[2];
    g_outdir = argv[3];
    size_t batch_bytes = (size_t)256 << 20;
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--qoi")) g_qoi = 1;
        else if (!strcmp(argv[i], "--batch-mb") && i + 1 < argc) batch_bytes = (size_t)atol(argv[++i]) << 20;
        else DIE("unknown option %s", argv[i]);
    }
    if (batch_bytes < ((size_t)1 << 20)) batch_bytes = (size_t)1 << 20;

    /* the directory: names and sizes only */
    DIR *dir = opendir(g_indir);
    if (!dir) DIE("cannot open %s", g_indir);
    struct dirent *e;
    while ((e = readdir(dir)) != NULL) {
        if (g_encode ? !ends_with(e->d_name, ".pam") : !(ends_with(e->d_name, ".sqoa") || ends_with(e->d_name, ".qoi"))) continue;
        char path[4096];
        snprintf(path, sizeof path, "%s/%s", g_indir, e->d_name);
        FILE *f = fopen(path, "rb");
        if (!f) DIE("cannot read %s", path);
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fclose(f);
        g_files = realloc(g_files, (size_t)(g_nfiles + 1) * sizeof *g_files);
        snprintf(g_files[g_nfiles].name, sizeof g_files[g_nfiles].name, "%s", e->d_name);
        g_files[g_nfiles++].fsize = sz > 0 ? (size_t)sz : 0;
    }
    closedir(dir);
    if (!g_nfiles) DIE("no input files in %s", g_indir);
    qsort(g_files, (size_t)g_nfiles, sizeof *g_files, by_name);

    /* sub-batches of about batch_bytes of input; buffers sized for the largest one */
    g_bfirst = calloc((size_t)g_nfiles + 2, sizeof *g_bfirst);
    size_t in_cap = 0, acc = 0;
    int max_n = 0;
    for (int i = 0; i < g_nfiles; i++) {
        const size_t padded = (g_files[i].fsize + 16 + 255) & ~(size_t)255;
        if (acc && acc + padded > batch_bytes) { g_bfirst[++g_nbatches] = i; acc = 0; }
        acc += padded;
        if (acc > in_cap) in_cap = acc;
        if (i + 1 - g_bfirst[g_nbatches] > max_n) max_n = i + 1 - g_bfirst[g_nbatches];
    }
    g_bfirst[++g_nbatches] = g_nfiles;
    for (int s = 0; s < 2; s++) {
        CK(sqoa_hip_host_alloc(in_cap, (void **)&g_inb[s].in));
        g_inb[s].in_cap = in_cap;
        g_inb[s].it = calloc((size_t)max_n, sizeof(item));
        g_outb[s].it = calloc((size_t)max_n, sizeof(item));
        g_outb[s].off = calloc((size_t)max_n + 1, sizeof(size_t));
        g_outb[s].len = calloc((size_t)max_n, sizeof(int));
        g_outb[s].status = calloc((size_t)max_n, sizeof(int));
    }

    pthread_t reader, writer;
    pthread_create(&reader, NULL, reader_main, NULL);
    pthread_create(&writer, NULL, writer_main, NULL);

    void *stream;
    CK(sqoa_hip_stream_create(0, &stream));
    unsigned long long d_in = 0, d_out = 0, d_blob = 0, d_offs = 0;
    size_t d_in_cap = 0, d_out_cap = 0, d_blob_cap = 0, d_offs_cap = 0;
    sqoa_hip_job *jobs = calloc((size_t)max_n, sizeof *jobs);
    sqoa_hip_result *res = calloc((size_t)max_n, sizeof *res);
    size_t *slot_off = calloc((size_t)max_n + 1, sizeof *slot_off);
    item *cur = calloc((size_t)max_n, sizeof *cur); /* this sub-batch's images (the reader reuses the input buffer's own table) */
    unsigned long long *offs = calloc((size_t)max_n + 1, sizeof *offs);
    size_t in_total = 0;
#define GROW(ptr, cap, need) do { if ((cap) < (need)) { if (ptr) CK(sqoa_hip_free(0, ptr)); CK(sqoa_hip_malloc(0, (need), &(ptr))); (cap) = (need); } } while (0)

    for (int k = 0; k < g_nbatches; k++) {
        inbuf *ib = &g_inb[k & 1];
        outbuf *ob = &g_outb[k & 1];
        wait_for(&g_in, k & 1, k);      /* the reader has filled it */
        const int n = ib->n, first = ib->first;
        memcpy(cur, ib->it, (size_t)n * sizeof(item));
        in_total += ib->in_bytes;
        GROW(d_in, d_in_cap, ib->in_bytes);
        CK(sqoa_hip_memcpy_h2d_async(0, d_in, ib->in, ib->in_bytes, stream));
        size_t out_total = 0;
        for (int i = 0; i < n; i++) {
            const item *it = &cur[i];
            sqoa_desc d = {.width = it->w, .height = it->h, .channels = (unsigned char)it->ch, .colorspace = SQOA_SRGB, .qoi_compat = (unsigned char)g_qoi};
            size_t need = !it->ok ? 256 : g_encode ? sqoa_hip_encode_bound(&d) : (((size_t)it->w * it->h * it->ch + 255) & ~(size_t)255);
            slot_off[i] = out_total;
            out_total += need;
            memset(&jobs[i], 0, sizeof jobs[i]);
            jobs[i].src = it->ok ? d_in + it->off + it->hdr : 0; /* a file that could not be read is a refused job */
            jobs[i].src_size = (unsigned)it->size;
            jobs[i].dst_cap = need > 0xffffffffu ? 0xffffffffu : (unsigned)need;
            jobs[i].desc = d;
            if (!g_encode) jobs[i].desc.channels = 0; /* sqoa_read(path, &desc, 0): channels as stored (sqoaconv.c:65) */
        }
        slot_off[n] = out_total;
        GROW(d_out, d_out_cap, out_total);
        for (int i = 0; i < n; i++) jobs[i].dst = d_out + slot_off[i];
        CK(sqoa_hip_stream_sync(0, stream)); /* the input is on the device: the reader may refill this buffer */
        set_to(&g_in, k & 1, -1);
        sqoa_hip_plan *plan;
        CK(sqoa_hip_plan_create(0, g_encode ? SQOA_HIP_ENCODE : SQOA_HIP_DECODE, jobs, n, &plan));
        CK(sqoa_hip_plan_launch(plan, stream));
        CK(sqoa_hip_plan_results(plan, res, n));

        wait_for(&g_out, k & 1, -1);    /* the writer is done with this output buffer */
        size_t need_host = 0;
        if (g_encode) { /* exact-size blob */
            size_t cap = 0;
            for (int i = 0; i < n; i++) cap += res[i].status ? 0 : (size_t)res[i].len;
            unsigned long long total = 0;
            GROW(d_blob, d_blob_cap, cap ? cap : 1);
            GROW(d_offs, d_offs_cap, ((size_t)n + 1) * 8);
            CK(sqoa_hip_pack(plan, d_blob, cap, d_offs, &total, stream));
            CK(sqoa_hip_memcpy_d2h(0, offs, d_offs, ((size_t)n + 1) * 8));
            need_host = (size_t)total;
        } else {
            need_host = out_total;
            for (int i = 0; i <= n; i++) offs[i] = slot_off[i];
        }
        if (ob->out_cap < need_host) {
            if (ob->out) CK(sqoa_hip_host_free(ob->out));
            ob->out_cap = need_host + need_host / 4 + 4096;
            CK(sqoa_hip_host_alloc(ob->out_cap, (void **)&ob->out));
        }
        if (need_host) CK(sqoa_hip_memcpy_d2h_async(0, ob->out, g_encode ? d_blob : d_out, need_host, stream));
        ob->first = first;
        ob->n = n;
        memcpy(ob->it, cur, (size_t)n * sizeof(item));
        for (int i = 0; i < n; i++) { ob->off[i] = (size_t)offs[i]; ob->len[i] = res[i].len; ob->status[i] = res[i].status; }
        CK(sqoa_hip_stream_sync(0, stream));
        sqoa_hip_plan_destroy(plan);
        set_to(&g_out, k & 1, k);       /* over to the writer */
    }
    pthread_join(reader, NULL);
    pthread_join(writer, NULL);
    printf("%s: %d files in %d sub-batch(es), %zu bytes in, %d failed\n", argv[1], g_nfiles, g_nbatches, in_total, g_failed);
    return g_failed != 0;
}
