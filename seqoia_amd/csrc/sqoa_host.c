/*
 * sqoa_host.c -- the four legacy entry points of seqoia.h, in plain C.
 *
 * This is the drop-in boundary: same signatures, same malloc()/free()
 * ownership, same NULL / 0 failure rules as the reference (seqoia.h:336, :350,
 * :363, :374; definitions :456-650, :652-809, :814-866).  Everything per-pixel
 * happens on the GPU through the batch layer (sqoa_hip.h) with a batch of one:
 * copy in, one launch, copy the exact-size result out.  That round trip is
 * PCIe-bound and exists for drop-in correctness; throughput work uses plans on
 * device-resident buffers.
 *
 * There is NO CPU fallback: without a usable gfx950 device every entry point
 * fails (NULL / 0), silently like the reference; SQOA_HIP_VERBOSE=1 makes the
 * first failure say why on stderr.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/seqoia.h"
#include "../../include/sqoa_hip.h"

#define SQOA_PIXELS_MAX 400000000u /* reference seqoia.h:432 */

/* Per-thread device context: the reference is re-entrant (only a const table
 * at seqoia.h:439), so concurrent callers must not share mutable state. */
typedef struct {
    int device;
    sqoa_hip_plan *plan[2];
    sqoa_hip_job last[2][32]; /* the job table each plan holds: a caller that repeats a call (sqoabench does) skips the upload */
    int last_n[2];
    unsigned long long d_in, d_out;
    size_t cap_in, cap_out;
} host_ctx;

static pthread_key_t g_key;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void ctx_free(void *p) {
    host_ctx *c = p;
    if (!c) return;
    for (int k = 0; k < 2; k++)
        if (c->plan[k]) sqoa_hip_plan_destroy(c->plan[k]);
    if (c->d_in) sqoa_hip_free(c->device, c->d_in);
    if (c->d_out) sqoa_hip_free(c->device, c->d_out);
    free(c);
}

static void key_init(void) { pthread_key_create(&g_key, ctx_free); }

/* The reference library never prints (failure is NULL / 0, seqoia.h:333-334, :343-344, :357-358, :368-369) and neither
 * does this one unless SQOA_HIP_VERBOSE is set in the environment: then the first failure says why, once. */
static void complain(const char *what) {
    static int said = 0;
    const char *v = getenv("SQOA_HIP_VERBOSE");
    if (v && *v && *v != '0' && !__sync_lock_test_and_set(&said, 1))
        fprintf(stderr, "libsqoa_mi355x: %s: %s (no CPU fallback)\n", what, sqoa_hip_last_error());
}

static host_ctx *ctx_get(void) {
    pthread_once(&g_once, key_init);
    host_ctx *c = pthread_getspecific(g_key);
    if (c) return c;
    if (sqoa_hip_device_count() <= 0) {
        complain("no HIP device");
        return NULL;
    }
    c = calloc(1, sizeof *c);
    if (!c) return NULL;
    const char *dev = getenv("SQOA_HIP_DEVICE");
    c->device = dev ? atoi(dev) : 0;
    pthread_setspecific(g_key, c);
    return c;
}

static int grow(host_ctx *c, unsigned long long *d, size_t *cap, size_t need) {
    if (*cap >= need) return 0;
    if (*d) sqoa_hip_free(c->device, *d);
    *d = 0;
    *cap = 0;
    need = (need + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    if (sqoa_hip_malloc(c->device, need, d)) return -1;
    *cap = need;
    return 0;
}

/* n jobs through the thread's plan of the given kind (the plan is created for SQOA_STRIPS_MAX jobs once) */
static int run_jobs(host_ctx *c, int kind, const sqoa_hip_job *jobs, int n, sqoa_hip_result *res) {
    if (!c->plan[kind]) {
        sqoa_hip_job first[32];
        memset(first, 0, sizeof first);
        memcpy(first, jobs, (size_t)n * sizeof *jobs);
        if (sqoa_hip_plan_create(c->device, kind, first, 32, &c->plan[kind])) { complain("plan"); return -1; }
    }
    if (c->last_n[kind] != n || memcmp(c->last[kind], jobs, (size_t)n * sizeof *jobs)) {
        if (sqoa_hip_plan_update(c->plan[kind], jobs, n)) { c->last_n[kind] = 0; return -1; }
        memcpy(c->last[kind], jobs, (size_t)n * sizeof *jobs);
        c->last_n[kind] = n;
    }
    if (sqoa_hip_plan_launch(c->plan[kind], NULL)) { complain("launch"); return -1; }
    if (sqoa_hip_plan_results(c->plan[kind], res, n)) { complain("results"); return -1; }
    return 0;
}

/* A large SQOA-native image goes to the GPU as several strip jobs (sqoa_hip.h), i.e. several workgroups: one image is
 * otherwise one workgroup on one compute unit (0.57 ms per Mpixel however idle the other 255 are).  A strip must start at a
 * pixel that differs from its predecessor, so that no run (seqoia.h:544-561) crosses the cut: from each nominal position the
 * cut moves forward to the first such pixel. */
#define SQOA_STRIPS_MAX 32
#define SQOA_STRIP_MIN_PX 32768u

static unsigned plan_strips(const unsigned char *px, size_t npx, unsigned stride, size_t *cut /* [SQOA_STRIPS_MAX + 1] */) {
    unsigned want = (unsigned)(npx / SQOA_STRIP_MIN_PX);
    if (want > SQOA_STRIPS_MAX) want = SQOA_STRIPS_MAX;
    unsigned n = 0;
    cut[n++] = 0;
    for (unsigned k = 1; k < want; k++) {
        size_t i = npx * k / want;
        if (i <= cut[n - 1]) i = cut[n - 1] + 1;
        while (i < npx && !memcmp(px + i * stride, px + (i - 1) * stride, stride)) i++;
        if (i >= npx) break; /* the rest of the image is one run */
        cut[n++] = i;
    }
    cut[n] = npx;
    return n;
}

void *sqoa_encode(const void *data, const sqoa_desc *desc, int *out_len) {
    /* reference seqoia.h:465-480 */
    if (!data || !out_len || !desc || desc->width == 0 || desc->height == 0 || desc->channels < 1 ||
        desc->channels > 6 || desc->colorspace > 1 || desc->height >= SQOA_PIXELS_MAX / desc->width)
        return NULL;
    if (desc->channels < 3 && desc->qoi_compat) return NULL;
    host_ctx *c = ctx_get();
    if (!c) return NULL;

    const unsigned stride = (desc->channels < 3 ? 1u : 3u) + ((desc->channels & 1) ? 0u : 1u);
    const size_t npx = (size_t)desc->width * desc->height;
    const size_t in_bytes = npx * stride;
    size_t cut[SQOA_STRIPS_MAX + 1];
    unsigned nstrips = 1;
    cut[0] = 0;
    cut[1] = npx;
    if (!desc->qoi_compat && npx >= 2 * (size_t)SQOA_STRIP_MIN_PX) nstrips = plan_strips(data, npx, stride, cut);

    sqoa_hip_job jobs[SQOA_STRIPS_MAX];
    size_t off[SQOA_STRIPS_MAX + 1];
    memset(jobs, 0, sizeof jobs);
    off[0] = 0;
    for (unsigned k = 0; k < nstrips; k++) {
        sqoa_desc d = *desc;
        if (nstrips > 1) { d.width = (unsigned)(cut[k + 1] - cut[k]); d.height = 1; }
        const size_t bound = sqoa_hip_encode_bound(&d);
        jobs[k].dst_cap = bound > 0xffffffffu ? 0xffffffffu : (unsigned)bound;
        jobs[k].desc = d;
        if (nstrips > 1)
            jobs[k].reserved = SQOA_HIP_JOB_NOHEADER | (k ? SQOA_HIP_JOB_PREV : 0u) | (k + 1 < nstrips ? SQOA_HIP_JOB_NOEND : 0u);
        off[k + 1] = off[k] + bound;
    }
    /* device buffers: pixels | one worst-case slot per strip | (strips only) the packed stream and its offsets */
    const size_t blob_at = off[nstrips], offs_at = blob_at + ((off[nstrips] + 255) & ~(size_t)255);
    if (grow(c, &c->d_in, &c->cap_in, in_bytes) || grow(c, &c->d_out, &c->cap_out, offs_at + 8 * (SQOA_STRIPS_MAX + 1))) return NULL;
    for (unsigned k = 0; k < nstrips; k++) {
        jobs[k].src = c->d_in + cut[k] * stride;
        jobs[k].dst = c->d_out + off[k];
    }
    if (sqoa_hip_memcpy_h2d(c->device, c->d_in, data, in_bytes)) return NULL;
    sqoa_hip_result res[SQOA_STRIPS_MAX];
    if (run_jobs(c, SQOA_HIP_ENCODE, jobs, (int)nstrips, res)) return NULL;
    size_t total = nstrips > 1 ? 15 : 0; /* magic, width, height, channels, colorspace, start byte */
    for (unsigned k = 0; k < nstrips; k++) {
        if (res[k].status != SQOA_HIP_ST_OK || res[k].len < 0) return NULL;
        total += (size_t)res[k].len;
    }
    if (total > 0x7fffffffu) return NULL;

    unsigned char *out = malloc(total); /* caller free()s: sqoabench.c:537, seqoia.h:834 */
    if (!out) return NULL;
    if (nstrips == 1) {
        if (sqoa_hip_memcpy_d2h(c->device, out, c->d_out, total)) { free(out); return NULL; }
    } else {
        /* header as seqoia.h:497-514 writes it; the strips' streams, compacted on the device, behind it */
        static const unsigned char magic[4] = {'S', 'q', 'o', 'a'};
        memcpy(out, magic, 4);
        for (int k = 0; k < 4; k++) { out[4 + k] = (unsigned char)(desc->width >> (24 - 8 * k)); out[8 + k] = (unsigned char)(desc->height >> (24 - 8 * k)); }
        out[12] = (unsigned char)stride;
        out[13] = desc->colorspace;
        out[14] = 0x31;
        unsigned long long packed = 0;
        if (sqoa_hip_pack(c->plan[SQOA_HIP_ENCODE], c->d_out + blob_at, total - 15, c->d_out + offs_at, &packed, NULL) || packed != total - 15 ||
            sqoa_hip_memcpy_d2h(c->device, out + 15, c->d_out + blob_at, total - 15)) { free(out); return NULL; }
    }
    *out_len = (int)total;
    return out;
}

static unsigned be32(const unsigned char *b) {
    return (unsigned)b[0] << 24 | (unsigned)b[1] << 16 | (unsigned)b[2] << 8 | b[3];
}

void *sqoa_decode(const void *data, int size, sqoa_desc *desc, int channels) {
    /* reference seqoia.h:662-668; channels < 0 is undefined there, NULL here */
    if (!data || !desc || channels > 4 || channels < 0 || size < 14 + 8) return NULL;
    const unsigned char *b = data;
    const unsigned magic = be32(b);
    /* the reference fills *desc before it validates (seqoia.h:672-677) */
    desc->width = be32(b + 4);
    desc->height = be32(b + 8);
    desc->channels = b[12];
    desc->colorspace = b[13];
    desc->qoi_compat = b[14] != 0x31;
    /* seqoia.h:679-688 */
    if (desc->width == 0 || desc->height == 0 || desc->channels < 1 || desc->channels > 6 || desc->colorspace > 1 ||
        !(magic == 0x716f6966u || magic == 0x53716f61u) || (magic == 0x716f6966u && !desc->qoi_compat) ||
        desc->height >= SQOA_PIXELS_MAX / desc->width)
        return NULL;
    const int col = desc->channels < 3 ? 1 : 3;
    const int out_ch = channels ? channels : col + ((desc->channels & 1) ? 0 : 1); /* seqoia.h:699-702 */

    host_ctx *c = ctx_get();
    if (!c) return NULL;
    const size_t out_bytes = (size_t)desc->width * desc->height * out_ch;
    if (grow(c, &c->d_in, &c->cap_in, (size_t)size + 16) || grow(c, &c->d_out, &c->cap_out, out_bytes)) return NULL;
    if (sqoa_hip_memcpy_h2d(c->device, c->d_in, data, (size_t)size)) return NULL;

    sqoa_hip_job job;
    memset(&job, 0, sizeof job);
    job.src = c->d_in;
    job.dst = c->d_out;
    job.src_size = (unsigned)size;
    job.dst_cap = out_bytes > 0xffffffffu ? 0xffffffffu : (unsigned)out_bytes;
    job.desc.width = desc->width;
    job.desc.height = desc->height;
    job.desc.channels = (unsigned char)channels;
    sqoa_hip_result res;
    if (run_jobs(c, SQOA_HIP_DECODE, &job, 1, &res) || res.status != SQOA_HIP_ST_OK || (size_t)res.len != out_bytes) return NULL;

    void *out = malloc(out_bytes); /* caller free()s: sqoabench.c:454, :493 */
    if (!out) return NULL;
    if (sqoa_hip_memcpy_d2h(c->device, out, c->d_out, out_bytes)) { free(out); return NULL; }
    return out;
}

#ifndef SQOA_NO_STDIO

int sqoa_write(const char *filename, const void *data, const sqoa_desc *desc) { /* seqoia.h:814-836 */
    FILE *f = fopen(filename, "wb");
    if (!f) return 0;
    int size = 0;
    void *enc = sqoa_encode(data, desc, &size);
    if (!enc) { fclose(f); return 0; }
    fwrite(enc, 1, (size_t)size, f);
    fflush(f);
    const int err = ferror(f);
    fclose(f);
    free(enc);
    return err ? 0 : size;
}

void *sqoa_read(const char *filename, sqoa_desc *desc, int channels) { /* seqoia.h:838-866 */
    FILE *f = fopen(filename, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    if (size <= 0 || size > 0x7fffffffL || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return NULL; }
    void *data = malloc((size_t)size);
    if (!data) { fclose(f); return NULL; }
    const size_t got = fread(data, 1, (size_t)size, f);
    fclose(f);
    void *px = got == (size_t)size ? sqoa_decode(data, (int)size, desc, channels) : NULL;
    free(data);
    return px;
}

#endif /* SQOA_NO_STDIO */
